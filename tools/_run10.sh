set -o pipefail
mkdir -p gpurun_out/r2g
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail 5 -k "wgrad" > gpurun_out/r2g/k_tests.log 2>&1; rc=$?; echo "kernel tests rc=$rc"; tail -2 gpurun_out/r2g/k_tests.log
[ $rc -eq 0 ] || exit 1
for l in 8 0; do echo "WGRAD_LW=$l"; GCSSL_WGRAD_LW=$l FTS="none" bash tools/tile_ab.sh "wgrad 1024 4 256 512" "wgrad 1024 16 64 128" "wgrad 1024 8 128 256" 2>&1 | grep -v amdgpu.ids; done
for v in 8 0 8 0; do GCSSL_WGRAD_LW=$v timeout -k 10 200 python bench.py --steps 30 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2g/b$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench wgrad_lw=$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"; done
