set -o pipefail
mkdir -p gpurun_out/r2g
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail 5 -k "conv_fwd or conv_dgrad" > gpurun_out/r2g/k_tests.log 2>&1; rc=$?; echo "kernel tests rc=$rc"; tail -2 gpurun_out/r2g/k_tests.log
[ $rc -eq 0 ] || exit 1
for l in 4 0; do echo "PERSIST_LW=$l"; GCSSL_PERSIST_LW=$l FTS="none" bash tools/tile_ab.sh "fwd 768 16 64 128" "fwd 256 32 64 128" 2>&1 | grep -v amdgpu.ids; done
for v in 4 0 4 0; do GCSSL_PERSIST_LW=$v timeout -k 10 200 python bench.py --steps 30 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2g/b$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench persist_lw=$v', d['value'], d['ms_per_step'])"; done
