set -o pipefail
mkdir -p gpurun_out/r2g
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail 5 -k "conv_fwd or conv_dgrad" > gpurun_out/r2g/k_tests.log 2>&1; rc=$?; echo "kernel tests rc=$rc"; tail -3 gpurun_out/r2g/k_tests.log
for l in 0 1; do echo "LW8=$l"; GCSSL_RING_LW8=$l FTS="none" bash tools/tile_ab.sh "fwd 768 8 128 256" "dgrad 768 4 256 512" "fwd 768 4 256 512" 2>&1 | grep -v amdgpu.ids; done
for i in 1 2; do timeout -k 10 200 python bench.py --steps 30 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2g/b.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"; done
