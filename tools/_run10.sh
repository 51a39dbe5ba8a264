set -o pipefail
mkdir -p gpurun_out/r2g
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail 5 -k "fused_convT or instance_norm or slab" > gpurun_out/r2g/k_tests.log 2>&1; rc=$?; echo "kernel tests rc=$rc"; tail -3 gpurun_out/r2g/k_tests.log
[ $rc -eq 0 ] || { grep -n "^E " gpurun_out/r2g/k_tests.log | head; exit 1; }
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -m gpu -q --maxfail 5 > gpurun_out/r2g/e_tests.log 2>&1; echo "engine tests rc=$?"; tail -3 gpurun_out/r2g/e_tests.log
for v in 1 0 1 0; do GCSSL_UP4_PRESUM=$v timeout -k 10 200 python bench.py --steps 30 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2g/b$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench presum=$v', d['value'], d['ms_per_step'])"; done
