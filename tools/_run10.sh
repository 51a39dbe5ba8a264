set -o pipefail
mkdir -p gpurun_out/r2g
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail 5 -k "fused_convT" > gpurun_out/r2g/k_tests.log 2>&1; rc=$?; echo "fused tests rc=$rc"; tail -2 gpurun_out/r2g/k_tests.log
[ $rc -eq 0 ] || { grep -n "^E " gpurun_out/r2g/k_tests.log | head -5; exit 1; }
timeout -k 10 100 python tools/convt_bench.py 768 16 128; timeout -k 10 100 python tools/convt_bench.py 768 8 256
for i in 1 2; do timeout -k 10 200 python bench.py --steps 30 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2g/b.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"; done
