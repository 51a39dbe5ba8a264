set -o pipefail
mkdir -p gpurun_out/r2g
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "instance_norm or act_bwd or in_" --maxfail 5 > gpurun_out/r2g/norm_tests.log 2>&1; echo "norm tests rc=$?"; tail -5 gpurun_out/r2g/norm_tests.log
timeout -k 10 200 python tools/norm_bench.py fp16 2>&1 | grep "768x8x8x128\|256x16x16x64"
GCSSL_BENCH_VERBOSE=1 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2g/b.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"
