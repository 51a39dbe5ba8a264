set -o pipefail
mkdir -p gpurun_out/r2g
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail 5 -k "conv_fwd or conv_dgrad" > gpurun_out/r2g/k_tests.log 2>&1; echo "kernel tests rc=$?"; tail -3 gpurun_out/r2g/k_tests.log
for v in 1 0; do GCSSL_C8_DGRAD=$v GCSSL_BENCH_VERBOSE=1 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2g/b$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench c8dgrad=$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"; grep "c1\.gp" gpurun_out/r2g/b$v.err; done
