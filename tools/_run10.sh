set -o pipefail
mkdir -p gpurun_out/r2g
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail 5 -k "conv_fwd or conv_dgrad or slab or split" > gpurun_out/r2g/k_tests.log 2>&1; rc=$?; echo "kernel tests rc=$rc"; tail -2 gpurun_out/r2g/k_tests.log
for i in 1 2 3; do GCSSL_BENCH_VERBOSE=1 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2g/b.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['d_convs']['frac'])"; done
grep "c4.fwd\|down4.fwd" gpurun_out/r2g/b.err
