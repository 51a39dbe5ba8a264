set -o pipefail
mkdir -p gpurun_out/r2e
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv_" --maxfail 5 > gpurun_out/r2e/conv_tests.log 2>&1; echo "conv tests rc=$?"; tail -3 gpurun_out/r2e/conv_tests.log
for w in 1 0; do
  GCSSL_WGRAD_DMA=$w GCSSL_BENCH_VERBOSE=1 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2e/w$w.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wdma=$w', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"
done
grep "wgrad" gpurun_out/r2e/w1.err | head -20
