set -o pipefail
mkdir -p gpurun_out/r2d
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "conv_fwd or conv_dgrad or conv_wgrad or split_k" 2>&1 | tail -3
for i in 1 2; do
GCSSL_LIB=$PWD/gan-calibrated-semi-supervised-learning_amd/libgcssl_prev.so timeout -k 10 200 python bench.py --steps 20 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prev', d['value'], d['sustained_ms_per_step'], d['roofline']['all_convs'])"
GCSSL_BENCH_VERBOSE=1 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 2>gpurun_out/r2d/new.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new ', d['value'], d['sustained_ms_per_step'], d['roofline']['all_convs'], d['roofline']['kernel'], d['roofline']['frac'])"
done
grep probe gpurun_out/r2d/new.err | head -24
