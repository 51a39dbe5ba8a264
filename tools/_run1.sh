set -o pipefail
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -q --maxfail=25 -x -k "not full_size" > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
python tools/fp16_ranges.py > gpurun_out/r2a/ranges.log 2>&1; echo "ranges rc=$?"
python tools/mode_error.py --speed --out gpurun_out/r2a/mode_error.json > gpurun_out/r2a/mode_error.log 2>&1; echo "mode_error rc=$?"
GCSSL_BENCH_VERBOSE=1 python bench.py --steps 20 --warmup 3 --dtype bf16 > gpurun_out/r2a/bench_bf16.json 2> gpurun_out/r2a/bench_bf16.err; echo "bench bf16 rc=$?"
GCSSL_BENCH_VERBOSE=1 python bench.py --steps 20 --warmup 3 --dtype fp16 --no-cpu-baseline > gpurun_out/r2a/bench_fp16.json 2> gpurun_out/r2a/bench_fp16.err; echo "bench fp16 rc=$?"
python bench.py --gpus 2 --steps 2 > gpurun_out/r2a/bench_2gpu.out 2>&1; echo "bench --gpus 2 rc=$? (expected non-zero on a 1-GPU box)"
cat gpurun_out/r2a/bench_fp16.json
