set -o pipefail
mkdir -p gpurun_out/r2b
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^$" gpurun_out/r2b/pytest.log | grep "FAILED\|passed\|failed\|Error" | tail -20
GCSSL_BENCH_VERBOSE=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --dtype fp16 --no-cpu-baseline > gpurun_out/r2b/bench_fp16.json 2> gpurun_out/r2b/bench_fp16.err; echo "bench fp16 rc=$?"
GCSSL_FUSED_UP=1 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --dtype fp16 --no-cpu-baseline > gpurun_out/r2b/bench_fp16_up4only.json 2> gpurun_out/r2b/bench_fp16_up4only.err; echo "bench up4-only rc=$?"
python -c "
import json
for f in ('bench_fp16','bench_fp16_up4only'):
    d=json.load(open('gpurun_out/r2b/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['sustained_ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['avg_us'], d['roofline']['all_convs'], d['roofline']['d_convs'])
"
grep "G.up" gpurun_out/r2b/bench_fp16.err | head -6
