set -o pipefail
mkdir -p gpurun_out/r2g
GCSSL_FORCE_TILE=ring256 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail 5 -k "conv_fwd or conv_dgrad" > gpurun_out/r2g/k_tests.log 2>&1; rc=$?; echo "ring256 kernel tests rc=$rc"; tail -3 gpurun_out/r2g/k_tests.log
[ $rc -eq 0 ] || exit 1
FTS="none ring256" bash tools/tile_ab.sh "fwd 768 16 64 128" "dgrad 768 8 128 256" "dgrad 768 4 128 512" 2>&1 | grep -v amdgpu.ids
for v in 1 0 1 0; do GCSSL_RING256=$v GCSSL_BENCH_VERBOSE=1 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2g/b$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench ring256=$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"; done
python - <<'PY'
import re
rows = {}
for v in ("1", "0"):
    for l in open(f"gpurun_out/r2g/b{v}.err"):
        m = re.match(r"\[probe\] (\S+)\s+(\d+)/iter\s+([\d.]+) us", l)
        if m: rows.setdefault(m.group(1), {})[v] = float(m.group(3))
for k, d in sorted(rows.items(), key=lambda kv: -abs(kv[1].get("1", 0) - kv[1].get("0", 0))):
    if abs(d.get("1", 0) - d.get("0", 0)) > 2.0: print(f"{k:26s} on {d.get('1'):7.1f}  off {d.get('0'):7.1f}")
PY
