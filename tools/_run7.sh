set -o pipefail
mkdir -p gpurun_out/r2d
for ft in none 256x128 256x64 128x128; do
  if [ $ft = none ]; then unset GCSSL_FORCE_TILE; else export GCSSL_FORCE_TILE=$ft; fi
  GCSSL_BENCH_VERBOSE=1 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2d/ft_$ft.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$ft', d['value'], d['ms_per_step'], d['roofline']['all_convs'])"
done
python - <<'PY'
import re
rows = {}
for ft in ("none", "256x128", "256x64", "128x128"):
    for l in open(f"gpurun_out/r2d/ft_{ft}.err"):
        m = re.match(r"\[probe\] (\S+)\s+(\d+)/iter\s+([\d.]+) us", l)
        if m: rows.setdefault(m.group(1), {})[ft] = float(m.group(3))
print(f"{'label':26s} {'none':>8s} {'256x128':>8s} {'256x64':>8s} {'128x128':>8s}")
for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get('none', 0)):
    if 'wgrad' in k: continue
    print(f"{k:26s} " + " ".join(f"{v.get(ft, float('nan')):8.1f}" for ft in ("none", "256x128", "256x64", "128x128")))
PY
