#!/bin/bash
# same-box A/B of environment knobs on the headline bench (GPU box): tools/ab_env.sh <outdir> "<ENV=VAL ...>" ["<ENV=VAL ...>" ...]
# (the first variant should be "" = the default build; every variant is one `bench.py --no-also --no-cpu-baseline` run)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
i=0
for v in "$@" "" ; do
  i=$((i+1))
  env $v timeout -k 10 200 python bench.py --no-also --no-cpu-baseline ${AB_ARGS} > $O/v$i.json 2> $O/v$i.err || { echo "variant [$v] failed"; tail -n 3 $O/v$i.err; continue; }
  python - "$v" $O/v$i.json <<'PY'
import json, sys
b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"[{sys.argv[1] or 'default'}]  {b['value']:.0f} images/s  {b['ms_per_step']:.4f} ms  sustained {b['sustained_ms_per_step']}  d_convs {b['roofline']['d_convs']['frac']}")
PY
done
