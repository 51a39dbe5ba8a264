#!/bin/bash
# same-box A/B of the working tree against the tree in ab_prev/ (a built checkout of an earlier commit; not committed):
# alternate the two bench commands and print images/s of the sustained window.  usage: tools/archive/ab_prev.sh <outdir> [bench args]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
mkdir -p $O
for i in 1 2 3; do
  for t in prev new; do
    if [ $t = prev ]; then D=$R/ab_prev; else D=$R; fi
    (cd $D && timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --sustain-s 3 "$@" > $O/${t}_$i.json 2> $O/${t}_$i.err)
    python - <<PY
import json
d = json.load(open("$O/${t}_$i.json"))
print("$t", $i, d["dtype"], d["value"], d["sustained"]["images_per_s"], d["roofline"]["d_convs"]["frac"])
PY
  done
done
