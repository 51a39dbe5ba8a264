#!/bin/bash
# True kernel durations (rocprofv3 kernel trace) of one conv shape under the forced-tile knob:
#   tools/archive/tile_ab.sh "fwd 768 8 128 256" "dgrad 768 8 128 256" ...
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tile_ab
mkdir -p $O
for shape in "$@"; do
  for ft in ${FTS:-none 128x128 256x128 256x64}; do
    tag=$(echo "$shape-$ft" | tr ' ' '_')
    if [ $ft = none ]; then unset GCSSL_FORCE_TILE; else export GCSSL_FORCE_TILE=$ft; fi
    rm -rf $O/$tag
    (cd $R && timeout -k 10 100 rocprofv3 --kernel-trace --output-format csv -d $O/$tag -o r -- python3 tools/conv_bench.py $shape fp16 20 > $O/$tag.log 2>&1) || { echo "$tag failed"; continue; }
    python3 - <<PY
import csv, glob
f = glob.glob("$O/$tag/**/*kernel_trace.csv", recursive=True)[0]
d = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "conv_" in k and "prep" not in k:
        d.setdefault(k[:70], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = v[len(v) // 2:]
    print("$shape", "$ft", k, f"{sum(v)/len(v):.1f} us x{len(v)}")
PY
  done
done
