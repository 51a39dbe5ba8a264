#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: two separate rocprofv3 --pmc passes, --kernel-trace only) of one conv launch shape,
# e.g. the bench's dominant kernel G.up4.fwd of the batched generator forward:  tools/archive/pmc_traffic.sh dgrad 768 32 64 128 tag
set -o pipefail
KIND=$1; N=$2; HI=$3; CIN=$4; COUT=$5; TAG=${6:-traffic}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o r -- python $GRAFT_REPO_ROOT/tools/conv_bench.py $KIND $N $HI $CIN $COUT bf16 5 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done
python - <<PY
import csv, glob, json
out = {}
for i, name in ((1, "FETCH_SIZE"), (2, "WRITE_SIZE")):
    f = glob.glob("$OUT/p%d/**/*counter_collection.csv" % i, recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == name and "conv_" in r["Kernel_Name"] and "prep" not in r["Kernel_Name"]]
    vals = vals[len(vals) // 2:]                       # the timed launches (warm)
    out[name] = sum(vals) / len(vals)
    out[name + "_launches"] = len(vals)
    open("$OUT/pass%d.csv" % i, "w").write(open(f).read())
fetch = out["FETCH_SIZE"] * 1024 * 2                   # KiB; doubled on gfx950 (MI355X_MICROARCH.md: 128-B requests tallied at 64 B)
write = out["WRITE_SIZE"] * 1024
res = dict(command="tools/archive/pmc_traffic.sh $KIND $N $HI $CIN $COUT (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, conv_bench.py)",
           counters_avg_per_launch=out, hbm_fetch_bytes_corrected_x2=fetch, hbm_write_bytes=write, hbm_bytes_per_launch=fetch + write)
json.dump(res, open("$OUT/traffic.json", "w"), indent=1)
print(json.dumps(res))
PY
