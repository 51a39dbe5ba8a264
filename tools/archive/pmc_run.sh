#!/bin/bash
# One rocprofv3 --pmc pass (kernel trace only, as the pool requires; the program directly after `--`) over a python tool:
#   tools/archive/pmc_run.sh TAG "COUNTER1 COUNTER2 ..." tools/convt_bench.py 768 16 128
# writes gpurun_out/pmc_TAG/ and prints the per-kernel averages of every counter (tools/pmc_summary.py).
set -o pipefail
TAG=$1; CTRS=$2; shift 2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
(cd $GRAFT_REPO_ROOT && timeout -k 10 240 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -o r -- python3 "$@" > $OUT/run.log 2>&1) || { echo "pass $TAG failed"; tail -5 $OUT/run.log; exit 1; }
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT
