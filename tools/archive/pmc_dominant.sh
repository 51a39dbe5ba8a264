#!/bin/bash
# SQ + HBM-traffic PMC passes for the bench's dominant kernel (G.up4.fwd == conv dgrad-form N=256 32x32 Cin=64 Cout=128).
# Separate rocprofv3 runs per counter group, --kernel-trace only (pool rule); each pass is bounded by its own timeout.
set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_dom
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o r -- python $GRAFT_REPO_ROOT/tools/conv_bench.py dgrad 256 32 64 128 bf16 5 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; exit 1; }
  echo "pass $i done" >> $OUT/progress.txt
done
ls $OUT
