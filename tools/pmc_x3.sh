#!/bin/bash
# SQ / LDS counter passes of ONE conv configuration (GPU box): tools/pmc_x3.sh <outdir> fwd 768 16 64 128 fp16x3
# each pass its own rocprofv3 run (--kernel-trace + --pmc only), program directly after `--`
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ -f $O/counters.txt ] || (timeout -k 5 60 rocprofv3 -L > $O/counters.txt 2>&1; true)
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf $O/p$i
  (cd $R && timeout -k 10 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o r -- python3 tools/conv_bench.py "$@" 5 > $O/p$i.log 2>&1) || echo "pass $i failed: $(tail -n 2 $O/p$i.log)"
  (cd $R && python tools/pmc_summary.py $O/p$i conv_ 2>&1 | head -8)
done
