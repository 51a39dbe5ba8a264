#!/bin/bash
# Round-4 PMC passes for the conv launches that lead bench.py's per-label table at the headline configuration (batch 256, 32x32,
# n_critic 2, bf16), for the split-precision mode's leading launches (fp16x3) and for the re-crop stage.  Per launch shape: separate
# rocprofv3 runs for FETCH_SIZE, WRITE_SIZE and two SQ sets, --kernel-trace only (pool rule), the program directly after `--`,
# each bounded by its own timeout.  Writes gpurun_out/<out>/<tag>/pN/ and gpurun_out/<out>/round4_pmc_dominant.json (copied to
# profiles/ by tools/final_run.sh).  usage (GPU box): tools/pmc_round4.sh <outdir under gpurun_out> [bf16|fp16]
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_r4}
mkdir -p $OUT
DT=${2:-bf16}
export GCSSL_CT_FUSED_ONLY=1
# tag | program + args          (every program takes its dtype: round 3's dominant-label record was fp16 under a bf16 heading)
CASES=(
 "G.up4.fwd[n=768]|tools/convt_bench.py 768 16 128 $DT"
 "D.c2.fwd[n=768]|tools/conv_bench.py fwd_in 768 16 64 128 $DT 5"
 "D.c3.fwd[n=768]|tools/conv_bench.py fwd_in 768 8 128 256 $DT 5"
 "D.c4.fwd[n=768]|tools/conv_bench.py fwd_in 768 4 256 512 $DT 5"
 "D.c2-4.wgrad|tools/wgrad_batch_bench.py 256 32 $DT 5"
 "D.c2.wgrad|tools/conv_bench.py wgrad 1024 16 64 128 $DT 5"
 "D.c4.wgrad|tools/conv_bench.py wgrad 1024 4 256 512 $DT 5"
 "D.c3.dgrad|tools/conv_bench.py dgrad 768 8 128 256 $DT 5"
 "x3.D.c2.fwd[n=768]|tools/conv_bench.py fwd 768 16 64 128 fp16x3 5"
 "x3.D.c3.dgrad|tools/conv_bench.py dgrad 768 8 128 256 fp16x3 5"
 "x3.D.c2.wgrad|tools/conv_bench.py wgrad 1024 16 64 128 fp16x3 5"
 "recrop[B=256,1280x720]|tools/recrop_bench.py 256 32 32 1280 720"
)
SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS")
for c in "${CASES[@]}"; do
  IFS='|' read -r tag prog <<< "$c"
  d=$OUT/$(echo "$tag" | tr '[]=,' '____')
  mkdir -p $d
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    (cd $R && timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d/p$i -o r -- python3 $prog > $d/p$i.log 2>&1) || { echo "$tag pass $i failed"; tail -3 $d/p$i.log; exit 1; }
    find $d/p$i -name "*agent_info.csv" -delete
  done
  echo "$tag done" >> $OUT/progress.txt
done
cd $R && python3 tools/pmc_round4_summary.py $OUT $DT round4_pmc_dominant.json tools/pmc_round4.sh
