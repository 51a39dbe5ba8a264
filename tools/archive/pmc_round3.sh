#!/bin/bash
# Round-3 PMC passes for the conv launches that lead bench.py's per-label table at the headline configuration
# (batch 256, 32x32, n_critic 2, bf16 by default).  Per launch shape: separate rocprofv3 runs for FETCH_SIZE, WRITE_SIZE and two SQ sets,
# --kernel-trace only (pool rule), the program directly after `--`, each bounded by its own timeout.  Writes
# gpurun_out/pmc_r3/<tag>/pN/ and gpurun_out/pmc_r3/round3_pmc_dominant.json (copied to profiles/ by hand).
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_r3
mkdir -p $OUT
DT=${1:-bf16}
# rocprofv3's own --stats table of the bench command (csv output), for profiles/round3_rocprofv3_kernel_stats.csv
rm -rf $OUT/bench_stats
(cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o r -- python3 bench.py --no-cpu-baseline --no-also > $OUT/bench_stats.json 2> $OUT/bench_stats.err) || echo "bench stats pass failed"
rm -f $OUT/bench_stats/*kernel_trace.csv $OUT/bench_stats/*.db
# tag | kernel-name filter | program + args
CASES=(
 "D.c2.fwd[n=768]|conv_dma|tools/conv_bench.py fwd_in 768 16 64 128 $DT 5"
 "D.c3.fwd[n=768]|conv_dma|tools/conv_bench.py fwd_in 768 8 128 256 $DT 5"
 "D.c4.fwd[n=768]|conv_dma|tools/conv_bench.py fwd_in 768 4 256 512 $DT 5"
 "D.c2.wgrad|conv_wgrad|tools/conv_bench.py wgrad 1024 16 64 128 $DT 5"
 "D.c3.wgrad|conv_wgrad|tools/conv_bench.py wgrad 1024 8 128 256 $DT 5"
 "D.c4.wgrad|conv_wgrad|tools/conv_bench.py wgrad 1024 4 256 512 $DT 5"
 "D.c2.dgrad|dgrad_img|tools/archive/actb_bench.py 768 5"
 "D.c3.dgrad|conv_dma|tools/conv_bench.py dgrad 768 8 128 256 $DT 5"
 "D.c4.dgrad|conv_dma|tools/conv_bench.py dgrad 768 4 256 512 $DT 5"
 "D.c1.fwd[n=768]|conv_|tools/conv_bench.py fwd16 768 32 8 64 $DT 5"
 "D.c1.gp_dgrad|conv_|tools/conv_bench.py dgrad 256 32 8 64 $DT 5"
 "G.up4.fwd[n=768]|convt_in_relu|tools/convt_bench.py 768 16 128"
)
SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE")
for c in "${CASES[@]}"; do
  IFS='|' read -r tag flt prog <<< "$c"
  d=$OUT/$(echo "$tag" | tr '[]=' '___')
  mkdir -p $d
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    (cd $R && timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d/p$i -o r -- python3 $prog > $d/p$i.log 2>&1) || { echo "$tag pass $i failed"; tail -3 $d/p$i.log; exit 1; }
  done
  echo "$tag done" >> $OUT/progress.txt
done
cd $R && python3 tools/archive/pmc_round2_summary.py $OUT $DT round3_pmc_dominant.json tools/pmc_round3.sh
