#!/bin/bash
# GPU-side gaps at graph boundaries: kernel traces of the critic's chain replayed as two graphs (the product's), as one graph, as
# three, and of the full two-stream iteration.  usage (GPU box): tools/graph_gaps.sh <outdir> [dtype]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; DT=${2:-bf16}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in critic_only critic_fused critic_steps full; do
  rm -rf $O/t_$m
  (cd $R && timeout -k 10 200 rocprofv3 --kernel-trace -d $O/t_$m -o r -- python3 tools/graph_gap_probe.py $m 150 $DT > $O/t_$m.log 2>&1) || { echo "trace $m failed"; tail -n 3 $O/t_$m.log; exit 1; }
  grep "us per iteration" $O/t_$m.log
  db=$(ls $O/t_$m/*results.db | head -1)
  if [ $m == full ]; then python $R/tools/prof_timeline.py $db > $O/tl_$m.txt 2>&1; else python $R/tools/prof_timeline.py $db 20 gp_norm_kernel 2 > $O/tl_$m.txt 2>&1; fi
  python $R/tools/gap_report.py $O/tl_$m.txt 8
  rm -rf $O/t_$m
done
