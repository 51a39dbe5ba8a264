#!/bin/bash
# the two chains of an iteration alone and together -- kernel traces (durations, co-residency) and the L2
# (TCC) hit / miss / write-back / invalidate counters of the same three runs.  usage (GPU box): tools/interference.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in critic_only gen_only full; do
  rm -rf $O/t_$m
  (cd $R && timeout -k 10 200 rocprofv3 --kernel-trace -d $O/t_$m -o r -- python3 tools/graph_gap_probe.py $m 150 > $O/t_$m.log 2>&1) || { echo "trace $m failed"; tail -n 3 $O/t_$m.log; exit 1; }
done
cd $R
python tools/interference.py $(ls $O/t_critic_only/*results.db | head -1) $(ls $O/t_gen_only/*results.db | head -1) $(ls $O/t_full/*results.db | head -1) > $O/interference_trace.txt 2>&1
cat $O/interference_trace.txt
cd /tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_ALL_TC_OP_INV_EVICT_sum TCC_NORMAL_WRITEBACK_sum TCC_WRITEBACK_sum"; do
  i=$((i+1))
  for m in critic_only gen_only full; do
    rm -rf $O/p${i}_$m
    (cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p${i}_$m -o r -- python3 tools/graph_gap_probe.py $m 30 > $O/p${i}_$m.log 2>&1) || { echo "pmc $i $m failed"; tail -n 3 $O/p${i}_$m.log; }
  done
done
cd $R
python tools/interference_pmc.py $O > $O/interference_pmc.txt 2>&1; cat $O/interference_pmc.txt
for d in $O/t_* $O/p?_*; do [ -d $d ] && find $d -name "*.db" -delete; [ -d $d ] && find $d -name "*agent_info.csv" -delete; done
