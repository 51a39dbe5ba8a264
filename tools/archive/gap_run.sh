#!/bin/bash
# usage (GPU box): tools/archive/gap_run.sh <outdir under gpurun_out> <mode> <marker> <every>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$2
timeout -k 10 200 python3 $R/tools/graph_gap_probe.py $2 > $O/$2.plain.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof_$2 -o r -- python3 $R/tools/graph_gap_probe.py $2 > $O/$2.txt 2>&1; echo "rc=$?"
cd $R
python tools/prof_timeline.py $(ls $O/prof_$2/*results.db | head -1) 20 $3 $4 > $O/timeline_$2.txt 2>&1
rm -rf $O/prof_$2
