#!/bin/bash
# rocprofv3 kernel trace of the default bench command (--no-cpu-baseline --no-also: the headline configuration only) and its
# per-iteration summary.  usage (on the GPU box): tools/prof_bench.sh <outdir under gpurun_out> [extra bench args]
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python3 $R/bench.py --no-cpu-baseline --no-also "$@" > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "prof rc=$?"
cd $R
python tools/prof_summary.py $(ls $O/prof/*results.db | head -1) --csv $O/kernel_stats.csv > $O/kernel_summary.txt; head -70 $O/kernel_summary.txt
python tools/prof_timeline.py $(ls $O/prof/*results.db | head -1) > $O/timeline.txt 2>&1
# per-LABEL in-graph durations (bench.py's roofline reads the committed copy: profiles/round4_label_durations.json)
python tools/prof_labels.py $O/bench_under_rocprof.json $(ls $O/prof/*results.db | head -1) --out $O/label_durations.json > $O/label_durations.txt 2>&1
rm -rf $O/prof
