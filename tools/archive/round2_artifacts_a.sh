#!/bin/bash
# Round-2 evidence, part A: the full GPU suite, the mode-error table, the default bench line and its rocprofv3 kernel trace.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round2
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail 10 > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
timeout -k 10 300 python tools/mode_error.py --speed --out $O/round2_mode_error.json > $O/mode_error.log 2>&1; echo "mode_error rc=$?"; tail -5 $O/mode_error.log
cp $O/round2_mode_error.json $R/profiles/round2_mode_error.json 2>/dev/null
timeout -k 10 600 python bench.py > $O/round2_bench.json 2> $O/round2_bench.err; echo "bench rc=$?"; cat $O/round2_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r -- python3 $R/bench.py --no-cpu-baseline > $O/round2_bench_under_rocprof.json 2> $O/round2_bench_under_rocprof.err; echo "prof rc=$?"
cd $R
python tools/prof_summary.py $(ls $O/prof/*results.db | head -1) --csv $O/round2_kernel_stats.csv > $O/round2_kernel_summary.txt; head -12 $O/round2_kernel_summary.txt
rm -rf $O/prof                                   # the raw trace (tens of MB) stays on the box: gpurun copies back <= 64 MiB
# rocprofv3's own --stats table (csv) of the same bench command
rm -rf $O/bench_stats
(cd $R && cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o r -- python3 $R/bench.py --no-cpu-baseline > $O/bench_stats.json 2> $O/bench_stats.err) || echo "bench stats pass failed"
rm -f $O/bench_stats/r_kernel_trace.csv $O/bench_stats/*.db; ls $O/bench_stats | head
