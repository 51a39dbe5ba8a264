set -o pipefail
mkdir -p gpurun_out/r2g
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r2g/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2g/prof -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --dtype fp16 --sustain-s 0 > $GRAFT_REPO_ROOT/gpurun_out/r2g/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2g/bench_prof.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls gpurun_out/r2g/prof/*results.db | head -1) --csv gpurun_out/r2g/kernel_stats.csv > gpurun_out/r2g/prof_summary.txt; head -50 gpurun_out/r2g/prof_summary.txt
