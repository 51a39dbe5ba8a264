set -o pipefail
mkdir -p gpurun_out/r2f
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2f/prof -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --dtype fp16 --sustain-s 0 > $GRAFT_REPO_ROOT/gpurun_out/r2f/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2f/bench_prof.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
ls gpurun_out/r2f/prof | head
python tools/prof_summary.py $(ls gpurun_out/r2f/prof/*results.db | head -1) --csv gpurun_out/r2f/kernel_stats.csv | head -70
