set -o pipefail
mkdir -p gpurun_out/r2g
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail 5 -k "conv_fwd or conv_dgrad" > gpurun_out/r2g/k_tests.log 2>&1; echo "kernel tests rc=$?"; tail -3 gpurun_out/r2g/k_tests.log
GCSSL_BENCH_VERBOSE=1 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2g/b1.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"; grep "c1\.\|down1" gpurun_out/r2g/b1.err
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r2g/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2g/prof -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --dtype fp16 --sustain-s 0 > $GRAFT_REPO_ROOT/gpurun_out/r2g/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2g/bench_prof.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py $(ls gpurun_out/r2g/prof/*results.db | head -1) --csv gpurun_out/r2g/kernel_stats.csv > gpurun_out/r2g/prof_summary.txt; grep "_c8_\|iterations" gpurun_out/r2g/prof_summary.txt
