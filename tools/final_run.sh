#!/bin/bash
# Everything the round's committed artefacts come from, in one GPU-box call: the -m gpu suite, the default bench line, the kernel
# trace + timeline + per-label in-graph durations of the headline configuration, the same for the split-precision mode, the
# one-rank RCCL rehearsal, the interference probe and the PMC passes.  usage (GPU box): tools/final_run.sh <outdir under gpurun_out> [parts]
# parts (default all): t = tests, b = bench, p = profiles, x = split-mode profile, d = DP rehearsal, i = interference, c = PMC
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; P=${2:-tbpxdic}
cd $R
if [[ $P == *t* ]]; then timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gpu.log 2>&1; tail -n 2 $O/gpu.log; fi
if [[ $P == *b* ]]; then timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-160 $O/bench.json; fi
if [[ $P == *p* ]]; then tools/prof_bench.sh $1/prof > /dev/null 2>&1; head -n 2 $O/prof/kernel_summary.txt; head -n 3 $O/prof/label_durations.txt; fi
if [[ $P == *x* ]]; then tools/prof_bench.sh $1/prof_x3 --dtype fp16x3 > /dev/null 2>&1; head -n 2 $O/prof_x3/kernel_summary.txt; fi
if [[ $P == *d* ]]; then GCSSL_FORCE_DP=1 timeout -k 10 200 python bench.py --no-also --no-cpu-baseline > $O/bench_dp1.json 2>/dev/null; cut -c1-100 $O/bench_dp1.json; fi
if [[ $P == *i* ]]; then tools/interference.sh $1/intf > $O/intf.log 2>&1; tail -n 12 $O/intf/interference_trace.txt; fi
if [[ $P == *c* ]]; then tools/pmc_round4.sh $1/pmc > $O/pmc.log 2>&1; tail -n 14 $O/pmc.log; fi
