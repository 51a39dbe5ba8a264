#!/bin/bash
# Everything the round's final artefacts come from, in one GPU-box call: the -m gpu suite, the default bench line, the kernel
# trace + timeline of the headline configuration, the one-rank RCCL rehearsal, and the PMC passes of c2's fused data gradient.
# usage (GPU box): tools/final_run.sh <outdir under gpurun_out>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu.log 2>&1; tail -n 2 $O/gpu.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-120 $O/bench.json
tools/prof_bench.sh $1/prof > /dev/null 2>&1; head -n 2 $O/prof/kernel_summary.txt
GCSSL_FORCE_DP=1 timeout -k 10 200 python bench.py --no-also --no-cpu-baseline > $O/bench_dp1.json 2>/dev/null; cut -c1-100 $O/bench_dp1.json
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/pmc_r3/D.c2.dgrad; rm -rf $d; mkdir -p $d; i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd $R && timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d/p$i -o r -- python3 tools/actb_bench.py 768 5 > $d/p$i.log 2>&1) || { echo "pmc pass $i failed"; exit 1; }
done
echo "pmc ok"
