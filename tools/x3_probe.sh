#!/bin/bash
# split-precision modes on the GPU box: kernel parity, conv timings of the bench shapes, engine error + speed
# usage: tools/x3_probe.sh <outdir under gpurun_out>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "split_precision" > $O/kern.log 2>&1; tail -n 5 $O/kern.log
for m in ${X3_MODES:-fp16x3 bf16x3} ; do
 for a in "fwd 768 16 64 128" "fwd 768 8 128 256" "fwd 768 4 256 512" "dgrad 768 16 64 128" "dgrad 768 8 128 256" "dgrad 768 4 256 512" "wgrad 1024 16 64 128" "wgrad 1024 8 128 256" "wgrad 1024 4 256 512" "dgrad 768 32 64 128" "fwd 256 4 256 512" "dgrad 256 8 128 256"; do
  timeout -k 10 60 python tools/conv_bench.py $a $m 20 >> $O/conv.log 2>&1 || exit 1
 done
done
cat $O/conv.log
timeout -k 10 400 python tools/mode_error.py --speed --modes ${X3_ERR_MODES:-fp16x3,bf16x3} --out $O/mode_error.json > $O/mode.log 2>&1; cat $O/mode.log
