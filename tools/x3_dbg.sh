#!/bin/bash
# timing experiments on the split-precision forward kernel: which phase of a K step sets its time (GPU box)
# GCSSL_X3_DEBUG bits: 1 no global loads, 2 no MFMA phase, 4 no split + LDS stores (results are garbage); then forced tiles
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for d in 0 1 2 4 3 5 6 7; do
  for a in "fwd 768 16 64 128" "fwd 768 8 128 256"; do
    echo -n "dbg=$d  " >> $O/dbg.log
    GCSSL_X3_DEBUG=$d timeout -k 10 60 python tools/conv_bench.py $a fp16x3 20 2>/dev/null >> $O/dbg.log || exit 1
  done
done
for t in 1 2; do for a in "fwd 768 16 64 128" "fwd 768 8 128 256" "fwd 768 4 256 512"; do
    echo -n "tile=$t  " >> $O/dbg.log
    GCSSL_X3_TILE=$t timeout -k 10 60 python tools/conv_bench.py $a fp16x3 20 2>/dev/null >> $O/dbg.log || exit 1
done; done
cat $O/dbg.log
