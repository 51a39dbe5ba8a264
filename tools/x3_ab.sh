#!/bin/bash
# A/B of the split-precision conv forms on the bench shapes (GPU box): waves per workgroup x tile
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for w in 4 8; do for t in 1 2; do
  for a in "fwd 768 16 64 128" "fwd 768 8 128 256" "fwd 768 4 256 512" "dgrad 768 16 64 128" "dgrad 768 8 128 256" "dgrad 768 32 64 128" "dgrad 256 8 128 256" "fwd 256 8 128 256"; do
    echo -n "waves=$w tile=$t  " >> $O/ab.log
    GCSSL_X3_WAVES=$w GCSSL_X3_TILE=$t timeout -k 10 60 python tools/conv_bench.py $a fp16x3 20 2>/dev/null >> $O/ab.log || exit 1
  done
done; done
cat $O/ab.log
