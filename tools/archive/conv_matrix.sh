#!/bin/bash
# forced-tile experiment matrix for single conv layers (bf16, LDS-DMA kernels); run on the GPU box
for cfg in "fwd 768 16 64 128" "fwd 768 8 128 256" "fwd 768 4 256 512" "dgrad 768 16 64 128" "dgrad 768 8 128 256" \
           "dgrad 256 32 64 128" "dgrad 256 16 64 256" "fwd 256 16 64 128" "fwd 256 8 128 256"; do
  for t in default 128x64 128x128 256x64 256x128; do
    echo -n "$t : "
    if [ $t = default ]; then python tools/conv_bench.py $cfg bf16 30 2>/dev/null | tail -1
    else GCSSL_FORCE_TILE=$t python tools/conv_bench.py $cfg bf16 30 2>/dev/null | tail -1; fi
  done
done
