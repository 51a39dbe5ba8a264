#!/bin/bash
# tile/wave experiment matrix for single conv layers (bf16)
for cfg in "fwd 768 16 64 128" "fwd 768 8 128 256" "fwd 768 4 256 512" "dgrad 768 8 128 256" "fwd 256 4 256 512" "dgrad 256 32 64 128"; do
  for t in 1 300 100000; do for w in 4 8; do
    echo -n "TILE_WGS=$t WAVES=$w : "; GCSSL_TILE_WGS=$t GCSSL_DMA_WAVES=$w python tools/conv_bench.py $cfg bf16 30 2>/dev/null | tail -1
  done; done
done
