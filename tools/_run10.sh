set -o pipefail
for d in 0 3 7 8 19 23; do echo "RING_DEBUG=$d"; GCSSL_RING_DEBUG=$d FTS="ring" bash tools/tile_ab.sh "fwd 768 8 128 256" 2>&1 | grep -v amdgpu.ids; done
