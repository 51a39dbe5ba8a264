set -o pipefail
FTS="none lw64" bash tools/tile_ab.sh "fwd 768 16 64 128" "dgrad 768 16 64 128" "dgrad 768 8 128 256" "dgrad 256 32 64 128" 2>&1 | grep -v amdgpu.ids
GCSSL_FORCE_TILE=lw64 timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --maxfail 5 -k "conv_fwd or conv_dgrad" 2>&1 | tail -2
