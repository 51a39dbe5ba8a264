set -o pipefail
mkdir -p gpurun_out/r2c
timeout -k 10 120 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "fused_convT" 2>&1 | tail -3
timeout -k 10 200 python tools/convt_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2c/v2.log
