set -o pipefail
mkdir -p gpurun_out/r2g
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_simple_gen_gpu.py -m gpu -q --maxfail 5 > gpurun_out/r2g/k_tests.log 2>&1; rc=$?; echo "kernel tests rc=$rc"; tail -3 gpurun_out/r2g/k_tests.log
[ $rc -eq 0 ] || exit 1
for v in 1 0 1 0; do GCSSL_EPI_LDS=$v GCSSL_BENCH_VERBOSE=1 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2g/b$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench epi_lds=$v', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['d_convs']['frac'])"; done
GCSSL_RING_DEBUG=3 FTS="ring" bash tools/tile_ab.sh "fwd 768 8 128 256" 2>&1 | grep -v amdgpu.ids
FTS="none" bash tools/tile_ab.sh "fwd 768 8 128 256" "dgrad 768 4 256 512" "fwd 256 8 128 256" 2>&1 | grep -v amdgpu.ids
