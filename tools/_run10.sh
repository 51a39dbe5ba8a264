set -o pipefail
mkdir -p gpurun_out/r2h
timeout -k 10 900 python -m pytest tests -m gpu -q --maxfail 10 > gpurun_out/r2h/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 gpurun_out/r2h/gpu_tests.log
for i in 1 2; do timeout -k 10 200 python bench.py --steps 30 --warmup 3 --dtype fp16 --no-cpu-baseline --probe-steps 2 --sustain-s 0 2>gpurun_out/r2h/b.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])"; done
