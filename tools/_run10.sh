set -o pipefail
mkdir -p gpurun_out/r2h
for cfg in "--batch 512 --size 32" "--batch 128 --size 64" "--batch 256 --size 64" "--batch 128 --size 128" "--batch 64 --size 128 --n_critic 5" "--batch 1024 --size 32" "--generator simple" "--batch 64 --size 32 --dtype fp32" "--batch 32 --size 32"; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --probe-steps 1 --sustain-s 0 $cfg 2>gpurun_out/r2h/cfg.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['dtype'], d['value'], d['ms_per_step'], d['finite_after_run'])" || { echo "$cfg FAILED"; tail -3 gpurun_out/r2h/cfg.err; }
done
