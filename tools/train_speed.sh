#!/bin/bash
# train.py end to end, eager loop vs --graph, on the synthetic source at the bench shape with a device-resident pool of 8 batches
# (without --synthetic_pool the source builds every batch on the host: ~0.1 s each, and both loops measure numpy).  usage (GPU box): tools/train_speed.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
for mode in "" "--graph"; do
  t0=$(date +%s.%N)
  timeout -k 10 400 python train.py --batch_size 256 --img_size 32 --n_critic 2 --n_epochs 3 --iters_per_epoch 1000 \
      --patience 10 --synthetic_pool 8 --save_dir $O/run$mode $mode > $O/train$mode.log 2> $O/train$mode.err
  t1=$(date +%s.%N)
  python - $O/run$mode/training_history.json "$mode" $t0 $t1 <<'PY'
import json, sys
h = json.load(open(sys.argv[1]))
e = h[-1]
print(f"mode [{sys.argv[2]}]: {float(sys.argv[4]) - float(sys.argv[3]):.1f} s in all; last epoch: {e['iterations']} iterations in {e['train_seconds']:.3f} s = "
      f"{e['train_seconds'] / e['iterations'] * 1e3:.3f} ms per iteration = {256 * e['iterations'] / e['train_seconds']:.0f} images/s")
PY
done
rm -f $O/run*/G_best.pth            # (checkpoints: tens of MB each; gpurun_out is capped)
