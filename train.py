#!/usr/bin/env python3
"""Training harness with the reference's surface (cgan/cgan_train_enhanced.py:139-168,256-261,481-489): same
argument names and defaults (read from config.yaml), Adam(lr, (beta1, beta2)) for both nets, the same logged scalars
and the same checkpoint dict keys {'generator','discriminator','epoch','delta_iou','config'} -- driving the MI355X
step engine.  Data: `--source synthetic` (default; the tensor contract of SURVEY §8a row I, no files needed) or
`--source dataset`: CalibratorDataset over --data_dir (YOLO txt + jpg) with the images decoded once into
an HBM atlas, the pred/gt patches and the per-step re-crop cut on the GPU.  Any iterable yielding
(pred_patch, gt_patch, delta_true, pred_box, refine_fn) can feed the loop.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
from pathlib import Path

import torch
import yaml

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
PKG = "gan-calibrated-semi-supervised-learning_amd"


def build_parser(config: dict) -> argparse.ArgumentParser:
    """Argument surface of cgan/cgan_train_enhanced.py:145-168 (+ the MI355X-specific switches at the end)."""
    p = argparse.ArgumentParser()
    p.add_argument("--data_dir", type=str, default=config["data_dir"])
    p.add_argument("--img_size", type=int, default=config["img_size"])
    p.add_argument("--batch_size", type=int, default=config["batch_size"])
    p.add_argument("--n_epochs", type=int, default=config["n_epochs"])
    p.add_argument("--lr", type=float, default=config["lr"])
    p.add_argument("--beta1", type=float, default=config["beta1"])
    p.add_argument("--beta2", type=float, default=config["beta2"])
    p.add_argument("--lambda_iou", type=float, default=config["lambda_iou"])
    p.add_argument("--use_eiou", action="store_true", default=config.get("use_eiou", True))
    p.add_argument("--pure_eiou", action="store_true", default=config.get("pure_eiou", True))
    p.add_argument("--spectral_norm", action="store_true", default=config["spectral_norm"])
    p.add_argument("--delta_scale", type=float, default=config["delta_scale"])
    p.add_argument("--generator_type", type=str, default=config["generator_type"])
    p.add_argument("--patience", type=int, default=config["early_stop"]["patience"])
    p.add_argument("--min_delta", type=float, default=config["early_stop"]["min_delta"])
    p.add_argument("--train_split", type=float, default=config["train_split"])
    p.add_argument("--val_split", type=float, default=config["val_split"])
    p.add_argument("--save_dir", type=str, default=config["save_dir"])
    p.add_argument("--seed", type=int, default=config["seed"])
    p.add_argument("--lambda_gp", type=float, default=config.get("lambda_gp", 10.0))
    p.add_argument("--n_critic", type=int, default=config.get("n_critic", 5))
    # MI355X-specific
    p.add_argument("--compute_dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    p.add_argument("--iters_per_epoch", type=int, default=20, help="synthetic source: iterations per epoch")
    p.add_argument("--source", default="synthetic", choices=["synthetic", "dataset"])
    return p


def load_config() -> dict:
    with open(ROOT / PKG / "config.yaml", "r", encoding="utf-8") as f:
        return yaml.safe_load(f)


def split_lengths(n: int, val_split: float):
    """(train_len, val_len) exactly as cgan/cgan_train_enhanced.py:220-221 computes them (`train_split` is parsed there and
    never used): 18 523 pairs at val_split 0.1 -> 16 671 / 1 852, the numbers the reference's own run logged."""
    val_len = max(1, int(val_split * n))
    return n - val_len, val_len


def shard_indices(indices, rank: int, world: int, batch: int):
    """Rank `rank`'s share of `indices` for data-parallel training: strided, then cut to the number of WHOLE batches the
    shortest shard holds, so every rank runs the same number of iterations (each iteration all-reduces: one rank with an
    extra batch would wait for ever)."""
    per_rank = len(indices) // world                       # the shortest strided shard
    keep = (per_rank // batch) * batch
    return list(indices[rank::world][:keep])


def allreduce_mean(values, device):
    """mean over the ranks of a list of python floats (identity without a process group)"""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return list(values)
    backend = torch.distributed.get_backend()
    t = torch.tensor(list(values), dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    torch.distributed.all_reduce(t)
    return (t / torch.distributed.get_world_size()).tolist()


def synthetic_source(synth, seed, batch, size, n_critic, device, iters):
    T = torch.from_numpy
    for it in range(iters):
        inp = synth.step_inputs(seed + it, batch, size, n_critic, tag="train")
        refined = [T(r).to(device) for r in inp["refined"]]
        yield (T(inp["pred"]).to(device), T(inp["gt"]).to(device), T(inp["delta_true"]).to(device),
               T(inp["pred_box"]).to(device), lambda delta, k, r=refined: r[k])


def dataset_source(ds, refine_mod, indices, batch, size, device, seed):
    """CalibratorDataset batches entirely on the device (cgan/cgan_train_enhanced.py:262-303 + the re-crop of :313,358):
    shuffled, last partial batch dropped (the engine's buffers are sized for one batch size)."""
    g = torch.Generator().manual_seed(seed)
    order = [indices[i] for i in torch.randperm(len(indices), generator=g).tolist()]
    atlas = ds.atlas(device)
    for b0 in range(0, len(order) - batch + 1, batch):
        pred_patch, gt_patch, delta_true, pred_box, img_idx = ds.gpu_batch(order[b0:b0 + batch], device)
        refine = lambda delta, k, ii=img_idx, pb=pred_box, fp=pred_patch: refine_mod.get_refined_patch_batch(
            atlas, ii, pb, delta, size, fallback_patches=fp)
        yield pred_patch, gt_patch, delta_true, pred_box, refine


def main(argv=None):
    config = load_config()
    args = build_parser(config).parse_args(argv)
    if args.generator_type not in ("unet", "simple"):
        raise SystemExit(f"unknown generator_type {args.generator_type!r} (cgan/cgan_train_enhanced.py:26-31: 'unet' or 'simple')")
    torch.manual_seed(args.seed)
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs an MI355X (the HIP path has no CPU fallback)")
    dist_mod = importlib.import_module(PKG + ".dist")
    rank, world, local = dist_mod.init_from_env()
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    models = importlib.import_module(PKG + ".models")
    engine = importlib.import_module(PKG + ".engine")
    losses = importlib.import_module(PKG + ".losses")
    synth = importlib.import_module(PKG + ".synth")
    if args.generator_type == "simple":                                              # get_generator(), :26-31
        netG = models.GeneratorSimpleRegressor(delta_scale=args.delta_scale)
    else:
        netG = models.GeneratorUNet(delta_scale=args.delta_scale)
    netD = models.Discriminator(spectral_norm=args.spectral_norm)
    netG.apply(models.weights_init_normal); netD.apply(models.weights_init_normal)
    if rank == 0:
        print(f"Generator parameters: {sum(p.numel() for p in netG.parameters()):,}")
        print(f"Discriminator parameters: {sum(p.numel() for p in netD.parameters()):,}")
    eng = engine.StepEngine(netG.state_dict(), netD.state_dict(), batch=args.batch_size // world, size=args.img_size,
                            n_critic=args.n_critic, dtype=args.compute_dtype, device=device, lr=args.lr,
                            betas=(args.beta1, args.beta2), delta_scale=args.delta_scale, lambda_gp=args.lambda_gp,
                            lambda_iou=args.lambda_iou, seed=args.seed + rank,
                            allreduce=dist_mod.GradAverager() if world > 1 else None, generator_type=args.generator_type,
                            spectral_norm=args.spectral_norm)            # (config.yaml `spectral_norm`, cgan/models.py:228-238)
    if world > 1:
        dist_mod.broadcast_state([eng.D.p, eng.G.p] + eng.u + eng.v)
    train_idx, val_idx = None, []
    if args.source == "dataset":
        dataset_mod = importlib.import_module(PKG + ".dataset")
        refine_mod = importlib.import_module(PKG + ".refine")
        ds = dataset_mod.CalibratorDataset(args.data_dir, img_size=args.img_size)
        g = torch.Generator().manual_seed(args.seed)
        perm = torch.randperm(len(ds), generator=g).tolist()
        n_train, _ = split_lengths(len(ds), args.val_split)
        train_idx = shard_indices(perm[:n_train], rank, world, args.batch_size // world)   # :219-231 (train part), sharded by rank
        val_idx = perm[n_train:]                                                     # the validation part (every rank evaluates it: same LR decisions)
        if rank == 0:
            print(f"dataset: {len(ds)} (pred, gt) pairs, {len(train_idx)} for training on this rank, {ds.atlas(device).n} images")
        if len(train_idx) < args.batch_size // world:
            raise SystemExit("fewer training pairs than one batch")
    # ReduceLROnPlateau(mode='max', factor=0.5, patience=5) on delta_iou for both optimisers (:260-261,427-428): torch's own
    # scheduler logic on placeholder optimisers; the resulting rates go to the engine's device-side optimiser state
    sched = [torch.optim.lr_scheduler.ReduceLROnPlateau(torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=args.lr),
                                                        mode="max", factor=0.5, patience=5) for _ in range(2)]
    out_root = Path(args.save_dir); out_root.mkdir(parents=True, exist_ok=True)
    ckpt_best = out_root / "G_best.pth"
    best, history, epochs_no_improve = -1.0, [], 0
    for epoch in range(1, args.n_epochs + 1):
        stats = dict(loss_G=0.0, loss_D=0.0, loss_iou=0.0, loss_wgan=0.0, loss_gp=0.0, wasserstein_distance=0.0)
        n = 0
        iou_b = iou_a = 0.0
        if train_idx is not None:
            source = dataset_source(ds, refine_mod, train_idx, args.batch_size // world, args.img_size, device, args.seed + epoch)
        else:
            source = synthetic_source(synth, args.seed + 1000 * epoch + rank, args.batch_size // world, args.img_size,
                                      args.n_critic, device, args.iters_per_epoch)
        for pred, gt, delta_true, pred_box, refine in source:
            log = eng.iteration(pred, gt, delta_true, pred_box, refine)
            stats["loss_D"] += sum(log["d_loss"]) / args.n_critic
            stats["loss_gp"] += sum(log["gp"]) / args.n_critic
            stats["wasserstein_distance"] += sum(log["wd"]) / args.n_critic
            stats["loss_G"] += log["loss_g"]; stats["loss_iou"] += log["loss_iou"]; stats["loss_wgan"] += log["loss_wgan"]
            n += 1
            # validation metric of :395-420 on the training batch (eval-mode box transform + plain IoU)
            gtb = losses.apply_delta_to_bbox(pred_box, delta_true, training=False)
            cal = losses.apply_delta_to_bbox(pred_box, log["delta_pred"], training=False)
            iou_b += float(losses.iou_metric(pred_box, gtb).mean()); iou_a += float(losses.iou_metric(cal, gtb).mean())
        for k in stats:
            stats[k] /= max(n, 1)
        delta_iou = (iou_a - iou_b) / max(n, 1)
        if world > 1:
            # every rank must take the same scheduler / early-stop / NaN decisions: a rank-local metric would give the replicas
            # different learning rates for the same averaged gradient, a rank-local `break` would leave the others in a collective
            keys = sorted(stats)
            red = allreduce_mean([stats[k] for k in keys] + [delta_iou], device)
            stats = dict(zip(keys, red[:-1])); delta_iou = red[-1]
        Bv = args.batch_size // world
        if train_idx is not None and len(val_idx) >= Bv:
            # validation of :395-420 on the held-out pairs: eval-mode G (dropout off), eval-mode box transform, plain IoU
            sb = sa = 0.0
            nv = 0
            for b0 in range(0, len(val_idx) - Bv + 1, Bv):
                pv, _, dtv, pbv, _ = ds.gpu_batch(val_idx[b0:b0 + Bv], device)
                dv = eng.generator_delta(pv, train=False)
                gtb = losses.apply_delta_to_bbox(pbv, dtv, training=False)
                cal = losses.apply_delta_to_bbox(pbv, dv, training=False)
                sb += float(losses.iou_metric(pbv, gtb).sum()); sa += float(losses.iou_metric(cal, gtb).sum()); nv += Bv
            delta_iou = sa / nv - sb / nv
            if world > 1:
                # every rank evaluated the same held-out pairs with (nominally) the same weights, but float-atomic order can
                # move a last bit: the scheduler / best-checkpoint / early-stop decisions below must be taken on ONE value, or a
                # rank could `break` at a patience tie while its peers enter the next all-reduce
                delta_iou = allreduce_mean([delta_iou], device)[0]
        for sc in sched:
            sc.step(delta_iou)
        eng.set_lr(lr_g=sched[0].optimizer.param_groups[0]["lr"], lr_d=sched[1].optimizer.param_groups[0]["lr"])
        history.append(dict(epoch=epoch, delta_iou=delta_iou, **stats))
        if rank == 0:
            print(f"[Epoch {epoch}/{args.n_epochs}] G: {stats['loss_G']:.3f} D: {stats['loss_D']:.3f} EIoU: {stats['loss_iou']:.3f} "
                  f"WGAN: {stats['loss_wgan']:.3f} GP: {stats['loss_gp']:.3f} WD: {stats['wasserstein_distance']:.3f} "
                  f"dIoU: {delta_iou:.4f}")
        if not all(map(lambda v: v == v and abs(v) != float("inf"), (stats["loss_G"], stats["loss_D"]))):
            print("Warning: NaN or Inf detected in losses! Stopping."); break          # :473-478
        if delta_iou > best + args.min_delta:                        # (delta_iou is identical on every rank)
            best, epochs_no_improve = delta_iou, 0
            if rank == 0:
                gsd, dsd = eng.state_dicts()
                torch.save({"generator": {k: v.cpu() for k, v in gsd.items()},
                            "discriminator": {k: v.cpu() for k, v in dsd.items()},
                            "epoch": epoch, "delta_iou": delta_iou,
                            "config": dict(config, generator_type=args.generator_type, delta_scale=args.delta_scale)},
                           ckpt_best)                                                   # :481-489 (config as built at :194-214)
        else:
            epochs_no_improve += 1                                   # :497-500
            if epochs_no_improve >= args.patience:
                if rank == 0:
                    print("Early stopping triggered.")
                break
    if rank == 0:
        with open(out_root / "training_history.json", "w") as f:
            json.dump(history, f, indent=2)
        print(f"Training complete. Best Delta IoU = {best:.4f}")
    return history


if __name__ == "__main__":
    main()
