#!/usr/bin/env python3
"""Training harness with the reference's surface (cgan/cgan_train_enhanced.py:139-168,256-261,481-489): same
argument names and defaults (read from config.yaml), Adam(lr, (beta1, beta2)) for both nets, the same logged scalars
and the same checkpoint dict keys {'generator','discriminator','epoch','delta_iou','config'} -- driving the MI355X
step engine.  Data: `--source synthetic` (default; the tensor contract of SURVEY §8a row I, no files needed) or
`--source dataset`: CalibratorDataset over --data_dir (YOLO txt + jpg) with the images decoded once into
an HBM atlas, the pred/gt patches and the per-step re-crop cut on the GPU.  Any iterable yielding
(pred_patch, gt_patch, delta_true, pred_box, refine_fn, extra) can feed the loop (extra: what `--graph` stages into the
re-crop stage's static buffers -- GraphLoop).
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time
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
    p.add_argument("--compute_dtype", default="bf16", choices=["bf16", "fp16", "fp32", "fp16x3", "bf16x3"],
                   help="MFMA operand type; fp16x3 / bf16x3: fp32 tensors, split-precision contraction (fp32-grade results)")
    p.add_argument("--graph", action="store_true",
                   help="replay the iteration as captured hipGraphs (engine.GraphedIteration): batches are staged into static "
                        "buffers, the logged scalars stay on the device and are read back once per epoch")
    p.add_argument("--iters_per_epoch", type=int, default=20, help="synthetic source: iterations per epoch")
    p.add_argument("--source", default="synthetic", choices=["synthetic", "dataset"])
    p.add_argument("--synthetic_pool", type=int, default=0,
                   help="synthetic source: generate this many batches once, keep them in HBM and cycle (0: a fresh batch per iteration, built on the host)")
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


_SYNTH_POOL = {}


def synthetic_source(synth, seed, batch, size, n_critic, device, iters, pool=0):
    """pool > 0: `pool` batches are generated once, kept on the device and cycled (the generator below builds every batch with
    numpy on the host, ~0.1 s per batch of 256: fine for a smoke run, 50x slower than the step it feeds)."""
    T = torch.from_numpy

    def make(s):
        inp = synth.step_inputs(s, batch, size, n_critic, tag="train")
        refined = [T(r).to(device) for r in inp["refined"]]
        return (T(inp["pred"]).to(device), T(inp["gt"]).to(device), T(inp["delta_true"]).to(device),
                T(inp["pred_box"]).to(device), lambda delta, k, r=refined: r[k], dict(refined=refined))
    if pool > 0:
        key = (batch, size, n_critic, str(device), pool)
        if key not in _SYNTH_POOL:
            _SYNTH_POOL[key] = [make(seed + j) for j in range(pool)]
        for it in range(iters):
            yield _SYNTH_POOL[key][(seed + it) % pool]
        return
    for it in range(iters):
        yield make(seed + it)


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
        yield pred_patch, gt_patch, delta_true, pred_box, refine, dict(img_idx=img_idx, atlas=atlas)


class GraphLoop:
    """The training loop's inner step as hipGraph replays (``--graph``).  The first batch it sees sizes the static buffers and is
    what the graphs are captured on; every later batch is copied into them (GraphedIteration.replay(batch=..., next_pred=...):
    the pipelined generator forward needs the NEXT batch's pred one replay early, so the loop looks one batch ahead).  The
    re-crop stage is refine.RefineStage on static buffers (dataset source) or the batch's precomputed patches (synthetic
    source).  Nothing is read back per iteration: one small launch appends the iteration's scalars -- per critic step the three
    group means and the gradient penalty (StepEngine.step_log), the generator's EIoU sum and WGAN mean, its predicted deltas
    and the batch's boxes -- to a device-side history that `drain()` turns into the reference's epoch sums
    (cgan/cgan_train_enhanced.py:335-337,372-374,395-420)."""

    ROWS = 64

    def __init__(self, eng, engine_mod, refine_mod, losses, ops, first, size: int, lambda_gp: float, lambda_iou: float):
        pred, gt, delta_true, pred_box, _, extra = first
        self.eng, self.losses, self.ops = eng, losses, ops
        self.lambda_gp, self.lambda_iou = lambda_gp, lambda_iou
        eng.enable_step_log()
        self.inp = tuple(t.detach().float().contiguous().clone() for t in (pred, gt, delta_true, pred_box))
        self.refined = self.stage = None
        if "refined" in extra:
            self.refined = [r.clone() for r in extra["refined"]]
            refine = lambda delta, k: self.refined[k]
        else:
            self.stage = refine_mod.RefineStage(extra["atlas"], extra["img_idx"].clone(), self.inp[3], size, eng.c + 1,
                                                fallback=self.inp[0])
            assert self.stage.pred_box.data_ptr() == self.inp[3].data_ptr()         # (staged with the batch: the same buffer)
            refine = self.stage
        self.gi = engine_mod.GraphedIteration(eng, *self.inp, refine, batch_g_critic=True)
        B, c = eng.B, eng.c
        self.width = 4 * c + 2 + 12 * B
        self.hist = torch.zeros(self.ROWS, self.width, device=pred.device)
        self._rows = [None] * self.ROWS
        self.n = 0
        self.sums = None

    def _append(self):
        j = self.n % self.ROWS
        if self._rows[j] is None:
            eng, B, c, h = self.eng, self.eng.B, self.eng.c, self.hist[j]
            o = 4 * c + 2
            self._rows[j] = self.ops.ReplicaSum(
                [(eng.step_log, h[0:4 * c], 4 * c, False), (eng.eiou_acc, h[4 * c:4 * c + 1], 1, False),
                 (eng.wgan_mean, h[4 * c + 1:o], 1, False), (eng.delta_log, h[o:o + 4 * B], 4 * B, False),
                 (self.inp[3], h[o + 4 * B:o + 8 * B], 4 * B, False), (self.inp[2], h[o + 8 * B:o + 12 * B], 4 * B, False)], 1, 0)
        self._rows[j].run()
        self.n += 1
        if self.n % self.ROWS == 0:
            self._fold(self.ROWS)

    def step(self, batch, nxt):
        """one iteration on `batch`; nxt: the batch after it (or None)"""
        extra = batch[5]
        if self.refined is not None:
            for dst, src in zip(self.refined, extra["refined"]):
                dst.copy_(src, non_blocking=True)
        else:
            self.stage.img_idx.copy_(extra["img_idx"].to(torch.int32), non_blocking=True)
        self.gi.replay(batch=batch[:4], next_pred=None if nxt is None else nxt[0])
        self._append()

    def _fold(self, rows: int):
        """history rows -> running sums of the epoch (one host sync)"""
        if rows == 0:
            return
        eng, L = self.eng, self.losses
        B, c = eng.B, eng.c
        h = self.hist[:rows]
        sl = h[:, :4 * c].view(rows, c, 4).double()
        wd = sl[:, :, 0] - sl[:, :, 1]
        gp = sl[:, :, 3]
        loss_iou = 1.0 + h[:, 4 * c].double()
        loss_wgan = -h[:, 4 * c + 1].double()
        o = 4 * c + 2
        delta, pb, dt = (h[:, o + 4 * B * i:o + 4 * B * (i + 1)].reshape(rows * B, 4) for i in range(3))
        gtb = L.apply_delta_to_bbox(pb, dt, training=False)
        cal = L.apply_delta_to_bbox(pb, delta, training=False)
        iou_b = L.iou_metric(pb, gtb).view(rows, B).double().mean(1).sum()
        iou_a = L.iou_metric(cal, gtb).view(rows, B).double().mean(1).sum()
        vals = torch.stack([(-wd + self.lambda_gp * gp).mean(1).sum(), gp.mean(1).sum(), wd.mean(1).sum(),
                            (self.lambda_iou * loss_iou + loss_wgan).sum(), loss_iou.sum(), loss_wgan.sum(), iou_b, iou_a]).tolist()
        self.sums = vals if self.sums is None else [a + b for a, b in zip(self.sums, vals)]

    def drain(self):
        """-> (dict of the epoch's sums, iterations) and reset"""
        self._fold(self.n % self.ROWS)
        keys = ("loss_D", "loss_gp", "wasserstein_distance", "loss_G", "loss_iou", "loss_wgan", "iou_before", "iou_after")
        out = dict(zip(keys, self.sums if self.sums is not None else [0.0] * len(keys)))
        if self.gi._g_owed:
            # batch_g_critic: a replay's history row carries the PREVIOUS iteration's WGAN term (0 in the first row after a
            # finish()); the last iteration's arrives with the forward finish() runs.  Only the epoch sums are used, so:
            self.gi.finish()
            w = -float(self.eng.wgan_mean)
            out["loss_wgan"] += w; out["loss_G"] += w
        n, self.n, self.sums = self.n, 0, None
        return out, n


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
    loop = None                                                      # --graph: the captured iteration + its static buffers (built on the first batch)
    for epoch in range(1, args.n_epochs + 1):
        stats = dict(loss_G=0.0, loss_D=0.0, loss_iou=0.0, loss_wgan=0.0, loss_gp=0.0, wasserstein_distance=0.0)
        n = 0
        t_epoch = time.perf_counter()
        iou_b = iou_a = 0.0
        if train_idx is not None:
            source = dataset_source(ds, refine_mod, train_idx, args.batch_size // world, args.img_size, device, args.seed + epoch)
        else:
            source = synthetic_source(synth, args.seed + 1000 * epoch + rank, args.batch_size // world, args.img_size,
                                      args.n_critic, device, args.iters_per_epoch, pool=args.synthetic_pool)
        if args.graph:
            # the first iteration of the run is eager (it also finishes the library's lazy set-up, which must not happen inside a
            # capture); from the second on, graph replays with a one-batch look-ahead
            it = iter(source)
            cur = next(it, None)
            if cur is not None and loop is None:
                source = [cur]                                     # (the eager loop below takes this one batch)
            else:
                source = []
        for pred, gt, delta_true, pred_box, refine, _extra in source:
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
        if args.graph:
            if loop is None and cur is not None:
                loop = GraphLoop(eng, engine, importlib.import_module(PKG + ".refine"), losses, importlib.import_module(PKG + ".ops"),
                                 cur, args.img_size, args.lambda_gp, args.lambda_iou)
                cur = next(it, None)                               # (the first batch went through the eager iteration above)
            while cur is not None:
                nxt = next(it, None)
                loop.step(cur, nxt)
                cur = nxt
            if loop is not None:
                sums, ng = loop.drain()
                for k in stats:
                    stats[k] += sums[k]
                iou_b += sums["iou_before"]; iou_a += sums["iou_after"]; n += ng
        torch.cuda.synchronize()
        train_seconds = time.perf_counter() - t_epoch               # the training iterations of the epoch (validation not included)
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
        history.append(dict(epoch=epoch, delta_iou=delta_iou, train_seconds=train_seconds, iterations=n, **stats))
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
