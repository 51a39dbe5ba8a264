#!/usr/bin/env python3
"""bench.py -- images/s of the full cGAN WGAN-GP iteration (n_critic critic steps + 1 generator step, optimiser steps
and gradient all-reduces included) on synthetic 32x32x3 batches, BASELINE.json configs[1]: B=256 per GPU, bf16 MFMA.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line (rank 0).  See DESIGN.md "Measurement" for how every field is obtained.
"""
from __future__ import annotations

import argparse
import importlib
import json
import math
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
PKG = "gan-calibrated-semi-supervised-learning_amd"

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (about 6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}      # dense, /opt/skills/guides/MI355X_MICROARCH.md:42-43
F_D = {32: 0.0535e9, 64: 0.2141e9, 128: 0.8564e9}       # forward FLOPs / image
F_G = {32: 0.2029e9, 64: 0.8116e9, 128: 3.2464e9}


def synthetic_inputs(synth, seed, B, S, c, dev, gtype="unet"):
    T = torch.from_numpy
    inp = synth.step_inputs(seed, B, S, c, tag="bench", generator_type=gtype)
    return dict(pred=T(inp["pred"]).to(dev), gt=T(inp["gt"]).to(dev), delta_true=T(inp["delta_true"]).to(dev),
                pred_box=T(inp["pred_box"]).to(dev), refined=[T(r).to(dev) for r in inp["refined"]]), inp


def cpu_baseline(synth, seed, B, S, c, budget_s=25.0, gtype="unet"):
    """The CPU oracle (oracle/cgan_oracle.py, a port pinned to the reference's golden vectors) on the host cores.
    torch's CPU ops stop scaling long before 128 threads on this step (B=256 convs of 2x2..16x16 maps), so the thread count
    is chosen first: one B=64 iteration per candidate in {8, 16, 32, 64}, the fastest runs the timed sample."""
    from oracle import cgan_oracle as O
    T = torch.from_numpy
    g = {k: T(v) for k, v in (synth.simple_generator_state(seed) if gtype == "simple" else synth.generator_state(seed)).items()}
    d = {k: T(v) for k, v in synth.discriminator_state(seed).items()}

    def make(b):
        inp = synth.step_inputs(seed, b, S, c, tag="bench", generator_type=gtype)
        orc = O.StepOracle(g, d, n_critic=c, generator_type=gtype)
        refined = [T(r) for r in inp["refined"]]
        args = (T(inp["pred"]), T(inp["gt"]), T(inp["delta_true"]), T(inp["pred_box"]), lambda dl, k: refined[k],
                [T(a) for a in inp["alpha"]], [[T(m) for m in ms] for ms in inp["masks"]])
        return orc, args
    ncpu = os.cpu_count() or 1
    t_all = time.perf_counter()
    cands = [n for n in (8, 16, 32, 64) if n <= ncpu] or [ncpu]   # (all 256 hyper-threads of the GPU box: 211 s per iteration)
    sweep = {}
    orc, args = make(min(B, 64))
    for n in cands:
        torch.set_num_threads(n)
        t0 = time.perf_counter(); orc.iteration(*args); first = time.perf_counter() - t0   # warm-up (thread pool, allocator)
        if sweep and first > 3.0 * min(sweep.values()):                  # already far slower than the best so far: stop climbing
            break
        t0 = time.perf_counter(); orc.iteration(*args); sweep[n] = time.perf_counter() - t0
        if time.perf_counter() - t_all > 0.3 * budget_s:
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    orc, args = make(B)
    t0 = time.perf_counter(); orc.iteration(*args); warm = time.perf_counter() - t0
    n, t0 = 0, time.perf_counter()
    while True:
        orc.iteration(*args); n += 1
        el = time.perf_counter() - t0
        if (time.perf_counter() - t_all) + el / n > budget_s or n >= 20:
            break
    return dict(value=B * n / el, unit="images/s", cores=best, kind="port", host_cpus=ncpu,
                thread_sweep_ms_per_iter_B64={str(k): round(v * 1e3, 1) for k, v in sweep.items()},
                sample=f"{n} timed iterations (+1 warm-up, {warm:.1f} s) of the same B={B}, {S}x{S}, n_critic={c} synthetic step, "
                       f"fp32, oracle/cgan_oracle.StepOracle on torch CPU ops with {best} threads (fastest of {sorted(sweep)} on "
                       f"a B={min(B, 64)} iteration)")


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start N fresh ranks (one per GPU) BEFORE this process touches a GPU,
    relay rank 0's JSON line, fail if any rank fails.  (The parent never initialises HIP: device_count() does not.)"""
    dist_mod = importlib.import_module(PKG + ".dist")
    have = torch.cuda.device_count()
    if have < args.gpus and not os.environ.get("GCSSL_SINGLE_DEVICE"):
        print(f"[bench] --gpus {args.gpus} but this node exposes {have} GPU(s): refusing to report a {args.gpus}-GPU number",
              file=sys.stderr)
        return 2
    codes, out0 = dist_mod.launch_local_ranks([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], args.gpus)
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    if any(c != 0 for c in codes) or not lines:
        print(f"[bench] rank exit codes {codes}; no result reported", file=sys.stderr)
        return 1
    print(lines[-1])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--n_critic", type=int, default=2)
    ap.add_argument("--dtype", default=os.environ.get("GCSSL_BENCH_DTYPE", "fp16"), choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--generator", default="unet", choices=["unet", "simple"],
                    help="generator_type (cgan/cgan_train_enhanced.py:26-31); the headline config is the default U-Net")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--probe-steps", type=int, default=5)
    ap.add_argument("--sustain-s", type=float, default=2.0,
                    help="length of the second, sustained timing window (0 = skip); reported beside the --steps window")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))                # N fresh child ranks; nothing below runs in this process

    dist_mod = importlib.import_module(PKG + ".dist")
    # RCCL prints a version banner on STDOUT when its communicator is created: keep stdout to the one JSON line by pointing
    # fd 1 at stderr until the process group and the communicator (first collective) exist
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        rank, world, local = dist_mod.init_from_env()
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                             f"(python bench.py --gpus N starts them itself)")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        if torch.distributed.is_initialized():
            torch.distributed.all_reduce(torch.zeros(1, device=dev))
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    rccl_ranks = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    synth = importlib.import_module(PKG + ".synth")
    engine = importlib.import_module(PKG + ".engine")
    B, S, c = args.batch, args.size, args.n_critic
    T = torch.from_numpy
    g = {k: T(v) for k, v in (synth.simple_generator_state(42) if args.generator == "simple" else synth.generator_state(42)).items()}
    d = {k: T(v) for k, v in synth.discriminator_state(42).items()}
    averager = dist_mod.GradAverager() if (world > 1 or dist_mod.force_dp()) else None
    eng = engine.StepEngine(g, d, batch=B, size=S, n_critic=c, dtype=args.dtype, device=dev, seed=42 + rank,
                            allreduce=averager, keep_clipped_grads=False,
                            overlap=int(os.environ.get("GCSSL_OVERLAP", "0")), generator_type=args.generator)
    data, _ = synthetic_inputs(synth, 42 + rank, B, S, c, dev, args.generator)          # resident in HBM before anything is timed
    refine = lambda delta, k: data["refined"][k]
    call = (data["pred"], data["gt"], data["delta_true"], data["pred_box"], refine)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (eager), then capture
    for _ in range(max(1, args.warmup)):
        eng.run_iteration(*call)
    torch.cuda.synchronize()
    graphed = None
    if not args.no_graph:
        try:
            graphed = engine.GraphedIteration(eng, *call)
            graphed.replay(); torch.cuda.synchronize()
        except Exception as e:                                           # report, then fall back to eager launches
            print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            graphed = None
    step = graphed.replay if graphed is not None else (lambda: eng.run_iteration(*call))

    def timed(nsteps):
        """exactly nsteps iterations between barrier+sync on both sides, max over ranks -> seconds"""
        barrier()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t)
        return el

    # ---- timed region
    el = timed(args.steps)
    ms = el / args.steps * 1e3
    value = world * B * args.steps / el
    # ---- a second, sustained window (clocks and thermals settle; long enough for an external GPU-busy sampler to see it)
    sustained = None
    if args.sustain_s > 0:
        n2 = max(args.steps, int(math.ceil(args.sustain_s / (ms * 1e-3))))
        if world > 1:                                                    # every rank must run the same count
            t = torch.tensor([n2], device=dev, dtype=torch.int64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            n2 = int(t)
        el2 = timed(n2)
        sustained = dict(steps=n2, seconds=round(el2, 3), ms_per_step=round(el2 / n2 * 1e3, 4),
                         images_per_s=round(world * B * n2 / el2, 1))

    # ---- sanity of the state the timed replays left behind: finite weights, finite last critic loss
    m = eng.means.tolist()
    d_loss_last = -(m[0] - m[1]) + eng.lambda_gp * float(eng.gp_sum)
    finite = bool(torch.isfinite(eng.D.p).all()) and bool(torch.isfinite(eng.G.p).all()) and math.isfinite(d_loss_last)
    if world > 1:
        t = torch.tensor([0 if finite else 1], device=dev, dtype=torch.int64)
        torch.distributed.all_reduce(t)
        finite = int(t) == 0
    if not finite:
        raise SystemExit(f"[bench] non-finite state after the timed region (last d_loss {d_loss_last}): result invalid")

    # ---- roofline of the dominant kernel: HIP events around every MFMA conv launch, same buffers, eager launches
    eng.enable_probe(True)
    for _ in range(args.probe_steps):
        eng.run_iteration(*call)
    prof = eng.probe_summary()
    eng.enable_probe(False)
    tot = {k: v[0] * v[1] for k, v in prof.items()}
    if os.environ.get("GCSSL_BENCH_VERBOSE") and rank == 0:
        for k in sorted(tot, key=tot.get, reverse=True):
            n, t, f, nb, st = prof[k]
            print(f"[probe] {k:24s} {n // args.probe_steps:3d}/iter  {t * 1e3:8.1f} us  {f / (t * 1e-3) / 1e12:8.1f} TF/s  "
                  f"{f / 1e9:7.2f} GF  algo {nb / 1e6:7.1f} MB  stored {st / 1e6:7.1f} MB  {nb / (t * 1e-3) / 1e9:7.0f} GB/s", file=sys.stderr)
    dom = max(tot, key=tot.get)
    n_dom, ms_dom, fl_dom, by_dom, st_dom = prof[dom]
    conv_ms = sum(tot.values()) / args.probe_steps
    conv_flops = sum(v[0] * v[2] for v in prof.values()) / args.probe_steps
    d_ms = sum(v for k, v in tot.items() if k.startswith("D.")) / args.probe_steps
    d_flops = sum(v[0] * v[2] for k, v in prof.items() if k.startswith("D.")) / args.probe_steps
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    ach = fl_dom / (ms_dom * 1e-3) / 1e12
    ach_gbs = by_dom / (ms_dom * 1e-3) / 1e9
    # the roof is a property of the layer (MFMA for every layer with >= 64 input channels, HBM for the 8-channel
    # first layers); algorithmic bytes = input + output + weights once, in the compute dtype (engine._algorithmic_bytes)
    hbm_bound = engine.roofline_bound(dom) == "hbm"
    # HBM bytes per launch of the dominant kernel: not measurable inside this run (PMC counters need rocprofv3); taken from
    # the committed rocprofv3 --pmc passes of this build (FETCH_SIZE x2-corrected + WRITE_SIZE, tools/pmc_traffic.sh) when
    # this run is the configuration they were taken on, else null
    traffic = traffic_src = None
    pmc = ROOT / "profiles" / "round2_pmc_dominant.json"
    if pmc.exists():
        rec = json.loads(pmc.read_text())
        lab = rec.get("labels", {}).get(dom)
        if lab and lab.get("hbm_bytes_per_launch") and rec.get("config") == [B, S, c, args.dtype, args.generator]:
            traffic, traffic_src = lab["hbm_bytes_per_launch"], f"profiles/{pmc.name} (committed rocprofv3 --pmc passes of this label's launch shape, not this run)"
    roofline = dict(bound="hbm" if hbm_bound else "mfma", kernel=dom,
                    achieved=round(ach_gbs if hbm_bound else ach, 2), peak=HBM_PEAK_GBS if hbm_bound else peak,
                    unit="GB/s" if hbm_bound else "TFLOP/s", frac=round(ach_gbs / HBM_PEAK_GBS if hbm_bound else ach / peak, 4),
                    traffic=traffic, traffic_source=traffic_src, launches=n_dom, avg_us=round(ms_dom * 1e3, 2),
                    algorithmic=dict(flops=fl_dom, bytes=by_dom, stored_bytes=st_dom, tflops=round(ach, 2), gbs=round(ach_gbs, 1),
                                     mfma_frac=round(ach / peak, 4), hbm_frac=round(ach_gbs / HBM_PEAK_GBS, 4)),
                    all_convs=dict(tflops=round(conv_flops / (conv_ms * 1e-3) / 1e12, 2), ms_per_iter=round(conv_ms, 3),
                                   frac=round(conv_flops / (conv_ms * 1e-3) / 1e12 / peak, 4)),
                    # MFMA utilisation of the critic's conv stack (every D.* conv launch of an iteration)
                    d_convs=dict(tflops=round(d_flops / (d_ms * 1e-3) / 1e12, 2), ms_per_iter=round(d_ms, 3),
                                 frac=round(d_flops / (d_ms * 1e-3) / 1e12 / peak, 4)))
    flop_iter = ((12 * c + 1) * F_D[S] + (c + 3) * F_G[S]) * B * world if (S in F_D and args.generator == "unet") else None

    out = dict(metric="images/sec (G+D step)", value=round(value, 1), unit="images/s", n_gpus=world, rccl_ranks=rccl_ranks,
               steps=args.steps, warmup=args.warmup, ms_per_step=round(ms, 4), higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype=args.dtype, data="synthetic",
               sustained_ms_per_step=sustained["ms_per_step"] if sustained else None, sustained=sustained,
               finite_after_run=finite, last_d_loss=round(d_loss_last, 6),
               config=dict(workload=f"cGAN WGAN-GP iteration (n_critic={c} critic steps + 1 generator step), "
                                    f"{S}x{S}x3, batch {B}/GPU, reference G ({'U-Net' if args.generator == 'unet' else 'GeneratorSimpleRegressor'}) "
                                    f"+ D (SN PatchGAN)",
                           global_batch=B * world, img_size=S, n_critic=c, parallelism=f"dp{world}",
                           launch="hipGraph replay" if graphed is not None else "eager",
                           algorithmic_tflops=round(flop_iter / (ms * 1e-3) / 1e12, 2) if flop_iter else None),
               roofline=roofline)
    mode_err = ROOT / "profiles" / "round2_mode_error.json"            # measured by tools/mode_error.py on the GPU box
    if mode_err.exists():
        rec = json.loads(mode_err.read_text())
        out["mode_error"] = rec.get("modes", {}).get(args.dtype)
        out["parity_mode_images_per_s"] = rec.get("parity_mode_images_per_s")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(synth, 42, B, S, c, gtype=args.generator)
    if rank == 0:
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
