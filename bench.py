#!/usr/bin/env python3
"""bench.py -- images/s of the full cGAN WGAN-GP iteration (n_critic critic steps + 1 generator step, optimiser steps
and gradient all-reduces included) on synthetic 32x32x3 batches.  The headline is BASELINE.json configs[1] as named:
B=256 per GPU, 32x32, bf16 MFMA operands with fp32 accumulation.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line (rank 0).  Besides the headline it carries, all MEASURED BY THIS RUN (single-GPU runs only):
  * "also": the same iteration in the other BASELINE / north_star configurations, each with its own bounded timed window --
    fp16 at B=256 32x32 (the 16-bit mode with the smaller error), fp16 B=512 32x32 (configs[3]), fp16 B=128 64x64
    (configs[4] / north_star's 64x64), and the fp32-MFMA parity mode at B=256 32x32;
  * "mode_error": the error of the 16-bit modes against the pinned CPU oracle on the bench configuration, taken against the
    oracle iteration the cpu_baseline leg runs anyway;
  * "roofline": the dominant conv launch (HIP events on the launch stream) with the kernel it ran; only `traffic` (HBM bytes
    from PMC counters, which need rocprofv3) comes from a committed file and says so in `traffic_source`;
  * "saturations": fp16 gradient stores that clipped at +-65504 during the whole run (0 for bf16 / fp32).
See DESIGN.md "Measurement" for how every field is obtained.
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
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3,       # dense, /opt/skills/guides/MI355X_MICROARCH.md:42-43
                    # split-precision modes: every algorithmic multiply-add is THREE 16-bit MFMA multiply-adds (hi*hi + lo*hi +
                    # hi*lo), so the roof of the algorithmic FLOP rate is a third of the 16-bit pipe's
                    "fp16x3": 2500.0 / 3, "bf16x3": 2500.0 / 3}
F_D = {32: 0.0535e9, 64: 0.2141e9, 128: 0.8564e9}       # forward FLOPs / image
F_G = {32: 0.2029e9, 64: 0.8116e9, 128: 3.2464e9}
T = torch.from_numpy


def synthetic_inputs(synth, seed, B, S, c, dev, gtype="unet"):
    inp = synth.step_inputs(seed, B, S, c, tag="bench", generator_type=gtype)
    return dict(pred=T(inp["pred"]).to(dev), gt=T(inp["gt"]).to(dev), delta_true=T(inp["delta_true"]).to(dev),
                pred_box=T(inp["pred_box"]).to(dev), refined=[T(r).to(dev) for r in inp["refined"]]), inp


def initial_state(synth, gtype):
    g = {k: T(v) for k, v in (synth.simple_generator_state(42) if gtype == "simple" else synth.generator_state(42)).items()}
    d = {k: T(v) for k, v in synth.discriminator_state(42).items()}
    return g, d


def cpu_baseline(synth, seed, B, S, c, budget_s=25.0, gtype="unet"):
    """The CPU oracle (oracle/cgan_oracle.py, a port pinned to the reference's golden vectors) on the host cores.
    torch's CPU ops stop scaling long before 128 threads on this step (B=256 convs of 2x2..16x16 maps), so the thread count
    is chosen first: one B=64 iteration per candidate in {8, 16, 32, 64}, the fastest runs the timed sample.
    -> (record, reference): `reference` = (inputs, log, taps) of the first full-size oracle iteration from the initial weights,
    which measure_mode_error() compares the HIP engine with (the checker role of the oracle; nothing timed uses it)."""
    from oracle import cgan_oracle as O
    g, d = initial_state(synth, gtype)

    def make(b):
        inp = synth.step_inputs(seed, b, S, c, tag="bench", generator_type=gtype)
        orc = O.StepOracle(g, d, n_critic=c, generator_type=gtype)
        refined = [T(r) for r in inp["refined"]]
        args = (T(inp["pred"]), T(inp["gt"]), T(inp["delta_true"]), T(inp["pred_box"]), lambda dl, k: refined[k],
                [T(a) for a in inp["alpha"]], [[T(m) for m in ms] for ms in inp["masks"]])
        return orc, args, inp
    ncpu = os.cpu_count() or 1
    t_all = time.perf_counter()
    cands = [n for n in (8, 16, 32, 64) if n <= ncpu] or [ncpu]   # (all 256 hyper-threads of the GPU box: 211 s per iteration)
    sweep = {}
    orc, args, _ = make(min(B, 64))
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
    orc, args, inp = make(B)
    taps = {}
    t0 = time.perf_counter(); ref = orc.iteration(*args, taps=taps); warm = time.perf_counter() - t0
    n, t0 = 0, time.perf_counter()
    while True:
        orc.iteration(*args); n += 1
        el = time.perf_counter() - t0
        if (time.perf_counter() - t_all) + el / n > budget_s or n >= 20:
            break
    rec = dict(value=B * n / el, unit="images/s", cores=best, kind="port", host_cpus=ncpu,
               thread_sweep_ms_per_iter_B64={str(k): round(v * 1e3, 1) for k, v in sweep.items()},
               sample=f"{n} timed iterations (+1 warm-up, {warm:.1f} s) of the same B={B}, {S}x{S}, n_critic={c} synthetic step, "
                      f"fp32, oracle/cgan_oracle.StepOracle on torch CPU ops with {best} threads (fastest of {sorted(sweep)} on "
                      f"a B={min(B, 64)} iteration)")
    return rec, (inp, ref, taps)


def measure_mode_error(engine, synth, reference, dtype, B, S, c, dev, gtype):
    """One eager iteration of a fresh engine in compute mode `dtype` on the oracle's fixture (same weights, inputs, alphas,
    dropout masks) against the oracle iteration of the cpu_baseline leg: max-norm relative error of the critic's scores and
    G's delta, relative error of the scalars; the second critic step sits behind an Adam update (recorded)."""
    import numpy as np
    inp, ref, taps = reference
    g, d = initial_state(synth, gtype)
    eng = engine.StepEngine(g, d, batch=B, size=S, n_critic=c, dtype=dtype, device=dev, generator_type=gtype)
    refined = [T(r).to(dev) for r in inp["refined"]]
    log = eng.iteration(T(inp["pred"]).to(dev), T(inp["gt"]).to(dev), T(inp["delta_true"]).to(dev), T(inp["pred_box"]).to(dev),
                        lambda dl, k: refined[k], alphas=[T(a).to(dev).view(-1).contiguous() for a in inp["alpha"]],
                        masks=[[T(m).to(dev) for m in ms] for ms in inp["masks"]])
    torch.cuda.synchronize()

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
    sgn = lambda a, b: float((a - b) / abs(b))
    r6 = lambda v: float(f"{v:.4g}")
    out = dict(scores=r6(max(rel(log["real"][0].cpu().reshape(-1), taps["real_validity"].reshape(-1)),
                             rel(log["fake"][0].cpu().reshape(-1), taps["fake_validity"].reshape(-1)))),
               delta=r6(rel(log["delta_pred"].cpu(), ref["delta_pred"])),
               wd=r6(abs(sgn(log["wd"][0], ref["wd"][0]))), gp=r6(abs(sgn(log["gp"][0], ref["gp"][0]))),
               d_grad_norm=r6(sgn(log["d_grad_norm"][0], ref["d_grad_norm"][0])),
               loss_iou=r6(abs(sgn(log["loss_iou"], ref["loss_iou"]))), g_grad_norm=r6(sgn(log["g_grad_norm"], ref["g_grad_norm"])),
               step2_gp=r6(abs(sgn(log["gp"][1], ref["gp"][1]))), step2_d_grad_norm=r6(sgn(log["d_grad_norm"][1], ref["d_grad_norm"][1])),
               saturations=eng.saturations())
    del eng
    return out


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start N fresh ranks (one per GPU) BEFORE this process touches a GPU,
    relay rank 0's JSON line, fail if any rank fails.  (The parent never initialises HIP: device_count() does not.)"""
    dist_mod = importlib.import_module(PKG + ".dist")
    have = torch.cuda.device_count()
    if have < args.gpus and not os.environ.get("GCSSL_SINGLE_DEVICE"):
        print(f"[bench] --gpus {args.gpus} but this node exposes {have} GPU(s): refusing to report a {args.gpus}-GPU number",
              file=sys.stderr)
        return 2
    codes, out0 = dist_mod.launch_local_ranks([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], args.gpus,
                                              timeout=float(os.environ.get("GCSSL_LAUNCH_TIMEOUT", "1500")))
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    if any(c != 0 for c in codes) or not lines:
        print(f"[bench] rank exit codes {codes}; no result reported", file=sys.stderr)
        return 1
    print(lines[-1])
    return 0


class Runner:
    """One configuration of the iteration on this rank's GPU: engine, resident synthetic inputs, hipGraph capture, timing."""

    def __init__(self, engine, synth, dist_mod, dev, rank, world, B, S, c, dtype, gtype, warmup, no_graph=False, recrop=None):
        self.engine, self.dev, self.rank, self.world = engine, dev, rank, world
        self.B, self.S, self.c, self.dtype, self.gtype = B, S, c, dtype, gtype
        g, d = initial_state(synth, gtype)
        averager = dist_mod.GradAverager() if (world > 1 or dist_mod.force_dp()) else None
        self.eng = engine.StepEngine(g, d, batch=B, size=S, n_critic=c, dtype=dtype, device=dev, seed=42 + rank,
                                     allreduce=averager, keep_clipped_grads=False, generator_type=gtype)
        data, _ = synthetic_inputs(synth, 42 + rank, B, S, c, dev, gtype)          # resident in HBM before anything is timed
        refine = lambda delta, k: data["refined"][k]
        if recrop:
            # the re-crop stage (cgan/cgan_train_enhanced.py:37-137) INSIDE the iteration: n_critic + 1 calls of
            # refine.RefineStage per iteration -- eval-mode box transform + Pillow-exact crop / letterbox / BICUBIC resize from an
            # HBM-resident atlas of synthetic source images (recrop = (n_images, width, height)), captured into the graphs
            import numpy as np
            rf = importlib.import_module(PKG + ".refine")
            ni, w, h = recrop
            rng = np.random.default_rng(1234 + rank)
            atlas = rf.ImageAtlas([rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for _ in range(ni)], dev)
            idx = torch.from_numpy(rng.integers(0, ni, B).astype(np.int32)).to(dev)
            self.refine_stage = rf.RefineStage(atlas, idx, data["pred_box"], S, c + 1, fallback=data["pred"])
            refine = self.refine_stage
        self.call = (data["pred"], data["gt"], data["delta_true"], data["pred_box"], refine)
        for _ in range(max(1, warmup)):                                             # warm-up (eager), then capture
            self.eng.run_iteration(*self.call)
        torch.cuda.synchronize()
        self.graphed = None
        if not no_graph:
            # a failed capture is an ERROR (the caller exits non-zero): an eager number under a line that says "hipGraph replay"
            # by default would be a silent downgrade.  --no-graph asks for the eager form explicitly.
            # batch_g_critic: iteration i's value-only critic forward as a fourth group of iteration i+1's first critic forward
            # (GraphedIteration docstring; every replay still runs exactly one such forward -- the previous iteration's)
            self.graphed = engine.GraphedIteration(self.eng, *self.call, batch_g_critic=True)
            self.graphed.replay(); torch.cuda.synchronize()
        self.step = self.graphed.replay if self.graphed is not None else (lambda: self.eng.run_iteration(*self.call))

    def barrier(self):
        if self.world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(self, nsteps):
        """exactly nsteps iterations between barrier+sync on both sides, max over ranks -> seconds"""
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            self.step()
        self.barrier()
        el = time.perf_counter() - t0
        if self.world > 1:
            t = torch.tensor([el], device=self.dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t)
        return el

    def finite(self):
        """sanity of the state the timed replays left behind: finite weights, finite last critic loss"""
        eng = self.eng
        if self.graphed is not None:
            self.graphed.finish()                                  # (the value-only forward the last replay still owes)
        m = eng.means.tolist()
        d_loss_last = -(m[0] - m[1]) + eng.lambda_gp * float(eng.gp_sum)
        ok = bool(torch.isfinite(eng.D.p).all()) and bool(torch.isfinite(eng.G.p).all()) and math.isfinite(d_loss_last)
        if self.world > 1:
            t = torch.tensor([0 if ok else 1], device=self.dev, dtype=torch.int64)
            torch.distributed.all_reduce(t)
            ok = int(t) == 0
        return ok, d_loss_last

    def probe(self, probe_steps, verbose=False):
        """HIP events around every MFMA conv launch, same buffers, eager launches -> per-label table + aggregates.
        Every launch is timed two ways in the same pass (engine.enable_probe): ALONE between two events -- behind the idle gap
        of an eager launch and an event packet, which costs a 10-40 us kernel 3-10 us that no graph replay pays -- and as
        PROBE_REPEATS further back-to-back launches between two events, i.e. as one link of a chain of launches, which is
        what it is inside the replayed graph (rocprofv3's per-kernel durations of the graph replay agree with this one).
        The table's `t` / the aggregates use the chained time; the lone-launch time rides along as *_single."""
        eng = self.eng
        if self.graphed is not None:
            self.graphed.finish()
        eng.enable_probe(True, repeats=PROBE_REPEATS)
        for _ in range(probe_steps):
            eng.run_iteration(*self.call)
        raw = eng.probe_summary()
        self.grids = dict(eng.probe_grids)
        eng.enable_probe(False)
        prof = {k: (v[0], v[6] if v[6] is not None else v[1], v[2], v[3], v[4], v[5], v[1]) for k, v in raw.items()}
        tot = {k: v[0] * v[1] for k, v in prof.items()}
        if verbose and self.rank == 0:
            for k in sorted(tot, key=tot.get, reverse=True):
                n, t, f, nb, st, kern, t1 = prof[k]
                print(f"[probe] {k:24s} {n // probe_steps:3d}/iter  {t * 1e3:8.1f} us (alone {t1 * 1e3:6.1f})  {f / (t * 1e-3) / 1e12:8.1f} TF/s  "
                      f"{f / 1e9:7.2f} GF  algo {nb / 1e6:7.1f} MB  stored {st / 1e6:7.1f} MB  {nb / (t * 1e-3) / 1e9:7.0f} GB/s  {kern}",
                      file=sys.stderr)
        peak = MFMA_PEAK_TFLOPS[self.dtype]

        def agg(pred):
            fl = sum(v[0] * v[2] for k, v in prof.items() if pred(k)) / probe_steps
            out = {}
            for name, col in (("", 1), ("_single", 6)):
                ms = sum(v[0] * v[col] for k, v in prof.items() if pred(k)) / probe_steps
                tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
                out.update({"tflops" + name: round(tf, 2), "ms_per_iter" + name: round(ms, 3), "frac" + name: round(tf / peak, 4)})
            return out
        # the critic's MFMA-bound launches alone (every layer with >= 64 input channels; the 8-channel first
        # layers are HBM-bound and are priced against that roof when one of them is the dominant launch)
        d_mfma = agg(lambda k: k.startswith("D.") and engine_mod_roofline_bound(k) == "mfma")
        d = agg(lambda k: k.startswith("D."))
        d["mfma_bound_launches"] = d_mfma
        return prof, tot, agg(lambda k: True), d


PROBE_REPEATS = 4


def engine_mod_roofline_bound(label):
    return importlib.import_module(PKG + ".engine").roofline_bound(label)


def dtype_symbol(dtype):
    return {"bf16": "__bf16", "fp16": "_Float16", "fp32": "float", "fp16x3": "float", "bf16x3": "float"}[dtype]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (weak scaling)")
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--n_critic", type=int, default=2)
    ap.add_argument("--dtype", default=os.environ.get("GCSSL_BENCH_DTYPE", "bf16"), choices=["bf16", "fp16", "fp32", "fp16x3", "bf16x3"],
                    help="MFMA operand type of the headline (BASELINE configs[1] names bf16); fp16x3 / bf16x3: fp32 tensors, operands "
                         "split hi + lo inside the conv kernels, 3 MFMAs per K step (the parity-grade throughput modes)")
    ap.add_argument("--generator", default="unet", choices=["unet", "simple"],
                    help="generator_type (cgan/cgan_train_enhanced.py:26-31); the headline config is the default U-Net")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the other configurations (the `also` array)")
    ap.add_argument("--also-s", type=float, default=2.0, help="length of each `also` configuration's timed window")
    ap.add_argument("--probe-steps", type=int, default=5)
    ap.add_argument("--preroll-s", type=float, default=0.5, help="untimed replays in front of the timed region (clock ramp)")
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
        if os.environ.get("GCSSL_MAIN_PRIO"):              # A/B knob: the rank's compute stream at another priority (-1 = high)
            torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=int(os.environ["GCSSL_MAIN_PRIO"])))
            print(f"[bench] main stream priority {torch.cuda.current_stream().priority} of range {torch.cuda.Stream.priority_range()}", file=sys.stderr)
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
    verbose = bool(os.environ.get("GCSSL_BENCH_VERBOSE"))

    run = Runner(engine, synth, dist_mod, dev, rank, world, B, S, c, args.dtype, args.generator, args.warmup, args.no_graph)
    # ---- untimed pre-roll: a fresh box ramps its clocks over the first few hundred ms of load (measured: the first 70-ms window
    # right behind the capture read 90k images/s on a box whose next windows all read 118k); the W eager warm-up iterations of
    # the contract are done (Runner), these replays are more of the same and are reported as `preroll_steps`
    preroll = 0
    if args.preroll_s > 0:
        e0 = run.timed(3)
        preroll = max(3, int(args.preroll_s / (e0 / 3)))
        if world > 1:
            t = torch.tensor([preroll], device=dev, dtype=torch.int64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            preroll = int(t)
        run.timed(preroll)
        preroll += 3
    # ---- timed region
    el = run.timed(args.steps)
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
        el2 = run.timed(n2)
        sustained = dict(steps=n2, seconds=round(el2, 3), ms_per_step=round(el2 / n2 * 1e3, 4),
                         images_per_s=round(world * B * n2 / el2, 1))
    finite, d_loss_last = run.finite()
    if not finite:
        raise SystemExit(f"[bench] non-finite state after the timed region (last d_loss {d_loss_last}): result invalid")

    sat = run.eng.saturations()                    # (before the probe pass: its repeated launches leave the engine's state scrap)
    # ---- roofline of the dominant kernel
    prof, tot, all_convs, d_convs = run.probe(args.probe_steps, verbose)
    # ---- in-graph durations.  The probe above times every label in an EAGER serial pass; inside the replayed
    # graphs the two chains run side by side and a launch can take longer (round 3: G.up4.fwd 69.9 us in the probe, 79.3 us in
    # the rocprofv3 trace of the replay).  profiles/round4_label_durations.json holds, per label, the average duration of its
    # launches in a committed rocprofv3 --kernel-trace of THIS command on this build (tools/prof_labels.py joins the trace to
    # the labels by kernel name and grid size; tools/prof_bench.sh produces both).  When it matches this run's configuration
    # the dominant label is picked by IN-GRAPH time and both durations are reported.
    in_graph, in_graph_src = {}, None
    lab_file = ROOT / "profiles" / "round4_label_durations.json"
    if lab_file.exists():
        rec = json.loads(lab_file.read_text())
        if rec.get("config") == [B, S, c, args.dtype, args.generator]:
            in_graph = {k: v["in_graph_avg_us"] for k, v in rec.get("labels", {}).items() if v.get("in_graph_avg_us")}
            in_graph_src = f"profiles/{lab_file.name} (rocprofv3 --kernel-trace of this command, committed; {rec.get('source', '')})"
    tot_ig = {k: (prof[k][0] / max(args.probe_steps, 1)) * in_graph.get(k, prof[k][1] * 1e3) for k in prof}   # us per iteration
    dom = max(tot_ig, key=tot_ig.get) if in_graph else max(tot, key=tot.get)
    n_dom, ms_dom, fl_dom, by_dom, st_dom, kern_dom, ms_dom_single = prof[dom]
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    ach = fl_dom / (ms_dom * 1e-3) / 1e12
    ach_gbs = by_dom / (ms_dom * 1e-3) / 1e9
    ig_us = in_graph.get(dom)
    labels = {k: dict(launches_per_iter=round(v[0] / max(args.probe_steps, 1), 2), probe_us=round(v[1] * 1e3, 2), gflop=round(v[2] / 1e9, 3),
                      algorithmic_mb=round(v[3] / 1e6, 2), kernel=v[5], workgroups=run.grids.get(k, 0),
                      in_graph_us=in_graph.get(k)) for k, v in prof.items()}
    # the roof is a property of the layer (MFMA for every layer with >= 64 input channels, HBM for the 8-channel
    # first layers); algorithmic bytes = input + output + weights once, in the compute dtype (engine._algorithmic_bytes)
    hbm_bound = engine.roofline_bound(dom) == "hbm"
    # HBM bytes per launch of the dominant kernel: not measurable inside this run (PMC counters need rocprofv3); taken from
    # the committed rocprofv3 --pmc passes of this build (FETCH_SIZE x2-corrected + WRITE_SIZE, tools/pmc_round3.sh) when
    # this run is the configuration they were taken on, else null
    traffic = traffic_src = None
    for name in ("round4_pmc_dominant.json", "round3_pmc_dominant.json", "round2_pmc_dominant.json"):
        pmc = ROOT / "profiles" / name
        if pmc.exists():
            rec = json.loads(pmc.read_text())
            lab = rec.get("labels", {}).get(dom)
            if lab and lab.get("hbm_bytes_per_launch") and rec.get("config") == [B, S, c, args.dtype, args.generator]:
                traffic, traffic_src = lab["hbm_bytes_per_launch"], f"profiles/{pmc.name} (committed rocprofv3 --pmc passes of this label's launch shape, not this run)"
                break
    roofline = dict(bound="hbm" if hbm_bound else "mfma", kernel=dom,
                    kernel_symbol=kern_dom.replace("<O,", f"<{dtype_symbol(args.dtype)},").replace("<T,", f"<{dtype_symbol(args.dtype)},"),
                    kernel_symbol_note="the kernel template expression the dispatcher launched for this label (gcssl_last_kernel); "
                                       "rocprofv3's --kernel-trace lists it under this name with the same template arguments",
                    achieved=round(ach_gbs if hbm_bound else ach, 2), peak=HBM_PEAK_GBS if hbm_bound else peak,
                    unit="GB/s" if hbm_bound else "TFLOP/s", frac=round(ach_gbs / HBM_PEAK_GBS if hbm_bound else ach / peak, 4),
                    traffic=traffic, traffic_source=traffic_src, launches=n_dom, avg_us=round(ms_dom * 1e3, 2),
                    avg_us_single=round(ms_dom_single * 1e3, 2),
                    # the same label inside the replayed graphs, from the committed kernel trace (null when none matches this run)
                    in_graph_avg_us=ig_us, in_graph_source=in_graph_src,
                    frac_in_graph=(round((by_dom / (ig_us * 1e-6) / 1e9) / HBM_PEAK_GBS if hbm_bound else (fl_dom / (ig_us * 1e-6) / 1e12) / peak, 4)
                                   if ig_us else None),
                    dominant_by="in-graph time per iteration (committed trace)" if in_graph else "eager probe time per iteration",
                    timing=f"HIP events in an eager pass behind the timed region, on the launch stream; avg_us = per launch inside "
                           f"{PROBE_REPEATS} back-to-back launches between two events (a link of a launch chain, as in the graph "
                           f"replay: what rocprofv3 --kernel-trace reports for the replay); avg_us_single = one launch alone "
                           f"between two events (adds the eager launch gap and the event packets)",
                    algorithmic=dict(flops=fl_dom, bytes=by_dom, stored_bytes=st_dom, tflops=round(ach, 2), gbs=round(ach_gbs, 1),
                                     mfma_frac=round(ach / peak, 4), hbm_frac=round(ach_gbs / HBM_PEAK_GBS, 4)),
                    all_convs=all_convs,
                    # MFMA utilisation of the critic's conv stack (every D.* conv launch of an iteration)
                    d_convs=d_convs)
    flop_iter = ((12 * c + 1) * F_D[S] + (c + 3) * F_G[S]) * B * world if (S in F_D and args.generator == "unet") else None

    out = dict(metric="images/sec (G+D step)", value=round(value, 1), unit="images/s", n_gpus=world, rccl_ranks=rccl_ranks,
               steps=args.steps, warmup=args.warmup, preroll_steps=preroll, warmup_total_steps=args.warmup + 1 + preroll,
               warmup_note="`warmup` eager iterations + 1 capture-check replay + `preroll_steps` untimed replays (--preroll-s of clock "
                           "ramp) run before the timed `steps`; `sustained` is a second, longer window behind them",
               ms_per_step=round(ms, 4), higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype=args.dtype, data="synthetic",
               sustained_ms_per_step=sustained["ms_per_step"] if sustained else None, sustained=sustained,
               finite_after_run=finite, last_d_loss=round(d_loss_last, 6), saturations=sat,
               config=dict(workload=f"cGAN WGAN-GP iteration (n_critic={c} critic steps + 1 generator step), "
                                    f"{S}x{S}x3, batch {B}/GPU, reference G ({'U-Net' if args.generator == 'unet' else 'GeneratorSimpleRegressor'}) "
                                    f"+ D (SN PatchGAN)",
                           baseline_config="BASELINE.json configs[1] (32x32x3, batch 256, bf16, 1 GPU)"
                           if (B, S, c, args.dtype, args.generator) == (256, 32, 2, "bf16", "unet") else None,
                           global_batch=B * world, img_size=S, n_critic=c, parallelism=f"dp{world}",
                           launch=("hipGraph replay" + (" (software-pipelined across iterations: the replayed graph ends with the next "
                                                        "iteration's batched generator forward; DESIGN.md 2)"
                                                        if getattr(run.graphed, "pipelined", False) else "")
                                   + ("; the generator step's value-only critic forward of replay i runs as a fourth group of replay "
                                      "i+1's first critic forward: every timed replay runs exactly one (the previous replay's), the last "
                                      "one's is run after the timed region (GraphedIteration.finish)"
                                      if getattr(run.graphed, "batch_g", False) else ""))
                           if run.graphed is not None else "eager",
                           algorithmic_tflops=round(flop_iter / (ms * 1e-3) / 1e12, 2) if flop_iter else None),
               roofline=roofline, labels=labels)
    del run
    torch.cuda.empty_cache()

    # ---- the other BASELINE / north_star configurations, each in its own bounded window (single-GPU runs)
    if rank == 0 and world == 1 and not args.no_also and args.generator == "unet":
        also = []
        alt = "fp16" if args.dtype != "fp16" else "bf16"
        for (b2, s2, dt2, label) in [(256, 32, alt, f"{alt} operands at the headline shape"),
                                     (512, 32, "fp16", "BASELINE configs[3] per-GPU shape (SVHN: batch 512, fp16, fp32 loss accumulation)"),
                                     (128, 64, "fp16", "BASELINE configs[4] / north_star 64x64 (STL shape: batch 128)"),
                                     (256, 32, "fp16x3", "split-precision parity-grade throughput mode at the headline shape (fp32 tensors, "
                                                         "conv operands split hi + lo into fp16 halves, 3 MFMAs per K step)"),
                                     (256, 32, "fp32", "exact fp32-MFMA parity mode at the headline shape")]:
            if (b2, s2, dt2) == (B, S, args.dtype):
                continue
            try:
                r2 = Runner(engine, synth, dist_mod, dev, rank, world, b2, s2, c, dt2, "unet", 2, args.no_graph)
                e0 = r2.timed(3)
                r2.timed(max(3, int(0.3 / (e0 / 3))))                    # untimed pre-roll, as for the headline window
                e0 = r2.timed(3)
                k2 = max(3, min(args.steps * 4, int(args.also_s / (e0 / 3))))
                e2 = r2.timed(k2)
                ok2, _ = r2.finite()
                _, _, all2, d2 = r2.probe(2)
                also.append(dict(config=label, dtype=dt2, batch=b2, img_size=s2, n_critic=c, steps=k2, seconds=round(e2, 3),
                                 ms_per_step=round(e2 / k2 * 1e3, 4), images_per_s=round(b2 * k2 / e2, 1), finite_after_run=ok2,
                                 launch="hipGraph replay" if r2.graphed is not None else "eager",
                                 d_convs=d2, all_convs=all2, saturations=r2.eng.saturations()))
                del r2
                torch.cuda.empty_cache()
            except Exception as e:                                           # a failing side configuration must not hide the headline
                also.append(dict(config=label, dtype=dt2, batch=b2, img_size=s2, error=f"{type(e).__name__}: {e}"))
        # ---- the headline configuration with the re-crop stage in the loop
        try:
            ni, w, h = 32, 1280, 720
            r3 = Runner(engine, synth, dist_mod, dev, rank, world, B, S, c, args.dtype, "unet", 2, args.no_graph, recrop=(ni, w, h))
            e0 = r3.timed(3)
            r3.timed(max(3, int(0.3 / (e0 / 3))))
            e0 = r3.timed(3)
            k3 = max(3, min(args.steps * 4, int(args.also_s / (e0 / 3))))
            e3 = r3.timed(k3)
            ok3, _ = r3.finite()
            # the stage alone: HIP events around n_critic + 1 eager calls, on the launch stream
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dl = r3.eng.gfa.delta
            reps = 20
            ev0.record()
            for _ in range(reps):
                for k in range(c + 1):
                    r3.refine_stage(dl[k * B:(k + 1) * B], k)
            ev1.record(); torch.cuda.synchronize()
            us_call = ev0.elapsed_time(ev1) / (reps * (c + 1)) * 1e3
            also.append(dict(config=f"end-to-end with the GPU re-crop stage in the loop ({ni} synthetic {w}x{h} source images resident in HBM; "
                                    f"{c + 1} re-crop calls per iteration, captured in the graphs; cgan/cgan_train_enhanced.py:37-137,313-315,358-360)",
                             dtype=args.dtype, batch=B, img_size=S, n_critic=c, steps=k3, seconds=round(e3, 3),
                             ms_per_step=round(e3 / k3 * 1e3, 4), images_per_s=round(B * k3 / e3, 1), finite_after_run=ok3,
                             launch="hipGraph replay" if r3.graphed is not None else "eager",
                             recrop_us_per_call=round(us_call, 1), recrop_patches_per_s=round(B / us_call * 1e6, 0),
                             recrop_calls_per_iter=c + 1))
            del r3
            torch.cuda.empty_cache()
        except Exception as e:
            also.append(dict(config="end-to-end with the GPU re-crop stage in the loop", error=f"{type(e).__name__}: {e}"))
        out["also"] = also
        by_dt = {a["dtype"]: a["images_per_s"] for a in also
                 if "images_per_s" in a and (a["batch"], a["img_size"]) == (B, S) and "recrop_us_per_call" not in a}
        by_dt[args.dtype] = round(value, 1)
        # the fastest mode at the headline shape whose measured error (mode_error below) meets north_star's 1e-3: the
        # split-precision mode; the exact-fp32 MFMA mode rides along
        out["parity_mode_images_per_s"] = by_dt.get("fp16x3", by_dt.get("fp32"))
        out["parity_mode"] = "fp16x3" if "fp16x3" in by_dt else "fp32"
        out["exact_fp32_mode_images_per_s"] = by_dt.get("fp32")

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"], reference = cpu_baseline(synth, 42, B, S, c, gtype=args.generator)
        # measured error of the 16-bit modes (and the parity mode) on this configuration against that oracle iteration
        try:
            modes = [args.dtype] + [m for m in ("bf16", "fp16", "fp16x3", "bf16x3", "fp32") if m != args.dtype]
            out["mode_error"] = dict(source="measured in this run: one eager engine iteration per mode against the first oracle "
                                            "iteration of the cpu_baseline leg (same initial weights, inputs, alphas, dropout masks); "
                                            "scores / delta: max-norm relative; scalars: relative; d_grad_norm signed",
                                     **{m: measure_mode_error(engine, synth, reference, m, B, S, c, dev, args.generator) for m in modes})
        except Exception as e:
            out["mode_error"] = dict(error=f"{type(e).__name__}: {e}")
    if rank == 0:
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
