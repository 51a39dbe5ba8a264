#!/usr/bin/env python3
"""Measured error of each compute mode of the HIP engine against the pinned CPU oracle at the bench configuration
(B=256, 32x32, n_critic=2: BASELINE configs[1]) on the SAME fixture inputs, alphas and dropout masks -- the quantities
SURVEY.md 0 maps BASELINE's "logits / feature-matching loss / real-fake scores" onto: critic scores, G's delta, WD, GP,
EIoU, and the un-clipped gradient norms.  Run on the GPU box:

    python tools/mode_error.py [--out profiles/round2_mode_error.json] [--speed]

`measure()` is also what tests/test_engine_gpu.py::test_16bit_mode_error_vs_oracle asserts on."""
from __future__ import annotations

import argparse
import importlib
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = "gan-calibrated-semi-supervised-learning_amd"
T = torch.from_numpy


def rel(a, b) -> float:
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def oracle_reference(synth, B=256, S=32, c=2, seed=42, threads=32):
    """One oracle iteration on the fixture inputs -> (inputs, log, taps)."""
    from oracle import cgan_oracle as O
    g = {k: T(v) for k, v in synth.generator_state(seed).items()}
    d = {k: T(v) for k, v in synth.discriminator_state(seed).items()}
    inp = synth.step_inputs(seed, B, S, c, tag="fullsize")
    torch.set_num_threads(min(threads, torch.get_num_threads()))
    orc = O.StepOracle(g, d, n_critic=c)
    refined = [T(r) for r in inp["refined"]]
    taps = {}
    ref = orc.iteration(T(inp["pred"]), T(inp["gt"]), T(inp["delta_true"]), T(inp["pred_box"]), lambda dl, k: refined[k],
                        [T(a) for a in inp["alpha"]], [[T(m) for m in ms] for ms in inp["masks"]], taps=taps)
    return (g, d, inp), ref, taps


def measure(dtype: str, state, ref, taps, B=256, S=32, c=2) -> dict:
    """Errors of one engine iteration in compute mode `dtype` relative to the oracle's (first critic step and generator
    step: pure functions of the fixture; the second critic step is reported too but sits behind an Adam update whose
    lr*sign(g) steps amplify any difference)."""
    engine = importlib.import_module(PKG + ".engine")
    g, d, inp = state
    eng = engine.StepEngine(g, d, batch=B, size=S, n_critic=c, dtype=dtype, device="cuda:0")
    refined = [T(r).cuda() for r in inp["refined"]]
    log = eng.iteration(T(inp["pred"]).cuda(), T(inp["gt"]).cuda(), T(inp["delta_true"]).cuda(), T(inp["pred_box"]).cuda(),
                        lambda dl, k: refined[k], alphas=[T(a).cuda().view(-1).contiguous() for a in inp["alpha"]],
                        masks=[[T(m).cuda() for m in ms] for ms in inp["masks"]])
    torch.cuda.synchronize()
    sgn = lambda a, b: float((a - b) / abs(b))
    out = dict(
        scores_real=rel(log["real"][0].cpu().reshape(-1), taps["real_validity"].reshape(-1)),
        scores_fake=rel(log["fake"][0].cpu().reshape(-1), taps["fake_validity"].reshape(-1)),
        delta=rel(log["delta_pred"].cpu(), ref["delta_pred"]),
        wd=abs(sgn(log["wd"][0], ref["wd"][0])), gp=abs(sgn(log["gp"][0], ref["gp"][0])),
        d_loss=abs(sgn(log["d_loss"][0], ref["d_loss"][0])),
        d_grad_norm=sgn(log["d_grad_norm"][0], ref["d_grad_norm"][0]),
        loss_iou=abs(sgn(log["loss_iou"], ref["loss_iou"])),
        g_grad_norm=sgn(log["g_grad_norm"], ref["g_grad_norm"]),
        step2_gp=abs(sgn(log["gp"][1], ref["gp"][1])), step2_d_grad_norm=sgn(log["d_grad_norm"][1], ref["d_grad_norm"][1]),
        finite=bool(torch.isfinite(eng.D.p).all() and torch.isfinite(eng.G.p).all()))
    out["scores"] = max(out["scores_real"], out["scores_fake"])
    return out


def speed(dtype: str, state, B=256, S=32, c=2, iters=40) -> float:
    """images/s of graph-replayed iterations in mode `dtype` (device-drawn alpha/masks, like bench.py)"""
    engine = importlib.import_module(PKG + ".engine")
    g, d, inp = state
    eng = engine.StepEngine(g, d, batch=B, size=S, n_critic=c, dtype=dtype, device="cuda:0", keep_clipped_grads=False)
    refined = [T(r).cuda() for r in inp["refined"]]
    call = (T(inp["pred"]).cuda(), T(inp["gt"]).cuda(), T(inp["delta_true"]).cuda(), T(inp["pred_box"]).cuda(),
            lambda dl, k: refined[k])
    for _ in range(3):
        eng.run_iteration(*call)
    gi = engine.GraphedIteration(eng, *call)
    gi.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        gi.replay()
    torch.cuda.synchronize()
    return B * iters / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "profiles" / "round2_mode_error.json"))
    ap.add_argument("--speed", action="store_true", help="also time every mode (graph replay)")
    ap.add_argument("--modes", default="fp32,bf16,fp16")
    args = ap.parse_args()
    synth = importlib.import_module(PKG + ".synth")
    state, ref, taps = oracle_reference(synth)
    rec = dict(config=dict(batch=256, size=32, n_critic=2, seed=42, inputs="synth.step_inputs(tag='fullsize')",
                           reference="oracle/cgan_oracle.StepOracle (pinned to the reference's golden vectors), fp32 CPU"),
               tolerance="north_star: 1e-3 relative on scores / delta; this table is the measured error per mode",
               modes={}, images_per_s={})
    for m in args.modes.split(","):
        rec["modes"][m] = {k: (round(v, 7) if isinstance(v, float) else v) for k, v in measure(m, state, ref, taps).items()}
        print(m, json.dumps(rec["modes"][m]))
        if args.speed:
            rec["images_per_s"][m] = round(speed(m, state), 1)
            print(m, "images/s", rec["images_per_s"][m])
    rec["parity_mode_images_per_s"] = rec["images_per_s"].get("fp32")
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(rec, indent=1))
    print("wrote", args.out)


if __name__ == "__main__":
    main()
