#!/usr/bin/env python3
"""Per-layer attribution of the 16-bit modes' error, by EMULATION on the CPU (no GPU needed): the first critic step of the
bench configuration (B=256, 32x32) in fp32 torch ops with the roundings the HIP engine applies in its 16-bit modes switched
on one class / one layer at a time:

  A_l  the input activation of critic layer l stored in 16 bits (l = 1: the packed input pair; l = 5: the head's input)
  W_l  the packed conv weights of layer l in 16 bits (the 1/sigma of the spectral norm stays an fp32 epilogue factor)
  G_l  the gradient w.r.t. layer l's conv output stored in 16 bits (gb_zs of the gradient-penalty chain and dzs of the
       backward: both are MFMA operands of the data- and weight-gradient convs)
  T_l  the second-order adjoint w.r.t. layer l's activation (gt_a of the reverse gradient-penalty chain) in 16 bits

Everything else (accumulators, pre-norm values, statistics, incoming gradients of the norm kernels, weight gradients) is
fp32 in the engine and here.  Reported against the all-fp32 run: critic scores (max-norm relative), Wasserstein term, gradient
penalty, un-clipped gradient norm.  usage: python tools/attribution_cpu.py [fp16|bf16] [B]"""
import importlib, json, sys, time
from pathlib import Path
import numpy as np
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = "gan-calibrated-semi-supervised-learning_amd"
T = torch.from_numpy
D_IDX = (0, 2, 5, 8)


def Q(x, dt):
    return x + (x.to(dt).float() - x).detach()


class RoundGrad(torch.autograd.Function):
    """identity whose backward rounds the gradient (differentiably: the second-order flow passes unrounded)"""
    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return Q(g, ctx.dt), None


class RoundAdjoint(torch.autograd.Function):
    """identity whose backward's backward rounds: the adjoint of the first-order gradient w.r.t. this tensor is rounded"""
    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return RoundGrad.apply(g, ctx.dt), None


def sn_sigma(sd, i):
    w = sd[f"model.{i}.weight_orig"]
    wm = w.reshape(w.shape[0], -1)
    with torch.no_grad():
        v = torch.mv(wm.t(), sd[f"model.{i}.weight_u"]); v = v / v.norm().clamp_min(1e-12)
        u = torch.mv(wm, v); u = u / u.norm().clamp_min(1e-12)
        sd[f"model.{i}.weight_u"], sd[f"model.{i}.weight_v"] = u, v
    return torch.dot(u, torch.mv(wm, v))


def d_forward(sd, pred, other, on, dt):
    """on: set of switches like 'A1', 'W3', 'G2', 'T4'"""
    x = torch.cat([pred, other], 1)
    for li, i in enumerate(D_IDX):
        l = li + 1
        if f"A{l}" in on:
            x = Q(x, dt)
        if f"T{l - 1}" in on and l > 1:
            x = RoundAdjoint.apply(x, dt)
        w = sd[f"model.{i}.weight_orig"]
        sigma = sn_sigma(sd, i)
        wq = Q(w, dt) if f"W{l}" in on else w
        z = F.conv2d(x, wq, None, stride=2, padding=1) / sigma + sd[f"model.{i}.bias"].view(1, -1, 1, 1)
        if f"G{l}" in on:
            z = RoundGrad.apply(z, dt)
        if li > 0:
            z = F.instance_norm(z, eps=1e-5)
        x = F.leaky_relu(z, 0.2)
    if "A5" in on:
        x = Q(x, dt)
    if "T4" in on:
        x = RoundAdjoint.apply(x, dt)
    return F.conv2d(x, sd["model.11.weight"], None, stride=1, padding=1)


def critic_step(sd0, pred, gt, refined, alpha, on, dt):
    sd = {k: v.clone() for k, v in sd0.items()}
    keys = [k for i in D_IDX for k in (f"model.{i}.bias", f"model.{i}.weight_orig")] + ["model.11.weight"]
    for k in keys:
        sd[k].requires_grad_(True)
    real = d_forward(sd, pred, gt, on, dt)
    fake = d_forward(sd, pred, refined, on, dt)
    a = alpha.view(-1, 1, 1, 1)
    ip = (a * pred + (1 - a) * pred).detach().requires_grad_(True)
    io = (a * gt + (1 - a) * refined).detach().requires_grad_(True)
    di = d_forward(sd, ip, io, on, dt)
    g1, g2 = torch.autograd.grad(di, [ip, io], torch.ones_like(di), create_graph=True)
    B = pred.shape[0]
    nrm = torch.sqrt((g1.reshape(B, -1) ** 2).sum(1) + (g2.reshape(B, -1) ** 2).sum(1) + 1e-12)
    gp = ((nrm - 1) ** 2).mean()
    wd = real.mean() - fake.mean()
    grads = torch.autograd.grad(-wd + gp, [sd[k] for k in keys])
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads))
    return dict(real=real.detach(), fake=fake.detach(), wd=float(wd), gp=float(gp), gnorm=float(total),
                grads={k: g.detach() for k, g in zip(keys, grads)})


def main():
    dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "fp16"]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    synth = importlib.import_module(PKG + ".synth")
    sd = {k: T(v) for k, v in synth.discriminator_state(42).items()}
    inp = synth.step_inputs(42, B, 32, 2, tag="fullsize")
    pred, gt, refined, alpha = T(inp["pred"]), T(inp["gt"]), T(inp["refined"][0]), T(inp["alpha"][0]).view(-1)
    t0 = time.time()
    ref = critic_step(sd, pred, gt, refined, alpha, set(), dt)
    print(f"# fp32 reference: wd {ref['wd']:.6g} gp {ref['gp']:.6g} grad-norm {ref['gnorm']:.6g}  ({time.time() - t0:.1f} s)", flush=True)
    L = range(1, 5)
    cfgs = [("all", {f"{c}{l}" for c in "AWGT" for l in L} | {"A5"})]
    cfgs += [(f"{c}* (all layers)", {f"{c}{l}" for l in L} | ({"A5"} if c == "A" else set())) for c in "AWGT"]
    cfgs += [(f"layer {l} (A W G T)", {f"{c}{l}" for c in "AWGT"}) for l in L]
    cfgs += [(f"A{l}", {f"A{l}"}) for l in range(1, 6)] + [(f"W{l}", {f"W{l}"}) for l in L]
    cfgs += [(f"G{l}", {f"G{l}"}) for l in L] + [(f"T{l}", {f"T{l}"}) for l in L]
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    rows = []
    for name, on in cfgs:
        r = critic_step(sd, pred, gt, refined, alpha, on, dt)
        row = dict(cfg=name, scores=max(rel(r["real"], ref["real"]), rel(r["fake"], ref["fake"])),
                   wd=abs(r["wd"] - ref["wd"]) / abs(ref["wd"]), gp=(r["gp"] - ref["gp"]) / ref["gp"],
                   gnorm=(r["gnorm"] - ref["gnorm"]) / ref["gnorm"],
                   # (conv weights only: the biases in front of an InstanceNorm have an exactly-zero true gradient)
                   worst_grad=max(rel(r["grads"][k], ref["grads"][k]) for k in r["grads"] if "weight" in k))
        rows.append(row)
        print(f"{name:22s} scores {row['scores']:9.2e}  wd {row['wd']:9.2e}  gp {row['gp']:+9.2e}  grad-norm {row['gnorm']:+9.2e}  "
              f"worst weight-grad tensor {row['worst_grad']:9.2e}", flush=True)
    out = ROOT / "profiles" / f"round3_attribution_cpu_{sys.argv[1] if len(sys.argv) > 1 else 'fp16'}_B{B}.json"
    out.write_text(json.dumps(dict(dtype=str(dt), batch=B, reference=dict(wd=ref["wd"], gp=ref["gp"], gnorm=ref["gnorm"]), rows=rows), indent=1))
    print("wrote", out)


if __name__ == "__main__":
    main()
