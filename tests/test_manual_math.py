"""The hand-derived schedule (oracle/manual_step.py: the spec the HIP engine follows) against the
autograd oracle (oracle/cgan_oracle.py), CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import cgan_oracle as O
from oracle import manual_step as M

T = torch.from_numpy


def _state(synth, seed):
    g = {k: T(v) for k, v in synth.generator_state(seed).items()}
    d = {k: T(v) for k, v in synth.discriminator_state(seed).items()}
    return g, d


def test_in_double_backward_formula():
    torch.manual_seed(0)
    z = torch.randn(3, 5, 4, 4, dtype=torch.float64, requires_grad=True)
    dn = torch.randn(3, 5, 4, 4, dtype=torch.float64, requires_grad=True)
    q = torch.randn(3, 5, 4, 4, dtype=torch.float64)
    n = torch.nn.functional.instance_norm(z, eps=M.EPS)
    (dz,) = torch.autograd.grad(n, z, dn, create_graph=True)
    g_dn, g_z = torch.autograd.grad((dz * q).sum(), [dn, z])
    mu, r = M.in_stats(z.detach())
    xh = (z.detach() - mu) * r
    assert rel_err(M.in_bwd(xh, r, dn.detach()), dz.detach()) < 1e-12
    a_dn, a_z = M.in_bwd_bwd(xh, r, dn.detach(), q)
    assert rel_err(a_dn, g_dn) < 1e-11
    assert rel_err(a_z, g_z) < 1e-11


@pytest.mark.parametrize("B,S,seed", [(3, 32, 5), (2, 64, 6)])
def test_d_step_grads_match_autograd(synth, B, S, seed):
    _, d = _state(synth, seed)
    inp = synth.step_inputs(seed, B, S, 1, tag="manual")
    pred, gt, refined, alpha = T(inp["pred"]), T(inp["gt"]), T(inp["refined"][0]), T(inp["alpha"][0])
    # autograd oracle
    da = {k: v.clone() for k, v in d.items()}
    for k in O.D_PARAM_KEYS:
        da[k].requires_grad_(True)
    real = O.d_forward(da, pred, gt, True)
    fake = O.d_forward(da, pred, refined, True)
    taps = {}
    gp = O.gradient_penalty(da, (pred, gt), (pred, refined), alpha, taps=taps)
    wd = real.mean() - fake.mean()
    loss = -wd + 1.0 * gp
    ref = dict(zip(O.D_PARAM_KEYS, torch.autograd.grad(loss, [da[k] for k in O.D_PARAM_KEYS])))
    # manual
    dm = {k: v.clone() for k, v in d.items()}
    grads, log = M.d_step_grads(dm, pred, gt, refined, alpha, 1.0)
    assert rel_err(log["real"], real.detach()) < 1e-5
    assert rel_err(log["fake"], fake.detach()) < 1e-5
    assert rel_err(log["d_interp"], taps["d_interp"]) < 1e-5
    assert abs(log["gp"] - float(gp)) < 1e-5 * max(1, abs(float(gp)))
    assert abs(log["wd"] - float(wd)) < 1e-5
    gpg = torch.cat([taps["gp_grad_pred"], taps["gp_grad_other"]], 1)
    assert rel_err(log["gp_grad"], gpg) < 1e-5
    for i in O.D_IDX:
        assert rel_err(dm[f"model.{i}.weight_u"], da[f"model.{i}.weight_u"]) < 1e-6
        assert rel_err(dm[f"model.{i}.weight_v"], da[f"model.{i}.weight_v"]) < 1e-6
    for k in O.D_PARAM_KEYS:
        if k in ("model.2.bias", "model.5.bias", "model.8.bias"):
            assert float(grads[k].abs().max()) < 1e-3 * float(ref["model.0.bias"].abs().max())
            continue
        e = rel_err(grads[k], ref[k])
        assert e < 2e-4, (k, e)


def test_eiou_box_gradient_known_answer():
    fix = load_golden("loss_vectors")
    bbox, delta = T(fix["bbox"]), T(fix["delta"])
    # the fixture's target boxes were made from this delta_true stream (make_golden.loss_vectors)
    from conftest import load_pkg
    dt = T(load_pkg("synth").normal("kv.dt", 7, (16, 4), 0.1))
    loss, g, cal = M.eiou_box_loss_and_grad(bbox, delta, dt)
    assert abs(float(loss) - float(fix["hybrid_total"])) < 1e-6
    assert rel_err(cal, fix["apply_train"]) < 1e-6
    assert rel_err(g, fix["hybrid_grad_delta"]) < 1e-5


@pytest.mark.parametrize("B,S,seed", [(3, 32, 7), (2, 64, 8)])
def test_g_forward_backward_match_autograd(synth, B, S, seed):
    g, _ = _state(synth, seed)
    inp = synth.step_inputs(seed, B, S, 1, tag="manualg")
    pred, box, dt = T(inp["pred"]), T(inp["pred_box"]), T(inp["delta_true"])
    masks = [T(m) for m in inp["masks"][0]]
    ga = {k: v.clone().requires_grad_(True) for k, v in g.items()}
    delta = O.g_forward(ga, pred, 0.3, masks)
    loss = O.eiou_loss(O.apply_delta_to_bbox(box, delta, True), O.apply_delta_to_bbox(box, dt, True))
    ref = dict(zip(O.G_PARAM_KEYS, torch.autograd.grad(loss, [ga[k] for k in O.G_PARAM_KEYS])))
    dm, li, grads = M.g_forward_backward(g, pred, 0.3, masks, box, dt)
    assert rel_err(dm, delta.detach()) < 1e-5
    assert abs(li - float(loss)) < 1e-6
    for k in O.G_PARAM_KEYS:
        e = rel_err(grads[k], ref[k])
        assert e < 2e-4, (k, e)


def test_clip_and_adam_matches_torch_optim():
    torch.manual_seed(1)
    ps = [torch.randn(50, 7), torch.randn(33)]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))
    m = [torch.zeros_like(p) for p in ps]; v = [torch.zeros_like(p) for p in ps]
    for t in (1, 2, 3):
        gs = [torch.randn_like(p) * 3 for p in ps]
        for r, g in zip(ref, gs):
            r.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        opt.step()
        tot = M.clip_and_adam(ps, gs, m, v, t)
        assert abs(tot - float(tn)) < 1e-4 * float(tn)
        for p, r in zip(ps, ref):
            assert float((p - r.detach()).abs().max()) < 1e-7
