"""The drop-in module surface (models.Discriminator / GeneratorUNet / losses.compute_gradient_penalty) on the HIP
kernels with torch.autograd, first and second order, against the CPU oracle with the same weights and inputs."""
import numpy as np
import pytest
import torch

from conftest import load_golden, load_pkg, rel_err
from oracle import cgan_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def build(synth, seed):
    models = load_pkg("models")
    G, D = models.GeneratorUNet(0.3).cuda(), models.Discriminator(True).cuda()
    gsd = {k: T(v) for k, v in synth.generator_state(seed).items()}
    dsd = {k: T(v) for k, v in synth.discriminator_state(seed).items()}
    assert set(G.state_dict().keys()) == set(gsd) and set(D.state_dict().keys()) == set(dsd)
    G.load_state_dict(gsd); D.load_state_dict(dsd)
    return G, D, gsd, dsd


def test_state_dict_keys_and_init(synth):
    models = load_pkg("models")
    G, D = models.GeneratorUNet(), models.Discriminator()          # None -> config.yaml defaults
    assert G.delta_scale == 0.3 and D.spectral_norm is True
    assert sum(p.numel() for p in G.parameters()) == 6294788 and sum(p.numel() for p in D.parameters()) == 2767808
    G.apply(models.weights_init_normal); D.apply(models.weights_init_normal)
    w = D.model[2].weight_orig
    assert abs(float(w.std()) - 0.02) < 1e-3 and float(D.model[2].bias.abs().max()) == 0.0
    assert abs(float(G.up4[0].weight.std()) - 0.02) < 2e-3


@pytest.mark.parametrize("B,S,seed", [(3, 32, 5), (2, 64, 6)])
def test_critic_step_via_autograd_matches_oracle(synth, B, S, seed):
    losses = load_pkg("losses")
    G, D, gsd, dsd = build(synth, seed)
    inp = synth.step_inputs(seed, B, S, 1, tag="mod")
    pred, gt, refined, alpha = (T(inp[k]) for k in ("pred", "gt")), None, None, None
    pred, gt = T(inp["pred"]), T(inp["gt"])
    refined, alpha = T(inp["refined"][0]), T(inp["alpha"][0])
    # oracle (CPU autograd)
    da = {k: v.clone() for k, v in dsd.items()}
    for k in O.D_PARAM_KEYS:
        da[k].requires_grad_(True)
    real = O.d_forward(da, pred, gt, True); fake = O.d_forward(da, pred, refined, True)
    gp = O.gradient_penalty(da, (pred, gt), (pred, refined), alpha)
    loss = -(real.mean() - fake.mean()) + 1.0 * gp
    ref = dict(zip(O.D_PARAM_KEYS, torch.autograd.grad(loss, [da[k] for k in O.D_PARAM_KEYS])))
    # HIP modules, the reference's own call pattern (cgan/cgan_train_enhanced.py:305-330)
    D.train()
    p, g, r = pred.cuda(), gt.cuda(), refined.cuda()
    real_v = D(p, g)
    fake_v = D(p, r)
    gp_v = losses.compute_gradient_penalty(D, (p, g), (p, r), "cuda", alpha=alpha.cuda())
    d_loss = -(torch.mean(real_v) - torch.mean(fake_v)) + 1.0 * gp_v
    d_loss.backward()
    assert rel_err(real_v.detach().cpu(), real.detach()) < 2e-4
    assert rel_err(fake_v.detach().cpu(), fake.detach()) < 2e-4
    assert abs(float(gp_v) - float(gp)) < 2e-4 * abs(float(gp))
    named = dict(D.named_parameters())
    for k in O.D_PARAM_KEYS:
        if k in ("model.2.bias", "model.5.bias", "model.8.bias"):
            continue
        e = rel_err(named[k].grad.cpu(), ref[k])
        assert e < 1e-3, (k, e)
    for i in O.D_IDX:      # three train-mode forwards advanced u,v three times
        assert rel_err(D.state_dict()[f"model.{i}.weight_u"].cpu(), da[f"model.{i}.weight_u"]) < 1e-4


@pytest.mark.parametrize("B,S,seed", [(3, 32, 7)])
def test_generator_step_via_autograd_matches_oracle(synth, B, S, seed):
    losses = load_pkg("losses")
    G, D, gsd, dsd = build(synth, seed)
    inp = synth.step_inputs(seed, B, S, 1, tag="modg")
    pred, box, dt = T(inp["pred"]), T(inp["pred_box"]), T(inp["delta_true"])
    masks = [T(m) for m in inp["masks"][0]]
    ga = {k: v.clone().requires_grad_(True) for k, v in gsd.items()}
    delta = O.g_forward(ga, pred, 0.3, masks)
    loss = O.eiou_loss(O.apply_delta_to_bbox(box, delta, True), O.apply_delta_to_bbox(box, dt, True))
    ref = dict(zip(O.G_PARAM_KEYS, torch.autograd.grad(loss, [ga[k] for k in O.G_PARAM_KEYS])))
    G.train()
    dl = G(pred.cuda(), masks=[m.cuda() for m in masks])
    crit = losses.HybridLoss(lambda_iou=1.0)
    cal = losses.apply_delta_to_bbox(box.cuda(), dl, training=True)
    gtb = losses.apply_delta_to_bbox(box.cuda(), dt.cuda(), training=True)
    tot, li = crit(dl, dt.cuda(), cal, gtb)
    tot.backward()
    assert rel_err(dl.detach().cpu(), delta.detach()) < 2e-4 and abs(float(li) - float(loss)) < 1e-5
    named = dict(G.named_parameters())
    for k in O.G_PARAM_KEYS:
        e = rel_err(named[k].grad.cpu(), ref[k])
        assert e < 1e-3, (k, e)
    # eval mode: dropout off, deterministic
    G.eval()
    with torch.no_grad():
        e1, e2 = G(pred.cuda()), G(pred.cuda())
    assert rel_err(e1.cpu(), e2.cpu()) < 1e-6      # split-K / pooling use float atomics: order-dependent last bits
    assert rel_err(e1.cpu(), O.g_forward(gsd, pred, 0.3, None)) < 2e-4


def test_simple_generator_module_matches_oracle(synth):
    """models.GeneratorSimpleRegressor (generator_type "simple", cgan/models.py:147-216): state_dict keys, forward and the
    parameter gradients of the EIoU loss through the module's autograd path, against the oracle."""
    models, losses = load_pkg("models"), load_pkg("losses")
    B, S, seed = 3, 32, 9
    gsd = {k: T(v) for k, v in synth.simple_generator_state(seed).items()}
    G = models.GeneratorSimpleRegressor(0.3).cuda()
    assert [k for k, _ in G.named_parameters()] == O.GS_PARAM_KEYS and set(G.state_dict()) == set(gsd)
    G.load_state_dict(gsd)
    inp = synth.step_inputs(seed, B, S, 1, tag="modgs", generator_type="simple")
    pred, box, dt = T(inp["pred"]), T(inp["pred_box"]), T(inp["delta_true"])
    masks = [T(m) for m in inp["masks"][0]]
    ga = {k: v.clone().requires_grad_(True) for k, v in gsd.items()}
    delta = O.g_simple_forward(ga, pred, 0.3, masks)
    loss = O.eiou_loss(O.apply_delta_to_bbox(box, delta, True), O.apply_delta_to_bbox(box, dt, True))
    ref = dict(zip(O.GS_PARAM_KEYS, torch.autograd.grad(loss, [ga[k] for k in O.GS_PARAM_KEYS])))
    G.train()
    dl = G(pred.cuda(), masks=[m.cuda() for m in masks])
    cal = losses.apply_delta_to_bbox(box.cuda(), dl, training=True)
    gtb = losses.apply_delta_to_bbox(box.cuda(), dt.cuda(), training=True)
    tot, li = losses.HybridLoss(lambda_iou=1.0)(dl, dt.cuda(), cal, gtb)
    tot.backward()
    assert rel_err(dl.detach().cpu(), delta.detach()) < 2e-4 and abs(float(li) - float(loss)) < 1e-5
    named = dict(G.named_parameters())
    for k in O.GS_PARAM_KEYS:
        if k.startswith("features.") and k.endswith(".bias"):
            continue                                # exactly-zero gradient (InstanceNorm removes the bias): rounding noise
        e = rel_err(named[k].grad.cpu(), ref[k])
        assert e < 1e-3, (k, e)
    G.eval()
    with torch.no_grad():
        e1 = G(pred.cuda())
    assert rel_err(e1.cpu(), O.g_simple_forward(gsd, pred, 0.3, None)) < 2e-4
    G.compute_dtype = "bf16"                        # throughput mode: bounded, not the parity claim
    with torch.no_grad():
        e2 = G(pred.cuda())
    assert rel_err(e2.cpu(), e1.cpu()) < 6e-2


def test_inputs_are_validated():
    models = load_pkg("models")
    D = models.Discriminator(True).cuda()
    with pytest.raises(ValueError):
        D(torch.zeros(2, 3, 28, 28, device="cuda"), torch.zeros(2, 3, 28, 28, device="cuda"))   # the reference raises too
    with pytest.raises(RuntimeError):
        D(torch.zeros(2, 3, 32, 32), torch.zeros(2, 3, 32, 32))                                   # CPU tensors: no fallback
