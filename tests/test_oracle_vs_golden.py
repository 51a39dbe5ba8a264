"""Pins the CPU oracle (oracle/cgan_oracle.py) to golden vectors produced by the REFERENCE's own
models.py/losses.py (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import check_pinned, load_golden, rel_err
from oracle import cgan_oracle as O

T = torch.from_numpy
TOL = 2e-5   # oracle vs reference, fp32 CPU both sides


def _gtype(name):
    return "simple" if "simple" in name else "unet"


def _gstate(synth, seed, gtype="unet"):
    return synth.simple_generator_state(seed) if gtype == "simple" else synth.generator_state(seed)


def _sn(name):
    return "nosn" not in name                             # Discriminator(spectral_norm=False): config.yaml `spectral_norm: false`


def _state(synth, seed, gtype="unet", spectral_norm=True):
    g = {k: T(v) for k, v in _gstate(synth, seed, gtype).items()}
    d = {k: T(v) for k, v in synth.discriminator_state(seed, spectral_norm).items()}
    return g, d


def _wsum(sd):
    return np.array([float(np.asarray(v, dtype=np.float64).sum()) for _, v in sorted(sd.items())]
                    + [float(np.abs(np.asarray(v, dtype=np.float64)).sum()) for _, v in sorted(sd.items())])


@pytest.mark.parametrize("name", ["fwd_B2_S32", "fwd_B2_S64", "step_B4_S32", "fwd_simple_B2_S32", "step_simple_B4_S32",
                                  "step_nosn_B4_S32"])
def test_deterministic_weights_regenerate_bit_exact(synth, name):
    fix = load_golden(name)
    seed = int(fix["meta"][0])
    assert np.array_equal(_wsum(_gstate(synth, seed, _gtype(name))), fix["wsum_g"])
    assert np.array_equal(_wsum(synth.discriminator_state(seed, _sn(name))), fix["wsum_d"])


def test_loss_known_answers():
    fix = load_golden("loss_vectors")
    bbox, delta, tgt = T(fix["bbox"]), T(fix["delta"]), T(fix["target"])
    assert rel_err(O.apply_delta_to_bbox(bbox, delta, True), fix["apply_train"]) < 1e-6
    assert rel_err(O.apply_delta_to_bbox(bbox, delta, False), fix["apply_eval"]) < 1e-6
    x = T(fix["x"])
    assert rel_err(O.smooth_clamp(x, -1.5, 1.5), fix["smooth_clamp"]) < 1e-6
    assert rel_err(O.smooth_clamp(x, 0.02, 0.8, 1.0), fix["smooth_clamp_t1"]) < 1e-6
    assert rel_err(O.iou_metric(bbox, tgt), fix["iou"]) < 1e-6
    assert abs(float(O.eiou_loss(bbox, tgt)) - float(fix["eiou"])) < 1e-6
    d = delta.clone().requires_grad_(True)
    tot = O.eiou_loss(O.apply_delta_to_bbox(bbox, d, True), T(fix["hybrid_gtb"]))
    tot.backward()
    assert abs(float(tot) - float(fix["hybrid_total"])) < 1e-6
    assert rel_err(d.grad, fix["hybrid_grad_delta"]) < 1e-5


@pytest.mark.parametrize("name", ["fwd_B2_S32", "fwd_B2_S64"])
def test_forward_activations(synth, name):
    fix = load_golden(name)
    seed, B, S = (int(v) for v in fix["meta"])
    inp = synth.step_inputs(seed, B, S, 1, tag=name)
    pred, gt = T(inp["pred"]), T(inp["gt"])
    g, d = _state(synth, seed)
    for mode in ("eval", "train"):      # same order as the generator (u,v advance in train)
        taps = {}
        with torch.no_grad():
            score = O.d_forward(d, pred, gt, train=(mode == "train"), taps=taps)
            masks = [T(m) for m in inp["masks"][0]] if mode == "train" else None
            delta = O.g_forward(g, pred, 0.3, masks, taps=taps)
        assert rel_err(score, fix[f"{mode}.d_out"]) < TOL
        assert rel_err(delta, fix[f"{mode}.g_delta"]) < TOL
        for j in range(1, 5):
            check_pinned(fix, f"{mode}.d.act{j}", taps[f"d.a{j}"], TOL, synth)
        for n, k in (("down1", "g.d1"), ("down2", "g.d2"), ("down3", "g.d3"), ("down4", "g.d4"),
                     ("up1", "g.u1"), ("up2", "g.u2"), ("up3", "g.u3"), ("up4", "g.u4")):
            check_pinned(fix, f"{mode}.g.{n}", taps[k], TOL, synth)
        for i in O.D_IDX:
            assert rel_err(d[f"model.{i}.weight_u"], fix[f"{mode}.u.{i}"]) < TOL
            assert rel_err(d[f"model.{i}.weight_v"], fix[f"{mode}.v.{i}"]) < TOL


def test_simple_generator_forward_activations(synth):
    """GeneratorSimpleRegressor (generator_type "simple", cgan/models.py:147-216): every ReLU output, the pooled features
    and delta, in eval and train mode, against the reference module's own forward."""
    name = "fwd_simple_B2_S32"
    fix = load_golden(name)
    seed, B, S = (int(v) for v in fix["meta"])
    inp = synth.step_inputs(seed, B, S, 1, tag=name, generator_type="simple")
    g, _ = _state(synth, seed, "simple")
    for mode in ("eval", "train"):
        taps = {}
        masks = [T(m) for m in inp["masks"][0]] if mode == "train" else None
        with torch.no_grad():
            delta = O.g_simple_forward(g, T(inp["pred"]), 0.3, masks, taps=taps)
        assert rel_err(delta, fix[f"{mode}.g_delta"]) < TOL
        for j in range(8):
            check_pinned(fix, f"{mode}.gs.a{j}", taps[f"gs.a{j}"], TOL, synth)
        check_pinned(fix, f"{mode}.gs.feat", taps["gs.feat"], TOL, synth)


def run_oracle_case(synth, name):
    fix = load_golden(name)
    seed, B, S, n_critic, iters, gray = (int(v) for v in fix["meta"])
    gtype = _gtype(name)
    g, d = _state(synth, seed, gtype, _sn(name))
    orc = O.StepOracle(g, d, n_critic=n_critic, generator_type=gtype, spectral_norm=_sn(name))
    logs, taps0 = [], {}
    for it in range(iters):
        inp = synth.step_inputs(seed + 1000 * it, B, S, n_critic, tag=name, generator_type=gtype)
        if gray:
            for key in ("pred", "gt"):
                z = np.zeros_like(inp[key]); z[:, :, 2:30, 2:30] = inp[key][:, :1, 2:30, 2:30]
                inp[key] = z
        refined = [T(r) for r in inp["refined"]]
        logs.append(orc.iteration(T(inp["pred"]), T(inp["gt"]), T(inp["delta_true"]), T(inp["pred_box"]),
                                  lambda delta, k: refined[k], [T(a) for a in inp["alpha"]],
                                  [[T(m) for m in ms] for ms in inp["masks"]],
                                  taps=taps0 if it == 0 else None))
    return fix, orc, logs, taps0


@pytest.mark.parametrize("name", ["step_B4_S32", "step_B2_S64", "step_B2_S128", "step_mnist_B4_S32",
                                  "step_simple_B4_S32", "step_simple_B2_S64", "step_nosn_B4_S32"])
def test_training_step(synth, name):
    fix, orc, logs, taps = run_oracle_case(synth, name)
    n_critic, iters = int(fix["meta"][3]), int(fix["meta"][4])
    full = f"it0.c0.d_interp" in fix
    for it in range(iters):
        lg = logs[it]
        # iteration 0 is a pure function of the fixture inputs; later iterations inherit Adam's
        # ~lr*sign(g) first steps (a near-zero gradient may flip sign), so they get the north-star
        # tolerance (1e-3) instead of the fp32-rounding one.
        tol = 5e-5 if it == 0 else 1e-3
        for c in range(n_critic):
            sc = fix[f"it{it}.c{c}.scalars"]
            got = np.array([lg["d_loss"][c], lg["gp"][c], lg["wd"][c], lg["d_grad_norm"][c]])
            # (without the spectral norm the critic's weights are 1/sigma ~ 7x larger in effect: the second critic step, behind
            #  one Adam update, measured 5.7e-5 on the gradient norm)
            assert rel_err(got, sc) < (tol if (c == 0 or _sn(name)) else max(tol, 2e-4)), (it, c, got, sc)
        gs = fix[f"it{it}.gscalars"]
        got = np.array([lg["loss_g"], lg["loss_iou"], lg["loss_wgan"], lg["g_grad_norm"]])
        assert rel_err(got, gs) < tol, (it, got, gs)
        assert rel_err(lg["delta_pred"], fix[f"it{it}.delta_pred"]) < tol
    if full:
        assert rel_err(taps["d_interp"], fix["it0.c0.d_interp"]) < TOL
        check_pinned(fix, "it0.c0.gp_grad_pred", taps["gp_grad_pred"], 5e-5, synth)
        check_pinned(fix, "it0.c0.gp_grad_other", taps["gp_grad_other"], 5e-5, synth)
        assert rel_err(taps["real_validity"], fix["it0.c0.real_validity"]) < TOL
        assert rel_err(taps["fake_validity"], fix["it0.c0.fake_validity"]) < TOL
        for k in orc.d_keys:
            if k in ("model.2.bias", "model.5.bias", "model.8.bias"):
                continue   # exactly-zero gradient (bias cancelled by InstanceNorm): rounding noise only
            check_pinned(fix, f"it0.c0.dgrad.{k}", taps[f"d.grad.{k}"], 2e-4, synth)
        for k in orc.g_keys:
            if k.startswith("features.") and k.endswith(".bias"):
                continue   # conv bias in front of InstanceNorm: exactly-zero gradient, rounding noise only
            check_pinned(fix, f"it0.ggrad.{k}", taps[f"g.grad.{k}"], 2e-4, synth)
    # state after the last iteration.  Adam's first steps are ~lr*sign(g): compare with an
    # absolute tolerance of a few % of lr per step taken.
    last = iters - 1
    lr = 2e-4
    for k, v in orc.d.items():
        if k.endswith(".bias") and k != "model.0.bias":
            continue
        a = v.detach().numpy().reshape(-1)
        key = f"it{last}.D.{k}"
        if key in fix:
            assert np.abs(a - fix[key].reshape(-1)).max() < 0.05 * lr * (last + 1) * n_critic + 1e-6, k
        else:
            smp = a[synth.sample_indices(a.size, 256)]
            bad = np.abs(smp - fix[key + "@sample"]) > 0.05 * lr * (last + 1) * n_critic + 1e-6
            assert bad.mean() <= 0.01, (k, bad.mean())
    for k, v in orc.g.items():
        if k.startswith("features.") and k.endswith(".bias"):
            continue       # Adam turns the rounding-noise gradient of these biases into +-lr steps of either sign
        a = v.detach().numpy().reshape(-1)
        key = f"it{last}.G.{k}"
        if key in fix:
            assert np.abs(a - fix[key].reshape(-1)).max() < 0.05 * lr * (last + 1) + 1e-6, k
        else:
            smp = a[synth.sample_indices(a.size, 256)]
            bad = np.abs(smp - fix[key + "@sample"]) > 0.05 * lr * (last + 1) + 1e-6
            assert bad.mean() <= 0.01, (k, bad.mean())
