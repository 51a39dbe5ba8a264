"""CPU-only checks of the harness surface: argument names/defaults mirror the reference's argparse
block and config; the losses module mirrors cgan/losses.py (known-answer vectors); models expose the reference's
state-dict keys without needing a GPU."""
import importlib
import sys

import numpy as np
import torch

from conftest import ROOT, load_golden, load_pkg, rel_err


def test_train_argument_surface_matches_reference():
    sys.path.insert(0, str(ROOT))
    train = importlib.import_module("train")
    cfg = train.load_config()
    args = train.build_parser(cfg).parse_args([])
    expect = dict(img_size=128, batch_size=128, n_epochs=500, lr=2e-4, beta1=0.5, beta2=0.999, lambda_iou=1.0,
                  spectral_norm=True, delta_scale=0.3, generator_type="unet", patience=20, min_delta=1e-5,
                  train_split=0.8, val_split=0.2, save_dir="runs/exp", seed=42, lambda_gp=1.0, n_critic=2,
                  use_eiou=True, pure_eiou=True)
    for k, v in expect.items():
        assert getattr(args, k) == v, k
    assert train.build_parser(cfg).parse_args(["--n_critic", "5", "--lambda_gp", "10"]).n_critic == 5


def test_losses_surface_known_answers():
    L = load_pkg("losses")
    fix = load_golden("loss_vectors")
    T = torch.from_numpy
    bbox, delta, tgt = T(fix["bbox"]), T(fix["delta"]), T(fix["target"])
    assert rel_err(L.apply_delta_to_bbox(bbox, delta, training=True), fix["apply_train"]) < 1e-6
    assert rel_err(L.apply_delta_to_bbox(bbox, delta, training=False), fix["apply_eval"]) < 1e-6
    assert rel_err(L.smooth_clamp(T(fix["x"]), -1.5, 1.5), fix["smooth_clamp"]) < 1e-6
    assert rel_err(L.iou_metric(bbox, tgt), fix["iou"]) < 1e-6
    assert abs(float(L.EIoULoss()(bbox, tgt)) - float(fix["eiou"])) < 1e-6
    d = delta.clone().requires_grad_(True)
    tot, li = L.HybridLoss(lambda_iou=1.0)(d, None, L.apply_delta_to_bbox(bbox, d, True), T(fix["hybrid_gtb"]))
    tot.backward()
    assert abs(float(tot) - float(fix["hybrid_total"])) < 1e-6 and rel_err(d.grad, fix["hybrid_grad_delta"]) < 1e-5


def test_models_expose_reference_state_dict_keys(synth):
    M = load_pkg("models")
    G, D = M.GeneratorUNet(), M.Discriminator()
    assert set(G.state_dict()) == set(synth.generator_state(1))
    assert set(D.state_dict()) == set(synth.discriminator_state(1))
    for k, v in synth.discriminator_state(1).items():
        assert tuple(D.state_dict()[k].shape) == v.shape, k
    for k, v in synth.generator_state(1).items():
        assert tuple(G.state_dict()[k].shape) == v.shape, k
    GS = M.GeneratorSimpleRegressor()                                # generator_type "simple" (cgan/models.py:147-216)
    assert [k for k, _ in GS.named_parameters()] == list(synth.simple_generator_state(1))
    for k, v in synth.simple_generator_state(1).items():
        assert tuple(GS.state_dict()[k].shape) == v.shape, k
    D2 = M.Discriminator(spectral_norm=False)
    assert "model.0.weight" in D2.state_dict() and "model.0.weight_orig" not in D2.state_dict()
