"""The reference's ``cgan/models.py`` Python surface on top of the HIP kernels: ``GeneratorUNet``,
``GeneratorSimpleRegressor``, ``Discriminator``, ``weights_init_normal`` with identical constructor arguments, parameter/buffer names
(state_dict keys of SURVEY.md §2.1, incl. ``weight_orig/weight_u/weight_v``) and forward signatures.

Parameters live in ordinary ``nn.Parameter``s in PyTorch layout (so ``state_dict()`` / ``load_state_dict()`` /
optimisers work unchanged); ``forward`` packs them for the kernels and runs the same launch sequences as the step
engine.  ``autograd_nets.py`` holds the differentiable (first- and second-order) wrappers; ``engine.StepEngine`` is the
preallocated, batched, graph-captured fast path for the whole training iteration.
"""
from __future__ import annotations

from pathlib import Path

import torch
import torch.nn as nn
import yaml


def _config_default(key):
    with open(Path(__file__).parent / "config.yaml", "r", encoding="utf-8") as f:
        return yaml.safe_load(f)[key]


def weights_init_normal(m: nn.Module) -> None:
    """Conv* weights ~ N(0, 0.02), biases 0; norm layers have no affine parameters here (cgan/models.py:37-48).
    ``nn.Linear`` is left at its default init, as in the reference."""
    name = m.__class__.__name__
    if name.find("Conv") != -1:
        w = getattr(m, "weight_orig", None)
        if w is None:
            w = m.weight
        nn.init.normal_(w.data, 0.0, 0.02)
        if getattr(m, "bias", None) is not None:
            nn.init.constant_(m.bias.data, 0.0)


class Conv4x4(nn.Module):
    """Parameter holder for Conv2d(k4) / spectrally-normalised Conv2d: weight [Cout][Cin][4][4] (+bias, +u, v)."""

    def __init__(self, cin, cout, bias=False, spectral=False):
        super().__init__()
        w = nn.Parameter(torch.empty(cout, cin, 4, 4))
        nn.init.kaiming_uniform_(w, a=5 ** 0.5)
        if spectral:
            self.weight_orig = w
            self.register_buffer("weight_u", nn.functional.normalize(torch.randn(cout), dim=0, eps=1e-12))
            self.register_buffer("weight_v", nn.functional.normalize(torch.randn(cin * 16), dim=0, eps=1e-12))
        else:
            self.weight = w
        if bias:
            bound = 1.0 / (cin * 16) ** 0.5
            self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))
        else:
            self.bias = None


class ConvT4x4(nn.Module):
    """Parameter holder for ConvTranspose2d(k4,s2,p1,bias=False): weight [Cin][Cout][4][4]."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cin, cout, 4, 4))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self.bias = None


class _Holder(nn.Module):
    """nn.Sequential-like container whose child names are the reference's Sequential indices."""

    def __init__(self, children: dict):
        super().__init__()
        for k, v in children.items():
            self.add_module(str(k), v)

    def __getitem__(self, idx):
        return getattr(self, str(idx))


class GeneratorUNet(nn.Module):
    """4-down / 4-up U-Net -> 4-d box correction (cgan/models.py:89-141)."""

    def __init__(self, delta_scale: float = None):
        super().__init__()
        if delta_scale is None:
            delta_scale = _config_default("delta_scale")
        self.delta_scale = float(delta_scale)
        chans = [(3, 64), (64, 128), (128, 256), (256, 512)]
        for k, (ci, co) in enumerate(chans, 1):
            blk = nn.Module()
            blk.model = _Holder({0: Conv4x4(ci, co)})
            self.add_module(f"down{k}", blk)
        for k, (ci, co) in enumerate([(512, 256), (512, 128), (256, 64)], 1):
            blk = nn.Module()
            blk.model = _Holder({0: ConvT4x4(ci, co)})
            self.add_module(f"up{k}", blk)
        self.up4 = _Holder({0: ConvT4x4(128, 64)})
        self.fc_delta = _Holder({1: nn.Linear(64, 4)})
        self.compute_dtype = "fp32"      # "bf16": bf16 MFMA operands (throughput mode); fp32 is the parity mode

    def forward(self, x: torch.Tensor, masks=None) -> torch.Tensor:
        from .autograd_nets import generator_forward
        return generator_forward(self, x, masks)


class Conv3x3(nn.Module):
    """Parameter holder for Conv2d(cin, cout, 3, padding=1): weight [Cout][Cin][3][3] + bias, nn.Conv2d's default init."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        bound = 1.0 / (cin * 9) ** 0.5
        self.bias = nn.Parameter(torch.empty(cout).uniform_(-bound, bound))


class GeneratorSimpleRegressor(nn.Module):
    """Plain CNN regressor -> 4-d box correction (cgan/models.py:147-216; generator_type "simple"): four blocks of
    2 x [Conv3x3, InstanceNorm, ReLU] + MaxPool2d(2,2), then AdaptiveAvgPool2d(1) and a 512-256-64-4 MLP with
    Dropout(0.5) and Tanh, times delta_scale.  Parameter names are the reference's (``features.N.*``, ``regressor.N.*``)."""

    def __init__(self, delta_scale: float = None):
        super().__init__()
        if delta_scale is None:
            delta_scale = _config_default("delta_scale")
        self.delta_scale = float(delta_scale)
        chans = [(3, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (256, 512), (512, 512)]
        self.features = _Holder({i: Conv3x3(ci, co) for i, (ci, co) in zip((0, 3, 7, 10, 14, 17, 21, 24), chans)})
        self.regressor = _Holder({2: nn.Linear(512, 256), 5: nn.Linear(256, 64), 8: nn.Linear(64, 4)})
        self.compute_dtype = "fp32"

    def forward(self, x: torch.Tensor, masks=None) -> torch.Tensor:
        from .autograd_nets import simple_generator_forward
        return simple_generator_forward(self, x, masks)


class Discriminator(nn.Module):
    """PatchGAN critic on (pred_patch, other_patch) pairs (cgan/models.py:222-258)."""

    def __init__(self, spectral_norm: bool = None):
        super().__init__()
        if spectral_norm is None:
            spectral_norm = _config_default("spectral_norm")
        self.spectral_norm = bool(spectral_norm)
        layers = {}
        for idx, (ci, co) in zip((0, 2, 5, 8), [(6, 64), (64, 128), (128, 256), (256, 512)]):
            layers[idx] = Conv4x4(ci, co, bias=True, spectral=self.spectral_norm)
        layers[11] = Conv4x4(512, 1)
        self.model = _Holder(layers)
        self.compute_dtype = "fp32"

    def forward(self, pred_patch: torch.Tensor, other_patch: torch.Tensor) -> torch.Tensor:
        from .autograd_nets import discriminator_forward
        return discriminator_forward(self, pred_patch, other_patch)
