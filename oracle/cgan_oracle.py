"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A from-scratch, functional CPU restatement (torch CPU ops + autograd, fp32) of the hot path of
1213ray/GAN-Calibrated-Semi-Supervised-Learning: the WGAN-GP cGAN training step.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the checker / the timed CPU baseline.  The product path
(``gan-calibrated-semi-supervised-learning_amd``) never imports it and fails loudly when the HIP
library is missing.

Pinning: this restatement is checked against golden vectors generated in the dev container by
importing the reference's own ``cgan/models.py`` + ``cgan/losses.py`` (script:
``tests/golden/make_golden.py``; test: ``tests/test_oracle_vs_golden.py``).  The reference has no
tests/golden vectors of its own.

Every function cites the reference lines it restates (paths relative to the reference root).
State is held in plain dicts of tensors keyed exactly like the reference ``state_dict()``s.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
D_IDX = (0, 2, 5, 8)
IN_EPS = 1e-5          # nn.InstanceNorm2d default eps (cgan/models.py:59,73,114,241)
LRELU = 0.2            # cgan/models.py:60,242


# ----------------------------------------------------------------------------------------------
# spectral norm (legacy torch.nn.utils.spectral_norm hook, used at cgan/models.py:237-238)
# ----------------------------------------------------------------------------------------------
def sn_weight(w_orig: Tensor, u: Tensor, v: Tensor, train: bool) -> Tensor:
    """One power iteration (train mode, in place on u,v, no grad), sigma = u^T W v, W/sigma.

    eps=1e-12 in both normalisations; sigma is differentiable w.r.t. ``w_orig`` with u,v constant.
    """
    wm = w_orig.reshape(w_orig.shape[0], -1)
    if train:
        with torch.no_grad():
            vn = torch.mv(wm.t(), u)
            vn = vn / vn.norm().clamp_min(1e-12)
            un = torch.mv(wm, vn)
            un = un / un.norm().clamp_min(1e-12)
            v.copy_(vn)
            u.copy_(un)
    sigma = torch.dot(u.detach().clone(), torch.mv(wm, v.detach().clone()))
    return w_orig / sigma


# ----------------------------------------------------------------------------------------------
# Discriminator (cgan/models.py:222-258)
# ----------------------------------------------------------------------------------------------
def d_forward(sd: Dict[str, Tensor], pred: Tensor, other: Tensor, train: bool = True,
              spectral_norm: bool = True, taps: Optional[dict] = None) -> Tensor:
    """cat(pred,other) -> 4x [SN-conv k4s2p1 + bias (-> InstanceNorm) -> LeakyReLU(0.2)]
    -> conv k4 s1 p1 (no bias).  cgan/models.py:245-258."""
    x = torch.cat([pred, other], dim=1)
    for li, i in enumerate(D_IDX):
        w = sd[f"model.{i}.weight_orig"] if spectral_norm else sd[f"model.{i}.weight"]
        if spectral_norm:
            w = sn_weight(w, sd[f"model.{i}.weight_u"], sd[f"model.{i}.weight_v"], train)
        x = F.conv2d(x, w, sd[f"model.{i}.bias"], stride=2, padding=1)
        if li > 0:
            x = F.instance_norm(x, eps=IN_EPS)
        x = F.leaky_relu(x, LRELU)
        if taps is not None:
            taps[f"d.a{li + 1}"] = x.detach().clone()
    return F.conv2d(x, sd["model.11.weight"], None, stride=1, padding=1)


# ----------------------------------------------------------------------------------------------
# GeneratorUNet (cgan/models.py:54-141)
# ----------------------------------------------------------------------------------------------
def _drop(x: Tensor, mask: Optional[Tensor]) -> Tensor:
    # nn.Dropout(0.5) in train mode: x * keep / (1-p)
    return x if mask is None else x * (mask.to(x.dtype) * 2.0)


def g_forward(sd: Dict[str, Tensor], x: Tensor, delta_scale: float,
              masks: Optional[Sequence[Tensor]] = None, taps: Optional[dict] = None) -> Tensor:
    """masks=None is eval mode (dropout off); otherwise the three keep-masks of the Dropout(0.5)
    sites in order down4, up1, up2 (cgan/models.py:106,109,110)."""
    def down(x, w, norm, mask=None):
        x = F.conv2d(x, w, None, stride=2, padding=1)
        if norm:
            x = F.instance_norm(x, eps=IN_EPS)
        return _drop(F.leaky_relu(x, LRELU), mask)

    def up(x, w, mask=None):
        x = F.conv_transpose2d(x, w, None, stride=2, padding=1)
        return _drop(F.relu(F.instance_norm(x, eps=IN_EPS)), mask)

    m = masks if masks is not None else (None, None, None)
    d1 = down(x, sd["down1.model.0.weight"], False)
    d2 = down(d1, sd["down2.model.0.weight"], True)
    d3 = down(d2, sd["down3.model.0.weight"], True)
    d4 = down(d3, sd["down4.model.0.weight"], True, m[0])
    u1 = torch.cat((up(d4, sd["up1.model.0.weight"], m[1]), d3), 1)
    u2 = torch.cat((up(u1, sd["up2.model.0.weight"], m[2]), d2), 1)
    u3 = torch.cat((up(u2, sd["up3.model.0.weight"]), d1), 1)
    u4 = up(u3, sd["up4.0.weight"])
    pooled = u4.mean(dim=(2, 3))                               # AdaptiveAvgPool2d(1)+Flatten
    raw = torch.tanh(F.linear(pooled, sd["fc_delta.1.weight"], sd["fc_delta.1.bias"]))
    if taps is not None:
        for k, t in (("g.d1", d1), ("g.d2", d2), ("g.d3", d3), ("g.d4", d4),
                     ("g.u1", u1), ("g.u2", u2), ("g.u3", u3), ("g.u4", u4), ("g.pooled", pooled)):
            taps[k] = t.detach().clone()
    return raw * delta_scale


GS_CONV_IDX = (0, 3, 7, 10, 14, 17, 21, 24)      # nn.Conv2d positions in GeneratorSimpleRegressor.features (models.py:161-197)
GS_FC_IDX = (2, 5, 8)                             # nn.Linear positions in .regressor (models.py:200-211)


def g_simple_forward(sd: Dict[str, Tensor], x: Tensor, delta_scale: float,
                     masks: Optional[Sequence[Tensor]] = None, taps: Optional[dict] = None) -> Tensor:
    """GeneratorSimpleRegressor.forward (cgan/models.py:147-216): four blocks of 2x [Conv3x3(+bias), InstanceNorm, ReLU] +
    MaxPool2d(2,2), then AdaptiveAvgPool2d(1) -> Linear(512,256) ReLU Dropout -> Linear(256,64) ReLU Dropout ->
    Linear(64,4) -> Tanh, times delta_scale.  masks=None is eval mode; otherwise the two keep-masks [B,256], [B,64]."""
    h = x
    for j, i in enumerate(GS_CONV_IDX):
        h = F.conv2d(h, sd[f"features.{i}.weight"], sd[f"features.{i}.bias"], padding=1)
        h = F.relu(F.instance_norm(h, eps=IN_EPS))
        if taps is not None:
            taps[f"gs.a{j}"] = h.detach().clone()
        if j & 1:
            h = F.max_pool2d(h, 2, 2)
    feat = h.mean(dim=(2, 3))
    m = masks if masks is not None else (None, None)
    h1 = _drop(F.relu(F.linear(feat, sd["regressor.2.weight"], sd["regressor.2.bias"])), m[0])
    h2 = _drop(F.relu(F.linear(h1, sd["regressor.5.weight"], sd["regressor.5.bias"])), m[1])
    raw = torch.tanh(F.linear(h2, sd["regressor.8.weight"], sd["regressor.8.bias"]))
    if taps is not None:
        taps["gs.feat"], taps["gs.h1"], taps["gs.h2"] = feat.detach().clone(), h1.detach().clone(), h2.detach().clone()
    return raw * delta_scale


# ----------------------------------------------------------------------------------------------
# losses (cgan/losses.py)
# ----------------------------------------------------------------------------------------------
def smooth_clamp(x: Tensor, lo: float, hi: float, temperature: float = 0.5) -> Tensor:
    """cgan/losses.py:99-106."""
    center = (lo + hi) / 2
    return lo + (hi - lo) * torch.sigmoid((x - center) / temperature)


def apply_delta_to_bbox(bbox: Tensor, delta: Tensor, training: bool = True) -> Tensor:
    """cgan/losses.py:108-150."""
    r = 1.5
    d = smooth_clamp(delta, -r, r) if training else torch.clamp(delta, -r, r)
    cx = bbox[:, 0] + d[:, 0] * bbox[:, 2]
    cy = bbox[:, 1] + d[:, 1] * bbox[:, 3]
    w = bbox[:, 2] * torch.exp(torch.clamp(d[:, 2], -1.0, 1.0))
    h = bbox[:, 3] * torch.exp(torch.clamp(d[:, 3], -1.0, 1.0))
    if training:
        cx, cy = smooth_clamp(cx, 0.05, 0.95), smooth_clamp(cy, 0.05, 0.95)
        w, h = smooth_clamp(w, 0.02, 0.8), smooth_clamp(h, 0.02, 0.8)
    else:
        cx, cy = torch.clamp(cx, 0.05, 0.95), torch.clamp(cy, 0.05, 0.95)
        w, h = torch.clamp(w, 0.02, 0.8), torch.clamp(h, 0.02, 0.8)
    return torch.stack([cx, cy, w, h], dim=-1)


def _corners(b: Tensor):
    return (b[:, 0] - b[:, 2] / 2, b[:, 1] - b[:, 3] / 2, b[:, 0] + b[:, 2] / 2, b[:, 1] + b[:, 3] / 2)


def iou_metric(pred: Tensor, target: Tensor, eps: float = 1e-6) -> Tensor:
    """cgan/losses.py:152-183."""
    px1, py1, px2, py2 = _corners(pred)
    tx1, ty1, tx2, ty2 = _corners(target)
    iw = torch.clamp(torch.min(px2, tx2) - torch.max(px1, tx1), min=0)
    ih = torch.clamp(torch.min(py2, ty2) - torch.max(py1, ty1), min=0)
    inter = iw * ih
    union = (px2 - px1) * (py2 - py1) + (tx2 - tx1) * (ty2 - ty1) - inter
    return inter / (union + eps)


def eiou_loss(pred: Tensor, target: Tensor, eps: float = 1e-6) -> Tensor:
    """1 - mean(IoU - rho^2/c^2 - dw^2/(Cw^2+eps) - dh^2/(Ch^2+eps)).  cgan/losses.py:19-73."""
    px1, py1, px2, py2 = _corners(pred)
    tx1, ty1, tx2, ty2 = _corners(target)
    iou = iou_metric(pred, target, eps)
    ew = torch.max(px2, tx2) - torch.min(px1, tx1)
    eh = torch.max(py2, ty2) - torch.min(py1, ty1)
    c2 = ew ** 2 + eh ** 2
    rho2 = (pred[:, 0] - target[:, 0]) ** 2 + (pred[:, 1] - target[:, 1]) ** 2
    dw2 = (pred[:, 2] - target[:, 2]) ** 2
    dh2 = (pred[:, 3] - target[:, 3]) ** 2
    eiou = iou - rho2 / (c2 + eps) - dw2 / (ew ** 2 + eps) - dh2 / (eh ** 2 + eps)
    return 1 - eiou.mean()


def gradient_penalty(sd_d: Dict[str, Tensor], real: Sequence[Tensor], fake: Sequence[Tensor],
                     alpha: Tensor, spectral_norm: bool = True, taps: Optional[dict] = None) -> Tensor:
    """cgan/losses.py:185-233 with alpha (B,1,1,1) supplied instead of drawn (:199)."""
    b = real[0].shape[0]
    a = alpha.expand_as(real[0])
    ip = (a * real[0] + (1 - a) * fake[0]).detach().requires_grad_(True)
    io = (a * real[1] + (1 - a) * fake[1]).detach().requires_grad_(True)
    d_i = d_forward(sd_d, ip, io, True, spectral_norm)
    g = torch.autograd.grad(d_i, [ip, io], torch.ones_like(d_i), create_graph=True,
                            retain_graph=True, only_inputs=True)
    nrm = torch.sqrt((g[0].reshape(b, -1) ** 2).sum(1) + (g[1].reshape(b, -1) ** 2).sum(1) + 1e-12)
    if taps is not None:
        taps["d_interp"] = d_i.detach().clone()
        taps["gp_grad_pred"] = g[0].detach().clone()
        taps["gp_grad_other"] = g[1].detach().clone()
        taps["gp_norm"] = nrm.detach().clone()
    return ((nrm - 1) ** 2).mean()


# ----------------------------------------------------------------------------------------------
# optimiser pieces (torch.nn.utils.clip_grad_norm_ + torch.optim.Adam as called at
# cgan/cgan_train_enhanced.py:256-257,331-332,368-369)
# ----------------------------------------------------------------------------------------------
def clip_grad_norm_(grads: List[Tensor], max_norm: float = 1.0) -> Tensor:
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


class Adam:
    """torch.optim.Adam (no amsgrad, no weight decay, eps 1e-8) restated."""

    def __init__(self, params: List[Tensor], lr: float, betas=(0.5, 0.999), eps: float = 1e-8):
        self.params, self.lr, self.b1, self.b2, self.eps = params, lr, betas[0], betas[1], eps
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]
        self.t = 0

    @torch.no_grad()
    def step(self, grads: List[Tensor]):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2s = math.sqrt(1 - self.b2 ** self.t)
        for p, g, m, v in zip(self.params, grads, self.m, self.v):
            m.lerp_(g, 1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            p.addcdiv_(m, (v.sqrt() / bc2s).add_(self.eps), value=-self.lr / bc1)


# ----------------------------------------------------------------------------------------------
# the training step (cgan/cgan_train_enhanced.py:304-369)
# ----------------------------------------------------------------------------------------------
D_PARAM_KEYS = [k for i in D_IDX for k in (f"model.{i}.bias", f"model.{i}.weight_orig")] + ["model.11.weight"]
G_PARAM_KEYS = ["down1.model.0.weight", "down2.model.0.weight", "down3.model.0.weight",
                "down4.model.0.weight", "up1.model.0.weight", "up2.model.0.weight",
                "up3.model.0.weight", "up4.0.weight", "fc_delta.1.weight", "fc_delta.1.bias"]
GS_PARAM_KEYS = ([k for i in GS_CONV_IDX for k in (f"features.{i}.weight", f"features.{i}.bias")]
                 + [k for i in GS_FC_IDX for k in (f"regressor.{i}.weight", f"regressor.{i}.bias")])   # named_parameters() order


class StepOracle:
    """Holds G/D state + two Adams and runs reference-ordered iterations.

    ``refine_fn(delta, k)`` stands in for ``get_refined_patch_batch``
    (cgan/cgan_train_enhanced.py:37-137): it must return a (B,3,S,S) tensor with no autograd
    edge to ``delta``.
    """

    def __init__(self, sd_g: Dict[str, Tensor], sd_d: Dict[str, Tensor], lr: float = 2e-4,
                 betas=(0.5, 0.999), delta_scale: float = 0.3, lambda_gp: float = 1.0,
                 lambda_iou: float = 1.0, n_critic: int = 2, generator_type: str = "unet", spectral_norm: bool = True):
        # generator_type: get_generator(), cgan/cgan_train_enhanced.py:26-31
        # spectral_norm: Discriminator(spectral_norm=...), cgan/models.py:228-238 (config.yaml `spectral_norm`): plain convs
        # under the keys model.N.weight when off
        self.sn = spectral_norm
        self.d_keys = D_PARAM_KEYS if spectral_norm else [k.replace("weight_orig", "weight") for k in D_PARAM_KEYS]
        self.g_keys = GS_PARAM_KEYS if generator_type == "simple" else G_PARAM_KEYS
        self.g_fwd = g_simple_forward if generator_type == "simple" else g_forward
        self.g = {k: v.clone().float() for k, v in sd_g.items()}
        self.d = {k: v.clone().float() for k, v in sd_d.items()}
        for k in self.g_keys:
            self.g[k].requires_grad_(True)
        for k in self.d_keys:
            self.d[k].requires_grad_(True)
        self.opt_g = Adam([self.g[k] for k in self.g_keys], lr, betas)
        self.opt_d = Adam([self.d[k] for k in self.d_keys], lr, betas)
        self.delta_scale, self.lambda_gp, self.lambda_iou, self.n_critic = \
            delta_scale, lambda_gp, lambda_iou, n_critic

    def iteration(self, pred: Tensor, gt: Tensor, delta_true: Tensor, pred_box: Tensor,
                  refine_fn: Callable[[Tensor, int], Tensor], alphas: Sequence[Tensor],
                  masks: Sequence[Optional[Sequence[Tensor]]], taps: Optional[dict] = None) -> dict:
        log = {"d_loss": [], "gp": [], "wd": [], "d_grad_norm": []}
        for c in range(self.n_critic):                                       # :304
            tp = taps if (taps is not None and c == 0) else None
            for k in self.d_keys:                                            # :305 zero_grad
                self.d[k].grad = None
            real = d_forward(self.d, pred, gt, True, self.sn)                # :308
            with torch.no_grad():                                            # :311-315
                delta_det = self.g_fwd(self.g, pred, self.delta_scale, masks[c])
                refined = refine_fn(delta_det, c)
            fake = d_forward(self.d, pred, refined, True, self.sn)           # :316
            gp = gradient_penalty(self.d, (pred, gt), (pred, refined), alphas[c], self.sn, taps=tp)  # :319-324
            wd = real.mean() - fake.mean()                                   # :327
            d_loss = -wd + self.lambda_gp * gp                               # :328
            grads = list(torch.autograd.grad(d_loss, [self.d[k] for k in self.d_keys]))  # :330
            if tp is not None:
                tp["real_validity"], tp["fake_validity"] = real.detach().clone(), fake.detach().clone()
                tp["delta_detached"] = delta_det.detach().clone()
                for k, g in zip(self.d_keys, grads):
                    tp[f"d.grad.{k}"] = g.detach().clone()
            total = clip_grad_norm_(grads, 1.0)                              # :331
            self.opt_d.step(grads)                                           # :332
            log["d_loss"].append(float(d_loss)); log["gp"].append(float(gp))
            log["wd"].append(float(wd)); log["d_grad_norm"].append(float(total))
        # ---- generator update (:345-369) ----
        for k in self.g_keys:
            self.g[k].grad = None
        delta_pred = self.g_fwd(self.g, pred, self.delta_scale, masks[self.n_critic])       # :348
        cal = apply_delta_to_bbox(pred_box, delta_pred, True)                # :351
        gtb = apply_delta_to_bbox(pred_box, delta_true, True)                # :352
        loss_iou = eiou_loss(cal, gtb)                                       # :353-355
        loss_reg = self.lambda_iou * loss_iou
        refined_g = refine_fn(delta_pred.detach(), self.n_critic)            # :358-360 (no edge)
        with torch.no_grad():                                                # value only
            fake_g = d_forward(self.d, pred, refined_g, True, self.sn)       # :361 (advances u,v)
        loss_wgan = -fake_g.mean()                                           # :362
        grads = list(torch.autograd.grad(loss_reg, [self.g[k] for k in self.g_keys]))         # :366
        if taps is not None:
            taps["delta_pred"] = delta_pred.detach().clone()
            taps["fake_validity_for_G"] = fake_g.detach().clone()
            for k, g in zip(self.g_keys, grads):
                taps[f"g.grad.{k}"] = g.detach().clone()
        total_g = clip_grad_norm_(grads, 1.0)                                # :368
        self.opt_g.step(grads)                                               # :369
        log.update(loss_iou=float(loss_iou), loss_wgan=float(loss_wgan),
                   loss_g=float(loss_reg + loss_wgan), g_grad_norm=float(total_g),
                   delta_pred=delta_pred.detach().clone())
        return log
