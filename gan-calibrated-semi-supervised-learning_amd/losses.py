"""The reference's ``cgan/losses.py`` Python surface (same names, argument meaning and defaults).

The (B,4) box math is negligible work and must stay differentiable for arbitrary
callers, so these are written with torch tensor ops (device-agnostic, autograd-capable).  The training engine does
NOT use them on its hot path: it calls the fused HIP kernel ``gcssl_eiou_fwd_bwd`` (box transform + EIoU + analytic
gradient in one launch).  ``compute_gradient_penalty`` keeps the reference signature and runs the critic through
the HIP kernels via ``models.Discriminator`` (first- and second-order autograd supported there).
"""
from __future__ import annotations

import torch
import torch.nn as nn


class EIoULoss(nn.Module):
    """1 - mean(IoU - rho^2/c^2 - dw^2/(Cw^2+eps) - dh^2/(Ch^2+eps)).  cgan/losses.py:10-73."""

    def __init__(self, eps: float = 1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, pred_boxes, target_boxes):
        px1, py1 = pred_boxes[:, 0] - pred_boxes[:, 2] / 2, pred_boxes[:, 1] - pred_boxes[:, 3] / 2
        px2, py2 = pred_boxes[:, 0] + pred_boxes[:, 2] / 2, pred_boxes[:, 1] + pred_boxes[:, 3] / 2
        tx1, ty1 = target_boxes[:, 0] - target_boxes[:, 2] / 2, target_boxes[:, 1] - target_boxes[:, 3] / 2
        tx2, ty2 = target_boxes[:, 0] + target_boxes[:, 2] / 2, target_boxes[:, 1] + target_boxes[:, 3] / 2
        iw = torch.clamp(torch.min(px2, tx2) - torch.max(px1, tx1), min=0)
        ih = torch.clamp(torch.min(py2, ty2) - torch.max(py1, ty1), min=0)
        inter = iw * ih
        union = (px2 - px1) * (py2 - py1) + (tx2 - tx1) * (ty2 - ty1) - inter
        iou = inter / (union + self.eps)
        ew = torch.max(px2, tx2) - torch.min(px1, tx1)
        eh = torch.max(py2, ty2) - torch.min(py1, ty1)
        c2 = ew ** 2 + eh ** 2
        rho2 = (pred_boxes[:, 0] - target_boxes[:, 0]) ** 2 + (pred_boxes[:, 1] - target_boxes[:, 1]) ** 2
        dw2 = (pred_boxes[:, 2] - target_boxes[:, 2]) ** 2
        dh2 = (pred_boxes[:, 3] - target_boxes[:, 3]) ** 2
        eiou = iou - rho2 / (c2 + self.eps) - dw2 / (ew ** 2 + self.eps) - dh2 / (eh ** 2 + self.eps)
        return 1 - eiou.mean()


class HybridLoss(nn.Module):
    """lambda_iou * EIoU; returns (total, iou_loss).  cgan/losses.py:75-97 (the delta arguments are unused there too)."""

    def __init__(self, lambda_iou: float = 1.0, **kwargs):
        super().__init__()
        self.lambda_iou = lambda_iou
        self.iou_loss = EIoULoss()

    def forward(self, pred_deltas, target_deltas, pred_boxes, target_boxes):
        iou_loss = self.iou_loss(pred_boxes, target_boxes)
        return self.lambda_iou * iou_loss, iou_loss


def smooth_clamp(x, min_val, max_val, temperature: float = 0.5):
    """cgan/losses.py:99-106."""
    center = (min_val + max_val) / 2
    return min_val + (max_val - min_val) * torch.sigmoid((x - center) / temperature)


def apply_delta_to_bbox(bbox, delta, training: bool = True):
    """cgan/losses.py:108-150."""
    rng = 1.5
    d = smooth_clamp(delta, -rng, rng) if training else torch.clamp(delta, -rng, rng)
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


def apply_delta_to_bbox_inference(bbox, delta):
    """The box transform of the reference's inference script, cgan/inference.py:69-89.  It is NOT apply_delta_to_bbox(
    training=False) although its docstring says so: the deltas are clamped to [-2, 2] (not 1.5), exp() is
    not clamped to [-1, 1], and w, h end in [0.01, 0.9] (not [0.02, 0.8]).  Kept as it is so that infer.py reproduces the
    reference's output files; validation during training (cgan/cgan_train_enhanced.py:395-420) uses the other one."""
    d = torch.clamp(delta, -2.0, 2.0)
    cx = torch.clamp(bbox[:, 0] + d[:, 0] * bbox[:, 2], 0.05, 0.95)
    cy = torch.clamp(bbox[:, 1] + d[:, 1] * bbox[:, 3], 0.05, 0.95)
    w = torch.clamp(bbox[:, 2] * torch.exp(d[:, 2]), 0.01, 0.9)
    h = torch.clamp(bbox[:, 3] * torch.exp(d[:, 3]), 0.01, 0.9)
    return torch.stack([cx, cy, w, h], dim=-1)


def iou_metric(pred_boxes, target_boxes, eps: float = 1e-6):
    """cgan/losses.py:152-183."""
    px1, py1 = pred_boxes[:, 0] - pred_boxes[:, 2] / 2, pred_boxes[:, 1] - pred_boxes[:, 3] / 2
    px2, py2 = pred_boxes[:, 0] + pred_boxes[:, 2] / 2, pred_boxes[:, 1] + pred_boxes[:, 3] / 2
    tx1, ty1 = target_boxes[:, 0] - target_boxes[:, 2] / 2, target_boxes[:, 1] - target_boxes[:, 3] / 2
    tx2, ty2 = target_boxes[:, 0] + target_boxes[:, 2] / 2, target_boxes[:, 1] + target_boxes[:, 3] / 2
    iw = torch.clamp(torch.min(px2, tx2) - torch.max(px1, tx1), min=0)
    ih = torch.clamp(torch.min(py2, ty2) - torch.max(py1, ty1), min=0)
    inter = iw * ih
    union = (px2 - px1) * (py2 - py1) + (tx2 - tx1) * (ty2 - ty1) - inter
    return inter / (union + eps)


def compute_gradient_penalty(discriminator, real_samples, fake_samples, device, alpha=None):
    """WGAN-GP penalty, cgan/losses.py:185-233 (same order of operations; ``alpha`` may be supplied for parity
    runs, otherwise it is drawn as torch.rand(B,1,1,1) like the reference).  ``discriminator`` is any callable
    ``(pred, other) -> scores`` that supports double backward, e.g. ``models.Discriminator``."""
    batch_size = real_samples[0].size(0)
    if alpha is None:
        alpha = torch.rand(batch_size, 1, 1, 1, device=device)
    alpha = alpha.expand_as(real_samples[0])
    interpolated_pred = (alpha * real_samples[0] + (1 - alpha) * fake_samples[0]).detach()
    interpolated_other = (alpha * real_samples[1] + (1 - alpha) * fake_samples[1]).detach()
    interpolated_pred.requires_grad_(True)
    interpolated_other.requires_grad_(True)
    d_interpolated = discriminator(interpolated_pred, interpolated_other)
    gradients = torch.autograd.grad(outputs=d_interpolated, inputs=[interpolated_pred, interpolated_other],
                                    grad_outputs=torch.ones_like(d_interpolated), create_graph=True,
                                    retain_graph=True, only_inputs=True)
    gp = gradients[0].view(batch_size, -1)
    go = gradients[1].view(batch_size, -1)
    norm = torch.sqrt(torch.sum(gp ** 2, dim=1) + torch.sum(go ** 2, dim=1) + 1e-12)
    return torch.mean((norm - 1) ** 2)
