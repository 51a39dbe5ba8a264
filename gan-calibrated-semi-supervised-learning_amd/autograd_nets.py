"""Differentiable wrappers (torch.autograd.Function) that run the networks of ``models.py`` on the HIP kernels.
Filled in after the step engine (round 1 milestone order: engine first)."""
from __future__ import annotations


def generator_forward(module, x, masks=None):
    raise NotImplementedError("autograd wrapper under construction")


def discriminator_forward(module, pred, other):
    raise NotImplementedError("autograd wrapper under construction")
