"""torch.autograd glue for the drop-in modules of ``models.py``.

The conv / InstanceNorm / activation stacks run on the HIP kernels through two network-level autograd Functions per
net (forward, and a *differentiable* backward), so that the reference's own call pattern works unchanged:

    d = D(ip, io); g = autograd.grad(d, [ip, io], ones, create_graph=True); gp(g).backward()    (cgan/losses.py:210-231)

* ``DNetFn.backward`` calls ``DNetBwdFn.apply`` -- the first-order backward is itself a recorded op;
* ``DNetBwdFn.backward`` is the hand-written double backward (same schedule as ``engine.StepEngine.d_compute`` and
  ``oracle/manual_step.py``): a forward-conv chain of the adjoints, wgrads against the first-order gradients, the
  InstanceNorm double-backward terms, and those terms pushed back through the saved forward activations.
* Spectral norm's sigma = u^T W v and W/sigma are a handful of torch ops on the master weights (so autograd
  differentiates the quotient to any order); the power iteration itself runs on the HIP kernel, in place on the
  module's ``weight_u/weight_v`` buffers like the reference's hook.

This path allocates its buffers per call (arbitrary batch sizes, several live forwards); the preallocated, batched,
graph-captured path is ``engine.StepEngine``.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib, ops
from .ops import LRELU, RELU

D_CH = [(6, 64), (64, 128), (128, 256), (256, 512)]
G_DOWN = [(3, 64), (64, 128), (128, 256), (256, 512)]
G_UP = [(512, 256), (512, 128), (256, 64), (128, 64)]
F32 = torch.float32


def _pad8(c):
    return max(8, c)


def _check_input(x: torch.Tensor, name: str):
    if not x.is_cuda:
        raise RuntimeError(f"{name}: the HIP path needs CUDA/HIP tensors (there is no CPU fallback)")
    S = x.shape[-1]
    if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != S or S < 32 or S & (S - 1):
        raise ValueError(f"{name}: expected (B,3,S,S) with S a power of two >= 32, got {tuple(x.shape)}")
    return x.contiguous().float()


def _wgrad(x, dy, cin_p, cin_real, cout, shape, extra=None):
    """weight gradient [Cout][Cin_real][4][4] of conv(x)->dy (+ optional second (x, dy) pair into the same reduce)."""
    N, Hi = x.shape[0], x.shape[1]
    pairs = [(x, dy)] + ([extra] if extra is not None else [])
    ns = [ops.wgrad_splits(p[0].shape[0], Hi, Hi, cin_p, cout) for p in pairs]
    slab = torch.empty(sum(ns), cout, 16, cin_p, device=x.device, dtype=F32)
    off = 0
    for (xx, dd), n in zip(pairs, ns):
        ops.conv_wgrad(xx, dd, slab[off:], cin_p, cout)
        off += n
    dw = torch.empty(shape, device=x.device, dtype=F32)
    ops.wgrad_reduce(slab, sum(ns), dw, cout, cin_p, cin_real)
    return dw


# ======================================================================================================== critic
class DTape:
    """One critic forward (single sample group) with everything its first- and second-order backward need."""

    def __init__(self, code: int):
        self.code, self.T = code, _lib.torch_dtype(code)

    def forward(self, pred, other, Ws: List[torch.Tensor], bs: List[torch.Tensor], W5: torch.Tensor):
        dev, T = pred.device, self.T
        B, S = pred.shape[0], pred.shape[-1]
        self.B, self.S = B, S
        self.x0 = torch.empty(B, S, S, 8, device=dev, dtype=T)
        ops.pack_pair(pred, other, self.x0)
        self.wf, self.wt = [], []
        for (cin, cout), w in zip(D_CH, Ws):
            cp = _pad8(cin)
            wf = torch.empty(cout, 16, cp, device=dev, dtype=T); wt = torch.empty(cp, 16, cout, device=dev, dtype=T)
            ops.prep_conv_weight(w.detach().contiguous(), wf, wt, cout, cin, cp, self.code)
            self.wf.append(wf); self.wt.append(wt)
        self.w5p = torch.empty(16, 512, device=dev, dtype=F32)
        ops.prep_c5_weight(W5.detach().contiguous(), self.w5p)
        self.a, self.z, self.mean, self.rstd = [], [None], [None], [None]
        src = self.x0
        for l, (cin, cout) in enumerate(D_CH):
            s = S >> (l + 1)
            bias = bs[l].detach().contiguous() if bs[l] is not None else None
            if l == 0:
                a = torch.empty(B, s, s, cout, device=dev, dtype=T)
                ops.conv_fwd(src, self.wf[0], a, 8, cout, bias=bias, act=LRELU)
            else:
                z = torch.empty(B, s, s, cout, device=dev, dtype=F32)
                ops.conv_fwd(src, self.wf[l], z, cin, cout, bias=bias)
                a = torch.empty(B, s, s, cout, device=dev, dtype=T)
                mean = torch.empty(B, cout, device=dev); rstd = torch.empty(B, cout, device=dev)
                ops.in_act_fwd(z, a, mean, rstd, cout, LRELU)
                self.z.append(z); self.mean.append(mean); self.rstd.append(rstd)
            self.a.append(a)
            src = a
        self.h = (S >> 4) - 1
        out = torch.empty(B, self.h, self.h, device=dev, dtype=F32)
        ops.c5_fwd(self.a[3], self.w5p, out)
        self.ws = self.ws_g = torch.empty(2 * B * 512, device=dev, dtype=F32)    # (the step engine keeps a separate one for the generator)
        return out.view(B, 1, self.h, self.h)

    def _backward_chain(self, da3, zt=None, keep=False):
        """dgrad/wgrad chain from the gradient wrt a4 (`da3`, fp32 NHWC) down to the input; returns
        (dx0 fp32 NHWC8, [dW0..3], [db0..3]); with keep=True remembers the first-order tensors for double backward."""
        dev, B, S = da3.device, self.B, self.S
        dW, db = [None] * 4, [None] * 4
        da = da3
        if keep:
            self.gb_a, self.gb_zs = [None] * 4, [None] * 4
        for l in (3, 2, 1, 0):
            cin, cout = D_CH[l]
            cp, s = _pad8(cin), S >> (l + 1)
            dzs = torch.empty(B, s, s, cout, device=dev, dtype=self.T)
            db[l] = torch.zeros(cout, device=dev)
            if l > 0:
                ops.in_act_bwd(self.z[l], self.mean[l], self.rstd[l], dzs, cout, LRELU, da=da,
                               zt=None if zt is None else zt[l], zt_n0=0, dbias=db[l], ws=self.ws)
            else:
                ops.act_bwd(da, self.a[0], dzs, 64, dbias=db[0])
            if keep:
                self.gb_a[l], self.gb_zs[l] = da, dzs
            xin = self.x0 if l == 0 else self.a[l - 1]
            dW[l] = (xin, dzs)                     # weight gradient is formed by the caller (possibly fused with a second pair)
            nxt = torch.empty(B, 2 * s, 2 * s, cp if l == 0 else cin, device=dev, dtype=F32)
            ops.conv_dgrad(dzs, self.wt[l], nxt, cp if l == 0 else cin, cout)
            da = nxt
        return da, dW, db

    def backward(self, d_out: torch.Tensor):
        dev, B = d_out.device, self.B
        self.d_out = d_out.reshape(B, self.h, self.h).contiguous().float()
        da3 = torch.empty(B, self.S >> 4, self.S >> 4, 512, device=dev, dtype=F32)
        ops.c5_dgrad(da3, self.w5p, dout=self.d_out)
        dw5 = torch.zeros(512, 16, device=dev)
        ops.c5_wgrad(self.a[3], dw5, 512, dout=self.d_out)
        dx0, pairs, db = self._backward_chain(da3, keep=True)
        dW = [_wgrad(x, dy, _pad8(ci), ci, co, (co, ci, 4, 4)) for (x, dy), (ci, co) in zip(pairs, D_CH)]
        d_pred = torch.empty(B, 3, self.S, self.S, device=dev); d_other = torch.empty_like(d_pred)
        ops.unpack_grad(dx0, d_pred, d_other)
        return d_pred, d_other, dW, db, dw5.view(1, 512, 4, 4)

    def double_backward(self, gt_pred: Optional[torch.Tensor], gt_other: Optional[torch.Tensor]):
        """Adjoint of backward(): given adjoints of (d_pred, d_other) return (adjoint of d_out, dW, db, dW5)."""
        dev, B, S, T = self.x0.device, self.B, self.S, self.T
        zero = torch.zeros(B, 3, S, S, device=dev)
        gt_x = torch.empty(B, S, S, 8, device=dev, dtype=T)
        ops.pack_pair((gt_pred if gt_pred is not None else zero).contiguous().float(),
                      (gt_other if gt_other is not None else zero).contiguous().float(), gt_x)
        src, chain_pairs, zt = gt_x, [], [None] * 4
        for l, (cin, cout) in enumerate(D_CH):
            cp, s = _pad8(cin), S >> (l + 1)
            gt_z = torch.empty(B, s, s, cout, device=dev, dtype=F32)
            ops.conv_fwd(src, self.wf[l], gt_z, cp, cout)
            chain_pairs.append((src, self.gb_zs[l]))
            gt_a = torch.empty(B, s, s, cout, device=dev, dtype=T)
            if l == 0:
                ops.act_bwd(gt_z, self.a[0], gt_a, 64)
            else:
                zt[l] = torch.empty(B, s, s, cout, device=dev, dtype=F32)
                ops.in_dbl_bwd(self.gb_a[l], gt_z, None, self.z[l], self.mean[l], self.rstd[l], gt_a, zt[l], cout, LRELU)
            src = gt_a
        gt_out = torch.empty(B, self.h, self.h, device=dev, dtype=F32)
        ops.c5_fwd(src, self.w5p, gt_out)
        dw5 = torch.zeros(512, 16, device=dev)
        ops.c5_wgrad(src, dw5, 512, dout=self.d_out)
        # the double-backward terms zt re-enter the backward of the forward pass (no gradient arrives from the output)
        da3 = torch.zeros(B, S >> 4, S >> 4, 512, device=dev, dtype=F32)
        _, pairs, db = self._backward_chain(da3, zt=zt, keep=False)
        dW = [_wgrad(x, dy, _pad8(ci), ci, co, (co, ci, 4, 4), extra=cp_)
              for (x, dy), cp_, (ci, co) in zip(pairs, chain_pairs, D_CH)]
        return gt_out.view(B, 1, self.h, self.h), dW, db, dw5.view(1, 512, 4, 4)


class DNetBwdFn(Function):
    """First-order backward of the critic as a differentiable op (its backward is the WGAN-GP double backward)."""

    @staticmethod
    def forward(ctx, d_out, tape, *params):
        d_pred, d_other, dW, db, dw5 = tape.backward(d_out)
        ctx.tape = tape
        ctx.set_materialize_grads(False)
        outs = [d_pred, d_other]
        for l in range(4):
            outs += [dW[l], db[l]]
        outs.append(dw5)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_pred, g_other, *g_w):
        if any(g is not None for g in g_w):
            raise NotImplementedError("differentiating through the critic's WEIGHT gradients is not supported "
                                      "(the reference never does; only input gradients are, for WGAN-GP)")
        gt_out, dW, db, dw5 = ctx.tape.double_backward(g_pred, g_other)
        grads = []
        for l in range(4):
            grads += [dW[l], db[l]]
        return (gt_out, None, *grads, dw5)


class DNetFn(Function):
    @staticmethod
    def forward(ctx, pred, other, code, *params):
        Ws, bs, W5 = list(params[0:8:2]), list(params[1:8:2]), params[8]
        tape = DTape(code)
        out = tape.forward(pred, other, Ws, bs, W5)
        ctx.tape = tape
        ctx.save_for_backward(*params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        outs = DNetBwdFn.apply(d_out.contiguous(), ctx.tape, *ctx.saved_tensors)
        return (outs[0], outs[1], None, *outs[2:])


_sn_states = {}


def discriminator_forward(module, pred: torch.Tensor, other: torch.Tensor) -> torch.Tensor:
    """Discriminator.forward (cgan/models.py:255-258) on the HIP kernels, differentiable to second order."""
    pred, other = _check_input(pred, "pred_patch"), _check_input(other, "other_patch")
    code = _lib.dtype_code(getattr(module, "compute_dtype", "fp32"))
    layers = [getattr(module.model, str(i)) for i in (0, 2, 5, 8)]
    params = []
    if module.spectral_norm:
        if module.training:                      # one power iteration, in place on weight_u / weight_v (no grad)
            with torch.no_grad():
                key = id(module)
                ptrs = tuple(t.data_ptr() for L in layers for t in (L.weight_orig, L.weight_u, L.weight_v))
                st = _sn_states.get(key)
                if st is None or st[0] != ptrs:
                    st = (ptrs, ops.SnState([L.weight_orig.detach() for L in layers], [L.weight_u for L in layers],
                                            [L.weight_v for L in layers], 1, pred.device))
                    _sn_states[key] = st
                st[1].iterate(0, True)
        for L in layers:
            w = L.weight_orig
            sigma = torch.dot(L.weight_u.detach(), torch.mv(w.reshape(w.shape[0], -1), L.weight_v.detach()))
            params += [w / sigma, L.bias]
    else:
        for L in layers:
            params += [L.weight, L.bias]
    params.append(getattr(module.model, "11").weight)
    return DNetFn.apply(pred, other, code, *params)


# ======================================================================================================== generator
class GTape:
    def __init__(self, code: int):
        self.code, self.T = code, _lib.torch_dtype(code)

    def forward(self, x, Wd, Wu, fc_w, fc_b, scale, masks):
        dev, T = x.device, self.T
        B, S = x.shape[0], x.shape[-1]
        self.B, self.S, self.scale = B, S, scale

        def act(s, c, dt=T):
            return torch.empty(B, s, s, c, device=dev, dtype=dt)
        self.x8 = act(S, 8)
        ops.pack_pair(x, None, self.x8)
        self.gd_wf, self.gd_wt, self.gu_wf, self.gu_wt = [], [], [], []
        for (cin, cout), w in zip(G_DOWN, Wd):
            cp = _pad8(cin)
            wf = torch.empty(cout, 16, cp, device=dev, dtype=T); wt = torch.empty(cp, 16, cout, device=dev, dtype=T)
            ops.prep_conv_weight(w.detach().contiguous(), wf, wt, cout, cin, cp, self.code)
            self.gd_wf.append(wf); self.gd_wt.append(wt)
        for (cint, coutt), w in zip(G_UP, Wu):
            wf = torch.empty(cint, 16, coutt, device=dev, dtype=T); wt = torch.empty(coutt, 16, cint, device=dev, dtype=T)
            ops.prep_conv_weight(w.detach().contiguous(), wf, wt, cint, coutt, coutt, self.code)
            self.gu_wf.append(wf); self.gu_wt.append(wt)
        self.cat3, self.cat2, self.cat1 = act(S // 2, 128), act(S // 4, 256), act(S // 8, 512)
        self.d4 = act(S // 16, 512)
        d1, d2, d3 = self.cat3[..., 64:], self.cat2[..., 128:], self.cat1[..., 256:]
        self.masks = None
        if masks is not None:
            self.masks = [m.permute(0, 2, 3, 1).contiguous().to(torch.uint8) for m in masks]
        mk = self.masks if self.masks is not None else [None, None, None]
        self.zd, self.dstat = [None], [None]
        ops.conv_fwd(self.x8, self.gd_wf[0], d1, 8, 64, act=LRELU)
        srcs, dsts = [d1, d2, d3], [d2, d3, self.d4]
        for k in (1, 2, 3):
            cin, cout = G_DOWN[k]
            z = act(S >> (k + 1), cout, F32)
            ops.conv_fwd(srcs[k - 1], self.gd_wf[k], z, cin, cout)
            mean = torch.empty(B, cout, device=dev); rstd = torch.empty(B, cout, device=dev)
            ops.in_act_fwd(z, dsts[k - 1], mean, rstd, cout, LRELU, mask=mk[0] if k == 3 else None)
            self.zd.append(z); self.dstat.append((mean, rstd))
        self.ins = [self.d4, self.cat1, self.cat2, self.cat3]
        self.u4 = act(S, 64)
        outs = [self.cat1[..., :256], self.cat2[..., :128], self.cat3[..., :64], self.u4]
        self.zu, self.ustat = [], []
        poolsum = torch.zeros(B, 64, device=dev)
        for k, (cint, coutt) in enumerate(G_UP):
            z = act(S >> (3 - k), coutt, F32)
            ops.conv_dgrad(self.ins[k], self.gu_wt[k], z, coutt, cint)
            mean = torch.empty(B, coutt, device=dev); rstd = torch.empty(B, coutt, device=dev)
            ops.in_act_fwd(z, outs[k], mean, rstd, coutt, RELU, mask=mk[k + 1] if k < 2 else None,
                           pool=poolsum if k == 3 else None)
            self.zu.append(z); self.ustat.append((mean, rstd))
        self.pooled = torch.empty(B, 64, device=dev); self.traw = torch.empty(B, 4, device=dev)
        delta = torch.empty(B, 4, device=dev)
        self.fc_w = fc_w.detach().contiguous()
        ops.pool_fc_tanh_fwd(self.u4, self.fc_w, fc_b.detach().contiguous(), scale, self.pooled, self.traw, delta,
                             pool_sum=poolsum)
        self.ws = torch.empty(2 * B * 512, device=dev)
        return delta

    def backward(self, g_delta):
        dev, B, S, T = g_delta.device, self.B, self.S, self.T
        mk = self.masks if self.masks is not None else [None, None, None]

        def act(s, c, dt=F32):
            return torch.empty(B, s, s, c, device=dev, dtype=dt)
        dfc_w = torch.zeros(4, 64, device=dev); dfc_b = torch.zeros(4, device=dev); dab = torch.empty(B, 64, device=dev)
        ops.head_bwd(g_delta.contiguous().float(), self.traw, self.pooled, self.fc_w, self.scale, B, S * S, dfc_w, dfc_b, dab)
        dcat = [act(S // 16, 512), act(S // 8, 512), act(S // 4, 256), act(S // 2, 128)]
        dWu, dWd = [None] * 4, [None] * 4
        for k in (3, 2, 1, 0):
            cint, coutt = G_UP[k]
            dz = act(S >> (3 - k), coutt, T)
            mean, rstd = self.ustat[k]
            if k == 3:
                ops.in_act_bwd(self.zu[3], mean, rstd, dz, coutt, RELU, da_bcast=dab, ws=self.ws)
            else:
                ops.in_act_bwd(self.zu[k], mean, rstd, dz, coutt, RELU, da=dcat[k + 1][..., :coutt],
                               mask=mk[k + 1] if k < 2 else None, ws=self.ws)
            dWu[k] = _wgrad(dz, self.ins[k], coutt, coutt, cint, (cint, coutt, 4, 4))
            ops.conv_fwd(dz, self.gu_wf[k], dcat[k], coutt, cint)
        d_act = [self.cat3[..., 64:], self.cat2[..., 128:], self.cat1[..., 256:]]
        dskip = [dcat[3][..., 64:], dcat[2][..., 128:], dcat[1][..., 256:]]
        dd = dcat[0]
        for k in (3, 2, 1, 0):
            cin, cout = G_DOWN[k]
            cp = _pad8(cin)
            dz = act(S >> (k + 1), cout, T)
            if k == 3:
                ops.in_act_bwd(self.zd[3], *self.dstat[3], dz, 512, LRELU, da=dd, mask=mk[0], ws=self.ws)
            elif k > 0:
                ops.in_act_bwd(self.zd[k], *self.dstat[k], dz, cout, LRELU, da=dd, da2=dskip[k], ws=self.ws)
            else:
                ops.act_bwd(dd, d_act[0], dz, 64, da2=dskip[0])
            xin = self.x8 if k == 0 else d_act[k - 1]
            dWd[k] = _wgrad(xin, dz, cp, cin, cout, (cout, cin, 4, 4))
            if k > 0:
                dd = act(S >> k, cin)
                ops.conv_dgrad(dz, self.gd_wt[k], dd, cin, cout)
        return dWd, dWu, dfc_w, dfc_b


class GNetFn(Function):
    @staticmethod
    def forward(ctx, x, code, scale, masks, *params):
        tape = GTape(code)
        delta = tape.forward(x, list(params[0:4]), list(params[4:8]), params[8], params[9], scale, masks)
        ctx.tape = tape
        return delta

    @staticmethod
    @once_differentiable
    def backward(ctx, g_delta):
        dWd, dWu, dfc_w, dfc_b = ctx.tape.backward(g_delta)
        return (None, None, None, None, *dWd, *dWu, dfc_w, dfc_b)


def generator_forward(module, x: torch.Tensor, masks=None) -> torch.Tensor:
    """GeneratorUNet.forward (cgan/models.py:125-141).  In train mode the three Dropout(0.5) keep-masks are drawn on
    the device unless given (NCHW, parity runs)."""
    x = _check_input(x, "x")
    code = _lib.dtype_code(getattr(module, "compute_dtype", "fp32"))
    if module.training and masks is None:
        B, S = x.shape[0], x.shape[-1]
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        masks = []
        for j, (c, s) in enumerate(((512, S // 16), (256, S // 8), (128, S // 4))):
            m = torch.empty(B, s, s, c, device=x.device, dtype=torch.uint8)
            ops.dropout_mask_gen(m, seed + j)
            masks.append(m.permute(0, 3, 1, 2))
    if not module.training:
        masks = None
    params = [getattr(module, f"down{k}").model[0].weight for k in (1, 2, 3, 4)]
    params += [getattr(module, f"up{k}").model[0].weight for k in (1, 2, 3)] + [module.up4[0].weight]
    fc = module.fc_delta[1]
    params += [fc.weight, fc.bias]
    return GNetFn.apply(x, code, module.delta_scale, masks, *params)


# ------------------------------------------------------------------------------------------------------------------
# GeneratorSimpleRegressor (cgan/models.py:147-216): the module-level path runs the same kernel schedule as the step
# engine (gen_simple.SimpleGenerator) on a small host object that stands in for the engine
# ------------------------------------------------------------------------------------------------------------------
class _SimpleHost:
    """What gen_simple.SimpleGenerator needs from its owner: sizes, the flat parameter / gradient buffers, the head's
    output buffers and the InstanceNorm-backward scratch."""

    def __init__(self, params: dict, B: int, S: int, code: int, scale: float, dev):
        from .engine import FlatParams
        self.B, self.S, self.code, self.T, self.dev = B, S, code, _lib.torch_dtype(code), dev
        self.mma = code                                              # (conv dtype code == storage code: the module API has no split modes)
        self.delta_scale, self.seed = scale, 0
        self.G = FlatParams(params, list(params.keys()), dev)
        self.ws = self.ws_g = torch.empty(2 * B * 512, device=dev, dtype=F32)    # (the step engine keeps a separate one for the generator)
        self.g_traw = torch.empty(B, 4, device=dev, dtype=F32)
        self.g_delta = torch.empty(B, 4, device=dev, dtype=F32)
        self.x8 = torch.empty(B, S, S, 8, device=dev, dtype=self.T)

    def _conv(self, _label, _flops, fn, *args, **kw):
        return fn(*args, **kw)


class GSimpleFn(Function):
    @staticmethod
    def forward(ctx, x, module, code, masks, train, *params):
        from .gen_simple import GS_PARAM_KEYS, SimpleGenerator
        B, S = x.shape[0], x.shape[-1]
        key = (B, S, code, x.device)
        cache = module.__dict__.setdefault("_hip_hosts", {})
        if key not in cache:                                   # buffers are kept per (batch, size, dtype): inference loops
            host = _SimpleHost({k: p.detach() for k, p in zip(GS_PARAM_KEYS, params)}, B, S, code, module.delta_scale, x.device)
            cache[key] = (host, SimpleGenerator(host))
        host, gen = cache[key]
        for k, p in zip(GS_PARAM_KEYS, params):
            host.G.views[k].copy_(p.detach())
        gen.prep()
        ops.pack_pair(x, None, host.x8)
        if train:
            if masks is None:
                host.seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
            gen.set_masks(masks, 0)
        ctx.gen, ctx.host = gen, host
        return gen.forward(host.x8, train).clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, g_delta):
        from .gen_simple import GS_PARAM_KEYS
        host = ctx.host
        host.G.g.zero_()
        ctx.gen.backward(g_delta.contiguous().float())
        return (None, None, None, None, None, *[host.G.gviews[k].clone() for k in GS_PARAM_KEYS])


def simple_generator_forward(module, x: torch.Tensor, masks=None) -> torch.Tensor:
    """GeneratorSimpleRegressor.forward (cgan/models.py:213-216).  In train mode the two Dropout(0.5) keep-masks
    ([B,256], [B,64]) are drawn on the device unless given (parity runs).  One live forward per (batch, size) at a time:
    the activations a backward needs stay in the module's cached buffers until the next forward."""
    from .gen_simple import GS_PARAM_KEYS
    x = _check_input(x, "x")
    code = _lib.dtype_code(getattr(module, "compute_dtype", "fp32"))
    sd = dict(module.named_parameters())
    return GSimpleFn.apply(x, module, code, masks if module.training else None, bool(module.training),
                           *[sd[k] for k in GS_PARAM_KEYS])
