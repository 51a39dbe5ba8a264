"""Thin launch wrappers over the C ABI (include/gcssl.h).  They only marshal arguments: every tensor here is a
caller-owned device buffer, activations are NHWC views `[N][H][W][C]` (possibly channel slices of a wider
buffer: the pixel stride is taken from ``t.stride(2)``), and nothing is allocated or synchronised.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import call

LRELU, RELU = 1, 2


def _ld(t: torch.Tensor) -> int:
    """pixel stride (elements) of an NHWC view; requires the view to be dense in (n,h,w) and unit stride in c."""
    assert t.dim() == 4 and t.stride(3) == 1, "expected an NHWC view with contiguous channels"
    ld = t.stride(2)
    assert t.stride(1) == ld * t.shape[2] and t.stride(0) == ld * t.shape[2] * t.shape[1], "non-dense NHWC view"
    return ld


def code(t: torch.Tensor) -> int:
    return _lib.dtype_code(t.dtype)


# ---- boundary
def pack_pair(a, b, out, reps=1):
    """reps > 1: out holds reps copies of the packed batch back to back ([reps * B][S][S][8])"""
    B, _, S, _ = a.shape
    call("gcssl_pack_pair", code(out), a, b, out, B, S, int(reps))


def pack_interp(pred, gt, refined, alpha, out):
    B, _, S, _ = pred.shape
    call("gcssl_pack_interp", code(out), pred, gt, refined, alpha, out, B, S)


def unpack_grad(g, ga, gb):
    B, S = g.shape[0], g.shape[1]
    call("gcssl_unpack_grad", g, ga, gb, B, S)


# ---- weights
def prep_conv_weight(w, wf, wt, cout, cin, cinp, dt):
    call("gcssl_prep_conv_weight", dt, w, wf, wt, cout, cin, cinp)


class PrepBatch:
    """Argument block for gcssl_prep_conv_weights: layers = [(w, wf, wt, cout, cin, cinp), ...] (<= 8)."""

    def __init__(self, layers, dt, c5=None):
        """c5 = (w5 [1][C][4][4], w5p [16][C]): the critic head's fp32 re-pack rides along (prep_c5_weight)"""
        self.n, self.dt = len(layers), dt
        self._keep = (layers, c5)
        self._w = _lib.ptr_array([l[0] for l in layers])
        self._wf = _lib.ptr_array([l[1] for l in layers])
        self._wt = _lib.ptr_array([l[2] for l in layers])
        self._co, self._ci, self._cp = (_lib.int_array([l[k] for l in layers]) for k in (3, 4, 5))
        self._c5 = (c5[0], c5[1], c5[1].shape[1]) if c5 is not None else (None, None, 0)

    def run(self):
        call("gcssl_prep_conv_weights", self.dt, self.n, self._w, self._wf, self._wt, self._co, self._ci, self._cp, *self._c5)


def prep_c5_weight(w, wp):
    call("gcssl_prep_c5_weight", w, wp, wp.shape[1])


# ---- conv k4 s2 p1
def conv_fwd(x, wf, y, cin, cout, bias=None, gscale=None, group_n=0, act=0, split_stride=0, dt=None):
    """y may be fp32 while x is bf16 (pre-InstanceNorm tensors are kept in fp32).  split_stride > 0: a K-split launch
    stores its partial sums in fp32 slabs y + k*split_stride instead of adding atomically (see conv_splits).
    dt: conv dtype code when it is not the tensors' own -- _lib.F32_F16X3 / F32_BF16X3, the split-precision MFMA modes on
    fp32 tensors (the same keyword on conv_dgrad / conv_wgrad / conv3_fwd / conv3_wgrad)."""
    N, Hi, Wi, _ = x.shape
    out_f32 = 1 if (y.dtype == torch.float32 and x.dtype != torch.float32) else 0
    call("gcssl_conv4x4s2_fwd", code(x) if dt is None else dt, x, _ld(x), wf, bias, gscale, group_n, y, _ld(y), N, Hi, Wi, cin, cout, act,
         out_f32, int(split_stride))


def conv_splits(kind, dt, N, Hi, cin, cout, act=0, out_f32=1) -> int:
    """K split the dispatcher uses for a forward ('fwd') or dgrad-form ('dgrad') conv of these shapes."""
    if kind == "fwd":
        r = _lib.call_nostream("gcssl_conv4x4s2_fwd_splits", dt, N, Hi, Hi, cin, cout, act, out_f32)
    else:
        r = _lib.call_nostream("gcssl_conv4x4s2_dgrad_splits", dt, N, Hi, Hi, cin, cout, out_f32)
    if r <= 0:
        raise RuntimeError(f"conv_splits{(kind, N, Hi, cin, cout)} -> {r}")
    return r


def conv_in_act_ok(dt, N, Hi, cin, cout) -> bool:
    """Does the one-launch conv + InstanceNorm + LeakyReLU form serve these shapes (gcssl_conv4x4s2_in_act_ok)?"""
    r = _lib.call_nostream("gcssl_conv4x4s2_in_act_ok", dt, N, Hi, Hi, cin, cout)
    if r < 0:
        raise RuntimeError(f"conv_in_act_ok{(N, Hi, cin, cout)} -> {r}")
    return bool(r)


def conv_in_act_fwd(x, wf, a, mean, rstd, cin, cout, bias=None, gscale=None, group_n=0, mask=None, apre=None, apre_n0=0):
    """Conv2d(k4,s2,p1) + InstanceNorm + LeakyReLU(0.2) [+ dropout mask] in one launch: a (16-bit, may be a channel slice)
    and the fp32 statistics; apre: the un-masked activation of samples >= apre_n0 (for the backward of a masked layer)."""
    N, Hi, Wi, _ = x.shape
    call("gcssl_conv4x4s2_in_act_fwd", code(x), x, _ld(x), wf, bias, gscale, group_n, a, _ld(a), mean, rstd, mask, apre,
         _ld(apre) if apre is not None else 0, int(apre_n0), N, Hi, Wi, cin, cout, LRELU)


def conv_in_act_x3_ok(dt, N, Hi, cin, cout) -> bool:
    """split-precision modes: does the one-launch conv + InstanceNorm + LeakyReLU form on fp32 tensors serve these shapes?"""
    r = _lib.call_nostream("gcssl_conv4x4s2_in_act_x3_ok", dt, N, Hi, Hi, cin, cout)
    if r < 0:
        raise RuntimeError(f"conv_in_act_x3_ok{(N, Hi, cin, cout)} -> {r}")
    return bool(r)


def conv_in_act_x3_fwd(x, wf, z, a, mean, rstd, cin, cout, bias=None, gscale=None, group_n=0, dt=None):
    """Conv2d(k4,s2,p1) + InstanceNorm + LeakyReLU(0.2) in one launch on fp32 tensors (split-precision MFMA): a, the statistics and
    (z not None) the pre-norm values."""
    N, Hi, Wi, _ = x.shape
    call("gcssl_conv4x4s2_in_act_x3_fwd", dt, x, _ld(x), wf, bias, gscale, group_n, z, _ld(z) if z is not None else 0, a, _ld(a),
         mean, rstd, N, Hi, Wi, cin, cout)


def conv_dgrad(dy, wt, dx, cin, cout, gscale=None, group_n=0, split_stride=0, dt=None):
    """dx: [N][Hi][Wi][>=cin] (fp32 output allowed whatever dy's dtype), dy: [N][Hi/2][Wi/2][>=cout]."""
    N, Hi, Wi, _ = dx.shape
    out_f32 = 1 if (dx.dtype == torch.float32 and dy.dtype != torch.float32) else 0
    call("gcssl_conv4x4s2_dgrad", code(dy) if dt is None else dt, dy, _ld(dy), wt, gscale, group_n, dx, _ld(dx), N, Hi, Wi, cin, cout, out_f32,
         int(split_stride))


def conv_dgrad_act_bwd_ok(dt, N, Hi, cin, cout, with_sums) -> bool:
    r = _lib.call_nostream("gcssl_conv4x4s2_dgrad_act_bwd_ok", dt, N, Hi, Hi, cin, cout, int(bool(with_sums)))
    if r < 0:
        raise RuntimeError(f"conv_dgrad_act_bwd_ok{(N, Hi, cin, cout)} -> {r}")
    return bool(r)


def conv_dgrad_act_bwd(dy, wt, a, dzs, cin, cout, gscale=None, group_n=0, bias=None, dbias=None, cdot=None, nrep=1, rep_stride=0,
                       sat=None, dt=None):
    """conv_dgrad (fp32 dx) + act_bwd of the norm-less layer in front of the conv as ONE launch: dzs = lrelu'(a) dx gscale in the
    compute dtype; dzs / a: [N][Hi][Wi][>=cin], dy: [N][Hi/2][Wi/2][>=cout].  dt: a split-precision code (fp32 tensors)."""
    N, Hi, Wi, _ = dzs.shape
    call("gcssl_conv4x4s2_dgrad_act_bwd", code(dy) if dt is None else dt, dy, _ld(dy), wt, a, _ld(a), gscale, group_n, bias, dzs, _ld(dzs), dbias, cdot,
         nrep, rep_stride, sat, N, Hi, Wi, cin, cout)


def conv_fwd_act_bwd_ok(dt, N, Hi, cin, cout) -> bool:
    r = _lib.call_nostream("gcssl_conv4x4s2_fwd_act_bwd_ok", dt, N, Hi, Hi, cin, cout)
    if r < 0:
        raise RuntimeError(f"conv_fwd_act_bwd_ok{(N, Hi, cin, cout)} -> {r}")
    return bool(r)


def conv_fwd_act_bwd(x, wf, a, y, cin, cout, gscale=None, group_n=0, dotx=None, dot_out=None, sat=None, dt=None):
    """conv_fwd (fp32 v, first layer 8 -> 64) + act_bwd + the <dotx, v> sum as ONE launch: y = lrelu'(a) v in the compute
    dtype; x: [N][Hi][Wi][>=cin], a / y / dotx: [N][Hi/2][Wi/2][>=cout].  dt: a split-precision code (fp32 tensors)."""
    N, Hi, Wi, _ = x.shape
    call("gcssl_conv4x4s2_fwd_act_bwd", code(x) if dt is None else dt, x, _ld(x), wf, gscale, group_n, a, _ld(a), y, _ld(y), dotx,
         _ld(dotx) if dotx is not None else 0, dot_out, sat, N, Hi, Wi, cin, cout)


def convt_in_relu_fwd(x, wt, mean, rstd, K, z32=None, z_n0=0, a=None, pool=None, cnt=None):
    """ConvTranspose2d(K -> 64, k4 s2 p1) + InstanceNorm + ReLU in one launch (csrc/convt_fused.hip).  x: [N][H][H][>=K],
    H in (8, 16), N*H*H a multiple of 256; wt: the dgrad pack [64][16][K].  Optional outputs: a (16-bit activation, may be a
    channel slice of a concat buffer), z32 (fp32 pre-norm values, written for samples >= z_n0 only), pool ([N][64] sums),
    cnt ([N][64] counts of positive outputs; with pool it lets in_act_bwd skip its statistics pass)."""
    N, H, _, _ = x.shape
    call("gcssl_convT4x4s2_in_relu_fwd", code(x), x, _ld(x), wt, z32, _ld(z32) if z32 is not None else 0, int(z_n0), a,
         _ld(a) if a is not None else 0, mean, rstd, pool, cnt, N, H, K, 64)


def convt_fused_ok(dt: int, n: int, h: int, cout_t: int) -> bool:
    """shapes the fused kernel takes: 16-bit dtype, 64 output channels, 8x8 or 16x16 inputs in whole 256-pixel blocks"""
    return dt != _lib.F32 and cout_t == 64 and h in (8, 16) and (n * h * h) % 256 == 0


def wgrad_splits(N, Hi, Wi, cin, cout) -> int:
    r = _lib.call_nostream("gcssl_conv4x4s2_wgrad_splits", N, Hi, Wi, cin, cout)
    if r <= 0:
        raise RuntimeError(f"gcssl_conv4x4s2_wgrad_splits: bad geometry {(N, Hi, Wi, cin, cout)}")
    return r


def conv_wgrad(x, dy, slab, cin, cout, dt=None):
    N, Hi, Wi, _ = x.shape
    call("gcssl_conv4x4s2_wgrad", code(x) if dt is None else dt, x, _ld(x), dy, _ld(dy), slab, N, Hi, Wi, cin, cout)


class WgradBatch:
    """Argument block for gcssl_conv4x4s2_wgrad_batch: layers = [(x, dy, slab, cin, cout), ...] (<= 3; x: [N][Hi][Wi][>=cin],
    dy: [N][Hi/2][Wi/2][>=cout]).  One launch when every layer takes the filter-row LDS-DMA kernel, else one per layer."""

    def __init__(self, layers, dt=None):
        self._keep = layers
        self.n = len(layers)
        self.dt = code(layers[0][0]) if dt is None else dt
        self._x = _lib.ptr_array([l[0] for l in layers]); self._dy = _lib.ptr_array([l[1] for l in layers])
        self._slab = _lib.ptr_array([l[2] for l in layers])
        self._ldx = _lib.int_array([_ld(l[0]) for l in layers]); self._lddy = _lib.int_array([_ld(l[1]) for l in layers])
        self._N = _lib.int_array([l[0].shape[0] for l in layers]); self._H = _lib.int_array([l[0].shape[1] for l in layers])
        self._W = _lib.int_array([l[0].shape[2] for l in layers])
        self._ci = _lib.int_array([l[3] for l in layers]); self._co = _lib.int_array([l[4] for l in layers])

    def run(self):
        call("gcssl_conv4x4s2_wgrad_batch", self.dt, self.n, self._x, self._ldx, self._dy, self._lddy, self._slab, self._N, self._H,
             self._W, self._ci, self._co)


def conv_wgrad_batch(batch: "WgradBatch", dt=None):
    """(the engine's launch wrapper passes dt for the split-precision modes: then the layers go out one by one)"""
    if dt is not None:
        batch.dt = dt
    batch.run()


def wgrad_reduce(slab, nsplit, dw, cout, cin, cin_real, coef=None, cscale=None, u=None, v=None, nrank=0,
                 accumulate=False):
    """u, v: 2-D [nrank][>=cout] / [nrank][>=cin_real*16] (row strides are taken from the tensors)."""
    call("gcssl_wgrad_reduce", slab, nsplit, dw, cout, cin, cin_real, coef, cscale, u,
         u.stride(0) if u is not None else 0, v, v.stride(0) if v is not None else 0, nrank,
         2 if accumulate == "zeroed" else int(accumulate))


class ReplicaSum:
    """Argument block for gcssl_sum_replicas: segments = [(src_replica0, dst, length, accumulate), ...] (<= 8)."""

    def __init__(self, segments, nrep, rep_stride):
        self._keep = segments
        self.n, self.nrep, self.rep_stride = len(segments), nrep, rep_stride
        self.acc = sum(1 << i for i, s in enumerate(segments) if s[3])
        self._src = _lib.ptr_array([s[0] for s in segments])
        self._dst = _lib.ptr_array([s[1] for s in segments])
        self._len = _lib.int_array([s[2] for s in segments])

    def run(self):
        call("gcssl_sum_replicas", self.n, self._src, self._dst, self._len, self.nrep, self.rep_stride, self.acc)


class ReduceBatch:
    """Argument block for gcssl_wgrad_reduce_batch: layers = [dict(slab, nsplit, dw, cout, cin, cin_real[, coef, u, v])]
    (<= 8); u, v are 2-D history tensors that share row strides."""

    def __init__(self, layers, nrank=0, accumulate="zeroed", nrep=0, rep_stride=0):
        """optional per layer: coef_rep (replica 0 of the striped coefficient sums, added to coef), bias_rep + dbias (the
        striped bias-gradient sums and where their total goes): the fold of ReplicaSum, nrep replicas rep_stride floats apart"""
        self.n, self.nrank = len(layers), nrank
        self.nrep, self.rep_stride = nrep, rep_stride
        opt = lambda key: (ctypes.c_void_p * len(layers))(*[l[key].data_ptr() if l.get(key) is not None else None for l in layers]) \
            if any(l.get(key) is not None for l in layers) else None
        self._crep, self._brep, self._dbias = opt("coef_rep"), opt("bias_rep"), opt("dbias")
        self._keep = layers
        self._slab = _lib.ptr_array([l["slab"] for l in layers])
        self._dw = _lib.ptr_array([l["dw"] for l in layers])
        self._ns, self._co, self._ci, self._cr = (_lib.int_array([l[k] for l in layers])
                                                  for k in ("nsplit", "cout", "cin", "cin_real"))
        if nrank:
            self._coef, self._u, self._v = (_lib.ptr_array([l[k] for l in layers]) for k in ("coef", "u", "v"))
            self.su, self.sv = layers[0]["u"].stride(0), layers[0]["v"].stride(0)
            assert all(l["u"].stride(0) == self.su and l["v"].stride(0) == self.sv for l in layers)
        else:
            self._coef = self._u = self._v = None
            self.su = self.sv = 0
        self.acc = 2 if accumulate == "zeroed" else int(accumulate)

    def run(self):
        call("gcssl_wgrad_reduce_batch", self.n, self._slab, self._ns, self._dw, self._co, self._ci, self._cr, self._coef,
             self._u, self._v, self.su, self.sv, self.nrank, self.acc, self._crep, self._brep, self._dbias, self.nrep,
             self.rep_stride)


# ---- critic head
def c5_fwd(x, wp, out, group_mean=None, groups=0):
    """group_mean (zeroed by the caller): += the means of `groups` equal chunks of out (== group_mean(out, groups, ...))"""
    N, Hi, Wi, _ = x.shape
    call("gcssl_conv4x4s1_c1_fwd", code(x), x, _ld(x), wp, out, group_mean, int(groups), N, Hi, Wi, wp.shape[1])


def _c4(consts):
    c = [float(v) for v in consts] + [0.0] * (4 - len(consts))
    return c[:4]


def c5_dgrad(dx, wp, dout=None, consts=(0.0, 0.0, 0.0), group_n=1):
    """consts: up to four per-sample-group constants standing in for dout"""
    N, Hi, Wi, _ = dx.shape
    call("gcssl_conv4x4s1_c1_dgrad", code(dx), dout, *_c4(consts), group_n, wp, dx, _ld(dx), N, Hi, Wi, wp.shape[1])


def c5_dgrad_defer(dx, w_raw, consts=(0.0, 0.0, 0.0), group_n=1):
    """c5_dgrad's constant form, recorded instead of launched: the NEXT PrepBatch.run() of this thread carries it (one launch
    less).  w_raw: the head conv's own weight [1][C][4][4] (fp32); dx fp32.  The caller re-packs right after and reads dx later."""
    N, Hi, Wi, _ = dx.shape
    assert dx.dtype == torch.float32
    r = _lib.call_nostream("gcssl_conv4x4s1_c1_dgrad_defer", *_c4(consts), group_n, w_raw, dx, _ld(dx), N, Hi, Wi, w_raw.shape[1])
    if r != 0:
        raise RuntimeError(f"gcssl_conv4x4s1_c1_dgrad_defer -> {_lib.ERRORS.get(r, r)}")


def c5_wgrad(x, dw, C, dout=None, consts=(0.0, 0.0, 0.0), group_n=1):
    """dw: fp32 [C][16] (PyTorch layout of the [1][C][4][4] weight), accumulated atomically."""
    N, Hi, Wi, _ = x.shape
    call("gcssl_conv4x4s1_c1_wgrad", code(x), x, _ld(x), dout, *_c4(consts), group_n, dw, N, Hi, Wi, C)


# ---- norm / activation
def in_act_fwd(z, a, mean, rstd, C, act, mask=None, pool=None, nslab=1, slab_stride=0):
    """z: fp32 pre-norm tensor; a: activation output in the compute dtype.  nslab > 1: z is the first of nslab split-K
    slabs slab_stride floats apart; they are summed on load and the total is written back to z.
    a None (see in_act_fwd_pool_only_ok): only the statistics and the pool sums."""
    N, H, W, _ = z.shape
    assert z.dtype == torch.float32
    call("gcssl_in_act_fwd", code(a) if a is not None else _lib.F32, z, _ld(z), a, _ld(a) if a is not None else 0, mean, rstd, mask,
         pool, nslab, int(slab_stride), N, H * W, C, act)


def in_act_fwd_pool_only_ok(HW: int, C: int) -> bool:
    """may in_act_fwd be called with a=None (statistics + pool sums, no activation store)?  gcssl_in_act_fwd's LDS-resident form."""
    return 256 < HW <= 1024 and C % 32 == 0


def in_act_bwd(z, mean, rstd, dzs, C, act, da=None, da2=None, da_bcast=None, mask=None, zt=None, zt_n0=0,
               gscale=None, group_n=0, bias=None, dbias=None, cdot=None, ws=None, nrep=1, rep_stride=0, da_nslab=1,
               da_slab_stride=0, presum_cnt=None, presum_pos=None, presum_pos_scale=1.0, sat=None):
    """da_nslab > 1: da is the first of that many split-K slabs (da_slab_stride floats apart), added on load.  nrep > 1: dbias / cdot point at replica 0 of nrep replicas rep_stride floats apart (fold with ReplicaSum).
    z: the fp32 pre-norm tensor, or the 16-bit un-masked activation a fused conv_in_act_fwd left (z_kind 1)."""
    N, H, W, _ = z.shape
    assert all(t is None or t.dtype == torch.float32 for t in (da, da2, zt))
    z_kind = 0 if z.dtype == torch.float32 else 1
    assert z_kind == 0 or z.dtype == dzs.dtype
    call("gcssl_in_act_bwd", code(dzs), da, _ld(da) if da is not None else 0, da2, _ld(da2) if da2 is not None else 0,
         da_bcast, z, _ld(z), z_kind, mean, rstd, mask, zt, zt_n0, gscale, group_n, bias, dzs, _ld(dzs), dbias, cdot, nrep, rep_stride,
         da_nslab, int(da_slab_stride), ws, presum_cnt, presum_pos, float(presum_pos_scale), sat, N, H * W, C, act)


def in_dbl_bwd(gb_a, qz, gb_zs, z, mean, rstd, gt_a, zt, C, act, cdot=None, q_nslab=1, q_slab_stride=0, sat=None):
    N, H, W, _ = z.shape
    assert gb_a.dtype == torch.float32 and qz.dtype == torch.float32
    assert zt.dtype == torch.float32
    z_kind = 0 if z.dtype == torch.float32 else 1                  # 1: the 16-bit activation of a fused conv_in_act_fwd
    assert z_kind == 0 or z.dtype == gt_a.dtype
    call("gcssl_in_dbl_bwd", code(gt_a), gb_a, _ld(gb_a), qz, _ld(qz), gb_zs, _ld(gb_zs) if gb_zs is not None else 0,
         z, _ld(z), z_kind, mean, rstd, gt_a, _ld(gt_a), zt, cdot, q_nslab, int(q_slab_stride), sat, N, H * W, C, act)


def act_bwd(da, a, dzs, C, da2=None, gscale=None, group_n=0, bias=None, dbias=None, cdot=None, nrep=1, rep_stride=0, sat=None,
            dotx=None, dot_out=None):
    """da/da2: fp32 incoming gradients; a and dzs in the compute dtype.  dotx (compute dtype) + dot_out: dot_out += sum dotx*da."""
    N, H, W, _ = a.shape
    assert da.dtype == torch.float32 and (da2 is None or da2.dtype == torch.float32)
    call("gcssl_act_bwd", code(a), da, _ld(da), da2, _ld(da2) if da2 is not None else 0, a, _ld(a), gscale, group_n,
         bias, dzs, _ld(dzs), dbias, cdot, nrep, rep_stride, sat, dotx, _ld(dotx) if dotx is not None else 0, dot_out, N, H * W, C)


def dot_accum(x, y, C, out):
    """out += sum x*y; x in the compute dtype, y fp32."""
    N, H, W, _ = x.shape
    assert y.dtype == torch.float32
    call("gcssl_dot_accum", code(x), x, _ld(x), y, _ld(y), N * H * W, C, out)


# ---- spectral norm
class SnState:
    """Argument block for gcssl_sn_power_iter over the critic's spectrally-normalised layers."""

    def __init__(self, ws, us, vs, nslots, device):
        self.n = len(ws)
        self.rows = [w.shape[0] for w in ws]
        self.cols = [w[0].numel() for w in ws]
        self.ws, self.us, self.vs = ws, us, vs
        tbuf = torch.zeros(2 * sum(self.cols), device=device)      # two halves per layer (chain parity): zero on entry, left zero
        self.t, off = [], 0
        for c in self.cols:
            self.t.append(tbuf[off:off + 2 * c]); off += 2 * c
        self._tbuf = tbuf
        self.s = [torch.empty(r, device=device) for r in self.rows]
        self.nslots = nslots
        self.su, self.sv = max(self.rows), max(self.cols)
        self.sigma = torch.zeros(self.n, nslots, device=device)
        self.isig = torch.zeros(self.n, nslots, device=device)
        self.u_hist = torch.zeros(self.n, nslots, self.su, device=device)
        self.v_hist = torch.zeros(self.n, nslots, self.sv, device=device)
        self._rows, self._cols = _lib.int_array(self.rows), _lib.int_array(self.cols)
        self.refresh_ptrs()

    def refresh_ptrs(self):
        self._w, self._u, self._v = _lib.ptr_array(self.ws), _lib.ptr_array(self.us), _lib.ptr_array(self.vs)
        self._t, self._s = _lib.ptr_array(self.t), _lib.ptr_array(self.s)

    def iterate(self, slot: int, iterate=True, zero=None, defer_finish=False):
        """iterate: True / k = that many chained power iterations filling slots slot.., False / 0 = sigma only.
        zero: a float tensor the closing launch also clears.  defer_finish: leave the chain's closing step to the NEXT
        PrepBatch.run() (gcssl_sn_defer_finish: one launch less; the caller re-packs right after)."""
        if defer_finish and iterate:
            _lib.lib().gcssl_sn_defer_finish(1)
        call("gcssl_sn_power_iter", self.n, self._w, self._u, self._v, self._t, self._s, self._rows, self._cols,
             self.sigma, self.isig, self.u_hist, self.v_hist, self.su, self.sv, slot, self.nslots, int(iterate),
             zero, zero.numel() if zero is not None else 0)


# ---- GP, optimiser, heads
def pack_fake_interp(pred, gt, refined, alpha, out_fake, out_interp, seed=0, counter=None, out_real=None):
    """alpha None: drawn on the device from (seed, counter[0], sample).  out_real: also pack the real group (pred, gt) -- pack_pair
    in the same launch."""
    B, _, S, _ = pred.shape
    if out_real is None:
        call("gcssl_pack_fake_interp", code(out_fake), pred, gt, refined, alpha, int(seed), counter, out_fake, out_interp, B, S)
    else:
        call("gcssl_pack_groups", code(out_fake), pred, gt, refined, alpha, int(seed), counter, out_real, out_fake, out_interp, B, S)


def gp_norm(g, B, lambda_gp, nrm, coef, gp_sum, scaled=None, sat=None):
    """scaled (optional, compute dtype): receives g * coef[n] in the same launch (== scale_rows afterwards).
    sat (here and on the norm / activation backward wrappers): int32 device counter of fp16 stores that clipped."""
    call("gcssl_gp_norm", g, g.numel() // B, B, float(lambda_gp), nrm, coef, gp_sum,
         code(scaled) if scaled is not None else 0, scaled, sat)


def last_grid() -> int:
    """workgroups of the most recent GCSSL_LAUNCH of this process (gcssl_last_grid)"""
    return int(_lib.lib().gcssl_last_grid())


def last_kernel() -> str:
    """the kernel template expression the most recent conv entry point launched (gcssl_last_kernel)"""
    return _lib.lib().gcssl_last_kernel().decode()


def scale_rows(x, coef, y, B):
    call("gcssl_scale_rows", code(y), x, coef, y, x.numel() // B, B)


ADAM_STATE = 264        # doubles in a clip_adam state block (GCSSL_ADAM_STATE, include/gcssl.h)


def clip_adam(p, g, m, v, state, lr, b1, b2, eps=1e-8, max_norm=1.0, write_clipped=False, grad_scale=1.0):
    """write_clipped: False/0 leave g, True/1 store the clipped gradient, 2 zero g (fused zero_grad).  state: ADAM_STATE doubles.
    grad_scale: the optimiser sees g * grad_scale (1/world after a data-parallel SUM all-reduce)."""
    assert state.numel() >= ADAM_STATE and state.dtype == torch.float64
    call("gcssl_clip_adam", p, g, m, v, p.numel(), state, float(lr), float(b1), float(b2), float(eps),
         float(max_norm), int(write_clipped), float(grad_scale))


def pool_fc_tanh_fwd(x, w, bias, scale, pooled, traw, delta, pool_sum=None):
    N, H, W, _ = x.shape
    call("gcssl_pool_fc_tanh_fwd", code(x), x, _ld(x), pool_sum, w, bias, float(scale), pooled, traw, delta, N, H * W, 64)


def head_bwd(g_delta, traw, pooled, w, scale, B, HW, dw, db, da_bcast):
    call("gcssl_head_bwd", g_delta, traw, pooled, w, float(scale), B, HW, dw, db, da_bcast)


def eiou_fwd_bwd(pred_box, delta, delta_true, lambda_iou, g_delta, calibrated, loss_acc):
    call("gcssl_eiou_fwd_bwd", pred_box, delta, delta_true, pred_box.shape[0], float(lambda_iou), g_delta,
         calibrated, loss_acc)


def dropout_mask_gen(out, seed, counter=None):
    call("gcssl_dropout_mask_gen", out, out.numel(), int(seed), counter)


def uniform_gen(out, seed, counter=None):
    call("gcssl_uniform_gen", out, out.numel(), int(seed), counter)


def group_mean(x, groups, out):
    call("gcssl_group_mean", x, groups, x.numel() // groups, out)


# ---- GeneratorSimpleRegressor pieces (cgan/models.py:147-216): 3x3 convs, MaxPool2d, the regressor head
def conv3_wk(c: int) -> int:
    """elements per packed 3x3 weight row for c input channels (9*c rounded up to 64)."""
    return _lib.call_nostream("gcssl_conv3x3_wk", c)


class Prep3Batch:
    """Argument block for gcssl_conv3x3_prep_weights: layers = [(w, wf|None, wt|None, cout, cin, cinp), ...] (<= 8)."""

    def __init__(self, layers, dt):
        self.n, self.dt = len(layers), dt
        self._keep = layers
        null = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() if t is not None else None for t in ts])
        self._w, self._wf, self._wt = (null([l[k] for l in layers]) for k in (0, 1, 2))
        self._co, self._ci, self._cp = (_lib.int_array([l[k] for l in layers]) for k in (3, 4, 5))

    def run(self):
        call("gcssl_conv3x3_prep_weights", self.dt, self.n, self._w, self._wf, self._wt, self._co, self._ci, self._cp)


def conv3_fwd(x, w, y, cin, cout, bias=None, dt=None):
    """3x3 s1 p1 conv (or, with the rotated-transposed pack and swapped channel counts, its data gradient)."""
    N, H, W, _ = x.shape
    out_f32 = 1 if (y.dtype == torch.float32 and x.dtype != torch.float32) else 0
    call("gcssl_conv3x3_fwd", code(x) if dt is None else dt, x, _ld(x), w, bias, y, _ld(y), N, H, W, cin, cout, out_f32)


def conv3_wgrad_splits(N, H, cin, cout) -> int:
    r = _lib.call_nostream("gcssl_conv3x3_wgrad_splits", N, H, H, cin, cout)
    if r <= 0:
        raise RuntimeError(f"conv3_wgrad_splits{(N, H, cin, cout)} -> {r}")
    return r


def conv3_wgrad(x, dy, slab, cin, cout, dt=None):
    N, H, W, _ = x.shape
    call("gcssl_conv3x3_wgrad", code(x) if dt is None else dt, x, _ld(x), dy, _ld(dy), slab, N, H, W, cin, cout)


class Reduce3Batch:
    """Argument block for gcssl_conv3x3_wgrad_reduce: layers = [dict(slab, nsplit, dw, cout, cin, cin_real), ...] (<= 8)."""

    def __init__(self, layers):
        self.n = len(layers)
        self._keep = layers
        self._slab = _lib.ptr_array([l["slab"] for l in layers])
        self._dw = _lib.ptr_array([l["dw"] for l in layers])
        self._ns, self._co, self._ci, self._cr = (_lib.int_array([l[k] for l in layers])
                                                  for k in ("nsplit", "cout", "cin", "cin_real"))

    def run(self):
        call("gcssl_conv3x3_wgrad_reduce", self.n, self._slab, self._ns, self._dw, self._co, self._ci, self._cr)


def maxpool2_fwd(a, o, C):
    N, H, W, _ = a.shape
    call("gcssl_maxpool2_fwd", code(a), a, _ld(a), o, _ld(o), N, H, W, C)


def maxpool2_bwd(a, dpool, da, C, bcast_scale=None):
    """dpool: [N][H/2][W/2][C] fp32, or (bcast_scale given) [N][C] broadcast over the pooled pixels times bcast_scale."""
    N, H, W, _ = a.shape
    if bcast_scale is None:
        call("gcssl_maxpool2_bwd", code(a), a, _ld(a), dpool, _ld(dpool), 0, 1.0, da, _ld(da), N, H, W, C)
    else:
        call("gcssl_maxpool2_bwd", code(a), a, _ld(a), dpool, dpool.stride(0), 1, float(bcast_scale), da, _ld(da), N, H, W, C)


def avgpool_fwd(x, feat, C):
    N, H, W, _ = x.shape
    call("gcssl_avgpool_fwd", code(x), x, _ld(x), feat, N, H * W, C)


def mlp_head_fwd(feat, w1t, b1, w2t, b2, w3, b3, delta_scale, h1, h2, traw, delta, m1=None, m2=None):
    """w1t [512][256], w2t [256][64]: transposed nn.Linear weights (contiguous); w3 [4][64] plain."""
    assert w1t.shape == (512, 256) and w2t.shape == (256, 64) and w1t.is_contiguous() and w2t.is_contiguous()
    call("gcssl_mlp_head_fwd", feat, w1t, b1, w2t, b2, w3, b3, m1, m2, float(delta_scale), h1, h2, traw, delta, feat.shape[0])


def mlp_head_bwd(gdelta, traw, h1, h2, feat, w1, w2, w3, delta_scale, train, dp1, dp2, dp3, dfeat, dw1, db1, dw2, db2, dw3, db3):
    call("gcssl_mlp_head_bwd", gdelta, traw, h1, h2, feat, w1, w2, w3, float(delta_scale), int(bool(train)), dp1, dp2, dp3,
         dfeat, dw1, db1, dw2, db2, dw3, db3, feat.shape[0])


#: the wrappers that take the conv dtype keyword `dt` (StepEngine injects its split-precision code into these)
CONV_FNS = (conv_fwd, conv_dgrad, conv_wgrad, conv3_fwd, conv3_wgrad, conv_in_act_x3_fwd, conv_fwd_act_bwd, conv_dgrad_act_bwd,
            conv_wgrad_batch)
