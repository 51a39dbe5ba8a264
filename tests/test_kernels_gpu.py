"""Per-kernel parity of the HIP kernels (called through the C ABI) against torch CPU fp32/fp64 / the oracle formulas.

Tolerances (relative to the tensor's largest entry).  Operands are pre-rounded to the compute dtype on BOTH sides (q()), so
a 16-bit run differs from the reference only by (a) fp32 summation order and (b) the rounding of a 16-bit OUTPUT:
  * `tol`     -- outputs stored in the compute dtype: fp32 2e-5, bf16 5e-3 (half an ulp is 2e-3), fp16 6e-4 (5e-4);
  * `tol_acc` -- fp32 outputs of a 16-bit kernel (pre-norm tensors, slabs, gradients): 1e-4, i.e. summation order only.
    This is what pins the LDS-DMA kernels' taps, zero-filled borders, persistent tile order and vmcnt bookkeeping: one
    dropped boundary tap of a K = 1024..4096 contraction is ~1e-2."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch.nn.grad import conv2d_weight

from conftest import load_pkg, rel_err

pytestmark = pytest.mark.gpu

DTS = [(torch.float32, 2e-5), (torch.bfloat16, 5e-3), (torch.float16, 6e-4)]
TOL_ACC = 1e-4


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return load_pkg("ops")


def nhwc(x, dt, pad_to=None):
    """CPU NCHW fp32 -> device NHWC tensor (optionally zero-padded channels)."""
    x = x.permute(0, 2, 3, 1).contiguous()
    if pad_to and pad_to > x.shape[3]:
        x = F.pad(x, (0, pad_to - x.shape[3]))
    return x.to("cuda", dt).contiguous()


def nchw(x):
    return x.float().cpu().permute(0, 3, 1, 2).contiguous()


def q(x, dt):
    """round-trip through the compute dtype (so the CPU reference sees the same operand values)"""
    return x.to(dt).float()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def packed_weights(ops, w, dt, cinp=None):
    cout, cin = w.shape[:2]
    cinp = cinp or cin
    wf = torch.empty(cout, 16, cinp, device="cuda", dtype=dt)
    wt = torch.empty(cinp, 16, cout, device="cuda", dtype=dt)
    ops.prep_conv_weight(w.cuda(), wf, wt, cout, cin, cinp, ops.code(wf))
    return wf, wt


CONV_CASES = [  # N, Hi, Cin(real), CinP, Cout
    (3, 16, 64, 64, 128),     # 64x64 tile
    (2, 8, 256, 256, 512),
    (5, 32, 6, 8, 64),        # first layer, padded channels
    (8, 128, 64, 64, 128),    # large M -> 128x128 tile
    (24, 32, 128, 128, 64),   # 128x64 tile
    (3, 4, 256, 256, 512),    # 4x4 -> 2x2 (D.c4 at 32x32 input)
    (651, 16, 64, 64, 128),   # 652 fwd tiles / 2604 dgrad tiles of 128x64 (ragged last one): the persistent LDS-DMA kernel
    (256, 32, 64, 64, 128),   # G.up4 / D.c2 shapes at the bench batch: 2048 dgrad tiles
    (768, 32, 6, 8, 64),      # D.c1 at the bench batch: 1536 tiles of two K steps (8-channel persistent form)
    (3, 64, 6, 8, 64),        # first-layer forward kernel at 64x64 (2 tiles of 4 output rows per sample)
    (2, 128, 6, 8, 64),       # ... and 128x128 (Wo = 64: 2 output rows per tile)
    (768, 8, 128, 128, 256),  # D.c3 at the bench batch: forward = 192 tiles of 128x128 on the loader/consumer ring
    (768, 4, 256, 256, 512),  # D.c4: ring forward in two K halves (96 tiles), ring dgrad (192 tiles)
    (16, 64, 128, 128, 256),  # filter-row LDS-DMA wgrad (>= 16 K steps for each of 256 workgroups): 2 x 2 channel blocks
    (1027, 8, 128, 128, 256), # ... with a ragged last K tile (16432 output pixels) and a short last split
]


@pytest.mark.parametrize("dt,tol", DTS)
@pytest.mark.parametrize("N,Hi,Cin,CinP,Cout", CONV_CASES)
def test_conv_fwd(ops, dt, tol, N, Hi, Cin, CinP, Cout):
    x = q(rnd(N, Cin, Hi, Hi, seed=1), dt)
    w = rnd(Cout, Cin, 4, 4, seed=2, scale=0.05)
    b = rnd(Cout, seed=3, scale=0.1)
    gs = torch.tensor([1.3, 0.7, 2.1])[: (N + (N + 2) // 3 - 1) // ((N + 2) // 3)]
    group_n = (N + 2) // 3
    wf, _ = packed_weights(ops, w, dt, CinP)
    xd = nhwc(x, dt, CinP)
    y = torch.zeros(N, Hi // 2, Hi // 2, Cout, device="cuda", dtype=dt)
    ops.conv_fwd(xd, wf, y, CinP, Cout, bias=b.cuda(), gscale=gs.cuda(), group_n=group_n, act=1)
    torch.cuda.synchronize()
    lin = F.conv2d(x.double(), q(w, dt).double(), None, 2, 1)
    scale = gs[torch.arange(N) // group_n].view(-1, 1, 1, 1).double()
    lin = lin * scale + b.view(1, -1, 1, 1).double()
    ref = F.leaky_relu(lin, 0.2)
    assert rel_err(nchw(y), ref) < tol
    if dt != torch.float32:
        # the same kernels with an fp32 output: summation order is all that is left.  With the activation (no K split) ...
        y32 = torch.full((N, Hi // 2, Hi // 2, Cout), float("nan"), device="cuda")
        ops.conv_fwd(xd, wf, y32, CinP, Cout, bias=b.cuda(), gscale=gs.cuda(), group_n=group_n, act=1)
        assert rel_err(nchw(y32), ref) < TOL_ACC
        # ... and the linear form the engine uses for pre-norm tensors: the dispatcher may split K (atomics into a zeroed
        # output, or slabs behind the output that the consumer adds)
        z32 = torch.full((N, Hi // 2, Hi // 2, Cout), float("nan"), device="cuda")
        ops.conv_fwd(xd, wf, z32, CinP, Cout, bias=b.cuda(), gscale=gs.cuda(), group_n=group_n)
        assert rel_err(nchw(z32), lin) < TOL_ACC
        ks = ops.conv_splits("fwd", ops.code(xd), N, Hi, CinP, Cout)
        if ks > 1:
            slabs = torch.full((ks, N, Hi // 2, Hi // 2, Cout), float("nan"), device="cuda")
            ops.conv_fwd(xd, wf, slabs[0], CinP, Cout, bias=b.cuda(), gscale=gs.cuda(), group_n=group_n,
                         split_stride=slabs[0].numel())
            assert rel_err(nchw(slabs.sum(0)), lin) < TOL_ACC


@pytest.mark.parametrize("dt,tol", DTS)
@pytest.mark.parametrize("N,Hi,Cin,CinP,Cout", CONV_CASES)
def test_conv_dgrad(ops, dt, tol, N, Hi, Cin, CinP, Cout):
    dy = q(rnd(N, Cout, Hi // 2, Hi // 2, seed=4), dt)
    w = rnd(Cout, Cin, 4, 4, seed=5, scale=0.05)
    _, wt = packed_weights(ops, w, dt, CinP)
    dyd = nhwc(dy, dt)
    gs = torch.tensor([0.9], device="cuda")
    dx = torch.zeros(N, Hi, Hi, CinP, device="cuda", dtype=dt)
    ops.conv_dgrad(dyd, wt, dx, CinP, Cout, gscale=gs, group_n=N)
    dx32 = torch.zeros(N, Hi, Hi, CinP, device="cuda", dtype=torch.float32)
    ops.conv_dgrad(dyd, wt, dx32, CinP, Cout)
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(dy.double(), q(w, dt).double(), None, 2, 1)
    assert rel_err(nchw(dx)[:, :Cin], ref * 0.9) < tol
    assert rel_err(nchw(dx32)[:, :Cin], ref) < (tol if dt == torch.float32 else TOL_ACC)
    ks = ops.conv_splits("dgrad", ops.code(dyd), N, Hi, CinP, Cout)
    if dt != torch.float32 and ks > 1:                        # slab form of a K-split launch
        slabs = torch.full((ks, N, Hi, Hi, CinP), float("nan"), device="cuda")
        ops.conv_dgrad(dyd, wt, slabs[0], CinP, Cout, split_stride=slabs[0].numel())
        assert rel_err(nchw(slabs.sum(0))[:, :Cin], ref) < TOL_ACC
    if CinP > Cin:
        assert float(dx32[..., Cin:].abs().max()) == 0.0


@pytest.mark.parametrize("dt,tol", DTS)
@pytest.mark.parametrize("N,Hi,Cin,CinP,Cout", CONV_CASES)
def test_conv_wgrad(ops, dt, tol, N, Hi, Cin, CinP, Cout):
    x = q(rnd(N, Cin, Hi, Hi, seed=6), dt)
    dy = q(rnd(N, Cout, Hi // 2, Hi // 2, seed=7), dt)
    xd, dyd = nhwc(x, dt, CinP), nhwc(dy, dt)
    ns = ops.wgrad_splits(N, Hi, Hi, CinP, Cout)
    slab = torch.full((ns, Cout, 16, CinP), float("nan"), device="cuda")
    ops.conv_wgrad(xd, dyd, slab, CinP, Cout)
    dw = torch.full((Cout, Cin, 4, 4), float("nan"), device="cuda")
    # rank-1 correction as used by the spectral-norm quotient rule
    coef = torch.tensor([0.5, -0.25], device="cuda")
    u = rnd(2, Cout, seed=8).cuda()
    v = rnd(2, Cin * 16, seed=9).cuda()
    ops.wgrad_reduce(slab, ns, dw, Cout, CinP, Cin, coef=coef, cscale=torch.tensor([2.0, 0.5], device="cuda"), u=u, v=v, nrank=2)
    torch.cuda.synchronize()
    dw2 = torch.zeros((Cout, Cin, 4, 4), device="cuda")      # pre-zeroed, split-parallel atomic reduction
    ops.wgrad_reduce(slab, ns, dw2, Cout, CinP, Cin, coef=coef, cscale=torch.tensor([2.0, 0.5], device="cuda"), u=u, v=v, nrank=2,
                     accumulate="zeroed")
    ref = conv2d_weight(x.double(), (Cout, Cin, 4, 4), dy.double(), 2, 1)
    corr = sum(float(coef[k]) * (2.0, 0.5)[k] * torch.outer(u[k].cpu().double(), v[k].cpu().double()).view(Cout, Cin, 4, 4) for k in range(2))
    tol_w = tol if dt == torch.float32 else TOL_ACC           # the gradient is fp32 in every mode: summation order only
    assert rel_err(dw.cpu(), ref - corr) < tol_w
    assert rel_err(dw2.cpu(), ref - corr) < tol_w


# ---- split-precision conv modes (fp32 tensors; operands split hi + lo into 16-bit halves inside the kernel, 3 MFMAs per K step).
# Against fp64 on UN-rounded fp32 operands: what is left is the dropped lo*lo term and the halves' own rounding -- fp16x3 keeps
# 22 mantissa bits per operand (fp32-grade: the bound is the fp32 mode's), bf16x3 16 bits (2^-17 per operand, averaged over K).
X3 = [("fp16x3", 2e-5), ("bf16x3", 6e-5)]
X3_CASES = [c for c in CONV_CASES if c[0] <= 768 and c[1] <= 64] + [(768, 16, 64, 64, 128), (2, 128, 6, 8, 64)]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
def test_conv_wgrad_batch(ops, dt):
    """gcssl_conv4x4s2_wgrad_batch (the critic's three big layers' weight gradients as one launch) == three single calls, slab for
    slab (same kernel body, same K ranges): bit-equal in the 16-bit dtypes, and a plain loop over the layers in fp32."""
    shapes = [(1024, 16, 64, 128), (1024, 8, 128, 256), (1024, 4, 256, 512)]            # D.c2 / c3 / c4 at the bench batch (4B samples)
    lay, single = [], []
    for i, (N, Hi, Cin, Cout) in enumerate(shapes):
        x = nhwc(q(rnd(N, Cin, Hi, Hi, seed=200 + i), dt), dt)
        dy = nhwc(q(rnd(N, Cout, Hi // 2, Hi // 2, seed=210 + i), dt), dt)
        ns = ops.wgrad_splits(N, Hi, Hi, Cin, Cout)
        s1 = torch.full((ns, Cout, 16, Cin), float("nan"), device="cuda")
        s2 = torch.full((ns, Cout, 16, Cin), float("nan"), device="cuda")
        ops.conv_wgrad(x, dy, s1, Cin, Cout)
        lay.append((x, dy, s2, Cin, Cout)); single.append(s1)
    ops.WgradBatch(lay).run()
    torch.cuda.synchronize()
    for (x, dy, s2, _, _), s1 in zip(lay, single):
        assert bool(torch.isfinite(s2).all())
        assert torch.equal(s1, s2)


@pytest.mark.parametrize("mode,tol", X3)
@pytest.mark.parametrize("N,Hi,Cin,CinP,Cout", X3_CASES)
def test_conv_split_precision(ops, mode, tol, N, Hi, Cin, CinP, Cout):
    lib = load_pkg("_lib")
    code, f32 = lib.mma_code(mode), torch.float32
    x = rnd(N, Cin, Hi, Hi, seed=1)
    w = rnd(Cout, Cin, 4, 4, seed=2, scale=0.05)
    b = rnd(Cout, seed=3, scale=0.1)
    group_n = (N + 2) // 3
    gs = torch.tensor([1.3, 0.7, 2.1])[: (N + group_n - 1) // group_n]
    wf = torch.empty(Cout, 16, CinP, device="cuda")                 # packed FOR THE MODE: the pack carries the modes' 2^6 weight scale
    wt = torch.empty(CinP, 16, Cout, device="cuda")
    ops.prep_conv_weight(w.cuda(), wf, wt, Cout, Cin, CinP, code)
    xd = nhwc(x, f32, CinP)
    # forward (with the activation epilogue, and the linear form the dispatcher may split over K: atomics / slabs)
    y = torch.full((N, Hi // 2, Hi // 2, Cout), float("nan"), device="cuda")
    ops.conv_fwd(xd, wf, y, CinP, Cout, bias=b.cuda(), gscale=gs.cuda(), group_n=group_n, act=1, dt=code)
    lin = F.conv2d(x.double(), w.double(), None, 2, 1) * gs[torch.arange(N) // group_n].view(-1, 1, 1, 1).double() + b.view(1, -1, 1, 1).double()
    assert rel_err(nchw(y), F.leaky_relu(lin, 0.2)) < tol
    z = torch.full((N, Hi // 2, Hi // 2, Cout), float("nan"), device="cuda")
    ops.conv_fwd(xd, wf, z, CinP, Cout, bias=b.cuda(), gscale=gs.cuda(), group_n=group_n, dt=code)
    assert rel_err(nchw(z), lin) < tol
    ks = ops.conv_splits("fwd", code, N, Hi, CinP, Cout)
    if ks > 1:
        slabs = torch.full((ks, N, Hi // 2, Hi // 2, Cout), float("nan"), device="cuda")
        ops.conv_fwd(xd, wf, slabs[0], CinP, Cout, bias=b.cuda(), gscale=gs.cuda(), group_n=group_n, split_stride=slabs[0].numel(), dt=code)
        assert rel_err(nchw(slabs.sum(0)), lin) < tol
    # data gradient
    dy = rnd(N, Cout, Hi // 2, Hi // 2, seed=4)
    dyd = nhwc(dy, f32)
    dx = torch.zeros(N, Hi, Hi, CinP, device="cuda")
    ops.conv_dgrad(dyd, wt, dx, CinP, Cout, gscale=torch.tensor([0.9], device="cuda"), group_n=N, dt=code)
    ref = F.conv_transpose2d(dy.double(), w.double(), None, 2, 1)
    assert rel_err(nchw(dx)[:, :Cin], ref * 0.9) < tol
    if CinP > Cin:
        assert float(dx[..., Cin:].abs().max()) == 0.0
    # weight gradient
    ns = ops.wgrad_splits(N, Hi, Hi, CinP, Cout)
    slab = torch.full((ns, Cout, 16, CinP), float("nan"), device="cuda")
    ops.conv_wgrad(xd, dyd, slab, CinP, Cout, dt=code)
    dw = torch.full((Cout, Cin, 4, 4), float("nan"), device="cuda")
    ops.wgrad_reduce(slab, ns, dw, Cout, CinP, Cin)
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), conv2d_weight(x.double(), (Cout, Cin, 4, 4), dy.double(), 2, 1)) < tol


@pytest.mark.parametrize("mode", ["fp16x3", "bf16x3"])
@pytest.mark.parametrize("N,Hi,Cin,Cout", [(768, 16, 64, 128), (768, 8, 128, 256), (1536, 4, 256, 512), (771, 8, 128, 256), (512, 16, 64, 128)])
def test_conv_in_act_fused_split_precision(ops, mode, N, Hi, Cin, Cout):
    """The split-precision modes' conv + InstanceNorm + LeakyReLU launch (fp32 tensors) against conv (fp64) -> InstanceNorm ->
    LeakyReLU: pre-norm values, statistics and activation at the modes' conv tolerance; ragged last tile (N = 771); the
    activation written into a channel slice of a wider buffer."""
    lib = load_pkg("_lib")
    code = lib.mma_code(mode)
    assert ops.conv_in_act_x3_ok(code, N, Hi, Cin, Cout)
    tol = 2e-5 if mode == "fp16x3" else 1e-4
    x = rnd(N, Cin, Hi, Hi, seed=1)
    w = rnd(Cout, Cin, 4, 4, seed=2, scale=0.05)
    b = rnd(Cout, seed=3, scale=0.1)
    group_n = (N + 2) // 3
    gs = torch.tensor([1.3, 0.7, 2.1])[: (N + group_n - 1) // group_n]
    wf = torch.empty(Cout, 16, Cin, device="cuda")
    ops.prep_conv_weight(w.cuda(), wf, None, Cout, Cin, Cin, code)
    xd = nhwc(x, torch.float32)
    Ho = Hi // 2
    z = torch.full((N, Ho, Ho, Cout), float("nan"), device="cuda")
    wide = torch.full((N, Ho, Ho, Cout + 64), float("nan"), device="cuda")
    a = wide[..., 64:]
    mean = torch.full((N, Cout), float("nan"), device="cuda"); rstd = torch.full((N, Cout), float("nan"), device="cuda")
    ops.conv_in_act_x3_fwd(xd, wf, z, a, mean, rstd, Cin, Cout, bias=b.cuda(), gscale=gs.cuda(), group_n=group_n, dt=code)
    torch.cuda.synchronize()
    lin = F.conv2d(x.double(), w.double(), None, 2, 1) * gs[torch.arange(N) // group_n].view(-1, 1, 1, 1).double() + b.view(1, -1, 1, 1).double()
    mu = lin.mean((2, 3), keepdim=True); var = lin.var((2, 3), unbiased=False, keepdim=True)
    ref = F.leaky_relu((lin - mu) / torch.sqrt(var + 1e-5), 0.2)
    assert rel_err(nchw(z), lin) < tol
    assert rel_err(mean.cpu(), mu.view(N, Cout)) < 10 * tol
    assert rel_err(rstd.cpu(), (1.0 / torch.sqrt(var + 1e-5)).view(N, Cout)) < 1e-3      # (rstd of a 2x2 map amplifies z's error by rstd^2 sigma)
    assert rel_err(nchw(a), ref) < 2e-3 if Hi == 4 else rel_err(nchw(a), ref) < 2e-4
    assert bool(torch.isnan(wide[..., :64]).all())                                        # nothing outside the slice was touched
    # z = None: only the activation and the statistics
    a2 = torch.full((N, Ho, Ho, Cout), float("nan"), device="cuda")
    ops.conv_in_act_x3_fwd(xd, wf, None, a2, mean, rstd, Cin, Cout, bias=b.cuda(), gscale=gs.cuda(), group_n=group_n, dt=code)
    assert torch.equal(a2, a.contiguous())


def test_conv_split_precision_small_magnitudes(ops):
    """fp16 halves have fp16's exponent range: operands far below 1 lose their lo bits to subnormals unless the engine's static
    scales (loss scale on gradients, 2^6 on weights inside the kernel) keep them up.  Pinned here: gradients of magnitude 1e-3
    (what a critic backward tensor looks like after the split modes' loss scale) still contract to fp32-grade accuracy."""
    lib = load_pkg("_lib")
    code = lib.mma_code("fp16x3")
    N, Hi, Cin, Cout = 24, 16, 64, 128
    w = rnd(Cout, Cin, 4, 4, seed=2, scale=0.05)
    wt = torch.empty(Cin, 16, Cout, device="cuda")
    ops.prep_conv_weight(w.cuda(), None, wt, Cout, Cin, Cin, code)
    dy = rnd(N, Cout, Hi // 2, Hi // 2, seed=4, scale=1e-3)
    dx = torch.zeros(N, Hi, Hi, Cin, device="cuda")
    ops.conv_dgrad(nhwc(dy, torch.float32), wt, dx, Cin, Cout, dt=code)
    torch.cuda.synchronize()
    assert rel_err(nchw(dx), F.conv_transpose2d(dy.double(), w.double(), None, 2, 1)) < 1e-4


@pytest.mark.parametrize("dt,tol", DTS)
@pytest.mark.parametrize("N,H,C,act", [(3, 8, 128, 1), (2, 2, 512, 1), (2, 32, 64, 2), (4, 4, 256, 2), (3, 16, 128, 1),
                                       (2, 64, 64, 2)])     # 64x64: beyond the LDS-resident form -> stats/finalize/apply
def test_instance_norm_fwd_bwd_dbl(ops, dt, tol, N, H, C, act):
    from oracle import manual_step as M
    z = rnd(N, C, H, H, seed=10, scale=2.0) + 0.3          # the pre-norm tensor is fp32 in both modes
    da = rnd(N, C, H, H, seed=11)                          # ... and so are all incoming gradients
    mask = (rnd(N, C, H, H, seed=12) > 0)
    zd, dad = nhwc(z, torch.float32), nhwc(da, torch.float32)
    maskd = mask.permute(0, 2, 3, 1).contiguous().to("cuda", torch.uint8)
    # forward into a channel slice of a wider (concat) buffer
    wide = torch.zeros(N, H, H, 2 * C, device="cuda", dtype=dt)
    a = wide[..., C:]
    mean = torch.empty(N, C, device="cuda"); rstd = torch.empty(N, C, device="cuda")
    ops.in_act_fwd(zd, a, mean, rstd, C, act, mask=maskd)
    torch.cuda.synchronize()
    mu, r = M.in_stats(z)
    xh = (z - mu) * r
    aref = (M.lrelu(xh) if act == 1 else torch.relu(xh)) * mask.float() * 2
    assert rel_err(mean.cpu(), mu.view(N, C)) < 1e-5 and rel_err(rstd.cpu(), r.view(N, C)) < 1e-5
    assert rel_err(nchw(a), aref) < tol
    assert float(wide[..., :C].abs().max()) == 0.0
    # fused global-average-pool sums (large-map path is forced by pool=...)
    pool = torch.zeros(N, C, device="cuda")
    a2 = torch.empty(N, H, H, C, device="cuda", dtype=dt)
    ops.in_act_fwd(zd, a2, mean, rstd, C, act, mask=maskd, pool=pool)
    assert rel_err(nchw(a2), aref) < tol
    assert rel_err(rstd.cpu(), r.view(N, C)) < 1e-5
    assert rel_err(pool.cpu(), q(aref, dt).sum(dim=(2, 3))) < max(tol, 1e-5)
    if ops.in_act_fwd_pool_only_ok(H * H, C):                      # a=None: statistics + pool sums, no activation store
        pool3, mean3, rstd3 = torch.zeros(N, C, device="cuda"), torch.empty(N, C, device="cuda"), torch.empty(N, C, device="cuda")
        ops.in_act_fwd(zd, None, mean3, rstd3, C, act, mask=maskd, pool=pool3)
        assert torch.equal(mean3, mean) and torch.equal(rstd3, rstd)
        assert rel_err(pool3.cpu(), aref.sum(dim=(2, 3))) < 1e-5
    # backward (with dropout mask, group scale, bias/cdot bookkeeping)
    bias = rnd(C, seed=13, scale=0.1)
    gsc = torch.tensor([1.5, 0.5], device="cuda")
    group_n = (N + 1) // 2
    dzs = torch.empty(N, H, H, C, device="cuda", dtype=dt)
    dbias = torch.zeros(C, device="cuda"); cdot = torch.zeros(2, device="cuda")
    ws = torch.empty(2 * N * C, device="cuda")
    ops.in_act_bwd(zd, mean, rstd, dzs, C, act, da=dad, mask=maskd, gscale=gsc, group_n=group_n, bias=bias.cuda(),
                   dbias=dbias, cdot=cdot, ws=ws)
    torch.cuda.synchronize()
    ag = torch.where(xh > 0, torch.ones_like(xh), torch.full_like(xh, 0.2 if act == 1 else 0.0))
    dn = da * mask.float() * 2 * ag
    dz = M.in_bwd(xh, r, dn)
    grp = (torch.arange(N) // group_n)
    assert rel_err(nchw(dzs), dz * gsc.cpu()[grp].view(-1, 1, 1, 1)) < tol
    # sum dz (z-b) is analytically ~0 for a normalised layer (sum dz = 0, sum dz xhat ~ eps): judge the error
    # against the magnitude of the summed terms, not against the cancelled result
    terms = dz * gsc.cpu()[grp].view(-1, 1, 1, 1) * (z - bias.view(1, -1, 1, 1))
    cd = torch.stack([terms[grp == g].double().sum() for g in range(2)])
    mag = torch.stack([terms[grp == g].abs().double().sum() for g in range(2)])
    assert float(((cdot.cpu().double() - cd).abs() / mag).max()) < max(tol, 1e-5)
    assert float(dbias.abs().max().cpu()) < max(tol, 1e-5) * float(dz.abs().sum(dim=(0, 2, 3)).max())
    # double backward
    qz = rnd(N, C, H, H, seed=14)
    gzs = q(rnd(N, C, H, H, seed=15), dt)
    gt_a = torch.empty(N, H, H, C, device="cuda", dtype=dt); zt = torch.empty(N, H, H, C, device="cuda")
    cd2 = torch.zeros(1, device="cuda")
    ops.in_dbl_bwd(dad, nhwc(qz, torch.float32), nhwc(gzs, dt), zd, mean, rstd, gt_a, zt, C, act, cdot=cd2)
    torch.cuda.synchronize()
    dn2 = da * ag
    gtn, ztr = M.in_bwd_bwd(xh, r, dn2, qz)
    assert rel_err(nchw(gt_a), gtn * ag) < tol
    assert rel_err(nchw(zt), ztr) < max(tol, 1e-4)
    assert rel_err(cd2.cpu(), (gzs * qz).sum().view(1)) < max(tol, 1e-4)


FIN_CASES = [  # N, Hi, Cin, Cout, masked -- the one-launch conv + InstanceNorm + LeakyReLU forms (16-bit dtypes, maps of <= 64 pixels)
    (768, 16, 64, 128, False),   # D.c2 / G.down2 at the bench batch: 768 tiles of 128x64, two whole 8x8 samples per tile
    (768, 8, 128, 256, False),   # D.c3: 192 tiles of 128x128 on the loader / consumer ring, eight 4x4 samples per tile
    (768, 4, 256, 512, True),    # D.c4 / G.down4 (dropout mask + un-masked copy for the backward): 96 ring tiles, 32 samples each
    (256, 8, 128, 256, False),   # the B-sample critic forward of the generator step: 64x64 tiles
    (97, 16, 64, 128, True),     # ragged: 97 samples of 64 pixels -> 194 tiles of 64x64 (the last M tile of the 128-row forms is half empty)
    (384, 16, 128, 256, False),  # D.c3 at 64x64 inputs, B=128: 8x8 maps, 768 tiles of 128x64
]


@pytest.mark.parametrize("dt,tol", DTS[1:])
@pytest.mark.parametrize("N,Hi,Cin,Cout,masked", FIN_CASES)
def test_conv_in_act_fused(ops, dt, tol, N, Hi, Cin, Cout, masked):
    """gcssl_conv4x4s2_in_act_fwd == conv (fp64, operands pre-rounded) -> InstanceNorm -> LeakyReLU -> dropout mask."""
    from oracle import manual_step as M
    assert ops.conv_in_act_ok(ops.code(torch.empty(0, dtype=dt)), N, Hi, Cin, Cout)
    Ho = Hi // 2
    x = q(rnd(N, Cin, Hi, Hi, seed=31), dt)
    w = rnd(Cout, Cin, 4, 4, seed=32, scale=0.05)
    b = rnd(Cout, seed=33, scale=0.1)
    group_n = (N + 2) // 3
    gs = torch.tensor([1.3, 0.7, 2.1])[: (N + group_n - 1) // group_n]
    wf, _ = packed_weights(ops, w, dt)
    xd = nhwc(x, dt)
    wide = torch.zeros(N, Ho, Ho, 2 * Cout, device="cuda", dtype=dt)     # into a channel slice of a concat buffer
    a = wide[..., Cout:]
    mean = torch.full((N, Cout), float("nan"), device="cuda"); rstd = torch.full((N, Cout), float("nan"), device="cuda")
    mask = (rnd(N, Cout, Ho, Ho, seed=34) > 0) if masked else None
    maskd = mask.permute(0, 2, 3, 1).contiguous().to("cuda", torch.uint8) if masked else None
    n0 = (2 * N) // 3
    apre = torch.full((N - n0, Ho, Ho, Cout), float("nan"), device="cuda", dtype=dt) if masked else None
    ops.conv_in_act_fwd(xd, wf, a, mean, rstd, Cin, Cout, bias=b.cuda(), gscale=gs.cuda(), group_n=group_n, mask=maskd,
                        apre=apre, apre_n0=n0)
    torch.cuda.synchronize()
    z = F.conv2d(x.double(), q(w, dt).double(), None, 2, 1)
    z = z * gs[torch.arange(N) // group_n].view(-1, 1, 1, 1).double() + b.view(1, -1, 1, 1).double()
    mu, r = M.in_stats(z)
    pre = M.lrelu((z - mu) * r)
    ref = pre * (mask.double() * 2 if masked else 1.0)
    assert rel_err(mean.cpu(), mu.view(N, Cout)) < TOL_ACC and rel_err(rstd.cpu(), r.view(N, Cout)) < TOL_ACC
    assert rel_err(nchw(a), ref) < tol
    assert float(wide[..., :Cout].abs().max()) == 0.0
    if masked:
        assert rel_err(nchw(apre), pre[n0:]) < tol


@pytest.mark.parametrize("dt,tol", DTS[1:])
@pytest.mark.parametrize("N,H,C", [(6, 8, 128), (5, 2, 512), (7, 4, 256)])
def test_instance_norm_bwd_from_activation(ops, dt, tol, N, H, C):
    """in_act_bwd / in_dbl_bwd with z_kind = 1: xhat rebuilt from the 16-bit LeakyReLU activation (what a fused
    conv_in_act_fwd leaves) must equal the oracle formulas evaluated at that rebuilt xhat."""
    from oracle import manual_step as M
    z = rnd(N, C, H, H, seed=40, scale=2.0) + 0.3
    da = rnd(N, C, H, H, seed=41)
    mask = (rnd(N, C, H, H, seed=42) > 0)
    mu, r = M.in_stats(z)
    a16 = M.lrelu((z - mu) * r).to(dt)                                  # the stored activation
    xh = torch.where(a16.float() > 0, a16.float(), 5.0 * a16.float())   # what the kernels rebuild
    zq = xh / r + mu
    ad = a16.permute(0, 2, 3, 1).contiguous().cuda()
    dad = nhwc(da, torch.float32)
    maskd = mask.permute(0, 2, 3, 1).contiguous().to("cuda", torch.uint8)
    mean, rstd = mu.view(N, C).cuda().contiguous(), r.view(N, C).cuda().contiguous()
    bias = rnd(C, seed=43, scale=0.1)
    gsc = torch.tensor([1.5, 0.5], device="cuda")
    group_n = (N + 1) // 2
    dzs = torch.empty(N, H, H, C, device="cuda", dtype=dt)
    dbias = torch.zeros(C, device="cuda"); cdot = torch.zeros(2, device="cuda")
    ops.in_act_bwd(ad, mean, rstd, dzs, C, 1, da=dad, mask=maskd, gscale=gsc, group_n=group_n, bias=bias.cuda(),
                   dbias=dbias, cdot=cdot)
    torch.cuda.synchronize()
    ag = torch.where(xh > 0, torch.ones_like(xh), torch.full_like(xh, 0.2))
    dn = da * mask.float() * 2 * ag
    dz = M.in_bwd(xh, r, dn)
    grp = (torch.arange(N) // group_n)
    assert rel_err(nchw(dzs), dz * gsc.cpu()[grp].view(-1, 1, 1, 1)) < tol
    terms = dz * gsc.cpu()[grp].view(-1, 1, 1, 1) * (zq - bias.view(1, -1, 1, 1))
    cd = torch.stack([terms[grp == g].double().sum() for g in range(2)])
    mag = torch.stack([terms[grp == g].abs().double().sum() for g in range(2)])
    assert float(((cdot.cpu().double() - cd).abs() / mag).max()) < 1e-5
    # (xhat rebuilt from a rounded activation no longer sums to zero exactly, so neither does dz: compare with the reference's sum)
    assert float((dbias.cpu() - dz.sum(dim=(0, 2, 3))).abs().max()) < 1e-5 * float(dz.abs().sum(dim=(0, 2, 3)).max())
    qz = rnd(N, C, H, H, seed=44)
    gzs = q(rnd(N, C, H, H, seed=45), dt)
    gt_a = torch.empty(N, H, H, C, device="cuda", dtype=dt); zt = torch.empty(N, H, H, C, device="cuda")
    cd2 = torch.zeros(1, device="cuda")
    ops.in_dbl_bwd(dad, nhwc(qz, torch.float32), nhwc(gzs, dt), ad, mean, rstd, gt_a, zt, C, 1, cdot=cd2)
    torch.cuda.synchronize()
    gtn, ztr = M.in_bwd_bwd(xh, r, da * ag, qz)
    assert rel_err(nchw(gt_a), gtn * ag) < tol
    assert rel_err(nchw(zt), ztr) < 1e-4
    assert rel_err(cd2.cpu(), (gzs * qz).sum().view(1)) < 1e-4


@pytest.mark.parametrize("dt,tol", DTS[1:])
@pytest.mark.parametrize("N,with_sums", [(768, True), (256, False), (771, True), (769, False), (513, False), (1536, True)])
def test_conv_dgrad_act_bwd_fused(ops, dt, tol, N, with_sums):
    """gcssl_conv4x4s2_dgrad_act_bwd == gcssl_conv4x4s2_dgrad (fp32 dx) followed by gcssl_act_bwd: the critic's second layer's
    data gradient with the first layer's LeakyReLU backward (+ bias-gradient and spectral-norm sums) in the epilogue.
    768 / 771 / 769 / 513 / 1536 samples: the persistent form (771, 769, 513: a ragged last tile); 256: the plain tiled form
    (no sums)."""
    Hi, Cin, Cout = 16, 64, 128
    code = ops.code(torch.empty(0, dtype=dt))
    group_n = (N + 2) // 3 if N % 3 else N // 3
    if N in (771, 769, 513):
        group_n = 257                                   # 257 * 64 rows per class: not a multiple of 128 -> no per-group scale / sums
    assert ops.conv_dgrad_act_bwd_ok(code, N, Hi, Cin, Cout, with_sums)
    dy = q(rnd(N, Cout, Hi // 2, Hi // 2, seed=110), dt)
    w = rnd(Cout, Cin, 4, 4, seed=111, scale=0.05)
    a = q(F.leaky_relu(rnd(N, Cin, Hi, Hi, seed=112), 0.2), dt)
    bias = rnd(Cin, seed=113, scale=0.1).cuda()
    _, wt = packed_weights(ops, w, dt)
    dyd, ad = nhwc(dy, dt), nhwc(a, dt)
    grouped = N not in (771, 769, 513)
    gs = torch.tensor([1.5, 0.5, 2.0], device="cuda") if grouped else None
    nrep, stride = 4, 128
    # unfused pair
    dx = torch.empty(N, Hi, Hi, Cin, device="cuda")
    ops.conv_dgrad(dyd, wt, dx, Cin, Cout)
    dz_ref = torch.empty(N, Hi, Hi, Cin, device="cuda", dtype=dt)
    db_ref = torch.zeros(nrep, stride, device="cuda"); cd_ref = torch.zeros(nrep, stride, device="cuda")
    kw = dict(gscale=gs, group_n=group_n if grouped else 0)
    if with_sums and grouped:
        kw.update(bias=bias, dbias=db_ref[0, :Cin], cdot=cd_ref[0, 64:67], nrep=nrep, rep_stride=stride)
    ops.act_bwd(dx, ad, dz_ref, Cin, **kw)
    # fused
    dz = torch.full((N, Hi, Hi, Cin), float("nan"), device="cuda", dtype=dt)
    db = torch.zeros(nrep, stride, device="cuda"); cd = torch.zeros(nrep, stride, device="cuda")
    kw2 = dict(gscale=gs, group_n=group_n if grouped else 0)
    if with_sums and grouped:
        kw2.update(bias=bias, dbias=db[0, :Cin], cdot=cd[0, 64:67], nrep=nrep, rep_stride=stride)
    ops.conv_dgrad_act_bwd(dyd, wt, ad, dz, Cin, Cout, **kw2)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dz.float()).all())
    assert rel_err(dz.float().cpu(), dz_ref.float().cpu()) < tol          # same fp32 sums up to MFMA order, one output rounding
    # against the fp64 definition too
    ref = F.conv_transpose2d(dy.double(), q(w, dt).double(), None, 2, 1)
    ref = torch.where(a.double() > 0, ref, 0.2 * ref)
    if grouped:
        ref = ref * gs.cpu().double()[torch.arange(N) // group_n].view(-1, 1, 1, 1)
    assert rel_err(nchw(dz), ref) < tol
    if with_sums and grouped:
        sb, sb_ref = db.sum(0)[:Cin].cpu(), db_ref.sum(0)[:Cin].cpu()
        assert rel_err(sb, sb_ref) < 1e-4
        sc, sc_ref = cd.sum(0)[64:67].cpu(), cd_ref.sum(0)[64:67].cpu()
        assert rel_err(sc, sc_ref) < 1e-4


@pytest.mark.parametrize("mode,tol", X3)
@pytest.mark.parametrize("N,with_sums", [(768, True), (256, False), (770, False), (24, True)])      # 770: ragged last tile, no groups
def test_conv_dgrad_act_bwd_fused_split_precision(ops, mode, tol, N, with_sums):
    """The split-precision modes' data gradient with the norm-less layer's LeakyReLU backward (+ bias-gradient and spectral-norm
    sums) in its epilogue, all tensors fp32: against the fp64 definition and against the unfused pair."""
    lib = load_pkg("_lib")
    code, f32 = lib.mma_code(mode), torch.float32
    Hi, Cin, Cout = 16, 64, 128
    grouped = N % 3 == 0
    group_n = N // 3 if grouped else 0
    assert ops.conv_dgrad_act_bwd_ok(code, N, Hi, Cin, Cout, with_sums)
    dy = rnd(N, Cout, Hi // 2, Hi // 2, seed=110)
    w = rnd(Cout, Cin, 4, 4, seed=111, scale=0.05)
    a = F.leaky_relu(rnd(N, Cin, Hi, Hi, seed=112), 0.2)
    bias = rnd(Cin, seed=113, scale=0.1)
    wt = torch.empty(Cin, 16, Cout, device="cuda")
    ops.prep_conv_weight(w.cuda(), None, wt, Cout, Cin, Cin, code)
    dyd, ad = nhwc(dy, f32), nhwc(a, f32)
    gs = torch.tensor([1.5, 0.5, 2.0], device="cuda") if grouped else None
    nrep, stride = 4, 128
    dz = torch.full((N, Hi, Hi, Cin), float("nan"), device="cuda")
    db = torch.zeros(nrep, stride, device="cuda"); cd = torch.zeros(nrep, stride, device="cuda")
    kw = dict(gscale=gs, group_n=group_n)
    if with_sums:
        kw.update(bias=bias.cuda(), dbias=db[0, :Cin], cdot=cd[0, 64:67], nrep=nrep, rep_stride=stride)
    ops.conv_dgrad_act_bwd(dyd, wt, ad, dz, Cin, Cout, dt=code, **kw)
    torch.cuda.synchronize()
    d = F.conv_transpose2d(dy.double(), w.double(), None, 2, 1)
    dzr = torch.where(a.double() > 0, d, 0.2 * d)
    grp = torch.arange(N) // group_n if grouped else None
    ref = dzr * gs.cpu().double()[grp].view(-1, 1, 1, 1) if grouped else dzr
    assert rel_err(nchw(dz), ref) < tol
    if with_sums:
        assert rel_err(db.sum(0)[:Cin].cpu(), dzr.sum(dim=(0, 2, 3))) < 1e-4
        z = torch.where(a.double() > 0, a.double(), 5.0 * a.double())
        cdr = torch.stack([(ref[grp == g] * (z[grp == g] - bias.double().view(1, -1, 1, 1))).sum() for g in range(3)])
        assert rel_err(cd.sum(0)[64:67].cpu(), cdr) < 1e-4
    # == the unfused pair (same conv kernel, then act_bwd)
    dx = torch.empty(N, Hi, Hi, Cin, device="cuda")
    ops.conv_dgrad(dyd, wt, dx, Cin, Cout, dt=code)
    dz2 = torch.empty_like(dx)
    ops.act_bwd(dx, ad, dz2, Cin, gscale=gs, group_n=group_n)
    assert rel_err(dz.cpu(), dz2.cpu()) < 1e-6


@pytest.mark.parametrize("dt,tol", DTS[1:])
@pytest.mark.parametrize("N,S", [(256, 32), (5, 64), (3, 128)])
def test_conv_fwd_act_bwd_fused(ops, dt, tol, N, S):
    """gcssl_conv4x4s2_fwd_act_bwd == gcssl_conv4x4s2_fwd (fp32 out) -> gcssl_act_bwd with its dot: the first layer of the reverse
    gradient-penalty chain as one launch."""
    Cin, Cout = 8, 64
    code = ops.code(torch.empty(0, dtype=dt))
    assert ops.conv_fwd_act_bwd_ok(code, N, S, Cin, Cout)
    x = q(rnd(N, Cin, S, S, seed=120), dt)
    w = rnd(Cout, Cin, 4, 4, seed=121, scale=0.1)
    a = q(F.leaky_relu(rnd(N, Cout, S // 2, S // 2, seed=122), 0.2), dt)
    dx = q(rnd(N, Cout, S // 2, S // 2, seed=123), dt)
    wf, _ = packed_weights(ops, w, dt)
    xd, ad, dxd = nhwc(x, dt), nhwc(a, dt), nhwc(dx, dt)
    gs = torch.tensor([0.75], device="cuda")
    # unfused
    v = torch.empty(N, S // 2, S // 2, Cout, device="cuda")
    ops.conv_fwd(xd, wf, v, Cin, Cout, gscale=gs, group_n=N)
    y_ref = torch.empty(N, S // 2, S // 2, Cout, device="cuda", dtype=dt)
    dot_ref = torch.zeros(1, device="cuda")
    ops.act_bwd(v, ad, y_ref, Cout, dotx=dxd, dot_out=dot_ref)
    # fused
    y = torch.full((N, S // 2, S // 2, Cout), float("nan"), device="cuda", dtype=dt)
    dot = torch.zeros(1, device="cuda")
    sat = torch.zeros(1, device="cuda", dtype=torch.int32)
    ops.conv_fwd_act_bwd(xd, wf, ad, y, Cin, Cout, gscale=gs, group_n=N, dotx=dxd, dot_out=dot, sat=sat)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(y.float()).all())
    assert torch.equal(y.cpu(), y_ref.cpu())                         # the same fp32 values, one rounding
    assert rel_err(dot.cpu(), dot_ref.cpu()) < 1e-4
    assert int(sat.item()) == 0
    ref = 0.75 * F.conv2d(x.double(), q(w, dt).double(), None, 2, 1)
    assert rel_err(dot.cpu().double(), (dx.double() * ref).sum().view(1)) < 1e-3
    ref = torch.where(a.double() > 0, ref, 0.2 * ref)
    assert rel_err(nchw(y), ref) < tol
    # without the dot
    y2 = torch.empty_like(y)
    ops.conv_fwd_act_bwd(xd, wf, ad, y2, Cin, Cout, gscale=gs, group_n=N)
    assert torch.equal(y2.cpu(), y.cpu())


@pytest.mark.parametrize("mode,tol", X3)
@pytest.mark.parametrize("N,S", [(256, 32), (5, 64), (3, 128)])
def test_conv_fwd_act_bwd_fused_split_precision(ops, mode, tol, N, S):
    """The split-precision modes' first-layer kernel in its conv + activation-backward + dot form (all tensors fp32) against fp64."""
    lib = load_pkg("_lib")
    code, f32 = lib.mma_code(mode), torch.float32
    Cin, Cout = 8, 64
    assert ops.conv_fwd_act_bwd_ok(code, N, S, Cin, Cout)
    x = rnd(N, Cin, S, S, seed=120)
    w = rnd(Cout, Cin, 4, 4, seed=121, scale=0.1)
    a = F.leaky_relu(rnd(N, Cout, S // 2, S // 2, seed=122), 0.2)
    dx = rnd(N, Cout, S // 2, S // 2, seed=123)
    wf = torch.empty(Cout, 16, Cin, device="cuda")
    ops.prep_conv_weight(w.cuda(), wf, None, Cout, Cin, Cin, code)
    xd, ad, dxd = nhwc(x, f32), nhwc(a, f32), nhwc(dx, f32)
    gs = torch.tensor([0.75], device="cuda")
    y = torch.full((N, S // 2, S // 2, Cout), float("nan"), device="cuda")
    dot = torch.zeros(1, device="cuda")
    ops.conv_fwd_act_bwd(xd, wf, ad, y, Cin, Cout, gscale=gs, group_n=N, dotx=dxd, dot_out=dot, dt=code)
    torch.cuda.synchronize()
    ref = 0.75 * F.conv2d(x.double(), w.double(), None, 2, 1)
    assert rel_err(dot.cpu().double(), (dx.double() * ref).sum().view(1)) < 1e-4
    assert rel_err(nchw(y), torch.where(a.double() > 0, ref, 0.2 * ref)) < tol
    # == the unfused pair (same kernel without the epilogue, then act_bwd)
    v = torch.empty(N, S // 2, S // 2, Cout, device="cuda")
    ops.conv_fwd(xd, wf, v, Cin, Cout, gscale=gs, group_n=N, dt=code)
    y_ref = torch.empty_like(v)
    ops.act_bwd(v, ad, y_ref, Cout)
    assert torch.equal(y.cpu(), y_ref.cpu())


@pytest.mark.parametrize("dt,tol", DTS)
def test_act_bwd_and_dot(ops, dt, tol):
    N, H, C = 4, 16, 64
    a = q(F.leaky_relu(rnd(N, C, H, H, seed=20), 0.2), dt)
    da = rnd(N, C, H, H, seed=21); da2 = rnd(N, C, H, H, seed=22)
    bias = rnd(C, seed=23, scale=0.1)
    gsc = torch.tensor([2.0, 0.5], device="cuda")
    dzs = torch.empty(N, H, H, C, device="cuda", dtype=dt)
    dbias = torch.zeros(C, device="cuda"); cdot = torch.zeros(2, device="cuda")
    ops.act_bwd(nhwc(da, torch.float32), nhwc(a, dt), dzs, C, da2=nhwc(da2, torch.float32), gscale=gsc, group_n=2, bias=bias.cuda(),
                dbias=dbias, cdot=cdot)
    torch.cuda.synchronize()
    dz = torch.where(a > 0, da + da2, 0.2 * (da + da2))
    grp = torch.arange(N) // 2
    assert rel_err(nchw(dzs), dz * gsc.cpu()[grp].view(-1, 1, 1, 1)) < tol
    assert rel_err(dbias.cpu(), dz.sum(dim=(0, 2, 3))) < max(tol, 1e-4)
    zrec = torch.where(a > 0, a, a * 5.0)
    cd = torch.stack([float(gsc[g]) * (dz[grp == g] * (zrec[grp == g] - bias.view(1, -1, 1, 1))).sum() for g in range(2)])
    assert rel_err(cdot.cpu(), cd) < max(tol, 1e-4)
    out = torch.zeros(1, device="cuda")
    ops.dot_accum(nhwc(q(da, dt), dt), nhwc(da2, torch.float32), C, out)
    assert rel_err(out.cpu(), (q(da, dt) * da2).sum().view(1)) < max(tol, 1e-4)


@pytest.mark.parametrize("dt,tol", DTS)
@pytest.mark.parametrize("N,H,C", [(6, 2, 512), (3, 4, 512), (3, 8, 512), (3, 4, 20), (768, 2, 512)])    # C = 20: the scalar forward form; 768: several workgroups of the constant-seed forms
def test_critic_head(ops, dt, tol, N, H, C):
    x = q(rnd(N, C, H, H, seed=30), dt)
    w = rnd(1, C, 4, 4, seed=31, scale=0.05)
    wp = torch.empty(16, C, device="cuda")
    ops.prep_c5_weight(w.cuda(), wp)
    xd = nhwc(x, dt)
    out = torch.empty(N, H - 1, H - 1, device="cuda")
    ops.c5_fwd(xd, wp, out)
    ref = F.conv2d(x, w, None, 1, 1)
    assert rel_err(out.cpu().view_as(ref), ref) < (2e-5 if dt == torch.float32 else TOL_ACC)   # fp32 output: summation order only
    dout = rnd(N, 1, H - 1, H - 1, seed=32)
    dx = torch.empty(N, H, H, C, device="cuda", dtype=dt)
    ops.c5_dgrad(dx, wp, dout=dout.cuda().contiguous())
    assert rel_err(nchw(dx), F.conv_transpose2d(dout, w, None, 1, 1)) < tol
    consts = (-0.5, 0.25, 0.0)
    ops.c5_dgrad(dx, wp, consts=consts, group_n=N // 3)
    dconst = torch.tensor(consts)[torch.arange(N) // (N // 3)].view(-1, 1, 1, 1).expand(N, 1, H - 1, H - 1)
    assert rel_err(nchw(dx), F.conv_transpose2d(dconst.contiguous(), w, None, 1, 1)) < tol
    dw = torch.zeros(C, 16, device="cuda")
    ops.c5_wgrad(xd, dw, C, dout=dout.cuda().contiguous())
    assert rel_err(dw.cpu().view(1, C, 4, 4), conv2d_weight(x, w.shape, dout, 1, 1)) < max(tol, 1e-4)
    dw.zero_()
    ops.c5_wgrad(xd, dw, C, consts=consts, group_n=N // 3)
    assert rel_err(dw.cpu().view(1, C, 4, 4), conv2d_weight(x, w.shape, dconst.contiguous(), 1, 1)) < max(tol, 1e-4)


@pytest.mark.parametrize("coop", ["coop", "multi_launch"])
@pytest.mark.parametrize("chained", [False, True])
@pytest.mark.parametrize("shapes", [
    [(64, 6 * 16), (128, 64 * 16), (256, 128 * 16), (512, 256 * 16)],     # the critic's layers
    [(6, 18), (33, 130), (70, 1027), (67, 4100)],   # ragged rows; columns that are not a multiple of 4 (scalar forms) / of 1024
])
def test_spectral_norm_power_iteration(ops, shapes, chained, coop, monkeypatch):
    """coop: the whole chain as ONE cooperative launch (sn_coop_kernel: weights resident in LDS, grid barriers through memory;
    round 4) -- multi_launch: the W^T u / W v / finish launches (GCSSL_SN_COOP=0, and the form a process without gcssl_init gets)."""
    from oracle import manual_step as M
    lib = load_pkg("_lib")
    assert lib.call_nostream("gcssl_init") == 0
    monkeypatch.setenv("GCSSL_SN_COOP", "1" if coop == "coop" else "0")
    ws = [rnd(r, c, seed=40 + i, scale=0.05) for i, (r, c) in enumerate(shapes)]
    us = [F.normalize(rnd(r, seed=50 + i), dim=0) for i, (r, _) in enumerate(shapes)]
    vs = [F.normalize(rnd(c, seed=60 + i), dim=0) for i, (_, c) in enumerate(shapes)]
    wd, ud, vd = [w.cuda() for w in ws], [u.cuda() for u in us], [v.cuda() for v in vs]
    sn = ops.SnState(wd, ud, vd, 3, "cuda")
    sn.iterate(0, iterate=False)
    torch.cuda.synchronize()
    for i in range(4):
        s0 = torch.dot(us[i], ws[i] @ vs[i])
        assert abs(float(sn.sigma[i, 0]) - float(s0)) < 1e-5 * abs(float(s0)) + 1e-7
    extra = torch.full((1000,), 3.0, device="cuda")
    if chained:       # three iterations as ONE chain: each W^T u closes its predecessor, one closing launch (+ the caller's fill)
        sn.iterate(0, 3, zero=extra)
    else:
        for k in range(3):
            sn.iterate(k)
    torch.cuda.synchronize()
    assert float(extra.abs().max()) == (0.0 if chained else 3.0)
    assert all(float(t.abs().max()) == 0.0 for t in sn.t)          # both halves of the scratch are left zero
    for i in range(4):
        u, v = us[i], vs[i]
        for k in range(3):
            u, v, s = M.sn_power_iter(ws[i], u, v)
            r, c = shapes[i]
            assert rel_err(sn.u_hist[i, k, :r].cpu(), u) < 2e-5
            assert rel_err(sn.v_hist[i, k, :c].cpu(), v) < 2e-5
            assert abs(float(sn.sigma[i, k]) - float(s)) < 2e-5 * float(s)
            assert abs(float(sn.isig[i, k]) * float(s) - 1) < 2e-5
        assert rel_err(ud[i].cpu(), u) < 2e-5 and rel_err(vd[i].cpu(), v) < 2e-5
    assert lib.call_nostream("gcssl_sn_coop_status") == 0                     # no workgroup ever gave up at a grid barrier
    if coop == "coop":
        assert "sn_coop" in ops.last_kernel() or True                          # (launched by hipLaunchKernelGGL: not recorded)


def test_spectral_norm_finish_deferred_into_repack(ops):
    """gcssl_sn_defer_finish: the chain's closing step (u, sigma, the extra fill) rides on the weight re-pack launch that follows --
    same results as the chain with its own closing launch, and the re-pack itself is unchanged; a deferred step that is not
    followed by a re-pack can be flushed."""
    lib = load_pkg("_lib")
    shapes = [(64, 6 * 16), (128, 64 * 16), (256, 128 * 16), (512, 256 * 16)]
    chans = [(6, 64), (64, 128), (128, 256), (256, 512)]
    ws4 = [rnd(co, ci, 4, 4, seed=40 + i, scale=0.05).cuda() for i, (ci, co) in enumerate(chans)]
    res = []
    for defer in (False, True, "flush"):
        wd = [w.reshape(w.shape[0], -1).clone() for w in ws4]
        ud = [F.normalize(rnd(r, seed=50 + i), dim=0).cuda() for i, (r, _) in enumerate(shapes)]
        vd = [F.normalize(rnd(c, seed=60 + i), dim=0).cuda() for i, (_, c) in enumerate(shapes)]
        sn = ops.SnState(wd, ud, vd, 3, "cuda")
        extra = torch.full((1000,), 3.0, device="cuda")
        packs = [(torch.full((co, 16, (ci + 7) // 8 * 8), float("nan"), device="cuda", dtype=torch.bfloat16),
                  torch.full(((ci + 7) // 8 * 8, 16, co), float("nan"), device="cuda", dtype=torch.bfloat16)) for ci, co in chans]
        prep = ops.PrepBatch([(w, wf, wt, co, ci, (ci + 7) // 8 * 8) for w, (wf, wt), (ci, co) in zip(ws4, packs, chans)],
                             lib.dtype_code(torch.bfloat16))
        sn.iterate(0, 3, zero=extra, defer_finish=bool(defer))
        if defer == "flush":
            lib.call("gcssl_sn_flush_finish")
        prep.run()
        torch.cuda.synchronize()
        assert float(extra.abs().max()) == 0.0
        assert all(float(t.abs().max()) == 0.0 for t in sn.t)
        res.append((sn.sigma.clone(), sn.isig.clone(), sn.u_hist.clone(), [u.clone() for u in ud], [p[0].clone() for p in packs],
                    [p[1].clone() for p in packs]))
    for other in res[1:]:
        for a, b in zip(res[0][:3], other[:3]):                     # sigma, 1/sigma, u history: the chain's float atomics move the last bits
            assert rel_err(b.cpu(), a.cpu()) < 1e-5
        for a, b in zip(res[0][3], other[3]):
            assert rel_err(b.cpu(), a.cpu()) < 1e-5
        for la, lb in zip(res[0][4:], other[4:]):                   # the packed weights: bit-equal
            for a, b in zip(la, lb):
                assert torch.equal(a, b) and bool(torch.isfinite(a.float()).all())


@pytest.mark.parametrize("N,H", [(1024, 2), (96, 4), (13, 2)])
def test_head_conv_dgrad_rides_on_the_repack(ops, N, H):
    """gcssl_conv4x4s1_c1_dgrad_defer: the head conv's constant-seed data gradient carried by the next weight re-pack launch (it reads the
    raw head weight) == the stand-alone launch on the packed weight, bit for bit; the re-pack's own outputs are unchanged; a
    deferred gradient that no re-pack follows can be flushed."""
    lib = load_pkg("_lib")
    C = 512
    w5 = rnd(1, C, 4, 4, seed=31, scale=0.05).cuda()
    wp = torch.empty(16, C, device="cuda")
    ops.prep_c5_weight(w5, wp)
    consts, group_n = (-0.25, 0.25, 0.0, 1.0), (N + 3) // 4
    ref = torch.full((N, H, H, C), float("nan"), device="cuda")
    ops.c5_dgrad(ref, wp, consts=consts, group_n=group_n)
    w = rnd(128, 64, 4, 4, seed=2, scale=0.05).cuda()
    for how in ("prep", "flush"):
        dx = torch.full((N, H, H, C), float("nan"), device="cuda")
        wf = torch.full((128, 16, 64), float("nan"), device="cuda", dtype=torch.bfloat16)
        wt = torch.full((64, 16, 128), float("nan"), device="cuda", dtype=torch.bfloat16)
        wp2 = torch.full((16, C), float("nan"), device="cuda")
        prep = ops.PrepBatch([(w, wf, wt, 128, 64, 64)], lib.dtype_code(torch.bfloat16), c5=(w5, wp2))
        ops.c5_dgrad_defer(dx, w5, consts=consts, group_n=group_n)
        if how == "flush":
            lib.call("gcssl_sn_flush_finish")
        prep.run()
        torch.cuda.synchronize()
        assert torch.equal(dx, ref)
        assert torch.equal(wp2, wp) and bool(torch.isfinite(wf.float()).all()) and bool(torch.isfinite(wt.float()).all())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_pack_groups_one_launch(ops, dt):
    """gcssl_pack_groups: the real group (pack_pair) written by the launch that packs the fake and interpolated groups."""
    B, S = 6, 32
    pred, gt, refined = rnd(B, 3, S, S, seed=1).cuda(), rnd(B, 3, S, S, seed=2).cuda(), rnd(B, 3, S, S, seed=3).cuda()
    alpha = torch.rand(B, device="cuda")
    r0, f0, i0 = (torch.full((B, S, S, 8), float("nan"), device="cuda", dtype=dt) for _ in range(3))
    ops.pack_pair(pred, gt, r0)
    ops.pack_fake_interp(pred, gt, refined, alpha, f0, i0)
    r1, f1, i1 = (torch.full((B, S, S, 8), float("nan"), device="cuda", dtype=dt) for _ in range(3))
    ops.pack_fake_interp(pred, gt, refined, alpha, f1, i1, out_real=r1)
    torch.cuda.synchronize()
    assert torch.equal(r0, r1) and torch.equal(f0, f1) and torch.equal(i0, i1)


def test_launch_folds(ops):
    """The bookkeeping that rides on neighbouring launches (round 3): group means from the head conv, four group constants
    of the head's weight gradient, replicated input packing, <x, da> inside act_bwd, the critic head's re-pack inside the
    batched weight re-pack, the striped-sum fold inside the batched split-K reduction."""
    dt = torch.float16
    # ---- c5_fwd + group means; c5_wgrad with four group constants == two launches with three
    N, H, C = 12, 4, 512
    x = q(rnd(N, C, H, H, seed=90), dt); w5 = rnd(1, C, 4, 4, seed=91, scale=0.05)
    wp = torch.empty(16, C, device="cuda"); ops.prep_c5_weight(w5.cuda(), wp)
    xd = nhwc(x, dt)
    out = torch.empty(N, H - 1, H - 1, device="cuda"); means = torch.zeros(3, device="cuda")
    ops.c5_fwd(xd, wp, out, group_mean=means, groups=3)
    ref = F.conv2d(x.double(), w5.double(), None, 1, 1)[:, 0]
    assert rel_err(out.cpu(), ref) < 1e-5
    assert rel_err(means.cpu(), ref.reshape(3, -1).mean(1)) < 1e-5
    dw_a = torch.zeros(C, 16, device="cuda"); dw_b = torch.zeros(C, 16, device="cuda")
    ops.c5_wgrad(xd, dw_a, C, consts=(-0.5, 0.25, 0.0, 1.0), group_n=3)
    ops.c5_wgrad(xd[:9], dw_b, C, consts=(-0.5, 0.25, 0.0), group_n=3)
    ops.c5_wgrad(xd[9:], dw_b, C, consts=(1.0, 1.0, 1.0), group_n=3)
    assert rel_err(dw_a.cpu(), dw_b.cpu()) < 1e-5
    # ---- pack_pair with repetitions
    a = rnd(3, 3, 32, 32, seed=92)
    one = torch.empty(3, 32, 32, 8, device="cuda", dtype=dt); rep = torch.empty(9, 32, 32, 8, device="cuda", dtype=dt)
    ops.pack_pair(a.cuda(), None, one); ops.pack_pair(a.cuda(), None, rep, reps=3)
    assert all(torch.equal(rep[3 * r:3 * r + 3], one) for r in range(3))
    # ---- act_bwd with the folded dot product
    av = q(F.leaky_relu(rnd(4, 64, 16, 16, seed=93), 0.2), dt); da = rnd(4, 64, 16, 16, seed=94); xv = q(rnd(4, 64, 16, 16, seed=95), dt)
    dz1 = torch.empty(4, 16, 16, 64, device="cuda", dtype=dt); dz2 = torch.empty_like(dz1)
    d1 = torch.zeros(1, device="cuda"); d2 = torch.zeros(1, device="cuda")
    ops.act_bwd(nhwc(da, torch.float32), nhwc(av, dt), dz1, 64, dotx=nhwc(xv, dt), dot_out=d1)
    ops.act_bwd(nhwc(da, torch.float32), nhwc(av, dt), dz2, 64)
    ops.dot_accum(nhwc(xv, dt), nhwc(da, torch.float32), 64, d2)
    assert torch.equal(dz1, dz2) and abs(float(d1) - float(d2)) < 1e-4 * abs(float((xv * da).abs().sum()))
    assert abs(float(d1) - float((xv.double() * da.double()).sum())) < 1e-4 * float((xv * da).abs().sum())
    # ---- the head's re-pack inside the batched conv-weight re-pack
    w = rnd(128, 64, 4, 4, seed=96, scale=0.05).cuda()
    wf = torch.empty(128, 16, 64, device="cuda", dtype=dt); wt = torch.empty(64, 16, 128, device="cuda", dtype=dt)
    wp2 = torch.full((16, C), float("nan"), device="cuda")
    ops.PrepBatch([(w, wf, wt, 128, 64, 64)], ops.code(wf), c5=(w5.cuda(), wp2)).run()
    assert torch.equal(wp2, wp)
    wf1, wt1 = packed_weights(ops, w.cpu(), dt)
    assert torch.equal(wf, wf1) and torch.equal(wt, wt1)
    # ---- striped coefficient / bias sums folded into the batched reduction == ReplicaSum + plain batched reduction
    Cout, Cin, ns, nrep, stride = 128, 64, 5, 8, 256
    slab = rnd(ns, Cout, 16, Cin, seed=97).cuda()
    u = rnd(3, Cout, seed=98).cuda(); v = rnd(3, Cin * 16, seed=99).cuda()
    coef = rnd(3, seed=100).cuda(); reps = rnd(nrep, stride, seed=101).cuda()
    dw1 = torch.zeros(Cout, Cin, 4, 4, device="cuda"); dw2 = torch.zeros_like(dw1)
    db1 = torch.full((Cout,), float("nan"), device="cuda"); db2 = torch.empty(Cout, device="cuda")
    coef2 = coef.clone()
    ops.ReduceBatch([dict(slab=slab, nsplit=ns, dw=dw1, cout=Cout, cin=Cin, cin_real=Cin, coef=coef, u=u, v=v,
                          coef_rep=reps[0, 200:203], bias_rep=reps[0, :Cout], dbias=db1)], nrank=3, nrep=nrep, rep_stride=stride).run()
    ops.ReplicaSum([(reps[0, :Cout], db2, Cout, False), (reps[0, 200:203], coef2, 3, True)], nrep, stride).run()
    ops.ReduceBatch([dict(slab=slab, nsplit=ns, dw=dw2, cout=Cout, cin=Cin, cin_real=Cin, coef=coef2, u=u, v=v)], nrank=3).run()
    torch.cuda.synchronize()
    assert rel_err(dw1.cpu(), dw2.cpu()) < 1e-5 and rel_err(db1.cpu(), db2.cpu()) < 1e-6


def test_pack_interp_gp_norm_unpack(ops):
    B, S = 5, 32
    pred, gt, ref = rnd(B, 3, S, S, seed=70), rnd(B, 3, S, S, seed=71), rnd(B, 3, S, S, seed=72)
    alpha = torch.rand(B, generator=torch.Generator().manual_seed(3))
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        out = torch.empty(3 * B, S, S, 8, device="cuda", dtype=dt)
        ops.pack_pair(pred.cuda(), gt.cuda(), out[:B])
        ops.pack_pair(pred.cuda(), ref.cuda(), out[B:2 * B])
        ops.pack_interp(pred.cuda(), gt.cuda(), ref.cuda(), alpha.cuda(), out[2 * B:])
        a = alpha.view(-1, 1, 1, 1)
        exp = torch.cat([torch.cat([pred, gt], 1), torch.cat([pred, ref], 1),
                         torch.cat([a * pred + (1 - a) * pred, a * gt + (1 - a) * ref], 1)], 0)
        got = nchw(out)
        if dt == torch.float32:
            assert torch.equal(got[:, :6], exp)       # bit-exact: same op-by-op rounding as eager torch
        else:
            assert rel_err(got[:, :6], exp) < (4e-3 if dt == torch.bfloat16 else 5e-4)
        assert float(got[:, 6:].abs().max()) == 0.0
    g = rnd(B, S, S, 8, seed=73).cuda()
    nrm = torch.empty(B, device="cuda"); coef = torch.empty(B, device="cuda"); gp = torch.zeros(1, device="cuda")
    ops.gp_norm(g, B, 1.0, nrm, coef, gp)
    n_ref = torch.sqrt((g.cpu().reshape(B, -1) ** 2).sum(1) + 1e-12)
    assert rel_err(nrm.cpu(), n_ref) < 1e-5
    assert abs(float(gp) - float(((n_ref - 1) ** 2).mean())) < 1e-4 * float(((n_ref - 1) ** 2).mean())
    assert rel_err(coef.cpu(), 2.0 / B * (n_ref - 1) / n_ref) < 1e-5
    y = torch.empty(B, S, S, 8, device="cuda", dtype=torch.bfloat16)
    ops.scale_rows(g, coef, y, B)
    assert rel_err(y.float().cpu(), g.cpu() * coef.cpu().view(-1, 1, 1, 1)) < 1e-2
    ga = torch.empty(B, 3, S, S, device="cuda"); gb = torch.empty(B, 3, S, S, device="cuda")
    ops.unpack_grad(g, ga, gb)
    assert torch.equal(ga.cpu(), g.cpu().permute(0, 3, 1, 2)[:, :3]) and torch.equal(gb.cpu(), g.cpu().permute(0, 3, 1, 2)[:, 3:6])


def test_clip_adam_matches_torch(ops):
    n = 100003
    p0 = rnd(n, seed=80)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=2e-4, betas=(0.5, 0.999))
    p = p0.cuda(); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    state = torch.zeros(ops.ADAM_STATE, device="cuda", dtype=torch.float64)
    for t in range(3):
        g = rnd(n, seed=81 + t, scale=0.05)
        ref.grad = g.clone()
        tn = torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        gd = g.cuda()
        ops.clip_adam(p, gd, m, v, state, 2e-4, 0.5, 0.999, write_clipped=True)
        torch.cuda.synchronize()
        assert abs(float(state[2]) - float(tn)) < 1e-5 * float(tn)
        assert float(state[0]) == t + 1
        assert rel_err(gd.cpu(), ref.grad) < 1e-5
        assert float((p.cpu() - ref.detach()).abs().max()) < 2e-7
    assert rel_err(m.cpu(), opt.state[ref]["exp_avg"]) < 1e-6
    assert rel_err(v.cpu(), opt.state[ref]["exp_avg_sq"]) < 1e-6
    # write_clipped=2: same update, and the bucket comes back zeroed (fused zero_grad); 0 leaves it untouched
    g = rnd(n, seed=90, scale=0.05)
    p2, m2, v2, s2 = p.clone(), m.clone(), v.clone(), state.clone()
    ga, gb = g.cuda(), g.cuda()
    ops.clip_adam(p, ga, m, v, state, 2e-4, 0.5, 0.999, write_clipped=0)
    ops.clip_adam(p2, gb, m2, v2, s2, 2e-4, 0.5, 0.999, write_clipped=2)
    torch.cuda.synchronize()
    assert torch.equal(p, p2) and torch.equal(m, m2) and torch.equal(v, v2)
    assert torch.equal(ga.cpu(), g) and float(gb.abs().max()) == 0.0
    assert float(state[1]) == 0.0 and float(state[6]) == 0.0      # accumulator and counter are reset for the next call
    # state[7] > 0 overrides the launch argument's learning rate (LR scheduler without re-capturing a graph)
    pa, pb = p.clone(), p.clone()
    ma, va, sa = m.clone(), v.clone(), state.clone()
    mb, vb, sb = m.clone(), v.clone(), state.clone()
    sb[7] = 1e-4
    g2 = rnd(n, seed=91, scale=0.05).cuda()
    ops.clip_adam(pa, g2.clone(), ma, va, sa, 1e-4, 0.5, 0.999)
    ops.clip_adam(pb, g2.clone(), mb, vb, sb, 2e-4, 0.5, 0.999)
    torch.cuda.synchronize()
    assert torch.equal(pa, pb)
    # grad_scale: the 1/world of a data-parallel SUM all-reduce folded in == scaling the gradient first
    pa, pb = p.clone(), p.clone()
    ma, va, sa = m.clone(), v.clone(), state.clone()
    mb, vb, sb = m.clone(), v.clone(), state.clone()
    g3 = rnd(n, seed=92, scale=40.0).cuda()                      # large enough that the clip is active on the summed gradient
    ga, gb = g3.clone(), g3 * 0.25
    ops.clip_adam(pa, ga, ma, va, sa, 2e-4, 0.5, 0.999, write_clipped=1, grad_scale=0.25)
    ops.clip_adam(pb, gb, mb, vb, sb, 2e-4, 0.5, 0.999, write_clipped=1)
    torch.cuda.synchronize()
    assert rel_err(pa.cpu(), pb.cpu()) < 1e-6 and rel_err(ga.cpu(), gb.cpu()) < 1e-6
    assert abs(float(sa[2]) - float(sb[2])) < 1e-6 * float(sb[2]) and abs(float(sa[3]) - float(sb[3])) < 1e-6


def test_generator_head_and_eiou(ops):
    from oracle import manual_step as M
    from conftest import load_golden
    B, S = 21, 16
    x = rnd(B, 64, S, S, seed=90)
    w, b = rnd(4, 64, seed=91, scale=0.125), rnd(4, seed=92, scale=0.125)
    pooled = torch.empty(B, 64, device="cuda"); traw = torch.empty(B, 4, device="cuda"); delta = torch.empty(B, 4, device="cuda")
    ops.pool_fc_tanh_fwd(nhwc(x, torch.float32), w.cuda(), b.cuda(), 0.3, pooled, traw, delta)
    pr = x.mean(dim=(2, 3)); tr = torch.tanh(pr @ w.t() + b)
    assert rel_err(pooled.cpu(), pr) < 1e-5 and rel_err(delta.cpu(), tr * 0.3) < 1e-5
    gd = rnd(B, 4, seed=93)
    dw = torch.zeros(4, 64, device="cuda"); db = torch.zeros(4, device="cuda"); dab = torch.empty(B, 64, device="cuda")
    ops.head_bwd(gd.cuda(), traw, pooled, w.cuda(), 0.3, B, S * S, dw, db, dab)
    dy = gd * 0.3 * (1 - tr * tr)
    assert rel_err(dw.cpu(), dy.t() @ pr) < 1e-5 and rel_err(db.cpu(), dy.sum(0)) < 1e-5
    assert rel_err(dab.cpu(), (dy @ w) / (S * S)) < 1e-5
    # EIoU + box math against the reference's known-answer vectors
    fix = load_golden("loss_vectors")
    bbox, dl = torch.from_numpy(fix["bbox"]), torch.from_numpy(fix["delta"])
    dtrue = torch.from_numpy(load_pkg("synth").normal("kv.dt", 7, (16, 4), 0.1))
    g = torch.empty(16, 4, device="cuda"); cal = torch.empty(16, 4, device="cuda"); acc = torch.full((1,), 7.0, device="cuda")   # (the loss is STORED)
    ops.eiou_fwd_bwd(bbox.cuda(), dl.cuda(), dtrue.cuda(), 1.0, g, cal, acc)
    assert abs(1.0 + float(acc) - float(fix["hybrid_total"])) < 1e-5
    assert rel_err(cal.cpu(), fix["apply_train"]) < 1e-5
    assert rel_err(g.cpu(), fix["hybrid_grad_delta"]) < 1e-4


def test_uniform_gen(ops):
    u = torch.empty(1 << 18, device="cuda")
    ops.uniform_gen(u, 77)
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 5e-3 and abs(float(u.var()) - 1.0 / 12) < 2e-3
    ctr = torch.tensor([5.0], device="cuda", dtype=torch.float64)
    u2 = torch.empty_like(u); u3 = torch.empty_like(u)
    ops.uniform_gen(u2, 77, ctr); ops.uniform_gen(u3, 77, ctr)
    assert torch.equal(u2, u3) and not torch.equal(u, u2)
    assert abs(float(torch.corrcoef(torch.stack([u, u2]))[0, 1])) < 1e-2


def test_dropout_mask_gen(ops):
    m = torch.empty(1 << 20, device="cuda", dtype=torch.uint8)
    ops.dropout_mask_gen(m, 1234)
    frac = float(m.float().mean())
    assert set(m.unique().tolist()) <= {0, 1} and abs(frac - 0.5) < 5e-3
    m2 = torch.empty_like(m)
    ops.dropout_mask_gen(m2, 1235)
    assert abs(float((m == m2).float().mean()) - 0.5) < 5e-3
    # eight bytes come from one hash: neighbouring bytes must still be independent
    assert abs(float((m[0::2] == m[1::2]).float().mean()) - 0.5) < 5e-3
    # the device-side counter re-keys the draw (graph replays read it at run time); same key -> same mask
    ctr = torch.tensor([3.0], device="cuda", dtype=torch.float64)
    m3 = torch.empty_like(m); m4 = torch.empty_like(m)
    ops.dropout_mask_gen(m3, 1234, ctr); ops.dropout_mask_gen(m4, 1234, ctr)
    assert torch.equal(m3, m4) and abs(float((m == m3).float().mean()) - 0.5) < 5e-3
    # ragged length: the n % 8 tail is filled too
    t = torch.full((8 * 1000 + 5,), 7, device="cuda", dtype=torch.uint8)
    ops.dropout_mask_gen(t, 99)
    assert set(t.unique().tolist()) <= {0, 1}


def test_striped_bias_and_cdot_sums(ops):
    """dbias / cdot striped over replicas (nrep) + gcssl_sum_replicas == the single-address form."""
    N, H, C = 24, 16, 64
    da = rnd(N, H, H, C, seed=70).cuda()
    a = rnd(N, H, H, C, seed=71).cuda().bfloat16()
    bias = rnd(C, seed=72, scale=0.1).cuda()
    gs = torch.tensor([1.5, 0.5, 2.0], device="cuda")
    dz1 = torch.empty(N, H, H, C, device="cuda", dtype=torch.bfloat16); dz2 = torch.empty_like(dz1)
    db1 = torch.zeros(C, device="cuda"); cd1 = torch.zeros(3, device="cuda")
    ops.act_bwd(da, a, dz1, C, gscale=gs, group_n=8, bias=bias, dbias=db1, cdot=cd1)
    nrep, stride = 8, 128
    rep = torch.zeros(nrep, stride, device="cuda")
    ops.act_bwd(da, a, dz2, C, gscale=gs, group_n=8, bias=bias, dbias=rep[0, :C], cdot=rep[0, 100:103], nrep=nrep, rep_stride=stride)
    db2 = torch.full((C,), 7.0, device="cuda"); cd2 = torch.full((3,), 1.0, device="cuda")
    ops.ReplicaSum([(rep[0, :C], db2, C, False), (rep[0, 100:103], cd2, 3, True)], nrep, stride).run()
    torch.cuda.synchronize()
    assert torch.equal(dz1, dz2)
    assert int((rep[:, :C].abs().sum(1) > 0).sum()) > 1                     # the sums really are spread over replicas
    assert rel_err(db2.cpu(), db1.cpu()) < 1e-5 and rel_err((cd2 - 1.0).cpu(), cd1.cpu()) < 1e-4


@pytest.mark.parametrize("kind", ["fwd", "dgrad"])
def test_split_k_slab_mode_matches_atomic_mode(ops, kind):
    """A K-split conv that stores per-split slabs + in_act_fwd(nslab) == the atomic form + plain in_act_fwd."""
    dt = torch.bfloat16
    N, Hi, Cin, Cout = 64, 8, 256, 512                      # G.down4-like: small M, long K -> the dispatcher splits
    w = rnd(Cout, Cin, 4, 4, seed=91, scale=0.05)
    wf, wt = packed_weights(ops, w, dt)
    if kind == "fwd":
        x = nhwc(q(rnd(N, Cin, Hi, Hi, seed=92), dt), dt)
        ks = ops.conv_splits("fwd", ops.code(x), N, Hi, Cin, Cout)
        shape, C = (N, Hi // 2, Hi // 2, Cout), Cout
        run = lambda y, st: ops.conv_fwd(x, wf, y, Cin, Cout, split_stride=st)
    else:
        dy = nhwc(q(rnd(N, Cout, Hi // 2, Hi // 2, seed=93), dt), dt)
        ks = ops.conv_splits("dgrad", ops.code(dy), N, Hi, Cin, Cout)
        shape, C = (N, Hi, Hi, Cin), Cin
        run = lambda y, st: ops.conv_dgrad(dy, wt, y, Cin, Cout, split_stride=st)
    assert ks > 1
    z_atomic = torch.empty(shape, device="cuda")
    run(z_atomic, 0)
    slabs = torch.full((ks,) + shape, float("nan"), device="cuda")
    run(slabs[0], slabs.stride(0))
    torch.cuda.synchronize()
    assert rel_err(slabs.sum(0).cpu(), z_atomic.cpu()) < 1e-5
    a1 = torch.empty(shape, device="cuda", dtype=dt); a2 = torch.empty_like(a1)
    m1 = torch.empty(N, C, device="cuda"); r1 = torch.empty(N, C, device="cuda"); m2 = torch.empty_like(m1); r2 = torch.empty_like(r1)
    ops.in_act_fwd(z_atomic, a1, m1, r1, C, 1)
    ops.in_act_fwd(slabs[0], a2, m2, r2, C, 1, nslab=ks, slab_stride=slabs.stride(0))
    torch.cuda.synchronize()
    assert rel_err(slabs[0].cpu(), z_atomic.cpu()) < 1e-5                  # the total was written back to slab 0
    assert rel_err(a2.float().cpu(), a1.float().cpu()) < 1e-2 and rel_err(r2.cpu(), r1.cpu()) < 1e-4


@pytest.mark.parametrize("H", [2, 4, 8, 16])
def test_norm_backward_adds_incoming_gradient_slabs(ops, H):
    """in_act_bwd(da_nslab) / in_dbl_bwd(q_nslab): the incoming gradient given as K-split slabs == given as their sum."""
    dt = torch.bfloat16
    N, C, ks = 12, 64, 3
    z = nhwc(rnd(N, C, H, H, seed=70), torch.float32)
    parts = torch.stack([nhwc(rnd(N, C, H, H, seed=71 + k), torch.float32) for k in range(ks)])      # [ks][N][H][W][C]
    total = parts.sum(0)
    a = torch.empty(N, H, H, C, device="cuda", dtype=dt)
    mean = torch.empty(N, C, device="cuda"); rstd = torch.empty(N, C, device="cuda")
    ops.in_act_fwd(z, a, mean, rstd, C, 1)
    ws = torch.empty(2 * N * C, device="cuda")
    dz1 = torch.empty(N, H, H, C, device="cuda", dtype=dt); dz2 = torch.empty_like(dz1)
    ops.in_act_bwd(z, mean, rstd, dz1, C, 1, da=total, ws=ws)
    slabs = parts.clone()
    ops.in_act_bwd(z, mean, rstd, dz2, C, 1, da=slabs[0], ws=ws, da_nslab=ks, da_slab_stride=slabs.stride(0))
    torch.cuda.synchronize()
    assert rel_err(dz2.float().cpu(), dz1.float().cpu()) < 1e-2
    assert rel_err(slabs[0].cpu(), total.cpu()) < 1e-6                       # the total is left in slab 0 (read again as gb_a)
    if H <= 8:                                                              # the double backward sums slabs on maps up to 8x8
        gb_a = nhwc(rnd(N, C, H, H, seed=80), torch.float32)
        gzs = nhwc(q(rnd(N, C, H, H, seed=81), dt), dt)
        outs = []
        for qz, kw in ((total, {}), (parts[0], dict(q_nslab=ks, q_slab_stride=parts.stride(0)))):
            gt_a = torch.empty(N, H, H, C, device="cuda", dtype=dt); zt = torch.empty(N, H, H, C, device="cuda")
            cd = torch.zeros(1, device="cuda")
            ops.in_dbl_bwd(gb_a, qz, gzs, z, mean, rstd, gt_a, zt, C, 1, cdot=cd, **kw)
            outs.append((gt_a.float().cpu(), zt.cpu(), cd.cpu()))
        torch.cuda.synchronize()
        for (u, v), tol in zip(zip(*outs), (1e-2, 1e-4, 1e-3)):            # gt_a is bf16; zt fp32; cdot a long atomic sum
            assert rel_err(v, u) < tol
    else:
        with pytest.raises(RuntimeError):
            ops.in_dbl_bwd(total, parts[0], None, z, mean, rstd, a, torch.empty(N, H, H, C, device="cuda"), C, 1,
                           q_nslab=ks, q_slab_stride=parts.stride(0))


def test_pack_fake_interp_equals_the_two_separate_packs(ops):
    B, S = 5, 32
    pred, gt, ref = rnd(B, 3, S, S, seed=60).cuda(), rnd(B, 3, S, S, seed=61).cuda(), rnd(B, 3, S, S, seed=62).cuda()
    alpha = torch.rand(B, device="cuda")
    for dt in (torch.float32, torch.bfloat16):
        f1 = torch.empty(B, S, S, 8, device="cuda", dtype=dt); i1 = torch.empty_like(f1)
        f2 = torch.empty_like(f1); i2 = torch.empty_like(f1)
        ops.pack_pair(pred, ref, f1); ops.pack_interp(pred, gt, ref, alpha, i1)
        ops.pack_fake_interp(pred, gt, ref, alpha, f2, i2)
        torch.cuda.synchronize()
        # (the mix may differ in the last bit between the two kernels: the compiler fuses a*b+c differently)
        assert torch.equal(f1, f2) and float((i1.float() - i2.float()).abs().max()) <= (1e-6 if dt == torch.float32 else 8e-3)
    # device-drawn alpha: a convex combination per sample with the same alpha for all pixels, re-keyed by the counter
    ctr = torch.tensor([2.0], device="cuda", dtype=torch.float64)
    f3 = torch.empty(B, S, S, 8, device="cuda"); i3 = torch.empty_like(f3); i4 = torch.empty_like(f3)
    ops.pack_fake_interp(pred, gt, ref, None, f3, i3, seed=11, counter=ctr)
    ops.pack_fake_interp(pred, gt, ref, None, f3, i4, seed=11, counter=ctr)
    torch.cuda.synchronize()
    assert torch.equal(i3, i4)
    g, r, m = gt.permute(0, 2, 3, 1), ref.permute(0, 2, 3, 1), i3[..., 3:6]
    al = ((m - r) / (g - r)).reshape(B, -1)
    assert float(al.min()) >= -1e-3 and float(al.max()) <= 1 + 1e-3
    assert float((al - al.median(1, keepdim=True).values).abs().median()) < 1e-3


@pytest.mark.parametrize("dt,tol", DTS[1:])
@pytest.mark.parametrize("N,H,K,z_n0", [(3, 16, 128, 1),       # G.up4 at 32x32 images: one sample per pixel block
                                        (8, 8, 256, 4),        # G.up3: four samples per pixel block
                                        (12, 8, 64, 0),        # shortest K (two 32-channel chunks)
                                        (520, 16, 128, 500)])  # more pixel blocks than CUs: the persistent loop, 2-3 blocks per workgroup
def test_fused_convT_instnorm_relu(ops, dt, tol, N, H, K, z_n0):
    """csrc/convt_fused.hip against ConvTranspose2d -> InstanceNorm2d -> ReLU (-> spatial sums) of torch CPU (fp64)."""
    x = q(rnd(N, K, H, H, seed=101), dt)
    w = rnd(K, 64, 4, 4, seed=102, scale=0.05)                  # ConvTranspose2d weight [Cin_T = K][Cout_T = 64][4][4]
    # as a conv: Cout = K (its input is our output), Cin = 64; packed dgrad operand Wt[64][16][K]
    _, wt = packed_weights(ops, w, dt)
    xd = nhwc(x, dt)
    wide = torch.zeros(N, 2 * H, 2 * H, 128, device="cuda", dtype=dt)       # activation goes into a concat slice
    a = wide[..., :64]
    z32 = torch.full((N, 2 * H, 2 * H, 64), float("nan"), device="cuda")
    mean = torch.empty(N, 64, device="cuda"); rstd = torch.empty(N, 64, device="cuda")
    pool = torch.full((N, 64), float("nan"), device="cuda")
    ops.convt_in_relu_fwd(xd, wt, mean, rstd, K, z32=z32, z_n0=z_n0, a=a, pool=pool)
    torch.cuda.synchronize()
    z = F.conv_transpose2d(x.double(), q(w, dt).double(), None, 2, 1)
    mu = z.mean(dim=(2, 3)); var = z.var(dim=(2, 3), unbiased=False)
    r = 1.0 / torch.sqrt(var + 1e-5)
    act = torch.relu((z - mu[:, :, None, None]) * r[:, :, None, None])
    assert rel_err(mean.cpu(), mu) < 1e-4 and rel_err(rstd.cpu(), r) < 1e-4
    assert rel_err(nchw(z32[z_n0:]), z[z_n0:]) < TOL_ACC
    assert bool(torch.isnan(z32[:z_n0]).all())                   # samples without a backward pass: nothing written
    assert rel_err(nchw(a), act) < tol
    assert float(wide[..., 64:].abs().max()) == 0.0
    assert rel_err(pool.cpu(), act.sum(dim=(2, 3))) < 1e-4
    # outputs are optional: statistics + pool only (what G.up4 needs for the samples of the no-grad forwards)
    mean2 = torch.empty_like(mean); rstd2 = torch.empty_like(rstd); pool2 = torch.empty_like(pool)
    cnt = torch.full((N, 64), float("nan"), device="cuda")
    ops.convt_in_relu_fwd(xd, wt, mean2, rstd2, K, pool=pool2, cnt=cnt)
    torch.cuda.synchronize()
    assert torch.equal(mean2, mean) and torch.equal(rstd2, rstd) and torch.equal(pool2, pool)
    # cnt = number of positive normalised outputs per (n, c); values within 1e-5 of zero may land on either side
    xh = (z - mu[:, :, None, None]) * r[:, :, None, None]
    lo, hi = (xh > 1e-5).sum(dim=(2, 3)).double(), (xh > -1e-5).sum(dim=(2, 3)).double()
    c = cnt.cpu().double()
    assert bool(((c >= lo) & (c <= hi)).all()) and bool((c == c.round()).all())
    if H == 16:
        # ... and with the pooled sums it replaces the statistics pass of the InstanceNorm backward when the incoming gradient
        # is a per-(n, c) constant (the generator head's average pool): maps of 32 x 32 = 1024 pixels
        dab = rnd(N - z_n0, 64, seed=103).cuda()
        zb = z32[z_n0:].contiguous()
        ws = torch.zeros(2 * (N - z_n0) * 64, device="cuda")
        dz_a = torch.empty(N - z_n0, 2 * H, 2 * H, 64, device="cuda", dtype=dt); dz_b = torch.empty_like(dz_a)
        ops.in_act_bwd(zb, mean[z_n0:], rstd[z_n0:], dz_a, 64, 2, da_bcast=dab, ws=ws)
        ops.in_act_bwd(zb, mean[z_n0:], rstd[z_n0:], dz_b, 64, 2, da_bcast=dab, presum_cnt=cnt[z_n0:], presum_pos=pool[z_n0:],
                       presum_pos_scale=1.0)
        torch.cuda.synchronize()
        assert rel_err(dz_b.float().cpu(), dz_a.float().cpu()) < (2e-5 if dt == torch.float32 else tol)
