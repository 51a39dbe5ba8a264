"""Per-kernel parity of the GeneratorSimpleRegressor pieces (cgan/models.py:147-216) against torch CPU fp32: the 3x3
stride-1 convolutions (forward, data gradient through the rotated pack, weight gradient + reduce), MaxPool2d(2,2) forward /
backward, AdaptiveAvgPool2d(1) and the regressor head.  Tolerances as in test_kernels_gpu.py: fp32 2e-5, bf16 2e-2."""
import pytest
import torch
import torch.nn.functional as F
from torch.nn.grad import conv2d_weight

from conftest import load_pkg, rel_err
from test_kernels_gpu import nchw, nhwc, q, rnd

pytestmark = pytest.mark.gpu

DTS = [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)]


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return load_pkg("ops")


def packed3(ops, w, dt, cinp=None, with_t=True):
    cout, cin = w.shape[:2]
    cinp = cinp or cin
    wf = torch.full((cout, ops.conv3_wk(cinp)), float("nan"), device="cuda", dtype=dt)
    wt = torch.full((cin, ops.conv3_wk(cout)), float("nan"), device="cuda", dtype=dt) if with_t else None
    ops.Prep3Batch([(w.cuda(), wf, wt, cout, cin, cinp)], ops.code(wf)).run()
    return wf, wt


C3_CASES = [  # N, H, Cin(real), CinP, Cout
    (3, 8, 64, 64, 128),
    (2, 16, 3, 8, 64),        # first layer: 3 channels padded to 8, K axis padded to 16 taps
    (2, 4, 256, 256, 512),
    (70, 32, 64, 64, 64),     # 560 tiles of 128x64 -> the 128-row tile
    (5, 2, 512, 512, 512),    # 2x2 maps (S = 16 ... or the last block at 32x32 is 4x4)
    (201, 32, 64, 64, 64),    # 1608 tiles (ragged last one) > resident workgroups: the persistent LDS-DMA kernel in bf16
    (90, 32, 3, 8, 64),       # 720 tiles of the 8-channel first layer
]


@pytest.mark.parametrize("dt,tol", DTS)
@pytest.mark.parametrize("case", C3_CASES)
def test_conv3x3_forward_dgrad_wgrad(ops, case, dt, tol):
    N, H, Cin, CinP, Cout = case
    x = q(rnd(N, Cin, H, H, seed=1), dt)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=0.05)
    b = rnd(Cout, seed=3, scale=0.1)
    dy = q(rnd(N, Cout, H, H, seed=4), dt)
    wf, wt = packed3(ops, w, dt, CinP, with_t=Cin >= 64)
    wq = q(w, dt)
    xd = nhwc(x, dt, CinP)
    # forward (+bias), fp32 output
    y = torch.empty(N, H, H, Cout, device="cuda", dtype=torch.float32)
    ops.conv3_fwd(xd, wf, y, CinP, Cout, bias=b.cuda())
    assert rel_err(nchw(y), F.conv2d(x, wq, b, padding=1)) < tol
    # forward into the compute dtype, written into a channel slice of a wider buffer
    wide = torch.zeros(N, H, H, 2 * Cout, device="cuda", dtype=dt)
    ops.conv3_fwd(xd, wf, wide[..., Cout:], CinP, Cout)
    assert rel_err(nchw(wide[..., Cout:]), F.conv2d(x, wq, None, padding=1)) < max(tol, 8e-3 if dt == torch.bfloat16 else 0)
    assert float(wide[..., :Cout].abs().max()) == 0.0
    dyd = nhwc(dy, dt)
    if Cin >= 64:
        dx = torch.empty(N, H, H, Cin, device="cuda", dtype=torch.float32)
        ops.conv3_fwd(dyd, wt, dx, Cout, Cin)
        assert rel_err(nchw(dx), F.conv_transpose2d(dy, wq, padding=1)) < tol
    # weight gradient: split-K slabs -> reduce into the PyTorch layout
    ns = ops.conv3_wgrad_splits(N, H, CinP, Cout)
    slab = torch.full((ns, Cout, 16, CinP), float("nan"), device="cuda")
    ops.conv3_wgrad(xd, dyd, slab, CinP, Cout)
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
    ops.Reduce3Batch([dict(slab=slab, nsplit=ns, dw=dw, cout=Cout, cin=CinP, cin_real=Cin)]).run()
    torch.cuda.synchronize()
    ref = conv2d_weight(x, (Cout, Cin, 3, 3), dy, padding=1)
    assert rel_err(dw.cpu(), ref) < tol


@pytest.mark.parametrize("dt,tol", DTS)
def test_maxpool_forward_backward(ops, dt, tol):
    N, H, C = 3, 8, 64
    a = q(rnd(N, C, H, H, seed=5), dt)
    a[0, :, :2, :2] = 0.25                                   # a window of ties: the first element takes the gradient
    a = torch.relu(a)                                        # many zero ties, as after the ReLU in front of every pool
    ad = nhwc(a, dt)
    o = torch.empty(N, H // 2, H // 2, C, device="cuda", dtype=dt)
    ops.maxpool2_fwd(ad, o, C)
    at = a.clone().requires_grad_(True)
    ref, idx = F.max_pool2d(at, 2, 2, return_indices=True)
    assert torch.equal(nchw(o), ref.detach())
    g = rnd(N, C, H // 2, H // 2, seed=6)
    ref.backward(g)
    da = torch.full((N, H, H, C), float("nan"), device="cuda")
    ops.maxpool2_bwd(ad, nhwc(g, torch.float32), da, C)
    assert torch.equal(nchw(da), at.grad)
    # AdaptiveAvgPool2d(1) backward folded in: a per-sample vector broadcast over the pooled pixels
    gv = rnd(N, C, seed=7)
    at.grad = None
    F.max_pool2d(at, 2, 2).mean(dim=(2, 3)).backward(gv)
    ops.maxpool2_bwd(ad, gv.cuda(), da, C, bcast_scale=1.0 / ((H // 2) ** 2))
    assert rel_err(nchw(da), at.grad) < 1e-6
    feat = torch.empty(N, C, device="cuda")
    ops.avgpool_fwd(o, feat, C)
    assert rel_err(feat.cpu(), ref.detach().mean(dim=(2, 3))) < 1e-6


@pytest.mark.parametrize("train", [True, False])
def test_regressor_head_forward_backward(ops, train):
    N = 7                                                    # not a multiple of the 4 samples per workgroup
    feat = rnd(N, 512, seed=8).requires_grad_(True)
    w1, b1 = rnd(256, 512, seed=9, scale=0.05).requires_grad_(True), rnd(256, seed=10, scale=0.05).requires_grad_(True)
    w2, b2 = rnd(64, 256, seed=11, scale=0.08).requires_grad_(True), rnd(64, seed=12, scale=0.05).requires_grad_(True)
    w3, b3 = rnd(4, 64, seed=13, scale=0.15).requires_grad_(True), rnd(4, seed=14, scale=0.05).requires_grad_(True)
    m1 = (rnd(N, 256, seed=15) > 0).to(torch.uint8)
    m2 = (rnd(N, 64, seed=16) > 0).to(torch.uint8)
    scale = 0.3
    h1 = torch.relu(feat @ w1.t() + b1)
    h2 = torch.relu((h1 * m1 * 2 if train else h1) @ w2.t() + b2)
    t = torch.tanh((h2 * m2 * 2 if train else h2) @ w3.t() + b3)
    delta = t * scale
    gd = rnd(N, 4, seed=17)
    delta.backward(gd)
    dev = lambda x: x.detach().cuda()
    H1, H2 = torch.empty(N, 256, device="cuda"), torch.empty(N, 64, device="cuda")
    T, D = torch.empty(N, 4, device="cuda"), torch.empty(N, 4, device="cuda")
    ops.mlp_head_fwd(dev(feat), dev(w1).t().contiguous(), dev(b1), dev(w2).t().contiguous(), dev(b2), dev(w3), dev(b3), scale,
                     H1, H2, T, D,
                     m1=m1.cuda() if train else None, m2=m2.cuda() if train else None)
    assert rel_err(D.cpu(), delta.detach()) < 1e-5 and rel_err(T.cpu(), t.detach()) < 1e-5
    dp1, dp2, dp3 = torch.empty(N, 256, device="cuda"), torch.empty(N, 64, device="cuda"), torch.empty(N, 4, device="cuda")
    dfeat = torch.empty(N, 512, device="cuda")
    gw = [torch.full_like(dev(p), float("nan")) for p in (w1, b1, w2, b2, w3, b3)]
    ops.mlp_head_bwd(gd.cuda(), T, H1, H2, dev(feat), dev(w1), dev(w2), dev(w3), scale, train, dp1, dp2, dp3, dfeat, *gw)
    torch.cuda.synchronize()
    assert rel_err(dfeat.cpu(), feat.grad) < 1e-5
    for got, p in zip(gw, (w1, b1, w2, b2, w3, b3)):
        assert rel_err(got.cpu(), p.grad) < 1e-5
