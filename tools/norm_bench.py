"""Where does the instance-norm backward spend its time?  usage: python tools/norm_bench.py [fp16|bf16]
Times gcssl_in_act_bwd / gcssl_in_act_fwd at the critic's and the generator's bench shapes with the optional terms switched
on one by one, and prints the bytes each variant moves."""
import importlib, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("gan-calibrated-semi-supervised-learning_amd.ops")
dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "fp16"]


def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def case(N, H, C, B):
    f32 = dict(device="cuda", dtype=torch.float32)
    z = torch.randn(N, H, H, C, **f32); da = torch.randn(N, H, H, C, **f32); da2 = torch.randn(N, H, H, C, **f32)
    zt = torch.randn(N - 2 * B if N > 2 * B else N, H, H, C, **f32)
    a = torch.empty(N, H, H, C, device="cuda", dtype=dt); dzs = torch.empty_like(a)
    mean = torch.empty(N, C, **f32); rstd = torch.empty(N, C, **f32)
    ws = torch.zeros(2 * N * C, **f32)
    gs = torch.ones(3, **f32); bias = torch.zeros(C, **f32)
    rep = torch.zeros(32, 1024, **f32)
    elems = N * H * H * C
    us = t(lambda: ops.in_act_fwd(z, a, mean, rstd, C, 1))
    print(f"[{N}x{H}x{H}x{C}] fwd                {us:6.1f} us  {elems * 6 / us / 1e6:6.2f} TB/s")
    for name, kw, bpe in (
            ("bwd plain", dict(da=da), 10),
            ("bwd +gscale", dict(da=da, gscale=gs, group_n=B), 10),
            ("bwd +da2", dict(da=da, da2=da2), 14),
            ("bwd +zt", dict(da=da, zt=zt, zt_n0=N - zt.shape[0], gscale=gs, group_n=B), 10 + 4 * zt.shape[0] / N),
            ("bwd +zt+bias/dbias/cdot r32", dict(da=da, zt=zt, zt_n0=N - zt.shape[0], gscale=gs, group_n=B, bias=bias,
                                                 dbias=rep[0, :C], cdot=rep[0, 960:963], nrep=32, rep_stride=1024), 10 + 4 * zt.shape[0] / N),
            ("bwd +bias/cdot only r32", dict(da=da, gscale=gs, group_n=B, bias=bias, cdot=rep[0, 960:963], nrep=32, rep_stride=1024), 10),
            ("bwd +bias/dbias only r32", dict(da=da, gscale=gs, group_n=B, bias=bias, dbias=rep[0, :C], nrep=32, rep_stride=1024), 10)):
        us = t(lambda: ops.in_act_bwd(z, mean, rstd, dzs, C, 1, ws=ws, **kw))
        print(f"[{N}x{H}x{H}x{C}] {name:28s} {us:6.1f} us  {elems * bpe / us / 1e6:6.2f} TB/s")


case(768, 8, 128, 256)      # D.c2
case(768, 4, 256, 256)      # D.c3
case(768, 2, 512, 256)      # D.c4
case(256, 8, 128, 256)      # G.down2 / up-side
case(256, 16, 64, 256)      # G.up3 (H*W = 256: 16 rows per lane)
