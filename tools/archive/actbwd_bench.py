"""Is the batched activation backward bound by its bias/SN atomics?  usage: python tools/archive/actbwd_bench.py"""
import importlib, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("gan-calibrated-semi-supervised-learning_amd.ops")
N, H, C = 768, 16, 64
da = torch.randn(N, H, H, C, device="cuda"); a = torch.randn(N, H, H, C, device="cuda").bfloat16()
dzs = torch.empty(N, H, H, C, device="cuda", dtype=torch.bfloat16)
gs = torch.ones(3, device="cuda"); bias = torch.zeros(C, device="cuda")
dbias = torch.zeros(C, device="cuda"); cdot = torch.zeros(3, device="cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("plain            %.1f us" % t(lambda: ops.act_bwd(da, a, dzs, C)))
print("gscale           %.1f us" % t(lambda: ops.act_bwd(da, a, dzs, C, gscale=gs, group_n=256)))
print("+dbias           %.1f us" % t(lambda: ops.act_bwd(da, a, dzs, C, gscale=gs, group_n=256, bias=bias, dbias=dbias)))
print("+dbias+cdot      %.1f us" % t(lambda: ops.act_bwd(da, a, dzs, C, gscale=gs, group_n=256, bias=bias, dbias=dbias, cdot=cdot)))
print("+cdot only       %.1f us" % t(lambda: ops.act_bwd(da, a, dzs, C, gscale=gs, group_n=256, bias=bias, cdot=cdot)))
