"""Standalone launcher of ONE conv configuration (for rocprofv3 --pmc / timing experiments).
usage: python tools/conv_bench.py {fwd|fwd16|fwd_in|dgrad|wgrad} N Hi Cin Cout [bf16|fp16|fp32|fp16x3|bf16x3] [reps]
(fwd_in: the one-launch conv + InstanceNorm + LeakyReLU form, 16-bit activation + fp32 statistics out)"""
import importlib, sys, time, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("gan-calibrated-semi-supervised-learning_amd.ops")
kind, N, Hi, Cin, Cout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
mode = sys.argv[6] if len(sys.argv) > 6 else "bf16"
_lib = importlib.import_module("gan-calibrated-semi-supervised-learning_amd._lib")
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32, "fp16x3": torch.float32, "bf16x3": torch.float32}[mode]
kw = {"dt": _lib.mma_code(mode)} if mode in _lib.SPLIT_MODES else {}      # split-precision modes: fp32 tensors, 3 x 16-bit MFMA
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 20
x = (torch.rand(N, Hi, Hi, Cin, device="cuda") * 2 - 1).to(dt)
dy = (torch.rand(N, Hi // 2, Hi // 2, Cout, device="cuda") * 2 - 1).to(dt)
w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.05
wf = torch.empty(Cout, 16, Cin, device="cuda", dtype=dt); wt = torch.empty(Cin, 16, Cout, device="cuda", dtype=dt)
ops.prep_conv_weight(w, wf, wt, Cout, Cin, Cin, kw.get('dt', ops.code(wf)))
y = torch.empty(N, Hi // 2, Hi // 2, Cout, device="cuda", dtype=torch.float32)
dx = torch.empty(N, Hi, Hi, Cin, device="cuda", dtype=torch.float32)
ns = ops.wgrad_splits(N, Hi, Hi, Cin, Cout)
slab = torch.empty(ns, Cout, 16, Cin, device="cuda")
a16 = torch.empty(N, Hi // 2, Hi // 2, Cout, device="cuda", dtype=dt)
mean = torch.empty(N, Cout, device="cuda"); rstd = torch.empty(N, Cout, device="cuda")
def run():
    if kind == "fwd": ops.conv_fwd(x, wf, y, Cin, Cout, **kw)
    elif kind == "fwd16": ops.conv_fwd(x, wf, a16, Cin, Cout, act=1)          # 16-bit output + LeakyReLU (the norm-less first layers)
    elif kind == "fwd_in": ops.conv_in_act_fwd(x, wf, a16, mean, rstd, Cin, Cout)
    elif kind == "dgrad": ops.conv_dgrad(dy, wt, dx, Cin, Cout, **kw)
    else: ops.conv_wgrad(x, dy, slab, Cin, Cout, **kw)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = 2.0 * N * (Hi // 2) ** 2 * Cout * 16 * Cin
print(f"{kind} N={N} Hi={Hi} Cin={Cin} Cout={Cout} {mode}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF/s  {ops.last_kernel()}")
