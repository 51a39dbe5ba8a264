#!/usr/bin/env python3
"""A measured bound on what Winograd F(2x2,3x3) could buy the simple generator's 3x3 layers.
F(2x2,3x3) replaces the 9*Cin-deep contraction per output pixel by 16 contractions of depth Cin per 2x2 output tile, i.e.
4*Cin per output pixel: 2.25x fewer MACs AND 2.25x fewer operand bytes through LDS.  The direct kernel run with its K loop cut
to 4 of its 9 steps (GCSSL_KCAP=4; results are garbage) does exactly that volume of fills + MFMAs with NO transform work, no
16 separate accumulator sets and no output transform: a lower bound on any Winograd form of this kernel.  GCSSL_KCAP=2 is
(almost) the launch skeleton.  Run on the GPU box, one process per setting:  GCSSL_KCAP=4 python tools/archive/winograd_bound.py"""
import importlib, os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("gan-calibrated-semi-supervised-learning_amd.ops")
N, H, C = 768, 32, 64
dt = torch.float16
x = (torch.rand(N, H, H, C, device="cuda") * 2 - 1).to(dt)
w = torch.randn(C, C, 3, 3, device="cuda") * 0.05
wk = ops.conv3_wk(C)
wf = torch.empty(C, wk, device="cuda", dtype=dt)
ops.Prep3Batch([(w, wf, None, C, C, C)], ops.code(wf)).run()
y = torch.empty(N, H, H, C, device="cuda", dtype=torch.float32)
for _ in range(5):
    ops.conv3_fwd(x, wf, y, C, C)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    ops.conv3_fwd(x, wf, y, C, C)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
fl = 2.0 * N * H * H * C * 9 * C
print(f"GCSSL_KCAP={os.environ.get('GCSSL_KCAP', '0 (all 9 K steps)')}: conv3x3 {C}->{C} at {H}x{H}, n={N}: {us:.1f} us per launch "
      f"({fl / us / 1e6:.0f} TFLOP/s if it were the whole contraction); kernel {ops.last_kernel()}")
