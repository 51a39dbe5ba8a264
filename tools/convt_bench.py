#!/usr/bin/env python3
"""Stand-alone timing of the fused transposed-conv kernel (csrc/convt_fused.hip) against the unfused pair (LDS-DMA dgrad-form
conv + InstanceNorm pass):  python tools/convt_bench.py [N H K [bf16|fp16]]"""
import importlib
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = "gan-calibrated-semi-supervised-learning_amd"
ops = importlib.import_module(PKG + ".ops")
_lib = importlib.import_module(PKG + "._lib")
_lib.call_nostream("gcssl_init")


def run(N, H, K, dt, iters=20, fused=True):
    x = (torch.rand(N, H, H, K, device="cuda") * 2 - 1).to(dt)
    w = (torch.rand(K, 64, 4, 4, device="cuda") * 2 - 1) * 0.05
    wf = torch.empty(K, 16, 64, device="cuda", dtype=dt); wt = torch.empty(64, 16, K, device="cuda", dtype=dt)
    ops.prep_conv_weight(w, wf, wt, K, 64, 64, ops.code(wt))
    z = torch.empty(N, 2 * H, 2 * H, 64, device="cuda"); a = torch.empty(N, 2 * H, 2 * H, 64, device="cuda", dtype=dt)
    mean = torch.empty(N, 64, device="cuda"); rstd = torch.empty(N, 64, device="cuda"); pool = torch.zeros(N, 64, device="cuda")
    def step():
        if fused:
            ops.convt_in_relu_fwd(x, wt, mean, rstd, K, z32=z, z_n0=N - N // 3, a=None, pool=pool)
        else:
            ops.conv_dgrad(x, wt, z, 64, K)
            ops.in_act_fwd(z, a, mean, rstd, 64, 2, pool=pool)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * N * 4 * H * H * 64 * 4 * K
    return us, fl / us / 1e6


if __name__ == "__main__":
    # (round 3's PMC record of the dominant label was taken in fp16 under a bf16 heading: this script ignored the dtype -- VERDICT r3 #10)
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[4]] if len(sys.argv) > 4 else torch.bfloat16
    if len(sys.argv) > 3:
        cases = [(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]))]
    else:
        cases = [(n, 16, 128) for n in (8, 32, 64, 128, 256, 512, 768)] + [(n, 8, 256) for n in (128, 512, 1024, 3072)]
    for N, H, K in cases:
        us_f, tf_f = run(N, H, K, dt, fused=True)
        us_u, tf_u = (run(N, H, K, dt, fused=False) if not os.environ.get("GCSSL_CT_FUSED_ONLY") else (0.0, 0.0))
        blocks = N * H * H // 256
        print(f"N={N:5d} H={H:2d} K={K:3d} blocks={blocks:4d}  fused {us_f:8.1f} us {tf_f:7.1f} TF/s ({us_f / -(-blocks // 256) / (K // 8):6.2f} us/step)   "
              f"unfused conv+norm {us_u:8.1f} us  rotate={os.environ.get('GCSSL_CT_ROTATE', '1')}")
