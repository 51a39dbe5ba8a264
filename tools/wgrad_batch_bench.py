"""Standalone launcher of the critic's batched weight-gradient launch (D.c2-4.wgrad: gcssl_conv4x4s2_wgrad_batch on the c2 / c3 / c4
shapes at 4B samples) for rocprofv3 --pmc / timing.  usage: python tools/wgrad_batch_bench.py [B=256] [S=32] [bf16|fp16] [reps]"""
import importlib, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
ops = importlib.import_module("gan-calibrated-semi-supervised-learning_amd.ops")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[3] if len(sys.argv) > 3 else "bf16"]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
lay, fl = [], 0.0
for l, (cin, cout) in enumerate([(64, 128), (128, 256), (256, 512)], start=1):
    N, Hi = 4 * B, S >> l
    x = (torch.rand(N, Hi, Hi, cin, device="cuda") * 2 - 1).to(dt)
    dy = (torch.rand(N, Hi // 2, Hi // 2, cout, device="cuda") * 2 - 1).to(dt)
    slab = torch.empty(ops.wgrad_splits(N, Hi, Hi, cin, cout), cout, 16, cin, device="cuda")
    lay.append((x, dy, slab, cin, cout)); fl += 2.0 * N * (Hi // 2) ** 2 * cout * 16 * cin
b = ops.WgradBatch(lay)
for _ in range(3): b.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): b.run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"wgrad batch c2-c4 B={B} S={S}: {ms * 1e3:.1f} us  {fl / ms / 1e9:.1f} TF/s  {ops.last_kernel()}  grid {ops.last_grid()}")
