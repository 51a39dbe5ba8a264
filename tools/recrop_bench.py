"""Throughput of the GPU re-crop stage (row f1) beside the reference's host loop restated with Pillow.
usage: python tools/recrop_bench.py [B] [S] [n_images] [W] [H]"""
import importlib
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
RF = importlib.import_module("gan-calibrated-semi-supervised-learning_amd.refine")

B, S, NI, W, H = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 256), (2, 32), (3, 32), (4, 1280), (5, 720)))
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(NI)]
atlas = RF.ImageAtlas(imgs, "cuda")
g = torch.Generator().manual_seed(1)
pred = torch.stack([torch.rand(B, generator=g) * 0.4 + 0.3, torch.rand(B, generator=g) * 0.4 + 0.3,
                    torch.rand(B, generator=g) * 0.4 + 0.1, torch.rand(B, generator=g) * 0.4 + 0.1], 1).cuda()
delta = (torch.randn(B, 4, generator=g) * 0.3).cuda()
idx = torch.randint(0, NI, (B,), generator=g).int().cuda()
for _ in range(3):
    out = RF.get_refined_patch_batch(atlas, idx, pred, delta, S)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    out = RF.get_refined_patch_batch(atlas, idx, pred, delta, S)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
refined = RF.apply_delta_eval(pred, delta).cpu().numpy()
crop_px = float(np.mean(np.clip(refined[:, 2], 0.05, 0.8) * W * np.clip(refined[:, 3], 0.05, 0.8) * H))
print(f"GPU  B={B} S={S} images {NI}x{W}x{H}: {ms * 1e3:.1f} us/call  {B / ms * 1e3:.0f} patches/s  "
      f"(mean crop {crop_px / 1e3:.0f} kpx -> {B * crop_px * 3 / ms / 1e6:.1f} GB/s of source pixels)")
try:
    from PIL import Image, ImageOps
    pil = [Image.fromarray(im) for im in imgs]
    pb, ix = pred.cpu().numpy(), idx.cpu().numpy()
    t0 = time.time()
    n = min(B, 64)
    for i in range(n):                      # the reference's loop body (cgan/cgan_train_enhanced.py:60-118), images cached
        img = pil[ix[i]]
        cx, cy, w, h = (float(v) for v in np.clip(refined[i], [0.1, 0.1, 0.05, 0.05], [0.9, 0.9, 0.8, 0.8]))
        x1, y1 = max(0, cx * W - w * W / 2), max(0, cy * H - h * H / 2)
        x2, y2 = min(W, cx * W + w * W / 2), min(H, cy * H + h * H / 2)
        crop = img.crop((int(x1), int(y1), int(x2), int(y2)))
        if crop.width != crop.height:
            pw, ph = max(crop.height - crop.width, 0), max(crop.width - crop.height, 0)
            crop = ImageOps.expand(crop, (pw // 2, ph // 2, pw - pw // 2, ph - ph // 2), fill=(128, 128, 128))
        crop = crop.resize((S, S), Image.BICUBIC)
        t = (torch.from_numpy(np.asarray(crop).copy()).permute(2, 0, 1).float().div(255) - 0.5) / 0.5
    dt = (time.time() - t0) / n
    print(f"CPU  Pillow loop (1 core, images already decoded): {dt * 1e6:.0f} us/patch  {1 / dt:.0f} patches/s")
except ImportError:
    print("Pillow not importable: no host baseline")
