"""Generates tests/golden/recrop.npz: inputs and expected outputs of the reference's re-crop stage
(get_refined_patch_batch, cgan/cgan_train_enhanced.py:37-137), produced with Pillow -- the library the reference calls --
following that function step by step (torchvision is not installed here, so ToTensor + Normalize(0.5, 0.5) are written
out: uint8 -> float32 / 255 -> (x - 0.5) / 0.5, which is what those transforms compute).

Run in the dev container:  python tests/golden/make_recrop_golden.py
"""
import sys
from pathlib import Path

import numpy as np
import torch
from PIL import Image, ImageOps

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))


def make_images():
    rng = np.random.default_rng(1234)
    imgs = []
    for (h, w) in ((120, 160), (143, 97), (200, 200), (64, 333)):
        yy, xx = np.mgrid[0:h, 0:w]
        smooth = np.stack([(xx * 255) // max(w - 1, 1), (yy * 255) // max(h - 1, 1), ((xx + yy) * 3) % 256], -1)
        noise = rng.integers(-40, 41, (h, w, 3))
        imgs.append(np.clip(smooth + noise, 0, 255).astype(np.uint8))
    return imgs


def reference_patch(img_np, refined_box, pred_box, size):
    """Lines 60-129 of the reference for one sample (image already 'opened': a PIL RGB image)."""
    img = Image.fromarray(img_np, "RGB")
    W, H = img.size
    cx, cy, w, h = refined_box
    cx = float(torch.clamp(cx, 0.1, 0.9)); cy = float(torch.clamp(cy, 0.1, 0.9))
    w = float(torch.clamp(w, 0.05, 0.8)); h = float(torch.clamp(h, 0.05, 0.8))
    px, py, pw, ph = cx * W, cy * H, w * W, h * H
    x1, y1 = max(0, px - pw / 2), max(0, py - ph / 2)
    x2, y2 = min(W, px + pw / 2), min(H, py + ph / 2)
    status = 0
    try:
        if x2 <= x1 or y2 <= y1 or (x2 - x1) < 10 or (y2 - y1) < 10:
            ocx, ocy, ow, oh = pred_box
            opx, opy = float(ocx) * W, float(ocy) * H
            opw, oph = float(ow) * W, float(oh) * H
            ox1, oy1 = max(0, opx - opw / 2), max(0, opy - oph / 2)
            ox2, oy2 = min(W, opx + opw / 2), min(H, opy + oph / 2)
            crop = img.crop((int(ox1), int(oy1), int(ox2), int(oy2)))
            status = 1
        else:
            crop = img.crop((int(x1), int(y1), int(x2), int(y2)))
        if crop.width != crop.height:
            pad_w = max(crop.height - crop.width, 0)
            pad_h = max(crop.width - crop.height, 0)
            padding = (pad_w // 2, pad_h // 2, pad_w - pad_w // 2, pad_h - pad_h // 2)
            crop = ImageOps.expand(crop, padding, fill=(128, 128, 128))
        if crop.size != (size, size):
            crop = crop.resize((size, size), Image.BICUBIC)
        t = torch.from_numpy(np.asarray(crop, np.uint8).copy()).permute(2, 0, 1).float().div(255)   # ToTensor
        t = (t - 0.5) / 0.5                                                                          # Normalize
        return t.numpy(), status
    except Exception:
        return np.zeros((3, size, size), np.float32), 2


def reference_letterbox(img_np, box, size):
    """CalibratorDataset._letterbox + transform (cgan/dataset.py:104-124, 50-53) for one box, with Pillow."""
    img = Image.fromarray(img_np, "RGB")
    W, H = img.size
    cx, cy, w, h = box
    px, py, pw, ph = float(cx) * W, float(cy) * H, float(w) * W, float(h) * H
    x1, y1 = max(0, px - pw / 2), max(0, py - ph / 2)
    x2, y2 = min(W, px + pw / 2), min(H, py + ph / 2)
    crop = img.crop((int(x1), int(y1), int(x2), int(y2)))
    pad_w = max(crop.height - crop.width, 0)
    pad_h = max(crop.width - crop.height, 0)
    padding = (pad_w // 2, pad_h // 2, pad_w - pad_w // 2, pad_h - pad_h // 2)
    crop_square = ImageOps.expand(crop, padding, fill=(128, 128, 128))
    crop_resized = crop_square.resize((size, size), Image.BICUBIC)
    t = torch.from_numpy(np.asarray(crop_resized, np.uint8).copy()).permute(2, 0, 1).float().div(255)
    return ((t - 0.5) / 0.5).numpy()


def reference_inference_patch(img_np, box, size):
    """crop_patch + letterbox + transform of cgan/inference.py:51-68,147-166 for one box, with Pillow."""
    img = Image.fromarray(img_np, "RGB")
    W, H = img.size
    cx, cy, w, h = (float(v) for v in box)
    px, py, pw, ph = cx * W, cy * H, w * W, h * H
    x1, y1 = max(0, px - pw / 2), max(0, py - ph / 2)
    x2, y2 = min(W, px + pw / 2), min(H, py + ph / 2)
    crop = img.crop((x1, y1, x2, y2))                       # float box: Pillow rounds
    pad_w = max(crop.height - crop.width, 0)
    pad_h = max(crop.width - crop.height, 0)
    sq = ImageOps.expand(crop, (pad_w // 2, pad_h // 2, pad_w - pad_w // 2, pad_h - pad_h // 2), fill=(128, 128, 128))
    res = sq.resize((size, size), Image.BICUBIC)
    t = torch.from_numpy(np.asarray(res, np.uint8).copy()).permute(2, 0, 1).float().div(255)
    return ((t - 0.5) / 0.5).numpy()


def main():
    imgs = make_images()
    rng = np.random.default_rng(99)
    n = 48
    idx = rng.integers(0, len(imgs), n).astype(np.int32)
    refined = np.stack([rng.uniform(0.0, 1.0, n), rng.uniform(0.0, 1.0, n), rng.uniform(0.01, 0.95, n),
                        rng.uniform(0.01, 0.95, n)], 1).astype(np.float32)
    pred = np.stack([rng.uniform(0.3, 0.7, n), rng.uniform(0.3, 0.7, n), rng.uniform(0.1, 0.5, n),
                     rng.uniform(0.1, 0.5, n)], 1).astype(np.float32)
    # hand-made corner cases: tiny refined box on the smallest image (-> fallback to the predicted box), a predicted box
    # that is empty too (-> failed), an exactly square crop that needs no padding, a crop that is exactly size x size
    idx[0], refined[0], pred[0] = 3, (0.5, 0.5, 0.05, 0.05), (0.5, 0.5, 0.4, 0.6)          # 64 x 333: h*0.05 < 10 px
    idx[1], refined[1], pred[1] = 3, (0.5, 0.5, 0.05, 0.05), (0.5, 0.5, 0.0, 0.0)          # both degenerate: 0 x 0 crop
    idx[5], refined[5], pred[5] = 3, (0.5, 0.5, 0.05, 0.05), (0.5, 0.5, -0.2, 0.3)         # crop raises -> except branch
    idx[6], refined[6], pred[6] = 3, (0.5, 0.5, 0.05, 0.05), (0.5, 0.5, 0.0, 0.3)          # 0 x 19 crop -> all-grey square
    idx[2], refined[2] = 2, (0.5, 0.5, 0.4, 0.4)                                            # 80 x 80 crop of 200 x 200
    idx[3], refined[3] = 2, (0.5, 0.5, 0.16, 0.16)                                          # 32 x 32: no resize at S=32
    idx[4], refined[4] = 0, (0.1, 0.9, 0.8, 0.8)                                            # clipped by the image border
    out = {"n_images": np.int32(len(imgs)), "img_idx": idx, "refined": refined, "pred": pred}
    for i, im in enumerate(imgs):
        out[f"img{i}"] = im
    for size in (32, 64):
        res = [reference_patch(imgs[idx[i]], torch.from_numpy(refined[i]), torch.from_numpy(pred[i]), size) for i in range(n)]
        out[f"patch{size}"] = np.stack([r[0] for r in res]).astype(np.float32)
        out[f"status{size}"] = np.array([r[1] for r in res], np.int32)
        if size == 32:
            out["infer32"] = np.stack([reference_inference_patch(imgs[idx[i]], refined[i], size) for i in range(n)]).astype(np.float32)
        out[f"letterbox{size}"] = np.stack([reference_letterbox(imgs[idx[i]], torch.from_numpy(refined[i]), size)
                                            for i in range(n)]).astype(np.float32)
    np.savez_compressed(ROOT / "tests" / "golden" / "recrop.npz", **out)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if not k.startswith("img") or k == "img_idx"})
    print("status32", np.bincount(out["status32"], minlength=3))


if __name__ == "__main__":
    main()
