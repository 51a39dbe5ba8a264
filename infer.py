#!/usr/bin/env python3
"""Calibrate the pseudo-labels of one image with a trained generator -- the surface of the reference's
cgan/inference.py:93-181 (same arguments, same txt formats, same checkpoint handling) on the MI355X path:

    python infer.py --weights runs/exp/G_best.pth --image demo.jpg --pred_txt demo_pred.txt --out_txt demo_calib.txt

`pred_txt`: YOLO txt, one `cls cx cy w h [conf ...]` row per box (normalised); `out_txt` receives the same rows with the
calibrated boxes.  All boxes of the image go through the GPU as ONE batch: crop + grey letterbox + BICUBIC resize with
the re-crop kernel (bit-exact with the Pillow calls of :51-68), eval-mode GeneratorUNet forward on the HIP kernels, the
inference-time box transform of :69-89.  Image decoding and the txt files stay on the host.
"""
from __future__ import annotations

import argparse
import importlib
import sys
from pathlib import Path
from typing import List

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
PKG = "gan-calibrated-semi-supervised-learning_amd"


def load_yolo_txt(txt_path: Path) -> List[List[float]]:
    """cgan/inference.py:27-39."""
    if not txt_path.exists():
        return []
    return [[float(x) for x in line.split()] for line in txt_path.read_text().splitlines() if line.strip()]


def save_yolo_txt(txt_path: Path, rows: List[List[float]]) -> None:
    """cgan/inference.py:41-47 (class id as int, everything else through str())."""
    with open(txt_path, "w") as f:
        for r in rows:
            r[0] = int(r[0])
            f.write(" ".join(map(str, r)) + "\n")


def resolve_config(weights: Path, checkpoint) -> tuple:
    """delta_scale / generator_type from the checkpoint's config, else from the file name, else 0.25 (:107-128)."""
    if isinstance(checkpoint, dict) and "config" in checkpoint:
        cfg = checkpoint["config"]
        return cfg.get("delta_scale", 0.25), cfg.get("generator_type", "unet")
    try:
        return float(weights.stem.split("=")[-1]), "unet"
    except (ValueError, IndexError):
        return 0.25, "unet"


def calibrate(netG, image: np.ndarray, preds: List[List[float]], img_size: int, device, chunk: int = 256) -> List[List[float]]:
    models_refine = importlib.import_module(PKG + ".refine")
    losses = importlib.import_module(PKG + ".losses")
    if not preds:
        return []
    boxes = torch.tensor([p[1:5] for p in preds], dtype=torch.float32, device=device)
    atlas = models_refine.ImageAtlas([image], device)
    idx = torch.zeros(len(preds), dtype=torch.int32, device=device)
    out = []
    with torch.no_grad():
        for b0 in range(0, len(preds), chunk):
            bb = boxes[b0:b0 + chunk]
            patches = models_refine.recrop(atlas, idx[b0:b0 + chunk], bb, None, img_size, letterbox="round")
            delta = netG(patches).float()
            out.append(losses.apply_delta_to_bbox_inference(bb, delta))
    cal = torch.cat(out).cpu().tolist()
    return [[int(p[0])] + c + p[5:] for p, c in zip(preds, cal)]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", type=str, required=True)
    ap.add_argument("--image", type=str, required=True)
    ap.add_argument("--pred_txt", type=str, required=True)
    ap.add_argument("--out_txt", type=str, required=True)
    ap.add_argument("--img_size", type=int, default=128)
    args = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("infer.py needs an MI355X (the HIP path has no CPU fallback)")
    device = torch.device("cuda")
    models = importlib.import_module(PKG + ".models")
    checkpoint = torch.load(args.weights, map_location="cpu", weights_only=False)
    delta_scale, generator_type = resolve_config(Path(args.weights), checkpoint)
    if generator_type == "simple":                                                   # cgan/inference.py:131-134
        netG = models.GeneratorSimpleRegressor(delta_scale=delta_scale).to(device)
    else:
        netG = models.GeneratorUNet(delta_scale=delta_scale).to(device)
    state = checkpoint["generator"] if isinstance(checkpoint, dict) and "generator" in checkpoint else checkpoint
    netG.load_state_dict(state)
    netG.eval()
    from PIL import Image
    image = np.asarray(Image.open(args.image).convert("RGB"))
    preds = load_yolo_txt(Path(args.pred_txt))
    rows = calibrate(netG, image, preds, args.img_size, device)
    save_yolo_txt(Path(args.out_txt), rows)
    print(f"saved {len(rows)} calibrated boxes -> {args.out_txt}")


if __name__ == "__main__":
    main()
