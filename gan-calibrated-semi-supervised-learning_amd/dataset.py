"""`CalibratorDataset`: the reference's on-disk formats and sample construction,
cgan/dataset.py:17-236, without torchvision and with a device-side batch path.

Directory layout (YOLO style, all coordinates normalised):
    images/*.jpg   labels_gt/*.txt (cls cx cy w h)   labels_pred/*.txt (cls cx cy w h conf)
Index building (`_prepare_index`, :128-153): every predicted box is matched to the ground-truth box of largest IoU,
kept when that IoU >= iou_thr (several predictions may share one ground truth, :181-205); the regression target is
`_bbox2delta` (:74-101).  `__getitem__` returns (pred_patch, gt_patch, delta_true, pred_box, img_path) like the
reference, produced on the host with Pillow.  `gpu_batch` returns the same patches for a whole batch from a
device-resident `ImageAtlas` through the re-crop kernel in letterbox mode (bit-exact with the host path)."""
from __future__ import annotations

import math
import os
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

DEFAULT_IMG_SIZE, DEFAULT_IOU_THR = 128, 0.25          # cgan/config.yaml: img_size, iou_threshold


def bbox_iou(box1, box2) -> float:
    """IoU of two (cx, cy, w, h) boxes, cgan/dataset.py:57-71."""
    b1x1, b1y1, b1x2, b1y2 = box1[0] - box1[2] / 2, box1[1] - box1[3] / 2, box1[0] + box1[2] / 2, box1[1] + box1[3] / 2
    b2x1, b2y1, b2x2, b2y2 = box2[0] - box2[2] / 2, box2[1] - box2[3] / 2, box2[0] + box2[2] / 2, box2[1] + box2[3] / 2
    inter = max(0, min(b1x2, b2x2) - max(b1x1, b2x1)) * max(0, min(b1y2, b2y2) - max(b1y1, b2y1))
    union = (b1x2 - b1x1) * (b1y2 - b1y1) + (b2x2 - b2x1) * (b2y2 - b2y1) - inter
    return float(inter / union) if union > 0 else 0.0


def bbox2delta(gt, pred) -> torch.Tensor:
    """(dx, dy, log dw, log dh) with the predicted box's geometric-mean size as the unit, cgan/dataset.py:74-101."""
    norm = max(math.sqrt(float(pred[2]) * float(pred[3])), 0.05)
    dx = (float(gt[0]) - float(pred[0])) / norm
    dy = (float(gt[1]) - float(pred[1])) / norm
    eps = 1e-6
    gw, gh = max(float(gt[2]), eps), max(float(gt[3]), eps)
    pw, ph = max(float(pred[2]), eps), max(float(pred[3]), eps)
    return torch.tensor([dx, dy, math.log(max(0.1, min(10.0, gw / pw))), math.log(max(0.1, min(10.0, gh / ph)))],
                        dtype=torch.float32)


def load_boxes(txt_path: Path, min_fields: int) -> torch.Tensor:
    """YOLO txt -> (n, 4); ground truth needs >= 5 fields per line, predictions >= 6 (the confidence), :155-179."""
    if txt_path.stat().st_size == 0:
        return torch.empty((0, 4))
    boxes = []
    for line in txt_path.read_text().strip().splitlines():
        parts = line.strip().split()
        if len(parts) >= min_fields:
            boxes.append([float(x) for x in parts[1:5]])
    return torch.tensor(boxes) if boxes else torch.empty((0, 4))


def greedy_matching(pred_boxes: torch.Tensor, gt_boxes: torch.Tensor, iou_thr: float) -> List[Tuple[int, int]]:
    """Every prediction takes its max-IoU ground truth if IoU >= thr (many-to-one), cgan/dataset.py:181-205."""
    if len(pred_boxes) == 0 or len(gt_boxes) == 0:
        return []
    iou = torch.zeros((len(pred_boxes), len(gt_boxes)))
    for i, pb in enumerate(pred_boxes):
        for j, gb in enumerate(gt_boxes):
            iou[i, j] = bbox_iou(pb, gb)
    best, idx = iou.max(dim=1)
    return [(i, int(idx[i])) for i in range(len(pred_boxes)) if best[i] >= iou_thr]


class CalibratorDataset(torch.utils.data.Dataset):
    def __init__(self, root, img_size: Optional[int] = None, iou_thr: Optional[float] = None):
        super().__init__()
        self.root = Path(root)
        self.img_dir, self.gt_dir, self.pred_dir = self.root / "images", self.root / "labels_gt", self.root / "labels_pred"
        self.img_size = int(img_size if img_size is not None else DEFAULT_IMG_SIZE)
        self.iou_thr = float(iou_thr if iou_thr is not None else DEFAULT_IOU_THR)
        self.samples: List[Tuple[Path, int, torch.Tensor, torch.Tensor, torch.Tensor]] = []
        self._prepare_index()
        self._atlas = None
        self._img_index = {}

    def _prepare_index(self) -> None:
        for txt_pred in sorted(self.pred_dir.glob("*.txt")):
            name = txt_pred.stem
            txt_gt, img_path = self.gt_dir / f"{name}.txt", self.img_dir / f"{name}.jpg"
            if not txt_gt.exists() or not img_path.exists():
                continue
            gt_boxes, pred_boxes = load_boxes(txt_gt, 5), load_boxes(txt_pred, 6)
            if len(gt_boxes) == 0 or len(pred_boxes) == 0:
                continue
            for pi, gi in greedy_matching(pred_boxes, gt_boxes, self.iou_thr):
                self.samples.append((img_path, 0, pred_boxes[pi], bbox2delta(gt_boxes[gi], pred_boxes[pi]), gt_boxes[gi]))

    def __len__(self) -> int:
        return len(self.samples)

    # ---- host path (Pillow), same return value as the reference's __getitem__ (:222-236)
    @staticmethod
    def _letterbox(img, bbox_xywh, out_size: int):
        from PIL import Image, ImageOps
        W, H = img.size
        cx, cy, w, h = bbox_xywh
        px, py, pw, ph = float(cx) * W, float(cy) * H, float(w) * W, float(h) * H
        x1, y1 = max(0, px - pw / 2), max(0, py - ph / 2)
        x2, y2 = min(W, px + pw / 2), min(H, py + ph / 2)
        crop = img.crop((int(x1), int(y1), int(x2), int(y2)))
        pad_w, pad_h = max(crop.height - crop.width, 0), max(crop.width - crop.height, 0)
        crop = ImageOps.expand(crop, (pad_w // 2, pad_h // 2, pad_w - pad_w // 2, pad_h - pad_h // 2), fill=(128, 128, 128))
        return crop.resize((out_size, out_size), Image.BICUBIC)

    @staticmethod
    def _to_tensor(pil_img) -> torch.Tensor:
        """ToTensor + Normalize([0.5]*3, [0.5]*3) (:50-53)."""
        t = torch.from_numpy(np.asarray(pil_img, np.uint8).copy()).permute(2, 0, 1).float().div(255)
        return (t - 0.5) / 0.5

    def __getitem__(self, idx: int):
        from PIL import Image
        img_path, _, pred_box, delta_true, gt_box = self.samples[idx]
        img = Image.open(img_path).convert("RGB")
        gt_patch = self._to_tensor(self._letterbox(img, gt_box, self.img_size))
        pred_patch = self._to_tensor(self._letterbox(img, pred_box, self.img_size))
        return pred_patch, gt_patch, delta_true, pred_box, str(img_path)

    # ---- device path: images decoded once into HBM, patches cut by the re-crop kernel
    def atlas(self, device="cuda"):
        if self._atlas is None:
            from . import refine
            paths = sorted({s[0] for s in self.samples})
            self._img_index = {p: i for i, p in enumerate(paths)}
            self._atlas = refine.ImageAtlas.from_paths([str(p) for p in paths], device)
        return self._atlas

    def gpu_batch(self, indices: Sequence[int], device="cuda"):
        """-> (pred_patch, gt_patch, delta_true, pred_box, img_idx): (B,3,S,S) fp32 x2, (B,4), (B,4), (B,) int32, all on
        the device; img_idx indexes self.atlas() for the re-crop stage of the training loop."""
        from . import refine
        atlas = self.atlas(device)
        rows = [self.samples[i] for i in indices]
        pred = torch.stack([r[2] for r in rows]).float().to(device)
        gt = torch.stack([r[4] for r in rows]).float().to(device)
        delta = torch.stack([r[3] for r in rows]).to(device)
        idx = torch.tensor([self._img_index[r[0]] for r in rows], dtype=torch.int32, device=device)
        pred_patch = refine.recrop(atlas, idx, pred, None, self.img_size, letterbox=True)
        gt_patch = refine.recrop(atlas, idx, gt, None, self.img_size, letterbox=True)
        return pred_patch, gt_patch, delta, pred, idx
