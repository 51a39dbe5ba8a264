"""GPU re-crop stage: the device-side counterpart of the reference's host loop
`get_refined_patch_batch` (cgan/cgan_train_enhanced.py:37-137).

The reference re-opens every source image with PIL on the host for every critic and generator step (its real
wall-clock bottleneck, about 60 patches/s).  Here the decoded RGB images live in HBM once (`ImageAtlas`) and one kernel
per call does box clamp -> crop -> grey letterbox -> Pillow-exact BICUBIC resize -> normalise for the whole batch, on the
training stream, with no host round trip (Delta never leaves the device)."""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib

call = _lib.call


class ImageAtlas:
    """Decoded RGB uint8 images (H, W, 3), concatenated in device memory."""

    def __init__(self, images: Sequence, device="cuda"):
        arrs = [np.ascontiguousarray(np.asarray(im, dtype=np.uint8)) for im in images]
        for a in arrs:
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("images must be (H, W, 3) uint8 RGB")
        sizes = [a.size for a in arrs]
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        flat = np.concatenate([a.reshape(-1) for a in arrs]) if arrs else np.zeros(0, np.uint8)
        self.data = torch.from_numpy(flat).to(device)
        self.off = torch.from_numpy(offs).to(device)
        self.w = torch.tensor([a.shape[1] for a in arrs], dtype=torch.int32, device=device)
        self.h = torch.tensor([a.shape[0] for a in arrs], dtype=torch.int32, device=device)
        self.max_side = max(max(a.shape[0], a.shape[1]) for a in arrs)
        self.n = len(arrs)
        self._ws = {}

    def workspace(self, B: int, S: int, max_side: int) -> torch.Tensor:
        """Scratch for the per-sample resize coefficient tables (kept between calls)."""
        key = (B, S, max_side)
        if key not in self._ws:
            n = _lib.lib().gcssl_recrop_ws_ints(B, S, max_side)
            if n < 0:
                raise RuntimeError(f"gcssl_recrop_ws_ints({B}, {S}, {max_side}) -> {_lib.ERRORS.get(n, n)}")
            self._ws[key] = torch.empty(n, dtype=torch.int32, device=self.data.device)
        return self._ws[key]

    @classmethod
    def from_paths(cls, paths: Sequence[str], device="cuda") -> "ImageAtlas":
        """Image.open(path).convert('RGB') (:69) once per distinct path; needs Pillow on the host."""
        from PIL import Image
        return cls([np.asarray(Image.open(p).convert("RGB")) for p in paths], device)


def apply_delta_eval(pred_bboxes: torch.Tensor, deltas: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """apply_delta_to_bbox(pred, delta, training=False) (cgan/losses.py:108-150) as one launch."""
    if out is None:
        out = torch.empty_like(pred_bboxes)
    call("gcssl_apply_delta_eval", pred_bboxes, deltas, out, pred_bboxes.shape[0])
    return out


def recrop(atlas: ImageAtlas, img_idx: torch.Tensor, refined_boxes: torch.Tensor, pred_bboxes: torch.Tensor,
           img_size: int, fallback_patches: Optional[torch.Tensor] = None, status: Optional[torch.Tensor] = None,
           out: Optional[torch.Tensor] = None, max_side: Optional[int] = None, letterbox=False) -> torch.Tensor:
    """letterbox=True: CalibratorDataset._letterbox semantics (cgan/dataset.py:104-124) -- refined_boxes are cropped as
    given (no clamp, validity test or fallback) -- i.e. the dataset's pred/gt patches.  letterbox="round": the same with
    the crop edges rounded to nearest, which is what cgan/inference.py:59-68 does (Image.crop on float coordinates)."""
    B = refined_boxes.shape[0]
    if out is None:
        out = torch.empty(B, 3, img_size, img_size, device=refined_boxes.device, dtype=torch.float32)
    if img_idx.dtype != torch.int32:
        img_idx = img_idx.to(torch.int32)
    ms = int(max_side if max_side is not None else atlas.max_side)
    call("gcssl_recrop_patches", atlas.data, atlas.data.numel(), atlas.off, atlas.w, atlas.h, img_idx,
         refined_boxes.contiguous(), pred_bboxes.contiguous() if pred_bboxes is not None else None, fallback_patches, out,
         status, atlas.workspace(B, img_size, ms), B, img_size, ms, 2 if letterbox == "round" else (1 if letterbox else 0))
    return out


def get_refined_patch_batch(atlas: ImageAtlas, img_idx: torch.Tensor, pred_bboxes: torch.Tensor,
                            deltas_pred: torch.Tensor, img_size: int, fallback_patches: Optional[torch.Tensor] = None,
                            status: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Same role and argument order as the reference's function, with (atlas, img_idx) in place of the list of image
    paths and everything on the device.  Returns (B, 3, img_size, img_size) fp32 with no autograd edge to deltas_pred."""
    refined = apply_delta_eval(pred_bboxes.detach().float().contiguous(), deltas_pred.detach().float().contiguous())
    return recrop(atlas, img_idx, refined, pred_bboxes.detach().float(), img_size, fallback_patches, status)


class RefineStage:
    """The re-crop stage as the engine's ``refine_fn(delta, k)`` with every buffer preallocated, so that the whole call is two
    kernel launches on static pointers -- capturable into GraphedIteration's hipGraphs (the reference's step
    contains this stage twice per critic step and once per generator step, cgan/cgan_train_enhanced.py:313-315,358-360).

    ``img_idx`` / ``pred_box`` / ``fallback`` are STATIC device tensors (the batch's source-image indices, predicted boxes and
    pred patches): a training loop that replays graphs refills them in place, as it does the engine's other inputs.  One output
    buffer per call slot k (critic steps 0..n_critic-1, generator step n_critic): slot n_critic's patch is read by the
    value-only critic forward long after the call, beside the next critic step's calls."""

    def __init__(self, atlas: ImageAtlas, img_idx: torch.Tensor, pred_box: torch.Tensor, img_size: int, n_slots: int,
                 fallback: Optional[torch.Tensor] = None, max_side: Optional[int] = None):
        B, dev = pred_box.shape[0], pred_box.device
        self.atlas, self.S = atlas, int(img_size)
        self.img_idx = img_idx.to(torch.int32).contiguous()
        self.pred_box = pred_box.detach().float().contiguous()
        self.fallback = fallback
        self.max_side = int(max_side if max_side is not None else atlas.max_side)
        self.boxes = [torch.empty(B, 4, device=dev) for _ in range(n_slots)]
        self.out = [torch.empty(B, 3, self.S, self.S, device=dev) for _ in range(n_slots)]
        # (one coefficient-table workspace per slot: the generator step's call may run beside a critic step's on another stream)
        n = _lib.lib().gcssl_recrop_ws_ints(B, self.S, self.max_side)
        if n < 0:
            raise RuntimeError(f"gcssl_recrop_ws_ints({B}, {self.S}, {self.max_side}) -> {_lib.ERRORS.get(n, n)}")
        self.ws = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(n_slots)]

    def __call__(self, delta: torch.Tensor, k: int) -> torch.Tensor:
        a = self.atlas
        apply_delta_eval(self.pred_box, delta, out=self.boxes[k])
        call("gcssl_recrop_patches", a.data, a.data.numel(), a.off, a.w, a.h, self.img_idx, self.boxes[k], self.pred_box,
             self.fallback, self.out[k], None, self.ws[k], self.pred_box.shape[0], self.S, self.max_side, 0)
        return self.out[k]
