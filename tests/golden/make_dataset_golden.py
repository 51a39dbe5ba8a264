#!/usr/bin/env python3
"""Pins row f2 (CalibratorDataset, cgan/dataset.py:128-205) to the reference's OWN record of its dataset.

The reference module cannot be imported here (it needs torchvision), but its committed W&B run
`cgan/wandb/run-20250718_183815-3pffojdl` logged what its CalibratorDataset produced on the committed directory
`datasets/500_100_100/cgan`: dataset/total_samples 18523, train_samples 16671, val_samples 1852 (val_split 0.1).  This
script builds OUR CalibratorDataset on that directory, checks those three numbers, and stores them together with a
checksum of every (pred_box, delta_true, gt_box) triple in tests/golden/dataset_500.json -- the regression fixture
tests/test_dataset.py::test_reference_dataset_record compares against (the data directory itself never leaves
/root/reference; the test is skipped where it is absent).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_dataset_golden.py
"""
import hashlib
import importlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")
DATA = REF / "datasets" / "500_100_100" / "cgan"
SUMMARY = REF / "cgan" / "wandb" / "run-20250718_183815-3pffojdl" / "files" / "wandb-summary.json"
PKG = "gan-calibrated-semi-supervised-learning_amd"


def dataset_record(ds) -> dict:
    """what the fixture stores about a CalibratorDataset: size and order-dependent checksums of its sample table"""
    pred = np.stack([s[2].numpy() for s in ds.samples]).astype(np.float32)
    delta = np.stack([s[3].numpy() for s in ds.samples]).astype(np.float32)
    gt = np.stack([s[4].numpy() for s in ds.samples]).astype(np.float32)
    names = "\n".join(s[0].name for s in ds.samples)
    return dict(total_samples=len(ds),
                images=len({s[0] for s in ds.samples}),
                sha256_names=hashlib.sha256(names.encode()).hexdigest(),
                sha256_pred_box=hashlib.sha256(pred.tobytes()).hexdigest(),
                sha256_gt_box=hashlib.sha256(gt.tobytes()).hexdigest(),
                # deltas go through log(): compare by value (sum / abs-sum / a strided sample), not by bit pattern
                delta_sum=[float(v) for v in delta.astype(np.float64).sum(0)],
                delta_abs_sum=[float(v) for v in np.abs(delta.astype(np.float64)).sum(0)],
                delta_sample=delta[:: max(1, len(delta) // 64)][:64].tolist())


def main():
    summary = json.loads(SUMMARY.read_text())
    want = {k: summary[f"dataset/{k}"] for k in ("total_samples", "train_samples", "val_samples")}
    ds = importlib.import_module(PKG + ".dataset").CalibratorDataset(DATA)
    train = importlib.import_module("train")
    n_train, n_val = train.split_lengths(len(ds), 0.1)                   # val_split of that run (its config.yaml:99-100)
    assert len(ds) == want["total_samples"], (len(ds), want)
    assert (n_train, n_val) == (want["train_samples"], want["val_samples"]), (n_train, n_val, want)
    rec = dict(source=dict(data="datasets/500_100_100/cgan", record=str(SUMMARY.relative_to(REF)), reference_logged=want,
                           val_split=0.1), **dataset_record(ds))
    out = ROOT / "tests" / "golden" / "dataset_500.json"
    out.write_text(json.dumps(rec, indent=1))
    print(f"{out}: {len(ds)} pairs over {rec['images']} images; reference logged {want}")


if __name__ == "__main__":
    main()
