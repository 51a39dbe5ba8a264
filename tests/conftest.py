import importlib
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

PKG = "gan-calibrated-semi-supervised-learning_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg(sub: str = ""):
    return importlib.import_module(PKG + ("." + sub if sub else ""))


@pytest.fixture(scope="session")
def synth():
    return load_pkg("synth")


def load_golden(name: str) -> dict:
    with np.load(GOLDEN / f"{name}.npz") as z:
        return {k: z[k] for k in z.files}


def rel_err(a, b) -> float:
    """max |a-b| / max(|b|_max, tiny): error relative to the tensor's scale."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def check_pinned(fix: dict, name: str, t, rtol: float, synth_mod) -> float:
    """Compare tensor ``t`` with what make_golden.pin() stored (full, or norm/sum/sample)."""
    a = np.asarray(t, dtype=np.float64)
    if name in fix:
        e = rel_err(a, fix[name])
        assert e <= rtol, f"{name}: rel err {e:.3e} > {rtol}"
        return e
    nrm = float(fix[name + "@norm"])
    flat = a.reshape(-1)
    smp = flat[synth_mod.sample_indices(flat.size, 256)]
    e1 = abs(float(np.sqrt((flat ** 2).sum())) - nrm) / max(nrm, 1e-30)
    e2 = float(np.abs(smp - fix[name + "@sample"]).max() / max(np.abs(fix[name + "@sample"]).max(), 1e-30))
    assert e1 <= rtol and e2 <= rtol, f"{name}: norm err {e1:.3e}, sample err {e2:.3e} > {rtol}"
    return max(e1, e2)
