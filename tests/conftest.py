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


def check_grad(fix: dict, name: str, t, synth_mod, rtol: float, frac_ok: float = 0.98, norm_rtol: float = 5e-3,
               outlier_rtol: float = 5e-2) -> float:
    """Gradient comparison that tolerates activation-kink flips -- and nothing else.

    A weight gradient is a sum over >1e5 positions of (upstream grad x ReLU/LeakyReLU mask x input).  When a
    pre-activation sits within fp32 rounding of 0 (measured: 2 of 524288 elements of G.up4 at S=64 have |xhat| < 1e-6)
    the two implementations may take different branches; that one element moves every weight-gradient entry of its
    channel by ~1e-2 of the tensor's max although everything upstream agrees to 1e-6.  A flip is localised (one channel: at
    most 1.6 % of a tensor's entries) and small, a real bug is neither, so: at least `frac_ok` of the pinned elements must
    agree within `rtol` of the tensor scale, EVERY element within `outlier_rtol` (no entry may be arbitrarily wrong), and
    the full-tensor L2 norm within `norm_rtol`."""
    a = np.asarray(t, dtype=np.float64).reshape(-1)
    if name in fix:
        b = np.asarray(fix[name], dtype=np.float64).reshape(-1)
        nrm_ref, sel = float(np.sqrt((b ** 2).sum())), a
    else:
        sel = a[synth_mod.sample_indices(a.size, 256)]
        b = np.asarray(fix[name + "@sample"], dtype=np.float64)
        nrm_ref = float(fix[name + "@norm"])
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(sel - b) / scale
    good = float((err <= rtol).mean())
    e_n = abs(float(np.sqrt((a ** 2).sum())) - nrm_ref) / max(nrm_ref, 1e-30)
    assert good >= frac_ok and e_n <= norm_rtol and float(err.max()) <= outlier_rtol, \
        f"{name}: {good:.3f} of elements within {rtol}, worst {float(err.max()):.2e}, norm err {e_n:.2e}"
    return 1.0 - good
