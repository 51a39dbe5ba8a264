#!/usr/bin/env python3
"""fp16 mode error against the pinned oracle as a function of the critic's static loss scale (GCSSL_LOSS_SCALE_D): is the
one-sided un-clipped gradient-norm error of the fp16 mode a range effect (saturation at the top / flush at the bottom) or
operand rounding?  Run on the GPU box:  python tools/archive/ls_sweep.py [scales...]  (GCSSL_FIN=0 in the environment: unfused norms)"""
import importlib, json, os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
import mode_error as ME

synth = importlib.import_module(ME.PKG + ".synth")
state, ref, taps = ME.oracle_reference(synth)
scales = [float(a) for a in sys.argv[1:]] or [1, 8, 32, 128, 512, 2048]
keys = ("scores", "delta", "wd", "gp", "d_grad_norm", "g_grad_norm", "step2_gp", "step2_d_grad_norm")
for dt, ls in [("fp32", None), ("bf16", None)] + [("fp16", s) for s in scales]:
    if ls is None:
        os.environ.pop("GCSSL_LOSS_SCALE_D", None)
    else:
        os.environ["GCSSL_LOSS_SCALE_D"] = str(ls)
    m = ME.measure(dt, state, ref, taps)
    print(dt, "loss_scale_d", ls, "FIN", os.environ.get("GCSSL_FIN", "1"), json.dumps({k: float(f"{m[k]:.3g}") for k in keys}), flush=True)
