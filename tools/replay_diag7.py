import importlib, sys, os
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import conftest
synth = conftest.load_pkg("synth")
import test_engine_gpu as TE
mode = sys.argv[1]; dtype = sys.argv[2]
engine, eng_e, call = TE._bench_like(synth, dtype, lr=0.0)
_, eng_g, call_g = TE._bench_like(synth, dtype, lr=0.0)
gi = engine.GraphedIteration(eng_g, *call_g)
def nan(t): return int((~torch.isfinite(t.float())).sum())
for it in range(3):
    eng_e.run_iteration(*call); gi.replay(); torch.cuda.synchronize()
    err = float((eng_g.G.g - eng_e.G.g).norm() / eng_e.G.g.norm())
    print(f"[{mode} {dtype}] it {it}: G err {err:.3e} nanG {nan(eng_g.G.g)} nanE {nan(eng_e.G.g)}")
    if mode == "ddiff":
        x = float((eng_g.D.g - eng_e.D.g).norm())
    elif mode == "alloc":
        x = torch.empty(2767808, device="cuda"); del x
    elif mode == "allocfill":
        x = torch.full((2767808,), 1000.0, device="cuda"); del x
    elif mode == "poison":
        xs = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 18, 1 << 20, 2767808, 1 << 23, 1 << 25)]; del xs
if mode == "poison":
    # which of the graph engine's tensors carry NaN now?
    for name in ("g_dab", "g_gdelta", "g_delta", "g_pooled", "g_poolsum", "g_traw"):
        print("   ", name, nan(getattr(eng_g, name)))
    for k in range(4):
        print("    g_dzu", k, nan(eng_g.g_dzu[k]), "g_zu", nan(eng_g.g_zu[k]), "umean", nan(eng_g.g_umean[k]), "urstd", nan(eng_g.g_urstd[k]), "dzd", nan(eng_g.g_dzd[k]))
    for k in ("g_dcat1", "g_dcat2", "g_dcat3", "g_dd4"):
        print("   ", k, nan(getattr(eng_g, k)))
    print("    G.g per key:", {k: nan(v) for k, v in eng_g.G.gviews.items() if nan(v)})
    print("    eager G.g per key:", {k: nan(v) for k, v in eng_e.G.gviews.items() if nan(v)})
