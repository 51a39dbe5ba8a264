import importlib, sys, os
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import conftest
synth = conftest.load_pkg("synth")
import test_engine_gpu as TE
mode = sys.argv[1]; dtype = "bf16"
engine, eng_e, call = TE._bench_like(synth, dtype, lr=0.0)
_, eng_g, call_g = TE._bench_like(synth, dtype, lr=0.0)
gi = engine.GraphedIteration(eng_g, *call_g)
for it in range(3):
    eng_e.run_iteration(*call); gi.replay(); torch.cuda.synchronize()
    err = float((eng_g.G.g - eng_e.G.g).norm() / eng_e.G.g.norm())
    print(f"[{mode}] it {it}: G err {err:.3e}")
    if "1" in mode:
        x = float(eng_g.D.state[0]) == float(eng_e.D.state[0]) == 2 * (it + 1)
        x = float(eng_g.G.state[0]) == float(eng_e.G.state[0]) == it + 1
    if "2" in mode:
        for fg, fe, name in ((eng_g.D, eng_e.D, "D"), (eng_g.G, eng_e.G, "G")):
            a, b = fg.g, fe.g
            ok = bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
            e2 = float((a - b).norm() / b.norm())
    if "3" in mode:
        for fg, fe, name in ((eng_g.D, eng_e.D, "D"), (eng_g.G, eng_e.G, "G")):
            per_key = sorted(((float((fg.gviews[k] - fe.gviews[k]).norm() / (fe.gviews[k].norm() + 1e-30)), k,
                              float(fe.gviews[k].norm()), float(fg.gviews[k].norm())) for k in fe.keys), reverse=True)[:3]
    if "4" in mode:
        for x, y in ((float(eng_e.gp_sum), float(eng_g.gp_sum)), (float(eng_e.eiou_acc), float(eng_g.eiou_acc)),
                     (float(eng_e.D.state[2]), float(eng_g.D.state[2])), (float(eng_e.G.state[2]), float(eng_g.G.state[2]))):
            pass
