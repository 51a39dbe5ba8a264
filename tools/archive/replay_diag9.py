import importlib, sys, os
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import conftest
from conftest import rel_err
synth = conftest.load_pkg("synth")
import test_engine_gpu as TE
flags = sys.argv[1]
def body(dtype):
    engine, eng_e, call = TE._bench_like(synth, dtype, lr=0.0)
    _, eng_g, call_g = TE._bench_like(synth, dtype, lr=0.0)
    gi = engine.GraphedIteration(eng_g, *call_g)
    if "P" in flags:
        assert float(eng_g.G.state[0]) == 0.0 and float(eng_g.D.state[0]) == 0.0
    for it in range(3):
        eng_e.run_iteration(*call)
        gi.replay()
        torch.cuda.synchronize()
        if "A" in flags:
            assert float(eng_g.D.state[0]) == float(eng_e.D.state[0]) == 2 * (it + 1)
            assert float(eng_g.G.state[0]) == float(eng_e.G.state[0]) == it + 1
        for fg, fe, name in ((eng_g.D, eng_e.D, "D"), (eng_g.G, eng_e.G, "G")):
            a, b = fg.g, fe.g
            if "F" in flags:
                assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
            err = float((a - b).norm() / b.norm())
            if "K" in flags:
                per_key = sorted(((float((fg.gviews[k] - fe.gviews[k]).norm() / (fe.gviews[k].norm() + 1e-30)), k,
                                  float(fe.gviews[k].norm()), float(fg.gviews[k].norm())) for k in fe.keys), reverse=True)[:3]
            print(f"[{flags}] it {it} {name} err {err:.3e}")
            if name == "G" and err > 0.1:
                def mx(t): return float(t.float().abs().nan_to_num(nan=1e30, posinf=1e30).max())
                for tag, e in (("graph", eng_g), ("eager", eng_e)):
                    print(f"   {tag}: pred_box {mx(call_g[3] if e is eng_g else call[3]):.3e} delta_true {mx(call_g[2] if e is eng_g else call[2]):.3e} "
                          f"delta {mx(e.g_delta):.3e} gdelta {mx(e.g_gdelta):.3e} cal {mx(e.g_cal):.3e} traw {mx(e.g_traw):.3e} pooled {mx(e.g_pooled):.3e} dab {mx(e.g_dab):.3e}")
                    print(f"   {tag}: " + " ".join(f"dzu{k} {mx(e.g_dzu[k]):.2e} zu{k} {mx(e.g_zu[k]):.2e} um{k} {mx(e.g_umean[k]):.2e} ur{k} {mx(e.g_urstd[k]):.2e}" for k in range(4)))
                    print(f"   {tag}: " + " ".join(f"dzd{k} {mx(e.g_dzd[k]):.2e}" for k in range(4)) + f" dcat3 {mx(e.g_dcat3):.2e} dcat2 {mx(e.g_dcat2):.2e} dcat1 {mx(e.g_dcat1):.2e} dd4 {mx(e.g_dd4):.2e}")
                    print(f"   {tag}: G.g per key " + " ".join(f"{k}={mx(v):.2e}" for k, v in e.G.gviews.items()))
        if "X" in flags:
            for x, y in ((float(eng_e.gp_sum), float(eng_g.gp_sum)), (float(eng_e.eiou_acc), float(eng_g.eiou_acc)),
                         (float(eng_e.D.state[2]), float(eng_g.D.state[2])), (float(eng_e.G.state[2]), float(eng_g.G.state[2]))):
                assert np.isfinite(x) and abs(x - y) <= 2e-3 * max(abs(x), 1e-6), (it, x, y)
body("bf16")
