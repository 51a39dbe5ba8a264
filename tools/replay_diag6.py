import importlib, sys, os
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import conftest
PKG = conftest.PKG
synth = conftest.load_pkg("synth")
import test_engine_gpu as TE
mode = sys.argv[1]
def run(dtype, pre_assert, n_it):
    engine, eng_e, call = TE._bench_like(synth, dtype, lr=0.0)
    _, eng_g, call_g = TE._bench_like(synth, dtype, lr=0.0)
    gi = engine.GraphedIteration(eng_g, *call_g)
    if pre_assert:
        assert float(eng_g.G.state[0]) == 0.0 and float(eng_g.D.state[0]) == 0.0
    for it in range(n_it):
        eng_e.run_iteration(*call)
        gi.replay()
        torch.cuda.synchronize()
        err = float((eng_g.G.g - eng_e.G.g).norm() / eng_e.G.g.norm())
        print(f"[{mode} {dtype} pre_assert={pre_assert}] it {it}: G err {err:.3e} |G.g| {float(eng_e.G.g.norm()):.4e} {float(eng_g.G.g.norm()):.4e} steps {float(eng_g.G.state[0])} {float(eng_e.G.state[0])}")
if mode == "func":
    for dt in ("bf16", "fp16"):
        try:
            TE.test_graph_replay_matches_eager_at_bench_config(synth, dt); print("[func", dt, "] passed")
        except AssertionError as e:
            print("[func", dt, "] FAILED", str(e)[:200])
elif mode == "pre":
    run("bf16", True, 3); run("fp16", True, 3)
else:
    run("bf16", False, 3); run("fp16", False, 3)
