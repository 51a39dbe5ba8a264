"""Graph replay vs eager launches at the bench configuration, lr = 0: which scalars / gradient tensors differ, per iteration.
usage: python tools/archive/graph_diag.py [fp16|bf16] [iters]"""
import importlib, sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
PKG = "gan-calibrated-semi-supervised-learning_amd"
synth = importlib.import_module(PKG + ".synth"); engine = importlib.import_module(PKG + ".engine")
dtype = sys.argv[1] if len(sys.argv) > 1 else "fp16"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T = torch.from_numpy


def make():
    seed, B, S, c = 42, 256, 32, 2
    g = {k: T(v) for k, v in synth.generator_state(seed).items()}
    d = {k: T(v) for k, v in synth.discriminator_state(seed).items()}
    eng = engine.StepEngine(g, d, batch=B, size=S, n_critic=c, dtype=dtype, device="cuda:0", seed=seed, lr=0.0)
    inp = synth.step_inputs(seed, B, S, c, tag="bench")
    refined = [T(r).cuda() for r in inp["refined"]]
    call = (T(inp["pred"]).cuda(), T(inp["gt"]).cuda(), T(inp["delta_true"]).cuda(), T(inp["pred_box"]).cuda(), lambda delta, k: refined[k])
    return eng, call


e1, c1 = make(); e2, c2 = make(); g1, cg1 = make(); g2, cg2 = make()
gi1 = engine.GraphedIteration(g1, *cg1); gi2 = engine.GraphedIteration(g2, *cg2)
for it in range(iters):
    e1.run_iteration(*c1); e2.run_iteration(*c2); gi1.replay(); gi2.replay()
    torch.cuda.synchronize()
    def scal(e): return [float(e.D.state[2]), float(e.G.state[2]), float(e.gp_sum), float(e.eiou_acc)] + e.means.tolist()
    print(f"--- iteration {it}")
    for name, e in (("eager1", e1), ("eager2", e2), ("graph1", g1), ("graph2", g2)):
        print(f"{name}: " + " ".join(f"{v:.7g}" for v in scal(e)))
    for name, a, b in (("eager2-eager1", e2, e1), ("graph1-eager1", g1, e1), ("graph2-graph1", g2, g1)):
        dd = float((a.D.g - b.D.g).norm() / b.D.g.norm()); dg = float((a.G.g - b.G.g).norm() / b.G.g.norm())
        worst = sorted(((float((a.D.gviews[k] - b.D.gviews[k]).norm() / (b.D.gviews[k].norm() + 1e-30)), k) for k in b.D.keys), reverse=True)[:3]
        du = max(float((a.u[l] - b.u[l]).abs().max()) for l in range(4))
        print(f"  {name}: D.g {dd:.3e} G.g {dg:.3e} u {du:.2e} worst {[(round(x, 5), k) for x, k in worst]}")
