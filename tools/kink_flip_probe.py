"""How often does the first critic step of the simple_B4_S32 fixture (fp32) land in its second mode -- a borderline LeakyReLU
element taking the other branch -- and does it depend on the weight-gradient launch form?  Repeats the per-tensor gradient check of
tests/test_engine_gpu.py::test_fp32_first_critic_step_gradients on fresh engines and prints the error of model.0.weight_orig.
usage: python tools/kink_flip_probe.py [repeats=20]"""
import importlib, sys
from pathlib import Path
import numpy as np
import torch
R = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(R)); sys.path.insert(0, str(R / "tests"))
import conftest                                                    # noqa: E402
T = torch.from_numpy
teg = importlib.import_module("test_engine_gpu")
synth = conftest.load_pkg("synth")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
name, mode = "step_simple_B4_S32", "fp32"
worst = []
for r in range(reps):
    fix, eng, (seed, B, S, n_critic, iters, gray) = teg.make_engine(synth, name, mode)
    inp = teg.inputs_for(synth, name, seed, 0, B, S, n_critic, gray)
    refined = [T(x).cuda() for x in inp["refined"]]
    eng.lr = 0.0
    eng.d_step(T(inp["pred"]).cuda(), T(inp["gt"]).cuda(), lambda d, k: refined[k], 0, T(inp["alpha"][0]).cuda().view(-1).contiguous(),
               [T(m).cuda() for m in inp["masks"][0]])
    torch.cuda.synchronize()
    coef = min(1.0, 1.0 / (float(eng.D.state[2]) + 1e-6))
    g = eng.D.gviews["model.0.weight_orig"].cpu().numpy().reshape(-1) / coef
    b = np.asarray(fix["it0.c0.dgrad.model.0.weight_orig"], dtype=np.float64).reshape(-1)
    worst.append(float(np.abs(g - b).max() / np.abs(b).max()))
print("worst element error / scale per run:", " ".join(f"{w:.1e}" for w in worst))
print(f"second mode (> 5e-4): {sum(w > 5e-4 for w in worst)} of {reps}")
