"""GPU-side gaps between graph launches (run under rocprofv3 --kernel-trace, then tools/prof_timeline.py).
usage: python tools/graph_gap_probe.py {critic_only|critic_fused|critic_steps|gen_only|critic_first|full|no_events} [iterations] [dtype]"""
import importlib, os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
PKG = bench.PKG
engine = importlib.import_module(PKG + ".engine"); synth = importlib.import_module(PKG + ".synth"); dist_mod = importlib.import_module(PKG + ".dist")
mode = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
run = bench.Runner(engine, synth, dist_mod, dev, 0, 1, 256, 32, 2, dtype, "unet", 3)
gi = run.graphed
assert gi.two_stream
if mode in ("critic_fused", "critic_steps"):
    # the critic's chain as ONE graph (critic_fused) / as one graph per critic step + the value-only forward (critic_steps): what
    # does a graph boundary cost on the GPU side?  (timing only)
    eng = gi.eng
    pred, gt, delta_true, pred_box, refine = run.call
    eng._in_g_branch = True

    def cap(fn):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            fn()
        return g

    def step(k):
        eng._d_dirty = True
        eng.d_pre(pred, gt, refine, k, None, None); eng.d_main(); eng.d_update()
    if mode == "critic_fused":
        extra = [cap(lambda: (step(0), step(1), eng.g_critic(pred)))]
    else:
        extra = [cap(lambda: step(0)), cap(lambda: step(1)), cap(lambda: eng.g_critic(pred))]
    eng._in_g_branch = False
for _ in range(20): gi.replay()
torch.cuda.synchronize()
main, side = torch.cuda.current_stream(), gi.side
hs = getattr(gi, "head_split", False)


def ca():                                                       # the critic's first segment (with its head graph, if split off)
    if hs: gi.c_a0.replay()
    (gi.c_a_g if getattr(gi, "batch_g", False) else gi.c_a).replay()      # (with the value-only forward's group where the iteration has it)
    if hs: gi.c_b0.replay()


t0 = time.perf_counter()
for _ in range(iters):
    if mode == "critic_only":
        ca(); gi.c_b.replay()
    elif mode in ("critic_fused", "critic_steps"):
        for g in extra: g.replay()
    elif mode == "gen_only":                       # the generator's chain alone (results are garbage after the first: timing only)
        gi.g_a.replay(); gi.g_b.replay()
    elif mode == "full":
        gi.replay()
    elif mode == "no_events":                      # both chains free-running (results are garbage: timing only)
        with torch.cuda.stream(side):
            gi.g_a.replay(); gi.g_b.replay()
        ca(); gi.c_b.replay()
    elif mode == "critic_first":
        ev0 = torch.cuda.Event(); ev0.record(main)
        ca()
        ev_ca = torch.cuda.Event(); ev_ca.record(main)
        side.wait_event(ev0)
        with torch.cuda.stream(side):
            gi.g_a.replay()
            ev_ga = torch.cuda.Event(); ev_ga.record(side)
            side.wait_event(ev_ca)
            gi.g_b.replay()
            ev_gb = torch.cuda.Event(); ev_gb.record(side)
        main.wait_event(ev_ga)
        gi.c_b.replay()
        main.wait_event(ev_gb)
torch.cuda.synchronize()
print(f"{mode}: {(time.perf_counter() - t0) / iters * 1e6:.1f} us per iteration")
