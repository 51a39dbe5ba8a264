"""Host-side cost of the graph launches of one iteration (two-stream form): wall-clock time of every replay() call and of
the whole iteration, GPU idle (synchronised) vs GPU busy (host running ahead).
usage (GPU box): python tools/archive/host_launch_probe.py"""
import importlib, os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
PKG = bench.PKG
engine = importlib.import_module(PKG + ".engine"); synth = importlib.import_module(PKG + ".synth"); dist_mod = importlib.import_module(PKG + ".dist")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
run = bench.Runner(engine, synth, dist_mod, dev, 0, 1, 256, 32, 2, "bf16", "unet", 3)
gi = run.graphed
for _ in range(50): gi.replay()
torch.cuda.synchronize()
if not getattr(gi, "two_stream", False):
    for mode in ("idle", "busy"):
        ts = []
        for _ in range(20):
            if mode == "idle": torch.cuda.synchronize()
            t0 = time.perf_counter(); gi.replay(); ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        print(f"one graph, GPU {mode}: replay() host time us: " + " ".join(f"{t * 1e6:.0f}" for t in ts))
    sys.exit(0)
main, side = torch.cuda.current_stream(), gi.side
def one(sync):
    out = {}
    def t(name, fn):
        t0 = time.perf_counter(); fn(); out[name] = out.get(name, 0.0) + (time.perf_counter() - t0) * 1e6
    if sync: torch.cuda.synchronize()
    ev0 = torch.cuda.Event(); t("ev", lambda: (ev0.record(main), side.wait_event(ev0)))
    with torch.cuda.stream(side):
        t("g_a", gi.g_a.replay)
        ev_ga = torch.cuda.Event(); t("ev", lambda: ev_ga.record(side))
    t("c_a", gi.c_a.replay)
    ev_ca = torch.cuda.Event(); t("ev", lambda: ev_ca.record(main))
    with torch.cuda.stream(side):
        t("ev", lambda: side.wait_event(ev_ca))
        t("g_b", gi.g_b.replay)
        ev_gb = torch.cuda.Event(); t("ev", lambda: ev_gb.record(side))
    t("ev", lambda: main.wait_event(ev_ga))
    t("c_b", gi.c_b.replay)
    t("ev", lambda: main.wait_event(ev_gb))
    return out
for mode in ("idle", "busy"):
    for i in range(12):
        o = one(mode == "idle")
        print(mode, " ".join(f"{k} {v:6.0f}" for k, v in o.items()), f" total {sum(o.values()):6.0f} us")
    torch.cuda.synchronize()
