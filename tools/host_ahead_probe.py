"""Is the host ahead of the GPU in the two-stream replay loop?  Times every host call of GraphedIteration.replay (no profiler)
and the final synchronize: a host that runs ahead leaves a long synchronize and short calls; a host that some call blocks leaves
the GPU waiting for its next graph launch.
usage: python tools/host_ahead_probe.py [iterations] [order]   order: ga_first (the product's) | ca_first"""
import importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
PKG = bench.PKG
engine = importlib.import_module(PKG + ".engine"); synth = importlib.import_module(PKG + ".synth"); dist_mod = importlib.import_module(PKG + ".dist")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
order = sys.argv[2] if len(sys.argv) > 2 else "ga_first"
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
run = bench.Runner(engine, synth, dist_mod, dev, 0, 1, 256, 32, 2, "bf16", "unet", 3)
gi = run.graphed
assert gi.two_stream and not gi.head_split, "this probe replays the four-graph form by hand (GCSSL_HEAD_SPLIT must be off)"
for _ in range(30): gi.replay()
torch.cuda.synchronize()
main, side = torch.cuda.current_stream(), gi.side
names = ["ev0+wait", "g_a", "c_a", "g_b", "c_b", "tail"]
acc = [0.0] * len(names); mx = [0.0] * len(names)
pc = time.perf_counter
t_begin = pc()
for it in range(iters):
    t = [pc()]
    ev0 = torch.cuda.Event(); ev0.record(main); side.wait_event(ev0); t.append(pc())
    if order == "ga_first":
        with torch.cuda.stream(side):
            gi.g_a.replay(); ev_ga = torch.cuda.Event(); ev_ga.record(side)
        t.append(pc())
        gi.c_a.replay(); ev_ca = torch.cuda.Event(); ev_ca.record(main); t.append(pc())
    else:
        gi.c_a.replay(); ev_ca = torch.cuda.Event(); ev_ca.record(main); t.append(pc())
        with torch.cuda.stream(side):
            gi.g_a.replay(); ev_ga = torch.cuda.Event(); ev_ga.record(side)
        t.append(pc())
        t[2], t[3] = t[3], t[2]                                     # (column order stays g_a, c_a: durations below use neighbours)
    with torch.cuda.stream(side):
        side.wait_event(ev_ca); gi.g_b.replay(); ev_gb = torch.cuda.Event(); ev_gb.record(side)
    t.append(pc())
    main.wait_event(ev_ga); gi.c_b.replay(); t.append(pc())
    main.wait_event(ev_gb); t.append(pc())
    if order == "ga_first":
        d = [t[i + 1] - t[i] for i in range(6)]
    else:
        d = [t[1] - t[0], t[2] - t[3], t[3] - t[1], t[4] - t[2], t[5] - t[4], t[6] - t[5]]
    for i, x in enumerate(d):
        acc[i] += x; mx[i] = max(mx[i], x)
t_host = pc() - t_begin
torch.cuda.synchronize()
t_all = pc() - t_begin
print(f"order {order}: {iters} iterations, host loop {t_host / iters * 1e6:.1f} us/iteration, with the final synchronize {t_all / iters * 1e6:.1f} us/iteration "
      f"(synchronize waited {(t_all - t_host) * 1e3:.2f} ms = {(t_all - t_host) / (t_all / iters):.1f} iterations of queued work)")
for n, a, m in zip(names, acc, mx):
    print(f"  {n:9s} avg {a / iters * 1e6:8.1f} us   max {m * 1e6:8.1f} us")
