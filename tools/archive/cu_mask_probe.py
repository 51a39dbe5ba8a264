"""Do CU-masked streams let the critic's and the generator's chains run side by side?  Both chains free-running (no events:
timing only, results are garbage) on two streams created with hipExtStreamCreateWithCUMask.
usage: python tools/archive/cu_mask_probe.py"""
import ctypes, importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
PKG = bench.PKG
engine = importlib.import_module(PKG + ".engine"); synth = importlib.import_module(PKG + ".synth"); dist_mod = importlib.import_module(PKG + ".dist")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[32 * w + b]) for w in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


run = bench.Runner(engine, synth, dist_mod, dev, 0, 1, 256, 32, 2, "bf16", "unet", 3)
gi = run.graphed
assert gi.two_stream
for _ in range(20): gi.replay()
torch.cuda.synchronize()


def timed(sc, sg, iters=150, what="both"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        if what in ("both", "g"):
            with torch.cuda.stream(sg):
                gi.g_a.replay(); gi.g_b.replay()
        if what in ("both", "c"):
            with torch.cuda.stream(sc):
                gi.c_a.replay(); gi.c_b.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


plain_c, plain_g = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
print(f"unmasked: critic alone {timed(plain_c, plain_g, what='c'):.0f} us, generator alone {timed(plain_c, plain_g, what='g'):.0f} us, "
      f"both {timed(plain_c, plain_g):.0f} us", flush=True)
for name, fn in (("low bits", lambda i, k: i < 32 * k), ("interleaved (bit % 8)", lambda i, k: (i % 8) < k)):
    for k in (4, 5, 6):                                   # critic gets k/8 of the mask bits, the generator the rest
        bc = [fn(i, k) for i in range(256)]
        sc, sg = masked_stream(bc), masked_stream([not b for b in bc])
        print(f"{name}: critic {sum(bc)} CUs / generator {256 - sum(bc)} CUs: critic alone {timed(sc, sg, what='c'):.0f} us, "
              f"generator alone {timed(sc, sg, what='g'):.0f} us, both {timed(sc, sg):.0f} us", flush=True)
