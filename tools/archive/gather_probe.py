"""Is a conv launch bound by WHERE its operand gathers hit?  (experiment tool, not part of the product path)

Builds two copies of the library into /tmp -- the shipped sources, and one in which every A-operand row of the LDS-DMA
conv kernels gathers from the SAME sample (n forced to 0 in the row offset: identical instruction stream, identical
number of gathered bytes, but a footprint of one sample that stays in the nearest cache) -- and times one conv
configuration with both.  Results of the second build are garbage; only its time matters.
usage: python tools/archive/gather_probe.py {fwd|dgrad|c3} N Hi Cin Cout [reps]"""
import re
import subprocess
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "gan-calibrated-semi-supervised-learning_amd"
FILES = ("igemm.hip", "norm.hip", "misc.hip", "recrop.hip", "simple_gen.hip")


def build(tag: str, flat: bool) -> Path:
    out = Path(f"/tmp/gcssl_probe_{tag}")
    out.mkdir(exist_ok=True)
    for f in FILES + ("common.h",):
        s = (PKG / "csrc" / f).read_text()
        if flat and f == "igemm.hip":
            s, k = re.subn(r"rowoff\[i\] = \(\(n \* ", "rowoff[i] = ((0 * ", s)
            assert k == 8, k
        (out / f).write_text(s)
    so = out / "libgcssl_probe.so"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           "-o", str(so)] + [str(out / f) for f in FILES]
    subprocess.run(cmd, check=True)
    return so


def main():
    kind, N, Hi, Cin, Cout = sys.argv[1], *map(int, sys.argv[2:6])
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 30
    sos = {"shipped": build("a", False), "one-sample gathers": build("b", True)}
    child = len(sys.argv) > 7
    if not child:                                                     # one process per build: the library binds once
        for name, so in sos.items():
            r = subprocess.run([sys.executable, __file__] + sys.argv[1:6] + [str(reps), str(so)], capture_output=True, text=True)
            print(f"{name:20s} {r.stdout.strip()} {r.stderr.strip()[-200:] if r.returncode else ''}")
        return
    import os
    os.environ["GCSSL_LIB"] = sys.argv[7]
    sys.path.insert(0, str(ROOT))
    import importlib
    ops = importlib.import_module(PKG.name + ".ops")
    dt = torch.bfloat16
    if kind == "c3":
        x = (torch.rand(N, Hi, Hi, Cin, device="cuda") * 2 - 1).to(dt)
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        wf = torch.empty(Cout, ops.conv3_wk(Cin), device="cuda", dtype=dt)
        ops.Prep3Batch([(w, wf, None, Cout, Cin, Cin)], ops.code(wf)).run()
        y = torch.empty(N, Hi, Hi, Cout, device="cuda", dtype=torch.float32)
        run = lambda: ops.conv3_fwd(x, wf, y, Cin, Cout)
        flops = 2.0 * N * Hi * Hi * Cout * 9 * Cin
    else:
        x = (torch.rand(N, Hi, Hi, Cin, device="cuda") * 2 - 1).to(dt)
        dy = (torch.rand(N, Hi // 2, Hi // 2, Cout, device="cuda") * 2 - 1).to(dt)
        w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.05
        wf = torch.empty(Cout, 16, Cin, device="cuda", dtype=dt)
        wt = torch.empty(Cin, 16, Cout, device="cuda", dtype=dt)
        ops.prep_conv_weight(w, wf, wt, Cout, Cin, Cin, ops.code(wf))
        y = torch.empty(N, Hi // 2, Hi // 2, Cout, device="cuda", dtype=torch.float32)
        dx = torch.empty(N, Hi, Hi, Cin, device="cuda", dtype=torch.float32)
        run = (lambda: ops.conv_fwd(x, wf, y, Cin, Cout)) if kind == "fwd" else (lambda: ops.conv_dgrad(dy, wt, dx, Cin, Cout))
        flops = 2.0 * N * (Hi // 2) ** 2 * Cout * 16 * Cin
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{kind} N={N} Hi={Hi} Cin={Cin} Cout={Cout}: {us:7.1f} us  {flops / us / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
