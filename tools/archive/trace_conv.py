"""Per-workgroup phase timing of conv_dma_kernel with s_memtime (experiment tool, not part of the product path).

Builds an INSTRUMENTED copy of csrc/igemm.hip into /tmp (the shipped library is untouched), runs one conv
configuration and prints, per phase, the mean / p50 / p95 shader cycles over the workgroups of the last launch:
  setup   kernel entry -> gather offsets ready        first   -> first K tile landed (DMA latency, pipeline fill)
  loop    remaining K loop                            epi     epilogue (stores / atomics)
usage: python tools/archive/trace_conv.py {fwd|dgrad} N Hi Cin Cout [reps]      (env knobs GCSSL_* apply as usual)"""
import ctypes
import os
import re
import subprocess
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "gan-calibrated-semi-supervised-learning_amd"
SRC = PKG / "csrc" / "igemm.hip"


def instrumented_source() -> str:
    s = SRC.read_text()
    s = s.replace('#include "common.h"', '#include "common.h"\n__device__ unsigned long long* g_trace_ptr;\n'
                  '#define TRACE(k) do { if (g_trace_ptr && threadIdx.x == 0) g_trace_ptr[((blockIdx.z * gridDim.y + blockIdx.y) * '
                  '(size_t)gridDim.x + blockIdx.x) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)\n'
                  '#define TRACE_ADD(k, v) do { if (g_trace_ptr && threadIdx.x == 0) g_trace_ptr[((blockIdx.z * gridDim.y + blockIdx.y) * '
                  '(size_t)gridDim.x + blockIdx.x) * 16 + (k)] += (v); } while (0)\n', 1)
    a = s.index("void conv_dma_kernel(ConvParams p) {")
    b = s.index("#endif", a)
    body = s[a:b]
    body = body.replace("    typedef bf16_t T;\n", "    typedef bf16_t T;\n    TRACE(0);\n", 1)
    body = body.replace("    const int nk_all = K / BK;\n", "    TRACE(1);\n    const int nk_all = K / BK;\n", 1)
    # per-step split for wave 0: [8] waiting for the tile (s_waitcnt), [9] barrier, [10] DMA issue, [11] LDS reads + MFMA
    body = body.replace("            if (t + 1 < t_end) asm volatile(\"s_waitcnt vmcnt(%0)\" ::\"n\"(NL) : \"memory\");",
                        "            const unsigned long long ta = __builtin_amdgcn_s_memtime();\n"
                        "            if (t + 1 < t_end) asm volatile(\"s_waitcnt vmcnt(%0)\" ::\"n\"(NL) : \"memory\");", 1)
    body = body.replace("            __builtin_amdgcn_s_barrier();          // everyone's part of tile t landed; everyone is done reading tile t-1\n",
                        "            const unsigned long long tb = __builtin_amdgcn_s_memtime();\n"
                        "            __builtin_amdgcn_s_barrier();\n", 1)
    body = body.replace("            __builtin_amdgcn_sched_barrier(0);\n",
                        "            __builtin_amdgcn_sched_barrier(0);\n            if (t == t_beg) TRACE(2);\n"
                        "            const unsigned long long tc = __builtin_amdgcn_s_memtime();\n", 1)
    body = body.replace("            const unsigned char* At = lds + slot * STAGE;\n",
                        "            __builtin_amdgcn_sched_barrier(0);\n            const unsigned long long td = __builtin_amdgcn_s_memtime();\n"
                        "            const unsigned char* At = lds + slot * STAGE;\n", 1)
    body = body.replace("            slot = slot == 2 ? 0 : slot + 1;\n",
                        "            __builtin_amdgcn_sched_barrier(0);\n            const unsigned long long te = __builtin_amdgcn_s_memtime();\n"
                        "            TRACE_ADD(8, tb - ta); TRACE_ADD(9, tc - tb); TRACE_ADD(10, td - tc); TRACE_ADD(11, te - td);\n"
                        "            slot = slot == 2 ? 0 : slot + 1;\n", 1)
    body = body.replace("    // ---- epilogue.", "    TRACE(3);\n    // ---- epilogue.", 1)
    body = body.rstrip() + "\n    __builtin_amdgcn_s_waitcnt(0); TRACE(4);\n    if (threadIdx.x == 0 && g_trace_ptr) { unsigned id; "
    body += "asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\" : \"=s\"(id)); g_trace_ptr[((blockIdx.z * gridDim.y + blockIdx.y) * (size_t)gridDim.x + blockIdx.x) * 16 + 5] = id; }\n"
    s = s[:a] + body + s[b:]
    s += '\nextern "C" int gcssl_set_trace(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_ptr), &p, sizeof(p)); }\n'
    assert s.count("TRACE(") >= 6 and s.count("TRACE_ADD(") >= 5
    return s


def build() -> Path:
    out = Path("/tmp/gcssl_trace")
    out.mkdir(exist_ok=True)
    (out / "igemm.hip").write_text(instrumented_source())
    for f in ("common.h", "norm.hip", "misc.hip", "recrop.hip", "simple_gen.hip"):
        (out / f).write_text((PKG / "csrc" / f).read_text())
    so = out / "libgcssl_trace.so"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value",
           "-o", str(so)] + [str(out / f) for f in ("igemm.hip", "norm.hip", "misc.hip", "recrop.hip", "simple_gen.hip")]
    subprocess.run(cmd, check=True)
    return so


def main():
    kind, N, Hi, Cin, Cout = sys.argv[1], *map(int, sys.argv[2:6])
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    so = build()
    sys.path.insert(0, str(ROOT))
    import importlib
    _lib = importlib.import_module(PKG.name + "._lib")
    _lib.LIB_PATH = so                                   # bind the instrumented build instead of the shipped one
    ops = importlib.import_module(PKG.name + ".ops")
    lib = _lib.lib()
    dt = torch.bfloat16
    x = (torch.rand(N, Hi, Hi, Cin, device="cuda") * 2 - 1).to(dt)
    dy = (torch.rand(N, Hi // 2, Hi // 2, Cout, device="cuda") * 2 - 1).to(dt)
    w = torch.randn(Cout, Cin, 4, 4, device="cuda") * 0.05
    wf = torch.empty(Cout, 16, Cin, device="cuda", dtype=dt)
    wt = torch.empty(Cin, 16, Cout, device="cuda", dtype=dt)
    ops.prep_conv_weight(w, wf, wt, Cout, Cin, Cin, ops.code(wf))
    y = torch.empty(N, Hi // 2, Hi // 2, Cout, device="cuda", dtype=torch.float32)
    dx = torch.empty(N, Hi, Hi, Cin, device="cuda", dtype=torch.float32)
    trace = torch.zeros(1 << 20, device="cuda", dtype=torch.int64)
    run = (lambda: ops.conv_fwd(x, wf, y, Cin, Cout)) if kind == "fwd" else (lambda: ops.conv_dgrad(dy, wt, dx, Cin, Cout))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    print(f"{kind} N={N} Hi={Hi} Cin={Cin} Cout={Cout}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us/launch (untraced)")
    ctypes.CDLL(str(so)).gcssl_set_trace(ctypes.c_void_p(trace.data_ptr()))
    run()
    torch.cuda.synchronize()
    t = trace.view(-1, 16).cpu()
    t = t[t[:, 0] != 0]
    n = t.shape[0]
    t0 = t[:, 0].min()
    span = int(t[:, 4].max() - t0)
    names = ["setup", "first", "loop", "epi"]
    print(f"workgroups {n}  kernel span {span} ticks   (s_memtime ticks; 100 MHz => 10 ns each if span ~ us*100)")
    for k, nm in enumerate(names):
        d = (t[:, k + 1] - t[:, k]).double()
        print(f"  {nm:6s} mean {d.mean():9.0f}  p50 {d.median():9.0f}  p95 {d.quantile(0.95):9.0f}")
    for k, nm in ((8, "wait"), (9, "barrier"), (10, "issue"), (11, "mfma")):
        d = t[:, k].double()
        print(f"  loop/{nm:8s} mean {d.mean():9.0f}  p50 {d.median():9.0f}  (summed over the K loop, wave 0)")
    life = (t[:, 4] - t[:, 0]).double()
    print(f"  life   mean {life.mean():9.0f}  p50 {life.median():9.0f}  p95 {life.quantile(0.95):9.0f}")
    start = (t[:, 0] - t0).double()
    qs = [0.1, 0.25, 0.5, 0.75, 0.9, 1.0]
    print("  start-time quantiles (ticks since first WG):", [int(start.quantile(q)) for q in qs])
    hw = t[:, 5]
    cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4)          # CU_ID [11:8], SE_ID [15:13] (gfx9 HW_ID layout)
    print("  distinct (se,cu) ids seen:", len(torch.unique(cu)))
    # concurrency: average number of workgroups alive
    print(f"  mean WGs alive = {float(life.sum()) / span:.1f}")


if __name__ == "__main__":
    main()
