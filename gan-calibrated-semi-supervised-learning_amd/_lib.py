"""ctypes binding of libgcssl_hip.so (the C ABI declared in include/gcssl.h).

The prototypes are parsed from the header itself, so Python can never drift from the ABI.  There is NO
fallback: if the library is missing or a call fails, a RuntimeError is raised (the product path never
routes through the CPU oracle).
"""
from __future__ import annotations

import ctypes
import os
import re
import subprocess
from pathlib import Path

import torch

PKG_DIR = Path(__file__).resolve().parent
ROOT = PKG_DIR.parent
HEADER = ROOT / "include" / "gcssl.h"
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libgcssl_hip.so"
SOURCES = ["igemm.hip", "norm.hip", "misc.hip", "recrop.hip", "simple_gen.hip", "convt_fused.hip"]

F32, BF16, F16 = 0, 1, 2
# split-precision conv modes (csrc/common.h): fp32 tensors, operands split hi + lo into 16-bit halves inside the conv kernels,
# three 16-bit MFMAs per K step.  Conv entry points only; everything else is called with F32.
F32_F16X3, F32_BF16X3 = 3, 4
SPLIT_MODES = {"fp16x3": F32_F16X3, "bf16x3": F32_BF16X3}
ERRORS = {-1: "GCSSL_EBADSHAPE", -2: "GCSSL_EBADDTYPE", -3: "GCSSL_EALIGN", -4: "GCSSL_ENULL"}

_CT = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
       "unsigned long long": ctypes.c_ulonglong}


def parse_header(path: Path = HEADER):
    """-> {name: (restype, [argtypes])} for every `gcssl_*` function declared in the header."""
    text = re.sub(r"/\*.*?\*/", "", path.read_text(), flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|int)\s+(gcssl_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = re.sub(r"\s+\w+$", "", a).replace("const ", "").strip()
                    argtypes.append(_CT[ty])
        protos[name] = (ctypes.c_char_p if ret.startswith("const char") else ctypes.c_int, argtypes)
    return protos


def abi_revision_of_header(path: Path = HEADER) -> int:
    m = re.search(r"#define\s+GCSSL_ABI_REVISION\s+(\d+)", path.read_text())
    if not m:
        raise RuntimeError(f"{path} does not define GCSSL_ABI_REVISION")
    return int(m.group(1))


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP sources for gfx950 into the in-tree shared library (hipcc cross-compiles without a GPU).
    One object per source, compiled in parallel (igemm.hip alone is most of the time), then linked."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [CSRC / s for s in SOURCES]
    hdr = CSRC / "common.h"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = PKG_DIR / "build"
    objdir.mkdir(exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]

    def compile_one(src: Path):
        obj = objdir / (src.stem + ".o")
        if not force and obj.exists() and obj.stat().st_mtime >= max(src.stat().st_mtime, hdr.stat().st_mtime):
            return obj, False
        cmd = [hipcc] + flags + ["-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src.name}:\n{r.stderr}")
        return obj, True
    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
        res = list(ex.map(compile_one, srcs))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not LIB_PATH.exists() or any(LIB_PATH.stat().st_mtime < o.stat().st_mtime for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH)] + [str(o) for o in objs]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB_PATH


_lib = None
_protos = None


def lib():
    global _lib, _protos
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               f"(there is no CPU fallback for the HIP path)")
        # GCSSL_LIB: load another build of the same ABI (same-box A/B of two builds; experiments only)
        _lib = ctypes.CDLL(os.environ.get("GCSSL_LIB") or str(LIB_PATH))
        _protos = parse_header()
        for name, (ret, argtypes) in _protos.items():
            fn = getattr(_lib, name)          # AttributeError here == header/ABI mismatch: fail loudly
            fn.restype, fn.argtypes = ret, argtypes
        want = abi_revision_of_header()
        have = _lib.gcssl_abi_revision()
        if have != want:
            raise RuntimeError(f"libgcssl_hip.so implements ABI revision {have}, include/gcssl.h declares {want}: rebuild "
                               f"(python -c 'import __graft_entry__ as g; g.build()')")
    return _lib


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _conv(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    return a


def call(name: str, *args):
    """Call an int-returning entry point with tensors/None/scalars; the current torch stream is appended."""
    fn = getattr(lib(), name)
    rc = fn(*[_conv(a) for a in args], stream_ptr())
    if rc != 0:
        raise RuntimeError(f"{name} failed: {ERRORS.get(rc, 'hipError ' + str(rc))}")


def call_nostream(name: str, *args):
    return getattr(lib(), name)(*[_conv(a) for a in args])


def ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def int_array(vals):
    return (ctypes.c_int * len(vals))(*vals)


def dtype_code(dt) -> int:
    if dt in (F32, "fp32", "f32", torch.float32):
        return F32
    if dt in (BF16, "bf16", torch.bfloat16):
        return BF16
    if dt in (F16, "fp16", "f16", torch.float16):
        return F16
    if dt in SPLIT_MODES or dt in SPLIT_MODES.values():
        return F32                            # storage / non-conv dtype of the split-precision modes (mma_code gives the conv code)
    raise ValueError(f"unsupported compute dtype {dt!r}")


def mma_code(dt) -> int:
    """dtype code the CONV entry points get for compute mode `dt`: the split-precision code for "fp16x3" / "bf16x3",
    dtype_code(dt) otherwise."""
    if dt in SPLIT_MODES:
        return SPLIT_MODES[dt]
    if dt in SPLIT_MODES.values():
        return dt
    return dtype_code(dt)


def torch_dtype(code: int):
    return {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[code]


def dtype_name(code: int) -> str:
    return {F32: "fp32", BF16: "bf16", F16: "fp16", F32_F16X3: "fp16x3", F32_BF16X3: "bf16x3"}[code]
