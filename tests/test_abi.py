"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/gcssl.h declares.
No compute calls are made here (no GPU)."""
import ctypes
import re
import subprocess

import pytest

from conftest import ROOT, load_pkg


@pytest.fixture(scope="module")
def lib_mod():
    m = load_pkg("_lib")
    m.build()
    return m


def test_header_declares_the_expected_surface(lib_mod):
    protos = lib_mod.parse_header()
    for name in ["gcssl_conv4x4s2_fwd", "gcssl_conv4x4s2_dgrad", "gcssl_conv4x4s2_wgrad", "gcssl_wgrad_reduce",
                 "gcssl_conv4x4s1_c1_fwd", "gcssl_conv4x4s1_c1_dgrad", "gcssl_conv4x4s1_c1_wgrad",
                 "gcssl_in_act_fwd", "gcssl_in_act_bwd", "gcssl_in_dbl_bwd", "gcssl_act_bwd",
                 "gcssl_sn_power_iter", "gcssl_gp_norm", "gcssl_clip_adam", "gcssl_pool_fc_tanh_fwd",
                 "gcssl_head_bwd", "gcssl_eiou_fwd_bwd", "gcssl_dropout_mask_gen", "gcssl_pack_pair",
                 "gcssl_pack_interp", "gcssl_unpack_grad", "gcssl_prep_conv_weight"]:
        assert name in protos, name


def test_every_declared_symbol_is_exported(lib_mod):
    handle = lib_mod.lib()
    protos = lib_mod.parse_header()
    for name, (_, argtypes) in protos.items():
        fn = getattr(handle, name)
        assert fn.argtypes == argtypes
    assert handle.gcssl_version().decode().startswith("gcssl-hip")
    # and nothing exported is undeclared
    out = subprocess.run(["nm", "-D", "--defined-only", str(lib_mod.LIB_PATH)], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\b(gcssl_\w+)\b", out))
    assert exported == set(protos), exported ^ set(protos)


def test_argument_validation_without_gpu(lib_mod):
    """Pure host-side checks return negative codes before any launch (safe on a CPU-only box)."""
    h = lib_mod.lib()
    assert h.gcssl_conv4x4s2_fwd(0, None, 8, None, None, None, 0, None, 64, 1, 8, 8, 8, 64, 0, 0, 0, None) == -4
    assert h.gcssl_conv4x4s2_fwd_splits(1, 256, 4, 4, 256, 512, 0, 1) >= 1     # dry run of the dispatcher: host only
    assert h.gcssl_conv4x4s2_dgrad_splits(1, 256, 4, 4, 256, 512, 1) >= 1
    assert h.gcssl_conv4x4s2_fwd_splits(1, 256, 12, 12, 256, 512, 0, 1) == -1  # non power-of-two spatial size
    assert h.gcssl_recrop_ws_ints(256, 32, 1280) > 0 and h.gcssl_recrop_ws_ints(0, 32, 1280) == -1
    assert h.gcssl_sum_replicas(1, None, None, None, 4, 64, 0, None) == -4
    assert h.gcssl_conv4x4s2_wgrad_splits(4, 12, 12, 64, 64) == -1          # non power-of-two spatial size
    assert h.gcssl_conv4x4s2_wgrad_splits(256, 16, 16, 64, 128) > 0
    assert h.gcssl_gp_norm(None, 10, 2, ctypes.c_float(1.0), None, None, None, 0, None, None, None) == -4
    assert h.gcssl_conv4x4s2_in_act_ok(0, 768, 8, 8, 128, 256) == 0          # fp32: the parity mode keeps conv -> z -> norm
    assert h.gcssl_conv4x4s2_in_act_ok(2, 768, 8, 8, 128, 256) == 1          # D.c3 at the bench batch, fp16: the fused launch
    assert h.gcssl_conv4x4s2_in_act_ok(2, 384, 32, 32, 64, 128) == 0         # 16x16 maps: a sample does not fit a tile
    assert h.gcssl_conv4x4s2_in_act_fwd(2, None, 64, None, None, None, 0, None, 64, None, None, None, None, 0, 0, 4, 8, 8, 64, 64, 1, None) == -4
    assert h.gcssl_conv4x4s2_fwd_act_bwd_ok(2, 256, 32, 32, 8, 64) == 1          # D.c1 of the reverse GP chain
    assert h.gcssl_conv4x4s2_fwd_act_bwd_ok(0, 256, 32, 32, 8, 64) == 0
    assert h.gcssl_conv4x4s2_fwd_act_bwd_ok(2, 256, 16, 16, 64, 128) == 0
    assert h.gcssl_conv4x4s2_fwd_act_bwd(2, None, 8, None, None, 0, None, 64, None, 64, None, 0, None, None, 4, 32, 32, 8, 64, None) == -4
    assert h.gcssl_last_kernel() is not None


def test_code_object_is_gfx950(lib_mod):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", str(lib_mod.LIB_PATH)],
                         capture_output=True, text=True).stdout
    blob = lib_mod.LIB_PATH.read_bytes()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"gfx90a" not in blob, out[:200]
