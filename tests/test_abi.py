"""CPU: the C-ABI shared library loads, exports every symbol include/dsmnet_hip.h
declares, and rejects bad arguments with error codes (no kernel is launched)."""
import ctypes
import os
import re

import pytest
import torch

from dsmnet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dsmnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsm_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    syms = declared_symbols()
    assert "dsm_conv3d_fwd" in syms and "dsm_concat_volume_fwd" in syms
    assert sorted(_lib.SIGNATURES) == syms


def test_library_exports_every_declared_symbol(hip_lib):
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(raw, name), "missing export: " + name
    assert hip_lib.dsm_abi_version() == 7
    assert hip_lib.dsm_strerror(0) == b"ok"
    assert b"not supported" in hip_lib.dsm_strerror(-2)


def test_argument_validation_returns_codes(hip_lib):
    null = None
    one = ctypes.c_void_p(16)
    assert hip_lib.dsm_corr1d_fwd(null, null, null, null, 1, 1, 1, 1, 1, 1, 1, 0, null) == -1
    assert hip_lib.dsm_corr1d_fwd(one, one, one, null, 1, 8, 4, 4, 4, 1, 2, 0, null) == -1  # even k
    assert hip_lib.dsm_corr1d_fwd(one, one, one, null, 1, 8, 4, 4, 4, 1, 3, 0, null) == -1  # no tmp
    assert hip_lib.dsm_corr1d_fwd(one, one, one, null, 1, 8, 4, 4, 4, 1, 1, 7, null) == -2  # dtype
    assert hip_lib.dsm_concat_volume_fwd(one, one, one, 1, 8, 4, 4, 0, 1, 1, 0, null) == -1  # D=0
    assert hip_lib.dsm_concat_volume_fwd(one, one, one, 1, 8, 4, 4, 4, 1, 5, 0, null) == -1  # layout
    assert hip_lib.dsm_concat_volume_fwd(one, one, one, 1, 6, 4, 4, 4, 1, 1, 0, null) == -2  # C%4
    assert hip_lib.dsm_soft_argmin_fwd(null, one, null, 1, 4, 4, 4, 4, 4, 4, 0, 0, 0, null) == -1
    a = _lib.Conv3dArgs()
    assert hip_lib.dsm_conv3d_fwd(ctypes.byref(a), null) == -1
    a.x = a.w_packed = a.y = 16
    a.B, a.Cin, a.Cout = 1, 32, 48
    a.Di = a.Hi = a.Wi = a.Do = a.Ho = a.Wo = 4
    a.stride = 1
    assert hip_lib.dsm_conv3d_fwd(ctypes.byref(a), null) == -2          # Cout not 1/32/64/128
    a.Cout, a.Do = 32, 5
    assert hip_lib.dsm_conv3d_fwd(ctypes.byref(a), null) == -1          # output larger than natural
    a.Do, a.x = 4, 20
    assert hip_lib.dsm_conv3d_fwd(ctypes.byref(a), null) == -4          # misaligned
    assert hip_lib.dsm_conv3d_packed_weight_bytes(32, 32, 0) == 32 * 32 * 27 * (4 + 6 + 4) + 16   # fp32 fragments + 3 bf16 planes + header, 2 fp16 planes
    assert hip_lib.dsm_conv3d_packed_weight_bytes(128, 128, 0) == 128 * 128 * 27 * (4 + 6 + 4) + 16   # (r03: four columns of 32 on small volumes)
    assert hip_lib.dsm_conv_packed_weight_bytes(128, 128, 1, 3) == 128 * 128 * 9 * (4 + 6 + 4) + 16
    assert hip_lib.dsm_conv_packed_weight_bytes(128, 32, 1, 1) == 128 * 32 * 4
    # ABI v5: weight gradients (flags argument), the 2-D entry point
    assert hip_lib.dsm_conv3d_wgrad(null, one, one, one, 1, 32, 32, 4, 4, 4, 4, 4, 4, 1, 0, 0, null, null, null) == -1
    assert hip_lib.dsm_conv3d_wgrad(one, one, one, one, 1, 32, 48, 4, 4, 4, 4, 4, 4, 1, 0, 0, null, null, null) == -2   # Cg % 32
    assert hip_lib.dsm_conv3d_wgrad(one, one, one, one, 1, 32, 32, 4, 4, 4, 4, 4, 4, 3, 0, 0, null, null, null) == -2   # stride
    assert hip_lib.dsm_conv3d_wgrad(one, one, one, one, 1, 32, 32, 4, 4, 4, 4, 4, 4, 1, 0, _lib.DSM_PREC_F16, null, null, null) == -1   # fp16 modes need both maxima
    assert hip_lib.dsm_conv2d_wgrad(one, one, one, null, 1, 32, 32, 8, 8, 8, 8, 1, 1, 0, 0, null, null, null) == -1
    assert hip_lib.dsm_conv2d_wgrad(one, one, one, one, 1, 32, 32, 8, 8, 4, 4, 2, 2, 0, 0, null, null, null) == -2      # stride 2 + dilation 2
    assert hip_lib.dsm_conv2d_wgrad(one, one, one, one, 1, 24, 32, 8, 8, 8, 8, 1, 1, 0, 0, null, null, null) == -2      # Cx % 32
    assert hip_lib.dsm_conv2d_wgrad(ctypes.c_void_p(20), one, one, one, 1, 32, 32, 8, 8, 8, 8, 1, 1, 0, 0, null, null, null) == -4
    # ABI v6: precision / x_amax of the split kernels, dsm_absmax
    a.x, a.precision = 16, 7
    assert hip_lib.dsm_conv3d_fwd(ctypes.byref(a), null) == -1            # unknown precision
    a.precision = _lib.DSM_PREC_F16X2
    assert hip_lib.dsm_conv3d_fwd(ctypes.byref(a), null) == -1            # a split kernel without x_amax
    buf = ctypes.create_string_buffer(96)
    a.x_amax = 16
    assert hip_lib.dsm_conv3d_plan(ctypes.byref(a), buf, 96) == 0 and b"f16x2" in buf.value
    a.precision = _lib.DSM_PREC_F16
    assert hip_lib.dsm_conv3d_plan(ctypes.byref(a), buf, 96) == 0 and b"_f16_" in buf.value
    a.precision = _lib.DSM_PREC_F32
    assert hip_lib.dsm_conv3d_plan(ctypes.byref(a), buf, 96) == 0 and b"bf16x3" in buf.value
    assert hip_lib.dsm_absmax(null, 4, one, null) == -1
    assert hip_lib.dsm_absmax(ctypes.c_void_p(20), 4, one, null) == -4


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: the product path fails loudly off the GPU."""
    from dsmnet_amd import costvolume as cv
    a, b = torch.zeros(1, 4, 2, 8), torch.zeros(1, 4, 2, 8)
    for fn in (lambda: cv.corr1d(a, b, 3), lambda: cv.concat_volume(a, b, 3, True),
               lambda: cv.soft_argmin(torch.zeros(1, 4, 2, 8))):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            fn()


def test_graft_entry_build_agrees_with_the_header(hip_lib):
    """__graft_entry__.build() asserts the ABI version: keep it in step with the header."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "dsmnet_hip.h")).read()
    version = int(re.search(r"#define DSM_ABI_VERSION (\d+)", header).group(1))
    entry = open(os.path.join(root, "__graft_entry__.py")).read()
    assert ("dsm_abi_version() == %d" % version) in entry
    assert hip_lib.dsm_abi_version() == version
