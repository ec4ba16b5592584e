"""ctypes binding of libdsmnet_hip.so (the C ABI of include/dsmnet_hip.h).

This is the whole reference-side binding: the reference is Python, so a
maintainer who wants the MI355X path adds exactly this stub.
There is no CPU fallback: if the library is missing the import of any op raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSM_LIB_PATH: another build of the same ABI (same-box A/B runs of kernel variants: scripts/ab_builds.sh)
LIB_PATH = os.environ.get("DSM_LIB_PATH") or os.path.join(_HERE, "csrc", "libdsmnet_hip.so")

c_int, c_void_p, c_size_t = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t

DSM_OK = 0
DSM_F32 = 0
DSM_NCDHW, DSM_NDHWC = 0, 1
DSM_CONV_FP32_MFMA, DSM_CONV_COUT1_CHUNKED, DSM_CONV_TM_SHIFT, DSM_CONV_BLOCKS_SHIFT = 1, 2, 4, 16
DSM_CONV_NO_NSPLIT = 4
DSM_CONV_NO_ONCE = 8
DSM_PREC_F32, DSM_PREC_F16, DSM_PREC_F16X2 = 0, 1, 2


class Bn3dArgs(ctypes.Structure):
    """struct dsm_bn3d_args (include/dsmnet_hip.h)."""
    _fields_ = [("y", c_void_p), ("residual", c_void_p), ("out", c_void_p), ("gamma", c_void_p),
                ("beta", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p),
                ("affine", c_void_p), ("workspace", c_void_p), ("gout", c_void_p), ("dy", c_void_p),
                ("dresidual", c_void_p), ("B", c_int), ("C", c_int),
                ("Dy", c_int), ("Hy", c_int), ("Wy", c_int), ("Dr", c_int), ("Hr", c_int), ("Wr", c_int),
                ("relu", c_int), ("momentum", ctypes.c_float), ("eps", ctypes.c_float),
                ("out_amax", c_void_p), ("dy_amax", c_void_p), ("dres_amax", c_void_p)]


class Conv3dArgs(ctypes.Structure):
    """struct dsm_conv3d_args (include/dsmnet_hip.h)."""
    _fields_ = [("x", c_void_p), ("w_packed", c_void_p), ("scale", c_void_p),
                ("shift", c_void_p), ("residual", c_void_p), ("y", c_void_p),
                ("B", c_int), ("Cin", c_int), ("Cout", c_int),
                ("Di", c_int), ("Hi", c_int), ("Wi", c_int),
                ("Do", c_int), ("Ho", c_int), ("Wo", c_int),
                ("Dr", c_int), ("Hr", c_int), ("Wr", c_int),
                ("stride", c_int), ("transposed", c_int), ("relu", c_int),
                ("kd", c_int), ("k", c_int), ("dil", c_int),
                ("flags", c_int), ("precision", c_int), ("x_amax", c_void_p), ("y_amax", c_void_p),
                ("vol_virtual", c_int), ("vol_mask_left", c_int)]


class BasicBlock2dArgs(ctypes.Structure):
    """struct dsm_basicblock2d_args (include/dsmnet_hip.h)."""
    _fields_ = [("x", c_void_p), ("y", c_void_p), ("w1_packed", c_void_p), ("w2_packed", c_void_p),
                ("scale1", c_void_p), ("shift1", c_void_p), ("scale2", c_void_p), ("shift2", c_void_p),
                ("x_amax", c_void_p), ("y_amax", c_void_p),
                ("B", c_int), ("H", c_int), ("W", c_int), ("C", c_int), ("precision", c_int), ("relu", c_int), ("no_skip", c_int)]


# name -> (restype, argtypes); must list every symbol declared in dsmnet_hip.h
SIGNATURES = {
    "dsm_abi_version": (c_int, []),
    "dsm_strerror": (ctypes.c_char_p, [c_int]),
    "dsm_corr1d_fwd": (c_int, [c_void_p] * 4 + [c_int] * 8 + [c_void_p]),
    "dsm_corr1d_bwd": (c_int, [c_void_p] * 6 + [c_int] * 8 + [c_void_p]),
    "dsm_concat_volume_fwd": (c_int, [c_void_p] * 3 + [c_int] * 8 + [c_void_p]),
    "dsm_concat_volume_bwd": (c_int, [c_void_p] * 3 + [c_int] * 8 + [c_void_p]),
    "dsm_soft_argmin_fwd": (c_int, [c_void_p] * 3 + [c_int] * 10 + [c_void_p]),
    "dsm_soft_argmin_bwd": (c_int, [c_void_p] * 5 + [c_int] * 10 + [c_void_p]),
    "dsm_conv3d_packed_weight_bytes": (c_size_t, [c_int] * 3),
    "dsm_conv3d_pack_weights": (c_int, [c_void_p] * 2 + [c_int] * 3 + [c_void_p]),
    "dsm_conv_pack_weights": (c_int, [c_void_p] * 2 + [c_int] * 5 + [c_void_p]),
    "dsm_absmax": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "dsm_conv3d_fwd": (c_int, [ctypes.POINTER(Conv3dArgs), c_void_p]),
    "dsm_basicblock2d_fwd": (c_int, [ctypes.POINTER(BasicBlock2dArgs), c_void_p]),
    "dsm_conv3d_plan": (c_int, [ctypes.POINTER(Conv3dArgs), ctypes.c_char_p, c_int]),
    "dsm_conv3d_wgrad": (c_int, [c_void_p] * 4 + [c_int] * 12 + [c_void_p] * 3),
    "dsm_conv2d_wgrad": (c_int, [c_void_p] * 4 + [c_int] * 11 + [c_void_p] * 3),
    "dsm_conv3d_cout1_bwd": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "dsm_deconv3d_cout1_bwd": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    "dsm_bn3d_train_fwd": (c_int, [ctypes.POINTER(Bn3dArgs), c_void_p]),
    "dsm_bn3d_train_bwd": (c_int, [ctypes.POINTER(Bn3dArgs), c_void_p]),
    "dsm_decoder_cat": (c_int, [c_void_p] * 5 + [c_int] * 11 + [c_void_p]),
    "dsm_stage_images_nhwc16": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p]),
    "dsm_volume_relayout": (c_int, [c_void_p] * 2 + [c_int] * 6 + [c_void_p]),
    "dsm_conv_packed_weight_bytes": (ctypes.c_size_t, [c_int] * 4),
    "dsm_spp_branch_floats": (ctypes.c_size_t, [c_int] * 3),
    "dsm_spp_pool8": (c_int, [c_void_p] * 2 + [c_int] * 3 + [c_void_p]),
    "dsm_spp_branches": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_void_p]),
    "dsm_spp_concat": (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_void_p] * 2),
    "dsm_warp_abs_error": (c_int, [c_void_p] * 4 + [c_int] * 6 + [ctypes.c_float, c_void_p]),
}

_lib = None


class DsmnetHipError(RuntimeError):
    pass


def _bind_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch ships its own libamdhip64.so.7 and this
    library is linked against the same soname, so the dynamic loader gives both whichever
    copy was loaded FIRST.  Loading ours before torch would bring in /opt/rocm's runtime,
    torch would then run on a runtime it was not built with, and launches on torch's
    streams fail (observed: hipErrorLaunchFailure on the first kernel).  Import torch and
    map its copy globally before ours."""
    import torch
    hip = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip):
        ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)


def load():
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DsmnetHipError(
            "libdsmnet_hip.so not found at %s -- build it with "
            "`python -m dsmnet_amd.csrc.build` (there is no CPU fallback)" % LIB_PATH)
    _bind_torch_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.dsm_abi_version() != 7:
        raise DsmnetHipError("libdsmnet_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(code, what):
    if code != DSM_OK:
        msg = load().dsm_strerror(code).decode()
        raise DsmnetHipError("%s failed: %s (code %d)" % (what, msg, code))
