"""Stereo cost-volume ops of the DSMnet path as ``torch.autograd.Function``s.

Every op is a thin host wrapper over the C ABI of ``include/dsmnet_hip.h``
(``libdsmnet_hip.so``, hand-written HIP for gfx950), launched on PyTorch's
current HIP stream.  PyTorch is plumbing here: device memory, streams, autograd
book-keeping.  There is no CPU path -- CPU tensors raise.

Reference code each op replaces (sunshinnnn/DSMnet):
  corr1d          models/util_conv.py:56-86           (Corr1d)
  concat_volume   models/gcnet.py:130-135, models/psmnet/stackhourglass.py:124-133
  soft_argmin     models/psmnet/stackhourglass.py:152-166 + submodule.py:56-63,
                  models/gcnet.py:104-111
  conv3d_block    models/psmnet/submodule.py:16-19, stackhourglass.py:22-62,
                  models/util_conv.py:150-179, models/util_fun.py:41-50
"""
import ctypes

import torch

from . import _lib

_CL3D = torch.channels_last_3d


class LaunchTimer(object):
    """Per-launch timing with HIP events recorded on the launch stream (the stream the
    kernels are enqueued on -- PyTorch's current stream).  Install with ``set_timer``;
    bench.py reads ``summary()`` after a synchronise.  Costs two event records per launch."""

    def __init__(self):
        self.records = []

    def start(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def stop(self, name, start, work):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.records.append((name, start, ev, work))

    def summary(self):
        """name -> dict(launches, ms, work); ``work`` is the algorithmic bytes (HBM-bound
        kernels) or FLOPs (MFMA kernels) summed over the launches.  Call after a sync."""
        out = {}
        for name, a, b, work in self.records:
            e = out.setdefault(name, {"launches": 0, "ms": 0.0, "work": 0.0})
            e["launches"] += 1
            e["ms"] += a.elapsed_time(b)
            e["work"] += work
        return out


_timer = None


def set_timer(timer):
    global _timer
    _timer = timer


class _timed(object):
    """``work``: algorithmic bytes or FLOPs of the launch (SURVEY.md section 8d counting)."""

    def __init__(self, name, work=0.0):
        self.name, self.work = name, work

    def __enter__(self):
        self.t0 = _timer.start() if _timer is not None else None

    def __exit__(self, *exc):
        if self.t0 is not None:
            _timer.stop(self.name() if callable(self.name) else self.name, self.t0, self.work)
        return False


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _require_device(name, *tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "%s: dsmnet_amd ops run on the MI355X through libdsmnet_hip.so only; got a %s "
                "tensor (there is no CPU fallback)" % (name, t.device.type))
        if t.dtype != torch.float32:
            raise TypeError("%s: only float32 is implemented, got %s" % (name, t.dtype))


def _same_shape(name, a, b):
    if a.shape != b.shape:
        # the reference asserts this in its model forwards (gcnet.py:127, dispnetcorr.py:67)
        raise ValueError("%s: left/right feature shapes differ: %s vs %s"
                         % (name, tuple(a.shape), tuple(b.shape)))


# ----------------------------------------------------------------------------
# (a1) Corr1d
# ----------------------------------------------------------------------------
class Corr1dFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fL, fR, D, stride, kernel_size):
        _require_device("corr1d", fL, fR)
        if fL.dim() != 4:
            raise ValueError("corr1d: expected (B,C,H,W) features, got %d-D" % fL.dim())
        _same_shape("corr1d", fL, fR)
        if kernel_size % 2 != 1:
            raise AssertionError("kernel_size must be odd")      # util_conv.py:83
        fL, fR = fL.contiguous(), fR.contiguous()
        B, C, H, W = fL.shape
        out = torch.empty((B, D, H, W), device=fL.device, dtype=fL.dtype)
        tmp = torch.empty_like(out) if kernel_size > 1 else None
        lib = _lib.load()
        with torch.cuda.device(fL.device), _timed("corr1d_fwd_kernel", 4.0 * (2 * B * C * H * W + B * D * H * W)):
            rc = lib.dsm_corr1d_fwd(_p(fL), _p(fR), _p(out), _p(tmp), B, C, H, W, D, stride,
                                    kernel_size, _lib.DSM_F32, _stream())
        _lib.check(rc, "dsm_corr1d_fwd")
        ctx.save_for_backward(fL, fR)
        ctx.cfg = (D, stride, kernel_size)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        fL, fR = ctx.saved_tensors
        D, stride, kernel_size = ctx.cfg
        g = grad_out.contiguous()
        B, C, H, W = fL.shape
        dfL, dfR = torch.empty_like(fL), torch.empty_like(fR)
        tmp = torch.empty_like(g) if kernel_size > 1 else None
        lib = _lib.load()
        with torch.cuda.device(fL.device):
            rc = lib.dsm_corr1d_bwd(_p(g), _p(fL), _p(fR), _p(dfL), _p(dfR), _p(tmp), B, C, H, W,
                                    D, stride, kernel_size, _lib.DSM_F32, _stream())
        _lib.check(rc, "dsm_corr1d_bwd")
        return dfL, dfR, None, None, None


def corr1d(fL, fR, D, stride=1, kernel_size=1):
    """``Corr1d(kernel_size, stride, D).forward(fL, fR)`` -> (B, D, H, W)."""
    return Corr1dFunction.apply(fL, fR, int(D), int(stride), int(kernel_size))


# ----------------------------------------------------------------------------
# (a2, a3) concatenation cost volume
# ----------------------------------------------------------------------------
def concat_volume_right(fL, fR, D):
    """The right-referenced volume of ``gcnet_LR`` (models/gcnet.py:155-164), (B, 2C, D, H, W)
    channels_last_3d: right features at every x, ``vol[:, C:, d, y, x] = fL[y, x + d]`` for
    ``x + d < W``.  Same kernel as ``concat_volume`` walking the other way.  Inference only."""
    _require_device("concat_volume_right", fL, fR)
    _same_shape("concat_volume_right", fL, fR)
    if fL.requires_grad or fR.requires_grad:
        if torch.is_grad_enabled():
            raise NotImplementedError("concat_volume_right has no backward (gcnet_LR is not reachable "
                                      "from the reference's model factory)")
    fL, fR = fL.contiguous(), fR.contiguous()
    B, C, H, W = fL.shape
    vol = torch.empty((B, 2 * C, D, H, W), device=fL.device, dtype=fL.dtype, memory_format=_CL3D)
    with torch.cuda.device(fL.device), _timed("volume_ndhwc_fwd_kernel", 4.0 * (2 * B * C * H * W + 2 * B * C * D * H * W)):
        rc = _lib.load().dsm_concat_volume_fwd(_p(fR), _p(fL), _p(vol), B, C, H, W, int(D), 2,
                                               _lib.DSM_NDHWC, _lib.DSM_F32, _stream())
    _lib.check(rc, "dsm_concat_volume_fwd")
    return vol


class ConcatVolumeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fL, fR, D, mask_left, channels_last):
        _require_device("concat_volume", fL, fR)
        if fL.dim() != 4:
            raise ValueError("concat_volume: expected (B,C,H,W) features, got %d-D" % fL.dim())
        _same_shape("concat_volume", fL, fR)
        fL, fR = fL.contiguous(), fR.contiguous()
        B, C, H, W = fL.shape
        fmt = _CL3D if channels_last else torch.contiguous_format
        vol = torch.empty((B, 2 * C, D, H, W), device=fL.device, dtype=fL.dtype,
                          memory_format=fmt)
        layout = _lib.DSM_NDHWC if channels_last else _lib.DSM_NCDHW
        lib = _lib.load()
        with torch.cuda.device(fL.device), \
                _timed("volume_ndhwc_fwd_kernel" if channels_last else "volume_ncdhw_fwd_kernel",
                       4.0 * (2 * B * C * H * W + 2 * B * C * D * H * W)):
            rc = lib.dsm_concat_volume_fwd(_p(fL), _p(fR), _p(vol), B, C, H, W, D,
                                           int(mask_left), layout, _lib.DSM_F32, _stream())
        _lib.check(rc, "dsm_concat_volume_fwd")
        ctx.cfg = (B, C, H, W, D, int(mask_left), layout, fmt)
        return vol

    @staticmethod
    def backward(ctx, gvol):
        B, C, H, W, D, mask_left, layout, fmt = ctx.cfg
        g = gvol.contiguous(memory_format=fmt)
        dfL = torch.empty((B, C, H, W), device=g.device, dtype=g.dtype)
        dfR = torch.empty_like(dfL)
        lib = _lib.load()
        with torch.cuda.device(g.device):
            rc = lib.dsm_concat_volume_bwd(_p(g), _p(dfL), _p(dfR), B, C, H, W, D, mask_left,
                                           layout, _lib.DSM_F32, _stream())
        _lib.check(rc, "dsm_concat_volume_bwd")
        return dfL, dfR, None, None, None


def concat_volume(fL, fR, D, mask_left, channels_last=True):
    """Concatenation cost volume (B, 2C, D, H, W).

    ``mask_left=False``: GCNet (gcnet.py:130-135); ``True``: PSMNet
    (stackhourglass.py:124-133).  ``channels_last`` selects the memory format of the
    result: ``torch.channels_last_3d`` (what ``conv3d_block`` consumes) or contiguous."""
    vol = ConcatVolumeFunction.apply(fL, fR, int(D), bool(mask_left), bool(channels_last))
    sl = getattr(fL, "_dsm_amax", None)
    if sl is not None and sl is getattr(fR, "_dsm_amax", None):
        vol._dsm_amax = sl                  # copies of the features: the same bound holds
    return vol


# ----------------------------------------------------------------------------
# (a6, a7) soft-argmin
# ----------------------------------------------------------------------------
class SoftArgminFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cost, out_size, negate, align_corners):
        _require_device("soft_argmin", cost)
        if cost.dim() == 5:
            if cost.shape[1] != 1:
                raise ValueError("soft_argmin: 5-D cost must have one channel")
            c4 = cost.reshape(cost.shape[0], *cost.shape[2:])
        elif cost.dim() == 4:
            c4 = cost
        else:
            raise ValueError("soft_argmin: expected (B,1,D,H,W) or (B,D,H,W)")
        c4 = c4.contiguous()
        B, Dc, Hc, Wc = c4.shape
        D, H, W = (Dc, Hc, Wc) if out_size is None else tuple(int(v) for v in out_size)
        disp = torch.empty((B, H, W), device=cost.device, dtype=cost.dtype)
        need_grad = ctx.needs_input_grad[0]
        stats = torch.empty((B, 2, H, W), device=cost.device, dtype=cost.dtype) if need_grad else None
        lib = _lib.load()
        # the x4 fast path (soft_argmin.hip: dsm_soft_argmin_fwd) has its own kernel
        kname = ("soft_argmin_up4_kernel" if (D == 4 * Dc and (Hc, Wc) != (H, W) and not align_corners)
                 else "soft_argmin_fwd_kernel")
        with torch.cuda.device(cost.device), _timed(kname, 4.0 * B * (Dc * Hc * Wc + H * W)):
            rc = lib.dsm_soft_argmin_fwd(_p(c4), _p(disp), _p(stats), B, Dc, Hc, Wc, D, H, W,
                                         int(negate), int(align_corners), _lib.DSM_F32, _stream())
        _lib.check(rc, "dsm_soft_argmin_fwd")
        if need_grad:
            ctx.save_for_backward(c4, disp, stats)
        ctx.cfg = (tuple(cost.shape), B, Dc, Hc, Wc, D, H, W, int(negate), int(align_corners))
        return disp

    @staticmethod
    def backward(ctx, gdisp):
        c4, disp, stats = ctx.saved_tensors
        shape, B, Dc, Hc, Wc, D, H, W, negate, align = ctx.cfg
        g = gdisp.contiguous()
        dcost = torch.empty_like(c4)
        lib = _lib.load()
        with torch.cuda.device(c4.device):
            rc = lib.dsm_soft_argmin_bwd(_p(c4), _p(disp), _p(stats), _p(g), _p(dcost), B, Dc, Hc,
                                         Wc, D, H, W, negate, align, _lib.DSM_F32, _stream())
        _lib.check(rc, "dsm_soft_argmin_bwd")
        return dcost.reshape(shape), None, None, None


def soft_argmin(cost, out_size=None, negate=False, align_corners=False):
    """Fused (trilinear upsample ->) softmax over disparity -> expectation.  -> (B, H, W)."""
    return SoftArgminFunction.apply(cost, out_size, bool(negate), bool(align_corners))


# ----------------------------------------------------------------------------
# (a4, a5) 3-D convolution block
# ----------------------------------------------------------------------------
def pack_conv3d_weight(weight, transposed):
    """torch Conv3d / ConvTranspose3d weight (k=3) -> MFMA fragment order (device)."""
    _require_device("pack_conv3d_weight", weight)
    if weight.dim() != 5 or tuple(weight.shape[2:]) != (3, 3, 3):
        raise ValueError("conv3d_block supports kernel_size=3 only, got %s" % (tuple(weight.shape),))
    cin, cout = (weight.shape[0], weight.shape[1]) if transposed else (weight.shape[1], weight.shape[0])
    w = weight.detach().contiguous()
    lib = _lib.load()
    nbytes = lib.dsm_conv3d_packed_weight_bytes(cin, cout, int(transposed))
    packed = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    with torch.cuda.device(w.device):
        rc = lib.dsm_conv3d_pack_weights(_p(w), _p(packed), cin, cout, int(transposed), _stream())
    _lib.check(rc, "dsm_conv3d_pack_weights")
    return packed


def to_channels_last_3d(x):
    """(B,C,D,H,W) in any strides -> NDHWC memory (no copy when it already is)."""
    if x.is_contiguous(memory_format=_CL3D):
        return x
    _require_device("to_channels_last_3d", x)
    xc = x.contiguous()
    B, C, D, H, W = xc.shape
    out = torch.empty((B, C, D, H, W), device=x.device, dtype=x.dtype, memory_format=_CL3D)
    with torch.cuda.device(x.device):
        rc = _lib.load().dsm_volume_relayout(_p(xc), _p(out), B, C, D, H, W, 1, _stream())
    _lib.check(rc, "dsm_volume_relayout")
    return out


def to_contiguous_3d(x):
    """NDHWC memory -> torch contiguous (B,C,D,H,W)."""
    if x.is_contiguous():
        return x
    if not x.is_contiguous(memory_format=_CL3D):
        return x.contiguous()
    B, C, D, H, W = x.shape
    out = torch.empty((B, C, D, H, W), device=x.device, dtype=x.dtype)
    with torch.cuda.device(x.device):
        rc = _lib.load().dsm_volume_relayout(_p(x), _p(out), B, C, D, H, W, 0, _stream())
    _lib.check(rc, "dsm_volume_relayout")
    return out


def conv3d_out_size(in_size, stride, transposed):
    if transposed:
        return tuple(2 * v for v in in_size)          # k3, s2, p1, op1
    return tuple((v - 1) // stride + 1 for v in in_size)


def conv3d_block(x, packed_weight, cout, scale=None, shift=None, residual=None, stride=1,
                 transposed=False, relu=False, out_size=None):
    """y = relu?(conv(x) * scale + shift (+ residual, cropped to the common size)).

    ``x`` is (B,Cin,D,H,W); it is consumed in NDHWC memory (converted if needed) and
    the result is returned as a channels_last_3d tensor.  With ``residual`` the output
    takes the element-wise minimum of the two spatial sizes -- ``myadd_3d`` semantics
    (stackhourglass.py:10-20).  ``x`` may be a ``VirtualVolume`` (Conv3d k3 s1 to 32 channels: the
    z-sliding kernel stages the never-materialised cost volume from the towers' output).
    Inference only.  What the MFMA kernels multiply in follows the ``conv_precision`` option."""
    virtual = x if isinstance(x, VirtualVolume) else None
    if virtual is not None:
        x = virtual.features                     # NHWC (2B, C, H, W): staged as the volume's planes
        if stride != 1 or transposed or cout != 32:
            raise ValueError("a virtual cost volume feeds a Conv3d(k3, s1) to 32 channels only")
        _require_device("conv3d_block", x, scale, shift, residual)
        B, cin, Di, Hi, Wi = virtual.shape
    else:
        _require_device("conv3d_block", x, scale, shift, residual)
        x = carry_amax(to_channels_last_3d(x), x)
        B, cin, Di, Hi, Wi = x.shape
    Do, Ho, Wo = conv3d_out_size((Di, Hi, Wi), stride, transposed)
    a = _lib.Conv3dArgs()
    if residual is not None:
        if residual.shape[0] != B or residual.shape[1] != cout:
            raise ValueError("conv3d_block: residual has shape %s, expected (%d,%d,...)"
                             % (tuple(residual.shape), B, cout))
        residual = to_channels_last_3d(residual)
        a.Dr, a.Hr, a.Wr = residual.shape[2:]
        Do, Ho, Wo = min(Do, a.Dr), min(Ho, a.Hr), min(Wo, a.Wr)
    if out_size is not None:                     # a corner of the natural output (bwd-data crops)
        Do, Ho, Wo = (min(n, int(o)) for n, o in zip((Do, Ho, Wo), out_size))
    y = torch.empty((B, cout, Do, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=_CL3D)
    a.x = x.data_ptr()
    a.w_packed = packed_weight.data_ptr()
    a.y = y.data_ptr()
    a.scale = None if scale is None else scale.data_ptr()
    a.shift = None if shift is None else shift.data_ptr()
    a.residual = None if residual is None else residual.data_ptr()
    a.B, a.Cin, a.Cout = B, cin, cout
    a.Di, a.Hi, a.Wi = Di, Hi, Wi
    a.Do, a.Ho, a.Wo = Do, Ho, Wo
    a.stride, a.transposed, a.relu = int(stride), int(transposed), int(relu)
    a.flags = _conv_flags()
    if virtual is not None:
        a.vol_virtual, a.vol_mask_left = 1, int(virtual.mask_left)
    keep = _set_precision(a, x, y if cout > 1 else None)
    # FLOPs as SURVEY.md section 8d counts them: 2*27*Cin*Cout per output voxel (conv) or
    # per input voxel (transposed conv)
    # (Cout = 1 runs on the VALU and is HBM-bound: input read once + output written)
    vox = B * (Di * Hi * Wi if transposed else Do * Ho * Wo)
    work = 54.0 * cin * cout * vox if cout > 1 else 4.0 * B * (cin * Di * Hi * Wi + Do * Ho * Wo)
    with torch.cuda.device(x.device), _timed(lambda: conv3d_plan_name(a), work):
        rc = _lib.load().dsm_conv3d_fwd(ctypes.byref(a), _stream())
    _lib.check(rc, "dsm_conv3d_fwd")
    del keep
    return y


def conv3d_plan_name(args):
    """Kernel variant ``dsm_conv3d_fwd`` picks for ``args`` (``dsm_conv3d_plan``)."""
    buf = ctypes.create_string_buffer(96)
    _lib.check(_lib.load().dsm_conv3d_plan(ctypes.byref(args), buf, 96), "dsm_conv3d_plan")
    return buf.value.decode()


# ----------------------------------------------------------------------------
# host-side options, absolute maxima of the fp16 precisions, the virtual cost volume
# ----------------------------------------------------------------------------
import os as _os

# host-side options (the library itself reads no environment variable): "conv_precision" starts from
# DSM_CONV_PRECISION=bf16x3|fp32|f16x2|f16 so that scripts and the parity tests can switch whole runs
_PRECISIONS = ("bf16x3", "fp32", "f16x2", "f16")


def _env_precision():
    v = _os.environ.get("DSM_CONV_PRECISION", "")
    if v and v not in _PRECISIONS:
        raise ValueError("DSM_CONV_PRECISION must be one of %s, got %r" % (_PRECISIONS, v))
    return v or "f16x2"


_OPTIONS = {"fuse_volume": True, "fuse_blocks": True, "conv_precision": _env_precision(), "conv_flags": 0}


def set_option(name, value):
    """Host-side switches for A/B runs and tests (no environment variable is read by the library):
    ``conv_precision`` -- what the 3x3(x3) MFMA convolutions multiply in:
        "bf16x3"  fp32 accuracy, three-term bf16 split, six MFMAs per product;
        "f16x2"   fp32 accuracy, two-term fp16 split of power-of-two-scaled operands, three MFMAs;
        "f16"     operands rounded to fp16, one MFMA, fp32 accumulate -- the reduced-precision mode of
                  BASELINE config #5 (forward, backward-data and weight gradients);
        "fp32"    the exact fp32-input MFMA (v_mfma_f32_32x32x2_f32);
    ``conv_fp32`` -- older spelling: True = "fp32", False = "bf16x3";
    ``fuse_volume`` -- PSMNet's / GCNet's eval forward never materialises the cost volume: the first
    3-D convolution stages it from the feature maps;
    ``fuse_blocks`` -- the towers' stride-1 64-channel BasicBlocks run as one launch each (fp16 modes);
    ``conv_flags`` -- raw dsm_conv3d_args.flags bits (tile height, grid size)."""
    if name == "conv_fp32":
        old = _OPTIONS["conv_precision"] == "fp32"
        _OPTIONS["conv_precision"] = "fp32" if value else "bf16x3"
        return old
    if name not in _OPTIONS:
        raise KeyError(name)
    old = _OPTIONS[name]
    if name == "conv_precision":
        if value not in _PRECISIONS:
            raise ValueError("conv_precision must be one of %s" % (_PRECISIONS,))
        _OPTIONS[name] = value
    else:
        _OPTIONS[name] = int(value) if name == "conv_flags" else bool(value)
    return old


def _conv_flags():
    """dsm_conv3d_args.flags of every convolution launch: precision "fp32" keeps the exact fp32-input
    MFMA kernels; ``conv_flags`` carries raw A/B bits (tile height, grid size: include/dsmnet_hip.h)."""
    return (_lib.DSM_CONV_FP32_MFMA if _OPTIONS["conv_precision"] == "fp32" else 0) | _OPTIONS["conv_flags"]


def get_option(name):
    if name == "conv_fp32":
        return _OPTIONS["conv_precision"] == "fp32"
    return _OPTIONS[name]


# ----------------------------------------------------------------------------
# absolute maxima for the fp16 precisions (include/dsmnet_hip.h: x_amax / y_amax)
# ----------------------------------------------------------------------------
# The fp16 split kernels scale every tensor by a power of two taken from its absolute maximum -- a
# device float that the PRODUCING launch writes from its epilogue (atomic max) and the consumer
# reads at kernel start: no host round trip, capturable in a hipGraph.  On the host the scalar rides
# on the tensor object as ``_dsm_amax``; a tensor without one (an input of the model, the result of
# a stock torch op) gets it from one ``dsm_absmax`` pass.  Slots come from a per-device arena that a
# model forward zeroes once (``amax_scope``), so that a forward costs one fill, not one per layer.
class _AmaxArena(object):
    SLOTS = 2048

    def __init__(self):
        self.buf, self.used, self.depth = {}, {}, 0

    def begin(self, device):
        key = (device.type, device.index)
        if key not in self.buf:
            self.buf[key] = torch.zeros(self.SLOTS, device=device, dtype=torch.float32)
        else:
            self.buf[key].zero_()
        self.used[key] = 0

    def slot(self, device):
        key = (device.type, device.index)
        if self.depth > 0 and key in self.buf and self.used[key] < self.SLOTS:
            i = self.used[key]
            self.used[key] = i + 1
            return self.buf[key][i:i + 1]
        return torch.zeros(1, device=device, dtype=torch.float32)


_ARENA = _AmaxArena()


class amax_scope(object):
    """``with amax_scope(device):`` around a model forward: the absolute-maximum slots of every
    launch inside come from one arena, zeroed once on entry (nested scopes share the outer one).
    A no-op unless an fp16 precision is selected."""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        if needs_amax():
            if _ARENA.depth == 0:
                _ARENA.begin(self.device)
            _ARENA.depth += 1
            self.entered = True
        else:
            self.entered = False
        return self

    def __exit__(self, *exc):
        if self.entered:
            _ARENA.depth -= 1
        return False


def needs_amax():
    return _OPTIONS["conv_precision"] in ("f16x2", "f16")


def absmax(x):
    """Device scalar holding max |x| (one pass, ``dsm_absmax``), also remembered on ``x``."""
    slot = _ARENA.slot(x.device)
    with torch.cuda.device(x.device), _timed("absmax_kernel", 4.0 * x.numel()):
        rc = _lib.load().dsm_absmax(_p(x), x.numel(), _p(slot), _stream())
    _lib.check(rc, "dsm_absmax")
    try:
        x._dsm_amax = slot
    except AttributeError:
        pass
    return slot


def amax_of(x):
    """The absolute-maximum scalar of ``x``: the one its producer attached, else a fresh pass."""
    slot = getattr(x, "_dsm_amax", None)
    return slot if slot is not None else absmax(x)


def carry_amax(dst, *srcs):
    """``dst`` holds values bounded by the maximum over ``srcs`` (a view, a slice, a concatenation):
    hand the bound on without a pass when a single source carries one."""
    slots = [getattr(t, "_dsm_amax", None) for t in srcs]
    if len(slots) == 1 and slots[0] is not None:
        dst._dsm_amax = slots[0]
    return dst


def _split_kernel_layer(a):
    """Does ``dsm_conv3d_fwd`` run this layer on a split-operand kernel (conv3d.hip make_plan kinds
    5 / 6)?  3x3(x3) taps, Cin % 16 == 0, Cout a multiple of 32 up to 64 (3-D; 128 on small volumes at
    stride 1) / 128 (2-D); stride 1, 3-D stride 2 to 64 channels, or transposed from Cin % 32 == 0."""
    kd, k = (a.kd or 3), (a.k or 3)
    if k != 3 or a.Cin % 16 or a.Cout % 32 or a.Cout > 128:
        return False
    if kd == 3 and a.Cout == 128:                    # four workgroup columns of 32: small volumes, stride 1 only
        tiles4 = a.B * a.Do * ((a.Ho + 3) // 4) * ((a.Wo + 31) // 32)
        return (not a.transposed) and a.stride == 1 and (a.dil or 1) == 1 and tiles4 <= 128
    if kd == 3 and a.Cout > 64:
        return False
    if a.transposed:
        return kd == 3 and a.Cin % 32 == 0
    return a.stride == 1 or (kd == 3 and a.Cout == 64)


def _set_precision(a, x, y):
    """precision / x_amax / y_amax of a ``dsm_conv3d_args``.  Returns the tensors the launch must
    keep alive.  ``x``, ``y``: the input and output tensors (NDHWC / NHWC memory); ``y`` None:
    the output needs no maximum (Cout = 1 heads)."""
    mode = _OPTIONS["conv_precision"]
    if mode not in ("f16x2", "f16"):
        return None
    a.precision = _lib.DSM_PREC_F16X2 if mode == "f16x2" else _lib.DSM_PREC_F16
    xa = None
    if _split_kernel_layer(a):                       # the others compute in fp32 and read no maximum
        xa = amax_of(x)
        a.x_amax = xa.data_ptr()
    ya = None
    if y is not None:
        ya = _ARENA.slot(y.device)
        a.y_amax = ya.data_ptr()
        y._dsm_amax = ya
    return xa, ya


class VirtualVolume(object):
    """A concatenation cost volume that is never written (models/psmnet/stackhourglass.py:124-133,
    models/gcnet.py:130-135): ``features`` is the towers' output for both views as ONE NHWC tensor
    (2B, C, H, W) [left maps, then right maps]; the first 3-D convolution (``conv3d_block``; the
    z-sliding kernel, csrc/conv_zs.hpp) stages plane d as [left | right shifted by d] with x < d
    zeroed.  ``shape``: the logical (B, 2C, D, H, W).  Inference only."""
    __slots__ = ("features", "shape", "mask_left")

    def __init__(self, features, D, mask_left):
        if features.dim() != 4 or features.shape[0] % 2:
            raise ValueError("VirtualVolume: features must be (2B, C, H, W), left maps then right maps")
        if not features.is_contiguous(memory_format=torch.channels_last):
            features = carry_amax(features.contiguous(memory_format=torch.channels_last), features)
        self.features = features
        B2, C, H, W = features.shape
        self.shape = (B2 // 2, 2 * C, int(D), H, W)
        self.mask_left = bool(mask_left)

    @property
    def device(self):
        return self.features.device


def virtual_volume_ok(C):
    """The z-sliding kernel can stage a virtual volume of 2C channels (C % 32 == 0) in every
    precision except "fp32" (which has no split kernels)."""
    return C % 32 == 0 and _OPTIONS["conv_precision"] != "fp32"


# ----------------------------------------------------------------------------
# 2-D convolution block (feature towers): same MFMA kernel, kd = 1
# ----------------------------------------------------------------------------
_CL2D = torch.channels_last


def pack_conv2d_weight(weight, cin_padded=None):
    """torch Conv2d weight (Cout, Cin, k, k), k in {1, 3} -> MFMA fragment order.  ``cin_padded``
    (a multiple of 16 >= Cin) zero-fills the extra input channels."""
    _require_device("pack_conv2d_weight", weight)
    cout, cin, kh, kw = weight.shape
    if kh != kw or kh not in (1, 3):
        raise ValueError("conv2d_block supports 1x1 and 3x3 kernels, got %dx%d" % (kh, kw))
    cin_p = cin if cin_padded is None else int(cin_padded)
    w = weight.detach().contiguous()
    lib = _lib.load()
    nbytes = lib.dsm_conv_packed_weight_bytes(cin_p, cout, 1, kh)
    if nbytes == 0:
        raise ValueError("pack_conv2d_weight: unsupported shape %s" % (tuple(weight.shape),))
    packed = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    with torch.cuda.device(w.device):
        rc = lib.dsm_conv_pack_weights(_p(w), _p(packed), cin, cin_p, cout, 1, kh, _stream())
    _lib.check(rc, "dsm_conv_pack_weights")
    return packed


def conv2d_block(x, packed_weight, cout, scale=None, shift=None, residual=None, stride=1,
                 relu=0, k=3, dilation=1):
    """y = relu?(conv2d(x) * scale + shift (+ residual)) on NHWC maps, "same" padding.
    ``x``: (B, Cin, H, W) in torch.channels_last memory, Cin a multiple of 16.  Inference only."""
    _require_device("conv2d_block", x, packed_weight, scale, shift, residual)
    if not x.is_contiguous(memory_format=_CL2D):
        x = carry_amax(x.contiguous(memory_format=_CL2D), x)
    B, cin, Hi, Wi = x.shape
    Ho, Wo = (Hi - 1) // stride + 1, (Wi - 1) // stride + 1
    a = _lib.Conv3dArgs()
    if residual is not None:
        if tuple(residual.shape) != (B, cout, Ho, Wo):
            raise ValueError("conv2d_block: residual shape %s != %s"
                             % (tuple(residual.shape), (B, cout, Ho, Wo)))
        if not residual.is_contiguous(memory_format=_CL2D):
            residual = residual.contiguous(memory_format=_CL2D)
        a.Dr, a.Hr, a.Wr = 1, Ho, Wo
    dev = packed_weight.device
    y = torch.empty((B, cout, Ho, Wo), device=dev, dtype=torch.float32, memory_format=_CL2D)
    a.x = x.data_ptr()
    a.w_packed = packed_weight.data_ptr()
    a.y = y.data_ptr()
    a.scale = None if scale is None else scale.data_ptr()
    a.shift = None if shift is None else shift.data_ptr()
    a.residual = None if residual is None else residual.data_ptr()
    a.B, a.Cin, a.Cout = B, cin, cout
    a.Di, a.Hi, a.Wi = 1, Hi, Wi
    a.Do, a.Ho, a.Wo = 1, Ho, Wo
    a.stride, a.transposed, a.relu = int(stride), 0, int(relu)
    a.kd, a.k, a.dil = 1, int(k), int(dilation)
    a.flags = _conv_flags()
    keep = _set_precision(a, x, y)
    work = 2.0 * k * k * cin * cout * B * Ho * Wo
    with torch.cuda.device(dev), _timed(lambda: conv3d_plan_name(a), work):
        rc = _lib.load().dsm_conv3d_fwd(ctypes.byref(a), _stream())
    _lib.check(rc, "dsm_conv3d_fwd")
    del keep
    return y


def basicblock2d_ok(x, cin, cout, stride, dilation):
    """Can a BasicBlock (two 3x3 convolutions + skip add) run as ONE launch (csrc/basicblock2d.hpp)?
    The towers' stride-1 64-channel blocks in the fp16 modes, eval."""
    return (_OPTIONS["fuse_blocks"] and _OPTIONS["conv_precision"] in ("f16x2", "f16") and cin in (32, 64) and
            cout == cin and stride == 1 and dilation == 1 and x.dim() == 4 and x.shape[1] == cin and
            4 * x.numel() < 2 ** 31)


def basicblock2d(x, packed1, scale1, shift1, packed2, scale2, shift2, relu=False, skip=True):
    """``conv2(relu(conv1(x) * scale1 + shift1)) * scale2 + shift2 + x`` (``relu``: ReLU after the add)
    on an NHWC map with 64 or 32 channels, both convolutions 3x3 / stride 1 / pad 1
    (models/psmnet/submodule.py:24-46, models/util_conv.py:181-210 with the BatchNorms folded;
    ``skip=False``: no ``+ x`` -- two convbn + ReLU layers in a row): one
    launch, the intermediate map stays in LDS.  ``packed*``: the layers' ``pack_conv2d_weight``
    buffers.  Inference only; fp16 modes only."""
    _require_device("basicblock2d", x, packed1, packed2, scale1, shift1, scale2, shift2)
    if not x.is_contiguous(memory_format=_CL2D):
        x = carry_amax(x.contiguous(memory_format=_CL2D), x)
    B, C, H, W = x.shape
    mode = _OPTIONS["conv_precision"]
    if C not in (32, 64) or mode not in ("f16x2", "f16"):
        raise ValueError("basicblock2d: 32 or 64 channels and an fp16 precision mode, got C=%d, %s" % (C, mode))
    y = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32, memory_format=_CL2D)
    a = _lib.BasicBlock2dArgs()
    a.x, a.y = x.data_ptr(), y.data_ptr()
    a.w1_packed, a.w2_packed = packed1.data_ptr(), packed2.data_ptr()
    a.scale1 = None if scale1 is None else scale1.data_ptr()
    a.shift1 = None if shift1 is None else shift1.data_ptr()
    a.scale2 = None if scale2 is None else scale2.data_ptr()
    a.shift2 = None if shift2 is None else shift2.data_ptr()
    a.B, a.H, a.W, a.C = B, H, W, C
    a.relu = 1 if relu else 0
    a.no_skip = 0 if skip else 1
    a.precision = _lib.DSM_PREC_F16X2 if mode == "f16x2" else _lib.DSM_PREC_F16
    xa = amax_of(x)
    ya = _ARENA.slot(y.device)
    a.x_amax, a.y_amax = xa.data_ptr(), ya.data_ptr()
    y._dsm_amax = ya
    work = 2.0 * 2 * 9 * C * C * B * H * W
    with torch.cuda.device(x.device), _timed("basicblock2d_%s_mfma_kernel<C=%d>" % (mode, C), work):
        rc = _lib.load().dsm_basicblock2d_fwd(ctypes.byref(a), _stream())
    _lib.check(rc, "dsm_basicblock2d_fwd")
    del xa
    return y


def warp_abs_error(left, right, disp, delt):
    """``|left - imwrap_BCHW(right, disp)|`` in one pass (csrc/warp.hip; utils/imwrap.py:37-72,
    models/iresnet.py:169-170).  ``left=None``: the warped map itself.  ``delt`` is the
    reference's random epsilon (a Python float drawn by the caller).  NCHW fp32; inference only."""
    _require_device("warp_abs_error", right, disp, left)
    B, C, H0, W0 = right.shape
    if disp.dim() != 4 or disp.shape[0] != B or disp.shape[1] != 1:
        raise ValueError("warp_abs_error: disp must be (B,1,H,W), got %s" % (tuple(disp.shape),))
    H, W = disp.shape[2:]
    if min(H, W, H0, W0) <= 1:
        raise ValueError("warp_abs_error: maps must be larger than 1x1")        # imwrap.py:48
    if left is not None and tuple(left.shape) != (B, C, H, W):
        raise ValueError("warp_abs_error: left %s does not match (B,C)=%s and disp %s"
                         % (tuple(left.shape), (B, C), (H, W)))
    right, disp = right.contiguous(), disp.contiguous()
    left = None if left is None else left.contiguous()
    out = torch.empty((B, C, H, W), device=right.device, dtype=torch.float32)
    n = out.numel()
    with torch.cuda.device(right.device), _timed("warp_abs_error_kernel",
                                                 4.0 * (right.numel() + disp.numel() + n * (2 if left is not None else 1))):
        rc = _lib.load().dsm_warp_abs_error(None if left is None else _p(left), _p(right), _p(disp),
                                            _p(out), B, C, H, W, H0, W0, float(delt), _stream())
    _lib.check(rc, "dsm_warp_abs_error")
    return out


def spp_head(raw, skip, w_t, scale, shift):
    """PSMNet's SPP head in three launches (csrc/spp.hip; models/psmnet/submodule.py:81-99,
    126-137): ``raw`` (B,64,H,W) and ``skip`` (B,128,H,W) channels_last -> the 320-channel
    concat [raw | skip | branch4 | branch3 | branch2 | branch1] (B,320,H,W) channels_last.
    ``w_t`` (4,128,32): the branch4..branch1 1x1 weights, input-channel major; ``scale`` /
    ``shift`` (4,32): their folded BN.  Inference only."""
    _require_device("spp_head", raw, skip, w_t, scale, shift)
    B, cr, H, W = raw.shape
    if cr != 64 or tuple(skip.shape) != (B, 128, H, W):
        raise ValueError("spp_head: expected raw (B,64,H,W) and skip (B,128,H,W), got %s and %s"
                         % (tuple(raw.shape), tuple(skip.shape)))
    if tuple(w_t.shape) != (4, 128, 32) or tuple(scale.shape) != (4, 32) or tuple(shift.shape) != (4, 32):
        raise ValueError("spp_head: w_t (4,128,32), scale/shift (4,32) expected")
    if H < 64 or W < 64:
        raise ValueError("spp_head: the 64x64 pooling branch needs H, W >= 64 at 1/4 resolution "
                         "(got %dx%d)" % (H, W))
    raw = raw.contiguous(memory_format=_CL2D)
    skip = skip.contiguous(memory_format=_CL2D)
    w_t, scale, shift = w_t.contiguous(), scale.contiguous(), shift.contiguous()
    lib = _lib.load()
    h8, w8 = H // 8, W // 8
    p8 = torch.empty((B, h8, w8, 128), device=raw.device, dtype=torch.float32)
    br = torch.empty(lib.dsm_spp_branch_floats(B, h8, w8), device=raw.device, dtype=torch.float32)
    out = torch.empty((B, 320, H, W), device=raw.device, dtype=torch.float32, memory_format=_CL2D)
    with torch.cuda.device(raw.device):
        with _timed("spp_pool8_kernel", 4.0 * (skip.numel() + p8.numel())):
            rc = lib.dsm_spp_pool8(_p(skip), _p(p8), B, H, W, _stream())
        _lib.check(rc, "dsm_spp_pool8")
        with _timed("spp_branches_kernel", 4.0 * (p8.numel() + br.numel())):
            rc = lib.dsm_spp_branches(_p(p8), _p(w_t), _p(scale), _p(shift), _p(br), B, h8, w8,
                                      _stream())
        _lib.check(rc, "dsm_spp_branches")
        with _timed("spp_concat_kernel", 4.0 * (raw.numel() + skip.numel() + out.numel())):
            ya = _ARENA.slot(out.device) if needs_amax() else None     # lastconv's x_amax, without a pass of its own
            rc = lib.dsm_spp_concat(_p(raw), _p(skip), _p(br), _p(out), B, H, W, _p(ya), _stream())
        _lib.check(rc, "dsm_spp_concat")
    if ya is not None:
        out._dsm_amax = ya
    return out


# ----------------------------------------------------------------------------
# Conv3d / ConvTranspose3d (k=3) with autograd: training through the 3-D trunk
# ----------------------------------------------------------------------------
def _wgrad_precision(x, g):
    """(precision, x_amax, g_amax) of a weight-gradient launch under the current ``conv_precision``."""
    mode = _OPTIONS["conv_precision"]
    if mode not in ("f16x2", "f16"):
        return _lib.DSM_PREC_F32, None, None
    return (_lib.DSM_PREC_F16X2 if mode == "f16x2" else _lib.DSM_PREC_F16), amax_of(x), amax_of(g)


def _wgrad(x_cl, g_cl, cx, cg, stride):
    """dW[g][c][tap] = sum_v X[v*stride + tap - 1][c] G[v][g]  ->  (cg, cx, 3, 3, 3)."""
    B = x_cl.shape[0]
    ws = torch.empty((cx // 32) * (cg // 32) * 27 * 1024, device=x_cl.device, dtype=torch.float32)
    dw = torch.empty((cg, cx, 3, 3, 3), device=x_cl.device, dtype=torch.float32)
    prec, xa, ga = _wgrad_precision(x_cl, g_cl)
    with torch.cuda.device(x_cl.device), _timed("conv3d_wgrad_%s_kernel<S=%d,%dx%d>" % (_OPTIONS["conv_precision"], stride, cx, cg),
                                                54.0 * cx * cg * B * g_cl.shape[2] * g_cl.shape[3] * g_cl.shape[4]):
        rc = _lib.load().dsm_conv3d_wgrad(_p(x_cl), _p(g_cl), _p(ws), _p(dw), B, cx, cg,
                                          x_cl.shape[2], x_cl.shape[3], x_cl.shape[4],
                                          g_cl.shape[2], g_cl.shape[3], g_cl.shape[4], stride,
                                          _conv_flags(), prec, _p(xa), _p(ga), _stream())
    _lib.check(rc, "dsm_conv3d_wgrad")
    return dw


class Conv3dFunction(torch.autograd.Function):
    """y = conv(x, weight) (+ bias) with k = 3, padding 1: ``nn.Conv3d(stride 1|2)`` or
    ``nn.ConvTranspose3d(stride 2, output_padding 1)`` -- forward and both gradients on the
    gfx950 kernels.  bwd-data is a convolution with re-packed weights on the forward kernels;
    bwd-weight is ``dsm_conv3d_wgrad``.  In the reference this is autograd through nn.Conv3d."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, transposed):
        _require_device("Conv3dFunction", x, weight, bias)
        x = carry_amax(to_channels_last_3d(x), x)
        cout = weight.shape[1] if transposed else weight.shape[0]
        packed = pack_conv3d_weight(weight, transposed)
        shift = None if bias is None else bias.detach().contiguous()
        scale = None if bias is None else torch.ones_like(shift)
        y = conv3d_block(x, packed, cout, scale, shift, None, stride, transposed, 0)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, transposed, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, transposed, has_bias = ctx.cfg
        gy = carry_amax(to_channels_last_3d(gy), gy)
        if needs_amax():
            amax_of(x), amax_of(gy)             # one pass each at most, shared by bwd-data and bwd-weight
        B, cin = x.shape[0], x.shape[1]
        cout = gy.shape[1]
        dx = dw = db = None
        w = weight.detach()
        if cout == 1 and transposed:                    # GCNet's head l37: ConvTranspose3d(C -> 1, s2)
            dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            dw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
            with torch.cuda.device(x.device):
                rc = _lib.load().dsm_deconv3d_cout1_bwd(
                    _p(x), _p(gy), _p(w.contiguous()), _p(dx), _p(dw), B, cin, x.shape[2],
                    x.shape[3], x.shape[4], gy.shape[2], gy.shape[3], gy.shape[4], _stream())
            _lib.check(rc, "dsm_deconv3d_cout1_bwd")
        elif cout == 1 and stride != 1:
            raise NotImplementedError("Conv3dFunction.backward: a stride-%d convolution to one "
                                      "channel is not a layer of the reference" % stride)
        elif cout == 1:                                 # classifier head, stride 1, not transposed
            packed = pack_conv3d_weight(w, False)       # [27][Cin]
            dwt = torch.empty(27 * cin, device=x.device, dtype=torch.float32)
            dx = torch.empty_like(x)
            with torch.cuda.device(x.device):
                rc = _lib.load().dsm_conv3d_cout1_bwd(
                    _p(x), _p(gy), _p(packed), _p(dx) if ctx.needs_input_grad[0] else None,
                    _p(dwt) if ctx.needs_input_grad[1] else None, B, cin, x.shape[2], x.shape[3],
                    x.shape[4], _stream())
            _lib.check(rc, "dsm_conv3d_cout1_bwd")
            if not ctx.needs_input_grad[0]:
                dx = None
            if ctx.needs_input_grad[1]:
                dw = dwt.view(27, cin).t().contiguous().view(1, cin, 3, 3, 3)
        else:
            if ctx.needs_input_grad[0]:
                if transposed:                          # dX = conv_s2(dY, W as (out=Cin, in=Cout))
                    dx = conv3d_block(gy, pack_conv3d_weight(w, False), cin, stride=2,
                                      out_size=x.shape[2:])
                elif stride == 1:                       # dX = conv_s1(dY, flipped W^T)
                    wt = w.flip(2, 3, 4).transpose(0, 1).contiguous()
                    dx = conv3d_block(gy, pack_conv3d_weight(wt, False), cin, stride=1)
                else:                                   # dX = convT_s2(dY, W), cropped to x
                    dx = conv3d_block(gy, pack_conv3d_weight(w, True), cin, stride=2,
                                      transposed=True, out_size=x.shape[2:])
                if tuple(dx.shape) != tuple(x.shape):
                    raise RuntimeError("Conv3dFunction.backward: dX shape %s != %s"
                                       % (tuple(dx.shape), tuple(x.shape)))
            if ctx.needs_input_grad[1]:
                if transposed:                          # roles swapped: X := dY, G := x
                    dw = _wgrad(gy, x, cout, cin, 2)    # (Cin, Cout, 3,3,3)
                else:
                    dw = _wgrad(x, gy, cin, cout, stride)
        if has_bias and ctx.needs_input_grad[2]:
            db = gy.sum(dim=(0, 2, 3, 4))
        return dx, dw, db, None, None


def conv3d(x, weight, bias=None, stride=1, transposed=False):
    return Conv3dFunction.apply(x, weight, bias, int(stride), bool(transposed))


# ----------------------------------------------------------------------------
# train-mode BatchNorm3d + cropped skip add + ReLU, fused (csrc/bn3d.hip)
# ----------------------------------------------------------------------------
class BnAddRelu3dFunction(torch.autograd.Function):
    """out = relu?( batch_norm(y; batch statistics) (+ residual, cropped to the common corner) )
    for NDHWC fp32 volumes, with explicit backward -- two launches each way instead of the
    reference's nn.BatchNorm3d + myadd_3d + F.relu chain (and their autograd nodes).
    ``relu``: 0 none, 1 after the addition (PSMNet), 2 before it (GCNet).  Running statistics are
    updated in place as nn.BatchNorm3d does."""

    @staticmethod
    def forward(ctx, y, gamma, beta, residual, running_mean, running_var, relu, momentum, eps):
        _require_device("bn_add_relu3d", y, gamma, beta, residual, running_mean, running_var)
        y = to_channels_last_3d(y)
        B, C, Dy, Hy, Wy = y.shape
        a = _lib.Bn3dArgs()
        a.B, a.C, a.Dy, a.Hy, a.Wy = B, C, Dy, Hy, Wy
        oshape = (B, C, Dy, Hy, Wy)
        if residual is not None:
            if residual.shape[0] != B or residual.shape[1] != C:
                raise ValueError("bn_add_relu3d: residual has shape %s" % (tuple(residual.shape),))
            residual = to_channels_last_3d(residual)
            a.Dr, a.Hr, a.Wr = residual.shape[2:]
            oshape = (B, C, min(Dy, a.Dr), min(Hy, a.Hr), min(Wy, a.Wr))
            a.residual = residual.data_ptr()
        out = torch.empty(oshape, device=y.device, dtype=torch.float32, memory_format=_CL3D)
        affine = torch.empty(4 * C, device=y.device, dtype=torch.float32)
        ws = torch.empty(2 * C, device=y.device, dtype=torch.float64)
        a.y, a.out, a.affine, a.workspace = y.data_ptr(), out.data_ptr(), affine.data_ptr(), ws.data_ptr()
        a.gamma = None if gamma is None else gamma.data_ptr()
        a.beta = None if beta is None else beta.data_ptr()
        a.running_mean = None if running_mean is None else running_mean.data_ptr()
        a.running_var = None if running_var is None else running_var.data_ptr()
        a.relu, a.momentum, a.eps = int(relu), float(momentum), float(eps)
        oa = None
        if needs_amax():                               # the next convolution's x_amax, from this epilogue
            oa = _ARENA.slot(y.device)
            a.out_amax = oa.data_ptr()
        with torch.cuda.device(y.device), _timed("bn3d_train_fwd_kernels", 4.0 * (2 * y.numel() + out.numel())):
            rc = _lib.load().dsm_bn3d_train_fwd(ctypes.byref(a), _stream())
        _lib.check(rc, "dsm_bn3d_train_fwd")
        ctx.save_for_backward(y, out if relu == 1 else None, affine)
        ctx.cfg = (int(relu), None if residual is None else tuple(residual.shape), gamma is not None,
                   beta is not None)
        if oa is not None:
            out._dsm_amax = oa
            ctx.out_amax = oa
        return out

    @staticmethod
    def backward(ctx, gout):
        y, out, affine = ctx.saved_tensors
        relu, rshape, has_gamma, has_beta = ctx.cfg
        gout = gout.contiguous(memory_format=_CL3D)
        B, C, Dy, Hy, Wy = y.shape
        a = _lib.Bn3dArgs()
        a.B, a.C, a.Dy, a.Hy, a.Wy = B, C, Dy, Hy, Wy
        dy = torch.empty_like(y, memory_format=_CL3D)
        dres = None
        if rshape is not None:
            a.Dr, a.Hr, a.Wr = rshape[2:]
            if ctx.needs_input_grad[3]:
                dres = torch.empty(rshape, device=y.device, dtype=torch.float32, memory_format=_CL3D)
                a.dresidual = dres.data_ptr()
            else:
                a.residual = y.data_ptr()          # only marks "a residual shaped the output corner"
        ws = torch.empty(2 * C, device=y.device, dtype=torch.float64)
        a.y, a.affine, a.workspace = y.data_ptr(), affine.data_ptr(), ws.data_ptr()
        a.out = None if out is None else out.data_ptr()
        a.gout, a.dy, a.relu = gout.data_ptr(), dy.data_ptr(), relu
        if needs_amax():                               # the maxima of the gradients this node hands on
            dy._dsm_amax = _ARENA.slot(y.device)
            a.dy_amax = dy._dsm_amax.data_ptr()
            if dres is not None:
                dres._dsm_amax = _ARENA.slot(y.device)
                a.dres_amax = dres._dsm_amax.data_ptr()
        with torch.cuda.device(y.device), _timed("bn3d_train_bwd_kernels", 4.0 * (3 * y.numel() + gout.numel())):
            rc = _lib.load().dsm_bn3d_train_bwd(ctypes.byref(a), _stream())
        _lib.check(rc, "dsm_bn3d_train_bwd")
        dbeta = ws[:C].float() if has_beta else None
        dgamma = ws[C:].float() if has_gamma else None
        return dy, dgamma, dbeta, dres, None, None, None, None, None


def bn_add_relu3d(y, gamma, beta, residual, running_mean, running_var, relu, momentum, eps):
    return BnAddRelu3dFunction.apply(y, gamma, beta, residual, running_mean, running_var, int(relu),
                                     float(momentum), float(eps))


def stage_images_nhwc16(left, right=None):
    """The towers' input staging in one launch: (B,C,H,W) image(s), C <= 16, -> (B or 2B, 16, H, W)
    channels_last with zero channels C..15; with ``right`` the two views share the batch
    (left first), as PSMNet's eval forward feeds them."""
    _require_device("stage_images_nhwc16", left, right)
    left = left.contiguous()
    B, C, H, W = left.shape
    if right is not None:
        if tuple(right.shape) != tuple(left.shape):
            raise ValueError("stage_images_nhwc16: the two views differ in shape")
        right = right.contiguous()
    out = torch.empty(((1 if right is None else 2) * B, 16, H, W), device=left.device,
                      dtype=torch.float32, memory_format=_CL2D)
    with torch.cuda.device(left.device):
        rc = _lib.load().dsm_stage_images_nhwc16(_p(left), _p(right), _p(out), B, C, H, W, _stream())
    _lib.check(rc, "dsm_stage_images_nhwc16")
    return out


# ----------------------------------------------------------------------------
# the 2-D towers in training: convolution with explicit gradients, batch-statistics BN
# ----------------------------------------------------------------------------
# (stride, Cout/32, k, dilation) variants of the 2-D MFMA convolution compiled in csrc/conv3d.hip
_CONV2D_VARIANTS = {(1, 1, 3, 1), (1, 2, 3, 1), (1, 4, 3, 1), (1, 4, 3, 2), (2, 1, 3, 1),
                    (2, 2, 3, 1), (1, 1, 1, 1), (1, 4, 1, 1), (2, 2, 1, 1)}


def conv2d_variant(cout, stride, k, dilation):
    return cout in (32, 64, 128) and (stride, cout // 32, k, dilation) in _CONV2D_VARIANTS


def _wgrad2d(x_cl, g_cl, stride, dilation):
    """dW[g][c][ky][kx] = sum_v X[v*stride + (k - 1)*dilation][c] G[v][g]  ->  (cg, cx, 3, 3)."""
    B, cx, Hx, Wx = x_cl.shape
    _, cg, Hg, Wg = g_cl.shape
    ws = torch.empty((cx // 32) * (cg // 32) * 9 * 1024, device=x_cl.device, dtype=torch.float32)
    dw = torch.empty((cg, cx, 3, 3), device=x_cl.device, dtype=torch.float32)
    prec, xa, ga = _wgrad_precision(x_cl, g_cl)
    with torch.cuda.device(x_cl.device), _timed("conv2d_wgrad_kernel", 18.0 * cx * cg * B * Hg * Wg):
        rc = _lib.load().dsm_conv2d_wgrad(_p(x_cl), _p(g_cl), _p(ws), _p(dw), B, cx, cg, Hx, Wx,
                                          Hg, Wg, int(stride), int(dilation), _conv_flags(), prec,
                                          _p(xa), _p(ga), _stream())
    _lib.check(rc, "dsm_conv2d_wgrad")
    return dw


class Conv2dFunction(torch.autograd.Function):
    """y = conv2d(x, weight), k in {1, 3}, padding = dilation * (k // 2), no bias: the ``nn.Conv2d`` of
    ``convbn`` (models/psmnet/submodule.py:10-13) with the forward, bwd-data (a convolution again,
    flipped transposed weights) and bwd-weight (``dsm_conv2d_wgrad``; 1x1: one GEMM) on the gfx950
    kernels, NHWC.  Gradients no kernel here covers (stride-2 bwd-data, the 3-channel image layer's
    bwd-weight, channel counts outside the compiled variants) come from
    ``aten.convolution_backward`` on the same tensors.  In the reference this is autograd through
    nn.Conv2d."""

    @staticmethod
    def forward(ctx, x, weight, stride, dilation):
        _require_device("Conv2dFunction", x, weight)
        cout, cin, k, _ = weight.shape
        x = x.contiguous(memory_format=_CL2D)
        xs = x
        if cin % 16:                                   # the image: 3 -> 16 staged channels
            xs = torch.zeros((x.shape[0], (cin + 15) // 16 * 16) + tuple(x.shape[2:]), device=x.device,
                             dtype=x.dtype).contiguous(memory_format=_CL2D)
            xs[:, :cin] = x
        packed = pack_conv2d_weight(weight, xs.shape[1])
        y = conv2d_block(xs, packed, cout, stride=stride, k=k, dilation=dilation)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (int(stride), int(dilation))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, dil = ctx.cfg
        cout, cin, k, _ = weight.shape
        gy = gy.contiguous(memory_format=_CL2D)
        w = weight.detach()
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dx = dw = None
        if need_dx and stride == 1 and conv2d_variant(cin, 1, k, dil) and cout % 16 == 0:
            wt = (w.flip(2, 3) if k == 3 else w).transpose(0, 1).contiguous()
            dx = conv2d_block(gy, pack_conv2d_weight(wt), cin, stride=1, k=k, dilation=dil)
        if need_dw and k == 3 and cin % 32 == 0 and cout % 32 == 0:
            dw = _wgrad2d(x, gy, stride, dil)
        elif need_dw and k == 1:
            xg = x if stride == 1 else x[:, :, ::stride, ::stride]
            dw = torch.matmul(gy.permute(0, 2, 3, 1).reshape(-1, cout).t(),
                              xg.permute(0, 2, 3, 1).reshape(-1, cin)).view(cout, cin, 1, 1)
        mask = [need_dx and dx is None, need_dw and dw is None, False]
        if mask[0] or mask[1]:
            pad = dil * (k // 2)
            rdx, rdw, _ = torch.ops.aten.convolution_backward(
                gy, x, w, None, [stride, stride], [pad, pad], [dil, dil], False, [0, 0], 1, mask)
            dx = rdx if mask[0] else dx
            dw = rdw if mask[1] else dw
        return dx, dw, None, None


def conv2d(x, weight, stride=1, dilation=1):
    return Conv2dFunction.apply(x, weight, int(stride), int(dilation))


def bn_add_relu2d(y, gamma, beta, residual, running_mean, running_var, relu, momentum, eps):
    """Train-mode BatchNorm2d (+ skip add) (+ ReLU) on NHWC maps: the 3-D kernels on (B, C, 1, H, W)
    views of the same memory."""
    y = y.contiguous(memory_format=_CL2D).unsqueeze(2)
    if residual is not None:
        residual = residual.contiguous(memory_format=_CL2D).unsqueeze(2)
    return bn_add_relu3d(y, gamma, beta, residual, running_mean, running_var, relu, momentum,
                         eps).squeeze(2)


# ----------------------------------------------------------------------------
# decoder level of DispNetC / iResNet: deconv bias + ReLU + x2 upsampling + myCat2d, one launch
# ----------------------------------------------------------------------------
def decoder_level(deconv, x, pr, skip):
    """``myCat2d(deconv(x), upsample(pr), skip)`` of the reference's decoders
    (models/dispnetcorr.py:89-132, models/iresnet.py:119-161,186-193; util_fun.py:7-15) with
    ``deconv`` = ``Sequential(ConvTranspose2d(bias), ReLU)`` or a bare ``ConvTranspose2d``.
    Eval mode on the GPU: the transposed convolution runs without its bias (stock kernel) and ONE
    launch does bias + ReLU + bilinear x2 upsampling of ``pr`` + the crops + the concatenation;
    otherwise (training, autograd, CPU) the stock ops."""
    import torch.nn as nn
    import torch.nn.functional as F
    conv = deconv[0] if isinstance(deconv, nn.Sequential) else deconv
    relu = isinstance(deconv, nn.Sequential) and len(deconv) > 1
    fast = (x.is_cuda and not torch.is_grad_enabled() and x.dtype == torch.float32 and
            isinstance(conv, nn.ConvTranspose2d) and
            (not relu or (len(deconv) == 2 and isinstance(deconv[1], nn.ReLU))))
    if not fast:
        seq = [deconv(x)]
        if pr is not None:
            seq.append(F.interpolate(pr, scale_factor=2, mode="bilinear", align_corners=False))
        if skip is not None:
            seq.append(skip)
        h = min(t.shape[2] for t in seq)
        w = min(t.shape[3] for t in seq)
        return torch.cat([t[:, :, :h, :w] for t in seq], dim=1)
    up = F.conv_transpose2d(x, conv.weight, None, conv.stride, conv.padding, conv.output_padding,
                            conv.groups, conv.dilation).contiguous()
    B, Cu, Hu, Wu = up.shape
    pr = None if pr is None else pr.contiguous()
    skip = None if skip is None else skip.contiguous()
    Cp, Hp, Wp = (0, 0, 0) if pr is None else pr.shape[1:]
    Cs, Hs, Ws = (0, 0, 0) if skip is None else skip.shape[1:]
    h = min([Hu] + ([2 * Hp] if Cp else []) + ([Hs] if Cs else []))
    w = min([Wu] + ([2 * Wp] if Cp else []) + ([Ws] if Cs else []))
    out = torch.empty((B, Cu + Cp + Cs, h, w), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device), _timed("decoder_cat_kernel", 8.0 * out.numel()):
        rc = _lib.load().dsm_decoder_cat(_p(up), _p(conv.bias), _p(pr), _p(skip), _p(out), B, Cu, Cp, Cs,
                                         Hu, Wu, Hp, Wp, Hs, Ws, int(relu), _stream())
    _lib.check(rc, "dsm_decoder_cat")
    return out
