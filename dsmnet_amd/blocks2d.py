"""2-D tower blocks: ``Sequential(Conv2d, BatchNorm2d)`` with the
reference's child indices (``convbn`` -- models/psmnet/submodule.py:10-13), run in eval mode
on the GPU as ONE launch of the MFMA convolution kernel with folded BN, fused ReLU and skip
add, on NHWC (``torch.channels_last``) maps.

Training mode, CPU tensors, and layer shapes the kernels do not cover run the stock torch
modules (autograd included); the SPP branches' 1x1 convolution with padding 1 has its own kernels
(csrc/spp.hip, ``costvolume.spp_head``).  Stride-1 3x3 layers compute on the bf16 pipe (bf16x3,
DESIGN.md 3.2a), stride-2 and 1x1 layers on the fp32-input MFMA.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import costvolume as cv
from . import blocks3d
from .blocks3d import _versions

# Training through the 2-D towers (DSM_TRAIN_2D, read by this host module):
#   auto  (default) a layer runs on costvolume.Conv2dFunction (forward, backward-data and weight
#         gradient on the MFMA kernels) when its output has at least _TRAIN_2D_MIN_PIXELS pixels, on
#         the stock torch layer otherwise; BatchNorm2d stays stock;
#   conv  every eligible layer on Conv2dFunction, stock BatchNorm2d;
#   fused the same plus the fused batch-statistics BN + skip + ReLU block (csrc/bn3d.hip);
#   stock the stock torch layers only.
# Measured (one replayed PSMNet step): at BASELINE config #5's shape (540 x 960,
# 4 pairs: 1/4-resolution maps of 135 x 240 x 4) conv 177 / fused 179 / stock 192 ms; at the 256 x 512
# crop (maps of 64 x 128 and smaller) stock 27.0 / conv 30.0 / fused 32.1 ms -- there the kernels
# launch 16-64 workgroups and the per-step weight re-packing dominates.
_TRAIN_2D_MODE = __import__("os").environ.get("DSM_TRAIN_2D", "auto")
_FUSED_TRAIN_2D = _TRAIN_2D_MODE in ("auto", "fused", "conv")
_FUSED_TRAIN_2D_BN = _TRAIN_2D_MODE == "fused"
_TRAIN_2D_MIN_PIXELS = int(__import__("os").environ.get("DSM_TRAIN_2D_MIN_PIXELS", 65536)) if _TRAIN_2D_MODE == "auto" else 0


class own_kernels_only(object):
    """``with own_kernels_only():`` every tower layer ``costvolume.Conv2dFunction`` covers runs on
    it whatever its size (the ``auto`` threshold is lifted).  Used by ``calibrate.calibrate_batchnorm``:
    a train-mode pass that never enters MIOpen's convolution solver search, whose first call per
    shape costs tens of milliseconds to seconds -- times eight ranks starting on one host."""

    def __enter__(self):
        global _TRAIN_2D_MIN_PIXELS, _FUSED_TRAIN_2D
        self.saved = (_TRAIN_2D_MIN_PIXELS, _FUSED_TRAIN_2D)
        _TRAIN_2D_MIN_PIXELS, _FUSED_TRAIN_2D = 0, True
        return self

    def __exit__(self, *exc):
        global _TRAIN_2D_MIN_PIXELS, _FUSED_TRAIN_2D
        _TRAIN_2D_MIN_PIXELS, _FUSED_TRAIN_2D = self.saved
        return False


def fused_ok(conv, x):
    """Can ``conv`` (an nn.Conv2d) run on the MFMA kernel for input ``x``?"""
    if not (isinstance(conv, nn.Conv2d) and x.is_cuda and x.dtype == torch.float32):
        return False
    k, s, d, p = conv.kernel_size, conv.stride, conv.dilation, conv.padding
    if k[0] != k[1] or s[0] != s[1] or d[0] != d[1] or p[0] != p[1] or conv.groups != 1:
        return False
    if k[0] not in (1, 3) or s[0] not in (1, 2) or d[0] not in (1, 2):
        return False
    if p[0] != d[0] * (k[0] - 1) // 2 or conv.padding_mode != "zeros":
        return False
    if conv.out_channels not in (32, 64, 128):
        return False
    cin = x.shape[1]
    if cin % 16 != 0 or cin < conv.in_channels:
        return False
    # variants compiled in csrc/conv3d.hip (DSM_CASE2D)
    key = (s[0], conv.out_channels // 32, k[0], d[0])
    return key in {(1, 1, 3, 1), (1, 2, 3, 1), (1, 4, 3, 1), (1, 4, 3, 2), (2, 1, 3, 1),
                   (2, 2, 3, 1), (1, 1, 1, 1), (1, 4, 1, 1), (2, 2, 1, 1)}


class _Folded2d(object):
    def __init__(self):
        self.key = None
        self.packed = self.scale = self.shift = None

    def get(self, conv, bn, cin_padded):
        srcs = [conv.weight, conv.bias]
        if bn is not None:
            srcs += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        key = _versions(*srcs) + (cin_padded,)
        if key != self.key:
            with torch.no_grad():
                self.packed = cv.pack_conv2d_weight(conv.weight, cin_padded)
                if bn is not None:
                    inv = torch.rsqrt(bn.running_var + bn.eps)
                    scale = bn.weight * inv if bn.weight is not None else inv
                    shift = -bn.running_mean * scale
                    if bn.bias is not None:
                        shift = shift + bn.bias
                    if conv.bias is not None:
                        shift = shift + conv.bias * scale
                    self.scale, self.shift = scale.contiguous(), shift.contiguous()
                elif conv.bias is not None:
                    self.scale = torch.ones_like(conv.bias)
                    self.shift = conv.bias.detach().clone()
                else:
                    self.scale = self.shift = None
            self.key = key
        return self.packed, self.scale, self.shift


def train_ok(conv, x):
    """Can ``conv`` run on ``costvolume.Conv2dFunction`` (forward on the MFMA kernels, explicit
    gradients) for the NCHW-shaped fp32 CUDA tensor ``x``?"""
    if not (_FUSED_TRAIN_2D and isinstance(conv, nn.Conv2d) and torch.is_tensor(x) and x.is_cuda and
            x.dtype == torch.float32 and conv.weight.dtype == torch.float32):
        return False
    k, s, d, p = conv.kernel_size, conv.stride, conv.dilation, conv.padding
    if k[0] != k[1] or s[0] != s[1] or d[0] != d[1] or p[0] != p[1] or conv.groups != 1:
        return False
    if conv.bias is not None or conv.padding_mode != "zeros" or p[0] != d[0] * (k[0] // 2):
        return False
    if x.shape[0] * ((x.shape[2] - 1) // s[0] + 1) * ((x.shape[3] - 1) // s[0] + 1) < _TRAIN_2D_MIN_PIXELS:
        return False
    return x.shape[1] == conv.in_channels and cv.conv2d_variant(conv.out_channels, s[0], k[0], d[0])


def _run_conv2d_autograd(conv, bn, x, residual, relu):
    """Training / autograd through a tower block: convolution with explicit gradients on the HIP
    kernels, then batch-statistics BatchNorm + skip add + ReLU as one autograd node of two launches
    each way (csrc/bn3d.hip on the (B, C, 1, H, W) view) -- instead of nn.Conv2d + nn.BatchNorm2d +
    add + ReLU and their autograd nodes (models/psmnet/submodule.py:10-13,24-46)."""
    y = cv.conv2d(x, conv.weight, conv.stride[0], conv.dilation[0])
    if (bn is not None and bn.training and _FUSED_TRAIN_2D_BN and blocks3d._FUSED_TRAIN_BN and
            y.shape[1] % 4 == 0 and y.shape[1] <= 256):
        momentum = bn.momentum if bn.momentum is not None else 0.1
        out = cv.bn_add_relu2d(y, bn.weight, bn.bias, residual,
                               bn.running_mean if bn.track_running_stats else None,
                               bn.running_var if bn.track_running_stats else None,
                               1 if relu else 0, momentum, bn.eps)
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        blocks3d._bump_running_stats(bn)      # running statistics changed through raw pointers
        return out
    if bn is not None:
        y = bn(y)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def run_conv2d(folded, conv, bn, x, residual=None, relu=False):
    """conv (+BN) (+skip) (+ReLU): fused when possible, stock torch otherwise."""
    training = bn is not None and bn.training
    if not training and not torch.is_grad_enabled() and fused_ok(conv, x):
        packed, scale, shift = folded.get(conv, bn, x.shape[1])
        return cv.conv2d_block(x, packed, conv.out_channels, scale, shift, residual,
                               stride=conv.stride[0], relu=1 if relu else 0,
                               k=conv.kernel_size[0], dilation=conv.dilation[0])
    if train_ok(conv, x):
        return _run_conv2d_autograd(conv, bn, x, residual, relu)
    if x.shape[1] != conv.in_channels:            # zero-padded staging channels
        x = x[:, : conv.in_channels]
    y = conv(x)
    if bn is not None:
        y = bn(y)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def run_basicblock_layers(fold1, c1, b1, fold2, c2, b2, x, relu_after, skip=True):
    """``bn2(conv2(relu(bn1(conv1(x))))) + x`` (``relu_after``: ReLU after the add) as ONE launch where
    ``costvolume.basicblock2d`` covers it -- eval mode on the GPU, no gradient, two stride-1 3x3
    convolutions of 32 or 64 channels, each with its BatchNorm -- else None (the caller runs the two
    layers)."""
    if b1 is None or b2 is None or c1.training or b1.training or b2.training or torch.is_grad_enabled():
        return None
    if not (fused_ok(c1, x) and fused_ok(c2, x) and c1.kernel_size[0] == 3 and c2.kernel_size[0] == 3 and
            c2.stride[0] == 1 and c2.dilation[0] == 1 and c2.in_channels == c2.out_channels == c1.out_channels and
            cv.basicblock2d_ok(x, c1.in_channels, c1.out_channels, c1.stride[0], c1.dilation[0])):
        return None
    p1, s1, h1 = fold1.get(c1, b1, x.shape[1])
    p2, s2, h2 = fold2.get(c2, b2, x.shape[1])
    return cv.basicblock2d(x, p1, s1, h1, p2, s2, h2, relu=relu_after, skip=skip)


def run_basicblock(cb1, cb2, x, downsample):
    """PSMNet's BasicBlock (models/psmnet/submodule.py:24-46): ``cb2(cb1(x, relu=True), residual=x)`` in one
    launch, or None."""
    if downsample is not None or not isinstance(cb1, ConvBN2d) or not isinstance(cb2, ConvBN2d):
        return None
    return run_basicblock_layers(cb1._folded, cb1[0], cb1[1], cb2._folded, cb2[0], cb2[1], x, False)


class ConvBN2d(nn.Sequential):
    """``convbn``'s Sequential(Conv2d, BatchNorm2d); ``forward(x)`` equals it."""

    def __init__(self, conv, bn):
        super(ConvBN2d, self).__init__(conv, bn)
        self._folded = _Folded2d()

    def forward(self, x, residual=None, relu=False):
        return run_conv2d(self._folded, self[0], self[1], x, residual, relu)


def stage_image_nhwc16(img):
    """(B,3,H,W) image -> (B,16,H,W) channels_last with zero channels 3..15, the 16-channel
    granularity the MFMA kernel stages (the first convolution's weights are zero-padded to match)."""
    return cv.stage_images_nhwc16(img, None)
