"""PSMNet building blocks with the reference's names and constructor signatures
(models/psmnet/submodule.py of sunshinnnn/DSMnet), MI355X-backed where the hot path runs.

  convbn_3d            -> blocks3d.ConvBN3d (same children: '0' Conv3d, '1' BatchNorm3d)
  disparityregression  -> HIP expectation over the disparity axis (submodule.py:56-63)
  convbn / BasicBlock / feature_extraction -- the 2-D tower: same
      module tree so reference checkpoints load; in eval mode every convolution runs on the
      MFMA kernels (blocks2d) and the SPP head on csrc/spp.hip; training takes stock layers.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import blocks2d
from ... import costvolume as cv
from ...blocks2d import ConvBN2d, _Folded2d, run_conv2d, stage_image_nhwc16
from ...blocks3d import ConvBN3d, _versions

# DSM_TRAIN_SPP=interp (read by this host module) keeps F.interpolate under autograd (A/B runs)
_SPP_MATMUL = __import__("os").environ.get("DSM_TRAIN_SPP", "matmul") != "interp"
# DSM_TRAIN_LAYOUT=nhwc: the stock train-mode tower layers run on channels_last maps (A/B runs)
_TRAIN_NHWC = __import__("os").environ.get("DSM_TRAIN_LAYOUT", "nchw") == "nhwc"


def convbn(in_planes, out_planes, kernel_size, stride, pad, dilation):
    # the reference pads by `dilation` whatever `pad` says (submodule.py:10-13); kept.
    del pad
    return ConvBN2d(
        nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride,
                  padding=dilation, dilation=dilation, bias=False),
        nn.BatchNorm2d(out_planes))


def convbn_3d(in_planes, out_planes, kernel_size, stride, pad):
    conv = nn.Conv3d(in_planes, out_planes, kernel_size=kernel_size, padding=pad,
                     stride=stride, bias=False)
    return ConvBN3d(conv, nn.BatchNorm3d(out_planes))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride, downsample, pad, dilation):
        super(BasicBlock, self).__init__()
        self.conv1 = nn.Sequential(convbn(inplanes, planes, 3, stride, pad, dilation),
                                   nn.ReLU(inplace=True))
        self.conv2 = convbn(planes, planes, 3, 1, pad, dilation)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        fused = blocks2d.run_basicblock(self.conv1[0], self.conv2, x, self.downsample)
        if fused is not None:                              # both convolutions in one launch (eval, 64 channels)
            return fused
        y = self.conv1[0](x, relu=True)                    # convbn + the Sequential's ReLU
        skip = x if self.downsample is None else self.downsample(x)
        return self.conv2(y, residual=skip)                # convbn + skip add, no ReLU after


class disparityregression(nn.Module):
    """``forward(x)``: x is a (B, D, H, W) probability volume; returns sum_d d * x[:, d].

    PSMNet's heads no longer call this (the softmax and the upsampling are fused into
    ``costvolume.soft_argmin``); it is kept, by name, for callers that still hold
    probabilities.  sum_d d*p_d == soft-argmin of log p, so it runs on the same kernel."""

    def __init__(self, maxdisp):
        super(disparityregression, self).__init__()
        self.maxdisp = maxdisp

    def forward(self, x):
        return cv.soft_argmin(torch.log(x.clamp_min(1e-38)), None, negate=False)


class feature_extraction(nn.Module):
    """2-D SPP tower, (B,3,H,W) -> (B,32,H/4,W/4).  Eval mode on the GPU: every convolution on
    the MFMA kernel (blocks2d) and the SPP head in three launches (csrc/spp.hip); training and
    CPU tensors take the stock torch layers."""

    def __init__(self, align_corners=False):
        super(feature_extraction, self).__init__()
        self.align_corners = bool(align_corners)   # of the SPP branches' bilinear upsampling
        self.inplanes = 32
        stem = []
        for cin, stride in ((3, 2), (32, 1), (32, 1)):
            stem += [convbn(cin, 32, 3, stride, 1, 1), nn.ReLU(inplace=True)]
        self.firstconv = nn.Sequential(*stem)
        self.layer1 = self._make_layer(BasicBlock, 32, 3, 1, 1, 1)
        self.layer2 = self._make_layer(BasicBlock, 64, 16, 2, 1, 1)
        self.layer3 = self._make_layer(BasicBlock, 128, 3, 1, 1, 1)
        self.layer4 = self._make_layer(BasicBlock, 128, 3, 1, 1, 2)
        for idx, pool in ((1, 64), (2, 32), (3, 16), (4, 8)):
            setattr(self, "branch%d" % idx, nn.Sequential(
                nn.AvgPool2d((pool, pool), stride=(pool, pool)),
                convbn(128, 32, 1, 1, 0, 1), nn.ReLU(inplace=True)))
        self.lastconv = nn.Sequential(
            convbn(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
            nn.Conv2d(128, 32, kernel_size=1, padding=0, stride=1, bias=False))

    def _make_layer(self, block, planes, blocks, stride, pad, dilation):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = ConvBN2d(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1,
                          stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, pad, dilation)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes, 1, None, pad, dilation) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    @staticmethod
    def _bilinear_matrix(n_in, n_out, align_corners, device):
        """(n_out, n_in) interpolation matrix of F.interpolate(mode="bilinear") along one axis."""
        i = torch.arange(n_out, dtype=torch.float32, device=device)
        if align_corners:
            src = i * ((n_in - 1) / (n_out - 1)) if n_out > 1 else torch.zeros_like(i)
        else:
            src = ((i + 0.5) * (n_in / n_out) - 0.5).clamp_min(0.0)
        i0 = src.floor().long().clamp_max(n_in - 1)
        i1 = (i0 + 1).clamp_max(n_in - 1)
        w1 = src - i0.to(src.dtype)
        m = torch.zeros((n_out, n_in), dtype=torch.float32, device=device)
        m.scatter_add_(1, i0[:, None], (1.0 - w1)[:, None])
        m.scatter_add_(1, i1[:, None], w1[:, None])
        return m

    def _upsample(self, y, size):
        """The branches' bilinear upsampling (submodule.py:118-128).  Under autograd on the GPU it is
        two small matrix products Ry @ y @ Rx^T (bilinear interpolation is separable and linear), so
        that its backward is two matrix products as well: torch's upsample_bilinear2d_backward
        scatters with atomics and took 0.30 ms per branch and side, 2.4 ms of a 27 ms training step
        (profiles/r02_train_kernel_stats.csv)."""
        if not (y.is_cuda and torch.is_grad_enabled() and _SPP_MATMUL):
            return F.interpolate(y, size=size, mode="bilinear", align_corners=self.align_corners)
        key = (y.shape[2], y.shape[3], size[0], size[1], self.align_corners, y.device)
        cache = self.__dict__.setdefault("_up_cache", {})
        if key not in cache:
            cache[key] = (self._bilinear_matrix(y.shape[2], size[0], self.align_corners, y.device),
                          self._bilinear_matrix(y.shape[3], size[1], self.align_corners, y.device).t().contiguous())
        ry, rxt = cache[key]
        return torch.matmul(ry, torch.matmul(y.contiguous(), rxt))

    def _spp_stock(self, raw, skip):
        """The reference's SPP head with stock torch ops (training, autograd, CPU)."""
        size = skip.shape[2:]
        # SPP pooling as a cascade: AvgPool(8), then 2x2 averages give AvgPool(16/32/64)
        # exactly (equal windows that tile, floor semantics preserved) -- the direct 64x64
        # pooling kernel costs 2.5 ms per forward at 384x1280 (profiles/r01_a_summary.md).
        pooled, pyramid = skip, []
        for i in (4, 3, 2, 1):
            pooled = F.avg_pool2d(pooled, 8 if i == 4 else 2)
            branch = getattr(self, "branch%d" % i)
            y = branch[1](pooled, relu=True)          # convbn + ReLU ([0] is the AvgPool2d)
            pyramid.append(self._upsample(y, size))
        return torch.cat([raw, skip] + pyramid, dim=1)

    def _spp_params(self):
        """(w_t (4,128,32), scale (4,32), shift (4,32)) of branch4..branch1 with BN folded,
        cached until a parameter or running statistic changes."""
        convs = [getattr(self, "branch%d" % i)[1] for i in (4, 3, 2, 1)]
        srcs = [t for c in convs for t in (c[0].weight, c[1].weight, c[1].bias,
                                           c[1].running_mean, c[1].running_var)]
        key = _versions(*srcs)
        if getattr(self, "_spp_key", None) != key:
            with torch.no_grad():
                w_t = torch.stack([c[0].weight.reshape(32, 128).t() for c in convs]).contiguous()
                inv = [torch.rsqrt(c[1].running_var + c[1].eps) for c in convs]
                scale = torch.stack([c[1].weight * i for c, i in zip(convs, inv)])
                shift = torch.stack([c[1].bias - c[1].running_mean * s
                                     for c, s in zip(convs, scale)])
            self._spp_cache = (w_t, scale.contiguous(), shift.contiguous())
            self._spp_key = key
        return self._spp_cache

    def forward(self, x, staged=False):
        """``staged``: ``x`` is already the (B,16,H,W) NHWC staging of the image(s)
        (``costvolume.stage_images_nhwc16``) -- eval fast path only."""
        fast = x.is_cuda and not self.training and not torch.is_grad_enabled()
        if staged and not fast:
            raise RuntimeError("feature_extraction: staged input exists on the eval fast path only")
        if fast and not staged:
            x = stage_image_nhwc16(x)                  # NHWC, 3 -> 16 staged channels
        elif _TRAIN_NHWC and x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        x = self.firstconv[0](x, relu=True)
        fc2, fc4 = self.firstconv[2], self.firstconv[4]           # two convbn + ReLU of 32 channels: one launch
        y = blocks2d.run_basicblock_layers(fc2._folded, fc2[0], fc2[1], fc4._folded, fc4[0], fc4[1], x, True, skip=False)
        x = y if y is not None else fc4(fc2(x, relu=True), relu=True)
        x = self.layer1(x)
        raw = self.layer2(x)
        skip = self.layer4(self.layer3(raw))
        if skip.is_cuda and not self.training and not torch.is_grad_enabled() and not self.align_corners:
            x = cv.spp_head(raw, skip, *self._spp_params())       # three launches (csrc/spp.hip)
        else:
            x = self._spp_stock(raw, skip)
        x = self.lastconv[0](x, relu=True)
        if not hasattr(self, "_last_fold"):
            self._last_fold = _Folded2d()
        return run_conv2d(self._last_fold, self.lastconv[2], None, x)
