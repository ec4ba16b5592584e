"""Layer toolbox with the reference's names (models/util_conv.py), MI355X-backed where the
cost-volume path runs.

  Corr1d                      -> HIP correlation kernel (util_conv.py:56-86)
  conv3d_bn / deconv3d_bn     -> blocks3d.ConvBN3d, one launch (util_conv.py:150-179)
  conv2d_bn / deconv2d_bn / conv_res / net_init -- 2-D layers outside the hot path: stock
      torch modules in the same Sequential layout (state-dict keys match the reference).
"""
import math

import torch
import torch.nn as nn

from .. import costvolume as cv
from ..blocks3d import Conv3dHip, ConvBN3d, ConvTranspose3dHip

flag_bn = False
flag_bias_default = True


def _default_act():
    return nn.ReLU(inplace=True)


activefun_default = _default_act()


class Corr1d(nn.Module):
    """``Corr1d(kernel_size, stride, D, simfun)``: plane ``i`` is the channel dot product of
    ``fL[..., x]`` with ``fR[..., x - i*stride]`` (zero for ``x < i*stride``), optionally box
    filtered.  Only the default similarity is implemented on the device."""

    def __init__(self, kernel_size=1, stride=1, D=1, simfun=None):
        super(Corr1d, self).__init__()
        if simfun is not None:
            raise NotImplementedError("Corr1d on the MI355X path supports the default "
                                      "dot-product similarity only (util_conv.py:68-69)")
        self.kernel_size, self.stride, self.D = kernel_size, stride, D

    def forward(self, fL, fR):
        return cv.corr1d(fL, fR, self.D, self.stride, self.kernel_size)

    def extra_repr(self):
        return "kernel_size=%d, stride=%d, D=%d" % (self.kernel_size, self.stride, self.D)


def _wrap(conv, out_planes, bn, activefun, norm):
    if not bn and not activefun:
        return conv
    layers = [conv]
    if bn:
        layers.append(norm(out_planes))
    if activefun:
        layers.append(activefun)
    return nn.Sequential(*layers)


def conv2d_bn(in_planes, out_planes, kernel_size=3, stride=1, flag_bias=flag_bias_default,
              bn=flag_bn, activefun=activefun_default):
    assert kernel_size % 2 == 1
    conv = nn.Conv2d(in_planes, out_planes, kernel_size, stride, padding=(kernel_size - 1) // 2,
                     bias=flag_bias)
    return _wrap(conv, out_planes, bn, activefun, nn.BatchNorm2d)


def deconv2d_bn(in_planes, out_planes, kernel_size=4, stride=2, flag_bias=flag_bias_default,
                bn=flag_bn, activefun=activefun_default):
    assert stride > 1
    p = (kernel_size - 1) // 2
    conv = nn.ConvTranspose2d(in_planes, out_planes, kernel_size, stride, padding=p,
                              output_padding=stride - (kernel_size - 2 * p), bias=flag_bias)
    return _wrap(conv, out_planes, bn, activefun, nn.BatchNorm2d)


def conv3d_bn(in_planes, out_planes, kernel_size=3, stride=1, flag_bias=flag_bias_default,
              bn=flag_bn, activefun=activefun_default):
    """3-D conv + BN + activation as ONE fused block (children '0','1','2' as in the reference)."""
    if not bn and not activefun:
        return Conv3dHip(in_planes, out_planes, kernel_size, stride,
                         padding=(kernel_size - 1) // 2, bias=flag_bias)
    conv = nn.Conv3d(in_planes, out_planes, kernel_size, stride, padding=(kernel_size - 1) // 2,
                     bias=flag_bias)
    return ConvBN3d(conv, nn.BatchNorm3d(out_planes) if bn else None, activefun or None)


def deconv3d_bn(in_planes, out_planes, kernel_size=4, stride=2, flag_bias=flag_bias_default,
                bn=flag_bn, activefun=activefun_default):
    """Transposed 3-D conv + BN + activation, fused.  The reference instantiates BatchNorm2d
    here (util_conv.py:176), which cannot take 5-D input; per-channel BN over (N,D,H,W) is
    what is meant, i.e. BatchNorm3d -- identical parameters and state-dict keys."""
    assert stride > 1
    p = (kernel_size - 1) // 2
    if not bn and not activefun:
        return ConvTranspose3dHip(in_planes, out_planes, kernel_size, stride, padding=p,
                                  output_padding=stride - (kernel_size - 2 * p), bias=flag_bias)
    conv = nn.ConvTranspose3d(in_planes, out_planes, kernel_size, stride, padding=p,
                              output_padding=stride - (kernel_size - 2 * p), bias=flag_bias)
    return ConvBN3d(conv, nn.BatchNorm3d(out_planes) if bn else None, activefun or None)


class BasicBlock(nn.Module):
    """2-D ResNet block of GCNet's tower (util_conv.py:181-210)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super(BasicBlock, self).__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        if x.is_cuda and not self.training and not torch.is_grad_enabled():
            # eval on the GPU: both 3x3 convolutions on the MFMA kernel (blocks2d), BN folded,
            # ReLU and the skip addition in their epilogues -- two launches per block
            from ..blocks2d import _Folded2d, run_conv2d, run_basicblock_layers
            if not hasattr(self, "_folds"):
                self._folds = (_Folded2d(), _Folded2d())
            if self.downsample is None:             # both convolutions, the add and the ReLU in ONE launch
                y = run_basicblock_layers(self._folds[0], self.conv1, self.bn1, self._folds[1], self.conv2,
                                          self.bn2, x, True)
                if y is not None:
                    return y
            skip = x if self.downsample is None else self.downsample(x)
            y = run_conv2d(self._folds[0], self.conv1, self.bn1, x, relu=True)
            return run_conv2d(self._folds[1], self.conv2, self.bn2, y, residual=skip, relu=True)
        y = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k != "_folds":                      # fold caches are per instance, re-made on use
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new


def conv_res(inplanes, planes, blocks, stride=1):
    downsample = None
    if stride != 1 or inplanes != planes:
        downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False),
                                   nn.BatchNorm2d(planes))
    layers = [BasicBlock(inplanes, planes, stride, downsample)]
    layers += [BasicBlock(planes, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


def net_init(net):
    """He-normal for Conv{1,2,3}d weights, BN to (1, 0); biases and transposed convolutions
    keep torch's default (util_conv.py:32-53)."""
    for m in net.modules():
        if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Conv3d)):
            fan = m.out_channels
            for k in m.kernel_size:
                fan *= k
            m.weight.data.normal_(0, math.sqrt(2.0 / fan))
        elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.weight.data.fill_(1)
            m.bias.data.zero_()
