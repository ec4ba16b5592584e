"""PSMNet (stacked hourglass) on the MI355X cost-volume path.

Same class names, constructor arguments, attribute tree (= state-dict keys) and return
convention as models/psmnet/stackhourglass.py of the reference; the body of ``forward``
hosts the calls to the HIP ops, because the reference's volume build
(stackhourglass.py:124-133) and heads (:152-166) are inline code with no callable
boundary.
"""
import math

import torch
import torch.nn as nn

from ... import costvolume as cv
from ...blocks3d import Chain3d, ConvBN3d
from .submodule import convbn_3d, disparityregression, feature_extraction  # noqa: F401


def myadd_3d(tensor1, tensor2):
    """Crop both to the common (d, h, w), then add (stackhourglass.py:10-20).  On the fused
    path this happens inside the convolution epilogue; kept for direct callers."""
    assert tensor1.dim() == 5
    d, h, w = (min(a, b) for a, b in zip(tensor1.shape[2:], tensor2.shape[2:]))
    return tensor1[:, :, :d, :h, :w] + tensor2[:, :, :d, :h, :w]


def _deconvbn_3d(cin, cout):
    return ConvBN3d(nn.ConvTranspose3d(cin, cout, kernel_size=3, padding=1, output_padding=1,
                                       stride=2, bias=False), nn.BatchNorm3d(cout))


class hourglass(nn.Module):
    def __init__(self, inplanes):
        super(hourglass, self).__init__()
        c = inplanes
        self.conv1 = Chain3d(convbn_3d(c, c * 2, kernel_size=3, stride=2, pad=1), nn.ReLU(inplace=True))
        self.conv2 = convbn_3d(c * 2, c * 2, kernel_size=3, stride=1, pad=1)
        self.conv3 = Chain3d(convbn_3d(c * 2, c * 2, kernel_size=3, stride=2, pad=1), nn.ReLU(inplace=True))
        self.conv4 = Chain3d(convbn_3d(c * 2, c * 2, kernel_size=3, stride=1, pad=1), nn.ReLU(inplace=True))
        self.conv5 = _deconvbn_3d(c * 2, c * 2)     # + presqu / pre
        self.conv6 = _deconvbn_3d(c * 2, c)         # + x

    def forward(self, x, presqu, postsqu, skip=None):
        """Reference signature plus ``skip``: when given, ``out + skip`` (the caller's
        ``myadd_3d(out, cost0)``, stackhourglass.py:139-145) is fused into conv6."""
        out = self.conv1(x)                                    # 1/4 -> 1/8
        pre = self.conv2(out, residual=postsqu, relu=True)     # relu(conv2 (+ postsqu))
        out = self.conv4(self.conv3(pre))                      # 1/8 -> 1/16
        post = self.conv5(out, residual=presqu if presqu is not None else pre, relu=True)
        return self.conv6(post, residual=skip), pre, post      # 1/8 -> 1/4


class PSMNet(nn.Module):
    """``PSMNet(maxdisp)`` as the reference; ``align_corners`` (extra, default False) selects the
    interpolation convention of every upsampling on the path -- the SPP branches and the trilinear
    upsampling fused into the soft-argmin heads.  The reference calls ``F.upsample`` without the
    argument (stackhourglass.py:152-166, submodule.py:126-137): that resolves to
    ``align_corners=False`` on torch >= 0.4.1 (the oracle and the goldens) and meant ``True`` on the
    PyTorch 0.3 the reference was written for -- pass ``True`` to run a checkpoint the way its
    authors ran it ("version drift")."""

    def __init__(self, maxdisp=192, align_corners=False):
        super(PSMNet, self).__init__()
        self.name = "psmnet"
        self.maxdisp = maxdisp
        self.align_corners = bool(align_corners)
        self.count_levels = 1
        self.feature_extraction = feature_extraction(align_corners=align_corners)
        relu = lambda: nn.ReLU(inplace=True)  # noqa: E731
        self.dres0 = Chain3d(convbn_3d(64, 32, 3, 1, 1), relu(), convbn_3d(32, 32, 3, 1, 1), relu())
        self.dres1 = Chain3d(convbn_3d(32, 32, 3, 1, 1), relu(), convbn_3d(32, 32, 3, 1, 1))
        self.dres2 = hourglass(32)
        self.dres3 = hourglass(32)
        self.dres4 = hourglass(32)
        for i in (1, 2, 3):
            setattr(self, "classif%d" % i, Chain3d(
                convbn_3d(32, 32, 3, 1, 1), relu(),
                nn.Conv3d(32, 1, kernel_size=3, padding=1, stride=1, bias=False)))
        self._init_weights()

    def _init_weights(self):
        # stackhourglass.py:100-112 -- He-normal for Conv2d/Conv3d, BN to (1, 0)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv3d)):
                fan = m.out_channels
                for k in m.kernel_size:
                    fan *= k
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm3d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def features(self, left, right):
        if self.training:
            return self.feature_extraction(left), self.feature_extraction(right)
        # eval: BN uses running statistics, so both views can share one batch
        fe = self.feature_extraction
        if left.is_cuda and not torch.is_grad_enabled() and left.shape[1] <= 16:
            # concatenation + NHWC staging (3 -> 16 channels) in one launch
            both = fe(cv.stage_images_nhwc16(left, right), staged=True)
        else:
            both = self.feature_extraction(torch.cat([left, right], dim=0))
        self.__dict__["_both"] = both            # the two views as one NHWC tensor (virtual volume)
        return (cv.carry_amax(both[: left.shape[0]], both), cv.carry_amax(both[left.shape[0]:], both))

    def regularise(self, cost):
        """3-D trunk (stackhourglass.py:135-149): volume -> the three head costs.  ``cost``: the
        volume as an fp32 tensor or (eval) a never-materialised ``costvolume.VirtualVolume``."""
        cost0 = self.dres0(cost)
        cost0 = self.dres1(cost0, residual=cost0)
        out1, pre1, post1 = self.dres2(cost0, None, None, skip=cost0)
        out2, pre2, post2 = self.dres3(out1, pre1, post1, skip=cost0)
        out3, pre3, post3 = self.dres4(out2, pre1, post2, skip=cost0)    # pre1, as the reference (:144)
        cost1 = self.classif1(out1)
        cost2 = self.classif2(out2, residual=cost1)
        cost3 = self.classif3(out3, residual=cost2)
        return cost1, cost2, cost3

    def forward(self, left, right, mode="train"):
        with cv.amax_scope(left.device):
            return self._forward(left, right)

    def _forward(self, left, right):
        self.__dict__["_both"] = None
        refimg_fea, targetimg_fea = self.features(left, right)
        both = self.__dict__.pop("_both")
        if (both is not None and cv.get_option("fuse_volume") and not torch.is_grad_enabled() and
                cv.virtual_volume_ok(both.shape[1])):
            # the volume is never written: dres0's first convolution (the z-sliding kernel) stages
            # plane d from the towers' output (shift by d, mask x < d: stackhourglass.py:124-133)
            cost = cv.VirtualVolume(both, self.maxdisp // 4, True)
        else:
            cost = cv.concat_volume(refimg_fea, targetimg_fea, self.maxdisp // 4, mask_left=True)
        size = (self.maxdisp, left.shape[2], left.shape[3])
        cost1, cost2, cost3 = self.regularise(cost)
        pred1 = cv.soft_argmin(cost1, size, align_corners=self.align_corners)
        pred2 = cv.soft_argmin(cost2, size, align_corners=self.align_corners)
        pred3 = cv.soft_argmin(cost3, size, align_corners=self.align_corners)
        return [0, 0, 0], [pred3, pred2, pred1]
