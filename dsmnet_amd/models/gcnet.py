"""GCNet on the MI355X cost-volume path: same names, attribute tree and return convention
as models/gcnet.py of the reference; the volume build (gcnet.py:130-135) and the soft-argmin
(:104-111), inline in the reference, are the HIP ops here."""
import torch
import torch.nn as nn

from .. import costvolume as cv
from .util_conv import conv2d_bn, conv3d_bn, conv_res, deconv3d_bn, net_init

flag_bias_t = True
flag_bn = True


def _act():
    return nn.ReLU(inplace=True)


class feature2d(nn.Module):
    """2-D tower at 1/2 resolution (models/gcnet.py:14-29).  Eval mode on the GPU: the eight
    residual blocks and the closing biased 3x3 convolution -- 17 of the 18 convolutions -- run on
    the MFMA kernel with folded BN (blocks2d); the 5x5 stride-2 stem (3 -> 32) stays a stock layer."""

    def __init__(self, num_F=32):
        super(feature2d, self).__init__()
        self.F = num_F
        self.conv1 = conv2d_bn(3, 32, kernel_size=5, stride=2, flag_bias=flag_bias_t, bn=flag_bn,
                               activefun=_act())
        self.block1 = conv_res(32, 32, blocks=8, stride=1)
        self.conv2 = nn.Conv2d(32, 32, kernel_size=3, stride=1, padding=1)

    def forward(self, x):
        x = self.block1(self.conv1(x))
        if x.is_cuda and not self.training and not torch.is_grad_enabled():
            from ..blocks2d import _Folded2d, run_conv2d
            if not hasattr(self, "_fold"):
                self._fold = _Folded2d()
            return run_conv2d(self._fold, self.conv2, None, x)    # NHWC: what the virtual volume stages from
        return self.conv2(x)


# (name, Cin multiplier, Cout multiplier, stride) of the 14 convolutions, in definition order
_CONVS = [("l19", 2, 1, 1), ("l20", 1, 1, 1), ("l21", 2, 2, 2), ("l22", 2, 2, 1), ("l23", 2, 2, 1),
          ("l24", 2, 2, 2), ("l25", 2, 2, 1), ("l26", 2, 2, 1), ("l27", 2, 2, 2), ("l28", 2, 2, 1),
          ("l29", 2, 2, 1), ("l30", 2, 4, 2), ("l31", 4, 4, 1), ("l32", 4, 4, 1)]
_DECONVS = [("l33", 4, 2), ("l34", 2, 2), ("l35", 2, 2), ("l36", 2, 1)]


class feature3d(nn.Module):
    """19-layer 3-D encoder-decoder + soft-argmin (gcnet.py:32-111), every layer one launch."""

    def __init__(self, num_F=32):
        super(feature3d, self).__init__()
        F_ = self.F = num_F
        for name, ci, co, s in _CONVS:
            setattr(self, name, conv3d_bn(F_ * ci, F_ * co, kernel_size=3, stride=s,
                                          flag_bias=flag_bias_t, bn=flag_bn, activefun=_act()))
        for name, ci, co in _DECONVS:
            setattr(self, name, deconv3d_bn(F_ * ci, F_ * co, kernel_size=3, stride=2,
                                            flag_bias=flag_bias_t, bn=flag_bn, activefun=_act()))
        self.l37 = deconv3d_bn(F_, 1, kernel_size=3, stride=2, bn=False, activefun=None)
        self.softmax = nn.Softmax2d()          # parameter-free; kept for attribute parity

    def cost(self, x, virtual=None):
        """x18 -> x37, the (B,1,2D,2h,2w) cost.  Skip additions are fused into the
        transposed convolutions (relu(bn(deconv)) + skip, cropped: gcnet.py:78-96).
        ``virtual``: the same volume as a never-materialised ``costvolume.VirtualVolume`` (eval):
        l19 (64 -> 32) then stages it from the towers' output on the z-sliding kernel; the fp32
        volume ``x`` still feeds the stride-2 l21."""
        x21 = self.l21(x)
        x24 = self.l24(x21)
        x27 = self.l27(x24)
        x32 = self.l32(self.l31(self.l30(x27)))
        x33 = self.l33(x32, residual=self.l29(self.l28(x27)))
        x34 = self.l34(x33, residual=self.l26(self.l25(x24)))
        x35 = self.l35(x34, residual=self.l23(self.l22(x21)))
        x20 = self.l20(self.l19(x if virtual is None else virtual))
        x36 = self.l36(x35, residual=x20)
        return self.l37(x36)

    def forward(self, x, mode="train", virtual=None):
        # `mode="test"` only frees intermediates early in the reference; nothing to do here
        return cv.soft_argmin(self.cost(x, virtual), None, negate=True).unsqueeze(1)


class gcnet(nn.Module):
    def __init__(self, maxdisparity=192):
        super(gcnet, self).__init__()
        self.name = "gcnet"
        self.D = maxdisparity // 2
        self.count_levels = 1
        self.layer2d = feature2d(32)
        self.layer3d = feature3d(32)
        net_init(self)

    def features(self, imL, imR):
        if self.training:
            return self.layer2d(imL), self.layer2d(imR)
        both = self.layer2d(torch.cat([imL, imR], dim=0))     # eval: BN uses running stats
        self.__dict__["_both"] = both
        return cv.carry_amax(both[: imL.shape[0]], both), cv.carry_amax(both[imL.shape[0]:], both)

    def forward(self, imL, imR, mode="train"):
        with cv.amax_scope(imL.device):
            return self._forward(imL, imR, mode)

    def _forward(self, imL, imR, mode):
        if imL.shape != imR.shape:
            raise ValueError("gcnet: imL and imR must have the same shape")   # gcnet.py:127
        self.__dict__["_both"] = None
        fL, fR = self.features(imL, imR)
        both = self.__dict__.pop("_both")
        xL = cv.concat_volume(fL, fR, self.D, mask_left=False)
        virtual = None
        if (both is not None and both.is_cuda and cv.get_option("fuse_volume") and
                not torch.is_grad_enabled() and cv.virtual_volume_ok(both.shape[1])):
            virtual = cv.VirtualVolume(both, self.D, False)
        oL = self.layer3d(xL, mode, virtual)[:, :, : imL.shape[-2], : imL.shape[-1]]
        return [0], [oL]


class gcnet_LR(nn.Module):
    """The two-sided GCNet of models/gcnet.py:139-167: left- and right-referenced volumes through
    the same trunk; ``forward(imL, imR) -> (oL, oR)``.  Not reachable from the reference's
    ``model_create_by_name`` (nor from this one's); eval / inference only for the right side."""

    def __init__(self, maxdisparity=192):
        super(gcnet_LR, self).__init__()
        self.name = "gcnet"
        self.D = maxdisparity // 2
        self.layer2d = feature2d(32)
        self.layer3d = feature3d(32)
        net_init(self)

    def forward(self, imL, imR):
        if imL.shape != imR.shape:
            raise ValueError("gcnet_LR: imL and imR must have the same shape")   # gcnet.py:151
        fL, fR = self.layer2d(imL), self.layer2d(imR)
        xL = cv.concat_volume(fL, fR, self.D, mask_left=False)
        xR = cv.concat_volume_right(fL, fR, self.D)
        crop = (slice(None), slice(None), slice(0, imL.shape[-2]), slice(0, imL.shape[-1]))
        return self.layer3d(xL)[crop], self.layer3d(xR)[crop]
