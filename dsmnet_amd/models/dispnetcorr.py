"""DispNetC on the MI355X cost-volume path: same names, attribute tree and return convention
as models/dispnetcorr.py; `self.corr` is the HIP Corr1d (D=41), everything else is the
reference's 2-D encoder/decoder; each decoder level's bias + ReLU + upsampling + myCat2d is one
HIP launch in eval mode (`costvolume.decoder_level`, csrc/decoder.hip)."""
import torch
import torch.nn as nn

from .. import costvolume as cv
from .util_conv import Corr1d, conv2d_bn, deconv2d_bn, net_init
from .util_fun import myCat2d

# (name, Cin, Cout, kernel, stride) of the encoder after the correlation
_ENCODER = [("conv3a", 64 + 41, 256, 5, 2), ("conv3b", 256, 256, 3, 1), ("conv4a", 256, 512, 3, 2),
            ("conv4b", 512, 512, 3, 1), ("conv5a", 512, 512, 3, 2), ("conv5b", 512, 512, 3, 1),
            ("conv6a", 512, 1024, 3, 2), ("conv6b", 1024, 1024, 3, 1)]
# decoder level -> (deconv Cin, width, skip channels)
_DECODER = {5: (1024, 512, 512), 4: (512, 256, 512), 3: (256, 128, 256), 2: (128, 64, 128),
            1: (64, 32, 64)}


def _layer(fn, *a, **k):
    return fn(*a, flag_bias=True, bn=False, activefun=nn.ReLU(inplace=True), **k)


class dispnetcorr(nn.Module):
    def __init__(self, maxdisparity=192):
        super(dispnetcorr, self).__init__()
        self.name = "dispnetcorr"
        self.D = maxdisparity
        self.delt = 1e-6
        self.count_levels = 7
        self.upsample = nn.Upsample(scale_factor=2, mode="bilinear")
        self.conv1 = _layer(conv2d_bn, 3, 64, kernel_size=7, stride=2)
        self.conv2 = _layer(conv2d_bn, 64, 128, kernel_size=5, stride=2)
        self.corr = Corr1d(kernel_size=1, stride=1, D=41, simfun=None)
        self.redir = _layer(conv2d_bn, 128, 64, kernel_size=1, stride=1)
        for name, cin, cout, k, s in _ENCODER:
            setattr(self, name, _layer(conv2d_bn, cin, cout, kernel_size=k, stride=s))
        self.pr6 = nn.Conv2d(1024, 1, kernel_size=3, stride=1, padding=1)
        for lvl in (5, 4, 3, 2, 1):
            cin, width, skip = _DECODER[lvl]
            setattr(self, "deconv%d" % lvl, _layer(deconv2d_bn, cin, width, kernel_size=4, stride=2))
            setattr(self, "iconv%d" % lvl, _layer(conv2d_bn, width + 1 + skip, width,
                                                  kernel_size=3, stride=1))
            setattr(self, "pr%d" % lvl, nn.Conv2d(width, 1, kernel_size=3, stride=1, padding=1))
        net_init(self)
        for lvl in range(1, 7):                     # prediction heads start small (:63-64)
            head = getattr(self, "pr%d" % lvl)
            head.weight.data = head.weight.data * 0.1

    def forward(self, imL, imR, mode="train"):
        if imL.shape != imR.shape:
            raise ValueError("dispnetcorr: imL and imR must have the same shape")   # :67
        maxD = max(self.D, imL.shape[-1])
        conv1L, conv1R = self.conv1(imL), self.conv1(imR)
        conv2L, conv2R = self.conv2(conv1L), self.conv2(conv1R)
        x = torch.cat([self.corr(conv2L, conv2R), self.redir(conv2L)], dim=1)
        skips = {2: conv2L, 1: conv1L}
        for name, _, _, _, _ in _ENCODER:
            x = getattr(self, name)(x)
            if name.endswith("b"):
                skips[int(name[4])] = x             # conv3b, conv4b, conv5b
        pr = self.pr6(x)
        out, out_scale = [pr], [6]
        for lvl in (5, 4, 3, 2, 1):
            # myCat2d(deconv(x), upsample(pr), skip): dispnetcorr.py:94-95 and the levels below
            x = getattr(self, "iconv%d" % lvl)(
                cv.decoder_level(getattr(self, "deconv%d" % lvl), x, pr, skips[lvl]))
            pr = getattr(self, "pr%d" % lvl)(x)
            out.insert(0, pr)
            out_scale.insert(0, lvl)
        out.insert(0, self.upsample(pr)[:, :, : imL.shape[-2], : imL.shape[-1]])
        out_scale.insert(0, 0)
        if mode == "test":
            out[-1] = out[-1].clamp(self.delt, maxD)   # the reference clamps out[-1] (:132)
        return out_scale, out
