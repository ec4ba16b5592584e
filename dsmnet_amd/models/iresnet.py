"""iResNet on the MI355X cost-volume path: same names, attribute tree and return convention
as models/iresnet.py.  Both correlations -- `corr` (D=81) and `r_corr` (kernel 3, stride 2,
D=41, the only strided/box-filtered use in the reference) -- are the HIP Corr1d; the 2-D
encoder/decoder/refinement layers are stock torch (outside the hot path)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import costvolume as cv
from .util_conv import Corr1d, conv2d_bn, deconv2d_bn, net_init
from .util_fun import myCat2d


def _layer(fn, *a, **k):
    return fn(*a, flag_bias=True, bn=False, activefun=nn.ReLU(inplace=True), **k)


def imwrap_BCHW(im_src, disp):
    """Warp ``im_src`` by ``disp`` with bilinear ``grid_sample`` (utils/imwrap.py:37-72, default
    arguments).  Like the reference it adds a tiny random epsilon drawn from the global
    generator (``:70``) before sampling."""
    bn, _, h0, w0 = im_src.shape
    _, c, h, w = disp.shape
    assert c == 1 and min(h, w, h0, w0) > 1
    x1 = -1.0 + (w - 1) * 2.0 / (w0 - 1)
    y1 = -1.0 + (h - 1) * 2.0 / (h0 - 1)
    gx = torch.linspace(-1.0, x1, w).view(1, 1, w).expand(bn, h, w).to(im_src)
    gy = torch.linspace(-1.0, y1, h).view(1, h, 1).expand(bn, h, w).to(im_src)
    grid = torch.stack([gx - disp.squeeze(1) * 2.0 / (w0 - 1), gy], dim=3)
    delt = 1e-4 * (torch.rand(1)[0] + 0.1)
    return F.grid_sample(im_src + delt.to(im_src), grid, mode="bilinear", padding_mode="zeros",
                         align_corners=False)


_ENCODER = [("conv3", 81 + 64, 256, 2), ("conv3_1", 256, 256, 1), ("conv4", 256, 512, 2),
            ("conv4_1", 512, 512, 1), ("conv5", 512, 512, 2), ("conv5_1", 512, 512, 1),
            ("conv6", 512, 1024, 2), ("conv6_1", 1024, 1024, 1)]
_DECODER = {5: (1024, 512, 512), 4: (512, 256, 512), 3: (256, 128, 256), 2: (128, 64, 128),
            1: (64, 32, 64), 0: (32, 32, 32)}


class iresnet(nn.Module):
    def __init__(self, maxdisparity=192):
        super(iresnet, self).__init__()
        self.name = "iresnet"
        self.D = maxdisparity
        self.delt = 1e-6
        self.count_levels = 7
        self.upsample = nn.Upsample(scale_factor=2, mode="bilinear")
        # shared multi-scale stem
        self.conv1 = _layer(conv2d_bn, 3, 64, kernel_size=7, stride=2)
        self.conv2 = _layer(conv2d_bn, 64, 128, kernel_size=5, stride=2)
        self.deconv1_s = _layer(deconv2d_bn, 64, 32, kernel_size=4, stride=2)
        self.deconv2_s = _layer(deconv2d_bn, 128, 32, kernel_size=8, stride=4)
        self.conv_de1_de2 = _layer(conv2d_bn, 64, 32, kernel_size=1, stride=1)
        # initial disparity network
        self.corr = Corr1d(kernel_size=1, stride=1, D=81, simfun=None)
        self.redir = _layer(conv2d_bn, 128, 64, kernel_size=1, stride=1)
        for name, cin, cout, s in _ENCODER:
            setattr(self, name, _layer(conv2d_bn, cin, cout, kernel_size=3, stride=s))
        self.pr6 = nn.Conv2d(1024, 1, kernel_size=3, stride=1, padding=1)
        for lvl in (5, 4, 3, 2, 1, 0):
            cin, width, skip = _DECODER[lvl]
            setattr(self, "deconv%d" % lvl, _layer(deconv2d_bn, cin, width, kernel_size=4, stride=2))
            setattr(self, "iconv%d" % lvl, _layer(conv2d_bn, width + 1 + skip, width,
                                                  kernel_size=3, stride=1))
            setattr(self, "pr%d" % lvl, nn.Conv2d(width, 1, kernel_size=3, stride=1, padding=1))
        # refinement network
        self.r_conv0 = _layer(conv2d_bn, 65, 32, kernel_size=3, stride=1)
        self.r_conv1 = _layer(conv2d_bn, 32, 64, kernel_size=3, stride=2)
        self.c_conv1 = _layer(conv2d_bn, 64, 64, kernel_size=3, stride=1)
        self.r_corr = Corr1d(kernel_size=3, stride=2, D=41, simfun=None)
        self.r_conv1_1 = _layer(conv2d_bn, 105, 64, kernel_size=3, stride=1)
        self.r_conv2 = _layer(conv2d_bn, 64, 128, kernel_size=3, stride=2)
        self.r_conv2_1 = _layer(conv2d_bn, 128, 128, kernel_size=3, stride=1)
        self.r_res2 = nn.Conv2d(128, 1, kernel_size=3, stride=1, padding=1)
        self.r_deconv1 = _layer(deconv2d_bn, 128, 64, kernel_size=4, stride=2)
        self.r_iconv1 = _layer(conv2d_bn, 129, 64, kernel_size=3, stride=1)
        self.r_res1 = nn.Conv2d(64, 1, kernel_size=3, stride=1, padding=1)
        self.r_deconv0 = _layer(deconv2d_bn, 64, 32, kernel_size=4, stride=2)
        self.r_iconv0 = _layer(conv2d_bn, 65, 32, kernel_size=3, stride=1)
        self.r_res0 = nn.Conv2d(32, 1, kernel_size=3, stride=1, padding=1)
        net_init(self)
        for head in [getattr(self, "pr%d" % i) for i in range(1, 7)] + \
                    [self.r_res2, self.r_res1, self.r_res0]:      # pr0 is not scaled (:83)
            head.weight.data = head.weight.data * 0.1

    def _stem(self, conv1, conv2, hw):
        d1 = self.deconv1_s(conv1)[:, :, : hw[0], : hw[1]]
        return self.conv_de1_de2(myCat2d(d1, self.deconv2_s(conv2)))

    def forward(self, imL, imR, mode="train", iter=1):
        if imL.shape != imR.shape:
            raise ValueError("iresnet: imL and imR must have the same shape")   # :87
        maxD = max(self.D, imL.shape[-1])
        hw = imL.shape[-2:]
        conv1L, conv1R = self.conv1(imL), self.conv1(imR)
        conv2L, conv2R = self.conv2(conv1L), self.conv2(conv1R)
        stemL, stemR = self._stem(conv1L, conv2L, hw), self._stem(conv1R, conv2R, hw)
        x = torch.cat([self.corr(conv2L, conv2R), self.redir(conv2L)], dim=1)
        skips = {2: conv2L, 1: conv1L, 0: stemL}
        for name, _, _, _ in _ENCODER:
            x = getattr(self, name)(x)
            if name.endswith("_1"):
                skips[int(name[4])] = x             # conv3_1, conv4_1, conv5_1
        pr = self.pr6(x)
        out, out_scale, keep = [pr], [6], {}
        for lvl in (5, 4, 3, 2, 1, 0):
            x = getattr(self, "iconv%d" % lvl)(
                cv.decoder_level(getattr(self, "deconv%d" % lvl), x, pr, skips[lvl]))
            pr = getattr(self, "pr%d" % lvl)(x)
            keep[lvl] = pr
            out.insert(0, pr)
            out_scale.insert(0, lvl)
        r_pr2, r_pr1, r_pr0 = keep[2], keep[1], keep[0]
        for _ in range(iter):
            if stemL.is_cuda and not torch.is_grad_enabled():
                # one HIP pass; the epsilon is drawn exactly as the reference draws it (imwrap.py:70)
                delt = float(1e-4 * (torch.rand(1)[0] + 0.1))
                err = cv.warp_abs_error(stemL, stemR, -r_pr0, delt)
            else:
                err = torch.abs(stemL - imwrap_BCHW(stemR, -r_pr0))
            r_conv0 = self.r_conv0(myCat2d(err, r_pr0, stemL))
            r_conv1 = self.r_conv1(r_conv0)
            r_corr = self.r_corr(self.c_conv1(conv1L), self.c_conv1(conv1R))
            r_conv1_1 = self.r_conv1_1(myCat2d(r_conv1, r_corr))
            r_conv2_1 = self.r_conv2_1(self.r_conv2(r_conv1_1))
            r_res2 = self.r_res2(r_conv2_1)
            r_pr2 = r_pr2 + r_res2
            out.insert(0, r_pr2); out_scale.insert(0, 2)
            r_iconv1 = self.r_iconv1(cv.decoder_level(self.r_deconv1, r_conv2_1, r_res2, r_conv1_1))
            r_res1 = self.r_res1(r_iconv1)
            r_pr1 = r_pr1 + r_res1
            out.insert(0, r_pr1); out_scale.insert(0, 1)
            r_iconv0 = self.r_iconv0(cv.decoder_level(self.r_deconv0, r_iconv1, r_res1, r_conv0))
            r_pr0 = r_pr0 + self.r_res0(r_iconv0)
            out.insert(0, r_pr0); out_scale.insert(0, 0)
        if mode == "test":
            out[-1] = out[-1].clamp(self.delt, maxD)
        return out_scale, out
