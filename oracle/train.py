"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's
supervised training objective and bookkeeping, SURVEY.md section 8f-3.

  loss_supervised        losses/loss.py:326-338   (diff1_dx / diff1_dy: :36-44)
  weight_adjust_levels   losses/loss.py:379-392
  losses_pyramid0        losses/loss.py:407-422
  lr_adjust              stereo.py:95-101
  accuracy (D1, EPE)     stereo.py:103-113

Pinned against the reference's own lines executed from the file text
(tests/golden/make_goldens.py, part "train"; fixture tests/golden/golden_train.npz).
Written in float64 numpy-style arithmetic on purpose: a second derivation, not a copy of the
product module dsmnet_amd/train.py.
"""
import numpy as np
import torch
import torch.nn.functional as F


def loss_supervised(disp_gt, disp, flag_smooth=False, factor=1.0):
    """Mean |gt - pred| over gt > 0, plus 0.1 x the clamped first-difference magnitude of the
    prediction over the same mask.  No valid pixel -> the integer 0 (loss.py:328-329)."""
    gt, d = disp_gt.double(), disp.double()
    mask = gt > 0
    n = int(mask.sum())
    if n == 0:
        return 0
    loss = ((gt - d).abs() * mask).sum() / n
    if flag_smooth:
        dx = torch.zeros_like(d)
        dy = torch.zeros_like(d)
        dx[..., :, :-1] = d[..., :, 1:] - d[..., :, :-1]        # zero in the last column
        dy[..., :-1, :] = d[..., 1:, :] - d[..., :-1, :]        # zero in the last row
        smooth = ((dx.abs() + dy.abs()) / factor).clamp(0, 1)
        loss = loss + 0.1 * (smooth * mask).sum() / n
    return loss


def weight_adjust_levels(count_levels, maxepoch, epoch):
    """Coarse-to-fine schedule of the pyramid weights (loss.py:379-392)."""
    w = [0.01] * count_levels
    if count_levels == 1 or epoch >= maxepoch:
        w[0] = 1
        return w
    x = (1 - epoch / float(maxepoch)) * (count_levels - 1)
    idx = int(x)
    frac = x - idx
    w[idx] = 1 - frac
    if idx < count_levels - 1:
        w[idx + 1] = frac
    return w


def losses_pyramid0(weight_levels, disp_gt, disps, scale_disps, flag_smooth=False):
    """Weighted sum over the outputs; level > 0 outputs are upsampled bilinearly by 2**level
    and cropped to the ground truth (loss.py:407-422)."""
    h, w = disp_gt.shape[-2:]
    total = 0
    for d, level in zip(disps, scale_disps):
        wt = weight_levels[level]
        if wt <= 0:
            continue
        if level > 0:
            d = F.interpolate(d.double(), scale_factor=2 ** level, mode="bilinear",
                              align_corners=False)[:, :, :h, :w]
        total = total + loss_supervised(disp_gt, d, flag_smooth, factor=1) * wt
    return total


def lr_adjust(lr0, epoch0, stride, epoch):
    """Learning rate in force at ``epoch``: ``None`` = untouched before ``epoch0``, then halved
    every ``stride`` epochs starting with one halving AT ``epoch0`` (stereo.py:95-101)."""
    if epoch < epoch0:
        return None
    n = ((epoch - epoch0) // stride) + 1
    return lr0 * (0.5 ** n)


def accuracy(disp, disp_gt):
    """(D1 in percent, EPE) over gt > 0; a pixel is good when its error is <= 3 px OR <= 5 % of
    the ground truth (stereo.py:103-113 -- the KITTI D1 uses AND; the reference's OR is kept)."""
    gt = np.asarray(disp_gt, dtype=np.float64)
    d = np.asarray(disp, dtype=np.float64)
    m = gt > 0
    diff = np.abs(gt - d)[m]
    epe = diff.mean()
    good = (diff <= 3) | (diff / gt[m] <= 0.05)
    return 100 - 100.0 * good.sum() / m.sum(), epe
