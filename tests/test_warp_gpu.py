"""Disparity warp + reconstruction error (csrc/warp.hip) against the
oracle's restatement of utils/imwrap.py:37-72 / models/iresnet.py:169-170."""
import pytest
import torch

from oracle import models as OM
from tests.helpers import seeded, maxerr

pytestmark = pytest.mark.gpu


def _delt(seed):
    torch.manual_seed(seed)
    return float(1e-4 * (torch.rand(1)[0] + 0.1))          # the draw imwrap_BCHW makes (:70)


@pytest.mark.parametrize("shape,scale", [((1, 32, 48, 160), 3.0), ((2, 5, 9, 33), 1.0),
                                         ((1, 3, 16, 20), 40.0), ((1, 8, 2, 2), 0.5)])
def test_warp_and_abs_error_match_oracle(hip_lib, shape, scale):
    """Smooth and wild disparities (|d| up to ~100 px on a 20-px-wide map: every tap out of
    bounds), ragged sizes, the 2x2 minimum."""
    from dsmnet_amd import costvolume as cv
    B, C, H, W = shape
    L, R = seeded(1, *shape), seeded(2, *shape)
    disp = seeded(3, B, 1, H, W) * scale
    torch.manual_seed(11)
    want = OM.imwarp(R, disp)
    got = cv.warp_abs_error(None, R.cuda(), disp.cuda(), _delt(11))
    assert maxerr(got.cpu(), want) <= 2e-5
    err = cv.warp_abs_error(L.cuda(), R.cuda(), disp.cuda(), _delt(11))
    assert maxerr(err.cpu(), (L - want).abs()) <= 2e-5


def test_warp_source_larger_than_disparity_map(hip_lib):
    """im_src (H0,W0) larger than disp (H,W): the grid covers the top-left corner (imwrap.py:51-56)."""
    from dsmnet_amd import costvolume as cv
    R, disp = seeded(5, 1, 4, 24, 40), seeded(6, 1, 1, 20, 31) * 2
    torch.manual_seed(3)
    want = OM.imwarp(R, disp)
    got = cv.warp_abs_error(None, R.cuda(), disp.cuda(), _delt(3))
    assert got.shape == (1, 4, 20, 31) and maxerr(got.cpu(), want) <= 2e-5


def test_warp_argument_checks(hip_lib):
    from dsmnet_amd import costvolume as cv
    R = torch.zeros(1, 4, 8, 8, device="cuda")
    with pytest.raises(ValueError):
        cv.warp_abs_error(None, R, torch.zeros(1, 2, 8, 8, device="cuda"), 1e-5)
    with pytest.raises(ValueError):
        cv.warp_abs_error(torch.zeros(1, 3, 8, 8, device="cuda"), R, torch.zeros(1, 1, 8, 8, device="cuda"), 1e-5)
    with pytest.raises(RuntimeError):
        cv.warp_abs_error(None, R.cpu(), torch.zeros(1, 1, 8, 8), 1e-5)
