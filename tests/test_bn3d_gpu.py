"""GPU parity of the fused train-mode BatchNorm3d + cropped skip add + ReLU (csrc/bn3d.hip) against
the stock torch ops it replaces (nn.BatchNorm3d in train mode, myadd_3d, F.relu: models/psmnet/
submodule.py:16-19, stackhourglass.py:10-20, models/util_conv.py:150-179), forward and backward."""
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import maxerr, seeded

pytestmark = pytest.mark.gpu


def _reference(y, gamma, beta, res, rm, rv, relu, momentum, eps):
    out = F.batch_norm(y, rm, rv, gamma, beta, True, momentum, eps)
    if relu == 2:
        out = out.relu()
    if res is not None:
        d, h, w = (min(a, b) for a, b in zip(out.shape[2:], res.shape[2:]))
        out = out[:, :, :d, :h, :w] + res[:, :, :d, :h, :w]
    if relu == 1:
        out = out.relu()
    return out


@pytest.mark.parametrize("C,yshape,rshape,relu", [
    (32, (2, 5, 9, 33), None, 1),
    (32, (1, 6, 13, 40), (6, 13, 40), 1),
    (64, (1, 4, 8, 20), (3, 7, 19), 1),          # the residual is the smaller one: out is its size
    (64, (1, 3, 7, 19), (4, 8, 20), 2),          # y is the smaller one; GCNet's ReLU-before-add
    (32, (1, 5, 9, 33), (5, 9, 33), 0),
])
def test_bn_add_relu3d_vs_torch(hip_lib, C, yshape, rshape, relu):
    from dsmnet_amd import costvolume as cv
    B = yshape[0]
    y = (seeded(1, B, C, *yshape[1:]) * 1.7 + 0.3).double().requires_grad_(True)
    gamma = (seeded(2, C).abs() + 0.5).double().requires_grad_(True)
    beta = seeded(3, C).double().requires_grad_(True)
    res = seeded(4, B, C, *rshape).double().requires_grad_(True) if rshape else None
    rm, rv = seeded(5, C).double(), seeded(6, C).abs().double() + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    want = _reference(y, gamma, beta, res, rm, rv, relu, 0.1, 1e-5)
    cot = seeded(7, *want.shape).double()
    grads = torch.autograd.grad(want, [t for t in (y, gamma, beta, res) if t is not None], cot)

    yg = y.detach().float().cuda().requires_grad_(True)
    gg, bg = gamma.detach().float().cuda().requires_grad_(True), beta.detach().float().cuda().requires_grad_(True)
    rg = res.detach().float().cuda().requires_grad_(True) if res is not None else None
    rmg, rvg = rm0.float().cuda(), rv0.float().cuda()
    out = cv.bn_add_relu3d(yg, gg, bg, rg, rmg, rvg, relu, 0.1, 1e-5)
    assert tuple(out.shape) == tuple(want.shape)
    assert maxerr(out, want) <= 2e-5 * max(1.0, want.abs().max().item())
    assert maxerr(rmg, rm) <= 1e-5 and maxerr(rvg, rv) <= 1e-5          # running statistics as nn.BatchNorm3d
    got = torch.autograd.grad(out, [t for t in (yg, gg, bg, rg) if t is not None], cot.float().cuda())
    for name, g, r in zip(("dy", "dgamma", "dbeta", "dres"), got, grads):
        assert g.shape == r.shape, name
        tol = 5e-5 * max(1.0, r.abs().max().item())
        assert maxerr(g, r) <= tol, "%s: %.3e > %.3e" % (name, maxerr(g, r), tol)
