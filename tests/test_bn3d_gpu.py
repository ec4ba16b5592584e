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


@pytest.mark.parametrize("rshape", [(5, 9, 33), (4, 8, 30)])
def test_bn_add_relu3d_residual_without_gradient(hip_lib, rshape):
    """A skip tensor that needs no gradient (the first block after a frozen stem; a cropped one
    included): forward as above, dy / dgamma / dbeta unchanged, and no gradient buffer for it."""
    from dsmnet_amd import costvolume as cv
    C, yshape = 32, (1, 5, 9, 33)
    y = (seeded(1, 1, C, *yshape[1:]) * 1.7 + 0.3).double().requires_grad_(True)
    gamma = (seeded(2, C).abs() + 0.5).double().requires_grad_(True)
    beta = seeded(3, C).double().requires_grad_(True)
    res = seeded(4, 1, C, *rshape).double()
    want = _reference(y, gamma, beta, res, None, None, 1, 0.1, 1e-5)
    cot = seeded(7, *want.shape).double()
    grads = torch.autograd.grad(want, [y, gamma, beta], cot)
    yg = y.detach().float().cuda().requires_grad_(True)
    gg, bg = gamma.detach().float().cuda().requires_grad_(True), beta.detach().float().cuda().requires_grad_(True)
    rg = res.float().cuda()                                   # requires_grad = False
    out = cv.bn_add_relu3d(yg, gg, bg, rg, None, None, 1, 0.1, 1e-5)
    assert maxerr(out, want) <= 2e-5 * max(1.0, want.abs().max().item())
    got = torch.autograd.grad(out, [yg, gg, bg], cot.float().cuda())
    assert rg.grad is None
    for name, g, r in zip(("dy", "dgamma", "dbeta"), got, grads):
        assert maxerr(g, r) <= 5e-5 * max(1.0, r.abs().max().item()), name


@pytest.mark.parametrize("res_grad", [True, False])
def test_bn_add_relu2d_vs_torch(hip_lib, res_grad):
    """The 2-D entry (the towers' BasicBlock tail, models/psmnet/submodule.py:24-46) is the 3-D
    kernel on the (B, C, 1, H, W) view."""
    from dsmnet_amd import costvolume as cv
    B, C, H, W = 2, 64, 11, 37
    y = (seeded(11, B, C, H, W) * 1.3 - 0.2).double().requires_grad_(True)
    gamma = (seeded(12, C).abs() + 0.5).double().requires_grad_(True)
    beta = seeded(13, C).double().requires_grad_(True)
    res = seeded(14, B, C, H, W).double().requires_grad_(res_grad)
    rm, rv = seeded(15, C).double(), seeded(16, C).abs().double() + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    want = (F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5) + res).relu()
    cot = seeded(17, *want.shape).double()
    wrt = [y, gamma, beta] + ([res] if res_grad else [])
    grads = torch.autograd.grad(want, wrt, cot)
    dev = lambda t, g=True: t.detach().float().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(g) \
        if t.dim() == 4 else t.detach().float().cuda().requires_grad_(g)
    yg, gg, bg, rg = dev(y), dev(gamma), dev(beta), dev(res, res_grad)
    rmg, rvg = rm0.float().cuda(), rv0.float().cuda()
    out = cv.bn_add_relu2d(yg, gg, bg, rg, rmg, rvg, 1, 0.1, 1e-5)
    assert maxerr(out, want) <= 2e-5 * max(1.0, want.abs().max().item())
    assert maxerr(rmg, rm) <= 1e-5 and maxerr(rvg, rv) <= 1e-5
    got = torch.autograd.grad(out, [yg, gg, bg] + ([rg] if res_grad else []), cot.float().cuda())
    for name, g, r in zip(("dy", "dgamma", "dbeta", "dres"), got, grads):
        assert maxerr(g, r) <= 5e-5 * max(1.0, r.abs().max().item()), name


def test_a_train_step_refolds_only_the_layers_it_touched(hip_lib):
    """The BN kernels write the running statistics through raw pointers; the host bumps exactly those
    tensors' version counters (blocks3d._bump_running_stats), so the next eval forward folds the
    NEW statistics -- equal to a freshly built block with the same state."""
    from dsmnet_amd import blocks3d
    import torch.nn as nn
    torch.manual_seed(3)
    blk = blocks3d.ConvBN3d(nn.Conv3d(32, 32, 3, 1, 1, bias=False), nn.BatchNorm3d(32)).cuda()
    other = blocks3d.ConvBN3d(nn.Conv3d(32, 32, 3, 1, 1, bias=False), nn.BatchNorm3d(32)).cuda()
    x = seeded(5, 1, 32, 4, 9, 20).cuda().contiguous(memory_format=torch.channels_last_3d)
    blk.eval(); other.eval()
    with torch.no_grad():
        before = blk(x).clone(); other(x)
    other_key = other._folded.key
    blk.train()
    blk(x)                                                     # running statistics move
    blk.eval()
    with torch.no_grad():
        after = blk(x)
        fresh = blocks3d.ConvBN3d(nn.Conv3d(32, 32, 3, 1, 1, bias=False), nn.BatchNorm3d(32)).cuda().eval()
        fresh.load_state_dict(blk.state_dict())
        want = fresh(x)
        other(x)
    assert maxerr(after, before) > 1e-3                        # the fold did change
    assert torch.equal(after, want)
    assert other._folded.key == other_key                      # the untouched layer kept its fold
