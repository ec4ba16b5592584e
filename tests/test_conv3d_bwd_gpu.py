"""GPU parity: gradients of the 3-D convolution blocks (bwd-data on the forward kernels with
re-packed weights, bwd-weight on the voxel-reduction MFMA kernel) against torch's CPU autograd,
and a PSMNet trunk training step against the oracle's autograd (BASELINE config #5 in fp32).

Tolerances: gradients are sums over up to 1e5 voxels of O(1) products in fp32 with atomics
(order varies): 1e-3 relative to the largest gradient entry."""
import pytest
import torch
import torch.nn.functional as F

from oracle import models as OM
from oracle import ops as OO
from tests.golden.make_goldens import randomise_bn
from tests.helpers import maxerr, seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv(hip_lib):
    from dsmnet_amd import costvolume
    return costvolume


@pytest.mark.parametrize("cin,cout,stride,transposed,shape,bias", [
    (32, 32, 1, False, (1, 6, 12, 40), False), (64, 32, 1, False, (1, 5, 9, 33), True),
    (64, 64, 1, False, (2, 3, 7, 35), False),
    (32, 64, 2, False, (1, 6, 12, 40), False), (64, 64, 2, False, (1, 5, 9, 37), True),
    (64, 64, 2, True, (1, 3, 6, 20), False), (64, 32, 2, True, (1, 3, 5, 33), True),
    (32, 1, 1, False, (1, 5, 9, 37), False),
])
def test_conv3d_function_gradients(cv, cin, cout, stride, transposed, shape, bias):
    B, D, H, W = shape
    x = seeded(1, B, cin, D, H, W).requires_grad_(True)
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = seeded(2, *wshape, scale=0.1).requires_grad_(True)
    b = seeded(3, cout, scale=0.1).requires_grad_(True) if bias else None
    if transposed:
        ref = F.conv_transpose3d(x, w, b, stride=2, padding=1, output_padding=1)
    else:
        ref = F.conv3d(x, w, b, stride=stride, padding=1)
    cot = seeded(4, *ref.shape)
    grads = torch.autograd.grad(ref, [t for t in (x, w, b) if t is not None], cot)
    xg = x.detach().cuda().requires_grad_(True)
    wg = w.detach().cuda().requires_grad_(True)
    bg = b.detach().cuda().requires_grad_(True) if bias else None
    y = cv.conv3d(xg, wg, bg, stride, transposed)
    assert maxerr(y, ref) <= 2e-4
    got = torch.autograd.grad(y, [t for t in (xg, wg, bg) if t is not None], cot.cuda())
    for name, g, r in zip(("dx", "dw", "db"), got, grads):
        tol = 1e-3 * max(1.0, r.abs().max().item())
        assert g.shape == r.shape, name
        assert maxerr(g, r) <= tol, "%s: %.3e > %.3e" % (name, maxerr(g, r), tol)


def _trunk_step_errors(seed):
    """Relative max-abs error of every checked gradient for one seeded input, plus the loss and
    running-statistics checks."""
    from dsmnet_amd import costvolume as cv
    from dsmnet_amd.models import model_create_by_name
    sd = randomise_bn(OM.init_state("psmnet", 0), 41)
    OM.apply_head_scale("psmnet", sd, 0.05)
    fl, fr = seeded(seed, 1, 32, 16, 40), seeded(seed + 1, 1, 32, 16, 40)
    size = (32, 64, 160)
    # oracle: leaves that require grad
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k
               else v.clone()) for k, v in sd.items()}
    ofl, ofr = fl.clone().requires_grad_(True), fr.clone().requires_grad_(True)
    n = OM.Net(osd, training=True)
    costs = OM.psmnet_trunk(n, OO.concat_volume(ofl, ofr, 8, True))
    oloss = sum(OO.soft_argmin(c, size).mean() for c in costs)
    keys = ["dres0.0.0.weight", "dres0.0.1.weight", "dres1.2.0.weight", "dres2.conv1.0.0.weight",
            "dres2.conv5.0.weight", "dres3.conv6.0.weight", "dres4.conv2.1.bias",
            "classif1.2.weight", "classif3.0.0.weight"]
    ogr = torch.autograd.grad(oloss, [osd[k] for k in keys] + [ofl, ofr])
    # product
    m = model_create_by_name("psmnet", 192)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    gfl, gfr = fl.cuda().requires_grad_(True), fr.cuda().requires_grad_(True)
    gc = m.regularise(cv.concat_volume(gfl, gfr, 8, True))
    loss = sum(cv.soft_argmin(c, size).mean() for c in gc)
    assert abs(loss.item() - oloss.item()) <= 1e-3 * max(1.0, abs(oloss.item()))
    for a, b in zip(gc, costs):                                   # train-mode forward: tight
        assert maxerr(a, b) <= 2e-5 * b.abs().max().item()
    params = dict(m.named_parameters())
    ggr = torch.autograd.grad(loss, [params[k] for k in keys] + [gfl, gfr])
    # running statistics were updated as nn.BatchNorm3d does
    assert maxerr(m.dres0[0][1].running_mean, osd["dres0.0.1.running_mean"]) <= 1e-4
    return {k: maxerr(g, r) / max(r.abs().max().item(), 1e-6)
            for k, g, r in zip(keys + ["fL", "fR"], ggr, ogr)}


def test_psmnet_trunk_training_step_vs_oracle(hip_lib):
    """Train-mode forward + backward through the whole 3-D trunk and the three fused heads:
    parameter and input gradients against the oracle's CPU autograd.

    The forward agrees to ~3e-6 on every input.  The backward of a ReLU network is discontinuous:
    an activation within rounding distance of zero can take the other side in the two
    implementations, and one flipped unit in a BatchNorm layer that sees 80 voxels per channel
    moves that layer's gradients by ~1 %.  That happens for about one seed in three, with the
    fp32-input MFMA as with the bf16x3 kernels (tests/tools/grad_check.py: seeds 71/81/91/101).  So:
    every seed must stay within the bound a few flips can explain, and at least one must be
    flip-free, where all gradients agree to 2e-3 (measured: 3e-6).  A wrong backward kernel fails
    both."""
    clean = False
    for seed in (81, 71, 101):
        errs = _trunk_step_errors(seed)
        assert max(errs.values()) <= 0.2, errs
        if max(errs.values()) <= 2e-3:
            clean = True
            break
    assert clean, errs


def test_psmnet_full_training_step(hip_lib):
    """BASELINE config #5 in miniature (fp32, 1 GPU): PSMNet in train mode, 256x512, D=192:
    forward with batch-statistics BN, smooth-L1 loss on the three heads, backward through the
    HIP trunk and the stock-torch towers, one SGD step; every parameter receives a finite
    gradient and the loss on the same pair goes down."""
    from dsmnet_amd import sharding
    from dsmnet_amd.models import model_create_by_name
    torch.manual_seed(0)
    m = model_create_by_name("psmnet", 192).cuda().train()
    for i in (1, 2, 3):                                   # sane logit scale for a random init
        getattr(m, "classif%d" % i)[2].weight.data.mul_(1e-3)
    g = torch.Generator().manual_seed(5)
    left = torch.rand(1, 3, 256, 512, generator=g).cuda()
    right = torch.roll(left, -6, dims=3)
    target = torch.full((1, 256, 512), 6.0, device="cuda")
    opt = torch.optim.SGD(m.parameters(), lr=1e-3)

    def step():
        opt.zero_grad()
        _, preds = m(left, right)
        loss = sum(w * F.smooth_l1_loss(p, target) for w, p in zip((1.0, 0.7, 0.5), preds))
        loss.backward()
        return loss

    l0 = step()
    missing = [n for n, p in m.named_parameters() if p.grad is None]
    assert not missing, missing[:5]
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    assert sharding.allreduce_gradients(m.parameters(), world=1) == 0    # single process: no-op
    opt.step()
    l1 = step()
    assert torch.isfinite(l0) and torch.isfinite(l1)
    assert l1.item() < l0.item(), (l0.item(), l1.item())
