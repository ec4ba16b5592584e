"""GPU parity: gradients of the 2-D tower layers in training (SURVEY.md section 8f-1 under autograd):
``costvolume.Conv2dFunction`` (forward and bwd-data on the MFMA convolution kernels, bwd-weight on
``dsm_conv2d_wgrad``) and the fused batch-statistics BN block against torch's CPU autograd, then the
whole PSMNet ``feature_extraction`` in train mode against the stock torch layers of the same module
tree on the CPU in fp64, with the stock fp32 layers on the GPU (``DSM_TRAIN_2D=stock`` semantics) as
the yardstick for fp32 backpropagation noise.

Tolerances: gradients are fp32 sums over up to 1e5 pixels with atomics (order varies): 1e-3
relative to the largest gradient entry, as for the 3-D layers (tests/test_conv3d_bwd_gpu.py)."""
import copy

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import maxerr, seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv(hip_lib):
    from dsmnet_amd import costvolume
    return costvolume


@pytest.fixture(autouse=True)
def fused_towers():
    """These tests run every layer on the own-kernel tower path (DSM_TRAIN_2D=fused), whatever its size."""
    from dsmnet_amd import blocks2d
    old = (blocks2d._FUSED_TRAIN_2D, blocks2d._FUSED_TRAIN_2D_BN, blocks2d._TRAIN_2D_MIN_PIXELS)
    blocks2d._FUSED_TRAIN_2D, blocks2d._FUSED_TRAIN_2D_BN, blocks2d._TRAIN_2D_MIN_PIXELS = True, True, 0
    yield
    blocks2d._FUSED_TRAIN_2D, blocks2d._FUSED_TRAIN_2D_BN, blocks2d._TRAIN_2D_MIN_PIXELS = old


@pytest.mark.parametrize("cin,cout,k,stride,dil,shape", [
    (32, 32, 3, 1, 1, (2, 13, 37)), (64, 64, 3, 1, 1, (1, 17, 70)), (64, 128, 3, 1, 1, (1, 9, 33)),
    (128, 128, 3, 1, 2, (2, 11, 35)), (32, 64, 3, 2, 1, (1, 14, 38)), (3, 32, 3, 2, 1, (1, 16, 40)),
    (320, 128, 3, 1, 1, (1, 8, 33)), (128, 32, 1, 1, 1, (2, 5, 9)), (64, 128, 1, 1, 1, (1, 9, 33)),
    (32, 64, 1, 2, 1, (1, 14, 38)), (32, 64, 3, 2, 1, (1, 15, 37)),
])
def test_conv2d_function_gradients(cv, cin, cout, k, stride, dil, shape):
    B, H, W = shape
    x = seeded(1, B, cin, H, W).requires_grad_(cin != 3)
    w = seeded(2, cout, cin, k, k, scale=0.1).requires_grad_(True)
    pad = dil * (k // 2)
    ref = F.conv2d(x.double(), w.double(), None, stride, pad, dil)
    cot = seeded(4, *ref.shape)
    wrt = [t for t in (x, w) if t.requires_grad]
    grads = torch.autograd.grad(ref, wrt, cot.double())
    xg = x.detach().cuda().requires_grad_(cin != 3)
    wg = w.detach().cuda().requires_grad_(True)
    y = cv.conv2d(xg, wg, stride, dil)
    assert maxerr(y, ref.float()) <= 2e-4 * max(1.0, ref.abs().max().item())
    got = torch.autograd.grad(y, [t for t in (xg, wg) if t.requires_grad], cot.cuda())
    for g, r in zip(got, grads):
        tol = 1e-3 * max(1.0, r.abs().max().item())
        assert g.shape == r.shape
        assert maxerr(g, r.float()) <= tol, "%.3e > %.3e" % (maxerr(g, r.float()), tol)


@pytest.mark.parametrize("relu,with_res", [(True, False), (False, True), (False, False)])
def test_convbn2d_train_block(cv, relu, with_res):
    """ConvBN2d in train mode (fused conv + BN + skip + ReLU autograd nodes) == the stock
    Sequential(Conv2d, BatchNorm2d) (+ skip) (+ ReLU) on the CPU in fp64, running statistics
    included."""
    from dsmnet_amd.models.psmnet.submodule import convbn
    torch.manual_seed(3)
    blk = convbn(64, 64, 3, 1, 1, 1)
    with torch.no_grad():
        blk[1].weight.uniform_(0.5, 1.5)
        blk[1].bias.uniform_(-0.3, 0.3)
    ref = copy.deepcopy(blk).double()
    x = seeded(5, 2, 64, 12, 36)
    res = seeded(6, 2, 64, 12, 36) if with_res else None
    xr = x.double().requires_grad_(True)
    yr = torch.nn.Sequential.forward(ref, xr)
    if with_res:
        yr = yr + res.double()
    if relu:
        yr = F.relu(yr)
    cot = seeded(7, *yr.shape)
    gref = torch.autograd.grad(yr, [xr, ref[0].weight, ref[1].weight, ref[1].bias], cot.double())
    blk = blk.cuda().train()
    xg = x.cuda().requires_grad_(True)
    y = blk(xg, residual=None if res is None else res.cuda(), relu=relu)
    assert y.grad_fn is not None and "BnAddRelu3d" in type(y.grad_fn).__name__ or "Squeeze" in type(y.grad_fn).__name__
    assert maxerr(y, yr.float()) <= 2e-4
    got = torch.autograd.grad(y, [xg, blk[0].weight, blk[1].weight, blk[1].bias], cot.cuda())
    for g, r in zip(got, gref):
        tol = 1e-3 * max(1.0, r.abs().max().item())
        assert maxerr(g, r.float()) <= tol, "%.3e > %.3e" % (maxerr(g, r.float()), tol)
    assert maxerr(blk[1].running_mean, ref[1].running_mean.float()) <= 1e-5
    assert maxerr(blk[1].running_var, ref[1].running_var.float()) <= 1e-4
    assert int(blk[1].num_batches_tracked) == 1


def test_feature_extraction_train_step():
    """PSMNet's whole 2-D tower in train mode: output and parameter gradients against the same
    module tree running stock torch layers on the CPU in fp64."""
    from dsmnet_amd import blocks2d
    from dsmnet_amd.models.psmnet.submodule import feature_extraction
    torch.manual_seed(11)
    net = feature_extraction()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.6, 1.4)
                m.bias.uniform_(-0.2, 0.2)
    ref = copy.deepcopy(net).double().train()
    img = seeded(12, 1, 3, 256, 256)          # the SPP head pools 64 x 64 at 1/4 resolution
    yr = ref(img.double())
    cot = seeded(13, *yr.shape)
    names = [n for n, _ in ref.named_parameters()]
    gref = torch.autograd.grad(yr, [p for _, p in ref.named_parameters()], cot.double())
    # the stock fp32 layers on the same GPU: the yardstick for what fp32 backpropagation through
    # 55 batch-normalised layers (and ReLU units that sit within rounding distance of zero) gives
    stock = copy.deepcopy(net).cuda().train()
    blocks2d._FUSED_TRAIN_2D = False
    try:
        ys = stock(img.cuda())
        gstock = torch.autograd.grad(ys, [p for _, p in stock.named_parameters()], cot.cuda())
    finally:
        blocks2d._FUSED_TRAIN_2D = True
    net = net.cuda().train()
    from dsmnet_amd import costvolume as cv
    t = cv.LaunchTimer()
    cv.set_timer(t)
    try:
        y = net(img.cuda())
        got = torch.autograd.grad(y, [p for _, p in net.named_parameters()], cot.cuda())
        torch.cuda.synchronize()
    finally:
        cv.set_timer(None)
    launched = t.summary()
    assert any("conv2d_wgrad_kernel" in k for k in launched), sorted(launched)
    assert any("bn3d_train_fwd_kernels" in k for k in launched), sorted(launched)
    assert maxerr(y, yr.float()) <= 1e-3 * max(1.0, yr.abs().max().item())
    worst = worst_stock = 0.0
    rows = []
    for n, g, gs, r in zip(names, got, gstock, gref):
        scale = max(1e-2, r.abs().max().item())
        e, es = maxerr(g, r.float()) / scale, maxerr(gs, r.float()) / scale
        worst, worst_stock = max(worst, e), max(worst_stock, es)
        rows.append((e, es, n))
    print("worst relative gradient error %.2e (stock fp32 layers: %.2e)" % (worst, worst_stock))
    for e, es, n in sorted(rows, reverse=True)[:12]:
        print("   %-32s %.2e   stock %.2e" % (n, e, es))
    # a ReLU input within rounding distance of zero takes either side in either implementation and
    # moves that layer's gradients by up to a few per cent (both columns show it, on different
    # layers): the bound is the stock layers' own worst case, and the medians must agree closely
    assert worst <= 3.0 * worst_stock + 1e-3
    med = sorted(r[0] for r in rows)[len(rows) // 2]
    med_stock = sorted(r[1] for r in rows)[len(rows) // 2]
    print("median %.2e (stock %.2e)" % (med, med_stock))
    assert med <= 3.0 * med_stock + 1e-4


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("hw,size", [((1, 2), (64, 128)), ((8, 16), (64, 128)), ((3, 5), (17, 33))])
def test_spp_upsample_as_matrix_products(hw, size, align):
    """The SPP branches' bilinear upsampling under autograd (two matrix products) equals
    F.interpolate in value and gradient (models/psmnet/submodule.py:118-128)."""
    from dsmnet_amd.models.psmnet.submodule import feature_extraction
    net = feature_extraction(align_corners=align)
    y = seeded(21, 2, 32, *hw).cuda().requires_grad_(True)
    out = net._upsample(y, size)
    ref = F.interpolate(y, size=size, mode="bilinear", align_corners=align)
    assert out.shape == ref.shape and maxerr(out, ref) <= 2e-6
    cot = seeded(22, *ref.shape).cuda()
    (g,) = torch.autograd.grad(out, y, cot, retain_graph=True)
    (gr,) = torch.autograd.grad(ref, y, cot)
    assert maxerr(g, gr) <= 1e-4 * max(1.0, gr.abs().max().item())
