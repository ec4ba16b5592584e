"""GPU parity: HIP corr1d / concat volume / soft-argmin (through the C ABI) against the
oracle on the same seeded inputs and against the golden fixtures generated from the
reference.  Tolerances are stated per test; integer-free fp32 throughout."""
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as OO
from tests.helpers import maxerr, seeded
from tests.test_oracle_golden import _softargmin_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv(hip_lib):
    from dsmnet_amd import costvolume
    return costvolume


def dev(t):
    return t.cuda()


# ------------------------------------------------------------------ corr1d --
def test_corr1d_golden_cases(cv, golden_ops):
    for case in golden_ops.meta["corr1d_cases"]:
        shp, seed, tag = case["shape"], case["seed"], case["tag"]
        fL = dev(seeded(seed, *shp)).requires_grad_(True)
        fR = dev(seeded(seed + 100, *shp)).requires_grad_(True)
        out = cv.corr1d(fL, fR, case["D"], case["s"], case["k"])
        cot = dev(seeded(seed + 200, *out.shape))
        gL, gR = torch.autograd.grad(out, (fL, fR), cot)
        # C<=128 fp32 dot products of N(0,1) data: sums ~ sqrt(C); 2e-4 abs is ~1e-5 rel
        golden_ops.compare(tag + ".out", out, 2e-4)
        golden_ops.compare(tag + ".dL", gL, 2e-4)
        golden_ops.compare(tag + ".dR", gR, 2e-4)


@pytest.mark.parametrize("shape,D,s,k", [
    ((1, 128, 96, 320), 41, 1, 1),      # BASELINE config #2: DispNetC 384x1280
    ((1, 128, 96, 320), 81, 1, 1),      # iResNet corr
    ((1, 64, 48, 160), 41, 2, 3),       # iResNet r_corr form (stride 2, box 3)
    ((2, 20, 7, 37), 11, 1, 1),         # ragged: W % 4 != 0, C not a chunk multiple
    ((1, 8, 3, 5), 9, 1, 1),            # D > W: planes past the width stay zero
    ((1, 8, 5, 30), 6, 3, 1),           # stride with no fast path
    ((1, 8, 5, 30), 4, 2, 5),           # wide box filter
])
def test_corr1d_vs_oracle(cv, shape, D, s, k):
    fL, fR = seeded(7, *shape), seeded(8, *shape)
    ref = OO.corr1d(fL.requires_grad_(True), fR.requires_grad_(True), D, s, k)
    cot = seeded(9, *ref.shape)
    rL, rR = torch.autograd.grad(ref, (fL, fR), cot)
    gl, gr = dev(fL.detach()).requires_grad_(True), dev(fR.detach()).requires_grad_(True)
    out = cv.corr1d(gl, gr, D, s, k)
    gL, gR = torch.autograd.grad(out, (gl, gr), dev(cot))
    tol = 3e-4
    assert maxerr(out, ref) <= tol
    assert maxerr(gL, rL) <= tol and maxerr(gR, rR) <= tol


def test_corr1d_linearity_full_size(cv):
    """Size-independent property at the BASELINE size: corr is bilinear."""
    shp = (1, 128, 96, 320)
    a, b, c = dev(seeded(1, *shp)), dev(seeded(2, *shp)), dev(seeded(3, *shp))
    lhs = cv.corr1d(a + 2 * c, b, 41)
    rhs = cv.corr1d(a, b, 41) + 2 * cv.corr1d(c, b, 41)
    assert maxerr(lhs, rhs) <= 2e-3
    assert cv.corr1d(a, b, 41)[0, 5, :, :5].abs().max().item() == 0.0


# ----------------------------------------------------------- concat volume --
@pytest.mark.parametrize("channels_last", [False, True])
def test_volume_golden_cases(cv, golden_ops, channels_last):
    for case in golden_ops.meta["volume_cases"]:
        shp, seed, tag = case["shape"], case["seed"], case["tag"]
        fL = dev(seeded(seed, *shp)).requires_grad_(True)
        fR = dev(seeded(seed + 100, *shp)).requires_grad_(True)
        vol = cv.concat_volume(fL, fR, case["D"], case["mask_left"], channels_last)
        assert vol.shape == (shp[0], 2 * shp[1], case["D"], shp[2], shp[3])
        cot = dev(seeded(seed + 200, *vol.shape))
        gL, gR = torch.autograd.grad(vol, (fL, fR), cot)
        golden_ops.compare(tag + ".vol", vol, 0.0)          # pure data movement: bit exact
        golden_ops.compare(tag + ".dL", gL, 1e-4)
        golden_ops.compare(tag + ".dR", gR, 1e-4)


@pytest.mark.parametrize("shape,D", [((1, 32, 24, 80), 48), ((2, 8, 5, 37), 12),
                                     ((1, 32, 3, 20), 40), ((1, 4, 2, 130), 3)])
@pytest.mark.parametrize("mask_left", [False, True])
@pytest.mark.parametrize("channels_last", [False, True])
def test_volume_vs_oracle(cv, shape, D, mask_left, channels_last):
    fL, fR = seeded(11, *shape), seeded(12, *shape)
    ref = OO.concat_volume(fL.requires_grad_(True), fR.requires_grad_(True), D, mask_left)
    cot = seeded(13, *ref.shape)
    rL, rR = torch.autograd.grad(ref, (fL, fR), cot)
    gl, gr = dev(fL.detach()).requires_grad_(True), dev(fR.detach()).requires_grad_(True)
    vol = cv.concat_volume(gl, gr, D, mask_left, channels_last)
    assert torch.equal(vol.cpu(), ref.detach())
    gL, gR = torch.autograd.grad(vol, (gl, gr), dev(cot))
    assert maxerr(gL, rL) <= 1e-4 and maxerr(gR, rR) <= 1e-4


@pytest.mark.parametrize("channels_last", [False, True])
def test_volume_full_size_properties(cv, channels_last):
    """PSMNet 384x1280 (D=192 -> 48 planes at 1/4): every plane d is the shifted,
    masked copy -- checked plane-wise on the device, no host copy of the 377 MB volume."""
    fL, fR = dev(seeded(21, 1, 32, 96, 320)), dev(seeded(22, 1, 32, 96, 320))
    vol = cv.concat_volume(fL, fR, 48, True, channels_last)
    assert vol.shape == (1, 64, 48, 96, 320)
    for d in (0, 1, 17, 47):
        assert torch.equal(vol[:, :32, d, :, d:], fL[..., d:])
        assert torch.equal(vol[:, 32:, d, :, d:], fR[..., :320 - d])
        if d:
            assert vol[:, :, d, :, :d].abs().max().item() == 0.0
    total = sum(float(fL[..., d:].double().sum() + fR[..., :320 - d].double().sum())
                for d in range(48))
    assert abs(float(vol.double().sum()) - total) <= 1e-6 * max(1.0, abs(total)) + 1e-3


# -------------------------------------------------------------- soft-argmin --
def test_softargmin_golden_cases(cv, golden_ops):
    for case in golden_ops.meta["softargmin_cases"]:
        c = dev(_softargmin_inputs(case)).requires_grad_(True)
        if case["form"] == "psm":
            out = cv.soft_argmin(c, tuple(case["out_size"]))
        else:
            out = cv.soft_argmin(c, None, negate=True).unsqueeze(1)
        cot = dev(seeded(case["seed"] + 200, *out.shape))
        (g,) = torch.autograd.grad(out, c, cot)
        # disparity in [0, 48): 1e-3 is the north-star bound; measured error is ~1e-5
        golden_ops.compare(case["tag"] + ".disp", out, 1e-3)
        golden_ops.compare(case["tag"] + ".dcost", g, 2e-3, 1e-3)


@pytest.mark.parametrize("cshape,osize,negate,align", [
    ((1, 1, 12, 8, 16), (48, 32, 64), False, False),
    ((2, 1, 12, 9, 17), (48, 35, 66), False, False),     # non-integer scale factors
    ((1, 1, 12, 8, 16), (48, 32, 64), False, True),      # PyTorch-0.3 era align_corners
    ((1, 1, 5, 6, 70), (7, 6, 300), False, False),       # D < 8: fewer lane segments
    ((1, 1, 25, 7, 11), (100, 27, 45), False, False),    # D >= 96: four disparity segments in the tiled adjoint
    ((2, 1, 24, 5, 9), (96, 20, 36), False, True),
    ((1, 1, 48, 12, 20), None, True, False),             # GCNet form
    ((2, 1, 192, 5, 33), None, True, False),
])
def test_softargmin_vs_oracle(cv, cshape, osize, negate, align):
    c = seeded(31, *cshape, scale=2.0)
    ref = OO.soft_argmin(c.requires_grad_(True), osize, negate, align)
    cot = seeded(32, *ref.shape)
    (rg,) = torch.autograd.grad(ref, c, cot)
    g = dev(c.detach()).requires_grad_(True)
    out = cv.soft_argmin(g, osize, negate, align)
    (gg,) = torch.autograd.grad(out, g, dev(cot))
    assert maxerr(out, ref) <= 1e-3
    assert maxerr(gg, rg) <= 1e-3 * max(1.0, rg.abs().max().item())


def test_softargmin_full_size_properties(cv):
    """PSMNet head at 384x1280: output bounded by [0, D-1]; a constant cost gives the
    mean disparity; shifting the cost by a constant changes nothing."""
    c = dev(seeded(41, 1, 1, 48, 96, 320, scale=2.0))
    out = cv.soft_argmin(c, (192, 384, 1280))
    assert out.shape == (1, 384, 1280)
    assert out.min().item() >= 0.0 and out.max().item() <= 191.0
    assert maxerr(cv.soft_argmin(c + 3.0, (192, 384, 1280)), out) <= 2e-3
    flat = cv.soft_argmin(torch.zeros_like(c), (192, 384, 1280))
    assert maxerr(flat, torch.full_like(flat, 95.5)) <= 1e-3


def test_soft_argmin_up4_isolated_peaks_force_the_exact_fallback(hip_lib):
    """The x4 head sums in ONE pass under a running maximum of the coarse samples; an isolated peak
    of height >> 88 leaves every fine value more than expf's range below that maximum, and the
    pixel is then redone with the exact two-pass form.  Costs that force the branch on most pixels
    (peaks of height 300..3000, one-hot softmax), plus moderate ones that never take it."""
    from dsmnet_amd import costvolume as cv
    from oracle import ops as OO
    g = torch.Generator().manual_seed(5)
    for scale in (1.0, 300.0, 3000.0):
        cost = torch.randn(1, 1, 12, 9, 17, generator=g)
        peaks = torch.randint(0, 12, (9, 17), generator=g)
        cost[0, 0].scatter_(0, peaks.unsqueeze(0), scale * (1.0 + torch.rand(1, 9, 17, generator=g)))
        want = OO.soft_argmin(cost, (48, 36, 68))
        got = cv.soft_argmin(cost.cuda(), (48, 36, 68))
        assert torch.isfinite(got).all()
        assert (got.cpu() - want).abs().max().item() <= 1e-3, scale


@pytest.mark.parametrize("relu,shapes", [
    (True, ((2, 24, 13, 22), (1, 6, 11), (16, 12, 21))),     # ragged: every operand a different size
    (True, ((1, 32, 16, 40), (1, 8, 20), (64, 16, 40))),
    (False, ((1, 8, 10, 12), None, (4, 10, 12))),            # bare deconv, no prediction map
    (True, ((1, 8, 10, 12), (2, 5, 6), None)),               # no skip
])
def test_decoder_level_equals_the_stock_ops(hip_lib, relu, shapes):
    """``costvolume.decoder_level`` (csrc/decoder.hip) = myCat2d(deconv(x), upsample(pr), skip) of
    models/dispnetcorr.py:89-132: copies bit-identical, the upsampled channel to 1e-6."""
    import torch.nn as nn
    from dsmnet_amd import costvolume as cv
    (B, Cu, Hu, Wu), prs, sks = shapes
    torch.manual_seed(3)
    conv = nn.ConvTranspose2d(12, Cu, 4, 2, 1, bias=True)
    deconv = (nn.Sequential(conv, nn.ReLU(inplace=True)) if relu else conv).cuda()
    x = torch.randn(B, 12, Hu // 2, Wu // 2, device="cuda")
    pr = None if prs is None else torch.randn(B, *prs, device="cuda")
    skip = None if sks is None else torch.randn(B, *sks, device="cuda")
    with torch.no_grad():
        got = cv.decoder_level(deconv, x, pr, skip)
    with torch.enable_grad():                                   # the stock branch
        want = cv.decoder_level(deconv, x.clone().requires_grad_(True), pr, skip).detach()
    assert got.shape == want.shape
    Cu_, Cp = Cu, (0 if pr is None else pr.shape[1])
    assert torch.equal(got[:, :Cu_], want[:, :Cu_])
    assert torch.equal(got[:, Cu_ + Cp:], want[:, Cu_ + Cp:])
    if Cp:
        assert (got[:, Cu_:Cu_ + Cp] - want[:, Cu_:Cu_ + Cp]).abs().max().item() <= 1e-6


@pytest.mark.parametrize("shape,D", [((1, 32, 6, 40), 12), ((2, 32, 5, 21), 9), ((1, 64, 3, 70), 48)])
def test_right_referenced_volume_vs_oracle(hip_lib, shape, D):
    """``concat_volume_right`` (gcnet_LR's xR, models/gcnet.py:155-164): data movement, bit-exact."""
    from dsmnet_amd import costvolume as cv
    from oracle import ops as OO
    fL, fR = seeded(81, *shape), seeded(82, *shape)
    got = cv.concat_volume_right(fL.cuda(), fR.cuda(), D)
    assert torch.equal(got.cpu().contiguous(), OO.concat_volume_right(fL, fR, D))


def test_right_referenced_volume_golden_cases(hip_lib):
    """The same against the fixtures written from the reference's own lines (models/gcnet.py:155-164
    executed by tests/golden/make_goldens.py), D > W included: bit-exact."""
    from tests.conftest import Golden
    from dsmnet_amd import costvolume as cv
    g = Golden("lr")
    assert any(c["D"] > c["shape"][3] for c in g.meta["volume_lr_cases"])
    for case in g.meta["volume_lr_cases"]:
        fL, fR = seeded(case["seed"], *case["shape"]), seeded(case["seed"] + 100, *case["shape"])
        got = cv.concat_volume_right(fL.cuda(), fR.cuda(), case["D"])
        g.compare(case["tag"] + ".xR", got, 0.0)


def test_gcnet_lr_left_output_equals_gcnet(hip_lib, golden_e2e):
    """gcnet_LR(imL, imR)[0] is gcnet's output; its right output equals the left output of the
    mirrored problem only up to the trunk's (non-symmetric) weights, so it is checked against the
    oracle trunk on the oracle's right-referenced volume."""
    from oracle import models as OM, ops as OO
    from tests.golden.make_goldens import images
    from tests.helpers import golden_state, maxerr
    from dsmnet_amd.models.gcnet import gcnet_LR
    sd, cfg = golden_state(golden_e2e, "gcnet")
    imL, imR = images(cfg["image_seed"], *cfg["hw"])
    m = gcnet_LR(192)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        oL, oR = m(imL.cuda(), imR.cuda())
        n = OM.Net(sd)
        fL, fR = OM.gcnet_features(n, imL), OM.gcnet_features(n, imR)
        wantL = OO.soft_argmin(OM.gcnet_trunk(n, OO.concat_volume(fL, fR, 96, False)), None, negate=True).unsqueeze(1)
        wantR = OO.soft_argmin(OM.gcnet_trunk(n, OO.concat_volume_right(fL, fR, 96)), None, negate=True).unsqueeze(1)
    h, w = imL.shape[-2:]
    assert maxerr(oL, wantL[:, :, :h, :w]) <= 1e-3
    assert maxerr(oR, wantR[:, :, :h, :w]) <= 1e-3


@pytest.mark.parametrize("shape,pair", [((2, 3, 17, 37), True), ((1, 3, 16, 40), False), ((1, 16, 5, 9), True)])
def test_stage_images_nhwc16(cv, shape, pair):
    """The towers' one-launch input staging == torch.cat of the views + zero channels, in NHWC memory
    (bit-identical: a copy)."""
    left, right = dev(seeded(51, *shape)), (dev(seeded(52, *shape)) if pair else None)
    out = cv.stage_images_nhwc16(left, right)
    ref = left if right is None else torch.cat([left, right], 0)
    assert out.shape == (ref.shape[0], 16) + tuple(shape[2:])
    assert out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out[:, : shape[1]], ref)
    assert float(out[:, shape[1]:].abs().max()) == 0.0 if shape[1] < 16 else True
