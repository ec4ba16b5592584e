"""GPU parity for the S3 path (csrc/conv_s3.hip): the pre-split fp32 storage format, the cost
volume written (or only implied) in it, and the z-sliding bf16x3 convolution -- against the
oracle's CPU restatements (oracle/ops.py) and torch's CPU convolution."""
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as OO
from tests.helpers import maxerr, seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv(hip_lib):
    from dsmnet_amd import costvolume
    return costvolume


def test_s3_is_a_lossless_encoding_of_fp32(cv):
    """v = hi + mid + lo exactly, for normal fp32 values of any magnitude and sign."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 3, 5, 37, generator=g)
    x = x * torch.exp(torch.randn(x.shape, generator=g) * 12.0)          # 1e-20 .. 1e+20
    x[0, :, 0, 0, :4] = torch.tensor([0.0, -0.0, 1.0, -3.0])
    x[1, 5, 1, 2, 7] = 3.3895313892515355e38                             # near the largest fp32
    s3 = cv.s3_from_tensor(x.cuda())
    assert s3.shape == tuple(x.shape) and s3.buf.numel() == x.numel() * 6
    back = s3.to_tensor()
    assert torch.equal(back.cpu().contiguous(), x)


@pytest.mark.parametrize("mask_left", [False, True])
@pytest.mark.parametrize("shape,D", [((1, 32, 16, 40), 48), ((2, 64, 5, 21), 9)])
def test_volume_s3_decodes_to_the_reference_volume(cv, shape, D, mask_left):
    fL, fR = seeded(11, *shape), seeded(12, *shape)
    want = OO.concat_volume(fL, fR, D, mask_left=mask_left)
    vol = cv.concat_volume_s3(fL.cuda(), fR.cuda(), D, mask_left)
    assert torch.equal(vol.to_tensor().cpu().contiguous(), want)          # data movement: bit-exact


def _reference_conv(x, w, scale, shift, res, relu):
    y = F.conv3d(x.double(), w.double(), padding=1)
    y = y * scale.double().view(1, -1, 1, 1, 1) + shift.double().view(1, -1, 1, 1, 1)
    if relu == 2:
        y = y.relu()
    if res is not None:
        d, h, w_ = (min(a, b) for a, b in zip(y.shape[2:], res.shape[2:]))
        y = y[:, :, :d, :h, :w_] + res.double()[:, :, :d, :h, :w_]
    if relu == 1:
        y = y.relu()
    return y.float()


@pytest.mark.parametrize("shape,cin,relu,res_shape,grid", [
    ((1, 6, 12, 40), 32, 1, None, 0),
    ((1, 6, 12, 40), 32, 0, None, 3),             # ranges that cross column borders mid-z
    ((2, 5, 9, 33), 64, 2, (5, 9, 33), 7),        # batch 2, ragged tiles, ReLU before the skip add
    ((1, 13, 17, 70), 32, 1, (12, 16, 69), 5),    # cropped skip: the output is the common corner
    ((1, 1, 8, 32), 32, 1, None, 0),              # one plane: both z-neighbours outside the volume
    ((1, 48, 24, 64), 64, 1, (48, 24, 64), 0),    # the dres0[0] shape class, 256 workgroups
])
def test_conv3d_s3_vs_cpu_fp64(cv, shape, cin, relu, res_shape, grid):
    B, D, H, W = shape
    x = seeded(21, B, cin, D, H, W)
    w = seeded(22, 32, cin, 3, 3, 3, scale=(2.0 / (27 * cin)) ** 0.5)
    scale, shift = seeded(23, 32).abs() + 0.5, seeded(24, 32)
    res = seeded(25, B, 32, *res_shape) if res_shape else None
    want = _reference_conv(x, w, scale, shift, res, relu)
    xs = cv.s3_from_tensor(x.cuda())
    packed = cv.pack_conv3d_s3_weight(w.cuda())
    y, ys3 = cv.conv3d_s3_block(xs, packed, scale.cuda(), shift.cuda(),
                                None if res is None else res.cuda(), relu=relu, out="both", grid=grid)
    assert tuple(y.shape) == tuple(want.shape)
    assert maxerr(y, want) <= 2e-5 * max(1.0, want.abs().max().item())
    assert torch.equal(ys3.to_tensor(), y)                                # the S3 output IS the fp32 output
    only = cv.conv3d_s3_block(xs, packed, scale.cuda(), shift.cuda(),
                              None if res is None else res.cuda(), relu=relu, out="s3", grid=grid)
    assert torch.equal(only.buf, ys3.buf)


def test_conv3d_s3_agrees_with_the_fp32_input_mfma_kernel(cv):
    """Same layer on the exact-fp32 MFMA kernel (conv3d.hip) and on the z-sliding bf16x3 kernel."""
    x = seeded(31, 1, 32, 7, 19, 45)
    w = seeded(32, 32, 32, 3, 3, 3, scale=0.05)
    old = cv.conv3d_block(x.cuda(), cv.pack_conv3d_weight(w.cuda(), False), 32, relu=1)
    new = cv.conv3d_s3_block(cv.s3_from_tensor(x.cuda()), cv.pack_conv3d_s3_weight(w.cuda()), relu=1)
    assert maxerr(old, new) <= 2e-5 * max(1.0, old.abs().max().item())


@pytest.mark.parametrize("mask_left", [False, True])
def test_virtual_volume_equals_the_materialised_one(cv, mask_left):
    """dres0[0] staged straight from the split feature maps (the volume never written) gives the
    same bits as the same kernel reading the materialised S3 volume."""
    fL, fR = seeded(41, 2, 32, 11, 53), seeded(42, 2, 32, 11, 53)
    D = 14
    w = seeded(43, 32, 64, 3, 3, 3, scale=0.04)
    packed = cv.pack_conv3d_s3_weight(w.cuda())
    real = cv.concat_volume_s3(fL.cuda(), fR.cuda(), D, mask_left)
    virt = cv.concat_volume_s3(fL.cuda(), fR.cuda(), D, mask_left, materialise=False)
    a = cv.conv3d_s3_block(real, packed, relu=1, grid=9)
    b = cv.conv3d_s3_block(virt, packed, relu=1, grid=9)
    assert torch.equal(a, b)
    want = F.conv3d(OO.concat_volume(fL, fR, D, mask_left=mask_left).double(), w.double(), padding=1).relu().float()
    assert maxerr(a, want) <= 2e-5 * max(1.0, want.abs().max().item())
    with pytest.raises(RuntimeError):
        virt.to_tensor()


@pytest.mark.parametrize("cin,cout,stride,transposed,shape", [
    (64, 32, 2, True, (1, 3, 5, 33)),        # hourglass conv6: the producer of the classifiers' input
    (64, 64, 2, True, (1, 3, 6, 20)),
    (32, 32, 1, False, (1, 5, 9, 37)),
    (32, 64, 2, False, (1, 6, 12, 40)),
])
def test_bf16x3_kernels_write_the_same_result_as_s3(cv, cin, cout, stride, transposed, shape):
    B, D, H, W = shape
    x = seeded(51, B, cin, D, H, W)
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = seeded(52, *wshape, scale=0.05)
    packed = cv.pack_conv3d_weight(w.cuda(), transposed)
    res_shape = tuple(2 * v - 1 for v in (D, H, W)) if transposed else None
    res = seeded(53, B, cout, *res_shape).cuda() if res_shape else None
    y, ys3 = cv.conv3d_block(x.cuda(), packed, cout, residual=res, stride=stride,
                             transposed=transposed, relu=1, out="both")
    ref = cv.conv3d_block(x.cuda(), packed, cout, residual=res, stride=stride,
                          transposed=transposed, relu=1)
    assert torch.equal(y, ref)
    assert torch.equal(ys3.to_tensor(), y)


def test_psmnet_paths_agree(cv, golden_e2e):
    """PSMNet eval forward (bf16x3) three ways -- virtual volume + z-sliding S3 layers, S3 layers on a
    materialised fp32 volume, and the r01 path (no S3 at all): each within 1e-3 px of the reference
    golden."""
    from tests.golden.make_goldens import images
    from tests.helpers import golden_state
    from dsmnet_amd.models import model_create_by_name
    sd, cfg = golden_state(golden_e2e, "psmnet")
    imL, imR = images(cfg["image_seed"], *cfg["hw"])
    m = model_create_by_name("psmnet", 192)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    outs = {}
    for name, s3, fuse in (("default", True, True), ("s3-on-fp32-volume", True, False), ("r01", False, False)):
        o0 = cv.set_option("conv_precision", "bf16x3")
        o1, o2 = cv.set_option("s3", s3), cv.set_option("fuse_volume", fuse)
        try:
            with torch.no_grad():
                outs[name] = m(imL.cuda(), imR.cuda())[1]
        finally:
            cv.set_option("s3", o1), cv.set_option("fuse_volume", o2), cv.set_option("conv_precision", o0)
        for pname, p in zip(("pred3", "pred2", "pred1"), outs[name]):
            golden_e2e.compare("e2e.psmnet." + pname, p, 1e-3)
    assert maxerr(outs["default"][0], outs["r01"][0]) <= 1e-3
