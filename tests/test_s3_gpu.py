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


@pytest.mark.parametrize("shape,cin,relu,res_shape,grid,virtual", [
    ((1, 6, 12, 40), 32, 1, None, 0, False),
    ((2, 5, 9, 33), 64, 2, (5, 9, 33), 7, False),       # ragged tiles (4-row tiles: 3 + partial), batch 2
    ((1, 13, 17, 70), 32, 1, (12, 16, 69), 5, False),   # cropped skip
    ((1, 9, 11, 53), 64, 1, None, 9, True),             # the virtual cost volume as input
])
def test_conv3d_s3_second_tiling_is_bit_identical(cv, shape, cin, relu, res_shape, grid, virtual):
    """dsm_conv3d_s3_args.tiling = 2 (4 x 32 tiles, two workgroups per CU, waves split over couts)
    computes the same arithmetic in the same order as tiling = 1: identical bits on both outputs."""
    B, D, H, W = shape
    w = seeded(22, 32, cin, 3, 3, 3, scale=(2.0 / (27 * cin)) ** 0.5).cuda()
    scale, shift = (seeded(23, 32).abs() + 0.5).cuda(), seeded(24, 32).cuda()
    res = seeded(25, B, 32, *res_shape).cuda() if res_shape else None
    if virtual:
        fL, fR = seeded(41, B, cin // 2, H, W).cuda(), seeded(42, B, cin // 2, H, W).cuda()
        xs = cv.concat_volume_s3(fL, fR, D, True, materialise=False)
    else:
        xs = cv.s3_from_tensor(seeded(21, B, cin, D, H, W).cuda())
    packed = cv.pack_conv3d_s3_weight(w)
    got = {}
    for tiling in (1, 2):
        old = cv.set_option("s3_tiling", tiling)
        try:
            got[tiling] = cv.conv3d_s3_block(xs, packed, scale, shift, res, relu=relu, out="both", grid=grid)
        finally:
            cv.set_option("s3_tiling", old)
    assert torch.equal(got[1][0], got[2][0])
    assert torch.equal(got[1][1].buf, got[2][1].buf)


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
    """PSMNet eval forward four ways -- default (virtual volume + z-sliding S3 layers), S3 layers on
    a materialised fp32 volume, S3 hand-over everywhere (towers and hourglasses too: the ``s3in``
    option), and the r01 path (no S3 at all): each within 1e-3 px of the reference golden."""
    from tests.golden.make_goldens import images
    from tests.helpers import golden_state
    from dsmnet_amd.models import model_create_by_name
    sd, cfg = golden_state(golden_e2e, "psmnet")
    imL, imR = images(cfg["image_seed"], *cfg["hw"])
    m = model_create_by_name("psmnet", 192)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    outs = {}
    for name, s3, fuse, s3in in (("default", True, True, False), ("s3-on-fp32-volume", True, False, False),
                                 ("s3-everywhere", True, True, True), ("r01", False, False, False)):
        o1, o2, o3 = cv.set_option("s3", s3), cv.set_option("fuse_volume", fuse), cv.set_option("s3in", s3in)
        try:
            with torch.no_grad():
                outs[name] = m(imL.cuda(), imR.cuda())[1]
        finally:
            cv.set_option("s3", o1), cv.set_option("fuse_volume", o2), cv.set_option("s3in", o3)
        for pname, p in zip(("pred3", "pred2", "pred1"), outs[name]):
            golden_e2e.compare("e2e.psmnet." + pname, p, 1e-3)
    assert maxerr(outs["default"][0], outs["r01"][0]) <= 1e-3


@pytest.mark.parametrize("cin,cout,stride,shape", [
    (32, 64, 2, (1, 6, 12, 40)), (64, 64, 2, (1, 5, 9, 37)), (64, 64, 1, (2, 3, 7, 35)),
    (32, 32, 1, (1, 5, 17, 33)), (64, 32, 1, (1, 4, 16, 64)),
])
def test_conv3d_bf16x3_reads_s3_input(cv, cin, cout, stride, shape):
    """The bf16x3 convolution with its input handed over pre-split (x_s3): equal to the same layer
    fed the fp32 tensor up to summation order, and to the CPU reference."""
    B, D, H, W = shape
    x = seeded(61, B, cin, D, H, W)
    w = seeded(62, cout, cin, 3, 3, 3, scale=0.05)
    sc, sh = seeded(63, cout).abs() + 0.5, seeded(64, cout)
    want = F.conv3d(x.double(), w.double(), stride=stride, padding=1)
    want = (want * sc.double().view(1, -1, 1, 1, 1) + sh.double().view(1, -1, 1, 1, 1)).relu().float()
    assert cv.conv_s3in_eligible(cin, cout, stride, False)
    xs = cv.s3_from_tensor(x.cuda())
    y, ys3 = cv.conv3d_block(xs, cv.pack_conv_weight_s3in(w.cuda()), cout, sc.cuda(), sh.cuda(),
                             stride=stride, relu=1, out="both")
    ref = cv.conv3d_block(x.cuda(), cv.pack_conv3d_weight(w.cuda(), False), cout, sc.cuda(), sh.cuda(),
                          stride=stride, relu=1)
    tol = 2e-5 * max(1.0, want.abs().max().item())
    assert maxerr(y, want) <= tol and maxerr(y, ref) <= tol
    assert torch.equal(ys3.to_tensor(), y)


@pytest.mark.parametrize("cin,cout,dil,hw,res", [
    (32, 32, 1, (192, 96), True),      # 16-row tiles
    (32, 32, 1, (20, 45), False),      # 8-row tiles, ragged
    (64, 64, 1, (33, 70), True),
    (64, 128, 1, (24, 40), False),
    (128, 128, 1, (17, 33), True),
    (128, 128, 2, (24, 50), True),     # dilation 2 (layer4)
])
def test_conv2d_bf16x3_reads_s3_input(cv, cin, cout, dil, hw, res):
    """2-D tower layers (models/psmnet/submodule.py:10-43) with the S3 hand-over between them."""
    H, W = hw
    x = seeded(71, 2, cin, H, W)
    w = seeded(72, cout, cin, 3, 3, scale=0.05)
    sc, sh = seeded(73, cout).abs() + 0.5, seeded(74, cout)
    r = seeded(75, 2, cout, H, W) if res else None
    want = F.conv2d(x.double(), w.double(), padding=dil, dilation=dil)
    want = want * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)
    if res:
        want = want + r.double()
    want = want.float()
    assert cv.conv_s3in_eligible(cin, cout, 1, False, kd=1, k=3, dil=dil)
    xs = cv.s3_from_tensor(x.cuda().unsqueeze(2))                      # the map as a (B,C,1,H,W) volume
    y, ys3 = cv.conv2d_block(xs, cv.pack_conv_weight_s3in(w.cuda()), cout, sc.cuda(), sh.cuda(),
                             None if r is None else r.cuda(), dilation=dil, out="both")
    assert tuple(y.shape) == (2, cout, H, W)
    assert maxerr(y, want) <= 2e-5 * max(1.0, want.abs().max().item())
    assert torch.equal(ys3.to_tensor().squeeze(2), y)
    ref = cv.conv2d_block(x.cuda(), cv.pack_conv2d_weight(w.cuda()), cout, sc.cuda(), sh.cuda(),
                          None if r is None else r.cuda(), dilation=dil)
    assert maxerr(y, ref) <= 2e-5 * max(1.0, want.abs().max().item())
