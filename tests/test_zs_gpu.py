"""GPU parity of the z-sliding convolution (dsmnet_amd/csrc/conv_zs.hpp; plan kind 7 of
dsm_conv3d_fwd): Conv3d(k3, s1, p1) to 32 channels -- dres0, dres1, classif*.0 of PSMNet
(models/psmnet/stackhourglass.py:73-98) -- in every precision mode, against torch's float64
convolution on the CPU; and the cost volume it stages without ever materialising it
(stackhourglass.py:124-133, models/gcnet.py:130-135) against the same kernel fed the volume the
oracle builds."""
import pytest
import torch
import torch.nn.functional as F

from oracle import ops as OO
from tests.helpers import maxerr, seeded
from tests.test_f16_gpu import F16X2_MAX, F16X2_RMS, F16_MAX, F16_RMS, errors, precision

pytestmark = pytest.mark.gpu
LIMITS = {"bf16x3": (2e-6, 6.5e-7), "f16x2": (F16X2_MAX, F16X2_RMS), "f16": (F16_MAX, F16_RMS)}


@pytest.fixture(scope="module")
def cv(hip_lib):
    from dsmnet_amd import costvolume
    return costvolume


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
    return y


def with_grid(cv, grid):
    """conv_flags with the persistent grid size forced (bits 16..31 of dsm_conv3d_args.flags)."""
    from dsmnet_amd import _lib
    return cv.set_option("conv_flags", grid << _lib.DSM_CONV_BLOCKS_SHIFT)


@pytest.mark.parametrize("shape,cin,relu,res_shape,grid", [
    ((1, 6, 12, 40), 32, 1, None, 0),
    ((1, 6, 12, 40), 32, 0, None, 3),             # ranges that cross column borders mid-z
    ((2, 5, 9, 33), 64, 2, (5, 9, 33), 7),        # batch 2, ragged tiles, ReLU before the skip add
    ((1, 13, 17, 70), 32, 1, (12, 16, 69), 5),    # cropped skip: the output is the common corner
    ((1, 1, 8, 32), 32, 1, None, 0),              # one plane: both z-neighbours outside the volume
    ((1, 2, 3, 5), 32, 0, None, 0),               # smaller than a tile in every direction
    ((1, 48, 24, 64), 64, 1, (48, 24, 64), 0),    # the dres0[0] shape class, 256 workgroups
    ((1, 7, 19, 45), 96, 1, None, 11),            # three channel groups
])
@pytest.mark.parametrize("mode", ["bf16x3", "f16x2", "f16"])
def test_conv3d_zs_vs_cpu_fp64(cv, mode, shape, cin, relu, res_shape, grid):
    B, D, H, W = shape
    x = seeded(21, B, cin, D, H, W)
    w = seeded(22, 32, cin, 3, 3, 3, scale=(2.0 / (27 * cin)) ** 0.5)
    scale, shift = seeded(23, 32).abs() + 0.5, seeded(24, 32)
    res = seeded(25, B, 32, *res_shape) if res_shape else None
    want = _reference_conv(x, w, scale, shift, res, relu)
    old = with_grid(cv, grid)
    try:
        with precision(cv, mode):
            y = cv.conv3d_block(x.cuda(), cv.pack_conv3d_weight(w.cuda(), False), 32, scale.cuda(), shift.cuda(),
                                None if res is None else res.cuda(), relu=relu)
    finally:
        cv.set_option("conv_flags", old)
    assert tuple(y.shape) == tuple(want.shape)
    emax, erms = errors(y, want)
    # the band was measured at K = 27 Cin <= 1728; rounding in an fp32 accumulation grows like sqrt(K)
    grow = max(1.0, cin / 64.0) ** 0.5
    assert emax <= LIMITS[mode][0] * grow and erms <= LIMITS[mode][1] * grow, (emax, erms)
    if mode != "bf16x3":
        assert y._dsm_amax.item() == y.abs().max().item()


def test_the_plan_of_a_32_channel_stride_1_layer_is_the_z_sliding_kernel(cv):
    from dsmnet_amd import _lib
    a = _lib.Conv3dArgs()
    a.x = a.w_packed = a.y = a.x_amax = 16
    a.B, a.Cin, a.Cout = 1, 64, 32
    a.Di = a.Do = 5
    a.Hi = a.Ho = 9
    a.Wi = a.Wo = 40
    a.stride = 1
    assert cv.conv3d_plan_name(a) == "conv3d_zs_bf16x3_mfma_kernel"
    a.precision, a.vol_virtual = _lib.DSM_PREC_F16X2, 1
    assert cv.conv3d_plan_name(a) == "conv3d_zs_f16x2_mfma_kernel<vol>"
    a.flags = _lib.DSM_CONV_FP32_MFMA                       # the exact fp32-input MFMA cannot stage a virtual volume
    with pytest.raises(_lib.DsmnetHipError):
        cv.conv3d_plan_name(a)
    a.flags, a.Cout = 0, 64
    with pytest.raises(_lib.DsmnetHipError):
        cv.conv3d_plan_name(a)


@pytest.mark.parametrize("mask_left", [False, True])
@pytest.mark.parametrize("mode", ["bf16x3", "f16x2", "f16"])
@pytest.mark.parametrize("shape,D,grid", [((2, 32, 11, 53), 14, 0), ((1, 32, 9, 33), 40, 5), ((1, 64, 5, 70), 6, 0)])
def test_virtual_volume_equals_the_materialised_one(cv, mode, mask_left, shape, D, grid):
    """The first 3-D convolution staged straight from the towers' output (the volume never
    written; D > W included) gives the same BITS as the same kernel reading the volume the oracle
    builds (oracle/ops.py concat_volume: stackhourglass.py:124-133 / gcnet.py:130-135)."""
    B, C, H, W = shape
    fL, fR = seeded(41, *shape), seeded(42, *shape)
    w = seeded(43, 32, 2 * C, 3, 3, 3, scale=0.04)
    packed = cv.pack_conv3d_weight(w.cuda(), False)
    vol = OO.concat_volume(fL, fR, D, mask_left=mask_left)
    both = torch.cat([fL, fR], 0).cuda().contiguous(memory_format=torch.channels_last)
    old = with_grid(cv, grid)
    try:
        with precision(cv, mode):
            real = cv.conv3d_block(vol.cuda(), packed, 32, relu=1)
            virt = cv.conv3d_block(cv.VirtualVolume(both, D, mask_left), packed, 32, relu=1)
    finally:
        cv.set_option("conv_flags", old)
    assert tuple(virt.shape) == (B, 32, D, H, W)
    if mode == "bf16x3":
        assert torch.equal(virt, real)
    else:
        # the materialised volume's own maximum can differ from the features' (masked columns) by
        # no more than a power-of-two step of the scale: same bits when the exponents agree
        assert maxerr(virt, real) <= LIMITS[mode][0] * real.abs().max().item()
    want = F.conv3d(vol.double(), w.double(), padding=1).relu()
    emax, erms = errors(virt, want)
    grow = max(1.0, 2 * C / 64.0) ** 0.5        # the band was measured at K = 27 Cin <= 1728 (see above)
    assert emax <= LIMITS[mode][0] * grow and erms <= LIMITS[mode][1] * grow, (emax, erms)


def test_psmnet_paths_agree(cv, golden_e2e):
    """PSMNet eval forward with the volume staged virtually (default) and materialised, in the two
    fp32-accurate modes: each within 1e-3 px of the reference golden."""
    from tests.golden.make_goldens import images
    from tests.helpers import golden_state
    from dsmnet_amd.models import model_create_by_name
    sd, cfg = golden_state(golden_e2e, "psmnet")
    imL, imR = images(cfg["image_seed"], *cfg["hw"])
    m = model_create_by_name("psmnet", 192)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    outs = {}
    for mode in ("bf16x3", "f16x2"):
        for fuse in (True, False):
            o1 = cv.set_option("fuse_volume", fuse)
            try:
                with precision(cv, mode), torch.no_grad():
                    outs[mode, fuse] = m(imL.cuda(), imR.cuda())[1]
            finally:
                cv.set_option("fuse_volume", o1)
            for pname, p in zip(("pred3", "pred2", "pred1"), outs[mode, fuse]):
                golden_e2e.compare("e2e.psmnet." + pname, p, 1e-3)
    assert maxerr(outs["f16x2", True][0], outs["bf16x3", True][0]) <= 1e-3
