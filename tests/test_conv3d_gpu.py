"""GPU parity: the fp32-MFMA 3-D convolution block (through the C ABI) against the
oracle's F.conv3d / F.conv_transpose3d + BN + cropped skip + ReLU on seeded inputs.

Tolerance: both sides are fp32 with fp32 accumulation over K = 27*Cin <= 3456 terms
of O(1) data scaled by He weights; outputs are O(1), summation order differs, so the
bound is 2e-4 absolute (measured ~2e-5)."""
import pytest
import torch

from oracle import ops as OO
from tests.helpers import maxerr, seeded

pytestmark = pytest.mark.gpu
TOL = 2e-4


@pytest.fixture(scope="module")
def cv(hip_lib):
    from dsmnet_amd import costvolume
    return costvolume


def run_case(cv, cin, cout, stride, transposed, shape, with_bn=True, with_res=False,
             relu=True, res_shrink=0, bias=False, seed=5):
    B, D, H, W = shape
    x = seeded(seed, B, cin, D, H, W)
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = seeded(seed + 1, *wshape, scale=(2.0 / (27 * cout)) ** 0.5)
    b = seeded(seed + 2, cout, scale=0.1) if bias else None
    bn = None
    if with_bn:
        g = torch.Generator().manual_seed(seed + 3)
        bn = (0.5 + torch.rand(cout, generator=g), torch.randn(cout, generator=g) * 0.2,
              torch.randn(cout, generator=g) * 0.3, 0.5 + torch.rand(cout, generator=g))
    osz = cv.conv3d_out_size((D, H, W), stride, transposed)
    res = None
    if with_res:
        rs = tuple(max(1, v - res_shrink) for v in osz)
        res = seeded(seed + 4, B, cout, *rs)
    with torch.no_grad():
        ref = OO.conv3d_block(x, w, b, stride, transposed, bn, res, relu)
    scale = shift = None
    if with_bn:
        gamma, beta, mean, var = bn
        scale = gamma / torch.sqrt(var + 1e-5)
        shift = beta - mean * scale + (b * scale if bias else 0)
    elif bias:
        scale, shift = torch.ones(cout), b
    packed = cv.pack_conv3d_weight(w.cuda(), transposed)
    y = cv.conv3d_block(x.cuda(), packed, cout,
                        None if scale is None else scale.cuda().contiguous(),
                        None if shift is None else shift.cuda().contiguous(),
                        None if res is None else res.cuda(), stride, transposed, relu)
    assert tuple(y.shape) == tuple(ref.shape), (y.shape, ref.shape)
    err = maxerr(y, ref)
    assert err <= TOL, "max abs err %.3e" % err
    return err


@pytest.mark.parametrize("cin,cout,stride,transposed", [
    (64, 32, 1, False), (32, 32, 1, False), (64, 64, 1, False), (128, 128, 1, False),
    (32, 64, 2, False), (64, 64, 2, False), (64, 128, 2, False),
    (64, 64, 2, True), (64, 32, 2, True), (128, 64, 2, True),
    (32, 1, 1, False), (32, 1, 2, True),
])
def test_conv3d_configs(cv, cin, cout, stride, transposed):
    run_case(cv, cin, cout, stride, transposed, (1, 6, 12, 40))


@pytest.mark.parametrize("shape", [(1, 5, 7, 37), (2, 3, 9, 33), (1, 1, 1, 1), (1, 2, 17, 70)])
@pytest.mark.parametrize("stride,transposed", [(1, False), (2, False), (2, True)])
def test_conv3d_ragged_shapes(cv, shape, stride, transposed):
    run_case(cv, 32, 32, stride, transposed, shape)
    run_case(cv, 32, 1, stride, transposed, shape) if stride == 1 or transposed else None


def test_conv3d_epilogue_variants(cv):
    run_case(cv, 32, 32, 1, False, (1, 6, 12, 40), with_bn=False, relu=False)
    run_case(cv, 32, 32, 1, False, (1, 6, 12, 40), with_res=True, relu=False)   # dres1 + cost0
    run_case(cv, 64, 64, 1, False, (1, 6, 12, 40), with_res=True, relu=True)    # conv2 + postsqu
    run_case(cv, 32, 1, 1, False, (1, 6, 12, 40), with_bn=False, with_res=True, relu=False)
    run_case(cv, 64, 32, 1, False, (1, 4, 8, 40), bias=True)                     # GCNet conv3d_bn
    run_case(cv, 32, 1, 2, True, (1, 4, 8, 20), with_bn=False, bias=True, relu=False)  # GCNet l37


def test_conv3d_crop_add(cv):
    """myadd_3d: deconv of a (34 -> 68) level against a 67-long skip (the 540x960 case,
    SURVEY.md section 7) -- output takes the smaller size."""
    run_case(cv, 64, 64, 2, True, (1, 3, 5, 17), with_res=True, res_shrink=1)
    run_case(cv, 64, 32, 2, True, (1, 3, 5, 17), with_res=True, res_shrink=1, relu=False)


def test_conv3d_large_tiles(cv):
    """Big enough for the 8-row (TM=2) tiles and the XCD-aware persistent schedule."""
    run_case(cv, 32, 32, 1, False, (1, 48, 64, 128))
    run_case(cv, 64, 64, 1, False, (1, 24, 64, 128), with_res=True)
    run_case(cv, 64, 64, 2, True, (1, 12, 32, 64), with_res=True)


def test_conv3d_cout1_zslide_segments(cv):
    """Classifier head 32->1 at sizes where the z-sliding kernel walks 8- and 16-plane segments
    (ragged last segment, ragged rows/columns, cropped skip as in PSMNet's cost2/cost3)."""
    run_case(cv, 32, 1, 1, False, (1, 40, 96, 320), with_bn=False, with_res=True, relu=False)
    run_case(cv, 32, 1, 1, False, (1, 37, 93, 317), with_bn=False, relu=False)
    run_case(cv, 32, 1, 1, False, (1, 33, 190, 610), with_bn=False, with_res=True, relu=False)


def test_relayout_roundtrip(cv):
    x = seeded(3, 2, 24, 5, 7, 9).cuda()
    cl = cv.to_channels_last_3d(x)
    assert cl.is_contiguous(memory_format=torch.channels_last_3d) and torch.equal(cl, x)
    back = cv.to_contiguous_3d(cl)
    assert back.is_contiguous() and torch.equal(back, x)


def test_bf16x3_split_agrees_with_fp32_mfma(cv):
    """The bf16x3 kernel (fp32 operands split exactly into three bf16 terms, six MFMAs per
    product, fp32 accumulate) against the fp32-input MFMA kernel on the same packed weights:
    a subprocess per mode, since the precision switch is read once per process."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, ".")
from dsmnet_amd import costvolume as cv
torch.manual_seed(0)
x = (torch.randn(1, 32, 20, 37, 70, device="cuda") * 3).contiguous(memory_format=torch.channels_last_3d)
w = torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.05
y = cv.conv3d_block(x, cv.pack_conv3d_weight(w, False), 32, None, None, None, 1, False, 0)
torch.save(y.cpu(), sys.argv[1])
'''
    outs = {}
    for mode in ("fp32", "bf16x3"):
        path = "/tmp/dsm_prec_%s_%d.pt" % (mode, os.getpid())
        env = dict(os.environ, DSM_CONV_PRECISION=mode)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        outs[mode] = torch.load(path, weights_only=True)
        os.remove(path)
    a, b = outs["fp32"], outs["bf16x3"]
    ref = torch.nn.functional.conv3d(
        (torch.manual_seed(0), torch.randn(1, 32, 20, 37, 70, device="cuda") * 3)[1].double().cpu(),
        (torch.randn(32, 32, 3, 3, 3, device="cuda") * 0.05).double().cpu(), padding=1)
    scale = ref.abs().max().item()
    e_fp32, e_split = (a.double() - ref).abs().max().item(), (b.double() - ref).abs().max().item()
    # both within a few fp32 ulps of the float64 result at this reduction length (K = 864) ...
    assert e_fp32 <= 2e-6 * scale and e_split <= 2e-6 * scale, (e_fp32, e_split, scale)
    # ... and the split path is not allowed to be meaningfully worse than the fp32 MFMA
    assert e_split <= 4 * max(e_fp32, 1e-7 * scale), (e_fp32, e_split)


@pytest.mark.parametrize("cin,cout,stride,transposed", [
    (32, 32, 1, False), (64, 32, 1, False), (64, 64, 1, False),      # conv_zs_kernel<3>, conv_split_kernel<3, 2, TM, 3, 1>
    (32, 64, 2, False), (64, 64, 2, False),                          # ... <3, 2, 1, 3, 1, S = 2>
    (64, 32, 2, True), (64, 64, 2, True),                            # deconv_split_kernel<3, 1|2>
])
@pytest.mark.parametrize("shape", [(2, 5, 9, 37), (1, 7, 18, 70)])
def test_bf16x3_variants_on_ragged_volumes(cv, cin, cout, stride, transposed, shape):
    """Every bf16x3 3-D variant on sizes that leave partial tiles in every dimension, batch 2,
    with the cropped skip add: tighter than the general bound (measured 3e-6 ... 2.2e-5 absolute, fp32 rounding at K = 864 ... 1728)."""
    old = cv.set_option("conv_precision", "bf16x3")
    try:
        err = run_case(cv, cin, cout, stride, transposed, shape, with_res=True,
                       res_shrink=0 if not transposed else 1, seed=31)
    finally:
        cv.set_option("conv_precision", old)
    assert err <= 5e-5, err
    from dsmnet_amd import _lib
    import ctypes
    a = _lib.Conv3dArgs()
    B, D, H, W = shape
    osz = cv.conv3d_out_size((D, H, W), stride, transposed)
    a.x = a.w_packed = a.y = 16
    a.B, a.Cin, a.Cout = B, cin, cout
    a.Di, a.Hi, a.Wi = D, H, W
    a.Do, a.Ho, a.Wo = osz
    a.stride, a.transposed, a.relu, a.kd, a.k, a.dil = stride, int(transposed), 1, 3, 3, 1
    assert "bf16x3" in cv.conv3d_plan_name(a)
