"""GPU parity: the 2-D form of the MFMA convolution (feature towers)
against torch's CPU F.conv2d + eval BatchNorm + skip + ReLU.  Same 2e-4 bound as the 3-D blocks."""
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import maxerr, seeded

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cv(hip_lib):
    from dsmnet_amd import costvolume
    return costvolume


@pytest.mark.parametrize("cin,cout,k,stride,dil,shape", [
    (32, 32, 3, 1, 1, (2, 24, 40)), (32, 32, 3, 1, 1, (1, 96, 160)),   # 8-row tiles
    (32, 32, 3, 1, 1, (2, 190, 630)),                                   # 16-row tiles (1/2-resolution maps), ragged
    (64, 64, 3, 1, 1, (2, 13, 37)), (128, 128, 3, 1, 1, (1, 9, 33)),
    (128, 128, 3, 1, 2, (2, 12, 40)),                                   # layer4: dilation 2
    (16, 32, 3, 2, 1, (2, 25, 70)), (32, 64, 3, 2, 1, (1, 24, 64)),     # stride 2
    (128, 32, 1, 1, 1, (2, 12, 40)), (64, 128, 1, 1, 1, (1, 7, 35)), (32, 64, 1, 2, 1, (2, 24, 64)),
    (320, 128, 3, 1, 1, (1, 12, 40)),                                   # lastconv: 20 chunks
])
@pytest.mark.parametrize("residual,relu", [(False, True), (True, False)])
def test_conv2d_variants(cv, cin, cout, k, stride, dil, shape, residual, relu):
    B, H, W = shape
    x = seeded(1, B, cin, H, W)
    w = seeded(2, cout, cin, k, k, scale=(2.0 / (k * k * cout)) ** 0.5)
    g = torch.Generator().manual_seed(3)
    gamma, beta = 0.5 + torch.rand(cout, generator=g), torch.randn(cout, generator=g) * 0.2
    mean, var = torch.randn(cout, generator=g) * 0.3, 0.5 + torch.rand(cout, generator=g)
    pad = dil * (k - 1) // 2
    ref = F.batch_norm(F.conv2d(x, w, None, stride, pad, dil), mean, var, gamma, beta, False)
    res = seeded(4, *ref.shape) if residual else None
    if res is not None:
        ref = ref + res
    if relu:
        ref = F.relu(ref)
    scale = gamma / torch.sqrt(var + 1e-5)
    shift = beta - mean * scale
    packed = cv.pack_conv2d_weight(w.cuda())
    y = cv.conv2d_block(x.cuda().contiguous(memory_format=torch.channels_last), packed, cout,
                        scale.cuda(), shift.cuda(), None if res is None else res.cuda(),
                        stride, 1 if relu else 0, k, dil)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert maxerr(y, ref) <= 2e-4


def test_first_conv_zero_padded_channels(cv):
    """3 input channels staged as 16 (zeros), weights zero-padded at pack time."""
    from dsmnet_amd.blocks2d import stage_image_nhwc16
    x = seeded(5, 2, 3, 31, 50)
    w = seeded(6, 32, 3, 3, 3, scale=0.2)
    ref = F.conv2d(x, w, None, 2, 1)
    packed = cv.pack_conv2d_weight(w.cuda(), 16)
    y = cv.conv2d_block(stage_image_nhwc16(x.cuda()), packed, 32, stride=2)
    assert maxerr(y, ref) <= 2e-4


def test_tower_fused_equals_stock(hip_lib):
    """PSMNet feature_extraction: fused eval path vs the same modules run as stock torch
    layers on the GPU (grad mode on selects the stock path)."""
    from dsmnet_amd.models.psmnet.submodule import feature_extraction
    torch.manual_seed(0)
    fe = feature_extraction().cuda()
    x = torch.randn(2, 3, 256, 512, device="cuda")
    fe.train()
    with torch.no_grad():
        fe(x)                                   # populate BN statistics (stock train path)
    fe.eval()
    with torch.no_grad():
        fused = fe(x)
    stock = fe(x)                               # grad enabled -> stock torch modules
    assert fused.shape == stock.shape == (2, 32, 64, 128)
    scale = stock.abs().max().item()
    assert maxerr(fused, stock.detach()) <= 2e-4 * max(1.0, scale)
