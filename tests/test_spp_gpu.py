"""PSMNet SPP head (csrc/spp.hip) against the oracle's restatement of
models/psmnet/submodule.py:81-99,126-137."""
import pytest
import torch

from oracle import models as OM
from tests.helpers import seeded, maxerr
from tests.golden.make_goldens import randomise_bn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(1, 96, 320), (2, 67, 91), (1, 135, 240), (1, 64, 64)])
def test_spp_head_matches_oracle(hip_lib, shape):
    from dsmnet_amd.models import model_create_by_name
    B, H, W = shape
    sd = randomise_bn(OM.init_state("psmnet", 0), 17)
    raw, skip = seeded(3, B, 64, H, W), seeded(4, B, 128, H, W)
    want = OM.psmnet_spp(OM.Net(sd, training=False), raw, skip)
    m = model_create_by_name("psmnet", 192)
    m.load_state_dict(sd, strict=True)
    fe = m.feature_extraction.cuda().eval()
    from dsmnet_amd import costvolume as cv
    with torch.no_grad():
        got = cv.spp_head(raw.cuda(), skip.cuda(), *fe._spp_params())
        stock = fe._spp_stock(raw.cuda(), skip.cuda())
    assert got.shape == want.shape == (B, 320, H, W)
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got[:, :192].cpu(), want[:, :192])          # copies: bit-exact
    assert maxerr(got.cpu(), want) <= 2e-5                          # pooled/conv/bilinear: fp32 rounding
    assert maxerr(stock.cpu(), want) <= 2e-5                        # the train-mode path agrees too


def test_spp_head_rejects_small_maps(hip_lib):
    from dsmnet_amd import costvolume as cv
    raw, skip = torch.zeros(1, 64, 40, 80, device="cuda"), torch.zeros(1, 128, 40, 80, device="cuda")
    w = torch.zeros(4, 128, 32, device="cuda")
    a = torch.zeros(4, 32, device="cuda")
    with pytest.raises(ValueError):
        cv.spp_head(raw, skip, w, a, a)


def test_spp_concat_reports_its_maximum(hip_lib):
    """The concat kernel raises the device scalar the fp16 convolution modes read as lastconv's x_amax
    (no separate pass over the 320-channel map)."""
    import torch
    from dsmnet_amd import costvolume as cv
    from tests.helpers import seeded
    old = cv.set_option("conv_precision", "f16x2")
    try:
        raw = seeded(3, 2, 64, 64, 96).cuda().contiguous(memory_format=torch.channels_last)
        skip = seeded(4, 2, 128, 64, 96).cuda().contiguous(memory_format=torch.channels_last)
        w_t = seeded(5, 4, 128, 32).cuda()
        scale, shift = (seeded(6, 4, 32).abs() + 0.5).cuda(), seeded(7, 4, 32).cuda()
        with cv.amax_scope(raw.device):
            out = cv.spp_head(raw, skip, w_t, scale, shift)
            assert out._dsm_amax.item() == out.abs().max().item()
    finally:
        cv.set_option("conv_precision", old)
