"""GPU parity, model level: the MI355X networks against the golden fixtures generated
from the reference and against the oracle run live on the host CPU.

Bound: the north-star's 1e-3 max-abs disparity error (BASELINE.json), on calibrated-BN
weights (SURVEY.md section 7 "Parity fragility")."""
import pytest
import torch
import torch.nn.functional as F

from oracle import models as OM
from tests.golden.make_goldens import images, randomise_bn
from tests.helpers import golden_state, maxerr, seeded

pytestmark = pytest.mark.gpu
DISP_TOL = 1e-3


def load(name, sd):
    from dsmnet_amd.models import model_create_by_name
    m = model_create_by_name(name, 192)
    m.load_state_dict(sd, strict=True)          # the checkpoint key contract
    return m.cuda().eval()


def test_psmnet_blocks_vs_golden(hip_lib, golden_blocks):
    """dres0 / hourglass / classif on (1,C,12,16,24) with non-trivial eval-mode BN."""
    meta = golden_blocks.meta["blocks"]
    sd = randomise_bn(OM.init_state("psmnet", meta["psm_state_seed"]), meta["psm_bn_seed"])
    m = load("psmnet", sd)
    x64 = seeded(meta["x64_seed"], *meta["x64_shape"]).cuda()
    x32 = seeded(meta["x32_seed"], *meta["x32_shape"]).cuda()
    with torch.no_grad():
        r0 = m.dres0(x64)
        o1, pre1, post1 = m.dres2(x32, None, None)
        o2, pre2, post2 = m.dres3(x32, pre1, post1)
        cl = m.classif1(x32)
    for nm, t in (("dres0", r0), ("hg1.out", o1), ("hg1.pre", pre1), ("hg1.post", post1),
                  ("hg2.out", o2), ("hg2.pre", pre2), ("hg2.post", post2), ("classif1", cl)):
        golden_blocks.compare("block3d.psm.eval." + nm, t, 5e-4)


def test_psmnet_end_to_end_256x512(hip_lib, golden_e2e):
    sd, cfg = golden_state(golden_e2e, "psmnet")
    imL, imR = images(cfg["image_seed"], *cfg["hw"])
    m = load("psmnet", sd)
    with torch.no_grad():
        scales, preds = m(imL.cuda(), imR.cuda())
        ref = OM.forward("psmnet", sd, imL, imR)
    assert scales == [0, 0, 0] and len(preds) == 3
    for nm, p, r in zip(("pred3", "pred2", "pred1"), preds, ref):
        assert p.shape == (1, 256, 512)
        golden_e2e.compare("e2e.psmnet." + nm, p, DISP_TOL)      # vs the reference itself
        assert maxerr(p, r) <= DISP_TOL, nm                      # vs the oracle, every pixel


def test_psmnet_rejects_train_mode(hip_lib):
    from dsmnet_amd.models import model_create_by_name
    m = model_create_by_name("psmnet", 192).cuda().train()
    x = torch.zeros(1, 64, 4, 8, 32, device="cuda")
    with pytest.raises(NotImplementedError):
        m.dres0(x)
