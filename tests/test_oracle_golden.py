"""CPU: the oracle restatement against the golden fixtures generated from the reference.

The fixtures (tests/golden/*.npz) were written by tests/golden/make_goldens.py from the
reference's own source; these tests keep the oracle pinned to them wherever the
repository goes (the reference tree itself does not travel)."""
import pytest
import torch

from tests.helpers import seeded
from oracle import models as OM
from oracle import ops as OO


def test_corr1d_matches_golden(golden_ops):
    for case in golden_ops.meta["corr1d_cases"]:
        shp, seed, tag = case["shape"], case["seed"], case["tag"]
        fL = seeded(seed, *shp).requires_grad_(True)
        fR = seeded(seed + 100, *shp).requires_grad_(True)
        out = OO.corr1d(fL, fR, case["D"], case["s"], case["k"])
        cot = seeded(seed + 200, *out.shape)
        gL, gR = torch.autograd.grad(out, (fL, fR), cot)
        golden_ops.compare(tag + ".out", out, 1e-5)
        golden_ops.compare(tag + ".dL", gL, 1e-4)
        golden_ops.compare(tag + ".dR", gR, 1e-4)


def test_corr1d_semantics():
    """The traps of SURVEY.md section 7: un-normalised dot product, stride multiplies the
    shift, planes past the width stay zero, box filter counts the zero padding."""
    fL, fR = seeded(1, 1, 3, 2, 5), seeded(2, 1, 3, 2, 5)
    out = OO.corr1d(fL, fR, 7, 2, 1)
    assert out.shape == (1, 7, 2, 5)
    assert torch.equal(out[:, 0], (fL * fR).sum(1))
    assert torch.allclose(out[0, 1, :, 2:], (fL[0, :, :, 2:] * fR[0, :, :, :3]).sum(0))
    assert out[0, 1, :, :2].abs().max() == 0 and out[0, 3:].abs().max() == 0
    box = OO.corr1d(fL, fR, 2, 1, 3)
    raw = OO.corr1d(fL, fR, 2, 1, 1)
    assert torch.allclose(box[0, 0, 0, 0], raw[0, 0, :2, :2].sum() / 9.0)
    with pytest.raises(AssertionError):
        OO.corr1d(fL, fR, 2, 1, 2)


def test_volume_matches_golden(golden_ops):
    for case in golden_ops.meta["volume_cases"]:
        shp, seed, tag = case["shape"], case["seed"], case["tag"]
        fL = seeded(seed, *shp).requires_grad_(True)
        fR = seeded(seed + 100, *shp).requires_grad_(True)
        vol = OO.concat_volume(fL, fR, case["D"], case["mask_left"])
        cot = seeded(seed + 200, *vol.shape)
        gL, gR = torch.autograd.grad(vol, (fL, fR), cot)
        golden_ops.compare(tag + ".vol", vol, 0.0)
        golden_ops.compare(tag + ".dL", gL, 1e-5)
        golden_ops.compare(tag + ".dR", gR, 1e-5)


def test_volume_mask_conventions():
    fL, fR = seeded(3, 1, 2, 1, 6), seeded(4, 1, 2, 1, 6)
    gc = OO.concat_volume(fL, fR, 4, mask_left=False)
    psm = OO.concat_volume(fL, fR, 4, mask_left=True)
    assert torch.equal(gc[:, :2, 3], fL)                      # GCNet: left at every x
    assert psm[0, :2, 3, :, :3].abs().max() == 0               # PSMNet: both halves masked
    assert torch.equal(psm[0, :2, 3, :, 3:], fL[0, :, :, 3:])
    assert torch.equal(gc[:, 2:], psm[:, 2:])
    assert torch.equal(gc[0, 2:, 2, :, 2:], fR[0, :, :, :4])


def _softargmin_inputs(case):
    kind, seed = case["kind"], case["seed"]
    c = seeded(seed, *case["cost_shape"], scale=3.0)
    if kind == "onehot":
        c = c * 0.01
        Dc, H, W = case["cost_shape"][2:]
        gseed = seed if case["form"] == "psm" else seed - 50 + 1
        idx = torch.randint(0, Dc, (H, W), generator=torch.Generator().manual_seed(gseed))
        c[0, 0].scatter_(0, idx.unsqueeze(0), 12.0 if case["form"] == "psm" else -12.0)
    return c


def test_softargmin_matches_golden(golden_ops):
    for case in golden_ops.meta["softargmin_cases"]:
        c = _softargmin_inputs(case).requires_grad_(True)
        if case["form"] == "psm":
            out = OO.soft_argmin(c, tuple(case["out_size"]))
            cot = seeded(case["seed"] + 200, *out.shape)
        else:
            out = OO.soft_argmin(c, None, negate=True).unsqueeze(1)
            cot = seeded(case["seed"] + 200, *out.shape)
        (g,) = torch.autograd.grad(out, c, cot)
        golden_ops.compare(case["tag"] + ".disp", out, 1e-4)
        golden_ops.compare(case["tag"] + ".dcost", g, 1e-4, 1e-4)


def test_state_dict_contract():
    """Key names / shapes of the oracle state dicts (strict-loaded into the reference
    modules by make_goldens.py) -- the checkpoint compatibility contract."""
    sd = OM.init_state("psmnet", 0)
    assert sd["dres0.0.0.weight"].shape == (32, 64, 3, 3, 3)
    assert sd["dres2.conv5.0.weight"].shape == (64, 64, 3, 3, 3)
    assert sd["dres2.conv6.0.weight"].shape == (64, 32, 3, 3, 3)      # (Cin, Cout, k, k, k)
    assert sd["classif3.2.weight"].shape == (1, 32, 3, 3, 3)
    n = sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k)
    assert n == 5224768
    g = OM.init_state("gcnet", 0)
    assert g["layer3d.l33.0.weight"].shape == (128, 64, 3, 3, 3)
    assert g["layer3d.l37.weight"].shape == (32, 1, 3, 3, 3)
    assert "layer3d.l33.1.running_mean" in g


def test_blocks_match_golden(golden_blocks):
    import torch.nn.functional as F
    from tests.golden.make_goldens import randomise_bn  # pure helper, no reference access
    m = golden_blocks.meta["blocks"]
    sd = randomise_bn(OM.init_state("psmnet", m["psm_state_seed"]), m["psm_bn_seed"])
    x64 = seeded(m["x64_seed"], *m["x64_shape"])
    x32 = seeded(m["x32_seed"], *m["x32_shape"])
    for training in (False, True):
        mode = "train" if training else "eval"
        n = OM.Net({k: v.clone() for k, v in sd.items()}, training=training)
        with torch.no_grad():
            m0 = F.relu(OM._cbn3(n, F.relu(OM._cbn3(n, x64, "dres0.0", 32)), "dres0.2", 32))
            a1, pre1, post1 = OM._hourglass(n, x32, "dres2", None, None)
            a2, pre2, post2 = OM._hourglass(n, x32, "dres3", pre1, post1)
        golden_blocks.compare("block3d.psm.%s.dres0" % mode, m0, 1e-4)
        golden_blocks.compare("block3d.psm.%s.hg1.out" % mode, a1, 1e-4)
        golden_blocks.compare("block3d.psm.%s.hg2.post" % mode, post2, 1e-4)
        golden_blocks.compare("block3d.psm.%s.hg2.out" % mode, a2, 1e-4)


@pytest.mark.parametrize("name", ["gcnet", "dispnetcorr"])
def test_e2e_oracle_matches_golden(golden_e2e, name):
    from tests.golden.make_goldens import images
    from tests.helpers import golden_state
    sd, cfg = golden_state(golden_e2e, name)
    imL, imR = images(cfg["image_seed"], *cfg["hw"])
    with torch.no_grad():
        out = OM.forward(name, sd, imL, imR)
    if name == "gcnet":
        golden_e2e.compare("e2e.gcnet.disp", out, 1e-3)
    else:
        for i, o in enumerate(out[1]):
            golden_e2e.compare("e2e.dispnetcorr.pr%d" % i, o, 1e-4)


def test_right_referenced_volume_matches_golden():
    """gcnet_LR's xR (models/gcnet.py:155-164): the oracle against the fixture written from the
    reference's own lines (tests/golden/make_goldens.py --only lr)."""
    from tests.conftest import Golden
    g = Golden("lr")
    for case in g.meta["volume_lr_cases"]:
        fL, fR = seeded(case["seed"], *case["shape"]), seeded(case["seed"] + 100, *case["shape"])
        g.compare(case["tag"] + ".xR", OO.concat_volume_right(fL, fR, case["D"]), 0.0)
