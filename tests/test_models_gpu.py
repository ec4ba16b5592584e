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


def test_psmnet_train_mode_forward_vs_golden(hip_lib, golden_blocks):
    """Train-mode BN (batch statistics) forward: HIP convolution + BatchNorm3d with batch
    stats, against the reference's train-mode goldens (gradients: test_conv3d_bwd_gpu.py)."""
    meta = golden_blocks.meta["blocks"]
    sd = randomise_bn(OM.init_state("psmnet", meta["psm_state_seed"]), meta["psm_bn_seed"])
    from dsmnet_amd.models import model_create_by_name
    m = model_create_by_name("psmnet", 192)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    x64 = seeded(meta["x64_seed"], *meta["x64_shape"]).cuda()
    x32 = seeded(meta["x32_seed"], *meta["x32_shape"]).cuda()
    with torch.no_grad():
        r0 = m.dres0(x64)
        o1, pre1, post1 = m.dres2(x32, None, None)
    golden_blocks.compare("block3d.psm.train.dres0", r0, 5e-4)
    golden_blocks.compare("block3d.psm.train.hg1.out", o1, 5e-4)
    golden_blocks.compare("block3d.psm.train.hg1.post", post1, 5e-4)
    assert int(m.dres0[0][1].num_batches_tracked) == 1
    y = m.dres0(x64)                                   # grad mode on: autograd graph recorded
    assert y.requires_grad


def test_gcnet_end_to_end_64x128(hip_lib, golden_e2e):
    """GCNet, D=192 (96 planes at 1/2), 64x128: left-unmasked volume, relu-before-skip
    epilogues, 128-channel levels, the 32->1 transposed head and the negated soft-argmin."""
    sd, cfg = golden_state(golden_e2e, "gcnet")
    imL, imR = images(cfg["image_seed"], *cfg["hw"])
    m = load("gcnet", sd)
    with torch.no_grad():
        scales, (disp,) = m(imL.cuda(), imR.cuda())
        ref = OM.forward("gcnet", sd, imL, imR)
    assert scales == [0] and disp.shape == (1, 1, 64, 128)
    golden_e2e.compare("e2e.gcnet.disp", disp, DISP_TOL)
    assert maxerr(disp, ref) <= DISP_TOL


def test_dispnetcorr_end_to_end_256x512(hip_lib, golden_e2e):
    """DispNetC (BASELINE config #1 shape, on the GPU): HIP Corr1d inside the stock 2-D net;
    all seven outputs.  The north-star's 1e-3 px is asserted where it holds (the full-resolution
    level of the pyramid; measured 6e-7 ... 1.5e-5), the measured error of every level is printed."""
    sd, cfg = golden_state(golden_e2e, "dispnetcorr")
    imL, imR = images(cfg["image_seed"], *cfg["hw"])
    m = load("dispnetcorr", sd)
    with torch.no_grad():
        scales, outs = m(imL.cuda(), imR.cuda())
    assert scales == list(range(7)) and len(outs) == 7
    errs = [golden_e2e.compare("e2e.dispnetcorr.pr%d" % i, o, DISP_TOL) for i, o in enumerate(outs)]
    print("DispNetC e2e max |disp - golden| per level: " + " ".join("%.1e" % e for e in errs))


def test_iresnet_end_to_end_256x512(hip_lib, golden_e2e):
    """iResNet: both correlations (D=81; kernel 3 / stride 2 / D=41) on the HIP path."""
    sd, cfg = golden_state(golden_e2e, "iresnet")
    imL, imR = images(cfg["image_seed"], *cfg["hw"])
    m = load("iresnet", sd)
    with torch.no_grad():
        torch.manual_seed(cfg["torch_seed"])       # imwrap's random epsilon
        scales, outs = m(imL.cuda(), imR.cuda())
    assert len(outs) == 10 and scales[:3] == [0, 1, 2]
    errs = [golden_e2e.compare("e2e.iresnet.out%d" % i, o, DISP_TOL) for i, o in enumerate(outs)]
    print("iResNet e2e max |disp - golden| per output: " + " ".join("%.1e" % e for e in errs))


def test_psmnet_540x960_crop_add(hip_lib):
    """The shape where myadd_3d really crops: 135 -> 68 -> 34 -> 68 -> 136
    vs 135.  Trunk only (volume from random features), against the oracle trunk."""
    from dsmnet_amd import costvolume as cv
    sd = randomise_bn(OM.init_state("psmnet", 0), 41)
    m = load("psmnet", sd)
    fl, fr = seeded(71, 1, 32, 17, 31) , seeded(72, 1, 32, 17, 31)     # odd sizes at every level
    with torch.no_grad():
        got = m.regularise(cv.concat_volume(fl.cuda(), fr.cuda(), 12, True))
        from oracle import ops as OO
        ref = OM.psmnet_trunk(OM.Net(sd), OO.concat_volume(fl, fr, 12, True))
    for g, r in zip(got, ref):
        assert g.shape == r.shape == (1, 1, 12, 17, 31)
        assert maxerr(g, r) <= 5e-4 * max(1.0, r.abs().max().item())


def test_psmnet_kitti_shape_375x1242(hip_lib, golden_e2e):
    """The reference's own timing shape (models/test_models_time.py:35: [1,3,375,1242]): odd
    sizes at every level (94 -> 47 -> 24 -> 48 vs 47: myadd_3d crops), non-integer upsampling
    ratios in H and W (generic soft-argmin kernel paths), against the oracle on the host CPU."""
    sd, _ = golden_state(golden_e2e, "psmnet")
    imL, imR = images(77, 375, 1242)
    m = load("psmnet", sd)
    with torch.no_grad():
        _, preds = m(imL.cuda(), imR.cuda())
        ref = OM.forward("psmnet", sd, imL, imR)
    for p, r in zip(preds, ref):
        assert p.shape == r.shape == (1, 375, 1242)
        assert maxerr(p, r) <= DISP_TOL


def test_psmnet_batch_of_two(hip_lib, golden_e2e):
    """B = 2: each pair of a batch equals the same pair run alone (pairs are independent --
    the property the multi-GPU sharding relies on)."""
    sd, cfg = golden_state(golden_e2e, "psmnet")
    a = images(cfg["image_seed"], *cfg["hw"])
    b = images(91, *cfg["hw"])
    m = load("psmnet", sd)
    with torch.no_grad():
        both = m(torch.cat([a[0], b[0]]).cuda(), torch.cat([a[1], b[1]]).cuda())[1]
        one = m(b[0].cuda(), b[1].cuda())[1]
    for pb, po in zip(both, one):
        assert pb.shape == (2, 256, 512)
        assert maxerr(pb[1:], po) <= 1e-4
    golden_e2e.compare("e2e.psmnet.pred3", both[0][:1], DISP_TOL)


def test_graphed_forward_equals_eager(hip_lib):
    """hipGraph replay of the whole PSMNet forward (dsmnet_amd/graphs.py): bit-identical to eager
    launches, and new inputs of the captured shape are picked up."""
    from dsmnet_amd.graphs import GraphedForward
    from dsmnet_amd.models import model_create_by_name
    torch.manual_seed(0)
    m = model_create_by_name("psmnet", 192).cuda().eval()
    for i in (1, 2, 3):
        with torch.no_grad():
            getattr(m, "classif%d" % i)[2].weight.mul_(1e-3)
    a, b = torch.rand(1, 3, 256, 512, device="cuda"), torch.rand(1, 3, 256, 512, device="cuda")
    c, d = torch.rand(1, 3, 256, 512, device="cuda"), torch.rand(1, 3, 256, 512, device="cuda")
    g = GraphedForward(m, a, b)
    with torch.no_grad():
        want_ab = [t.clone() for t in m(a, b)[1]]
        want_cd = [t.clone() for t in m(c, d)[1]]
    got_cd = [t.clone() for t in g(c, d)[1]]
    got_ab = [t.clone() for t in g(a, b)[1]]
    for w, x in zip(want_ab + want_cd, got_ab + got_cd):
        assert torch.equal(w, x)
    assert not torch.equal(got_ab[2], got_cd[2])
    with pytest.raises(ValueError):
        g(a[:, :, :128], b[:, :, :128])
    with pytest.raises(ValueError):
        GraphedForward(m.train(), a, b)


@pytest.mark.parametrize("net,size,tol", [("gcnet", (64, 128), 0.0), ("dispnetcorr", (256, 512), 1e-4),
                                         ("iresnet", (256, 512), 5e-3)])
def test_graphed_forward_other_models(hip_lib, net, size, tol):
    """The other three networks replay from a hipGraph too.  iResNet's warp draws a random
    epsilon per call in eager mode (utils/imwrap.py:70); a captured forward keeps the one drawn
    at capture time, hence its tolerance; DispNetC's stock MIOpen layers are not bit-reproducible
    between a captured and an eager run (GCNet, all HIP kernels of this library, is)."""
    from dsmnet_amd.graphs import GraphedForward
    from dsmnet_amd.models import model_create_by_name
    torch.manual_seed(0)
    m = model_create_by_name(net, 192).cuda().eval()
    a, b = torch.rand(1, 3, *size, device="cuda"), torch.rand(1, 3, *size, device="cuda")
    g = GraphedForward(m, a, b)
    with torch.no_grad():
        want = m(a, b)[1]
    got = g(a, b)[1]
    assert len(got) == len(want)
    for w, x in zip(want, got):
        assert w.shape == x.shape
        assert (w - x).abs().max().item() <= tol * max(1.0, w.abs().max().item())


def test_psmnet_raw_init_is_no_worse_than_the_cpu_fp32_path(hip_lib, golden_e2e):
    """The ill-conditioned case, pinned.  With the reference's RAW random
    initialisation (no head calibration) the cost entering the softmax has std ~3e3: the softmax
    is one-hot, fp32 rounding flips near-ties, and the reference's own fp32 forward is 0.2-0.3 px
    away from an fp64 run of itself -- 1e-3 px against the CPU path cannot hold there for ANY
    independent fp32 implementation.  What must hold: this path is not further from the fp64
    truth than the CPU fp32 path is, up to the scatter between two fp32 summation orders: factor
    2 on the largest and on the mean error of each head.  Which near-ties flip is chaotic (the
    largest error is one pixel's): measured r02 on two kernel generations, max 1.0-1.55x and
    mean 1.3-1.6x of the CPU path's (a third one -- the opt-in VALU first layer, exact fp32 FMAs --
    gave 2.01x on one head's largest error); a precision bug (a dropped bf16 term costs 2^-16
    relative) moves the mean by orders of magnitude."""
    from oracle import ops as OO
    sd, cfg = golden_state(golden_e2e, "psmnet")
    raw = 1.0 / float(golden_e2e.z["e2e.psmnet.head_scale"])      # undo the calibration of the heads
    OM.apply_head_scale("psmnet", sd, raw)
    imL, imR = images(cfg["image_seed"], *cfg["hw"])
    size = (192,) + tuple(cfg["hw"])
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        n64 = OM.Net(sd64)
        c64 = OM.psmnet_trunk(n64, OO.concat_volume(OM.psmnet_features(n64, imL.double()),
                                                    OM.psmnet_features(n64, imR.double()), 48, True))
        assert c64[2].std().item() > 500.0                           # it IS the ill-conditioned regime
        d64 = [OO.soft_argmin(c, size) for c in (c64[2], c64[1], c64[0])]
        d32 = OM.forward("psmnet", sd, imL, imR)
        m = load("psmnet", sd)
        dg = m(imL.cuda(), imR.cuda())[1]
    for name, a64, a32, ag in zip(("pred3", "pred2", "pred1"), d64, d32, dg):
        e_cpu = (a32.double() - a64).abs()
        e_gpu = (ag.double().cpu() - a64).abs()
        assert e_gpu.max().item() <= 2.0 * e_cpu.max().item() + 1e-3, \
            "%s: max |hip - fp64| %.4f vs |cpu32 - fp64| %.4f" % (name, e_gpu.max().item(), e_cpu.max().item())
        assert e_gpu.mean().item() <= 2.0 * e_cpu.mean().item() + 1e-6, \
            "%s: mean |hip - fp64| %.3e vs cpu %.3e" % (name, e_gpu.mean().item(), e_cpu.mean().item())


def test_gcnet_config3_full_size_256x512(hip_lib, golden_e2e):
    """BASELINE config #3 at its full size: GCNet, D=192, one 256x512 pair -- volume
    (1,64,96,128,256) = 805 MB, head l37 at 192x256x512 -- against the oracle on the host CPU
    (same synthetic checkpoint as the 64x128 golden case: calibrated BN + head scale)."""
    from dsmnet_amd import costvolume as cv
    sd, _ = golden_state(golden_e2e, "gcnet")
    imL, imR = images(29, 256, 512)
    m = load("gcnet", sd)
    with torch.no_grad():
        # the fixture's head scale was calibrated on the 64x128 pair; bring the soft-argmin logits
        # of THIS input to the same trained-network range (std 2) -- one factor on the linear
        # head l37, applied to the state dict both implementations then load
        fl, fr = m.features(imL.cuda(), imR.cuda())
        std = float(m.layer3d.cost(cv.concat_volume(fl, fr, m.D, mask_left=False)).std())
        OM.apply_head_scale("gcnet", sd, 2.0 / std)
        m = load("gcnet", sd)
        _, outs = m(imL.cuda(), imR.cuda())
        ref = OM.forward("gcnet", sd, imL, imR)
    out, r = outs[0], ref
    assert out.shape == r.shape == (1, 1, 256, 512)
    assert maxerr(out, r) <= DISP_TOL


def test_graphed_train_step_after_an_eager_step_on_the_same_model(hip_lib):
    """capturing a training step after an eager backward through the SAME model used
    to abort the process inside capture_end.  GraphedTrainStep now collects garbage, drops the
    gradients and synchronises before warm-up and capture.  Run in a child process so that an
    abort fails this test instead of ending the test run."""
    import subprocess
    import sys
    code = r'''
import torch
from dsmnet_amd import train
from dsmnet_amd.graphs import GraphedTrainStep
from dsmnet_amd.models import model_create_by_name
torch.manual_seed(0)
m = model_create_by_name("psmnet", 192).cuda()
with torch.no_grad():
    for i in (1, 2, 3):
        getattr(m, "classif%d" % i)[2].weight.mul_(1e-3)
g = torch.Generator().manual_seed(5)
left = torch.rand(1, 3, 256, 512, generator=g)
batch = torch.cat([left, torch.roll(left, -6, dims=3), torch.full((1, 1, 256, 512), 6.0)], 1).cuda()
lf = train.losses("supervised", 1, 0); lf.Weight_Adjust_levels(0)
opt = torch.optim.Adam(m.parameters(), lr=1e-4, capturable=True)
l0 = train.train_step(m, opt, lf, batch)[0]          # eager forward + backward + step
step = GraphedTrainStep(m, opt, lf, batch, warmup=2)  # then capture on the same model
l1 = float(step(batch)[0]); l2 = float(step(batch)[0])
assert l0 == l0 and l1 == l1 and l2 <= l1 * 1.05, (l0, l1, l2)
print("ok", l0, l1, l2)
'''
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       cwd=__import__("os").path.dirname(__import__("os").path.dirname(__file__)))
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])


def test_gcnet_feature2d_runs_on_the_hip_kernels(hip_lib, golden_e2e):
    """SURVEY 8f-1, second half: GCNet's 2-D tower (models/gcnet.py:14-29) -- its eight residual
    blocks are one launch each of the fused BasicBlock kernel in the fp16 modes (two launches of the
    MFMA convolution kernel each otherwise), the closing biased convolution one more launch
    (the 5x5 stride-2 stem stays stock); the tower's output equals the oracle's."""
    from dsmnet_amd import costvolume as cv
    sd, cfg = golden_state(golden_e2e, "gcnet")
    imL, _ = images(cfg["image_seed"], *cfg["hw"])
    m = load("gcnet", sd)
    timer = cv.LaunchTimer()
    cv.set_timer(timer)
    try:
        with torch.no_grad():
            got = m.layer2d(imL.cuda())
    finally:
        cv.set_timer(None)
    torch.cuda.synchronize()
    convs = sum(v["launches"] for k, v in timer.summary().items() if k.startswith("conv2d"))
    blocks = sum(v["launches"] for k, v in timer.summary().items() if k.startswith("basicblock2d"))
    assert (convs, blocks) in ((1, 8), (17, 0)), timer.summary().keys()
    assert blocks == (8 if cv.get_option("conv_precision") in ("f16x2", "f16") else 0)
    with torch.no_grad():
        want = OM.gcnet_features(OM.Net(sd), imL)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)   # NHWC: what the virtual volume stages from
    assert maxerr(got, want) <= 2e-5 * max(1.0, want.abs().max().item())


def test_psmnet_align_corners_switch(hip_lib, golden_e2e):
    """``PSMNet(maxdisp, align_corners=True)`` -- the PyTorch-0.3 meaning of the reference's
    ``F.upsample`` calls: every upsampling on the path switches convention
    (SPP branches, the trilinear upsampling fused into the heads), checked against the oracle's
    stages evaluated with the same convention."""
    import torch.nn.functional as TF
    from oracle import ops as OO
    from dsmnet_amd.models.psmnet.stackhourglass import PSMNet
    sd, cfg = golden_state(golden_e2e, "psmnet")
    imL, imR = images(cfg["image_seed"], *cfg["hw"])
    m = PSMNet(192, align_corners=True)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    orig = TF.interpolate

    def interp_true(x, *a, **k):                    # the oracle's SPP branches with align_corners=True
        if k.get("mode") == "bilinear":
            k["align_corners"] = True
        return orig(x, *a, **k)

    with torch.no_grad():
        _, got = m(imL.cuda(), imR.cuda())
        TF.interpolate = interp_true
        try:
            n = OM.Net(sd)
            fl, fr = OM.psmnet_features(n, imL), OM.psmnet_features(n, imR)
        finally:
            TF.interpolate = orig
        costs = OM.psmnet_trunk(n, OO.concat_volume(fl, fr, 48, True))
        want = [OO.soft_argmin(c, (192,) + tuple(cfg["hw"]), align_corners=True)
                for c in (costs[2], costs[1], costs[0])]
        _, default = load("psmnet", sd)(imL.cuda(), imR.cuda())
    for g, w in zip(got, want):
        assert maxerr(g, w) <= DISP_TOL
    assert maxerr(got[0], default[0]) > 10 * DISP_TOL          # the two conventions really differ
