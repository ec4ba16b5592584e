#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference, which never travels
to the GPU box).  The reference's Python-2 / PyTorch-0.3 source is read as
text from where it lies and executed by this container's torch with the shims
of SURVEY.md section 8c (oracle/reference_loader.py).  Inline code that has no
callable boundary (the two volume builds, the soft-argmin heads) is executed
*from the reference file's own lines* with two textual patches for Python 3
(``/4`` -> ``//4``, ``.cuda()`` removed); nothing of the reference is written
into the repository -- only inputs' seeds, outputs (sub-sampled where large)
and float64 checksums.

For every case the script also runs the oracle restatement (oracle/) on the
same inputs and REFUSES to write the fixture if the two disagree beyond the
stated tolerance, so a committed fixture certifies oracle == reference here.

Decision recorded in the fixtures' metadata: ``F.upsample`` in this container
(torch 2.10) resolves to ``align_corners=False``; the goldens therefore carry
align_corners=False semantics ("Version drift").

Usage:  python tests/golden/make_goldens.py [--only ops|blocks|e2e|train|lr]
"""
import argparse
import json
import os
import sys
import textwrap
import warnings

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import models as OM          # noqa: E402
from oracle import ops as OO             # noqa: E402
from oracle import reference_loader as RL  # noqa: E402

warnings.filterwarnings("ignore")
META = {"torch": torch.__version__, "align_corners": False,
        "reference": "sunshinnnn/DSMnet @ 2025-02-15"}


def seeded(seed, *shape, scale=1.0):
    """Seeded N(0,1) input, drawn in float64 and rounded (platform independent)."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).float()


def pack(t, stride=1):
    """Tensor -> dict(sample, sum, abssum); the sample is a strided sub-sample of the
    last two axes so large fields stay small in the repository."""
    t = t.detach()
    s = t[..., ::stride, ::stride] if stride > 1 else t
    return {"sample": s.numpy().astype(np.float32), "stride": np.int64(stride),
            "sum": np.float64(t.double().sum().item()),
            "abssum": np.float64(t.double().abs().sum().item()),
            "shape": np.asarray(t.shape, dtype=np.int64)}


def put(store, name, t, stride=1):
    for k, v in pack(t, stride).items():
        store["%s.%s" % (name, k)] = v


def check(name, ref, mine, tol):
    err = (ref - mine).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-30)
    status = "ok" if err <= tol * max(scale, 1.0) else "FAIL"
    print("  %-46s max|ref-oracle| = %.3e (scale %.3g) %s" % (name, err, scale, status))
    if status != "ok":
        raise SystemExit("oracle disagrees with the reference on %s" % name)
    return err


def ref_lines(relpath, first, last):
    """Lines first..last (1-based, inclusive) of a reference file, dedented."""
    with open(os.path.join(RL.REFERENCE_ROOT, relpath)) as fh:
        lines = fh.readlines()[first - 1:last]
    lines = [l for l in lines if l.strip() and not l.lstrip().startswith("#")]
    return textwrap.dedent("".join(lines))


class _Self(object):
    pass


# ----------------------------------------------------------------------------
# G1-G3: op-level goldens
# ----------------------------------------------------------------------------
def gen_ops(store):
    mods = RL.load()
    Corr1d = mods["util_conv"].Corr1d
    print("G1 corr1d (models/util_conv.py:56-86)")
    cases = []
    for (k, s, D) in ((1, 1, 41), (1, 1, 81), (3, 2, 41)):
        for shp, seed in (((2, 16, 9, 50), 11), ((1, 128, 64, 128), 12)):
            tag = "corr1d.k%d_s%d_D%d.%s" % (k, s, D, "x".join(map(str, shp)))
            fL = seeded(seed, *shp).requires_grad_(True)
            fR = seeded(seed + 100, *shp).requires_grad_(True)
            out = Corr1d(k, s, D)(fL, fR)
            cot = seeded(seed + 200, *out.shape)
            gL, gR = torch.autograd.grad(out, (fL, fR), cot)
            fL2, fR2 = fL.detach().clone().requires_grad_(True), fR.detach().clone().requires_grad_(True)
            mine = OO.corr1d(fL2, fR2, D, s, k)
            mL, mR = torch.autograd.grad(mine, (fL2, fR2), cot)
            check(tag + ".out", out, mine, 1e-6)
            check(tag + ".dL", gL, mL, 1e-5)
            check(tag + ".dR", gR, mR, 1e-5)
            stride = 1 if shp[0] == 2 else 4
            put(store, tag + ".out", out, stride)
            put(store, tag + ".dL", gL, stride)
            put(store, tag + ".dR", gR, stride)
            cases.append({"tag": tag, "k": k, "s": s, "D": D, "shape": list(shp), "seed": seed})
    META["corr1d_cases"] = cases

    print("G2 concat volume (models/gcnet.py:130-135, models/psmnet/stackhourglass.py:124-133)")
    vcases = []
    gc_src = ref_lines("models/gcnet.py", 130, 135).replace("Variable(", "(")
    psm_src = (ref_lines("models/psmnet/stackhourglass.py", 124, 133)
               .replace("self.maxdisp/4", "self.maxdisp//4").replace(".cuda()", "")
               .replace("Variable(", "("))
    for D in (48, 96):
        shp, seed = (1, 32, 16, 40), 21
        for mask_left in (False, True):
            tag = "volume.%s.D%d.%s" % ("psm" if mask_left else "gc", D, "x".join(map(str, shp)))
            fL = seeded(seed, *shp).requires_grad_(True)
            fR = seeded(seed + 100, *shp).requires_grad_(True)
            me = _Self()
            if mask_left:
                me.maxdisp = 4 * D
                ns = {"torch": torch, "self": me, "refimg_fea": fL, "targetimg_fea": fR}
                exec(psm_src, ns)
                vol = ns["cost"]
            else:
                me.D = D
                ns = {"torch": torch, "self": me, "fL": fL, "fR": fR}
                exec(gc_src, ns)
                vol = ns["xL"]
            cot = seeded(seed + 200, *vol.shape)
            gL, gR = torch.autograd.grad(vol, (fL, fR), cot)
            fL2, fR2 = fL.detach().clone().requires_grad_(True), fR.detach().clone().requires_grad_(True)
            mine = OO.concat_volume(fL2, fR2, D, mask_left)
            mL, mR = torch.autograd.grad(mine, (fL2, fR2), cot)
            check(tag + ".vol", vol, mine, 0.0)
            check(tag + ".dL", gL, mL, 1e-6)
            check(tag + ".dR", gR, mR, 1e-6)
            put(store, tag + ".vol", vol, 4)
            put(store, tag + ".dL", gL)
            put(store, tag + ".dR", gR)
            vcases.append({"tag": tag, "D": D, "mask_left": mask_left, "shape": list(shp), "seed": seed})
    META["volume_cases"] = vcases

    print("G3 soft-argmin (stackhourglass.py:152-166 + submodule.py:56-63, gcnet.py:104-111)")
    sub = mods["submodule"]
    psm_head = ref_lines("models/psmnet/stackhourglass.py", 163, 166)
    gc_head = ref_lines("models/gcnet.py", 104, 111).replace("Variable(", "(")
    gc_head = "\n".join(l for l in gc_head.splitlines() if not l.strip().startswith("if(mode"))
    gc_head = gc_head.replace("return out.unsqueeze(1)", "result = out.unsqueeze(1)")
    scases = []
    for kind, seed in (("normal", 31), ("onehot", 32)):
        # PSMNet form: x4 trilinear from (1,1,12,8,16) to (48,32,64)
        c = seeded(seed, 1, 1, 12, 8, 16, scale=3.0)
        if kind == "onehot":
            c = c * 0.01
            idx = torch.randint(0, 12, (8, 16), generator=torch.Generator().manual_seed(seed))
            c[0, 0].scatter_(0, idx.unsqueeze(0), 12.0)
        c.requires_grad_(True)
        left = torch.zeros(1, 3, 32, 64)
        me = _Self(); me.maxdisp = 48
        ns = {"F": F, "torch": torch, "self": me, "left": left, "cost3": c,
              "disparityregression": sub.disparityregression}
        exec(psm_head, ns)
        pred = ns["pred3"]
        cot = seeded(seed + 200, *pred.shape)
        (g,) = torch.autograd.grad(pred, c, cot)
        c2 = c.detach().clone().requires_grad_(True)
        mine = OO.soft_argmin(c2, (48, 32, 64))
        (mg,) = torch.autograd.grad(mine, c2, cot)
        tag = "softargmin.psm.%s" % kind
        check(tag + ".disp", pred, mine, 1e-6)
        check(tag + ".dcost", g, mg, 1e-5)
        put(store, tag + ".disp", pred); put(store, tag + ".dcost", g)
        scases.append({"tag": tag, "form": "psm", "kind": kind, "seed": seed,
                       "cost_shape": [1, 1, 12, 8, 16], "out_size": [48, 32, 64]})
        # GCNet form: softmax(-x) over 48 bins at (1,1,48,12,20)
        x = seeded(seed + 50, 1, 1, 48, 12, 20, scale=3.0)
        if kind == "onehot":
            x = x * 0.01
            idx = torch.randint(0, 48, (12, 20), generator=torch.Generator().manual_seed(seed + 1))
            x[0, 0].scatter_(0, idx.unsqueeze(0), -12.0)
        x.requires_grad_(True)
        me = _Self(); me.softmax = torch.nn.Softmax2d()
        ns = {"torch": torch, "self": me, "x37": x, "mode": "train"}
        exec(gc_head, ns)
        pred = ns["result"]
        cot = seeded(seed + 250, *pred.shape)
        (g,) = torch.autograd.grad(pred, x, cot)
        x2 = x.detach().clone().requires_grad_(True)
        mine = OO.soft_argmin(x2, None, negate=True).unsqueeze(1)
        (mg,) = torch.autograd.grad(mine, x2, cot)
        tag = "softargmin.gc.%s" % kind
        check(tag + ".disp", pred, mine, 1e-6)
        check(tag + ".dcost", g, mg, 1e-5)
        put(store, tag + ".disp", pred); put(store, tag + ".dcost", g)
        scases.append({"tag": tag, "form": "gc", "kind": kind, "seed": seed + 50,
                       "cost_shape": [1, 1, 48, 12, 20]})
    META["softargmin_cases"] = scases


# ----------------------------------------------------------------------------
# G4: 3-D blocks
# ----------------------------------------------------------------------------
def randomise_bn(sd, seed):
    """Non-trivial BN affine + running statistics so that eval-mode folding is
    actually exercised (the reference initialises them to 1/0/0/1)."""
    g = torch.Generator().manual_seed(seed)
    for k in sorted(sd):
        if k.endswith("running_mean"):
            p = k[: -len("running_mean")]
            c = sd[k].numel()
            sd[k] = (torch.randn(c, generator=g, dtype=torch.float64) * 0.3).float()
            sd[p + "running_var"] = (0.5 + torch.rand(c, generator=g, dtype=torch.float64)).float()
            sd[p + "weight"] = (0.5 + torch.rand(c, generator=g, dtype=torch.float64)).float()
            sd[p + "bias"] = (torch.randn(c, generator=g, dtype=torch.float64) * 0.2).float()
    return sd


def gen_blocks(store):
    mods = RL.load()
    print("G4 3-D blocks (submodule.py:16-19, stackhourglass.py:22-62,73-98; gcnet.py:32-101)")
    sd = randomise_bn(OM.init_state("psmnet", 0), 41)
    ref = mods["stackhourglass"].PSMNet(192)
    ref.load_state_dict(sd, strict=True)
    x64 = seeded(51, 1, 64, 12, 16, 24)
    x32 = seeded(52, 1, 32, 12, 16, 24)
    for training in (False, True):
        mode = "train" if training else "eval"
        ref.train(training)
        with torch.no_grad():
            ref.load_state_dict(sd, strict=True)     # reset running stats between modes
            r0 = ref.dres0(x64)
            o1, pre1, post1 = ref.dres2(x32, None, None)
            o2, pre2, post2 = ref.dres3(x32, pre1, post1)
            cl = ref.classif1(x32)
            sd2 = {k: v.clone() for k, v in sd.items()}
            n = OM.Net(sd2, training=training)
            m0 = F.relu(OM._cbn3(n, F.relu(OM._cbn3(n, x64, "dres0.0", 32)), "dres0.2", 32))
            a1, apre1, apost1 = OM._hourglass(n, x32, "dres2", None, None)
            a2, apre2, apost2 = OM._hourglass(n, x32, "dres3", apre1, apost1)
            t = F.relu(OM._cbn3(n, x32, "classif1.0", 32))
            mcl = n.conv(t, "classif1.2", 1, 3, 1, nd=3)
        for nm, r, m in (("dres0", r0, m0), ("hg1.out", o1, a1), ("hg1.pre", pre1, apre1),
                         ("hg1.post", post1, apost1), ("hg2.out", o2, a2), ("hg2.pre", pre2, apre2),
                         ("hg2.post", post2, apost2), ("classif1", cl, mcl)):
            tag = "block3d.psm.%s.%s" % (mode, nm)
            check(tag, r, m, 2e-5)
            put(store, tag, r)
    # GCNet: conv s2 + BN + ReLU (l21), deconv + BN + ReLU (l33), final deconv (l37)
    gsd = randomise_bn(OM.init_state("gcnet", 0), 42)
    gref = RL.fix_gcnet(mods["gcnet"].gcnet(192))
    gref.load_state_dict(gsd, strict=True)
    gref.eval()
    x128 = seeded(53, 1, 128, 6, 8, 12)
    with torch.no_grad():
        r21 = gref.layer3d.l21(x64)
        r33 = gref.layer3d.l33(x128)
        r37 = gref.layer3d.l37(x32)
        n = OM.Net({k: v.clone() for k, v in gsd.items()})
        m21 = F.relu(n.bn(n.conv(x64, "layer3d.l21.0", 64, 3, 2, bias=True, nd=3), "layer3d.l21.1"))
        m33 = F.relu(n.bn(n.conv(x128, "layer3d.l33.0", 64, 3, 2, bias=True, nd=3, transposed=True,
                                 out_pad=1), "layer3d.l33.1"))
        m37 = n.conv(x32, "layer3d.l37", 1, 3, 2, bias=True, nd=3, transposed=True, out_pad=1)
    for nm, r, m in (("l21", r21, m21), ("l33", r33, m33), ("l37", r37, m37)):
        tag = "block3d.gc.eval.%s" % nm
        check(tag, r, m, 2e-5)
        put(store, tag, r)
    META["blocks"] = {"psm_state_seed": 0, "psm_bn_seed": 41, "gc_state_seed": 0, "gc_bn_seed": 42,
                      "x64_seed": 51, "x32_seed": 52, "x128_seed": 53,
                      "x64_shape": [1, 64, 12, 16, 24], "x32_shape": [1, 32, 12, 16, 24],
                      "x128_shape": [1, 128, 6, 8, 12]}


# ----------------------------------------------------------------------------
# G5: end to end
# ----------------------------------------------------------------------------
def images(seed, h, w):
    """Synthetic pair, mirroring models/test_models_time.py:17-23: uniform [0,1)
    pixels then ImageNet normalisation (myTransforms/__init__.py:8)."""
    g = torch.Generator().manual_seed(seed)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    imL = torch.rand(1, 3, h, w, generator=g, dtype=torch.float64).float()
    # the right view is the left one shifted by a smooth disparity, so that the
    # cost volume has structure; exact content is irrelevant to parity.
    imR = torch.roll(imL, shifts=-7, dims=3)
    return (imL - mean) / std, (imR - mean) / std


def bn_stats(sd):
    return {k: v.clone() for k, v in sd.items() if "running_" in k or "num_batches" in k}


def gen_e2e(store):
    mods = RL.load()
    print("G5 end to end")
    # --- DispnetC 256x512 (BASELINE config #1: CPU plumbing case) -------------
    sd = OM.init_state("dispnetcorr", 0)
    ref = mods["dispnetcorr"].dispnetcorr(192)
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    imL, imR = images(61, 256, 512)
    with torch.no_grad():
        _, outs = ref(imL, imR)
        _, mine = OM.forward("dispnetcorr", sd, imL, imR)
    for i, (r, m) in enumerate(zip(outs, mine)):
        check("e2e.dispnetcorr.pr%d" % i, r, m, 1e-5)
        put(store, "e2e.dispnetcorr.pr%d" % i, r, 4 if i < 2 else 1)
    # --- iResNet 256x512 (the random epsilon of imwrap is seeded) -------------
    sd = OM.init_state("iresnet", 0)
    ref = mods["iresnet"].iresnet(192)
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    with torch.no_grad():
        torch.manual_seed(5); _, outs = ref(imL, imR)
        torch.manual_seed(5); _, mine = OM.forward("iresnet", sd, imL, imR)
    for i, (r, m) in enumerate(zip(outs, mine)):
        check("e2e.iresnet.out%d" % i, r, m, 1e-5)
        put(store, "e2e.iresnet.out%d" % i, r, 4 if r.shape[-1] >= 256 else 1)
    # --- GCNet 64x128, D=192, calibrated BN ----------------------------------
    sd = OM.init_state("gcnet", 0)
    gL, gR = images(62, 64, 128)
    OM.calibrate_bn("gcnet", sd, gL, gR)
    gc_factor = OM.calibrate_heads("gcnet", sd, gL, gR)
    store["e2e.gcnet.head_scale"] = np.float64(gc_factor)
    print("  gcnet head scale %.6e" % gc_factor)
    ref = RL.fix_gcnet(mods["gcnet"].gcnet(192))
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    with torch.no_grad():
        _, (oL,) = ref(gL, gR)
        mine = OM.forward("gcnet", sd, gL, gR)
    check("e2e.gcnet.disp", oL, mine, 1e-4)
    put(store, "e2e.gcnet.disp", oL)
    for k, v in bn_stats(sd).items():
        store["e2e.gcnet.bn." + k] = v.numpy()
    # --- PSMNet 256x512, D=192, calibrated BN --------------------------------
    # PSMNet.forward itself cannot run here (hard-coded .cuda(), Py2 '/');
    # the reference's own sub-modules are called around its volume-build and
    # head lines executed from the file text (SURVEY.md section 8c shim 4).
    sd = OM.init_state("psmnet", 0)
    pL, pR = images(63, 256, 512)
    OM.calibrate_bn("psmnet", sd, pL, pR)
    psm_factor = OM.calibrate_heads("psmnet", sd, pL, pR)
    store["e2e.psmnet.head_scale"] = np.float64(psm_factor)
    print("  psmnet head scale %.6e" % psm_factor)
    ref = mods["stackhourglass"].PSMNet(192)
    ref.load_state_dict(sd, strict=True)
    ref.eval()
    body = (ref_lines("models/psmnet/stackhourglass.py", 119, 166)
            .replace("self.maxdisp/4", "self.maxdisp//4").replace(".cuda()", "")
            .replace("Variable(", "("))
    ns = {"torch": torch, "F": F, "self": ref, "left": pL, "right": pR,
          "myadd_3d": mods["stackhourglass"].myadd_3d,
          "disparityregression": mods["submodule"].disparityregression}
    with torch.no_grad():
        exec(body, ns)
        preds = [ns["pred3"], ns["pred2"], ns["pred1"]]
        mine = OM.forward("psmnet", sd, pL, pR)
    for nm, r, m in zip(("pred3", "pred2", "pred1"), preds, mine):
        check("e2e.psmnet." + nm, r, m, 1e-4)
        put(store, "e2e.psmnet." + nm, r, 4)
    for k, v in bn_stats(sd).items():
        store["e2e.psmnet.bn." + k] = v.numpy()
    META["e2e"] = {"dispnetcorr": {"seed": 0, "image_seed": 61, "hw": [256, 512]},
                   "iresnet": {"seed": 0, "image_seed": 61, "hw": [256, 512], "torch_seed": 5},
                   "gcnet": {"seed": 0, "image_seed": 62, "hw": [64, 128]},
                   "psmnet": {"seed": 0, "image_seed": 63, "hw": [256, 512]}}



# ----------------------------------------------------------------------------
# G6: supervised objective and bookkeeping
# ----------------------------------------------------------------------------
def gen_train(store):
    """losses/loss.py does not parse under Python 3 (``print`` statements at :320-321) and
    imports SSIM/cv2-era modules, so the lines of the supervised path are executed from the
    file text as methods of a holder class -- the same approach as the inline volume builds."""
    from oracle import train as OT
    print("G6 supervised loss (losses/loss.py:36-44,326-338,379-392,407-422; stereo.py:95-113)")
    body = "".join("    " + l + "\n" for part in (
        ref_lines("losses/loss.py", 36, 44), ref_lines("losses/loss.py", 326, 338),
        ref_lines("losses/loss.py", 379, 392), ref_lines("losses/loss.py", 407, 422),
        ref_lines("stereo.py", 95, 101), ref_lines("stereo.py", 103, 113))
        for l in part.splitlines())
    ns = {"torch": torch, "F": F}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        exec("class Ref(object):\n" + body, ns)
    ref = ns["Ref"]()
    ref.lossfun = ref.loss_supervised

    cases = []
    for seed, shape, frac_valid, smooth in ((1, (2, 1, 24, 40), 0.7, True), (2, (1, 1, 17, 33), 1.0, False),
                                           (3, (1, 1, 8, 8), 0.0, True), (4, (3, 1, 16, 16), 0.3, True)):
        gt = seeded(seed, *shape).abs() * 40
        keep = torch.rand(shape, generator=torch.Generator().manual_seed(seed + 50)) < frac_valid
        gt = gt * keep
        pred = gt + seeded(seed + 100, *shape) * 3
        want = ref.loss_supervised(gt, pred, smooth, 1.0)
        mine = OT.loss_supervised(gt, pred, smooth, 1.0)
        if isinstance(want, int):
            assert want == 0 and mine == 0
            want = torch.zeros(())
        else:
            check("loss_supervised seed %d" % seed, want.double(), torch.as_tensor(mine), 1e-6)
        tag = "train.loss.%d" % seed
        store[tag] = np.float64(float(want))
        cases.append({"seed": seed, "shape": list(shape), "frac_valid": frac_valid, "smooth": smooth})
        d1, epe = ref.accuracy(pred, gt) if frac_valid > 0 else (torch.zeros(()), torch.zeros(()))
        if frac_valid > 0:
            od1, oepe = OT.accuracy(pred.numpy(), gt.numpy())
            check("accuracy D1 seed %d" % seed, torch.as_tensor(float(d1)), torch.as_tensor(float(od1)), 1e-6)
            check("accuracy EPE seed %d" % seed, torch.as_tensor(float(epe)), torch.as_tensor(float(oepe)), 1e-6)
        store["train.d1.%d" % seed] = np.float64(float(d1))
        store["train.epe.%d" % seed] = np.float64(float(epe))
    META["train_cases"] = cases

    # pyramid of 7 outputs (DispNetC / iResNet: scale_disps 0..6) under three schedules
    gt = seeded(7, 1, 1, 64, 128).abs() * 30
    disps = [seeded(70 + l, 1, 1, -(-64 // 2 ** l), -(-128 // 2 ** l)) * 2 + 20 for l in range(7)]
    scales = list(range(7))
    sched = []
    for epoch, maxepoch in ((0, 37), (10, 37), (36, 37), (37, 37), (5, 0)):
        ref.count_levels, ref.maxepoch_weight_adjust = 7, maxepoch
        ref.Weight_Adjust_levels(epoch)
        ow = OT.weight_adjust_levels(7, maxepoch, epoch)
        assert np.allclose(ref.weight_levels, ow, atol=1e-12), (ref.weight_levels, ow)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = ref.losses_pyramid0(gt, disps, scales, True)
        mine = OT.losses_pyramid0(ow, gt, disps, scales, True)
        check("losses_pyramid0 epoch %d/%d" % (epoch, maxepoch), want.double(), torch.as_tensor(mine), 1e-6)
        store["train.weights.%d_%d" % (epoch, maxepoch)] = np.asarray(ref.weight_levels, dtype=np.float64)
        store["train.pyramid.%d_%d" % (epoch, maxepoch)] = np.float64(float(want))
        sched.append([epoch, maxepoch])
    META["train_pyramid"] = {"gt_seed": 7, "disp_seed0": 70, "h": 64, "w": 128, "schedules": sched}

    class _Opt(object):
        def __init__(self):
            self.param_groups = [{"lr": -1.0}]
    lrs = []
    for epoch in (0, 49, 50, 69, 70, 131):
        o = _Opt()
        ref.lr_adjust(o, 50, 20, 1e-4, epoch)
        got = o.param_groups[0]["lr"]
        mine = OT.lr_adjust(1e-4, 50, 20, epoch)
        assert (got == -1.0 and mine is None) or abs(got - mine) < 1e-18, (epoch, got, mine)
        lrs.append(got)
    store["train.lr"] = np.asarray(lrs, dtype=np.float64)
    print("  lr schedule, weight schedule: oracle == reference")


def gen_lr(store):
    """gcnet_LR's two-sided volume build, models/gcnet.py:155-164, executed from the file text
    (``self.D`` is an int here: the Py2 ``maxdisparity/2`` of :143 is a float under Python 3)."""
    print("G2b right-referenced volume (models/gcnet.py:155-164)")
    src = ref_lines("models/gcnet.py", 155, 164).replace("Variable(", "(")
    cases = []
    for D, shp, seed in ((12, (1, 32, 6, 40), 31), (48, (2, 32, 5, 21), 32)):      # D > W in the second
        fL, fR = seeded(seed, *shp), seeded(seed + 100, *shp)
        me = _Self()
        me.D = D
        n, Fc, h, w = shp
        ns = {"torch": torch, "self": me, "fL": fL, "fR": fR, "n": n, "F": Fc, "h": h, "w": w}
        # D > W: for i >= w the reference's slices (`i:` / `:-i` on both sides) are all empty, so the
        # planes past the width keep their zeros in the shifted half -- pinned as well
        exec(src, ns)
        tag = "volume_lr.D%d.%s" % (D, "x".join(map(str, shp)))
        check(tag + ".xL", ns["xL"], OO.concat_volume(fL, fR, D, False), 0.0)
        check(tag + ".xR", ns["xR"], OO.concat_volume_right(fL, fR, D), 0.0)
        put(store, tag + ".xR", ns["xR"], 2)
        cases.append({"tag": tag, "D": D, "shape": list(shp), "seed": seed})
    META["volume_lr_cases"] = cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["ops", "blocks", "e2e", "train", "lr"], default=None)
    args = ap.parse_args()
    if not RL.available():
        raise SystemExit("needs the reference tree at %s (build container only)" % RL.REFERENCE_ROOT)
    torch.set_num_threads(os.cpu_count() or 1)
    for part, fn in (("ops", gen_ops), ("blocks", gen_blocks), ("e2e", gen_e2e), ("train", gen_train),
                     ("lr", gen_lr)):
        if args.only and args.only != part:
            continue
        store = {}
        fn(store)
        store["meta"] = np.frombuffer(json.dumps(META, sort_keys=True).encode(), dtype=np.uint8)
        path = os.path.join(HERE, "golden_%s.npz" % part)
        np.savez_compressed(path, **store)
        print("wrote %s (%.1f KiB, %d arrays)" % (path, os.path.getsize(path) / 1024.0, len(store)))


if __name__ == "__main__":
    main()
