import sys, os
sys.path.insert(0, ".")
import torch
from oracle import models as OM, ops as OO
from tests.golden.make_goldens import randomise_bn
from tests.helpers import seeded, maxerr
from dsmnet_amd import costvolume as cv
from dsmnet_amd.models import model_create_by_name
sd = randomise_bn(OM.init_state("psmnet", 0), 41)
OM.apply_head_scale("psmnet", sd, 0.05)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 71
fl, fr = seeded(S, 1, 32, 16, 40), seeded(S + 1, 1, 32, 16, 40)
size = (32, 64, 160)
osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
ofl, ofr = fl.clone().requires_grad_(True), fr.clone().requires_grad_(True)
n = OM.Net(osd, training=True)
costs = OM.psmnet_trunk(n, OO.concat_volume(ofl, ofr, 8, True))
oloss = sum(OO.soft_argmin(c, size).mean() for c in costs)
keys = ["dres0.0.0.weight", "dres0.0.1.weight", "dres1.2.0.weight", "dres2.conv1.0.0.weight", "dres2.conv5.0.weight", "dres3.conv6.0.weight", "dres4.conv2.1.bias", "classif1.2.weight", "classif3.0.0.weight"]
ogr = torch.autograd.grad(oloss, [osd[k] for k in keys] + [ofl, ofr])
m = model_create_by_name("psmnet", 192); m.load_state_dict(sd, strict=True); m = m.cuda().train()
gfl, gfr = fl.cuda().requires_grad_(True), fr.cuda().requires_grad_(True)
gc = m.regularise(cv.concat_volume(gfl, gfr, 8, True))
loss = sum(cv.soft_argmin(c, size).mean() for c in gc)
print(os.environ.get("DSM_CONV_PRECISION"), "seed", S, "loss", loss.item(), oloss.item())
params = dict(m.named_parameters())
ggr = torch.autograd.grad(loss, [params[k] for k in keys] + [gfl, gfr])
print("   " + " ".join("%s %.1e" % (k.split(".")[0] + "." + k.split(".")[-1][0], maxerr(g, r) / max(r.abs().max().item(), 1e-6)) for k, g, r in zip(keys + ["fL", "fR"], ggr, ogr)))
for i, (a, b) in enumerate(zip(gc, costs)):
    print("  cost%d fwd rel err %.2e" % (i + 1, maxerr(a, b) / b.abs().max().item()))
