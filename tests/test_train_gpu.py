"""Supervised training step on the GPU (BASELINE config #5 in miniature):
``train.train_step`` / ``validate_step`` through the HIP forward and backward kernels."""
import pytest
import torch

from oracle import train as OT

pytestmark = pytest.mark.gpu


def _batch(B, H, W, shift, seed):
    g = torch.Generator().manual_seed(seed)
    left = torch.rand(B, 3, H, W, generator=g)
    right = torch.roll(left, -shift, dims=3)
    disp = torch.full((B, 1, H, W), float(shift))
    disp[:, :, :, :shift] = 0                               # no ground truth where the view wraps
    return torch.cat([left, right, disp], 1).cuda()


def test_dispnetc_train_step_pyramid_loss_vs_oracle(hip_lib):
    """7-output pyramid (scale_disps 0..6): the loss train_step reports equals the oracle's
    objective evaluated on the model's own outputs; Adam steps reduce it."""
    from dsmnet_amd import train
    from dsmnet_amd.models import model_create_by_name
    torch.manual_seed(0)
    model = model_create_by_name("dispnetcorr", 192).cuda()
    lossfun = train.losses("supervised", model.count_levels, maxepoch_weight_adjust=37)
    lossfun.Weight_Adjust_levels(10)
    opt = train.make_optimizer(model, lr=1e-4)
    batch = _batch(2, 256, 512, 6, 3)
    model.train()
    scales, disps = model(batch[:, :3], batch[:, 3:6])
    want = OT.losses_pyramid0(lossfun.weight_levels, batch[:, 6:7].cpu(),
                              [d.detach().cpu() for d in disps], scales, True)
    got = lossfun({"disp_gt": batch[:, 6:7], "disps": disps, "scale_disps": scales, "flag_smooth": True})
    assert abs(float(got.detach()) - float(want)) <= 1e-4 * max(1.0, float(want))
    first = [train.train_step(model, opt, lossfun, batch) for _ in range(4)]
    assert all(torch.isfinite(torch.tensor(f)).all() for f in first)
    assert first[-1][0] < first[0][0]
    val = train.validate_step(model, lossfun, batch)
    assert all(v == v for v in val) and not model.training


def test_psmnet_train_step_and_lr_schedule(hip_lib):
    from dsmnet_amd import train
    from dsmnet_amd.models import model_create_by_name
    torch.manual_seed(0)
    model = model_create_by_name("psmnet", 192).cuda()
    for i in (1, 2, 3):
        getattr(model, "classif%d" % i)[2].weight.data.mul_(1e-3)
    lossfun = train.losses("supervised", model.count_levels, 0)
    lossfun.Weight_Adjust_levels(0)
    opt = train.make_optimizer(model, lr=1e-3)
    batch = _batch(1, 256, 512, 6, 5)
    l0 = train.train_step(model, opt, lossfun, batch)[0]
    train.lr_adjust(opt, 0, 20, 1e-3, 0)
    assert opt.param_groups[0]["lr"] == 5e-4
    for _ in range(2):
        l1, d1, epe = train.train_step(model, opt, lossfun, batch)
    assert l1 < l0 and 0 <= d1 <= 100 and epe >= 0
    # no ground truth at all: the reference's loss is the integer 0 and the step is skipped
    empty = batch.clone()
    empty[:, 6:7] = 0
    before = [p.detach().clone() for p in model.parameters()][:3]
    assert train.train_step(model, opt, lossfun, empty)[0] == 0.0
    assert all(torch.equal(a, b) for a, b in zip(before, list(model.parameters())[:3]))
