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
        with torch.no_grad():
            getattr(model, "classif%d" % i)[2].weight.mul_(1e-3)
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


def test_graphed_train_step_follows_the_eager_trajectory(hip_lib):
    """hipGraph replay of a whole PSMNet training step: same losses as eager steps from the same
    initial state (float atomics in two backward kernels: agreement to 1e-3, not bit-exact)."""
    import copy
    from dsmnet_amd import train
    from dsmnet_amd.graphs import GraphedTrainStep
    from dsmnet_amd.models import model_create_by_name
    torch.manual_seed(0)
    m1 = model_create_by_name("psmnet", 192).cuda()
    for i in (1, 2, 3):
        with torch.no_grad():
            getattr(m1, "classif%d" % i)[2].weight.mul_(1e-3)
    m2 = copy.deepcopy(m1)
    batches = [_batch(1, 256, 512, 6, s) for s in (5, 6, 7)]
    lf1, lf2 = train.losses("supervised", 1, 0), train.losses("supervised", 1, 0)
    lf1.Weight_Adjust_levels(0); lf2.Weight_Adjust_levels(0)
    o1 = torch.optim.Adam(m1.parameters(), lr=1e-4)
    eager = [train.train_step(m1, o1, lf1, b)[0] for b in batches + batches]
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-4, capturable=True)
    state = copy.deepcopy(m2.state_dict())
    step = GraphedTrainStep(m2, o2, lf2, batches[0], warmup=2)
    # the warm-up and the capture ran real steps: rewind model and optimizer, then replay
    m2.load_state_dict(state)
    for st in o2.state.values():              # in place: the graph holds these very tensors
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    graphed = [float(step(b)[0]) for b in batches + batches]
    for a, b in zip(eager, graphed):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (eager, graphed)
    assert graphed[-1] < graphed[0]
    with pytest.raises(ValueError):
        GraphedTrainStep(m2, torch.optim.Adam(m2.parameters(), lr=1e-4), lf2, batches[0])


def test_graphed_train_step_in_its_multi_rank_form(hip_lib):
    """The form a step takes with several ranks -- forward + backward captured, every gradient a view
    of one flat buffer, the (here: one-rank, no-op) all-reduce and the fused optimizer step outside
    the graph -- follows the eager trajectory too, and a batch without ground truth skips the update."""
    import copy
    from dsmnet_amd import train
    from dsmnet_amd.graphs import GraphedTrainStep
    from dsmnet_amd.models import model_create_by_name
    torch.manual_seed(0)
    m1 = model_create_by_name("psmnet", 192).cuda()
    for i in (1, 2, 3):
        with torch.no_grad():
            getattr(m1, "classif%d" % i)[2].weight.mul_(1e-3)
    m2 = copy.deepcopy(m1)
    batches = [_batch(1, 256, 512, 6, s) for s in (5, 6, 7)]
    lf1, lf2 = train.losses("supervised", 1, 0), train.losses("supervised", 1, 0)
    lf1.Weight_Adjust_levels(0); lf2.Weight_Adjust_levels(0)
    o1 = torch.optim.Adam(m1.parameters(), lr=1e-4)
    eager = [train.train_step(m1, o1, lf1, b)[0] for b in batches + batches]
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-4, capturable=True, fused=True)
    state = copy.deepcopy(m2.state_dict())
    step = GraphedTrainStep(m2, o2, lf2, batches[0], warmup=2, flat_gradients=True)
    assert step.flatgrads is not None and all(p.grad.data_ptr() >= step.flatgrads.flat.data_ptr()
                                              for p in m2.parameters())
    m2.load_state_dict(state)                 # the warm-up and the capture ran real steps: rewind
    for st in o2.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    graphed = [float(step(b)[0]) for b in batches + batches]
    for a, b in zip(eager, graphed):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (eager, graphed)
    assert graphed[-1] < graphed[0]
    before = [p.detach().clone() for p in m2.parameters()]
    empty = batches[0].clone()
    empty[:, 6:7] = 0                          # no ground-truth pixel: zero gradients, no update
    step(empty)
    assert float(step.flatgrads.extra) == 0.0
    assert all(torch.equal(a, b) for a, b in zip(before, m2.parameters()))
