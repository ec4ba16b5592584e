"""Supervised objective, schedules and metrics on CPU: the oracle
(oracle/train.py) and the product host logic (dsmnet_amd/train.py) against the golden values
computed by the reference's own lines (tests/golden/golden_train.npz)."""
import numpy as np
import pytest
import torch

from dsmnet_amd import train as T
from oracle import train as OT
from tests.helpers import seeded


@pytest.fixture(scope="module")
def gold(golden_train):
    return golden_train


def _case(c):
    seed, shape = c["seed"], tuple(c["shape"])
    gt = seeded(seed, *shape).abs() * 40
    keep = torch.rand(shape, generator=torch.Generator().manual_seed(seed + 50)) < c["frac_valid"]
    gt = gt * keep
    return gt, gt + seeded(seed + 100, *shape) * 3


def test_loss_supervised_and_accuracy_match_reference(gold):
    lf = T.losses("supervised", 1, 0)
    for c in gold.meta["train_cases"]:
        gt, pred = _case(c)
        want = float(gold["train.loss.%d" % c["seed"]])
        assert abs(float(OT.loss_supervised(gt, pred, c["smooth"])) - want) <= 1e-6 * max(1, want)
        assert abs(float(lf.loss_supervised(gt, pred, c["smooth"])) - want) <= 2e-6 * max(1, want)
        if c["frac_valid"] > 0:
            for fn in (lambda: OT.accuracy(pred.numpy(), gt.numpy()), lambda: T.accuracy(pred, gt)):
                d1, epe = fn()
                assert abs(float(d1) - float(gold["train.d1.%d" % c["seed"]])) <= 1e-4
                assert abs(float(epe) - float(gold["train.epe.%d" % c["seed"]])) <= 1e-5
        else:
            assert lf.loss_supervised(gt, pred, True) == 0 and OT.loss_supervised(gt, pred, True) == 0


def test_pyramid_loss_and_weight_schedule_match_reference(gold):
    m = gold.meta["train_pyramid"]
    gt = seeded(m["gt_seed"], 1, 1, m["h"], m["w"]).abs() * 30
    disps = [seeded(m["disp_seed0"] + l, 1, 1, -(-m["h"] // 2 ** l), -(-m["w"] // 2 ** l)) * 2 + 20
             for l in range(7)]
    for epoch, maxepoch in m["schedules"]:
        want_w = gold["train.weights.%d_%d" % (epoch, maxepoch)]
        want = float(gold["train.pyramid.%d_%d" % (epoch, maxepoch)])
        assert np.allclose(OT.weight_adjust_levels(7, maxepoch, epoch), want_w, atol=1e-12)
        lf = T.losses("supervised", count_levels=7, maxepoch_weight_adjust=maxepoch)
        lf.Weight_Adjust_levels(epoch)
        assert np.allclose(lf.weight_levels, want_w, atol=1e-12)
        got = lf({"disp_gt": gt, "disps": disps, "scale_disps": list(range(7)), "flag_smooth": True})
        assert abs(float(got) - want) <= 2e-6 * want
        assert abs(float(OT.losses_pyramid0(want_w, gt, disps, list(range(7)), True)) - want) <= 1e-6 * want


def test_initial_weights_and_psmnet_outputs():
    lf = T.losses("supervised", count_levels=7, maxepoch_weight_adjust=37)
    assert lf.weight_levels == [0, 0, 0, 0, 0, 0, 1]              # loss.py:360-362, before any epoch
    lf = T.losses("supervised-mask", count_levels=1, maxepoch_weight_adjust=0)
    assert lf.flag_mask and lf.weight_levels == [1]
    gt = seeded(1, 2, 1, 8, 12).abs() + 1
    preds = [gt[:, 0] + 0.5, gt[:, 0] - 0.25, gt[:, 0]]            # (B,H,W), as PSMNet returns them
    got = lf({"disp_gt": gt, "disps": preds, "scale_disps": [0, 0, 0], "flag_smooth": False})
    assert abs(float(got) - 0.75) < 1e-6
    with pytest.raises(NotImplementedError):
        T.losses("depthmono")


def test_lr_adjust_matches_reference(gold):
    want = gold["train.lr"]
    p = torch.nn.Parameter(torch.zeros(1))
    for epoch, w in zip((0, 49, 50, 69, 70, 131), want):
        opt = torch.optim.Adam([p], lr=123.0)
        T.lr_adjust(opt, 50, 20, 1e-4, epoch)
        o = OT.lr_adjust(1e-4, 50, 20, epoch)
        if w == -1.0:                                  # the generator's "untouched" sentinel
            assert o is None and opt.param_groups[0]["lr"] == 123.0
        else:
            assert abs(opt.param_groups[0]["lr"] - w) < 1e-18 and abs(o - w) < 1e-18


def test_loss_is_differentiable_and_masks_gradients():
    lf = T.losses("supervised", 1, 0)
    gt = torch.tensor([[[[0.0, 5.0], [7.0, 0.0]]]])
    pred = torch.tensor([[[[1.0, 4.0], [9.0, 3.0]]]], requires_grad=True)
    lf.loss_supervised(gt, pred, False).backward()
    assert torch.equal(pred.grad, torch.tensor([[[[0.0, -0.5], [0.5, 0.0]]]]))


def test_average_meter():
    m = T.AverageMeter()
    m.update(2.0, 2)
    m.update(5.0, 1)
    assert m.val == 5.0 and m.count == 3 and abs(m.avg - 3.0) < 1e-12


def test_capturable_loss_equals_reference_form(gold):
    lf, lc = T.losses("supervised", 1, 0), T.losses("supervised", 1, 0)
    lc.capturable = True
    for c in gold.meta["train_cases"]:
        gt, pred = _case(c)
        a, b = lf.loss_supervised(gt, pred, c["smooth"]), lc.loss_supervised(gt, pred, c["smooth"])
        assert abs(float(a) - float(b)) <= 2e-6 * max(1.0, float(a))
        assert torch.is_tensor(b)                      # a zero tensor where the reference returns int 0
