"""Synthetic-checkpoint conditioning for benchmarks and smoke runs.

The reference ships no weights (the README links are offline); benchmarks therefore use
the reference's random initialisation (models/psmnet/stackhourglass.py:100-112).  Raw, it
is numerically degenerate -- BatchNorm running statistics at (0, 1) blow activations up to
~1e7 and the soft-argmin logits to std ~3e3 (tests/tools/diag_psmnet_error.py)
-- so, as the goldens do, one train-mode pass populates the BN statistics and the last
(linear) classifier layers are scaled to trained-network logit magnitudes.  Neither step
changes the work per forward pass.
"""
import torch


def calibrate_batchnorm(model, left, right, passes=1):
    """Train-mode forward passes (batch statistics; running stats updated), no autograd.  The
    towers' convolutions run on this library's kernels whatever their size (no MIOpen solver
    search at start-up: ``blocks2d.own_kernels_only``)."""
    from .blocks2d import own_kernels_only
    was_training = model.training
    model.train()
    with torch.no_grad(), own_kernels_only():
        for _ in range(passes):
            model(left, right)
    model.train(was_training)
    return model


def calibrate_psmnet_heads(model, left, right, target_std=2.0):
    """Scale classif{1,2,3}.2.weight by one factor so that std(cost3) == target_std."""
    from . import costvolume as cv
    was_training = model.training
    model.eval()
    with torch.no_grad():
        fl, fr = model.features(left, right)
        cost3 = model.regularise(cv.concat_volume(fl, fr, model.maxdisp // 4, True))[2]
        factor = target_std / float(cost3.std())
        for i in (1, 2, 3):
            getattr(model, "classif%d" % i)[2].weight.mul_(factor)
    model.train(was_training)
    return factor
