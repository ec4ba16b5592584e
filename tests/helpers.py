"""Helpers shared by the test modules."""
import torch


def seeded(seed, *shape, scale=1.0):
    """Same seeded N(0,1) draw as tests/golden/make_goldens.py."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).float()


def maxerr(a, b):
    return (a.detach().double().cpu() - b.detach().double().cpu()).abs().max().item()


def golden_state(golden_e2e, name):
    """The synthetic checkpoint of an end-to-end golden case: oracle init from the recorded
    seed + the calibrated BN statistics and head scale stored in the fixture."""
    from oracle import models as OM
    cfg = golden_e2e.meta["e2e"][name]
    sd = OM.init_state(name, cfg["seed"])
    for k, v in golden_e2e.arrays("e2e.%s.bn." % name).items():
        sd[k] = torch.from_numpy(v.copy())
    key = "e2e.%s.head_scale" % name
    if key in golden_e2e.z.files:
        OM.apply_head_scale(name, sd, float(golden_e2e.z[key]))
    return sd, cfg
