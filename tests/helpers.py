"""Helpers shared by the test modules."""
import torch


def seeded(seed, *shape, scale=1.0):
    """Same seeded N(0,1) draw as tests/golden/make_goldens.py."""
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale).float()


def maxerr(a, b):
    return (a.detach().double().cpu() - b.detach().double().cpu()).abs().max().item()
