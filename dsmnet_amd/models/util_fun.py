"""Crop-to-common-size helpers with the reference's names (models/util_fun.py).  On the
fused 3-D path ``myAdd3d`` happens inside the convolution epilogue
(``block(x, residual=skip)``); these remain for direct callers and the 2-D decoders."""
import torch


def _common(tensors, first_axis):
    return [min(t.shape[a] for t in tensors) for a in range(first_axis, tensors[0].dim())]


def _crop(t, sizes, first_axis):
    idx = [slice(None)] * first_axis + [slice(0, s) for s in sizes]
    return t[tuple(idx)]


def myCat2d(*seq):
    assert seq[0].dim() == 4
    sizes = _common(seq, 2)
    return torch.cat([_crop(t, sizes, 2) for t in seq], dim=1)


def myCat3d(*seq):
    assert seq[0].dim() == 5
    sizes = _common(seq, 2)
    return torch.cat([_crop(t, sizes, 2) for t in seq], dim=1)


def myAdd2d(tensor1, tensor2):
    assert tensor1.dim() == 4
    sizes = _common((tensor1, tensor2), 2)
    return _crop(tensor1, sizes, 2) + _crop(tensor2, sizes, 2)


def myAdd3d(tensor1, tensor2):
    assert tensor1.dim() == 5
    sizes = _common((tensor1, tensor2), 2)
    return _crop(tensor1, sizes, 2) + _crop(tensor2, sizes, 2)


def myMax2d(disp_low, disp_high):
    assert disp_low.dim() == 4
    bn, c, h, w = disp_high.shape
    low = disp_low[:bn, :c, :h, :w]
    mask = disp_high < low
    disp_high[mask] = low[mask]
    return disp_high
