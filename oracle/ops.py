"""Oracle restatements of the four hot-path op families (CPU, torch, fp32/fp64).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

All functions are pure (inputs untouched, fresh outputs) and differentiable by
torch autograd, so the same restatement also yields the reference gradients
(the reference gets its backward from autograd through the identical slice
assignments).
"""
import torch
import torch.nn.functional as F


def corr1d(fL, fR, D, stride=1, kernel_size=1):
    """1-D left/right correlation.

    Follows ``models/util_conv.py:71-86`` (``Corr1d.forward``) with the default
    similarity ``(fL*fR).sum(dim=1)`` (``:68-69``): plane ``i`` holds the
    un-normalised channel dot product of ``fL[..., x]`` and
    ``fR[..., x - i*stride]`` for ``x >= i*stride`` and zero elsewhere; planes
    ``i >= w`` stay zero (the ``break`` at ``:79``); ``kernel_size > 1`` box
    filters every plane with ``AvgPool2d(k, 1, k//2)`` (``:82-85``), i.e. zero
    padding that counts in the divisor.
    """
    B, C, H, W = fL.shape
    planes = []
    for i in range(D):
        shift = i * stride
        plane = fL.new_zeros(B, H, W)
        if i == 0:
            plane = (fL * fR).sum(dim=1)
        elif i < W and shift < W:
            prod = (fL[..., shift:] * fR[..., : W - shift]).sum(dim=1)
            plane = F.pad(prod, (shift, 0))
        planes.append(plane)
    out = torch.stack(planes, dim=1)
    if kernel_size > 1:
        if kernel_size % 2 != 1:
            raise AssertionError("kernel_size must be odd")  # util_conv.py:83
        out = F.avg_pool2d(out, kernel_size, stride=1, padding=kernel_size // 2)
    return out


def concat_volume(fL, fR, D, mask_left):
    """Concatenation cost volume ``(B, 2C, D, H, W)``.

    ``mask_left=False`` is GCNet's build, ``models/gcnet.py:130-135``: the left
    features fill every ``x`` of every disparity plane, the right features are
    shifted by ``d`` and zero for ``x < d``.

    ``mask_left=True`` is PSMNet's build,
    ``models/psmnet/stackhourglass.py:124-133``: both halves are zero for
    ``x < d``.  The reference allocates a ``FloatTensor`` (fp32) there whatever
    the feature dtype; the oracle keeps the feature dtype (fp32 in every test).
    """
    B, C, H, W = fL.shape
    vol = fL.new_zeros(B, 2 * C, D, H, W)
    for d in range(D):
        if d >= W:
            if not mask_left:
                vol[:, :C, d] = fL
            continue
        if mask_left:
            vol[:, :C, d, :, d:] = fL[..., d:]
        else:
            vol[:, :C, d] = fL
        vol[:, C:, d, :, d:] = fR[..., : W - d]
    return vol


def concat_volume_right(fL, fR, D):
    """The right-referenced volume of ``gcnet_LR``, ``models/gcnet.py:155-164`` (``xR``): the
    RIGHT features fill every ``x`` of every disparity plane, the LEFT features are shifted the
    other way -- ``xR[:, F:, d, :, x] = fL[:, :, :, x + d]`` for ``x < W - d``, zero beyond."""
    B, C, H, W = fL.shape
    vol = fL.new_zeros(B, 2 * C, D, H, W)
    for d in range(D):
        vol[:, :C, d] = fR
        if d < W:
            vol[:, C:, d, :, : W - d] = fL[..., d:]
    return vol


def crop_add(a, b):
    """Skip-add with crop to the common ``(d, h, w)``.

    ``models/util_fun.py:41-50`` (``myAdd3d``) and
    ``models/psmnet/stackhourglass.py:10-20`` (``myadd_3d``).
    """
    d = min(a.shape[2], b.shape[2])
    h = min(a.shape[3], b.shape[3])
    w = min(a.shape[4], b.shape[4])
    return a[:, :, :d, :h, :w] + b[:, :, :d, :h, :w]


def soft_argmin(cost, out_size=None, negate=False, align_corners=False):
    """Soft-argmin disparity regression.

    PSMNet form (``out_size=(D, H, W)``, ``negate=False``),
    ``models/psmnet/stackhourglass.py:152-166`` +
    ``models/psmnet/submodule.py:60-63``: trilinear upsample of the
    ``(B, 1, Dc, Hc, Wc)`` cost, squeeze, softmax over the disparity axis
    (``F.softmax`` without ``dim`` on a 4-D tensor picks dim 1), expectation
    against ``arange(D)``.  Returns ``(B, H, W)``.

    GCNet form (``out_size=None``, ``negate=True``), ``models/gcnet.py:104-111``:
    ``Softmax2d(-x)`` on the ``(B, D, H, W)`` cost, same expectation.  Returns
    ``(B, H, W)``; the caller adds the channel axis (``:111``).

    ``align_corners=False`` is what ``F.upsample`` resolves to on the torch that
    executes the oracle ("Version drift").
    """
    if out_size is not None:
        if cost.dim() == 4:
            cost = cost.unsqueeze(1)
        cost = F.interpolate(cost, size=tuple(out_size), mode="trilinear",
                             align_corners=align_corners).squeeze(1)
    elif cost.dim() == 5:
        cost = cost.squeeze(1)
    if negate:
        cost = -cost
    prob = torch.softmax(cost, dim=1)
    disp = torch.arange(prob.shape[1], dtype=prob.dtype, device=prob.device)
    return prob.permute(0, 2, 3, 1).matmul(disp)


def conv3d_block(x, weight, bias=None, stride=1, transposed=False, bn=None,
                 residual=None, relu=False, training=False, eps=1e-5):
    """One 3-D regularisation layer: (de)conv k=3 -> BN -> (+skip, cropped) -> ReLU.

    Covers ``convbn_3d`` (``models/psmnet/submodule.py:16-19``), the
    ``ConvTranspose3d(k=3, s=2, p=1, op=1)`` + ``BatchNorm3d`` pairs of
    ``hourglass`` (``stackhourglass.py:38-42``), and ``conv3d_bn`` /
    ``deconv3d_bn`` (``models/util_conv.py:150-179``; the latter's
    ``BatchNorm2d`` is treated as per-channel BN over ``(N, D, H, W)``,
    SURVEY.md section 8c shim 3).

    ``bn`` is ``(gamma, beta, running_mean, running_var)`` or ``None``.
    """
    if transposed:
        y = F.conv_transpose3d(x, weight, bias, stride=stride, padding=1,
                               output_padding=stride - 1)
    else:
        y = F.conv3d(x, weight, bias, stride=stride, padding=1)
    if bn is not None:
        gamma, beta, mean, var = bn
        y = F.batch_norm(y, mean, var, gamma, beta, training=training, eps=eps)
    if residual is not None:
        y = crop_add(y, residual)
    if relu:
        y = F.relu(y)
    return y
