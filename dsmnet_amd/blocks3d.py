"""3-D regularisation blocks: reference-compatible module trees, one HIP launch each.

The reference builds its 3-D trunk from ``nn.Sequential(nn.Conv3d, nn.BatchNorm3d[, ReLU])``
(``convbn_3d`` -- models/psmnet/submodule.py:16-19; ``conv3d_bn`` / ``deconv3d_bn`` --
models/util_conv.py:150-179).  The classes here keep exactly those children under
exactly those indices -- so reference checkpoints load with ``load_state_dict``
unchanged -- but execute conv + folded BN + cropped skip-add + ReLU as a single
``dsm_conv3d_fwd`` call.  The stock ``nn.Conv3d`` / ``nn.BatchNorm3d`` children are
parameter holders only; their own ``forward`` is never used on this path.
"""
import torch
import torch.nn as nn

from . import costvolume as cv

RELU_NONE, RELU_AFTER_ADD, RELU_BEFORE_ADD = 0, 1, 2
_FUSED_TRAIN_BN = __import__("os").environ.get("DSM_TRAIN_BN", "fused") != "stock"
# test hook: called with (block output, relu mode) of every fused train-mode block that ends in a
# ReLU (the sign pattern of the ReLU's input is `out > 0` for mode 1)
_TRAIN_RELU_HOOK = [None]


_EPOCH = [0]


def invalidate_folded_caches():
    """Drop every cached packed weight / folded BN affine (3-D blocks, 2-D blocks, SPP head):
    they are re-made at the next forward.  The caches notice ordinary updates by themselves
    (optimizer steps, ``load_state_dict``, ``with torch.no_grad(): w.mul_(...)`` -- all bump the
    tensor's ``_version``); an in-place edit THROUGH ``.data`` (``w.data.mul_()``) does not, and
    needs this call afterwards.  Exported as ``dsmnet_amd.refold()``."""
    _EPOCH[0] += 1


def _bump_running_stats(bn):
    """The BN kernels update the running statistics through raw pointers: bump the two tensors'
    version counters by hand, so that exactly the folds made from THIS layer's statistics are re-made
    at the next eval forward (a global invalidation would re-pack every layer after every step)."""
    if bn.track_running_stats and bn.running_mean is not None:
        torch.autograd.graph.increment_version(bn.running_mean)
        torch.autograd.graph.increment_version(bn.running_var)


def _versions(*tensors):
    return tuple((t.data_ptr(), t._version) for t in tensors if t is not None) + (_EPOCH[0],)


class _Folded(object):
    """Packed weights + folded BN affine of one (conv, bn) pair, cached until a
    parameter or running statistic changes (``_version`` / storage address)."""

    def __init__(self):
        self.key = None
        self.packed = self.scale = self.shift = None

    def get(self, conv, bn):
        transposed = isinstance(conv, nn.ConvTranspose3d)
        srcs = [conv.weight, conv.bias]
        if bn is not None:
            srcs += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        key = _versions(*srcs)
        if key != self.key:
            with torch.no_grad():
                self.packed = cv.pack_conv3d_weight(conv.weight, transposed)
                cout = conv.out_channels
                if bn is not None:
                    inv = torch.rsqrt(bn.running_var + bn.eps)
                    scale = bn.weight * inv if bn.weight is not None else inv
                    shift = -bn.running_mean * scale
                    if bn.bias is not None:
                        shift = shift + bn.bias
                    if conv.bias is not None:
                        shift = shift + conv.bias * scale
                    self.scale, self.shift = scale.contiguous(), shift.contiguous()
                elif conv.bias is not None:
                    self.scale = torch.ones(cout, device=conv.weight.device)
                    self.shift = conv.bias.detach().clone()
                else:
                    self.scale = self.shift = None
            self.key = key
        return self.packed, self.scale, self.shift


def _check_conv(conv):
    k = conv.kernel_size
    if tuple(k) != (3, 3, 3) or tuple(conv.padding) != (1, 1, 1) or tuple(conv.dilation) != (1, 1, 1):
        raise ValueError("the gfx950 3-D block supports kernel_size=3, padding=1 only")
    s = conv.stride
    if s[0] != s[1] or s[1] != s[2] or s[0] not in (1, 2):
        raise ValueError("the gfx950 3-D block supports stride 1 or 2 only")
    if isinstance(conv, nn.ConvTranspose3d):
        if s[0] != 2 or tuple(conv.output_padding) != (1, 1, 1):
            raise ValueError("transposed 3-D block: stride=2, output_padding=1 only")


def _crop_add(y, residual):
    d, h, w = (min(a, b) for a, b in zip(y.shape[2:], residual.shape[2:]))
    return y[:, :, :d, :h, :w] + residual[:, :, :d, :h, :w]


def _run_block_batch_stats(folded, conv, bn, x, residual, relu):
    """Train-mode block: convolution (forward, bwd-data and bwd-weight on the HIP kernels via
    ``costvolume.Conv3dFunction``), then batch-statistics BatchNorm + cropped skip add + ReLU as
    ONE autograd node of two launches each way (``costvolume.bn_add_relu3d``, csrc/bn3d.hip).
    ``DSM_TRAIN_BN=stock`` (read by this host module) keeps the stock torch ops for A/B runs."""
    del folded
    y = cv.conv3d(x, conv.weight, conv.bias, conv.stride[0], isinstance(conv, nn.ConvTranspose3d))
    momentum = bn.momentum if bn.momentum is not None else 0.1
    if _FUSED_TRAIN_BN and y.shape[1] % 4 == 0 and y.shape[1] <= 256:
        out = cv.bn_add_relu3d(y, bn.weight, bn.bias, residual,
                               bn.running_mean if bn.track_running_stats else None,
                               bn.running_var if bn.track_running_stats else None,
                               relu, momentum, bn.eps)
        if bn.num_batches_tracked is not None:
            bn.num_batches_tracked += 1
        _bump_running_stats(bn)
        if _TRAIN_RELU_HOOK[0] is not None and relu != RELU_NONE:
            _TRAIN_RELU_HOOK[0](out, relu)
        return out
    y = torch.nn.functional.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias,
                                       True, momentum, bn.eps)
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    if relu == RELU_BEFORE_ADD:
        y = torch.relu(y)
    if residual is not None:
        y = _crop_add(y, residual)
    if relu == RELU_AFTER_ADD:
        y = torch.relu(y)
    return y


def _fast_eval(conv, bn, x):
    """Eval-mode, no autograd: the fused single-launch kernels apply."""
    if bn is not None and bn.training:
        return False
    if isinstance(x, cv.VirtualVolume):
        return True
    return not (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad))


def run_block(folded, conv, bn, x, residual=None, relu=RELU_NONE):
    """conv (+ BN) (+ cropped skip) (+ ReLU); one launch in eval mode.  ``x``: a tensor or (eval
    only) a ``costvolume.VirtualVolume``."""
    if not _fast_eval(conv, bn, x):
        if bn is not None and bn.training:
            return _run_block_batch_stats(folded, conv, bn, x, residual, relu)
        # autograd through a block without train-mode BN (bare convolutions such as classif*.2,
        # or eval-mode BN while fine-tuning): unfused path with explicit gradients
        y = cv.conv3d(x, conv.weight, conv.bias, conv.stride[0],
                      isinstance(conv, nn.ConvTranspose3d))
        if bn is not None:
            y = torch.nn.functional.batch_norm(y, bn.running_mean, bn.running_var, bn.weight,
                                               bn.bias, False, 0.0, bn.eps)
        if relu == RELU_BEFORE_ADD:
            y = torch.relu(y)
        if residual is not None:
            y = _crop_add(y, residual)
        if relu == RELU_AFTER_ADD:
            y = torch.relu(y)
        return y
    packed, scale, shift = folded.get(conv, bn)
    return cv.conv3d_block(x, packed, conv.out_channels, scale, shift, residual,
                           stride=conv.stride[0], transposed=isinstance(conv, nn.ConvTranspose3d),
                           relu=relu)


class ConvBN3d(nn.Sequential):
    """``Sequential(conv, [bn], [act])`` with the reference's child indices, fused.

    ``forward(x)`` equals the reference Sequential.  ``forward(x, residual=r, relu=True)``
    additionally fuses what the reference does around it: ``F.relu(seq(x) + r)`` with the
    crop of ``myadd_3d`` (PSMNet), or -- when the ReLU is a child of the Sequential, as in
    ``conv3d_bn`` / ``deconv3d_bn`` -- ``myAdd3d(seq(x), r)`` (GCNet)."""

    def __init__(self, conv, bn=None, act=None):
        layers = [conv]
        if bn is not None:
            layers.append(bn)
        if act is not None:
            layers.append(act)
        super(ConvBN3d, self).__init__(*layers)
        _check_conv(conv)
        self._bn_idx = 1 if bn is not None else None
        self._has_act = act is not None
        self._folded = _Folded()

    def forward(self, x, residual=None, relu=False):
        conv = self[0]
        bn = self[self._bn_idx] if self._bn_idx is not None else None
        if self._has_act:
            if relu:
                raise ValueError("this block already ends in its own ReLU")
            mode = RELU_BEFORE_ADD
        else:
            mode = RELU_AFTER_ADD if relu else RELU_NONE
        return run_block(self._folded, conv, bn, x, residual, mode)


class Chain3d(nn.Sequential):
    """A reference ``nn.Sequential`` of 3-D layers (``dres0``, ``classif1`` ...) run with
    peephole fusion: ``ConvBN3d`` followed by ``nn.ReLU`` becomes one launch; a bare
    ``nn.Conv3d`` child (``classif*.2``) runs through the same kernel; ``residual`` is
    added in the epilogue of the last convolution."""

    def __init__(self, *layers):
        super(Chain3d, self).__init__(*layers)
        self._plain = {}

    def forward(self, x, residual=None, relu=False):
        mods = list(self)
        convs = [i for i, m in enumerate(mods) if isinstance(m, (ConvBN3d, nn.Conv3d, nn.ConvTranspose3d))]
        last_conv = convs[-1]
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt_relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            res = residual if i == last_conv else None
            want_relu = nxt_relu or (relu and i == last_conv)
            if isinstance(m, ConvBN3d):
                if res is not None and nxt_relu:
                    raise ValueError("skip-add before an inner ReLU is not a reference pattern")
                x = m(x, residual=res, relu=want_relu)
            elif isinstance(m, (nn.Conv3d, nn.ConvTranspose3d)):
                _check_conv(m)
                folded = self._plain.setdefault(i, _Folded())
                x = run_block(folded, m, None, x, res, RELU_AFTER_ADD if want_relu else RELU_NONE)
            elif isinstance(m, nn.ReLU):
                raise ValueError("ReLU without a preceding convolution in a 3-D chain")
            else:
                raise TypeError("unsupported layer in a 3-D chain: %r" % (m,))
            i += 2 if nxt_relu else 1
        return x


class Conv3dHip(nn.Conv3d):
    """A bare ``nn.Conv3d`` (k3, p1) whose forward runs on the HIP kernel; state-dict keys are
    the stock ``weight`` / ``bias``.  A real subclass (not a rebound ``forward``): deep copies,
    pickling and ``torch.save(model)`` see an ordinary module with its own fold cache."""

    def __init__(self, *args, **kw):
        super(Conv3dHip, self).__init__(*args, **kw)
        _check_conv(self)
        self._folded = _Folded()

    def forward(self, x, residual=None, relu=False):
        return run_block(self._folded, self, None, x, residual, RELU_AFTER_ADD if relu else RELU_NONE)

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        import copy
        for k, v in self.__dict__.items():
            new.__dict__[k] = _Folded() if k == "_folded" else copy.deepcopy(v, memo)
        return new


class ConvTranspose3dHip(nn.ConvTranspose3d):
    """``nn.ConvTranspose3d`` (k3, s2, p1, op1) on the HIP kernel -- GCNet's ``l37``
    (models/gcnet.py:63)."""

    def __init__(self, *args, **kw):
        super(ConvTranspose3dHip, self).__init__(*args, **kw)
        _check_conv(self)
        self._folded = _Folded()

    def forward(self, x, residual=None, relu=False):
        return run_block(self._folded, self, None, x, residual, RELU_AFTER_ADD if relu else RELU_NONE)

    __deepcopy__ = Conv3dHip.__deepcopy__
