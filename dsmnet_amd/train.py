"""Supervised training step with the reference's objective and bookkeeping (SURVEY.md
section 8f-3): ``losses`` (losses/loss.py:341-392,407-422,515-516 -- ``loss_supervised`` :326-338),
``lr_adjust`` (stereo.py:95-101), ``accuracy`` (stereo.py:103-113), ``AverageMeter``
(utils/utils.py:87-118) and the body of the reference's train / validate loops
(stereo_supervised.py:43-119, 121-190) as ``train_step`` / ``validate_step``.

The forward and backward of the model run on the HIP kernels (``costvolume`` autograd
functions); the loss itself is a handful of element-wise stock torch ops on (B,1,H,W) maps.
Only the supervised objective is covered -- the self-supervised ones (depthmono / SsSMnet /
Cap_ds_lr) need the image-warping path, which SURVEY.md section 8f-4 ranks after this.
"""
import torch
import torch.nn.functional as F

from . import sharding


def _diff1_dx(img):
    return F.pad(img[:, :, :, 1:] - img[:, :, :, :-1], [0, 1, 0, 0])


def _diff1_dy(img):
    return F.pad(img[:, :, 1:] - img[:, :, :-1], [0, 0, 0, 1])


class losses(torch.nn.Module):
    """``losses(loss_name="supervised", count_levels, maxepoch_weight_adjust)``; call with the
    reference's argument dict ``{"disp_gt", "disps", "scale_disps", "flag_smooth"}``."""

    def __init__(self, loss_name="supervised", count_levels=1, maxepoch_weight_adjust=1):
        super(losses, self).__init__()
        name = loss_name.split("-")[0].lower()
        if "supervised" not in name:
            raise NotImplementedError(
                "only the supervised objective is built (loss_name=%r); the self-supervised "
                "ones need the image-warping path (SURVEY.md section 8f-4)" % loss_name)
        self.flag_mask = "mask" in loss_name
        self.lossfun = self.loss_supervised
        self.lossesfun = self.losses_pyramid0
        self.maxepoch_weight_adjust = maxepoch_weight_adjust
        self.count_levels = count_levels
        self.weight_levels = [0] * count_levels
        self.weight_levels[-1] = 1
        # capturable = True: no host-side test for "no valid pixel" (a device synchronisation);
        # the loss is then a zero tensor instead of the reference's integer 0 -- what a hipGraph
        # capture of the whole step needs (graphs.GraphedTrainStep sets it)
        self.capturable = False

    def Weight_Adjust_levels(self, epoch):
        """Coarse-to-fine: the unit weight slides from the coarsest output (epoch 0) to the
        finest (``maxepoch_weight_adjust``); every other level keeps 0.01."""
        n, maxepoch = self.count_levels, self.maxepoch_weight_adjust
        self.weight_levels = [0.01] * n
        if n == 1 or epoch >= maxepoch:
            self.weight_levels[0] = 1
            return
        x = (1 - epoch / float(maxepoch)) * (n - 1)
        idx = int(x)
        w = x - idx
        self.weight_levels[idx] = 1 - w
        if idx < n - 1:
            self.weight_levels[idx + 1] = w

    def loss_supervised(self, disp_gt, disp, flag_smooth=False, factor=1.0):
        mask = disp_gt > 0
        if self.capturable:                     # same value, fixed shapes, no synchronisation
            m = mask.to(disp.dtype)
            n = m.sum().clamp_min(1.0)
            loss = (torch.abs(disp_gt - disp) * m).sum() / n
            if flag_smooth:
                dxdy = (torch.abs(_diff1_dx(disp)) + torch.abs(_diff1_dy(disp))) / factor
                loss = loss + 0.1 * (dxdy.clamp(0, 1) * m).sum() / n
            return loss
        if not bool(mask.any()):
            return 0
        loss = torch.abs(disp_gt - disp)[mask].mean()
        if flag_smooth:
            dxdy = (torch.abs(_diff1_dx(disp)) + torch.abs(_diff1_dy(disp))) / factor
            loss = loss + 0.1 * dxdy[mask].clamp(0, 1).mean()
        return loss

    def losses_pyramid0(self, disp_gt, disps, scale_disps, flag_smooth=False):
        _, _, h, w = disp_gt.shape
        loss = 0
        for pred, level in zip(disps, scale_disps):
            weight = self.weight_levels[level]
            if weight <= 0:
                continue
            if pred.dim() == 3:          # PSMNet returns (B,H,W); the reference's diff1_dx asserts 4-D
                pred = pred.unsqueeze(1)
            if level > 0:
                pred = F.interpolate(pred, scale_factor=2 ** level, mode="bilinear",
                                     align_corners=False)[:, :, :h, :w]
            loss = loss + self.lossfun(disp_gt, pred, flag_smooth, factor=1) * weight
        return loss

    def forward(self, args):
        return self.lossesfun(**args)


def lr_adjust(optimizer, epoch0, stride, lr0, epoch):
    """Halve the learning rate every ``stride`` epochs from ``epoch0`` on (first halving at
    ``epoch0`` itself); untouched before."""
    if epoch < epoch0:
        return
    n = ((epoch - epoch0) // stride) + 1
    lr = lr0 * (0.5 ** n)
    for group in optimizer.param_groups:
        group["lr"] = lr


def accuracy(dispL, dispL_gt):
    """(D1 %, EPE) over gt > 0.  Good pixel: error <= 3 px OR <= 5 % (the reference's OR)."""
    if dispL.dim() == 3:
        dispL = dispL.unsqueeze(1)
    mask = dispL_gt > 0
    diff = (dispL_gt - dispL).abs()[mask]
    epe = diff.mean()
    good = (diff <= 3) | ((diff / dispL_gt[mask]) <= 0.05)
    d1 = 100 - 100.0 * good.sum() / mask.sum()
    return d1, epe


class AverageMeter(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def make_optimizer(model, lr=1e-4, betas=(0.9, 0.999)):
    """stereo.py:37-40 with main.py's defaults (:31-33)."""
    return torch.optim.Adam(model.parameters(), lr=lr, betas=betas)


def _split(batch):
    if batch.shape[1] < 7:
        raise ValueError("a supervised batch is (B, >=7, H, W): imL | imR | dispL")
    return batch[:, :3], batch[:, 3:6], batch[:, 6:7]


def _world(world):
    import torch.distributed as dist
    if world is not None:
        return world
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def train_step(model, optim, lossfun, batch, world=None):
    """One iteration of stereo_supervised.py:53-97: forward in train mode, pyramid loss with
    ``flag_smooth=True``, backward, (multi-GPU: one flat gradient all-reduce), Adam step.
    ``batch`` = (B,7,H,W) imL | imR | dispL, as the reference's loader yields it.
    Returns (loss, D1, EPE) as floats.

    Single process: a batch without any ground-truth pixel gives the reference's integer-0 loss
    and the step is skipped, as there (losses/loss.py:330-331).  Several ranks: EVERY rank runs
    backward and the all-reduce every step (the loss is the tensor form, zero -- with zero
    gradients -- on a rank without ground truth), so the collective calls always match; the
    number of ranks that had ground truth rides in the same bucket and the optimizer step is
    skipped on all ranks together when it is zero."""
    from . import costvolume as cv
    with cv.amax_scope(batch.device):            # fp16 convolution modes: one arena of maxima per step
        return _train_step(model, optim, lossfun, batch, world)


def _train_step(model, optim, lossfun, batch, world):
    model.train()
    world = _world(world)
    imL, imR, dispL = _split(batch)
    scale_dispLs, dispLs = model(imL, imR)
    args = {"disp_gt": dispL, "disps": dispLs, "scale_disps": scale_dispLs, "flag_smooth": True}
    optim.zero_grad()
    if world > 1:
        was = lossfun.capturable
        lossfun.capturable = True               # tensor loss on every rank, no host-side branch
        try:
            loss = lossfun(args)
        finally:
            lossfun.capturable = was
        loss.backward()
        has_gt = (dispL > 0).any().to(torch.float32).reshape(1)
        _, n_gt = sharding.allreduce_gradients(model.parameters(), world=world, extra=has_gt)
        if float(n_gt) > 0:
            optim.step()
    else:
        loss = lossfun(args)
        if torch.is_tensor(loss):              # the integer 0 when no pixel has ground truth
            loss.backward()
            optim.step()
    d1, epe = accuracy(dispLs[0].detach(), dispL)
    return float(loss.detach() if torch.is_tensor(loss) else loss), float(d1), float(epe)


def validate_step(model, lossfun, batch):
    """One iteration of stereo_supervised.py:129-162 (eval mode, no gradients)."""
    model.eval()
    imL, imR, dispL = _split(batch)
    with torch.no_grad():
        scale_dispLs, dispLs = model(imL, imR)
        loss = lossfun({"disp_gt": dispL, "disps": dispLs, "scale_disps": scale_dispLs,
                        "flag_smooth": True})
        d1, epe = accuracy(dispLs[0], dispL)
    return float(loss), float(d1), float(epe)
