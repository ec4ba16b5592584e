"""Multi-GPU use of the cost-volume path: one process per GPU, pairs sharded.

The reference is single-process, single-device (`stereo.py:24,32-34`; its only trace of data
parallelism is a commented-out DistributedDataParallel line).  A stereo pair's forward is
independent of every other pair's, so inference shards pairs across ranks with **no collective
on the data path**; the only exchange the path ever needs is the optional training-time
gradient all-reduce (PSMNet 5.22 M parameters = 20.9 MB fp32 per step, a
~0.24 ms ring over xGMI -- negligible next to the step, so one flat bucket, no overlap logic).

`torch.distributed` with backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the process group from torchrun's environment (RANK, WORLD_SIZE, MASTER_*).
    Returns (rank, world_size, local_rank).  Single-process runs need no group."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def pair_indices(num_pairs, rank, world):
    """Indices of the pairs rank ``rank`` owns: pair i -> rank i mod world.
    Every pair is owned exactly once; loads differ by at most one pair."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return list(range(rank, num_pairs, world))


def shard_batch(left, right, rank, world):
    """The slice of a (B,3,H,W) batch this rank processes (interleaved, like pair_indices)."""
    if left.shape != right.shape:
        raise ValueError("left/right batch shapes differ")
    idx = pair_indices(left.shape[0], rank, world)
    return left[idx], right[idx], idx


def gather_disparities(local, idx, num_pairs, world):
    """Optional: reassemble the per-pair outputs on every rank (evaluation convenience; not
    used by the benchmark).  ``local``: (len(idx), ...) tensor."""
    if world == 1:
        return local
    counts = [len(range(r, num_pairs, world)) for r in range(world)]
    pad = max(counts)
    buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = local.new_empty((num_pairs,) + tuple(local.shape[1:]))
    for r in range(world):
        out[r:num_pairs:world] = parts[r][: counts[r]]
    return out


def allreduce_gradients(parameters, world=None, extra=None):
    """Average gradients across ranks with ONE flat all-reduce (sum, then / world).
    Parameters without a gradient contribute zeros so that every rank reduces the same
    buffer -- every rank must call this every step, whatever its own batch held.
    ``extra``: an optional 1-D float tensor that rides in the same bucket and comes back
    SUMMED over ranks (not averaged) -- e.g. the count of ranks whose batch had ground truth.
    Returns the number of gradient elements reduced, or ``(that, extra_sum)`` with ``extra``."""
    if world is None:
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    params = [p for p in parameters if p.requires_grad]
    if world == 1 or not params:
        return 0 if extra is None else (0, extra.clone())
    parts = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]
    if extra is not None:
        parts.append(extra.to(parts[0].dtype).reshape(-1))
    flat = torch.cat(parts)
    _all_reduce_sum(flat)
    total = sum(p.numel() for p in params)
    flat[:total].div_(world)                     # one launch; the trailing `extra` stays a sum
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off: off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    if extra is None:
        return off
    return off, flat[off:].clone()


def _all_reduce_sum(flat):
    """dist.all_reduce(SUM); a CUDA tensor under the gloo backend (the rehearsal of the multi-rank
    path on fewer GPUs than ranks) is staged through the host."""
    if flat.is_cuda and dist.get_backend() == "gloo":
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)


class FlatGradients(object):
    """The gradients of ``parameters`` as views of ONE flat buffer (plus ``n_extra`` trailing floats):
    autograd accumulates into the views in place, so a step's gradient exchange is one memset
    before the backward, one all-reduce and one division after it -- no concatenation, no copies
    back, and nothing that changes an address between steps, which is what lets the backward live in
    a captured hipGraph while the collective stays an ordinary RCCL call (``graphs.GraphedTrainStep``
    with several ranks).  The reference has no counterpart (``stereo.py:34``, a commented-out
    DistributedDataParallel)."""

    def __init__(self, parameters, n_extra=0):
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradients: no parameter requires a gradient")
        self.total = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(self.total + n_extra, device=p0.device, dtype=p0.dtype)
        self.attach()

    def attach(self):
        """(Re)bind every ``p.grad`` to its view of the flat buffer."""
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off: off + n].view_as(p)
            off += n

    @property
    def extra(self):
        return self.flat[self.total:]

    def zero(self):
        self.flat.zero_()

    def allreduce(self, world=None):
        """Sum over ranks, gradients divided by ``world``; ``extra`` comes back summed."""
        if world is None:
            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world > 1:
            _all_reduce_sum(self.flat)
            self.flat[: self.total].div_(world)
        return self.extra
