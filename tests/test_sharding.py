"""CPU, world_size 2 (gloo): the pair partition and the gradient all-reduce of the N > 1 path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dsmnet_amd import sharding


def test_pair_partition_is_exact():
    for n in (0, 1, 7, 8, 33):
        for world in (1, 2, 3, 8):
            owned = sorted(i for r in range(world) for i in sharding.pair_indices(n, r, world))
            assert owned == list(range(n))
            sizes = [len(sharding.pair_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = sharding.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    left = torch.arange(5 * 3 * 2 * 4, dtype=torch.float32).view(5, 3, 2, 4)
    l, rr, idx = sharding.shard_batch(left, left + 1, rank, world)
    assert idx == list(range(rank, 5, world)) and torch.equal(l, left[idx])
    # "disparity" = a per-pair function; gather must restore pair order on every rank
    local = l.mean(dim=(1, 2, 3)) + 10 * rr.mean(dim=(1, 2, 3))
    full = sharding.gather_disparities(local, idx, 5, world)
    want = left.mean(dim=(1, 2, 3)) + 10 * (left + 1).mean(dim=(1, 2, 3))
    ok_gather = torch.allclose(full, want)
    # gradient all-reduce: rank-dependent grads, one param without grad on rank 1
    lin = torch.nn.Linear(3, 2)
    for p in lin.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    if rank == 1:
        lin.bias.grad = None
    n = sharding.allreduce_gradients(lin.parameters(), world)
    ok_grad = (n == 8 and torch.allclose(lin.weight.grad, torch.full((2, 3), 1.5))
               and torch.allclose(lin.bias.grad, torch.full((2,), 0.5)))
    # the same through one flat buffer (graphs.GraphedTrainStep with several ranks): gradients are
    # views of it, the trailing float comes back summed
    lin2 = torch.nn.Linear(3, 2)
    fg = sharding.FlatGradients(lin2.parameters(), n_extra=1)
    fg.zero()
    (lin2(torch.ones(1, 3)).sum() * (rank + 1)).backward()      # accumulates INTO the views
    assert lin2.weight.grad.data_ptr() == fg.flat.data_ptr()
    fg.extra.fill_(float(rank == 0))
    n_gt = fg.allreduce(world)
    ok_flat = (torch.allclose(lin2.weight.grad, torch.full((2, 3), 1.5)) and
               torch.allclose(lin2.bias.grad, torch.full((2,), 1.5)) and float(n_gt) == 1.0)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok_gather, ok_grad and ok_flat))


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1]
    assert all(r[1] and r[2] for r in results), results


class _ToyStereo(torch.nn.Module):
    """A stock-torch stand-in with the models' call contract: (imL, imR) -> (scales, disps)."""

    def __init__(self):
        super(_ToyStereo, self).__init__()
        self.count_levels = 2
        self.conv = torch.nn.Conv2d(6, 1, 3, padding=1)

    def forward(self, imL, imR, mode="train"):
        d0 = self.conv(torch.cat([imL, imR], 1))
        return [0, 1], [d0, torch.nn.functional.avg_pool2d(d0, 2)]


def _batch(seed):
    g = torch.Generator().manual_seed(seed)
    return torch.cat([torch.rand(2, 6, 8, 12, generator=g), torch.rand(2, 1, 8, 12, generator=g) * 9 + 1], 1)


def _train_worker(rank, world, port, q):
    from dsmnet_amd import train
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sharding.init_from_env("gloo")
    torch.manual_seed(0)
    model = _ToyStereo()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    lossfun = train.losses("supervised", 2, 4)
    lossfun.Weight_Adjust_levels(1)
    train.train_step(model, opt, lossfun, _batch(100 + rank))        # world from the process group
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, [p.detach().numpy().copy() for p in model.parameters()]))   # by value


def test_train_step_world_size_2_equals_averaged_gradients():
    from dsmnet_amd import train
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process: the mean of the two ranks' gradients, one SGD step
    torch.manual_seed(0)
    model = _ToyStereo()
    lossfun = train.losses("supervised", 2, 4)
    lossfun.Weight_Adjust_levels(1)
    grads = []
    for r in range(2):
        model.zero_grad()
        b = _batch(100 + r)
        s, d = model(b[:, :3], b[:, 3:6])
        lossfun({"disp_gt": b[:, 6:7], "disps": d, "scale_disps": s, "flag_smooth": True}).backward()
        grads.append([p.grad.clone() for p in model.parameters()])
    want = [p.detach() - 0.1 * (g0 + g1) / 2 for p, g0, g1 in zip(model.parameters(), *grads)]
    for r in (0, 1):
        for got, w in zip(results[r], want):
            assert torch.allclose(torch.from_numpy(got), w, atol=1e-6)


def _empty_gt_worker(rank, world, port, q, empty_ranks):
    from dsmnet_amd import train
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sharding.init_from_env("gloo")
    torch.manual_seed(0)
    model = _ToyStereo()
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    lossfun = train.losses("supervised", 2, 4)
    lossfun.Weight_Adjust_levels(1)
    b = _batch(100 + rank)
    if rank in empty_ranks:
        b[:, 6:7] = 0                       # no pixel with ground truth on this rank
    loss, _, _ = train.train_step(model, opt, lossfun, b)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, loss, [p.detach().numpy().copy() for p in model.parameters()]))


def _run_empty_gt(empty_ranks):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_gt_worker, args=(r, 2, port, q, empty_ranks)) for r in range(2)]
    for p in procs:
        p.start()
    results = {r: (loss, params) for r, loss, params in (q.get(timeout=120) for _ in procs)}   # a hang fails here
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def test_train_step_with_one_rank_without_ground_truth_does_not_deadlock():
    """a rank whose batch has no gt > 0 pixel used to skip backward AND the
    all-reduce while the other rank waited in it.  Now every rank reduces every step; the
    empty rank contributes zero gradients and the parameters stay identical across ranks."""
    from dsmnet_amd import train
    results = _run_empty_gt({1})
    assert results[1][0] == 0.0 and results[0][0] > 0.0
    torch.manual_seed(0)
    model = _ToyStereo()
    lossfun = train.losses("supervised", 2, 4)
    lossfun.Weight_Adjust_levels(1)
    b = _batch(100)
    s, d = model(b[:, :3], b[:, 3:6])
    lossfun({"disp_gt": b[:, 6:7], "disps": d, "scale_disps": s, "flag_smooth": True}).backward()
    want = [p.detach() - 0.1 * p.grad / 2 for p in model.parameters()]     # (g0 + 0) / 2
    for r in (0, 1):
        for got, w in zip(results[r][1], want):
            assert torch.allclose(torch.from_numpy(got), w, atol=1e-6)


def test_train_step_with_no_ground_truth_anywhere_skips_the_step_on_all_ranks():
    results = _run_empty_gt({0, 1})
    torch.manual_seed(0)
    untouched = [p.detach() for p in _ToyStereo().parameters()]
    for r in (0, 1):
        assert results[r][0] == 0.0
        for got, w in zip(results[r][1], untouched):
            assert torch.equal(torch.from_numpy(got), w)
