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
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok_gather, ok_grad))


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
