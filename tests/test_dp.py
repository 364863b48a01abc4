"""Data-parallel gradient reduction (ep24.dp) on CPU tensors with the gloo backend, world_size 2.

The reducer is device-agnostic: the same planning (buckets from the backward write ranges, cut points) and the
same launch / wait protocol drive RCCL on the GPUs.  The N=8 run itself belongs to the driver."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def run(fn, world=2):
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world, join=True)
    return [ret[r] for r in range(world)]


def _plan_case(rank, world):
    from ep24 import dp
    n = 1000
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    # backward "ops": op i writes [900-100*i, 1000-100*i) -> the buffer completes from its tail
    writes = [[(900 - 100 * i, 100)] for i in range(10)]
    red = dp.GradReducer(bucket_bytes=250 * 4, first_bucket_bytes=50 * 4)
    red.plan(flat, writes)
    cuts = red.cuts()
    assert cuts[0] == 0 and cuts[-1] == 10 and cuts == sorted(set(cuts))
    order = []
    for seg in range(len(cuts) - 1):
        # a bucket may only be launched once every op that writes into it has run
        for k in red._seg_buckets[seg]:
            lo, hi = red.buckets[k]
            last_writer = max(i for i, ws in enumerate(writes) for off, cnt in ws if off < hi and off + cnt > lo)
            assert last_writer < cuts[seg + 1]
            order.append((lo, hi))
        red.bucket_ready(seg)
    red.wait()
    # graduated sizes from the head of the buffer (what backward completes last): 50 | 250 | the rest in 750-element pieces
    assert sorted(order) == [(0, 50), (50, 300), (300, 1000)] and order[0] == (300, 1000)
    want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
    return bool(torch.equal(flat, want))


def test_bucket_plan_and_sum():
    assert run(_plan_case) == [True, True]


def _dp_vs_shards(rank, world):
    """Every rank: gradient of its own shard; after reduce + 1/world scaling == mean of the per-shard gradients
    computed in one process (per-rank BN statistics and num_fg normalisation stay local, like the legacy DDP)."""
    from ep24 import dp
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3))
    params = list(net.parameters())
    n = sum(p.numel() for p in params)
    flat = torch.zeros(n)
    off = 0
    writes = []
    for p in reversed(params):                    # backward order: last layer first
        writes.append([(n - off - p.numel(), p.numel())])
        off += p.numel()
    views, o = [], n
    for p in reversed(params):
        o -= p.numel()
        views.append((p, flat[o:o + p.numel()].view_as(p)))
    data = torch.randn(world, 5, 8, generator=torch.Generator().manual_seed(1))
    shard_grads = []
    for r in range(world):
        net.zero_grad()
        net(data[r]).pow(2).mean().backward()
        shard_grads.append([p.grad.clone() for p in params])
    for (p, v), g in zip(views, reversed(shard_grads[rank])):
        v.copy_(g)
    red = dp.GradReducer(bucket_bytes=64 * 4)
    red.plan(flat, writes)
    red.reduce_all()
    flat.mul_(1.0 / world)
    ok = True
    for (p, v), i in zip(views, reversed(range(len(params)))):
        mean = sum(sg[i] for sg in shard_grads) / world
        ok = ok and torch.allclose(v, mean, atol=1e-6)
    return bool(ok)


def test_dp_mean_equals_simulated_shards():
    assert run(_dp_vs_shards) == [True, True]


def test_requires_process_group():
    from ep24 import dp
    if dist.is_initialized():
        pytest.skip("process group already up")
    with pytest.raises(RuntimeError):
        dp.GradReducer()
