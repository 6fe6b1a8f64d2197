"""CPU suite: the data-parallel exchange (hipops.dist) with world_size 2 over gloo -- bucket
partition, asynchronous bucketed all-reduce(sum) of a flat gradient, early launch of one
slice (the decoder range) + reduce_rest for the remainder, and the scalar loss average."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from hipops import dist as hd
        n = 1000 * 64 + 64
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)
        want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        red = hd.GradReducer(g, bucket_elems=4096)
        red.reduce_range(12800, 51200)          # the slice that is ready first (decoder gradients)
        assert len(red.pending) == (51200 - 12800 + 4095) // 4096
        red.reduce_rest()                       # everything else exactly once
        red.wait()
        ok = torch.equal(g, want)
        # a second step must start from a clean state
        g2 = torch.ones(n) * (rank + 1)
        red2 = hd.GradReducer(g2, bucket_elems=1 << 20)
        red2.reduce_rest(); red2.wait()
        ok = ok and torch.equal(g2, torch.full((n,), float(sum(r + 1 for r in range(world)))))
        w = torch.full((10,), float(rank))
        hd.broadcast_flat(w, 0)
        ok = ok and torch.equal(w, torch.zeros(10))
        mean = hd.all_reduce_mean_scalar(float(rank + 1))
        ok = ok and abs(mean - (sum(r + 1 for r in range(world)) / world)) < 1e-12
        # the job's seed: drawn on rank 0 when the config has none, identical everywhere (each rank's own python RNG
        # is seeded differently here, as unseeded processes are)
        import random
        random.seed(1000 + rank)
        seeds = [None] * world
        dist.all_gather_object(seeds, hd.shared_seed(None))
        ok = ok and len(set(seeds)) == 1 and 1 <= seeds[0] <= 10000 and hd.shared_seed(77) == 77
        # bf16 gradient transport: all-to-all of bf16 shards, fp32 sum on arrival, all-gather; ragged bucket sizes
        gen = torch.Generator().manual_seed(5 + rank)
        gb = torch.randn(n, generator=gen)
        parts = [torch.randn(n, generator=torch.Generator().manual_seed(5 + r)) for r in range(world)]
        want_bf = sum(p.bfloat16().float() for p in parts).bfloat16().float()      # one rounding per addend, one per sum
        red3 = hd.GradReducer(gb, bucket_elems=5000, transport="bf16")
        red3.reduce_range(12800, 51200, early=True)
        red3.reduce_rest(); red3.wait()
        ok = ok and torch.equal(gb, want_bf)
        ok = ok and float((gb - sum(parts)).norm() / sum(parts).norm()) < 6e-3
        assert hd.backend_name() == "gloo"
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_split_buckets_cover_range_backwards():
    from hipops.dist import split_buckets
    b = split_buckets(128, 1000 * 64, 4096)
    assert b[0][1] == 64000 and b[-1][0] == 128
    assert all(x[0] == y[1] for x, y in zip(b[:-1], b[1:]))          # contiguous, emitted tail-first
    assert all(e - s <= 4096 for s, e in b)
    assert split_buckets(5, 5, 64) == []


def test_single_process_is_a_noop():
    from hipops import dist as hd
    g = torch.ones(100)
    r = hd.GradReducer(g)
    r.reduce_range(0, 50); r.reduce_rest(); r.wait()
    assert torch.equal(g, torch.ones(100)) and hd.world_size() == 1
    assert hd.all_reduce_mean_scalar(3.5) == 3.5
