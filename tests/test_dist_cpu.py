"""-m "not gpu": the N > 1 path on CPU with world_size-2 gloo (sharding, barrier, max-over-ranks, flat gradient mean)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from genima_amd import dist as gd


def test_shard_range_partitions():
    for n in (1, 7, 8, 9, 64):
        for world in (1, 2, 3, 8):
            spans = [gd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, l, w = gd.init_from_env("gloo")
    assert (r, w) == (rank, world)
    s, e = gd.shard_range(9, r, w)
    gd.barrier()
    tmax = gd.max_over_ranks(1.0 + rank)
    g = torch.arange(11, dtype=torch.float32) * (rank + 1)  # 11 % 2 != 0 exercises the tail path
    gd.allreduce_mean_flat(g)
    # the trainer's form: SUM in place + the world size; reduce-scatter -> all-gather -> tail all-reduce is the same call sequence
    # the RCCL backend runs (gloo's missing reduce-scatter is one `reduce` per destination rank)
    h = torch.arange(11, dtype=torch.float32) * (rank + 1)
    assert gd.allreduce_sum_flat(h) == world and h.tolist() == (torch.arange(11, dtype=torch.float32) * 3).tolist()
    # bucketed exchange driven by a (fake) backward walk: 10 "parameters" of 100 elements laid out in forward order, first used by
    # tape entries 0..9; the walk runs 9 -> 0, so buckets become final from the END of the buffer towards its front
    layout = {f"p{i}": (100 * i, (100,)) for i in range(10)}
    first_use = {f"p{i}": i for i in range(9)}  # p9 is never touched: final before the walk starts
    base = torch.arange(1000, dtype=torch.float32)
    grad = base * (rank + 1)
    bk = gd.GradBuckets(n_buckets=4, align=8)
    bk.begin(grad, layout, first_use, 10)
    order = []
    bk.entry_done(10)
    order.append(list(bk.fired))
    for i in range(9, -1, -1):
        bk.entry_done(i)
        order.append(list(bk.fired))
    assert bk.finish() == world
    fired_at = {b: next(k for k, o in enumerate(order) if b in o) for b in range(len(bk.ranges))}
    for b, (lo, hi) in enumerate(bk.ranges):
        need = min(first_use.get(f"p{i}", 10) for i in range(10) if 100 * i < hi and 100 * i + 100 > lo)
        assert fired_at[b] == 10 - need, (b, fired_at[b], need)  # order[k] is the state after entry 10 - k
    assert torch.equal(grad, base * 3), "bucketed exchange must equal the one-shot sum"
    q.put((rank, (s, e), tmax, g.tolist()))
    gd.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == (0, 5) and res[1][1] == (5, 9)
    assert res[0][2] == res[1][2] == 2.0
    expect = (torch.arange(11, dtype=torch.float32) * 1.5).tolist()
    assert res[0][3] == expect and res[1][3] == expect
