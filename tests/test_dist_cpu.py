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


class _FakeWork:
    def __init__(self, log, tag):
        self.log, self.tag = log, tag

    def wait(self):
        self.log.append(("wait",) + self.tag)


def _worker_rccl_order(rank, world, port, q):
    """The async branch the RCCL backend takes (`reduce_scatter_tensor` + `all_gather_into_tensor` with async_op=True, waited in
    finish()), driven at world size 2: the collectives are gloo-backed shims behind torch.distributed's names, so the ISSUE ORDER and
    the arithmetic of dist.GradBuckets' nccl path are executed for real with two ranks (no 2-GPU box is available to the builder)."""
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    gd.init_from_env("gloo")
    log = []
    real_reduce, real_all_gather, real_all_reduce = dist.reduce, dist.all_gather, dist.all_reduce

    def reduce_scatter_tensor(out, inp, op=None, async_op=False):
        chunks = inp.view(world, -1).clone()
        for r in range(world):
            real_reduce(chunks[r], dst=r, op=dist.ReduceOp.SUM)
        out.copy_(chunks[rank])
        log.append(("rs", inp.data_ptr(), inp.numel(), async_op))
        return _FakeWork(log, ("rs", inp.data_ptr()))

    def all_gather_into_tensor(out, inp, async_op=False):
        parts = [torch.empty_like(inp) for _ in range(world)]
        real_all_gather(parts, inp.clone())
        out.copy_(torch.cat(parts))
        log.append(("ag", out.data_ptr(), out.numel(), async_op))
        return _FakeWork(log, ("ag", out.data_ptr()))

    def all_reduce(t, op=None, async_op=False):
        real_all_reduce(t, op=dist.ReduceOp.SUM)
        log.append(("ar", t.data_ptr(), t.numel(), async_op))
        return _FakeWork(log, ("ar", t.data_ptr()))

    gd.dist.reduce_scatter_tensor, gd.dist.all_gather_into_tensor, gd.dist.all_reduce = reduce_scatter_tensor, all_gather_into_tensor, all_reduce
    gd.dist.get_backend = lambda *a, **k: "nccl"
    layout = {f"p{i}": (100 * i, (100,)) for i in range(10)}
    first_use = {f"p{i}": i for i in range(10)}
    base = torch.arange(1003, dtype=torch.float32)  # 1003: the last bucket keeps an odd tail -> the small all-reduce
    layout["tail"] = (1000, (3,))
    first_use["tail"] = 9
    grad = base * (rank + 1)
    bk = gd.GradBuckets(n_buckets=4, align=8)
    bk.begin(grad, layout, first_use, 10)
    # the trainer joins its weight-gradient stream exactly where an exchange is launched: fire_indices must name those entries
    assert bk.fire_indices == set(bk.ready) and len(bk.fire_indices) >= 2
    for i in range(10, -1, -1):
        before = len(log)
        bk.entry_done(i)
        assert (len(log) > before) == (i in bk.fire_indices), (i, bk.fire_indices)
    issued = [e for e in log if e[0] != "wait"]
    assert not any(e[0] == "wait" for e in log), "no collective may be waited on inside the backward walk"
    assert all(e[3] for e in issued), "the RCCL path must issue asynchronously"
    # buckets fire from the end of the buffer to its front; inside a bucket: reduce-scatter, all-gather, (tail all-reduce)
    kinds = [e[0] for e in issued]
    assert kinds[:3] == ["rs", "ag", "ar"] and kinds[3:] == ["rs", "ag"] * (len(bk.ranges) - 1), kinds
    starts = [e[1] for e in issued if e[0] == "rs"]
    assert starts == sorted(starts, reverse=True)
    n_works = len(bk.works)
    assert bk.finish() == world
    assert sum(1 for e in log if e[0] == "wait") == n_works == len(issued)
    assert torch.equal(grad, base * 3)
    q.put(rank)
    real_all_reduce(torch.zeros(1))
    dist.destroy_process_group()


def test_two_rank_async_bucket_issue_order():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_rccl_order, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    assert sorted(q.get(timeout=120) for _ in range(2)) == [0, 1]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher around it re-runs itself under torch.distributed.run: both
    ranks start (and, without GPUs here, each stops at its device check -- not at the old WORLD_SIZE assertion)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = env["CUDA_VISIBLE_DEVICES"] = ""  # same outcome on a GPU box: the ranks find no device
    for script in ("bench.py", "bench_train.py"):
        r = subprocess.run([sys.executable, os.path.join(root, script), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                           capture_output=True, text=True, timeout=600)
        assert "launching 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr, r.stderr[-2000:]
        assert "AssertionError: --gpus" not in r.stderr
        if script == "bench.py":
            # (torchrun tears the sibling down as soon as the first rank exits, so only ONE of the two messages is guaranteed to reach stderr)
            assert "rank 0 of 2 needs ROCm device 0" in r.stderr or "rank 1 of 2 needs ROCm device 1" in r.stderr, r.stderr[-2000:]
        assert r.returncode != 0
    cmd = gd.self_launch_command("bench.py", ["--gpus", "4"], 4, port=1234)
    assert cmd[1:9] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1", "--master-port", "1234"]
    assert cmd[9:] == ["bench.py", "--gpus", "4"]
