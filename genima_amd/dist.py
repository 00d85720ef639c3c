"""One-process-per-GPU helpers over torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).

Inference: tiled samples (episodes x frames) are independent, the 4 views inside one sample are not, so
ranks shard EPISODES and there is no data-path collective -- only the barrier / max-over-ranks the benchmark contract asks for.
Training (a16): ONE exchange per optimizer step -- the flat ControlNet gradient buffer is mean-reduced as reduce-scatter +
all-gather so every one of the 7 xGMI links of a GPU carries 1/N of the buffer per phase instead of a ring's per-link bound.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """Initialise from torch.distributed.run's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  -> (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GN_BENCH_SHARE_DEVICE") == "1":  # test aid: every rank on device 0 (bench.py)
        local = 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
    return rank, local, world


def self_launch_command(script: str, argv, n_ranks: int, port: int = None):
    """The command line that runs ``script argv`` as ``n_ranks`` processes of one node, one per GPU -- exactly what the driver's
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P`` wrapper does (the
    role of ``accelerate launch`` in front of diffusion/train_controlnet_genima.py:1216-1218)."""
    import socket
    import sys

    if port is None:
        with socket.socket() as s:  # a free port on the loopback (the container hostname may not resolve)
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script] + list(argv)


def maybe_self_launch(script: str, argv, n_ranks: int) -> None:
    """``python bench.py --gpus N`` with N > 1 and no launcher around it (WORLD_SIZE unset): re-run the same command under
    torch.distributed.run and exit with its status; rank 0 of the children prints the single JSON line.  Returns when the process is
    already one of the ranks (or N == 1)."""
    import subprocess
    import sys

    if n_ranks <= 1 or "WORLD_SIZE" in os.environ:
        return
    cmd = self_launch_command(script, argv, n_ranks)
    print(f"[genima_amd.dist] launching {n_ranks} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    sys.exit(subprocess.run(cmd, env=env).returncode)


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced, exhaustive partition of ``n_items`` episodes over ``world`` ranks."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(x: float, device="cpu") -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(x)
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _reduce_scatter_sum(mine: torch.Tensor, full: torch.Tensor, async_op: bool = False):
    """mine <- this rank's 1/world slice of the sum over ranks of ``full``.  RCCL: one ``reduce_scatter_tensor``.  gloo (CPU tests) has
    no reduce-scatter: the same result -- bit for bit on every rank -- through one ``reduce`` per destination rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() == "gloo":
        chunks = full.view(world, -1)
        for r in range(world):
            dist.reduce(chunks[r], dst=r, op=dist.ReduceOp.SUM)  # chunk r now holds the total on rank r (other ranks: partial garbage)
        mine.copy_(chunks[rank])
        return None
    return dist.reduce_scatter_tensor(mine, full, op=dist.ReduceOp.SUM, async_op=async_op)


def _exchange_sum(buf: torch.Tensor, scratch: torch.Tensor = None, async_op: bool = False):
    """In-place SUM over ranks of a flat f32 buffer as reduce-scatter + all-gather: each of a GPU's 7 xGMI links carries 1/N of the
    buffer per phase (a ring would be bound by one link).  A tail shorter than the world size goes through one small all-reduce.
    Returns the outstanding works (async) -- the collectives of one process group run in issue order on its own stream."""
    world = dist.get_world_size()
    n = buf.numel()
    if dist.get_backend() == "gloo" and buf.is_cuda:  # gloo moves device tensors only through all_reduce (1-GPU multi-process tests)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        return []
    main = n - n % world
    works = []
    if main:
        mine = scratch[: main // world] if scratch is not None else torch.empty(main // world, dtype=buf.dtype, device=buf.device)
        works.append(_reduce_scatter_sum(mine, buf[:main], async_op))
        works.append(dist.all_gather_into_tensor(buf[:main], mine, async_op=async_op))
    if n != main:
        works.append(dist.all_reduce(buf[main:], op=dist.ReduceOp.SUM, async_op=async_op))
    return [w for w in works if w is not None]


def allreduce_sum_flat(buf: torch.Tensor, force: bool = False) -> int:
    """In-place SUM over ranks of the flat gradient buffer (bit-identical on every rank afterwards).  Returns the number of ranks: the
    trainer folds the 1 / world of the mean into its unscale factor instead of spending a pass over 1.46 GB on it.
    ``force`` runs the collectives even at world size 1 (the 1-GPU test of the RCCL branch)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size()
    if world > 1 or force:
        _exchange_sum(buf)
    return world


def allreduce_mean_flat(buf: torch.Tensor) -> torch.Tensor:
    """Mean over ranks (kept for callers that want the mean in the buffer itself; the trainer uses the SUM form)."""
    world = allreduce_sum_flat(buf)
    if world > 1:
        buf.div_(world)
    return buf


class GradBuckets:
    """Gradient exchange overlapped with the backward walk (the role of DDP's buckets under ``accelerator.backward``,
    diffusion/train_controlnet_genima.py:1216-1218, :1402): the flat buffer is cut into ``n_buckets`` contiguous ranges; the tape
    (training.Graph) knows, for every parameter, the index of the entry that writes its gradient LAST in the reversed walk; a bucket's
    reduce-scatter + all-gather is launched (async, on the process group's stream, ordered behind the compute stream's work so far) as
    soon as every parameter inside it is final, while the backward keeps producing the buckets in front of it.  ``finish()`` makes the
    compute stream wait for all of them and returns the world size.  Results equal ``allreduce_sum_flat`` bit for bit."""

    def __init__(self, n_buckets: int = 8, align: int = 1024):
        self.n_buckets, self.align = max(1, n_buckets), align
        self._key = None
        self.works, self.fired = [], []

    def begin(self, grad: torch.Tensor, layout, first_use, n_tape: int):
        world = dist.get_world_size() if dist.is_initialized() else 1
        key = (grad.data_ptr(), grad.numel(), world)
        if self._key != key:  # the cut and the scratch slices are fixed for a given buffer
            n = grad.numel()
            step = max(self.align * world, -(-n // self.n_buckets))
            step = -(-step // (self.align * world)) * (self.align * world)
            self.ranges = [(a, min(n, a + step)) for a in range(0, n, step)]
            self.scratch = torch.empty(step // world + 1, dtype=grad.dtype, device=grad.device) if world > 1 else None
            self._key = key
        self.grad, self.world = grad, world
        # a bucket is final once the EARLIEST tape entry among its parameters has run (entries run from n_tape - 1 down to 0);
        # parameters no entry touches keep a zero gradient and are final from the start (index n_tape)
        ready = [n_tape] * len(self.ranges)
        for name, (off, shape) in layout.items():
            numel = 1
            for d in shape:
                numel *= d
            fu = first_use.get(name, n_tape)
            for b, (a, e) in enumerate(self.ranges):
                if off < e and off + numel > a:
                    ready[b] = min(ready[b], fu)
        self.ready, self.works, self.fired = ready, [], []
        self.fire_indices = set(ready)  # tape indices after which some bucket's exchange is launched (Graph.backward joins its side stream there)

    def entry_done(self, index: int):
        for b, r in enumerate(self.ready):
            if r == index:
                self.fired.append(b)
                if self.world > 1:
                    a, e = self.ranges[b]
                    # one scratch slice is enough: the process group runs its collectives in issue order on one stream
                    self.works += _exchange_sum(self.grad[a:e], self.scratch, async_op=dist.get_backend() != "gloo")

    def finish(self) -> int:
        assert len(self.fired) == len(self.ranges), f"buckets never fired: {sorted(set(range(len(self.ranges))) - set(self.fired))}"
        for w in self.works:
            w.wait()  # NCCL/RCCL: the current stream waits for the collective's event (no host sync)
        self.works = []
        return self.world


class AbiComm:
    """The same exchange through the C ABI (``gn_comm_*`` in include/genima_hip.h: RCCL reduce-scatter + all-gather on the library's own
    HIP stream, optional bf16 wire), for hosts that do not carry torch.distributed.  The RCCL unique id travels through whatever
    channel the caller has (here: torch.distributed's store when it is up; a single rank needs none)."""

    def __init__(self, engine, rank: int = 0, world: int = 1, bf16_wire: bool = False):
        import ctypes as C

        from ._lib import check
        self.E, self.lib, self.world, self.bf16 = engine, engine.lib, world, bf16_wire
        uid = (C.c_ubyte * 128)()
        if rank == 0:
            check(self.lib.gn_comm_unique_id(uid), "gn_comm_unique_id")
        if world > 1:
            obj = [bytes(uid)]
            dist.broadcast_object_list(obj, src=0)
            uid = (C.c_ubyte * 128).from_buffer_copy(obj[0])
        self._comm = C.c_void_p()
        check(self.lib.gn_comm_init(engine._ctx, rank, world, uid, C.byref(self._comm)), "gn_comm_init")
        self._scratch = None

    def __call__(self, buf: torch.Tensor) -> int:
        from ._lib import check
        need = int(self.lib.gn_comm_scratch_bytes(self._comm, buf.numel(), int(self.bf16)))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(max(need, 256), dtype=torch.uint8, device=buf.device)
        check(self.lib.gn_comm_allreduce_grads(self._comm, buf.data_ptr(), buf.numel(), int(self.bf16), self._scratch.data_ptr()),
              "gn_comm_allreduce_grads")
        check(self.lib.gn_comm_wait(self._comm), "gn_comm_wait")  # the compute stream waits for the exchange (no host sync)
        return self.world

    def __del__(self):
        try:
            if self._comm:
                self.lib.gn_comm_destroy(self._comm)
        except Exception:
            pass
