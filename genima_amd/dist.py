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
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
    return rank, local, world


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
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_mean_flat(buf: torch.Tensor) -> torch.Tensor:
    """In-place mean over ranks of a flat gradient buffer as reduce-scatter + all-gather (numel padded to the world size by
    the caller or handled here through a tail all-reduce).  Bit-identical on every rank afterwards."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return buf
    world = dist.get_world_size()
    n = buf.numel()
    main = n - n % world
    if main:
        chunks = buf[:main].view(world, main // world)
        mine = torch.empty_like(chunks[0])
        if dist.get_backend() == "gloo":  # gloo has no reduce_scatter: same result through all_reduce
            dist.all_reduce(chunks, op=dist.ReduceOp.SUM)
        else:
            dist.reduce_scatter_tensor(mine, buf[:main], op=dist.ReduceOp.SUM)
            dist.all_gather_into_tensor(buf[:main], mine)
    if n != main:
        dist.all_reduce(buf[main:], op=dist.ReduceOp.SUM)
    buf.div_(world)
    return buf
