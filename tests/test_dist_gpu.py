"""-m gpu: the data-parallel fine-tune path on ONE GPU.
  * RCCL for real at world size 1: torch.distributed's nccl backend (= RCCL) runs the reduce-scatter + all-gather branch of
    genima_amd/dist.py, and the C-ABI communicator (gn_comm_*, csrc/comm.hip) runs its own RCCL reduce-scatter + all-gather on its
    side stream, f32 and bf16 wire;
  * two PROCESSES sharing the GPU (gloo carries the device tensors; RCCL refuses two ranks on one device): each rank trains on its
    own half of a batch with the bucketed, backward-overlapped exchange -- afterwards both ranks hold bit-identical weights, equal
    (to summation-order rounding) to a single process trained on the whole batch.
Reference: accelerate DDP under accelerator.backward, diffusion/train_controlnet_genima.py:1216-1218, :1402-1408."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_rccl_branch_and_abi_comm_at_world_size_one():
    from genima_amd import dist as gd
    from genima_amd.engine import Engine

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.randn(1_000_003, generator=g, device="cuda")  # odd length: the tail path too
        y = x.clone()
        assert gd.allreduce_sum_flat(y, force=True) == 1  # reduce_scatter_tensor + all_gather_into_tensor on RCCL
        torch.cuda.synchronize()
        assert torch.equal(x, y)
        E = Engine("cuda:0")
        for bf16 in (False, True):
            comm = gd.AbiComm(E, rank=0, world=1, bf16_wire=bf16)
            z = x.clone()
            assert comm(z) == 1
            torch.cuda.synchronize()
            if bf16:  # the sum travelled as bf16: values come back rounded to bf16
                assert torch.equal(z, x.to(torch.bfloat16).float())
            else:
                assert torch.equal(z, x)
            del comm
    finally:
        dist.destroy_process_group()


def _setup(seed=0):
    from genima_amd import configs, schema, weights
    from genima_amd.scheduler import DDPMScheduler

    fam = configs.family("tiny")
    usd = weights.round_to(weights.synth_state_dict(schema.unet_schema(fam["unet"]), 1), torch.float16)
    csd = weights.round_to(weights.synth_state_dict(schema.controlnet_schema(fam["controlnet"]), 2), torch.float16)
    g = torch.Generator().manual_seed(seed)
    B, h = 2, 32
    lat = torch.zeros(B, h, h, 8).half()
    lat[..., :4] = (torch.randn(B, h, h, 4, generator=g) * 0.8).half()
    noi = torch.zeros(B, h, h, 8).half()
    noi[..., :4] = torch.randn(B, h, h, 4, generator=g).half()
    ctx = (torch.randn(B, 77, 128, generator=g) * 0.5).half()
    cond = torch.zeros(B, 8 * h, 8 * h, 8).half()
    cond[..., :3] = torch.rand(B, 8 * h, 8 * h, 3, generator=g).half()
    t = torch.tensor([801, 399])
    sa, s1 = DDPMScheduler().add_noise_coeffs(t)
    return fam, usd, csd, (lat, noi, t.float(), sa, s1, ctx, cond)


def _train(rows, allreduce, steps=2):
    from genima_amd.engine import Engine
    from genima_amd.packing import pack_state_dict
    from genima_amd.training import ControlNetTrainer

    fam, usd, csd, batch = _setup()
    tr = ControlNetTrainer(Engine("cuda:0"), fam["unet"], fam["controlnet"], pack_state_dict(usd, "cuda"), csd, lr=1e-4, loss_scale=1024.0,
                           allreduce=allreduce)
    args = [a[rows].cuda() for a in batch]
    losses = [float(tr.step(*args).cpu()) for _ in range(steps)]
    torch.cuda.synchronize()
    return tr, losses


def _rank_worker(rank, world, port, q):
    from genima_amd import dist as gd

    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    gd.init_from_env("gloo")
    buckets = gd.GradBuckets(n_buckets=6)
    tr, losses = _train(slice(rank, rank + 1), buckets)  # rank r trains on sample r
    fired = list(buckets.fired)
    q.put((rank, losses, tr.cn.master.cpu().numpy(), tr.last["grad_norm"], fired, tr.world))  # numpy: pickled by value
    gd.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_single_process_step():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, l0, m0, n0, f0, w0), (_, l1, m1, n1, f1, w1) = res
    m0, m1 = torch.from_numpy(m0), torch.from_numpy(m1)
    assert w0 == w1 == 2 and sorted(f0) == list(range(len(f0))) and len(f0) >= 2, (f0, f1)
    assert f0[0] != 0, "the LAST bucket of the flat buffer (zero convs / mid block) must be exchanged before the first (conv_in, time MLP)"
    assert torch.equal(m0, m1) and n0 == n1, "after the exchange every rank applies the same update: weights stay bit-identical"
    ref, lref = _train(slice(0, 2), None)  # one process, both samples
    mref = ref.cn.master.cpu()
    # the mean of the two per-rank losses is the batch loss; the weights agree to the rounding of a different summation order
    assert abs(0.5 * (l0[0] + l1[0]) - lref[0]) <= 1e-4 * lref[0]
    assert abs(n0 - ref.last["grad_norm"]) <= 2e-3 * ref.last["grad_norm"]
    upd = float((mref - m0).abs().mean())
    moved = float((mref - _initial_master()).abs().mean())
    assert moved > 0 and upd <= 0.03 * moved, (upd, moved)


def _initial_master():
    from genima_amd.engine import Engine
    from genima_amd.training import TrainParams

    _, _, csd, _ = _setup()
    return TrainParams(Engine("cuda:0"), csd).master.cpu()
