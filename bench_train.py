#!/usr/bin/env python
"""ControlNet fine-tune benchmark on MI355X: BASELINE.json's second metric, "ControlNet train steps/sec".

    python bench_train.py --gpus N --steps K --warmup W [--batch 8] [--family sd-turbo]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench_train.py --gpus N ...

Workload = BASELINE.json configs[3]: train_controlnet_genima.py SD-Turbo fine-tune at 512x512 (4 tiled 256x256 views), per-GPU
batch 8 (global batch 64 on 8 GPUs), data parallel with one RCCL all-reduce of the flat 1.46 GB fp32 gradient buffer per step.
A "step" is the whole step body of diffusion/train_controlnet_genima.py:1317-1408 on one synthetic batch already resident in HBM:
device-side augmentation (colour jitter + shared reflect-pad crop, README recipe), VAE encode + posterior sample, noise / timestep
sampling, CLIP text encode, ControlNet forward, frozen UNet forward, MSE, backward
(ControlNet dX + dW, UNet decoder dX), gradient all-reduce, unscale + global-norm clip, AdamW, f16 weight refresh, zero_grad.
Weights are seeded random-init tensors of the full SD-Turbo architecture (no checkpoints offline); the ControlNet uses random
(non-zero) output convs so that every gradient path carries real work.  Same JSON-line contract as bench.py (rank 0 prints it).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process GPU work on these hosts needs dmabuf IPC (RCCL / device-tensor sharing fail with hipIpcGetMemHandle otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

MFMA_PEAK_TF = 2500.0
# algorithmic TFLOP per sample: forward 2.25 (VAE-enc 1.13, CN 0.33, UNet 0.80 at 64x64 latents) + backward 1.37
TFLOP_PER_SAMPLE = {("sd-turbo", 512): 2.25 + 1.37,
                    # SDXL ControlNet fine-tune at 512x512: forward 3.42 + backward 3.01 TFLOP per sample
                    ("sdxl-turbo", 512): 3.42 + 3.01}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (configs[3]: 64 / 8 GPUs)")
    ap.add_argument("--resolution", type=int, default=512)
    ap.add_argument("--family", default="sd-turbo")
    ap.add_argument("--lr", type=float, default=1e-5)
    ap.add_argument("--augmentations", default="crop,colorjitter", help="the reference's --augmentations list (README.md:204); '' disables")
    ap.add_argument("--fp8", action="store_true", help="frozen UNet transformer Linears on the fp8 MFMA in the forward pass (configs[4])")
    ap.add_argument("--exchange", default="buckets", choices=["buckets", "flat", "abi", "abi-bf16"],
                    help="gradient exchange: bucketed RS+AG overlapped with the backward (default), one RS+AG after it, or the C-ABI "
                         "RCCL communicator (gn_comm_*; -bf16: bf16 on the wire)")
    ap.add_argument("--graph", action="store_true", help="capture the forward + backward walk into one hipGraph after two eager steps and replay it")
    ap.add_argument("--gemm-table", default=None, help="write the per-(shape, tile) HIP-event GEMM timing table of one extra step to this CSV")
    return ap.parse_args(argv)


def run(args, quiet: bool = False):
    """One measurement; returns the JSON-line dict on rank 0 (None elsewhere).  bench.py calls this for its ``train`` extra key."""
    from genima_amd import configs, dist, schema, weights
    from genima_amd.engine import Engine, save_tune_table
    from genima_amd.packing import pack_state_dict
    from genima_amd.scheduler import DDPMScheduler
    from genima_amd.training import ControlNetTrainer

    rank, local, world = dist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    fam = configs.family(args.family)
    E = Engine(dev, autotune=True)

    def synth(sch, seed):
        return weights.synth_state_dict(sch, seed, device=dev)

    unet_W = pack_state_dict(synth(schema.unet_schema(fam["unet"]), 1), dev, up_phases=False)  # the tape runs the fused-upsample 3x3 launch
    vae_W = pack_state_dict(synth(schema.vae_schema(fam["vae"]), 3), dev)
    text_W = pack_state_dict(synth(schema.clip_text_schema(fam["text"]), 4), dev)
    cn_sd = synth(schema.controlnet_schema(fam["controlnet"]), 2)
    exchange = None
    if world > 1:
        exchange = {"buckets": lambda: dist.GradBuckets(n_buckets=8), "flat": lambda: dist.allreduce_sum_flat,
                    "abi": lambda: dist.AbiComm(E, rank, world), "abi-bf16": lambda: dist.AbiComm(E, rank, world, bf16_wire=True)}[args.exchange]()
    tr = ControlNetTrainer(E, fam["unet"], fam["controlnet"], unet_W, cn_sd, lr=args.lr, allreduce=exchange,
                           hip_graph=bool(args.graph) and not args.gemm_table)
    del cn_sd
    text2_W = pack_state_dict(synth(schema.clip_text_schema(fam["text_2"]), 5), dev) if "text_2" in fam else None
    tr.attach_frozen(fam["vae"], vae_W, fam["text"], text_W, DDPMScheduler(), seed=1234 + rank,
                     text2_cfg=fam.get("text_2"), text2_W=text2_W, augmentations=args.augmentations or None)
    n_fp8 = tr.enable_fp8_frozen() if args.fp8 else 0

    B, R = args.batch, args.resolution
    g = torch.Generator(device=dev).manual_seed(77 + rank)
    V = fam["text"]["vocab_size"]
    ids = torch.zeros(B, 77, dtype=torch.int32)
    ids[:, :14] = torch.tensor([V - 2] + [320 + i for i in range(12)] + [V - 1], dtype=torch.int32)
    px = torch.zeros(B, R, R, 8, dtype=torch.float16, device=dev)
    px[..., :3] = (torch.rand(B, R, R, 3, generator=g, device=dev) * 2 - 1).half()
    cond = torch.zeros(B, R, R, 8, dtype=torch.float16, device=dev)
    cond[..., :3] = torch.rand(B, R, R, 3, generator=g, device=dev).half()
    batch = dict(pixel_values=px, conditioning_pixel_values=cond, input_ids=ids.to(dev))

    losses = []
    for _ in range(args.warmup):
        losses.append(tr.train_step(batch))
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(tr.train_step(batch))
    torch.cuda.synchronize()
    dist.barrier()
    dt = dist.max_over_ranks(time.perf_counter() - t0, device=dev)
    if args.gemm_table and rank == 0:  # one extra, untimed step with every GEMM launch bracketed by HIP events
        E.gemm_log = []
        tr.train_step(batch)
        rows = sorted(E.gemm_log_report().items(), key=lambda kv: -kv[1][1])
        E.gemm_log = None
        os.makedirs(os.path.dirname(os.path.abspath(args.gemm_table)), exist_ok=True)
        with open(args.gemm_table, "w") as f:
            f.write("conv|M|N|K|C1|C2|KH|stride|up2x|act|out_mode|residual[|b<batch>][|acc]|t<tile>,calls,total_ms,tflops\n")
            for k, (calls, ms, tf) in rows:
                f.write(f"{k},{calls},{ms:.4f},{tf:.1f}\n")
    if rank == 0:
        save_tune_table()
        ms = dt / args.steps * 1e3
        tf = TFLOP_PER_SAMPLE.get((args.family, R))
        achieved = tf * B * args.steps / dt if tf else None
        line = {
            "metric": "ControlNet train steps/sec", "value": args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16 (+ fp8 e4m3 forward Linears of the frozen UNet)" if args.fp8 else "f16",
            "data": f"synthetic (seeded random-init {args.family} weights, uniform random images, fixed 14-token prompt)",
            "samples_per_sec": B * world * args.steps / dt,
            "config": {"workload": ("BASELINE.json configs[4]: SDXL-Turbo ControlNet fine-tune, 512x512 (4x256x256 tiled views)"
                                    if args.family == "sdxl-turbo" else
                                    "BASELINE.json configs[3]: SD-Turbo ControlNet fine-tune, 512x512 (4x256x256 tiled views)"),
                       "family": args.family,
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": f"dp{world}", "gradient_exchange": args.exchange if world > 1 else None,
                       "hip_graph": bool(tr._graphs), "optimizer": "AdamW fp32 master, f16 compute, loss scale",
                       "trainable_params_padded": int(tr.cn.numel), "fp8_frozen_linears": n_fp8},
            "loss_first": float(losses[0]), "loss_last": float(losses[-1]), "grad_norm_last": tr.last.get("grad_norm"),
            "loss_scale": tr.loss_scale, "applied_steps": tr.opt_step,
            "roofline": ({"bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TF,
                          "frac_of_sustained_mfma_rate": achieved / 1680.0,  # what a pure MFMA stream holds
                          "traffic": None, "note": f"whole step, algorithmic FLOPs of SURVEY.md section 8 ({tf} TFLOP per sample)"}
                         if achieved else None),
            "peak_mem_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
        }
        return line
    return None


def main():
    args = parse_args()
    from genima_amd.dist import maybe_self_launch

    maybe_self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus)  # --gpus N > 1 without a launcher: become the launcher
    line = run(args)
    if line is not None:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
