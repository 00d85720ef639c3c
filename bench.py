#!/usr/bin/env python
"""Genima hot-path benchmark on MI355X (driver contract: one JSON line from rank 0).

    python bench.py --gpus N --steps K --warmup W [--workload tiled_b8|single_b1] [--graph]

A "step" is ONE pipeline call of the hot path (metric 1): CLIP text encode -> ControlNet cond-embedding ->
5 x (ControlNet + UNet + Euler step) -> VAE decode -> uint8 post-process, on a batch of synthetic tiled observations with
seeded random-init weights of the full SD-Turbo architecture (no checkpoints / datasets exist offline).  The default workload
is BASELINE.json configs[2] -- the configuration the metric is quoted on ("SD-Turbo 256x256, 5 steps, 4 views": 4-view tiled
512x512, batch = 8 episodes, one GPU).  metric = joint-target images/sec = 4*B*calls / time, whole job over all ranks; inputs
(token ids, uint8 control images, unit-variance latents) are resident in HBM when the timed region starts and the uint8
result stays in HBM (`value`); the D->H-inclusive rate of the reference's `gen_time` bracket is reported beside it.

N > 1: one process per GPU under torch.distributed.run; inference shards episodes across ranks with no data-path collective
(replicas), so scaling is "weak" (per-GPU batch fixed).

Extra objects on the JSON line (rank 0, N = 1 only): `roofline` (dominant kernel family: the MFMA implicit-GEMM conv /
linear kernel; per-op HIP-event timing of one replay of the recorded program), `roofline_extra` (attention: MFMA; GroupNorm
+SiLU: HBM) and `cpu_baseline` (the fp32 torch-CPU oracle timed on this box's host cores on a bounded sample).
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

import torch  # noqa: E402

MFMA_PEAK_TF = 2500.0   # dense f16/bf16, MI355X_MICROARCH.md
MFMA_SUSTAINED_TF = 1680.0  # measured: what a pure MFMA stream holds at the clock the chip sustains
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (B per GPU, H, W, description)
    "tiled_b8": (8, 512, 512, "BASELINE.json configs[2]: SD-Turbo + ControlNet 4-view tiled 512x512, 5 steps, batch=8 episodes"),
    "tiled_b1": (1, 512, 512, "SD-Turbo + ControlNet 4-view tiled 512x512, 5 steps, batch=1 episode"),
    "single_b1": (1, 256, 256, "BASELINE.json configs[1]: SD-Turbo + ControlNet 256x256 single view, 5 steps, batch=1"),
}
# algorithmic GFLOP per sample-call: CLIP + 5*(CN + UNet) + VAE decode
GFLOP_PER_CALL = {512: 8000.0, 256: 1888.0}


PMC_TRAFFIC_FILE = "r01_v7_pmc_traffic_tiled_b8.json"


def pmc_traffic_per_launch(family: str):
    """HBM bytes per launch of a kernel family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs,
    collected as MI355X_MICROARCH.md prescribes).  None when the file is missing."""
    path = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)  # tools/probes/pmc_traffic.sh re-collects it
    try:
        with open(path) as f:
            d = json.load(f)
        fe, wr = d["FETCH_SIZE"][family], d["WRITE_SIZE"][family]
        return (2.0 * fe["sum_counter"] / fe["launches"] + wr["sum_counter"] / wr["launches"]) * 1024.0
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


def synthetic_inputs(pipe, B, H, W, device, rank):
    from genima_amd import weights

    V = pipe.text_encoder.config["vocab_size"]
    ids = torch.zeros(B, 77, dtype=torch.int32)
    ids[:, :14] = torch.tensor([V - 2] + [320 + i for i in range(12)] + [V - 1], dtype=torch.int32)
    img = torch.from_numpy(weights.counter_bytes(100 + rank, "bench_ctrl", B * H * W * 3).reshape(B, H, W, 3))
    g = torch.Generator().manual_seed(2 + rank)  # diffusion_seed = 2 (controller/cfgs/eval_genima.yaml:32)
    lat = torch.randn(B, 4, H // 8, W // 8, generator=g).to(torch.float16)
    return ids.to(device), img.to(device), lat.to(device)


def per_op_profile(pipe, io, dump=None):
    """Replay the recorded program once op by op with HIP events on the engine's stream; aggregate by kernel family."""
    E = io.engine
    n = E.num_ops
    evs = [E.event() for _ in range(n + 1)]
    E.synchronize()
    E.event_record(evs[0])
    for i in range(n):
        E.run(i, i + 1)
        E.event_record(evs[i + 1])
    E.synchronize()
    agg = {}
    by_shape = {}
    for i, m in enumerate(E.meta[:n]):
        ms = E.event_elapsed_ms(evs[i], evs[i + 1])
        s = by_shape.setdefault((m["kind"], tuple(m["shape"])), dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
        s["ms"] += ms
        s["flops"] += m["flops"]
        s["bytes"] += m["bytes"]
        s["launches"] += 1
        a = agg.setdefault(m["kind"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
        a["ms"] += ms
        a["flops"] += m["flops"]
        a["bytes"] += m["bytes"]
        a["launches"] += 1
    for ev in evs:
        E.lib.gn_event_destroy(ev)
    if dump:
        with open(dump, "w") as f:
            f.write("kind,shape,launches,total_ms,avg_us,TFLOP/s,GB/s\n")
            for (k, shp), a in sorted(by_shape.items(), key=lambda kv: -kv[1]["ms"]):
                f.write(f"{k},{'x'.join(map(str, shp))},{a['launches']},{a['ms']:.3f},{1000 * a['ms'] / a['launches']:.1f},"
                        f"{a['flops'] / (a['ms'] * 1e-3) / 1e12:.1f},{a['bytes'] / (a['ms'] * 1e-3) / 1e9:.0f}\n")
    return agg


def host_threads() -> int:
    """Threads the CPU baseline may use: the scheduler affinity, capped by the cgroup CPU quota (the GPU boxes expose 256
    logical CPUs but cap the container at cpu.max = 16 CPUs; oversubscribing made torch 30x slower.  2 threads per quota CPU
    measured best: tools/probe_cpu.py)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, 2 * int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(max_seconds=30.0):
    """fp32 torch-CPU oracle on a bounded sample of the same workload: one denoise step (ControlNet + UNet forward) at full
    SD-Turbo width, B=1, latent 32x32 (one 256x256 view) -- 244.2 GFLOP algorithmic.  Scaled to the metric's unit by FLOPs:
    a 5-step tiled call is 8000 GFLOP per 4 joint-target images."""
    from genima_amd import configs, schema, weights
    from oracle import sd_torch as O

    cores = host_threads()
    torch.set_num_threads(cores)
    fam = configs.family("sd-turbo")
    t0 = time.time()
    gdev = "cuda" if torch.cuda.is_available() else "cpu"  # draw on the GPU (same bits as the numpy path), copy to host
    usd = {k: v.cpu() for k, v in weights.synth_state_dict(schema.unet_schema(fam["unet"]), 21, device=gdev).items()}
    csd = {k: v.cpu() for k, v in weights.synth_state_dict(schema.controlnet_schema(fam["controlnet"]), 22, device=gdev).items()}
    gen_s = time.time() - t0
    g = torch.Generator().manual_seed(0)
    x, ctx = torch.randn(1, 4, 32, 32, generator=g), torch.randn(1, 77, 1024, generator=g)
    cond, t = torch.rand(1, 3, 256, 256, generator=g), torch.tensor([999.0])
    times = []
    with torch.no_grad():
        while sum(times) < max_seconds * 0.5 and len(times) < 3:
            t0 = time.time()
            down, mid = O.controlnet_forward(csd, fam["controlnet"], x, t, ctx, cond)
            O.unet_forward(usd, fam["unet"], x, t, ctx, down, mid)
            times.append(time.time() - t0)
    best = min(times)
    gflop = 181.1 + 63.1
    gfs = gflop / best
    img_s = gfs / (8000.0 / 4.0)
    return {"value": img_s, "unit": "joint-target images/sec", "cores": cores, "kind": "port",
            "sample": f"1 denoise step (ControlNet+UNet fwd, {gflop:.1f} GFLOP) B=1 latent 32x32, fp32 torch-CPU oracle at full "
                      f"SD-Turbo width, best of {len(times)}: {best:.2f} s = {gfs:.0f} GFLOP/s on {cores} threads; scaled by "
                      f"FLOPs to the 8000-GFLOP 5-step tiled call (weights drawn in {gen_s:.0f} s, untimed)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5, help="timed pipeline calls")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="tiled_b8", choices=sorted(WORKLOADS))
    ap.add_argument("--denoise-steps", type=int, default=5)
    ap.add_argument("--family", default="sd-turbo")
    ap.add_argument("--graph", action="store_true", help="replay each call as one captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-act", action="store_true", help="skip the ACT controller forward after each pipeline call")
    ap.add_argument("--dump-ops", default=None, help="write the per-(kernel, shape) HIP-event timing table of one call to this CSV")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert world == max(1, args.gpus), f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from genima_amd import configs
    from genima_amd.pipeline import StableDiffusionControlNetPipeline

    B, H, W, desc = WORKLOADS[args.workload]
    pipe = StableDiffusionControlNetPipeline.from_synthetic(configs.family(args.family), seed=0, gen_device=dev)
    pipe.to(dev)
    for m in (pipe.vae, pipe.text_encoder, pipe.unet, pipe.controlnet):
        m._sd = None  # fp32 masters are not needed for inference; keep only the packed f16 copy resident
    torch.cuda.empty_cache()
    pipe.enable_hip_graph(args.graph)
    ids, img, lat = synthetic_inputs(pipe, B, H, W, dev, rank)

    # ACT controller forward on the generated joint-target images (BASELINE.json configs[2]: "... + ACT controller forward")
    act_agent = None
    if H == 512 and not args.no_act:
        from genima_amd.act import GenimaACT

        fam = configs.family(args.family)
        act_agent = GenimaACT(fam["act"], None, fam["act_text"], None, device=dev, seed=0)
        g = torch.Generator().manual_seed(7 + rank)
        act_state = torch.randn(B, 1, fam["act"]["state_dim"], generator=g).to(dev)
        Vc = fam["act_text"]["vocab_size"]
        act_tokens = torch.zeros(B, 1, 77, dtype=torch.int32)
        act_tokens[:, 0, :14] = torch.tensor([Vc - 2] + [320 + i for i in range(12)] + [Vc - 1], dtype=torch.int32)
        act_tokens = act_tokens.to(dev)

    def call(output_type="pt"):
        out = pipe(prompt_ids=ids, image=img, latents=lat, num_inference_steps=args.denoise_steps, guidance_scale=0.0,
                   output_type="pt" if act_agent is not None else output_type)
        if act_agent is not None:
            actions = act_agent.act_tiled(out.images, act_state, act_tokens)  # [B, 20, 8] on the device
            if output_type == "np":  # the reference's brackets: images to host PIL (gen_time) + actions to host (control_time)
                return out.images.cpu().numpy(), actions.float().cpu().numpy()
            return actions
        return out

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        call()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        call()
    barrier()
    dt = time.perf_counter() - t0
    from genima_amd.dist import max_over_ranks

    dt = max_over_ranks(dt, dev)  # the slowest rank defines the job's time
    calls = args.steps
    value = 4.0 * B * world * calls / dt

    # D->H-inclusive bracket (the reference's gen_time: ... -> uint8 -> host PIL), rank-local, informational
    barrier()
    t1 = time.perf_counter()
    for _ in range(max(1, min(3, calls))):
        call("np")
    torch.cuda.synchronize(dev)
    dt_host = (time.perf_counter() - t1) / max(1, min(3, calls))

    act_ms = None
    if act_agent is not None:  # controller forward alone (the reference's control_time bracket minus the D->H copy)
        tiled = pipe(prompt_ids=ids, image=img, latents=lat, num_inference_steps=args.denoise_steps, guidance_scale=0.0,
                     output_type="pt").images
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        for _ in range(5):
            act_agent.act_tiled(tiled, act_state, act_tokens)
        torch.cuda.synchronize(dev)
        act_ms = (time.perf_counter() - t2) / 5 * 1000.0

    out = {
        "metric": "joint-target images/sec (SD-Turbo + ControlNet, 4-view tiled 512x512, 5 steps, incl. CLIP text + VAE decode)"
        if H == 512 else "images/sec (SD-Turbo + ControlNet 256x256 single view, 5 steps, incl. CLIP text + VAE decode)",
        "value": value, "unit": "images/sec", "n_gpus": world, "steps": calls, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / calls, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (f32 accumulate)", "data": "synthetic (seeded random-init SD-Turbo-architecture weights, counter-PRNG images)",
        "config": {"workload": desc, "family": args.family, "per_gpu_batch": B, "global_batch": B * world,
                   "image": f"{H}x{W}", "denoise_steps": args.denoise_steps, "parallelism": f"replicas x{world} (episodes sharded, no collective)",
                   "hip_graph": bool(args.graph), "act_controller_forward": act_agent is not None,
                   "two_streams": bool(pipe.two_streams)},
        "images_per_sec_per_gpu": value / world,
        "act_controller_ms_per_call": act_ms,
        "value_incl_d2h_to_host": (4.0 if H == 512 else 1.0) * B / dt_host,
        "algorithmic_tflops_per_gpu": GFLOP_PER_CALL.get(H, 0.0) * B * calls / dt / 1000.0,
    }
    if H != 512:
        out["value"] = B * world * calls / dt
        out["images_per_sec_per_gpu"] = out["value"] / world

    if rank == 0 and world == 1 and not args.no_roofline:
        io = pipe.program(B, H, W, args.denoise_steps)
        if args.graph:
            io.engine.use_stream(io.stream)
        agg = per_op_profile(pipe, io, args.dump_ops)
        gemm_kinds = [k for k in agg if k.startswith("conv") or k == "linear"]
        g_ms = sum(agg[k]["ms"] for k in gemm_kinds)
        g_fl = sum(agg[k]["flops"] for k in gemm_kinds)
        n_l = sum(agg[k]["launches"] for k in gemm_kinds)
        ach = g_fl / (g_ms * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_dma_kernel / gemm_kernel<BM,BN,WM,WN,CONV> (MFMA implicit-GEMM conv3x3/1x1 + Linear)",
                           "achieved": ach, "peak": MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TF,
                           "traffic": pmc_traffic_per_launch("gemm") if args.workload == "tiled_b8" else None,
                           "traffic_note": "HBM bytes per launch of this kernel family from the committed rocprofv3 --pmc passes of this "
                                           f"workload (profiles/{PMC_TRAFFIC_FILE}: 2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes; "
                                           "the x2 is MI355X_MICROARCH.md's gfx950 FETCH_SIZE correction); not re-collected by this run",
                           "frac_of_sustained_mfma_rate": ach / MFMA_SUSTAINED_TF,
                           "sustained_note": f"a pure v_mfma_f32_32x32x16_f16 stream sustains {MFMA_SUSTAINED_TF:.0f} TFLOP/s on MI355X "
                                             "(19-21 ns per MFMA per SIMD at the ~1.6-1.7 GHz the chip holds under matrix load; "
                                             "tools/probes/mfma_coissue.hip); `frac` stays quoted against the nominal dense peak",
                           "launches_per_call": n_l, "avg_launch_ms": g_ms / max(1, n_l), "algorithmic_gflop_per_call": g_fl / 1e9,
                           "algorithmic_bytes_per_launch": sum(agg[k]["bytes"] for k in gemm_kinds) / max(1, n_l),
                           "share_of_call_time": g_ms / sum(a["ms"] for a in agg.values())}
        extra = []
        for k in sorted(agg, key=lambda k: -agg[k]["ms"]):
            a = agg[k]
            row = {"kernel": k, "ms_per_call": a["ms"], "launches": a["launches"]}
            if a["flops"] > 0:
                tf = a["flops"] / (a["ms"] * 1e-3) / 1e12
                row.update(bound="mfma", achieved=tf, peak=MFMA_PEAK_TF, unit="TFLOP/s", frac=tf / MFMA_PEAK_TF)
            else:
                gbs = a["bytes"] / (a["ms"] * 1e-3) / 1e9
                row.update(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS)
            extra.append(row)
        out["roofline_extra"] = extra
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
