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
# multi-process GPU work on these hosts needs dmabuf IPC (RCCL / device-tensor sharing fail with hipIpcGetMemHandle otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

MFMA_PEAK_TF = 2500.0   # dense f16/bf16, MI355X_MICROARCH.md
MFMA_SUSTAINED_TF = 1680.0  # measured: what a pure MFMA stream holds at the clock the chip sustains
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    # name: (B per GPU, H, W, description)
    "tiled_b8": (8, 512, 512, "BASELINE.json configs[2]: SD-Turbo + ControlNet 4-view tiled 512x512, 5 steps, batch=8 episodes"),
    "tiled_b4": (4, 512, 512, "SD-Turbo + ControlNet 4-view tiled 512x512, 5 steps, batch=4 episodes"),
    "tiled_b2": (2, 512, 512, "SD-Turbo + ControlNet 4-view tiled 512x512, 5 steps, batch=2 episodes"),
    "tiled_b1": (1, 512, 512, "SD-Turbo + ControlNet 4-view tiled 512x512, 5 steps, batch=1 episode"),
    "single_b1": (1, 256, 256, "BASELINE.json configs[1]: SD-Turbo + ControlNet 256x256 single view, 5 steps, batch=1"),
}
# algorithmic GFLOP per sample-call: CLIP + 5*(CN + UNet) + VAE decode
GFLOP_PER_CALL = {512: 8000.0, 256: 1888.0}


def _pmc_traffic_file():
    """The newest committed PMC traffic summary (profiles/rNN_vM_pmc_traffic_tiled_b8.json; tools/probes/pmc_traffic.sh writes them and
    stamps the commit + the library hash they were collected on)."""
    import glob
    import re

    def key(p):
        m = re.search(r"r(\d+)_v(\d+)_pmc_traffic_tiled_b8", p)
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_tiled_b8.json")), key=key)
    return os.path.basename(files[-1]) if files else "none"


PMC_TRAFFIC_FILE = _pmc_traffic_file()


def pmc_traffic_per_launch(family: str, launches_per_call=None):
    """HBM bytes per launch of a kernel family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs,
    collected as MI355X_MICROARCH.md prescribes).  None when the file is missing -- or when its stamp counts a different number of
    launches per call than the program this run replays (``launches_per_call``): the counters then belong to another lowering."""
    path = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)  # tools/probes/pmc_traffic.sh re-collects it
    try:
        with open(path) as f:
            d = json.load(f)
        fe, wr = d["FETCH_SIZE"][family], d["WRITE_SIZE"][family]
        stamped = d.get("stamp", {}).get("gemm_ops_per_call")
        if family == "gemm" and launches_per_call is not None and stamped is not None and abs(stamped - launches_per_call) > 0.5:
            return None
        # per OP of the recorded program (`launches_per_call` counts those: a split-K gn_gemm's reduce launch belongs to its op)
        return (2.0 * fe["sum_counter"] / fe.get("ops", fe["launches"]) + wr["sum_counter"] / wr.get("ops", wr["launches"])) * 1024.0
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        return None


def pmc_traffic_stamp():
    """{"commit", "lib_sha16"} of the PMC passes and whether that library is the one loaded now."""
    import hashlib

    try:
        with open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)) as f:
            st = json.load(f).get("stamp", {})
        with open(os.path.join(ROOT, "genima_amd", "libgenima_hip.so"), "rb") as f:
            st["same_library_as_this_run"] = hashlib.sha256(f.read()).hexdigest()[:16] == st.get("lib_sha16")
        # a rebuilt .so differs byte-wise from box to box; the hash of csrc/ + compile flags names the code itself
        from genima_amd.build import source_sha16

        st["same_sources_as_this_run"] = source_sha16() == st.get("src_sha16")
        return st
    except OSError:
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
        a = agg.setdefault(m["kind"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, ref_flops=0.0))
        a["ms"] += ms
        a["ref_flops"] += m.get("ref_flops", m["flops"])
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


def cpu_baseline(max_seconds=40.0):
    """The fp32 torch-CPU oracle (oracle/sd_torch.py, the restatement of the reference's diffusers path -- diffusers itself is not
    installable here) timed on the pieces of ONE tiled sample's 5-step call, each at the size the call runs it:
      CLIP-H text tower (B = 1)  +  5 x [ControlNet + UNet at the 64x64 latent of a tiled 512x512 sample]  +  VAE decode.
    The denoise step is timed once (the five steps are identical work); the VAE decode runs at its real size (the 64x64 latent of the
    tiled 512x512 sample), once.  One tiled sample = 4 joint-target images."""
    from genima_amd import configs, schema, weights
    from oracle import sd_torch as O

    cores = host_threads()
    torch.set_num_threads(cores)
    fam = configs.family("sd-turbo")
    gdev = "cuda" if torch.cuda.is_available() else "cpu"  # draw on the GPU (same bits as the numpy path), copy to host

    def synth(sch, seed):
        return {k: v.cpu() for k, v in weights.synth_state_dict(sch, seed, device=gdev).items()}

    t0 = time.time()
    usd, csd = synth(schema.unet_schema(fam["unet"]), 21), synth(schema.controlnet_schema(fam["controlnet"]), 22)
    vsd, tsd = synth(schema.vae_schema(fam["vae"], encoder=False), 23), synth(schema.clip_text_schema(fam["text"]), 24)
    gen_s = time.time() - t0
    g = torch.Generator().manual_seed(0)
    x, ctx = torch.randn(1, 4, 64, 64, generator=g), torch.randn(1, 77, 1024, generator=g)
    cond, t = torch.rand(1, 3, 512, 512, generator=g), torch.tensor([999.0])
    z = torch.randn(1, 4, 64, 64, generator=g)
    V = fam["text"]["vocab_size"]
    ids = torch.zeros(1, 77, dtype=torch.int64)
    ids[0, :14] = torch.tensor([V - 2] + [320 + i for i in range(12)] + [V - 1])

    def timed(fn, budget):
        ts = []
        while len(ts) < 2 and (not ts or sum(ts) + ts[-1] < budget):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
        return min(ts)

    def step():
        down, mid = O.controlnet_forward(csd, fam["controlnet"], x, t, ctx, cond)
        O.unet_forward(usd, fam["unet"], x, t, ctx, down, mid)

    with torch.no_grad():
        t_clip = timed(lambda: O.clip_text_forward(tsd, fam["text"], ids), 0.1 * max_seconds)
        t_step = timed(step, 0.6 * max_seconds)
        t_vae = timed(lambda: O.vae_decode(vsd, fam["vae"], z), 0.15 * max_seconds)
    per_sample = t_clip + 5.0 * t_step + t_vae
    return {"value": 4.0 / per_sample, "unit": "joint-target images/sec", "cores": cores, "kind": "port",
            "seconds_per_tiled_sample": per_sample,
            "sample": f"fp32 torch-CPU oracle at full SD-Turbo width on {cores} threads, pieces of one tiled 512x512 sample's 5-step call: "
                      f"CLIP-H text {t_clip:.2f} s + 5 x (ControlNet + UNet @ 64x64 latent, 1088 GFLOP) {t_step:.2f} s + VAE decode of the "
                      f"512x512 tile (2515 GFLOP) {t_vae:.2f} s = {per_sample:.1f} s per tiled sample (= 4 joint-target images); "
                      f"each piece best of <= 2 runs; weights drawn in {gen_s:.0f} s, untimed"}


def single_view_latency(pipe, dev, denoise_steps, rank, calls=10, workload="single_b1"):
    """BASELINE.json configs[1] as the evaluation loop sees it (B = frame_stack = 1, one 256x256 view) -- or, ``workload="tiled_b1"``, the
    call the real evaluation loop makes (controller/eval_genima.py:202-211: ONE 4-view tiled 512x512 observation per control step):
    latency of one 5-step call, stream replay of the recorded program and the same program as ONE hipGraph (the HIP form of the
    reference's ``torch_compile`` reduce-overhead flag)."""
    B, H, W, desc = WORKLOADS[workload]
    ids, img, lat = synthetic_inputs(pipe, B, H, W, dev, rank)
    res = {}
    skip = os.environ.get("GN_BENCH_SKIP", "").split(",")  # bisect aid (profiles/r05_v9_train_order.txt): graph / two_calls / single / tiled_b1
    for graph in ((False,) if "graph" in skip else (False, True)):
        pipe.enable_hip_graph(graph)
        for _ in range(3):
            pipe(prompt_ids=ids, image=img, latents=lat, num_inference_steps=denoise_steps, guidance_scale=0.0, output_type="pt")
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            pipe(prompt_ids=ids, image=img, latents=lat, num_inference_steps=denoise_steps, guidance_scale=0.0, output_type="pt")
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        res["hip_graph" if graph else "stream_replay"] = 1000.0 * ts[len(ts) // 2]
    pipe.enable_hip_graph(False)
    best = min(res.values())
    return {"workload": desc, "ms_per_call_median": res, "images_per_sec": (4.0 if H == 512 else 1.0) * 1000.0 / best,
            "weight_streaming_ideal_ms": 13.1e9 / 8e12 * 1e3,  # 13.1 GB of weights per call at 8 TB/s
            "frac_of_hbm_roofline": (13.1e9 / 8e12 * 1e3) / best}


def two_calls_in_flight(pipe, dev, denoise_steps, rank, calls=8):
    """Throughput with TWO tiled B = 8 calls in flight (two recorded programs with their own buffers on two HIP streams, looping back to
    back): what a host that serves two episode batches gets.  An extra, never the headline value -- `value` is one call at a time, as the
    reference's control loop runs it; a call's latency doubles here."""
    B, H, W, _ = WORKLOADS["tiled_b8"]
    ids, img, lat = synthetic_inputs(pipe, B, H, W, dev, rank)
    saved = dict(pipe._progs)
    progs, main = [], None
    try:
        for _ in range(2):
            pipe._progs.clear()
            progs.append(pipe.program(B, H, W, denoise_steps))
        main = progs[0].engine.stream
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for io, st in zip(progs, streams):
            io.engine.use_stream(st)
            io.ids.copy_(ids.to(torch.int32)); io.image_u8.copy_(img); io.noise.copy_(lat.permute(0, 2, 3, 1))
        torch.cuda.synchronize(dev)
        for _ in range(2):
            for io in progs:
                io.engine.run()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(calls):
            for io in progs:
                io.engine.run()
        torch.cuda.synchronize(dev)
        ms = 1000.0 * (time.perf_counter() - t0) / (2 * calls)
    finally:
        for io in progs:
            if main is not None:
                io.engine.use_stream(main)
        pipe._progs.clear()
        pipe._progs.update(saved)
    return {"workload": "two tiled_b8 calls in flight on two streams (own buffers)", "ms_per_call": ms, "images_per_sec": 4.0 * B * 1000.0 / ms}


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
    ap.add_argument("--no-train", action="store_true", help="skip the `train` extra (BASELINE.json's second metric: ControlNet train steps/s)")
    ap.add_argument("--no-single-view", action="store_true", help="skip the `single_view_b1` extra (configs[1] latency, hipGraph)")
    ap.add_argument("--train-steps", type=int, default=10)
    ap.add_argument("--no-act", action="store_true", help="skip the ACT controller forward after each pipeline call")
    ap.add_argument("--dump-ops", default=None, help="write the per-(kernel, shape) HIP-event timing table of one call to this CSV")
    args = ap.parse_args()

    from genima_amd.dist import maybe_self_launch

    maybe_self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus)  # --gpus N > 1 without a launcher: become the launcher
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test aid (tests/test_bench_gpu.py): GN_BENCH_SHARE_DEVICE=1 puts every rank on device 0 and GN_BENCH_BACKEND=gloo carries the
    # collectives (RCCL refuses two ranks on one device) -- the N > 1 code path of this script on a one-GPU box; never the measured setup
    share = os.environ.get("GN_BENCH_SHARE_DEVICE") == "1"
    if share:
        local = 0
    if not torch.cuda.is_available() or local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} of {world} needs ROCm device {local}, {torch.cuda.device_count()} visible")
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime

        dist.init_process_group(os.environ.get("GN_BENCH_BACKEND", "nccl"), rank=rank, world_size=world, timeout=datetime.timedelta(minutes=5))
    assert world == max(1, args.gpus), f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from genima_amd import configs
    from genima_amd.pipeline import StableDiffusionControlNetPipeline, StableDiffusionXLControlNetPipeline

    B, H, W, desc = WORKLOADS[args.workload]
    fam_cfg = configs.family(args.family)
    # the SDXL families (BASELINE.json configs[4]'s agent: controller/agent/sdxl_controlnet_agent.py) have two text towers
    pipe_cls = StableDiffusionXLControlNetPipeline if "text_2" in fam_cfg else StableDiffusionControlNetPipeline
    pipe = pipe_cls.from_synthetic(fam_cfg, seed=0, gen_device=dev)
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
    # The timed region: K calls, each one the reference's own bracket -- prompt + images in HBM -> denoise ->
    # VAE -> uint8 images AND the controller's actions on the HOST (gen_time + control_time of controller/eval_genima.py:202-247) -- i.e. every call ends in
    # its D->H copy and is therefore individually synchronised.  `value` = units / the whole bracket (barrier + synchronize on both sides, max over ranks);
    # the median call and the back-to-back rate without the D->H (rounds 1 - 5's `value`) ride along as extras.
    barrier()
    per_call = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tc = time.perf_counter()
        call("np")
        per_call.append(time.perf_counter() - tc)
    barrier()
    dt = time.perf_counter() - t0
    from genima_amd.dist import max_over_ranks

    dt = max_over_ranks(dt, dev)  # the slowest rank defines the job's time
    calls = args.steps
    units = 4.0 if H == 512 else 1.0
    value = units * B * world * calls / dt
    per_call.sort()
    median_call = per_call[len(per_call) // 2] if len(per_call) % 2 else 0.5 * (per_call[len(per_call) // 2 - 1] + per_call[len(per_call) // 2])

    # back-to-back bracket: the same K calls queued without waiting for their outputs (no D->H), rank-local, informational
    barrier()
    t1 = time.perf_counter()
    for _ in range(calls):
        call()
    torch.cuda.synchronize(dev)
    dt_b2b = (time.perf_counter() - t1) / calls

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

    label = {"sd-turbo": "SD-Turbo"}.get(args.family, args.family)  # other families (sdxl-turbo: configs[4]'s agent) name themselves
    out = {
        "metric": f"joint-target images/sec ({label} + ControlNet, 4-view tiled 512x512, 5 steps, incl. CLIP text + VAE decode)"
        if H == 512 else f"images/sec ({label} + ControlNet 256x256 single view, 5 steps, incl. CLIP text + VAE decode)",
        "value": value, "unit": "images/sec", "n_gpus": world, "steps": calls, "warmup": args.warmup,
        "ms_per_step": 1000.0 * dt / calls, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 (f32 accumulate)", "data": f"synthetic (seeded random-init {label}-architecture weights, counter-PRNG images)",
        "config": {"workload": desc, "family": args.family, "per_gpu_batch": B, "global_batch": B * world,
                   "image": f"{H}x{W}", "denoise_steps": args.denoise_steps, "parallelism": f"replicas x{world} (episodes sharded, no collective)",
                   "hip_graph": bool(args.graph), "act_controller_forward": act_agent is not None,
                   "two_streams": bool(pipe.two_streams)},
        "images_per_sec_per_gpu": value / world,
        "act_controller_ms_per_call": act_ms,
        "ms_per_call_median": 1000.0 * median_call,
        "value_from_median_call": units * B * world / median_call,
        "value_back_to_back_no_d2h": units * B * world / dt_b2b,
        "timed_region": "K individually synchronised calls, each incl. the uint8 image + action D->H copies (SURVEY 8(d))",
        "algorithmic_tflops_per_gpu": GFLOP_PER_CALL.get(H, 0.0) * B * calls / dt / 1000.0,
    }

    # BASELINE.json's second metric on the same launch: the ControlNet fine-tune step (configs[3], per-GPU batch 8), N ranks data
    # parallel with the bucketed RCCL reduce-scatter + all-gather of the flat gradient overlapped with the backward.
    # Measured IN THIS PROCESS.  At N = 1 it runs right behind the headline loop, before the other extras: behind them (two more recorded
    # programs with their hipGraphs, a second B = 8 buffer set, 2 000 event pairs of the per-op replay) the same step read 3 - 4 ms slower
    # (66.1 against 61.7 ms, profiles/r04_v7_train_inline_bisect.txt) and round 4 moved it into a process of its own; in front of them it reads what
    # `python bench_train.py` reads (profiles/r05_v9_train_order.txt).  GN_BENCH_TRAIN=subprocess / last restore the other two methods.
    def run_train_extra(free_pipeline: bool):
        nonlocal pipe, act_agent
        if free_pipeline:
            del pipe, act_agent
            pipe = act_agent = None
            import gc

            gc.collect()  # the recorded programs sit in reference cycles: free them (and their device buffers) now, not inside a timed train step
            torch.cuda.empty_cache()
        try:
            line, method = None, "in this process"
            if world == 1 and os.environ.get("GN_BENCH_TRAIN") == "subprocess":
                import subprocess

                r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_train.py"), "--steps", str(args.train_steps), "--warmup", "3"],
                                   capture_output=True, text=True, timeout=900)
                tail = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
                if r.returncode == 0 and tail:
                    line = json.loads(tail[-1])
                    method = "bench_train.py in a process of its own"
                else:
                    method = f"in this process (subprocess rc {r.returncode}: {(r.stderr or r.stdout).strip()[-200:]})"
            if line is None:
                import bench_train

                targs = bench_train.parse_args(["--gpus", str(world), "--steps", str(args.train_steps), "--warmup", "3"])  # (as `bench_train.py --steps 10 --warmup 3`: the trainer builds its lazily
                # derived state -- weight copies, their one-launch table, gc.freeze -- in its first steps, and consecutive steps overlap)
                line = bench_train.run(targs)
            if rank == 0 and line is not None:
                line["process"] = method + ("" if free_pipeline else ", right behind the headline loop (inference pipeline resident)")
                out["train"] = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "samples_per_sec",
                                                     "dtype", "config", "roofline", "peak_mem_gb", "loss_first", "loss_last", "scaling", "process") if k in line}
        except Exception as e:
            if rank == 0:
                out["train"] = {"error": repr(e)[:300]}
        import gc

        gc.unfreeze()  # (ControlNetTrainer(gc_freeze=True) parks every live object in the permanent generation)
        gc.collect()
        torch.cuda.empty_cache()

    train_first = world == 1 and os.environ.get("GN_BENCH_TRAIN", "first") == "first"
    if not args.no_train and train_first:
        run_train_extra(free_pipeline=False)

    if rank == 0 and world == 1 and not args.no_roofline:
        io = pipe.program(B, H, W, args.denoise_steps)
        if args.graph:
            io.engine.use_stream(io.stream)
        agg = per_op_profile(pipe, io, args.dump_ops)
        gemm_kinds = [k for k in agg if k.startswith("conv") or k in ("linear", "tblock")]  # tblock: fused chains of Linears (csrc/tblock.hip)
        g_ms = sum(agg[k]["ms"] for k in gemm_kinds)
        g_fl = sum(agg[k]["flops"] for k in gemm_kinds)
        n_l = sum(agg[k]["launches"] for k in gemm_kinds)
        ach = g_fl / (g_ms * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_dma_kernel / gemm_kernel<BM,BN,WM,WN,CONV> (MFMA implicit-GEMM conv3x3/1x1 + Linear) + tblock_kernel (fused Linear chains)",
                           "achieved": ach, "peak": MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TF,
                           "traffic": pmc_traffic_per_launch("gemm", n_l) if args.workload == "tiled_b8" else None,
                           "traffic_note": "HBM bytes per launch of this kernel family from the committed rocprofv3 --pmc passes of this "
                                           f"workload (profiles/{PMC_TRAFFIC_FILE}: 2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes; "
                                           "the x2 is MI355X_MICROARCH.md's gfx950 FETCH_SIZE correction); not re-collected by this run",
                           "traffic_stamp": pmc_traffic_stamp(),
                           "frac_of_sustained_mfma_rate": ach / MFMA_SUSTAINED_TF,
                           "achieved_reference_algorithm": sum(agg[k]["ref_flops"] for k in gemm_kinds) / (g_ms * 1e-3) / 1e12,
                           "reference_algorithm_note": "`achieved` counts the multiply-adds the launches EXECUTE; the upsampling 3x3 convs run as four 2x2 "
                                                       "phase convs (4/9 of the reference algorithm's MACs, SURVEY.md Appendix C) -- counted at the "
                                                       "reference's 2*MAC figure the family does `achieved_reference_algorithm` TFLOP/s",
                           "sustained_note": f"a pure v_mfma_f32_32x32x16_f16 stream sustains {MFMA_SUSTAINED_TF:.0f} TFLOP/s on MI355X "
                                             "(19-21 ns per MFMA per SIMD at the ~1.6-1.7 GHz the chip holds under matrix load; "
                                             "tools/probes/mfma_coissue.hip); `frac` stays quoted against the nominal dense peak",
                           "launches_per_call": n_l, "avg_launch_ms": g_ms / max(1, n_l), "algorithmic_gflop_per_call": g_fl / 1e9,
                           "algorithmic_bytes_per_launch": sum(agg[k]["bytes"] for k in gemm_kinds) / max(1, n_l),
                           "share_of_call_time": g_ms / sum(a["ms"] for a in agg.values())}
        extra = []
        for k in sorted(agg, key=lambda k: -agg[k]["ms"]):
            a = agg[k]
            if k == "stream":  # fork / main / join markers of the two-stream program: no launch, no bytes -- not a roofline row
                continue
            row = {"kernel": k, "ms_per_call": a["ms"], "launches": a["launches"]}
            if a["flops"] > 0:
                tf = a["flops"] / (a["ms"] * 1e-3) / 1e12
                row.update(bound="mfma", achieved=tf, peak=MFMA_PEAK_TF, unit="TFLOP/s", frac=tf / MFMA_PEAK_TF)
            else:
                gbs = a["bytes"] / (a["ms"] * 1e-3) / 1e9
                row.update(bound="hbm", achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS)
            extra.append(row)
        out["roofline_extra"] = extra
    if rank == 0 and world == 1 and not args.no_single_view:
        try:
            out["single_view_b1"] = single_view_latency(pipe, dev, args.denoise_steps, rank)
        except Exception as e:  # an extra must never cost the headline line
            out["single_view_b1"] = {"error": repr(e)[:300]}
        try:
            out["tiled_b1"] = single_view_latency(pipe, dev, args.denoise_steps, rank, workload="tiled_b1")
        except Exception as e:
            out["tiled_b1"] = {"error": repr(e)[:300]}
        if args.workload == "tiled_b8" and args.family == "sd-turbo" and "two_calls" not in os.environ.get("GN_BENCH_SKIP", "").split(","):
            try:
                out["two_calls_in_flight"] = two_calls_in_flight(pipe, dev, args.denoise_steps, rank)
            except Exception as e:
                out["two_calls_in_flight"] = {"error": repr(e)[:300]}

    if not args.no_train and not train_first:
        run_train_extra(free_pipeline=os.environ.get("GN_BENCH_TRAIN") != "last_keep")  # (last_keep: the bisect of what the extras leave behind)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline()
        except Exception as e:
            out["cpu_baseline"] = {"error": repr(e)[:300]}

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
