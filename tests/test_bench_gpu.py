"""-m gpu: the driver's own command lines, run as subprocesses."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable] + argv, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_single_gpu():
    out = _run(["bench.py", "--gpus", "1", "--steps", "1", "--warmup", "1", "--workload", "tiled_b1", "--no-train", "--no-cpu-baseline",
                "--no-single-view"])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["unit"] == "images/sec"
    assert out["roofline"]["bound"] == "mfma" and 0 < out["roofline"]["frac"] < 1


def test_bench_launcher_runs_ranks_under_torchrun():
    """One rank through the launcher path proper: `torch.distributed.run --nproc-per-node 1` around bench.py, RCCL initialised at world
    size 1 is what the box can offer (two ranks need two devices)."""
    from genima_amd import dist as gd

    cmd = gd.self_launch_command(os.path.join(ROOT, "bench.py"), ["--gpus", "1", "--steps", "1", "--warmup", "0", "--workload", "single_b1",
                                                                  "--no-train", "--no-cpu-baseline", "--no-single-view", "--no-roofline"], 1)
    out = _run(cmd[1:])
    assert out["n_gpus"] == 1 and out["value"] > 0


def test_bench_two_ranks_share_the_device_over_gloo():
    """The N > 1 code path of bench.py end to end on a one-GPU box: two ranks under torch.distributed.run, both on device 0, gloo carrying
    the barrier / max-over-ranks / gradient exchange (RCCL refuses two ranks on one device; the 8-GPU curve itself is the driver's to
    measure): one JSON line from rank 0, n_gpus = 2, the whole-job value = 2 x the per-rank batch over the slowest rank's time, and the
    `train` extra with world size 2 (dp2, summed gradients over both ranks)."""
    from genima_amd import dist as gd

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(GN_BENCH_SHARE_DEVICE="1", GN_BENCH_BACKEND="gloo")
    cmd = gd.self_launch_command(os.path.join(ROOT, "bench.py"), ["--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "single_b1",
                                                                  "--no-cpu-baseline", "--no-single-view", "--train-steps", "1"], 2)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2 and out["value"] > 0
    assert "roofline" not in out  # rank-0-at-N=1-only extras stay out of multi-rank lines
    assert out["train"]["n_gpus"] == 2 and out["train"]["config"]["parallelism"] == "dp2" and out["train"]["ms_per_step"] > 0
