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
