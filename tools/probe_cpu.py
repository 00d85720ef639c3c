import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e: print("no cpu.max", e)
import torch.nn.functional as F
x = torch.randn(1, 320, 32, 32); w = torch.randn(320, 320, 3, 3)
a = torch.randn(2048, 2048); b = torch.randn(2048, 2048)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    F.conv2d(x, w, padding=1); a @ b
    t = time.time(); [F.conv2d(x, w, padding=1) for _ in range(5)]; tc = (time.time() - t) / 5
    t = time.time(); [a @ b for _ in range(5)]; tm = (time.time() - t) / 5
    print(nt, "conv GFLOP/s", 2 * 320 * 320 * 9 * 1024 / tc / 1e9, "matmul GFLOP/s", 2 * 2048**3 / tm / 1e9)
