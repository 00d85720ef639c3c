"""What the in-call cost of a B = 1 launch is made of: each gn_gemm op of the recorded single-view program is timed
  (a) where it runs in the call (op-by-op replay: cold code, cold weights, its input fresh from the producer),
  (b) immediately repeated (code, weights, TLB all warm),
  (c) behind a SIBLING op of the same shape (another layer: the same kernel code just ran, the weights are cold).
(a) - (c) ~ instruction fetch / kernel-specific warm-up, (c) - (b) ~ the cold weight stream."""
import ctypes as C, os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench
from incall_tune import gemm_ops
from genima_amd import configs
from genima_amd.pipeline import StableDiffusionControlNetPipeline
wl = sys.argv[1] if len(sys.argv) > 1 else "single_b1"
dev = torch.device("cuda", 0)
pipe = StableDiffusionControlNetPipeline.from_synthetic(configs.family("sd-turbo"), seed=0, gen_device=dev); pipe.to(dev)
B, H, W, _ = bench.WORKLOADS[wl]
ids, img, lat = bench.synthetic_inputs(pipe, B, H, W, dev, 0)
for _ in range(2): pipe(prompt_ids=ids, image=img, latents=lat, num_inference_steps=5, guidance_scale=0.0, output_type="pt")
E = pipe.program(B, H, W, 5).engine
ops = gemm_ops(E)
n = E.num_ops
def timed(fn):
    a, b = E.event(), E.event(); E.event_record(a); fn(); E.event_record(b); return (a, b)
res = collections.defaultdict(lambda: [[], [], []])
for rep in range(2):
    evs = []
    pos = 0
    where = sorted((i, k) for k, lst in ops.items() if len(lst) >= 10 for i, _, _ in lst)
    sib = {}
    for k, lst in ops.items():
        idx = [i for i, _, _ in lst]
        for j, i in enumerate(idx): sib[i] = idx[j - 1] if j else idx[-1]
    for i, k in where:
        if i > pos: E.run(pos, i)
        ea = timed(lambda: E.run(i, i + 1))
        eb = timed(lambda: E.run(i, i + 1))
        E.run(sib[i], sib[i] + 1)        # (rewrites the sibling's output with the values it already holds or will be recomputed later)
        ec = timed(lambda: E.run(i, i + 1))
        evs.append((k, ea, eb, ec)); pos = i + 1
    E.run(pos, n); E.synchronize()
    for k, ea, eb, ec in evs:
        for s, (a, b) in enumerate((ea, eb, ec)):
            res[k][s].append(E.event_elapsed_ms(a, b) * 1e3)
            E.lib.gn_event_destroy(a); E.lib.gn_event_destroy(b)
print(f"{'shape key':52s} {'ops':>4s} {'in-call us':>10s} {'repeat us':>10s} {'sibling us':>10s}")
tot = [0.0, 0.0, 0.0]
for k, (a, b, c) in sorted(res.items(), key=lambda kv: -sum(kv[1][0])):
    m = [sum(x) / len(x) for x in (a, b, c)]
    nops = len(a) // 2
    for s in range(3): tot[s] += m[s] * nops
    print(f"{k:52s} {nops:4d} {m[0]:10.1f} {m[1]:10.1f} {m[2]:10.1f}")
print(f"total over these ops, ms per call: in-call {tot[0]/1e3:.2f}  repeat {tot[1]/1e3:.2f}  behind a sibling {tot[2]/1e3:.2f}")
