"""Same conv, weights hot (one tensor re-used: L2 / MALL resident) against cold (a ring of tensors larger than the 256 MB MALL, as in
the real call where every layer brings its own weights from HBM)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
E = Engine("cuda:0"); E.no_table = True; E.autotune = False
plan = [0, 0]
orig = E._gemm
def patched(d, keep):
    d.tile, d.splitk = plan
    return orig(d, keep)
E._gemm = patched
def h(*s, sc=0.5): return (torch.randn(*s, device="cuda") * sc).half()
for (B, hw, cin, cout, plans) in ((8, 16, 1280, 1280, [(9, 3), (15, 5), (23, 4)]), (8, 8, 1280, 1280, [(9, 0), (23, 0)]), (8, 32, 640, 640, [(20, 0), (23, 2)]),
                                  (8, 64, 320, 320, [(23, 0), (20, 0)]), (8, 16, 2560, 1280, [(15, 6)])):
    nring = max(2, int(600e6 / (cout * 9 * cin * 2)))
    x, b = h(B, hw, hw, cin), h(cout)
    ws = [h(cout, 9 * cin, sc=0.02) for _ in range(nring)]
    fl = 2.0 * B * hw * hw * cout * 9 * cin
    for (t, sk) in plans:
        plan[:] = [t, sk]
        out = []
        for ring in (1, nring):
            for i in range(2 * ring if ring > 1 else 3): E.conv2d(x, ws[i % ring], b)
            best = 1e9
            for _ in range(3):
                a, bb = E.event(), E.event(); E.event_record(a)
                for i in range(nring): E.conv2d(x, ws[i % ring], b)
                E.event_record(bb); best = min(best, E.event_elapsed_ms(a, bb) / nring)
            out.append(best)
        print(f"conv {cin}->{cout}@{hw} t{t}/sk{sk}: hot {out[0]*1e3:.1f} us  cold {out[1]*1e3:.1f} us ({nring} weight tensors, {cout*9*cin*2/1e6:.1f} MB each)", flush=True)
