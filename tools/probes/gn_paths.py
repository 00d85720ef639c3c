import os, sys, torch
sys.path.insert(0, os.getcwd())
from genima_amd.engine import Engine
E = Engine("cuda:0")
for b, hw, c in [(1, 4096, 320), (1, 1024, 640), (1, 256, 1280), (1, 64, 1280), (1, 4096, 640), (1, 1024, 1280), (8, 4096, 320), (8, 1024, 640), (8, 4096, 640)]:
    side = int(hw ** 0.5)
    x = torch.randn(b, side, side, c, device="cuda").half()
    g, bt = torch.ones(c, device="cuda").half(), torch.zeros(c, device="cuda").half()
    def t(fn):
        for _ in range(3): fn()
        e0, e1 = E.event(), E.event(); E.event_record(e0)
        for _ in range(20): fn()
        E.event_record(e1); return E.event_elapsed_ms(e0, e1) / 20 * 1e3
    a = t(lambda: E.groupnorm(x, g, bt, 32, 1e-5, act=1))
    s = t(lambda: E.groupnorm_stats(x, g, bt, 32, 1e-5))
    print(f"groupnorm {b}x{hw}x{c}: full {a:7.1f} us   statistics-only (2 launches) {s:7.1f} us", flush=True)
