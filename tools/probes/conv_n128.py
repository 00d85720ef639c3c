"""Every tile on the VAE's 128-channel 512x512 convs (the slowest large launches of the call: 630 TF/s)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from genima_amd.engine import Engine
E = Engine("cuda:0"); E.no_table = True; E.autotune = False
def h(*s, sc=0.5): return (torch.randn(*s, device="cuda") * sc).half()
names = ["R256x128","R128x128","R128x64","R64x64","R256x64","R128x256","D256x256","D256x128","D128x128","D128x64","D64x64","D256x64","D128x320","D256x320","PP256x256","S3_128x128","S3_128x64","S3_64x64","S3_256x64","S3_128x160","S3_64x160","S3_64x320","D128x160"]
for (B, hw, cin, cout) in ((8, 512, 128, 128), (8, 512, 256, 128), (8, 256, 256, 256)):
    x, w, b = h(B, hw, hw, cin), h(cout, 9 * cin, sc=0.02), h(cout)
    r = h(B, hw, hw, cout)
    fl = 2.0 * B * hw * hw * cout * 9 * cin
    for cfg in (0, 1, 4, 6, 7, 8, 9, 11, 14, 15, 18):
        E.lib.gn_set_gemm_tile_override(cfg)
        for _ in range(2): E.conv2d(x, w, b, residual=r)
        a, bb = E.event(), E.event(); E.event_record(a)
        for _ in range(5): E.conv2d(x, w, b, residual=r)
        E.event_record(bb); ms = E.event_elapsed_ms(a, bb) / 5
        print(f"conv {cin}->{cout}@{hw} cfg {cfg:2d} {names[cfg]:10s} {ms*1e3:8.1f} us {fl/ms/1e9:7.1f} TF/s", flush=True)
