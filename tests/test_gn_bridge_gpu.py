"""-m gpu: the GroupNorm bridge (include/genima_hip.h gn_stats_sink / gn_norm_in; csrc/gn_bridge.h) -- diffusers' ResnetBlock2D runs
conv -> GroupNorm -> SiLU -> conv and Transformer2DModel conv -> GroupNorm -> proj_in as separate passes (inside `self.pipe(...)`,
controller/agent/sd_controlnet_agent.py:67-76); here the producer adds the statistics of what it stores and the consumer normalises its own
A tiles (or one apply launch does).

  * producer side: the fixed-point (sum, sum of squares) a gn_gemm / its split-K reduce / gn_add_multi leave, against f64 sums of the f16
    values they stored, on every tile family, with concat offsets and group sizes that straddle tiles;
  * consumer side: conv / Linear with gn_gemm_desc.norm_in against torch fp32 `conv2d(silu(group_norm(x)))` on the same f16-rounded inputs at
    the 1e-3 bar and against the GroupNorm + conv launches it replaces -- every ring tile, K splits, the zero padding, a concatenated input,
    an appended k_append shortcut, row tiles that span several samples;
  * the apply-from-statistics GroupNorm launch;
  * a recorded program: producer -> bridge -> consumer plumbing (gn_program_set_sink, the memset op), replayed twice bit-identically."""
import pytest
import torch
import torch.nn.functional as F

from genima_amd._lib import ACT_NONE, ACT_SILU
from genima_amd.engine import Engine, Norm
from genima_amd.packing import pack_conv_weight
from util import assert_close, q16, randn_h, rel_l2

pytestmark = pytest.mark.gpu

FIX, FIXSQ = float(1 << 24), float(1 << 12)  # GN_STATS_SHIFT / GN_STATS_SHIFT_SQ


def _nchw(t):
    return t.float().permute(0, 3, 1, 2)


def _host_stats(tensors, groups):
    """f64 (sum, sum of squares) per (sample, group) of the channel concatenation of NHWC tensors -> int64 fixed point [B, groups, 2] (cuda)."""
    x = torch.cat([t.double().cpu().reshape(t.shape[0], -1, t.shape[-1]) for t in tensors if t.shape[-1] > 0], dim=-1)
    B, _, C = x.shape
    xg = x.reshape(B, -1, groups, C // groups)
    s, q = xg.sum(dim=(1, 3)), (xg * xg).sum(dim=(1, 3))
    out = torch.zeros(1, B, groups, 16, dtype=torch.int64)  # one replica; a 128-byte line per (sample, group): GN_STATS_LINE
    out[0, :, :, 0], out[0, :, :, 1] = s.mul(FIX).round().to(torch.int64), q.mul(FIXSQ).round().to(torch.int64)
    return out.cuda()


def _zero_stats(B, groups, replicas=1):
    return torch.zeros(replicas, B, groups, 16, dtype=torch.int64, device="cuda")


def _stats_close(st, ref, what, n):
    """n = elements of a (sample, group) slab.  The producers sum f32 partials per tile (then exact integer adds): 1e-5 of the slab's sum of
    squares; the plain sums can cancel to ~0, so they are judged against sqrt(n * sumsq) >= |sum|."""
    assert int(st[..., 2:].abs().max()) == 0, "only the first two words of a line are written"
    scale = torch.tensor([FIX, FIXSQ], dtype=torch.float64)
    a, b = st[..., :2].sum(0).double().cpu() / scale, ref[..., :2].sum(0).double().cpu() / scale  # (replicas summed: integer adds)
    tol_s = 1e-5 * (n * b[..., 1].abs()).sqrt() + 1e-3
    assert ((a[..., 0] - b[..., 0]).abs() <= tol_s).all(), (what, float((a[..., 0] - b[..., 0]).abs().max()))
    assert ((a[..., 1] - b[..., 1]).abs() <= 1e-5 * b[..., 1].abs() + 5e-2).all(), (what, float(((a[..., 1] - b[..., 1]).abs() / b[..., 1].abs().clamp_min(1e-9)).max()))


@pytest.mark.parametrize("tile,splitk", [(0, 0), (2, 1), (8, 1), (9, 1), (10, 1), (13, 1), (15, 1), (16, 1), (17, 1), (18, 1), (21, 1), (22, 1), (23, 1),
                                         (18, 3), (17, 2), (10, 4)])
def test_conv_producer_statistics(tile, splitk):
    """conv 3x3 320 -> 320 with bias + time shift + residual: the statistics block it leaves for a consumer whose concatenated input puts this
    tensor at channel offset 640 in groups of 30 (so groups straddle the 32 / 64 / 160-wide column tiles)."""
    E = Engine("cuda:0")
    B, H, C, N = 2, 16, 320, 320
    x, w, b = randn_h(B, H, H, C, seed=1), randn_h(N, C, 3, 3, seed=2, scale=0.03), randn_h(N, seed=3)
    res, sh = randn_h(B, H, H, N, seed=4), randn_h(B, N, seed=5)
    groups, cpg, coff = 32, 30, 640
    E.autotune = False
    st = _zero_stats(B, groups, replicas=1 + tile % 3)
    if tile:
        E.lib.gn_set_gemm_tile_override(tile - 1)
    try:
        y = E.conv2d(x, pack_conv_weight(w.float().cpu()).cuda(), b, shift=sh, residual=res, sink=(st, cpg, coff, H * H), splitk=splitk)
    finally:
        E.lib.gn_set_gemm_tile_override(-1)
    E.synchronize()
    ref = F.conv2d(_nchw(x), w.float(), b.float(), padding=1) + sh.float()[:, :, None, None] + _nchw(res)
    assert_close(_nchw(y), ref, 1e-3, f"conv tile {tile}")
    pad_lo = torch.zeros(B, H, H, coff, dtype=torch.float16, device="cuda")
    pad_hi = torch.zeros(B, H, H, groups * cpg - coff - N, dtype=torch.float16, device="cuda")
    _stats_close(st, _host_stats([pad_lo, y, pad_hi], groups), f"tile {tile} splitk {splitk}", H * H * cpg)


def test_linear_and_phase_conv_producer_statistics():
    E = Engine("cuda:0")
    # dense: proj_out + residual at C = 640 (8 samples of 64 rows: a 128 / 256-row tile spans several samples)
    B, R, C = 8, 64, 640
    a, w, b, res = randn_h(B, R, C, seed=1), randn_h(C, C, seed=2, scale=0.04), randn_h(C, seed=3), randn_h(B, R, C, seed=4)
    st = _zero_stats(B, 32)
    y = E.linear(a, w, b, residual=res, sink=(st, 20, 0, R))
    E.synchronize()
    assert_close(y, a.float() @ w.float().t() + b.float() + res.float(), 1e-3, "linear")
    _stats_close(st, _host_stats([y], 32), "linear sink", R * 20)
    # the four phase convs of an Upsample2D as one launch: the statistics of the whole upsampled tensor
    from genima_amd.packing import pack_upsample_phases

    B, H, C = 2, 16, 128
    x, w, b = randn_h(B, H, H, C, seed=5), randn_h(C, C, 3, 3, seed=6, scale=0.04), randn_h(C, seed=7)
    ER = Engine("cuda:0", record=True, autotune=False, gn_bridge=True)
    ER.gn_reduce_fuse = False
    W = {"u.weight": pack_conv_weight(w.float().cpu()).cuda(), "u.bias": b}
    W["u.up4.weight"] = pack_upsample_phases(w.float().cpu()).cuda()
    up = ER.conv2d_up2x(x, W["u.up4.weight"], b, name="up")
    n = ER.groupnorm(up, torch.ones(C, dtype=torch.float16, device="cuda"), torch.zeros(C, dtype=torch.float16, device="cuda"), 32, 1e-5, name="gn")
    assert ER.meta[-1]["shape"][-1] == "st", "the GroupNorm behind the phase convs must take the bridge"
    ER.run()
    ER.synchronize()
    ref = F.group_norm(q16(_nchw(up)), 32, eps=1e-5)
    assert_close(_nchw(n), ref, 1e-3, "GroupNorm from the phase convs' statistics")


def _gn_case(B, H, Wd, C1, C2, N, seed, ksize=3):
    g = torch.Generator().manual_seed(seed)
    x = q16(torch.randn(B, H, Wd, C1, generator=g) * 1.4 + 0.4 * torch.randn(B, 1, 1, C1, generator=g))
    x2 = q16(torch.randn(B, H, Wd, C2, generator=g) * 0.8 - 0.3) if C2 else None
    C = C1 + C2
    w = q16(torch.randn(N, C, ksize, ksize, generator=g) * (ksize * ksize * C) ** -0.5)
    b = q16(torch.randn(N, generator=g) * 0.2)
    gamma, beta = q16(1.0 + 0.2 * torch.randn(C, generator=g)), q16(0.2 * torch.randn(C, generator=g))
    return x, x2, w, b, gamma, beta


@pytest.mark.parametrize("B,H,C1,C2,N,tile,splitk,silu", [
    (2, 16, 320, 0, 320, 0, 0, True), (2, 16, 320, 0, 320, 16, 1, True), (2, 16, 320, 0, 320, 17, 1, False), (2, 16, 320, 0, 320, 18, 3, True),
    (2, 16, 320, 0, 320, 19, 1, True), (2, 16, 320, 0, 320, 20, 1, True), (2, 16, 320, 0, 320, 21, 2, True), (2, 16, 320, 0, 320, 22, 1, True),
    (1, 16, 640, 320, 640, 0, 0, True),    # the up blocks' concatenated input: groups of 30 straddle the two sources
    (4, 8, 1280, 1280, 1280, 17, 4, True),  # 8x8 level: a 128-row tile spans two samples
    (8, 8, 1280, 0, 1280, 19, 2, True),     # ... a 256-row tile four
    (1, 32, 320, 0, 320, 0, 0, True)])
def test_conv_with_groupnorm_in_its_a_path(B, H, C1, C2, N, tile, splitk, silu):
    E = Engine("cuda:0")
    E.autotune = False
    x, x2, w, b, gamma, beta = _gn_case(B, H, H, C1, C2, N, seed=H + C1 + tile)
    G, eps = 32, 1e-5
    xc = x if x2 is None else torch.cat([x, x2], dim=-1)
    n_ref = F.group_norm(_nchw(xc), G, gamma, beta, eps)
    ref = F.conv2d(q16(F.silu(n_ref) if silu else n_ref), w, b, padding=1)
    xd, x2d = x.half().cuda(), None if x2 is None else x2.half().cuda()
    wd, bd, gd, bed = pack_conv_weight(w).cuda(), b.half().cuda(), gamma.half().cuda(), beta.half().cuda()
    st = _host_stats([xd] + ([x2d] if x2d is not None else []), G)
    act = ACT_SILU if silu else ACT_NONE
    if tile:
        E.lib.gn_set_gemm_tile_override(tile - 1)
    try:
        y = E.conv2d(xd, wd, bd, x2=x2d, norm=Norm(gd, bed, G, eps, act), norm_stats=st, splitk=splitk)
    finally:
        E.lib.gn_set_gemm_tile_override(-1)
    n = E.groupnorm(xd, gd, bed, G, eps, act=act, x2=x2d)
    y0 = E.conv2d(n, wd, bd)
    E.synchronize()
    assert_close(_nchw(y), ref, 1e-3, f"norm_in conv {C1}+{C2}->{N} {H}x{H} tile {tile}")
    assert rel_l2(y, y0.float()) < 3e-4, rel_l2(y, y0.float())


def test_norm_in_with_appended_shortcut_time_shift_and_1x1():
    """ResnetBlock2D's second half as ONE launch: conv2(silu(norm2(h)) ) + conv_shortcut(x) -- the appended 1x1 segment reads x RAW -- and a
    1x1 conv / a Linear (Transformer2DModel.norm -> proj_in) with the GroupNorm in the A path."""
    E = Engine("cuda:0")
    E.autotune = False
    B, H, C, Cx, N, G, eps = 2, 16, 640, 320, 640, 32, 1e-5
    h, _, w3, b3, gamma, beta = _gn_case(B, H, H, C, 0, N, seed=3)
    g = torch.Generator().manual_seed(4)
    x = q16(torch.randn(B, H, H, Cx, generator=g))
    w1, b1 = q16(torch.randn(N, Cx, 1, 1, generator=g) * Cx ** -0.5), q16(torch.randn(N, generator=g) * 0.1)
    sh = q16(torch.randn(B, N, generator=g) * 0.3)
    ref = (F.conv2d(q16(F.silu(F.group_norm(_nchw(h), G, gamma, beta, eps))), w3, b3, padding=1) + F.conv2d(_nchw(x), w1, b1)
           + sh[:, :, None, None])
    wcat = torch.cat([pack_conv_weight(w3).cuda(), pack_conv_weight(w1).cuda()], dim=1).contiguous()
    hd, xd = h.half().cuda(), x.half().cuda()
    st = _host_stats([hd], G)
    y = E.conv2d(hd, wcat, (b3 + b1).half().cuda(), append=xd, shift=sh.half().cuda(), norm=Norm(gamma.half().cuda(), beta.half().cuda(), G, eps, ACT_SILU),
                 norm_stats=st)
    E.synchronize()
    assert_close(_nchw(y), ref, 1e-3, "norm_in + k_append + shift")
    # 1x1 conv and the token-major Linear: GroupNorm without activation
    xs, _, w, b, gamma, beta = _gn_case(2, 16, 16, 640, 0, 640, seed=9, ksize=1)
    n_ref = q16(F.group_norm(_nchw(xs), G, gamma, beta, 1e-6))
    ref = F.conv2d(n_ref, w, b)
    xd = xs.half().cuda()
    st = _host_stats([xd], G)
    nm = Norm(gamma.half().cuda(), beta.half().cuda(), G, 1e-6, ACT_NONE)
    y1 = E.conv2d(xd, pack_conv_weight(w).cuda(), b.half().cuda(), ksize=1, norm=nm, norm_stats=st)
    y2 = E.linear(xd.view(2, 256, 640), w.reshape(640, 640).half().cuda(), b.half().cuda(), norm=nm, norm_stats=st)
    E.synchronize()
    assert_close(_nchw(y1), ref, 1e-3, "norm_in 1x1 conv")
    assert_close(_nchw(y2.view(2, 16, 16, 640)), ref, 1e-3, "norm_in Linear (proj_in)")


@pytest.mark.parametrize("B,HW,C1,C2,silu", [(1, 4096, 320, 0, True), (8, 64, 1280, 1280, True), (2, 1024, 640, 320, False), (3, 256, 1280, 0, True)])
def test_groupnorm_apply_from_statistics(B, HW, C1, C2, silu):
    E = Engine("cuda:0")
    g = torch.Generator().manual_seed(HW + C1)
    x = q16(torch.randn(B, HW, C1, generator=g) * 2.0 + 0.7).half().cuda()
    x2 = q16(torch.randn(B, HW, C2, generator=g) - 0.2).half().cuda() if C2 else None
    C = C1 + C2
    gamma, beta = q16(1.0 + 0.3 * torch.randn(C, generator=g)).half().cuda(), q16(0.3 * torch.randn(C, generator=g)).half().cuda()
    st = _host_stats([x] + ([x2] if x2 is not None else []), 32)
    y = E.groupnorm(x, gamma, beta, 32, 1e-5, act=ACT_SILU if silu else ACT_NONE, x2=x2, stats_in=st)
    y0 = E.groupnorm(x, gamma, beta, 32, 1e-5, act=ACT_SILU if silu else ACT_NONE, x2=x2)
    E.synchronize()
    xc = (x if x2 is None else torch.cat([x, x2], dim=-1)).float().cpu()
    ref = F.group_norm(xc.permute(0, 2, 1), 32, gamma.float().cpu(), beta.float().cpu(), 1e-5)
    ref = (F.silu(ref) if silu else ref).permute(0, 2, 1)
    assert_close(y, ref, 1e-3, "apply from statistics")
    assert rel_l2(y, y0.float()) < 1e-4


def test_add_multi_statistics():
    E = Engine("cuda:0")
    shapes = [(2, 32, 32, 320), (2, 16, 16, 640), (2, 8, 8, 1280), (2, 8, 8, 1280)]
    pairs = [(randn_h(*s, seed=i), randn_h(*s, seed=10 + i)) for i, s in enumerate(shapes)]
    # consumer partitions: behind 640 channels in groups of 30; behind 1280 in groups of 60; offset 0 groups of 80; no sink
    specs = [(30, 640), (60, 1280), (80, 0), None]
    sts = [None if sp is None else _zero_stats(2, 32, replicas=4) for sp in specs]
    sinks = [None if sp is None else (st, sp[0], sp[1], s[1] * s[2]) for sp, st, s in zip(specs, sts, shapes)]
    outs = E.add_multi(pairs, sinks=sinks)
    E.synchronize()
    for (a, b), o, sp, st, s in zip(pairs, outs, specs, sts, shapes):
        assert torch.equal(o, (a.float() + b.float()).half())
        if sp is not None:
            cpg, coff = sp
            lo = torch.zeros(s[:3] + (coff,), dtype=torch.float16, device="cuda")
            hi = torch.zeros(s[:3] + (32 * cpg - coff - s[3],), dtype=torch.float16, device="cuda")
            _stats_close(st, _host_stats([lo, o, hi], 32), f"add_multi sink {sp}", s[1] * s[2] * cpg)


def test_recorded_resnet_through_the_bridge():
    """A recorded ResnetBlock2D (time shift, conv_shortcut through k_append) behind a producing conv: every GroupNorm of the block takes the
    bridge (no groupnorm op without statistics in the program), the result matches torch and the unbridged program, and two replays agree bit
    for bit (the memset op clears the statistics arena at the top of each)."""
    from genima_amd import graphs, packing

    g = torch.Generator().manual_seed(21)
    Cin, Cout, G, T = 320, 640, 32, 1280
    sd = {"r.norm1.weight": 1.0 + 0.1 * torch.randn(Cin, generator=g), "r.norm1.bias": 0.1 * torch.randn(Cin, generator=g),
          "r.conv1.weight": torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5, "r.conv1.bias": 0.1 * torch.randn(Cout, generator=g),
          "r.time_emb_proj.weight": torch.randn(Cout, T, generator=g) * T ** -0.5, "r.time_emb_proj.bias": 0.1 * torch.randn(Cout, generator=g),
          "r.norm2.weight": 1.0 + 0.1 * torch.randn(Cout, generator=g), "r.norm2.bias": 0.1 * torch.randn(Cout, generator=g),
          "r.conv2.weight": torch.randn(Cout, Cout, 3, 3, generator=g) * (9 * Cout) ** -0.5, "r.conv2.bias": 0.1 * torch.randn(Cout, generator=g),
          "r.conv_shortcut.weight": torch.randn(Cout, Cin, 1, 1, generator=g) * Cin ** -0.5, "r.conv_shortcut.bias": 0.1 * torch.randn(Cout, generator=g),
          "pre.weight": torch.randn(Cin, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5, "pre.bias": 0.1 * torch.randn(Cin, generator=g)}
    sd = {k: q16(v) for k, v in sd.items()}
    W = packing.pack_state_dict(sd, "cuda")
    B, H = 2, 16
    x0 = q16(torch.randn(B, Cin, H, H, generator=g))
    temb = q16(torch.randn(B, T, generator=g))
    sh_ref = F.linear(F.silu(temb), sd["r.time_emb_proj.weight"], sd["r.time_emb_proj.bias"])
    x = F.conv2d(x0, sd["pre.weight"], sd["pre.bias"], padding=1)
    h = F.conv2d(F.silu(F.group_norm(x, G, sd["r.norm1.weight"], sd["r.norm1.bias"], 1e-5)), sd["r.conv1.weight"], sd["r.conv1.bias"], padding=1)
    h = h + sh_ref[:, :, None, None]
    h = F.conv2d(F.silu(F.group_norm(h, G, sd["r.norm2.weight"], sd["r.norm2.bias"], 1e-5)), sd["r.conv2.weight"], sd["r.conv2.bias"], padding=1)
    ref = h + F.conv2d(x, sd["r.conv_shortcut.weight"], sd["r.conv_shortcut.bias"])
    outs = {}
    for bridge in (True, False):
        E = Engine("cuda:0", record=True, autotune=False, gn_bridge=bridge)
        E.gn_fuse_max_rows = 4096  # (the consumer-side normalisation is opt-in by rows: GN_BRIDGE_FUSE_MAX_ROWS)
        E.gn_reduce_fuse = False   # (the other route of round 5 -- GroupNorm in the split-K reduce, tests/test_norm_out_gpu.py -- would take these)
        x0d = x0.permute(0, 2, 3, 1).contiguous().half().cuda()
        shifts = q16(sh_ref).half().cuda()
        xd = E.conv2d(x0d, W["pre.weight"], W["pre.bias"], name="pre")
        Wm = dict(W)
        Wm["__meta__"] = {"temb_slices": {"r": (0, Cout)}}
        y = graphs.emit_resnet(E, Wm, "r", xd, None, shifts, G, 1e-5)
        kinds = [(m["kind"], m.get("shape")) for m in E.meta]
        gn_ops = [k for k in kinds if k[0] == "groupnorm"]
        if bridge:
            assert not gn_ops, f"both GroupNorms of the block fit the consumer's A path at {B * H * H} rows: {gn_ops}"
        else:
            assert len(gn_ops) == 2
        E.run()
        E.synchronize()
        a = y.clone()
        E.run()
        E.synchronize()
        assert torch.equal(a, y), "two replays of the recorded program must agree bit for bit"
        outs[bridge] = a.float().permute(0, 3, 1, 2).cpu()
    assert rel_l2(outs[True], ref) < 2e-3 and rel_l2(outs[False], ref) < 2e-3, (rel_l2(outs[True], ref), rel_l2(outs[False], ref))
    assert rel_l2(outs[True], outs[False]) < 6e-4, rel_l2(outs[True], outs[False])
    # above the row gate the GroupNorm is ONE apply launch fed by the producer's statistics
    E = Engine("cuda:0", record=True, autotune=False, gn_bridge=True)
    E.gn_fuse_max_rows = 0
    E.gn_reduce_fuse = False
    xd = E.conv2d(x0.permute(0, 2, 3, 1).contiguous().half().cuda(), W["pre.weight"], W["pre.bias"], name="pre")
    Wm = dict(W)
    Wm["__meta__"] = {"temb_slices": {"r": (0, Cout)}}
    y = graphs.emit_resnet(E, Wm, "r", xd, None, q16(sh_ref).half().cuda(), G, 1e-5)
    gn_ops = [m for m in E.meta if m["kind"] == "groupnorm"]
    assert len(gn_ops) == 2 and all(m["shape"][-1] == "st" for m in gn_ops), gn_ops
    E.run()
    E.synchronize()
    assert rel_l2(y.float().permute(0, 3, 1, 2).cpu(), ref) < 2e-3


@pytest.mark.parametrize("fuse_rows,graph", [(0, False), (4096, False), (4096, True)])
def test_tiny_pipeline_with_the_bridge_on(monkeypatch, fuse_rows, graph):
    """The whole recorded call (CLIP, 5 x (ControlNet || UNet) on two streams, VAE) with GN_BRIDGE=1 -- the apply-from-statistics launches
    (fuse_rows 0) and the consumer-side normalisation (4096) -- against the same pipeline with the bridge off: latents within f16 noise, uint8
    images within 1 LSB almost everywhere; a second call is bit-identical (the arena's memset op, integer atomics); as a captured hipGraph too."""
    import numpy as np

    from genima_amd import configs, weights
    from genima_amd.pipeline import StableDiffusionControlNetPipeline

    fam = configs.family("tiny")
    B, steps = 2, 3
    img = torch.from_numpy(weights.counter_bytes(3, "ctrl", B * 128 * 128 * 3).reshape(B, 128, 128, 3))
    lat = q16(torch.randn(B, 4, 16, 16, generator=torch.Generator().manual_seed(2))).half()
    outs = {}
    for on in (False, True):
        monkeypatch.setenv("GN_BRIDGE", "1" if on else "0")
        monkeypatch.setenv("GN_BRIDGE_FUSE_MAX_ROWS", str(fuse_rows))
        pipe = StableDiffusionControlNetPipeline.from_synthetic(fam, seed=20)
        pipe.to("cuda")
        pipe.enable_hip_graph(graph and on)
        ids = pipe.encode_ids(["tiled perspectives of a robot arm executing 'open the box'"] * B)
        a = pipe(prompt_ids=ids, image=img, num_inference_steps=steps, guidance_scale=0.0, latents=lat, output_type="np").images
        io = pipe.program(B, 128, 128, steps)
        kinds = [m["kind"] + ("/st" if m.get("shape") and m["shape"][-1] == "st" else "") for m in io.engine.meta]
        if on:
            assert kinds.count("groupnorm/st") > 10, "the recorded program must contain GroupNorm launches fed by producer statistics"
            if fuse_rows:
                assert kinds.count("groupnorm") + kinds.count("groupnorm/st") < outs["n_gn"], "some GroupNorms must have moved into their consumers"
        else:
            outs["n_gn"] = kinds.count("groupnorm")
        b = pipe(prompt_ids=ids, image=img, num_inference_steps=steps, guidance_scale=0.0, latents=lat, output_type="np").images
        assert np.array_equal(a, b), "a second call must be bit-identical"
        outs[on] = (a, io.latents.float().cpu().clone())
    d = np.abs(outs[True][0].astype(np.int32) - outs[False][0].astype(np.int32))
    e = rel_l2(outs[True][1], outs[False][1])
    print(f"bridge on vs off (fuse_rows {fuse_rows}): latents rel-L2 {e:.2e}; uint8 mean |diff| {d.mean():.4f}, max {d.max()}")
    assert e < 3e-3 and d.mean() < 0.3 and (d > 2).mean() < 1e-2
