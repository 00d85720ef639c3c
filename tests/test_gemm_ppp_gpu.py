"""-m gpu: the persistent skewed ping-pong GEMM (tile 25, csrc/gemm_ppp.hip) -- the conv / GEMM calls inside `self.pipe(...)`
(controller/agent/sd_controlnet_agent.py:67-76) on a grid of one workgroup per CU that walks the tile list.

  * against torch fp32 on the same f16 inputs at the kernel bar (1e-3), every epilogue the kernel carries (bias, SiLU, residual before / after the
    activation, time shift, scale), dense and conv (3x3, the four-phase upsampling conv), tile counts that exercise every segment kind: exactly one
    round (skew hand-offs only), 1.25 and 2.5 rounds (tail tiles split 4 and 2 ways along K), 3 rounds + a tail;
  * against tile 15 (gemm_pp.hip, one launch round per 256 tiles): a tile whose K range one workgroup walks is BIT-identical (every tile of the full
    rounds), a tile of the last partial round that is split along K differs by the rounding of a few f32 partial sums (<= 2e-4 relative on the
    tensor); the skewed walk (GN_PPP_SKEW=1, off by default) in a subprocess;
  * run to run bit-identical (fixed summation order of the hand-offs), no bounded wait ever gives up (gn_ppp_timeouts), flags self-clean (the second
    run reuses nothing stale: a recorded program replays the same flag region)."""
import pytest
import torch
import torch.nn.functional as F

from genima_amd._lib import ACT_NONE, ACT_SILU
from genima_amd.engine import Engine
from genima_amd.packing import pack_conv_weight
from util import assert_close, q16, randn_h, rel_l2

pytestmark = pytest.mark.gpu


def _with_tile(E, tile, fn):
    E.lib.gn_set_gemm_tile_override(tile - 1)
    try:
        y = fn()
        E.synchronize()
        return y
    finally:
        E.lib.gn_set_gemm_tile_override(-1)


def _ncu():
    return torch.cuda.get_device_properties(0).multi_processor_count


@pytest.mark.parametrize("M,N,K,act,res,res_first", [
    (16384, 1024, 640, ACT_NONE, False, False),    # 256 tiles: exactly one round
    (20480, 1024, 4096, ACT_SILU, False, False),   # 320 tiles: 1.25 rounds, the 64 tail tiles split along K (the planner's cost model splits long K only)
    (40960, 1024, 2048, ACT_NONE, True, False),    # 640 tiles: 2.5 rounds, tail split 2 ways, residual after the (absent) activation
    (8192, 2048, 320, ACT_SILU, True, True),       # 256 tiles, K = 5 tiles only, residual inside the activation
    (65536, 512, 1152, ACT_NONE, False, False),    # 512 tiles: two full rounds, no tail
    (28160, 2048, 256, ACT_SILU, True, False)])    # 880 tiles: 3 rounds + 112 tail tiles (2 ways), K = 4 tiles (the minimum)
def test_dense_vs_torch_and_tile15(M, N, K, act, res, res_first):
    if _ncu() != 256:
        pytest.skip("tile counts are written for 256 CUs")
    E = Engine("cuda:0")
    E.autotune = False
    x, w, b = randn_h(M, K, seed=1), randn_h(N, K, seed=2, scale=K ** -0.5), randn_h(N, seed=3)
    r = randn_h(M, N, seed=4) if res else None
    t0 = int(E.lib.gn_ppp_timeouts())

    def run():
        d = dict(act=act, residual=r)
        if res_first:
            # residual_before_act is a conv2d keyword; the dense path takes it through the descriptor: use a 1x1 conv view of the same problem
            return E.conv2d(x.view(1, M // 256, 256, K), w, b, ksize=1, pad=(0, 0, 0, 0), act=act, residual=r.view(1, M // 256, 256, N),
                            residual_before_act=True).view(M, N)
        return E.linear(x, w, b, **d)
    y25 = _with_tile(E, 25, run)
    y25b = _with_tile(E, 25, run)
    y15 = _with_tile(E, 15, run)
    ref = x.float() @ w.float().t() + b.float()
    if res and res_first:
        ref = ref + r.float()
    if act == ACT_SILU:
        ref = F.silu(ref)
    if res and not res_first:
        ref = ref + r.float()
    assert torch.equal(y25, y25b), "tile 25 must be bit-reproducible"
    assert_close(y25, ref, 1e-3, "tile 25 vs torch fp32")
    assert rel_l2(y25.float(), y15.float()) < 2e-4, rel_l2(y25.float(), y15.float())
    assert int(E.lib.gn_ppp_timeouts()) == t0, "a bounded hand-off wait gave up"
    # the tiles of the full rounds are computed by ONE workgroup over the whole K range: bit-identical to tile 15 (row-major tile order: the first
    # R * 256 tiles are the first R * 256 / tiles_n row bands); only the tail's tiles, where they are split along K, differ in rounding
    tiles_n = N // 256
    full = ((M // 256) * tiles_n // 256) * 256 // tiles_n * 256
    assert torch.equal(y25[:full], y15[:full]), "unshared tiles must be bit-identical to tile 15"


def test_skewed_walk_in_a_subprocess():
    """GN_PPP_SKEW=1 (read once per process by the library; off by default because it measured slower): workgroup c enters its first tile at K
    iteration c * nk / G and hands the partial sums to workgroup c - 1 -- every round-0 tile goes through a hand-off.  Same bars."""
    import os
    import subprocess
    import sys
    code = (
        "import torch, sys; sys.path.insert(0, 'tests');\n"
        "from genima_amd.engine import Engine; from util import randn_h, rel_l2\n"
        "E = Engine('cuda:0'); E.autotune = False\n"
        "for (M, N, K) in ((32768, 512, 640), (20480, 1024, 1280), (40960, 1024, 2048)):\n"
        "    x, w, b = randn_h(M, K, seed=1), randn_h(N, K, seed=2, scale=K ** -0.5), randn_h(N, seed=3)\n"
        "    ys = []\n"
        "    for t in (25, 25, 15):\n"
        "        E.lib.gn_set_gemm_tile_override(t - 1); ys.append(E.linear(x, w, b)); E.synchronize()\n"
        "    ref = x.float() @ w.float().t() + b.float()\n"
        "    assert torch.equal(ys[0], ys[1]); assert rel_l2(ys[0].float(), ys[2].float()) < 2e-4; assert rel_l2(ys[0].float(), ref) < 1e-3\n"
        "    assert not torch.equal(ys[0], ys[2]), 'the skewed walk shares every round-0 tile: some rounding must differ'\n"
        "assert int(E.lib.gn_ppp_timeouts()) == 0; print('OK')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, GN_PPP_SKEW="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("B,H,Cin,Cout,act,res,shift", [
    (4, 128, 128, 256, ACT_NONE, False, False),   # 256 tiles, K = 18 tiles (the VAE's 128 -> 256 level)
    (5, 128, 64, 256, ACT_SILU, True, False),     # 320 tiles: tail split 4 ways; residual
    (8, 64, 256, 512, ACT_NONE, False, True),     # 256 tiles of a 64 x 64 map (Wo = 64): time shift per sample
    (2, 256, 64, 512, ACT_NONE, True, False),     # 512 x 2 tiles: two rounds, Wo = 256
    (16, 32, 128, 512, ACT_SILU, False, False)])  # Wo = 32 < 64: a lane's rows sit two image rows apart
def test_conv3x3_vs_torch_and_tile15(B, H, Cin, Cout, act, res, shift):
    if _ncu() != 256:
        pytest.skip("tile counts are written for 256 CUs")
    E = Engine("cuda:0")
    E.autotune = False
    g = torch.Generator().manual_seed(B * 1000 + H + Cin)
    x = q16(torch.randn(B, H, H, Cin, generator=g))
    w = q16(torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5)
    b = q16(torch.randn(Cout, generator=g) * 0.2)
    r = q16(torch.randn(B, H, H, Cout, generator=g)) if res else None
    sh = q16(torch.randn(B, Cout, generator=g) * 0.3) if shift else None
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1)
    if shift:
        ref = ref + sh[:, :, None, None]
    if act == ACT_SILU:
        ref = F.silu(ref)
    if res:
        ref = ref + r.permute(0, 3, 1, 2)
    xd, wd, bd = x.half().cuda(), pack_conv_weight(w).cuda(), b.half().cuda()
    rd, sd = (None if r is None else r.half().cuda()), (None if sh is None else sh.half().cuda())
    t0 = int(E.lib.gn_ppp_timeouts())
    run = lambda: E.conv2d(xd, wd, bd, act=act, residual=rd, shift=sd)
    y25 = _with_tile(E, 25, run)
    y25b = _with_tile(E, 25, run)
    y15 = _with_tile(E, 15, run)
    assert torch.equal(y25, y25b)
    assert_close(y25.float().permute(0, 3, 1, 2), ref, 1e-3, "tile 25 conv vs torch fp32")
    assert rel_l2(y25.float(), y15.float()) < 2e-4
    assert int(E.lib.gn_ppp_timeouts()) == t0


def test_upsampling_phase_convs_and_recorded_replay():
    """The four phase convs of an Upsample2D as one persistent launch (blockIdx.z folded into the tile list: 4 x 256 tiles), eager and as a recorded
    program replayed three times (the same flag region every replay: the flags must come back clean)."""
    if _ncu() != 256:
        pytest.skip("tile counts are written for 256 CUs")
    from genima_amd.packing import pack_upsample_phases
    B, H, C, N = 4, 64, 256, 256   # phases of 16384 rows x 256 columns: 64 tiles each, 256 in all
    g = torch.Generator().manual_seed(5)
    x = q16(torch.randn(B, H, H, C, generator=g))
    w = q16(torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5)
    b = q16(torch.randn(N, generator=g) * 0.2)
    ref = F.conv2d(F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2, mode="nearest"), w, b, padding=1)
    w4 = pack_upsample_phases(w).cuda()
    E = Engine("cuda:0")
    E.autotune = False
    y25 = _with_tile(E, 25, lambda: E.conv2d_up2x(x.half().cuda(), w4, b.half().cuda()))
    y15 = _with_tile(E, 15, lambda: E.conv2d_up2x(x.half().cuda(), w4, b.half().cuda()))
    assert_close(y25.float().permute(0, 3, 1, 2), ref, 2e-3, "phase convs on tile 25 (the composed phase weights round once more than the 3x3)")
    assert rel_l2(y25.float(), y15.float()) < 2e-4
    R = Engine("cuda:0", record=True)
    R.autotune = False
    R.lib.gn_set_gemm_tile_override(24)
    try:
        xr = x.half().cuda()
        yr = R.conv2d_up2x(xr, w4, b.half().cuda(), name="up")
        for _ in range(3):
            yr.zero_()
            R.run()
            R.synchronize()
            assert torch.equal(yr, y25)
    finally:
        R.lib.gn_set_gemm_tile_override(-1)
    assert int(R.lib.gn_ppp_timeouts()) == 0


def test_ineligible_problems_fall_back_to_tile15():
    """A plan that names tile 25 for a problem it does not take (a partial tile, fewer tiles than CUs, GEGLU ...) runs tile 15 / the planner's fallback
    with the usual result -- gn_gemm never refuses a tile name."""
    E = Engine("cuda:0")
    E.autotune = False
    for (M, N, K) in ((1000, 512, 640), (4096, 512, 640), (16384, 1000, 640)):
        x, w, b = randn_h(M, K, seed=1), randn_h(N, K, seed=2, scale=0.04), randn_h(N, seed=3)
        y = _with_tile(E, 25, lambda: E.linear(x, w, b))
        assert_close(y, x.float() @ w.float().t() + b.float(), 1e-3)


@pytest.mark.parametrize("M,Nh,K", [
    (8192, 2560, 640),     # the 32 x 32 latent level's feed-forward projection at B = 8: 32 x 20 = 640 tiles, 2.5 rounds
    (2048, 5120, 1280),    # the 16 x 16 level: 8 x 40 = 320 tiles
    (32768, 1280, 320)])   # 128 x 10 = 1280 tiles, five rounds, K = 5 tiles
def test_feed_forward_variant_layernorm_fold_and_geglu(M, Nh, K):
    """BasicTransformerBlock.norm3 -> FeedForward.net[0] (GEGLU) as ONE persistent launch: the rows' LayerNorm statistics from the K loop's A fragments,
    hidden * gelu(gate) in the epilogue (W rows permuted in the loader so that a wave owns a hidden block and its gate block).  Against torch fp32
    layer_norm + linear + GEGLU at the kernel bar, against the tile the table names today (9: 128 x 128 LDS-DMA) within 3e-4 (the statistics are
    summed in another order), bit-reproducible; rows with a large mean so that the mean * c1 cancellation is exercised."""
    if _ncu() != 256:
        pytest.skip("tile counts are written for 256 CUs")
    from genima_amd.packing import pack_geglu
    E = Engine("cuda:0")
    E.autotune = False
    x = (randn_h(M, K, seed=31).float() + randn_h(M, 1, seed=32).float() * 3.0).half()
    w, b = randn_h(2 * Nh, K, seed=33, scale=K ** -0.5), randn_h(2 * Nh, seed=34, scale=0.3)
    gamma, beta = (1.0 + 0.3 * randn_h(K, seed=35).float()).half(), randn_h(K, seed=36, scale=0.2)
    ln = F.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5)
    full = ln @ w.float().t() + b.float()
    want = full[:, :Nh] * F.gelu(full[:, Nh:])
    wp, bp = pack_geglu(w.float().cpu(), b.float().cpu())
    wp, bp = wp.cuda(), bp.cuda()
    wg = (wp.float() * gamma.float()[None, :]).half().contiguous()
    c1 = wg.float().sum(dim=1).contiguous()
    c2 = (wp.float() @ beta.float() + bp.float()).half()
    run = lambda: E.linear(x, wg, c2, ln_c1=c1, act=5)
    y25 = _with_tile(E, 25, run)
    y25b = _with_tile(E, 25, run)
    y9 = _with_tile(E, 9, run)
    assert torch.equal(y25, y25b)
    assert_close(y25, want, 1e-3, "feed-forward variant vs torch fp32")
    assert rel_l2(y25.float(), y9.float()) < 3e-4, rel_l2(y25.float(), y9.float())
    assert int(E.lib.gn_ppp_timeouts()) == 0


def test_split_tail_inside_a_captured_graph():
    """A tile-25 launch whose last partial round is split along K (hand-off flags in use) captured into a hipGraph and replayed: the region it took at capture
    time stays its own (captured launches draw from a never-recycled half of the pool), the flags come back clean after every replay, and eager launches
    of the same problem in between (rotating through the other half) do not disturb it."""
    if _ncu() != 256:
        pytest.skip("tile counts are written for 256 CUs")
    M, N, K = 20480, 1024, 4096   # 320 tiles: 64 tail tiles, split along K
    x, w, b = randn_h(M, K, seed=1), randn_h(N, K, seed=2, scale=K ** -0.5), randn_h(N, seed=3)
    E = Engine("cuda:0")
    E.autotune = False
    y_eager = _with_tile(E, 25, lambda: E.linear(x, w, b))
    side = torch.cuda.Stream()
    R = Engine("cuda:0", record=True)
    R.autotune = False
    R.lib.gn_set_gemm_tile_override(24)
    try:
        yr = R.linear(x, w, b, name="y")
        R.use_stream(side)
        with torch.cuda.stream(side):
            R.run(); side.synchronize()
            R.capture()
            for _ in range(3):
                yr.zero_()
                R.launch(); side.synchronize()
                assert torch.equal(yr, y_eager)
                y2 = E.linear(x, w, b)   # (the override is process-wide: this is tile 25 too, eager, on the default stream)
                E.synchronize()
                assert torch.equal(y2, y_eager)
    finally:
        R.lib.gn_set_gemm_tile_override(-1)
    assert int(R.lib.gn_ppp_timeouts()) == 0
