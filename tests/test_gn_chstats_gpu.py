"""-m gpu: GroupNorm statistics taken in the epilogue of the producing conv / Linear (gn_gemm_desc.chstats -> gn_groupnorm_desc.chstats;
the ResnetBlock2D / Transformer2DModel GroupNorms of the reference, torch native_group_norm's statistics pass: SURVEY.md K2).

  * every block tile: the per-(32-row band, channel) sums / sums of squares equal those of the STORED f16 output (f64 reference) --
    bias, time shift, residual and activation included, ragged channel count (N % 32 != 0);
  * the consumer: GroupNorm(+SiLU) from producer statistics, one- and two-source, equals the stand-alone GroupNorm kernels on the same
    tensors (same f16 inputs; statistics differ only in summation order) and the fp32 torch reference;
  * the recorded-program plumbing (Engine: conv2d / linear ``stats=True`` -> groupnorm) replays to the eager result."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from genima_amd._lib import ACT_SILU, GemmDesc, GroupNormDesc, check
from genima_amd.engine import Engine, _ptr
from util import assert_close, randn_h

pytestmark = pytest.mark.gpu
N_TILES = 23


def _band_stats(y: torch.Tensor):
    """y f16 [M, N] -> f64 [M/32, 2, N]."""
    v = y.double().view(y.shape[0] // 32, 32, y.shape[1])
    return torch.stack([v.sum(1), (v * v).sum(1)], dim=1)


class _Capture:
    """Run one Engine GEMM with a chstats buffer attached (the eager path of the engine never asks for statistics itself)."""

    def __init__(self, E):
        self.E, self.st = E, None

    def __enter__(self):
        E, outer = self.E, self
        self._orig = E._gemm

        def gemm(d, keep, stats=False):
            if d.tile == 0 and d.splitk == 0:
                E.apply_plan(d, 0)
            d.splitk = 1
            assert int(E.lib.gn_gemm_chstats_band(C.byref(d))) == 32
            outer.st = torch.full((int(d.M) // 32, 2, int(d.N)), float("nan"), dtype=torch.float32, device="cuda")
            d.chstats = outer.st.data_ptr()
            E.run_gemm(d)
        E._gemm = gemm
        return self

    def __exit__(self, *a):
        self.E._gemm = self._orig


@pytest.mark.parametrize("tile", range(1, N_TILES + 1))
def test_chstats_every_tile(engine, tile):
    E = engine
    old, old_auto = getattr(E, "no_table", False), E.autotune
    E.no_table, E.autotune = True, False
    E.lib.gn_set_gemm_tile_override(tile - 1)
    try:
        # conv (ResnetBlock2D conv1: + bias + per-sample time shift) with a ragged channel count
        B, H, W, Cin, Cout = 2, 16, 24, 64, 136
        x, w, b = randn_h(B, H, W, Cin, seed=5), randn_h(Cout, 9 * Cin, seed=6, scale=(9 * Cin) ** -0.5), randn_h(Cout, seed=7, scale=0.3)
        shift, res = randn_h(B, Cout, seed=8), randn_h(B, H, W, Cout, seed=9)
        with _Capture(E) as cap:
            y = E.conv2d(x, w, b, shift=shift)
        want = _band_stats(y.view(-1, Cout)).float()
        assert torch.isfinite(cap.st).all(), f"tile {tile}: a (band, channel) slot was never written"
        assert_close(cap.st, want, rel=2e-6, what=f"tile {tile} conv chstats")
        with _Capture(E) as cap:
            y = E.conv2d(x, w, b, residual=res, act=1)   # silu(conv) + residual
        assert_close(cap.st, _band_stats(y.view(-1, Cout)).float(), rel=2e-6, what=f"tile {tile} conv+res chstats")
        # Linear (Transformer2DModel.proj_out + residual)
        M, N, K = 1024, 328, 192
        a, wl, bl, r = randn_h(M, K, seed=1), randn_h(N, K, seed=2, scale=K ** -0.5), randn_h(N, seed=3, scale=0.3), randn_h(M, N, seed=4)
        with _Capture(E) as cap:
            y = E.linear(a, wl, bl, residual=r)
        assert_close(y, a.float().cpu() @ wl.float().cpu().t() + bl.float().cpu() + r.float().cpu(), what=f"tile {tile} linear (with chstats)")
        assert_close(cap.st, _band_stats(y).float(), rel=2e-6, what=f"tile {tile} linear chstats")
    finally:
        E.lib.gn_set_gemm_tile_override(-1)
        E.no_table, E.autotune = old, old_auto


def _gn_from_stats(E, x, x2, gamma, beta, groups, eps, act, st1, st2):
    B, C1 = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C1)
    C2 = x2.shape[-1] if x2 is not None else 0
    out = torch.empty(tuple(x.shape[:-1]) + (C1 + C2,), dtype=torch.float16, device="cuda")
    d = GroupNormDesc()
    d.x, d.x2, d.gamma, d.beta, d.y = _ptr(x), _ptr(x2), _ptr(gamma), _ptr(beta), _ptr(out)
    d.B, d.HW, d.C1, d.C2, d.groups, d.act, d.eps = B, HW, C1, C2, groups, act, eps
    ws = E._workspace(int(E.lib.gn_groupnorm_workspace_bytes(C.byref(d))))
    d.workspace = ws.data_ptr()
    d.chstats, d.chstats2 = _ptr(st1), _ptr(st2)
    check(E.lib.gn_groupnorm_fwd(E._ctx, C.byref(d)), "gn_groupnorm_fwd")
    return out


@pytest.mark.parametrize("shape", [(2, 64, 64, 320, 0), (2, 32, 32, 640, 320), (1, 128, 128, 128, 0), (3, 16, 32, 96, 160)])
def test_groupnorm_from_producer_stats(engine, shape):
    E = engine
    B, H, W, C1, C2 = shape
    G, eps = 32, 1e-5
    x = randn_h(B, H, W, C1, seed=11, scale=1.5) + 0.25
    x2 = (randn_h(B, H, W, C2, seed=12, scale=0.7) - 0.5) if C2 else None
    gamma, beta = randn_h(C1 + C2, seed=13, scale=0.5) + 1.0, randn_h(C1 + C2, seed=14, scale=0.3)
    st1 = _band_stats(x.view(-1, C1)).float().contiguous()
    st2 = _band_stats(x2.view(-1, C2)).float().contiguous() if C2 else None
    for act in (0, ACT_SILU):
        y = _gn_from_stats(E, x, x2, gamma, beta, G, eps, act, st1, st2)
        y_old = E.groupnorm(x, gamma, beta, G, eps, act=act, x2=x2)
        cat = torch.cat([x, x2], dim=-1) if C2 else x
        ref = F.group_norm(cat.float().cpu().permute(0, 3, 1, 2), G, gamma.float().cpu(), beta.float().cpu(), eps)
        ref = (F.silu(ref) if act else ref).permute(0, 2, 3, 1)
        assert_close(y, ref, what=f"GroupNorm from producer stats {shape} act {act}")
        assert float((y.float() - y_old.float()).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-3


def test_recorded_program_uses_producer_stats():
    """conv2d(stats=True) -> groupnorm in a recorded program: the GroupNorm op carries the producer's statistics buffer, the replay equals
    the eager engine (stand-alone GroupNorm) within rounding and the fp32 reference within the kernel bar; a tensor overwritten by an op
    that takes no statistics falls back to the stand-alone kernels."""
    B, H, W, Cin, Cout, G = 2, 64, 64, 64, 320, 32
    x, w, b = randn_h(B, H, W, Cin, seed=21), randn_h(Cout, 9 * Cin, seed=22, scale=(9 * Cin) ** -0.5), randn_h(Cout, seed=23, scale=0.3)
    gamma, beta = randn_h(Cout, seed=24, scale=0.5) + 1.0, randn_h(Cout, seed=25, scale=0.3)
    R = Engine("cuda:0", record=True)
    R.gn_stats_min_bytes = 1
    h = R.conv2d(x, w, b, name="c", stats=True)
    assert h.data_ptr() in R._chstats
    y = R.groupnorm(h, gamma, beta, G, 1e-5, act=ACT_SILU, name="n")
    h2 = R.conv2d(x, w, b, name="c2", stats=True)
    z = R.add(h2, h2, name="sum")          # an elementwise producer: no statistics
    assert z.data_ptr() not in R._chstats
    y2 = R.groupnorm(z, gamma, beta, G, 1e-5, name="n2")
    R.run()
    torch.cuda.synchronize()
    E = Engine("cuda:0")
    he = E.conv2d(x, w, b)
    ye = E.groupnorm(he, gamma, beta, G, 1e-5, act=ACT_SILU)
    assert torch.equal(h, he)
    ref = F.silu(F.group_norm(he.float().cpu().permute(0, 3, 1, 2), G, gamma.float().cpu(), beta.float().cpu(), 1e-5)).permute(0, 2, 3, 1)
    assert_close(y, ref, what="recorded GroupNorm from conv statistics")
    assert float((y.float() - ye.float()).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-3
    assert torch.equal(y2, E.groupnorm(E.add(he, he), gamma, beta, G, 1e-5))
    # replay determinism
    y_first = y.clone()
    R.run()
    torch.cuda.synchronize()
    assert torch.equal(y, y_first)
