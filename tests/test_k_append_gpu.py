"""-m gpu: gn_gemm_desc.k_append -- a 1x1 conv of the block input appended along the K axis of a 3x3 conv: diffusers ResnetBlock2D's
conv2(h) + conv_shortcut(x) (inside `self.pipe(...)`, controller/agent/sd_controlnet_agent.py:67-76) as ONE launch, against the two
launches it replaces and an fp32 restatement, on every LDS-DMA tile, with a K split, and with a concatenated block input (two tensors)."""
import pytest
import torch
import torch.nn.functional as F

from genima_amd.engine import Engine
from genima_amd.packing import pack_conv_weight
from util import assert_close, randn_h

pytestmark = pytest.mark.gpu


def _ref(h, x, w3, b3, w1, b1):
    y = F.conv2d(h.float().permute(0, 3, 1, 2), w3.float(), b3.float(), padding=1) + F.conv2d(x.float().permute(0, 3, 1, 2), w1.float(), b1.float())
    return y.permute(0, 2, 3, 1)


def _problem(B, H, C, Cx, N, seed=0):
    h, x = randn_h(B, H, H, C, seed=seed), randn_h(B, H, H, Cx, seed=seed + 1)
    w3, b3 = randn_h(N, C, 3, 3, seed=seed + 2, scale=0.03), randn_h(N, seed=seed + 3)
    w1, b1 = randn_h(N, Cx, 1, 1, seed=seed + 4, scale=0.03), randn_h(N, seed=seed + 5)
    wcat = torch.cat([pack_conv_weight(w3.float().cpu()).cuda(), pack_conv_weight(w1.float().cpu()).cuda()], dim=1).contiguous()
    bcat = (b3.float() + b1.float()).half()
    return h, x, w3, b3, w1, b1, wcat, bcat


@pytest.mark.parametrize("tile", [0] + list(range(7, 25)))
def test_appended_shortcut_every_dma_tile(tile):
    E = Engine("cuda:0")
    E.autotune = False
    h, x, w3, b3, w1, b1, wcat, bcat = _problem(2, 16, 128, 192, 160, seed=tile)
    ref = _ref(h, x, w3, b3, w1, b1)
    if tile:
        E.lib.gn_set_gemm_tile_override(tile - 1)
    try:
        y = E.conv2d(h, wcat, bcat, append=x)
    finally:
        E.lib.gn_set_gemm_tile_override(-1)
    E.synchronize()
    assert_close(y, ref, 1e-3, f"tile {tile}")


@pytest.mark.parametrize("B,H,C,C2,C3,N,splitk", [(8, 8, 1280, 1280, 1280, 1280, 4), (1, 16, 640, 640, 320, 640, 3), (2, 32, 320, 320, 0, 320, 1),
                                                  (2, 64, 320, 640, 320, 320, 0), (1, 8, 1280, 1280, 640, 1280, 0)])
def test_appended_shortcut_matches_the_two_launches(B, H, C, C2, C3, N, splitk):
    E = Engine("cuda:0")
    Cx = C2 + C3
    h, x, w3, b3, w1, b1, wcat, bcat = _problem(B, H, C, Cx, N, seed=H)
    xa, xb = (x[..., :C2].contiguous(), x[..., C2:].contiguous()) if C3 else (x, None)
    y = E.conv2d(h, wcat, bcat, append=xa, append2=xb, splitk=splitk)
    sc = E.conv2d(xa, pack_conv_weight(w1.float().cpu()).cuda(), b1, ksize=1, x2=xb)
    y2 = E.conv2d(h, pack_conv_weight(w3.float().cpu()).cuda(), b3, residual=sc)
    E.synchronize()
    ref = _ref(h, x, w3, b3, w1, b1)
    e1, e2 = assert_close(y, ref, 1e-3, "appended"), assert_close(y2, ref, 1e-3, "two launches")
    assert e1 <= 1.2 * e2 + 1e-5  # one rounding instead of two: no further from fp32 than the launches it replaces


@pytest.mark.parametrize("tile", [0] + list(range(7, 25)))
def test_dense_append_every_dma_tile(tile):
    """y = [g | h] @ w.T + b + x on every LDS-DMA tile (rows not a multiple of any tile, the second operand with its own row stride)."""
    E = Engine("cuda:0")
    E.autotune = False
    M, K1, K2, N = 616, 256, 192, 320
    g, hbig, x = randn_h(M, K1, seed=tile), randn_h(M, K2 + 64, seed=tile + 1), randn_h(M, N, seed=tile + 2)
    h = hbig[:, :K2]  # row stride K2 + 64
    w, b = randn_h(N, K1 + K2, seed=tile + 3, scale=0.05), randn_h(N, seed=tile + 4)
    ref = torch.cat([g.float(), h.float()], dim=1) @ w.float().t() + b.float() + x.float()
    if tile:
        E.lib.gn_set_gemm_tile_override(tile - 1)
    try:
        y = E.linear(g, w, b, residual=x, append=h)
    finally:
        E.lib.gn_set_gemm_tile_override(-1)
    E.synchronize()
    assert_close(y, ref, 1e-3, f"tile {tile}")


@pytest.mark.parametrize("M,C,splitk", [(2048, 1280, 0), (8192, 640, 0), (64, 1280, 4), (1024, 320, 0)])
def test_composed_ff_out_and_proj_out(M, C, splitk):
    """Transformer2DModel's proj_out(ff.net.2(g) + h) + x as ONE GEMM over [g | h] with the composed weight (packing `ffo_pout`) against the
    two launches it replaces and fp32."""
    E = Engine("cuda:0")
    g, h, x = randn_h(M, 4 * C, seed=1), randn_h(M, C, seed=2), randn_h(M, C, seed=3)
    wf, bf = randn_h(C, 4 * C, seed=4, scale=0.02), randn_h(C, seed=5, scale=0.1)
    wp, bp = randn_h(C, C, seed=6, scale=0.03), randn_h(C, seed=7, scale=0.1)
    wcat = torch.cat([wp.double() @ wf.double(), wp.double()], dim=1).half().contiguous()
    bcat = (wp.double() @ bf.double() + bp.double()).half()
    y = E.linear(g, wcat, bcat, residual=x, append=h, splitk=splitk)
    y2 = E.linear(E.linear(g, wf, bf, residual=h), wp, bp, residual=x)
    E.synchronize()
    ref = ((g.float() @ wf.float().t() + bf.float() + h.float()) @ wp.float().t() + bp.float() + x.float())
    e1, e2 = assert_close(y, ref, 1e-3, "composed"), assert_close(y2, ref, 1e-3, "two launches")
    assert e1 <= 1.5 * e2 + 1e-5
