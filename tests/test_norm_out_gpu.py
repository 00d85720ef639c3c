"""-m gpu: GroupNorm (+ SiLU) inside the split-K reduce of the launch that wrote its input (gn_gemm_desc.norm_out, csrc/gemm.hip
splitk_reduce_gn_kernel): diffusers ResnetBlock2D's conv1 -> norm2 -> SiLU and conv2 -> the next block's norm1 / Transformer2DModel.norm
(inside `self.pipe(...)`, controller/agent/sd_controlnet_agent.py:67-76) without the GroupNorm launch, at the latent levels whose convs split K.

  * kernel level: conv / Linear with a K split and norm_out against torch fp32 `silu(group_norm(conv(x)))` on the same f16-rounded inputs (1e-3),
    the raw output against the plain reduce bit for bit, the normalised one against the GroupNorm launch it replaces;
  * recorded programs: a GroupNorm recorded behind a K-split conv moves into its reduce (no groupnorm op), behind an unsplit one it stays."""
import pytest
import torch
import torch.nn.functional as F

from genima_amd._lib import ACT_NONE, ACT_SILU, GenimaHipError
from genima_amd.engine import Engine, Norm
from genima_amd.packing import pack_conv_weight
from util import assert_close, q16, randn_h, rel_l2

pytestmark = pytest.mark.gpu


def _nchw(t):
    return t.float().permute(0, 3, 1, 2)


@pytest.mark.parametrize("B,H,C,N,splitk,tile,silu,shift,residual", [
    (8, 8, 1280, 1280, 6, 17, True, True, False),   # the 8x8 level at B = 8: conv1 + time shift -> norm2 -> SiLU
    (1, 8, 1280, 1280, 8, 18, True, False, True),   # ... at B = 1: conv2 + residual -> the next block's norm1
    (2, 16, 640, 1280, 3, 17, False, False, False),  # Transformer2DModel.norm (no activation) behind a resnet's conv2
    (1, 32, 320, 640, 2, 23, True, True, False),     # 32x32 level: 1024 rows x 20 channels per slab
    (4, 16, 1280, 1280, 5, 15, True, False, True)])  # the ping-pong tile's K split
def test_conv_split_k_with_groupnorm_in_the_reduce(B, H, C, N, splitk, tile, silu, shift, residual):
    E = Engine("cuda:0")
    E.autotune = False
    g = torch.Generator().manual_seed(H + C + splitk)
    x = q16(torch.randn(B, H, H, C, generator=g))
    w = q16(torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5)
    b = q16(torch.randn(N, generator=g) * 0.2)
    sh = q16(torch.randn(B, N, generator=g) * 0.3) if shift else None
    res = q16(torch.randn(B, H, H, N, generator=g)) if residual else None
    gamma, beta = q16(1.0 + 0.2 * torch.randn(N, generator=g)), q16(0.2 * torch.randn(N, generator=g))
    G, eps = 32, 1e-5
    h_ref = F.conv2d(_nchw(x), w, b, padding=1)
    if shift:
        h_ref = h_ref + sh[:, :, None, None]
    if residual:
        h_ref = h_ref + _nchw(res)
    n_ref = F.group_norm(q16(h_ref), G, gamma, beta, eps)
    n_ref = F.silu(n_ref) if silu else n_ref
    xd, wd, bd = x.half().cuda(), pack_conv_weight(w).cuda(), b.half().cuda()
    shd, rd = (None if sh is None else sh.half().cuda()), (None if res is None else res.half().cuda())
    gd, bed = gamma.half().cuda(), beta.half().cuda()
    act = ACT_SILU if silu else ACT_NONE
    E.lib.gn_set_gemm_tile_override(tile - 1)
    try:
        h, y = E.conv2d(xd, wd, bd, shift=shd, residual=rd, splitk=splitk, norm_out=Norm(gd, bed, G, eps, act))
        h0 = E.conv2d(xd, wd, bd, shift=shd, residual=rd, splitk=splitk)
    finally:
        E.lib.gn_set_gemm_tile_override(-1)
    y0 = E.groupnorm(h0, gd, bed, G, eps, act=act)
    E.synchronize()
    assert torch.equal(h, h0), "the raw output must not depend on who reduces the K slices"
    assert_close(_nchw(h), h_ref, 1e-3, "raw conv output")
    assert_close(_nchw(y), n_ref, 1e-3, "GroupNorm in the reduce")
    assert rel_l2(y, y0.float()) < 1e-4, rel_l2(y, y0.float())


def test_linear_split_k_norm_out_and_refusals():
    E = Engine("cuda:0")
    E.autotune = False
    B, R, K, N = 2, 64, 2560, 1280
    a, w, b = randn_h(B, R, K, seed=1), randn_h(N, K, seed=2, scale=0.02), randn_h(N, seed=3)
    gamma, beta = q16(1.0 + 0.1 * torch.randn(N)).half().cuda(), q16(0.1 * torch.randn(N)).half().cuda()
    h, y = E.linear(a, w, b, splitk=4, rows_per_batch=0, norm_out=Norm(gamma, beta, 32, 1e-6, ACT_NONE))
    E.synchronize()
    ref = a.float() @ w.float().t() + b.float()
    assert_close(h, ref, 1e-3, "linear raw")
    n_ref = F.group_norm(q16(ref).permute(0, 2, 1).cpu(), 32, gamma.float().cpu(), beta.float().cpu(), 1e-6).permute(0, 2, 1)
    assert_close(y, n_ref, 1e-3, "linear + GroupNorm in the reduce")
    with pytest.raises(GenimaHipError):  # a plan that does not split K cannot carry the fusion: loud, not silent
        E.linear(a, w, b, splitk=1, norm_out=Norm(gamma, beta, 32, 1e-6, ACT_NONE))


def test_recorded_groupnorm_moves_into_the_split_k_reduce():
    g = torch.Generator().manual_seed(5)
    B, H, C, N, G = 2, 8, 1280, 1280, 32
    x = q16(torch.randn(B, H, H, C, generator=g)).half().cuda()
    w1, b1 = pack_conv_weight(q16(torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5)).cuda(), q16(torch.randn(N, generator=g) * 0.1).half().cuda()
    w2, b2 = pack_conv_weight(q16(torch.randn(N, N, 3, 3, generator=g) * (9 * N) ** -0.5)).cuda(), q16(torch.randn(N, generator=g) * 0.1).half().cuda()
    gamma, beta = q16(1.0 + 0.1 * torch.randn(N, generator=g)).half().cuda(), q16(0.1 * torch.randn(N, generator=g)).half().cuda()
    outs = {}
    for fuse in (True, False):
        E = Engine("cuda:0", record=True, autotune=False)
        E.gn_reduce_fuse, E.gn_reduce_fuse_min_slabs = fuse, 0  # (production gates the route by slabs = B x groups >= 128)
        h = E.conv2d(x, w1, b1, splitk=4, name="c1")
        y = E.conv2d(h, w2, b2, norm=Norm(gamma, beta, G, 1e-5, ACT_SILU, "n2"), splitk=4, name="c2")
        kinds = [m["kind"] for m in E.meta]
        assert kinds.count("groupnorm") == (0 if fuse else 1), kinds
        assert sum(1 for m in E.meta if m.get("norm_out")) == (1 if fuse else 0)
        E.run()
        E.synchronize()
        a = y.clone()
        E.run()
        E.synchronize()
        assert torch.equal(a, y)
        outs[fuse] = a.float()
    assert rel_l2(outs[True], outs[False]) < 2e-4, rel_l2(outs[True], outs[False])
    # behind an unsplit conv the GroupNorm stays a launch
    E = Engine("cuda:0", record=True, autotune=False)
    E.gn_reduce_fuse_min_slabs = 0
    h = E.conv2d(x, w1, b1, splitk=1, name="c1")
    E.conv2d(h, w2, b2, norm=Norm(gamma, beta, G, 1e-5, ACT_SILU, "n2"), splitk=4, name="c2")
    assert [m["kind"] for m in E.meta].count("groupnorm") == 1
