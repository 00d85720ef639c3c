"""-m gpu: the nearest-2x upsample + 3x3 conv of diffusers' Upsample2D (UNet up_blocks.*.upsamplers.0, the VAE decoder's)
as four 2x2 PHASE convs on the source pixels (packing.pack_upsample_phases + Engine.conv2d_up2x: gn_gemm with KH = KW = 2, asymmetric
padding and the two-level output row pitch), against fp32 torch and against the fused-upsample 3x3 launch it replaces."""
import pytest
import torch
import torch.nn.functional as F

from genima_amd.packing import pack_conv_weight, pack_upsample_phases
from util import assert_close, randn_h, rel_l2

pytestmark = pytest.mark.gpu


def _ref(x, w, b):
    up = F.interpolate(x.float().cpu().permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest")
    return F.conv2d(up, w.float().cpu(), b.float().cpu(), padding=1).permute(0, 2, 3, 1)


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64), (1, 8, 24, 128, 72), (3, 4, 4, 64, 320)])
def test_conv2d_up2x_matches_torch_and_the_fused_launch(engine, shape):
    B, H, W, Cin, Cout = shape
    x = randn_h(B, H, W, Cin, seed=1)
    w = randn_h(Cout, Cin, 3, 3, seed=2, scale=(9 * Cin) ** -0.5)
    b = randn_h(Cout, seed=3, scale=0.3)
    w4 = pack_upsample_phases(w.float().cpu()).cuda()
    y = engine.conv2d_up2x(x, w4, b)   # one launch, blockIdx.z = phase
    assert tuple(y.shape) == (B, 2 * H, 2 * W, Cout)
    assert_close(y, _ref(x, w, b), what=f"phase upsample conv {shape}")
    engine.up_phases_one_launch = False
    try:
        y4 = engine.conv2d_up2x(x, w4, b)  # four launches, one per phase
    finally:
        engine.up_phases_one_launch = True
    assert torch.equal(y, y4), "the one-launch and the four-launch form run the same arithmetic"
    old = engine.conv2d(x, pack_conv_weight(w.float().cpu()).cuda(), b, upsample2x=True)
    assert rel_l2(y, old.float().cpu()) < 6e-4


@pytest.mark.parametrize("tile", range(1, 25))
def test_two_level_row_pitch_every_tile(engine, tile):
    """The strided-view output (gn_gemm_desc.out_row_width / ldo_hi) through every block tile, with and without split-K."""
    E = engine
    old, old_auto = getattr(E, "no_table", False), E.autotune
    E.no_table, E.autotune = True, False
    E.lib.gn_set_gemm_tile_override(tile - 1)
    try:
        B, H, W, Cin, Cout = 2, 8, 12, 64, 136
        x = randn_h(B, H, W, Cin, seed=5)
        w = randn_h(Cout, Cin, 3, 3, seed=6, scale=(9 * Cin) ** -0.5)
        b = randn_h(Cout, seed=7, scale=0.3)
        w4 = pack_upsample_phases(w.float().cpu()).cuda()
        ref = _ref(x, w, b)
        assert_close(E.conv2d_up2x(x, w4, b), ref, what=f"tile {tile} (one launch)")
        out = torch.full((B, 2 * H, 2 * W, Cout), float("nan"), dtype=torch.float16, device="cuda")
        for dy in (0, 1):
            for dx in (0, 1):
                E.conv2d(x, w4[2 * dy + dx], b, ksize=2, pad=(1 - dy, 1 - dx, dy, dx), out=out[:, dy::2, dx::2, :], splitk=2)
        assert_close(out, ref, what=f"tile {tile} split-K")
    finally:
        E.lib.gn_set_gemm_tile_override(-1)
        E.no_table, E.autotune = old, old_auto
