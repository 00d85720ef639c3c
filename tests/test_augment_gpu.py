"""-m gpu: device-side ColorJitter / reflect-pad + crop (genima_amd/augment.py) vs the torch restatement of the torchvision ops the
reference's augment_data chains (oracle/augment_torch.py).  Colour math is f32 on both sides, the HIP result is stored as f16:
tolerance = 1.5e-3 absolute on [0, 1] values (three f16 ulps near 1.0); crop is a pure gather: bit-exact."""
import itertools

import pytest
import torch

from genima_amd import augment
from genima_amd.host import nchw_to_nhwc
from oracle import augment_torch as OA
from util import q16

pytestmark = pytest.mark.gpu


def _images(B=2, H=48, W=40, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, H, W, generator=g)
    x[0, :, :4] = x[0, :1, :4]      # grey rows (max == min branch of rgb -> hsv)
    x[1, :, 5:9] = 0.0              # black
    x[1, 0, 10:14] = 1.0            # saturated primaries / ties between channels
    x[1, 1, 12:16] = 1.0
    return q16(x)


@pytest.mark.parametrize("order", list(itertools.permutations(range(4)))[::3])
def test_color_jitter_matches_torchvision_restatement(engine, order):
    x = _images()
    for factors in ((1.17, 0.83, 1.08, 0.043), (0.81, 1.19, 0.91, -0.05), (1.0, 1.0, 1.0, 0.0)):
        ref = OA.color_jitter(x, order, factors)
        y = augment.color_jitter(engine, nchw_to_nhwc(x, 8).half().cuda(), order, factors)
        got = y[..., :3].permute(0, 3, 1, 2).float().cpu()
        err = float((got - ref).abs().max())
        assert err <= 1.5e-3, (order, factors, err)
        assert float(y[..., 3:].abs().max()) == 0.0


def test_reflect_pad_crop_is_exact(engine):
    x = _images(B=3, H=32, W=24, seed=1)
    xd = nchw_to_nhwc(x, 8).half().cuda()
    for i, j in ((0, 0), (4, 4), (2, 2), (1, 3), (4, 0)):
        ref = OA.reflect_pad_crop(x, i, j)
        got = augment.reflect_pad_crop(engine, xd, i, j)[..., :3].permute(0, 3, 1, 2).float().cpu()
        assert torch.equal(got, ref), (i, j)


def test_augment_data_follows_the_reference_chain(engine):
    """crop,colorjitter as in README.md:204: jitter on the conditioning image only, one shared crop; same draws as the restatement."""
    px, cond = _images(seed=2) * 2 - 1, _images(seed=3)
    batch = dict(pixel_values=nchw_to_nhwc(px, 8).half().cuda(), conditioning_pixel_values=nchw_to_nhwc(cond, 8).half().cuda(),
                 input_ids=torch.zeros(2, 77, dtype=torch.int64))
    out = augment.augment_data(engine, "crop,colorjitter", batch, generator=torch.Generator().manual_seed(7))
    g = torch.Generator().manual_seed(7)
    order, factors = augment.draw_color_jitter(g)
    i, j = augment.draw_crop(generator=g)
    ref_c = OA.reflect_pad_crop(OA.color_jitter(cond, order, factors), i, j)
    ref_p = OA.reflect_pad_crop(px, i, j)
    got_c = out["conditioning_pixel_values"][..., :3].permute(0, 3, 1, 2).float().cpu()
    got_p = out["pixel_values"][..., :3].permute(0, 3, 1, 2).float().cpu()
    assert float((got_c - ref_c).abs().max()) <= 1.5e-3
    assert torch.equal(got_p, q16(ref_p))
    with pytest.raises(NotImplementedError):
        augment.augment_data(engine, "elastic", batch)
