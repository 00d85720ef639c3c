"""fp8 (OCP e4m3) quantisation + Linear on the fp8 MFMA vs the CPU oracle (oracle/fp8_torch.py), through the C ABI."""
import pytest
import torch
import torch.nn.functional as F

from oracle import fp8_torch
from tests.util import assert_close, randn_h

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from genima_amd.engine import Engine
    return Engine("cuda:0")


@pytest.mark.parametrize("rows,K", [(64, 320), (1000, 1280), (5, 24), (4096, 2048), (77, 10240)])
def test_quantize_fp8_rows_bit_exact(engine, rows, K):
    x = randn_h(rows, K, seed=rows + K, scale=2.0)
    x[rows // 2] = 0  # an all-zero row: scale 1, bytes 0
    q, s = engine.quantize_fp8(x)
    rq, rs = fp8_torch.quantize_rows(x.cpu())
    assert torch.equal(s[:rows].cpu(), rs), "row scales differ"
    got = q.cpu()[:, :K]
    assert torch.equal(got, rq.view(torch.uint8)), f"{int((got != rq.view(torch.uint8)).sum())} quantised bytes differ"
    assert int(q.cpu()[:, K:].abs().sum()) == 0, "pad bytes must be zero"


@pytest.mark.parametrize("M,N,K,act,res", [(256, 320, 320, None, False), (4096, 640, 2560, None, True), (300, 1280, 1280, "silu", False),
                                           (77, 2048, 1280, None, False), (8192, 1280, 5120, None, True), (33, 64, 48, "gelu", False)])
def test_linear_fp8_vs_oracle(engine, M, N, K, act, res):
    from genima_amd.engine import ACT_GELU, ACT_NONE, ACT_SILU
    x = randn_h(M, K, seed=1, scale=1.5)
    w = randn_h(N, K, seed=2, scale=K ** -0.5)
    b = randn_h(N, seed=3, scale=0.1)
    r = randn_h(M, N, seed=4) if res else None
    xq, xs = engine.quantize_fp8(x)
    wq, ws = engine.quantize_fp8(w)
    code = {None: ACT_NONE, "silu": ACT_SILU, "gelu": ACT_GELU}[act]
    y = engine.linear_fp8(xq, xs, wq, ws, b, act=code, residual=r)
    fn = {None: None, "silu": F.silu, "gelu": F.gelu}[act]
    ref = fp8_torch.linear_fp8(x.cpu(), w.cpu(), b.cpu(), fn, r.cpu() if res else None)
    assert_close(y, ref, what=f"fp8 linear {M}x{N}x{K} act={act} res={res}")


def test_linear_fp8_tracks_the_f16_linear(engine):
    """The fp8 result is a quantised estimate of the f16 Linear: ~3-4 % relative error per product, averaging down over K."""
    M, N, K = 2048, 1280, 1280
    x = randn_h(M, K, seed=5)
    w = randn_h(N, K, seed=6, scale=K ** -0.5)
    xq, xs = engine.quantize_fp8(x)
    wq, ws = engine.quantize_fp8(w)
    y8 = engine.linear_fp8(xq, xs, wq, ws).float()
    y16 = engine.linear(x, w).float()
    err = float((y8 - y16).norm() / y16.norm())
    assert err < 0.06, err


def test_linear_fp8_geglu(engine):
    from genima_amd.engine import ACT_GEGLU
    from genima_amd.packing import pack_geglu
    M, C = 512, 640
    x = randn_h(M, C, seed=7)
    w = randn_h(8 * C, C, seed=8, scale=C ** -0.5)
    b = randn_h(8 * C, seed=9, scale=0.1)
    wp, bp = pack_geglu(w.cpu().float(), b.cpu().float())
    wp, bp = wp.cuda(), bp.cuda()
    xq, xs = engine.quantize_fp8(x)
    wq, ws = engine.quantize_fp8(wp)
    y = engine.linear_fp8(xq, xs, wq, ws, bp, act=ACT_GEGLU)
    lin = fp8_torch.linear_fp8(x.cpu(), w.cpu(), b.cpu())  # per-row weight scales: row order does not matter
    hid, gate = lin.chunk(2, dim=-1)
    assert_close(y, hid * F.gelu(gate), what="fp8 GEGLU")


def test_engine_fp8_dispatch_of_registered_weights(engine):
    """enable_fp8: a later linear(x, w) on a registered weight runs the fp8 kernel; unregistered weights stay f16."""
    from genima_amd.engine import Engine
    E = Engine("cuda:0")
    x = randn_h(2048, 640, seed=10)
    w1 = randn_h(640, 640, seed=11, scale=0.04)
    w2 = randn_h(640, 640, seed=12, scale=0.04)
    E.enable_fp8([w1])
    y1, y2 = E.linear(x, w1), E.linear(x, w2)
    assert_close(y1, fp8_torch.linear_fp8(x.cpu(), w1.cpu()), what="dispatched fp8 linear")
    assert_close(y2, x.cpu().float() @ w2.cpu().float().t(), what="f16 linear beside it")
