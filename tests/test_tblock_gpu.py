"""-m gpu: the fused transformer-block chains (csrc/tblock.hip, gn_tblock) against (a) an fp32 torch restatement of the diffusers
BasicTransformerBlock ops they replace -- attn1.to_out.0 + residual, norm2 -> attn2.to_q, attn2.to_out.0 + residual, norm3 -> GEGLU
feed-forward + residual, proj_out + residual (the transformer blocks inside `self.pipe(...)`, controller/agent/sd_controlnet_agent.py:67-76)
-- on the same f16-rounded inputs at the 1e-3 bar, stage by stage, and (b) the gn_gemm launches they replace (same f16 rounding points,
same K order: agreement far inside the bar).  Every 128-row workgroup, both column halves and all 20 hidden chunks carry distinct data."""
import pytest
import torch
import torch.nn.functional as F

from genima_amd import packing
from genima_amd._lib import ACT_GEGLU
from util import assert_close, q16, rel_l2

pytestmark = pytest.mark.gpu

C = 320


def _weights(seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    sd = {}
    b = "t.transformer_blocks.0"
    for n in ("attn1.to_out.0", "attn2.to_out.0"):
        sd[f"{b}.{n}.weight"], sd[f"{b}.{n}.bias"] = r(C, C, sc=C ** -0.5), r(C, sc=0.2)
    for n in ("attn1.to_q", "attn1.to_k", "attn1.to_v", "attn2.to_q"):
        sd[f"{b}.{n}.weight"] = r(C, C, sc=C ** -0.5)
    sd[f"{b}.attn2.to_k.weight"], sd[f"{b}.attn2.to_v.weight"] = r(C, 1024, sc=1024 ** -0.5), r(C, 1024, sc=1024 ** -0.5)
    for n in ("norm1", "norm2", "norm3"):
        sd[f"{b}.{n}.weight"], sd[f"{b}.{n}.bias"] = 1.0 + r(C, sc=0.2), r(C, sc=0.2)
    sd[f"{b}.ff.net.0.proj.weight"], sd[f"{b}.ff.net.0.proj.bias"] = r(8 * C, C, sc=C ** -0.5), r(8 * C, sc=0.2)
    sd[f"{b}.ff.net.2.weight"], sd[f"{b}.ff.net.2.bias"] = r(C, 4 * C, sc=(4 * C) ** -0.5), r(C, sc=0.2)
    sd["t.proj_out.weight"], sd["t.proj_out.bias"] = r(C, C, sc=C ** -0.5), r(C, sc=0.2)
    return {k: q16(v) for k, v in sd.items()}


def _ref_mid(sd, a, res):
    b = "t.transformer_blocks.0"
    h1 = q16(a @ sd[f"{b}.attn1.to_out.0.weight"].T + sd[f"{b}.attn1.to_out.0.bias"] + res)
    q = F.layer_norm(h1, (C,), sd[f"{b}.norm2.weight"], sd[f"{b}.norm2.bias"], 1e-5) @ sd[f"{b}.attn2.to_q.weight"].T
    return h1, q


def _ref_tail(sd, a, res, x):
    b = "t.transformer_blocks.0"
    h2 = q16(a @ sd[f"{b}.attn2.to_out.0.weight"].T + sd[f"{b}.attn2.to_out.0.bias"] + res)
    pr = F.layer_norm(h2, (C,), sd[f"{b}.norm3.weight"], sd[f"{b}.norm3.bias"], 1e-5) @ sd[f"{b}.ff.net.0.proj.weight"].T + sd[f"{b}.ff.net.0.proj.bias"]
    hid = q16(pr[:, : 4 * C] * F.gelu(pr[:, 4 * C:]))
    h3 = q16(hid @ sd[f"{b}.ff.net.2.weight"].T + sd[f"{b}.ff.net.2.bias"] + h2)
    return h3 @ sd["t.proj_out.weight"].T + sd["t.proj_out.bias"] + x


@pytest.fixture(scope="module")
def packed():
    sd = _weights()
    W = packing.pack_state_dict(sd, "cuda")
    assert "t.transformer_blocks.0.tblock_tail.tape" in W and "t.transformer_blocks.0.tblock_mid.tape" in W
    return sd, W


@pytest.mark.parametrize("M", [128, 640, 4096])
def test_tblock_mid_vs_reference_and_unfused(engine, packed, M):
    sd, W = packed
    b = "t.transformer_blocks.0"
    g = torch.Generator().manual_seed(M)
    a, res = q16(torch.randn(M, C, generator=g)), q16(torch.randn(M, C, generator=g) * 2.0 + 0.3)
    ad, rd = a.half().cuda(), res.half().cuda()
    h1, q = engine.tblock_mid(ad, rd, W[b + ".tblock_mid.tape"])
    rh1, rq = _ref_mid(sd, a, res)
    assert_close(h1, rh1, what="mid: attn1.to_out + residual")
    assert_close(q, rq, what="mid: norm2 -> attn2.to_q")
    # the launches it replaces
    uh1 = engine.linear(ad, W[b + ".attn1.to_out.0.weight"], W[b + ".attn1.to_out.0.bias"], residual=rd)
    uq = engine.linear(uh1, W[b + ".attn2.to_q.ln_weight"], W[b + ".attn2.to_q.ln_c2"], ln_c1=W[b + ".attn2.to_q.ln_c1"])
    assert torch.equal(h1, uh1), "h1 differs from the gn_gemm launch (same K order, same rounding point)"
    assert rel_l2(q, uq.float()) < 2e-4


@pytest.mark.parametrize("M", [128, 640, 4096])
def test_tblock_tail_vs_reference_and_unfused(engine, packed, M):
    sd, W = packed
    b = "t.transformer_blocks.0"
    g = torch.Generator().manual_seed(M + 1)
    a, res, x = q16(torch.randn(M, C, generator=g)), q16(torch.randn(M, C, generator=g) * 2.0 - 0.2), q16(torch.randn(M, C, generator=g) * 3.0)
    ad, rd, xd = a.half().cuda(), res.half().cuda(), x.half().cuda()
    out = engine.tblock_tail(ad, rd, xd, W[b + ".tblock_tail.tape"])
    assert_close(out, _ref_tail(sd, a, res, x), what="tail: attn2.to_out .. proj_out")
    h2 = engine.linear(ad, W[b + ".attn2.to_out.0.weight"], W[b + ".attn2.to_out.0.bias"], residual=rd)
    hid = engine.linear(h2, W[b + ".ff.net.0.proj.ln_weight"], W[b + ".ff.net.0.proj.ln_c2"], ln_c1=W[b + ".ff.net.0.proj.ln_c1"], act=ACT_GEGLU)
    h3 = engine.linear(hid, W[b + ".ff.net.2.weight"], W[b + ".ff.net.2.bias"], residual=h2)
    un = engine.linear(h3, W["t.proj_out.weight"], W["t.proj_out.bias"], residual=xd)
    assert rel_l2(out, un.float()) < 3e-4, rel_l2(out, un.float())


@pytest.mark.parametrize("B,N", [(1, 128), (3, 256), (2, 2048)])
def test_tblock_front_vs_reference_and_unfused(engine, packed, B, N):
    """GroupNorm (no activation) -> proj_in -> norm1 -> q | k | v (V transposed per sample) against torch and against the launches replaced."""
    sd, W = packed
    W = dict(W)
    g = torch.Generator().manual_seed(N)
    gamma, beta = q16(1.0 + 0.2 * torch.randn(C, generator=g)), q16(0.2 * torch.randn(C, generator=g))
    w_in, b_in = q16(torch.randn(C, C, generator=g) * C ** -0.5), q16(torch.randn(C, generator=g) * 0.2)
    full = dict(sd)
    full.update({"t.norm.weight": gamma, "t.norm.bias": beta, "t.proj_in.weight": w_in, "t.proj_in.bias": b_in})
    W = packing.pack_state_dict(full, "cuda")
    assert "t.tblock_front.tape" in W
    x = q16(torch.randn(B, N, C, generator=g) * 1.5 + 0.5)
    b = "t.transformer_blocks.0"
    xn = F.group_norm(x.transpose(1, 2), 32, gamma, beta, 1e-6).transpose(1, 2)
    rh = q16(q16(xn) @ w_in.T + b_in)
    ln = F.layer_norm(rh, (C,), sd[f"{b}.norm1.weight"], sd[f"{b}.norm1.bias"], 1e-5)
    rq, rk, rv = (ln @ sd[f"{b}.attn1.to_{n}.weight"].T for n in "qkv")
    xd = x.half().cuda()
    st = engine.groupnorm_stats(xd, W["t.norm.weight"], W["t.norm.bias"], 32, 1e-6)
    h, qk, vt = engine.tblock_front(xd, st, W["t.tblock_front.tape"], N)
    assert_close(h, rh, what="front: GroupNorm + proj_in")
    assert_close(qk[:, :, :C], rq, what="front: q")
    assert_close(qk[:, :, C:], rk, what="front: k")
    assert_close(vt[:, :, :N], rv.transpose(1, 2), what="front: V^T")
    if vt.shape[2] != N:
        assert float(vt[:, :, N:].abs().max()) == 0.0
    # the launches it replaces
    n0 = engine.groupnorm(xd, W["t.norm.weight"], W["t.norm.bias"], 32, 1e-6)
    h0 = engine.linear(n0, W["t.proj_in.weight"], W["t.proj_in.bias"])
    qk0, vt0 = engine.linear(h0, W[b + ".attn1.to_qkv.ln_weight"], W[b + ".attn1.to_qkv.ln_c2"], ln_c1=W[b + ".attn1.to_qkv.ln_c1"], split_n=2 * C,
                             rows_per_batch=N, pad_cols=vt.shape[2])
    assert rel_l2(h, h0.float()) < 2e-4 and rel_l2(qk, qk0.float()) < 3e-4 and rel_l2(vt, vt0.float()) < 3e-4


def test_tblock_rejects_what_it_is_not_built_for(engine, packed):
    from genima_amd._lib import GenimaHipError

    _, W = packed
    tape = W["t.transformer_blocks.0.tblock_mid.tape"]
    assert not engine.tblock_supported(100, C) and not engine.tblock_supported(256, 640) and engine.tblock_supported(256, C)
    with pytest.raises(GenimaHipError):
        engine.tblock_mid(torch.zeros(100, C, dtype=torch.float16, device="cuda"), torch.zeros(100, C, dtype=torch.float16, device="cuda"), tape)
    with pytest.raises(GenimaHipError):  # the tail's tape is not the mid's
        engine.tblock_mid(torch.zeros(128, C, dtype=torch.float16, device="cuda"), torch.zeros(128, C, dtype=torch.float16, device="cuda"),
                          W["t.transformer_blocks.0.tblock_tail.tape"])


def test_transformer_graph_with_and_without_the_fused_chains(engine, packed):
    """graphs.emit_transformer end to end (GroupNorm, proj_in, q | k | v, both attentions): the two-launch route against the six-launch one."""
    from genima_amd import graphs

    sd, W = packed
    W = dict(W)
    g = torch.Generator().manual_seed(5)
    W["t.norm.weight"], W["t.norm.bias"] = (1.0 + 0.1 * torch.randn(C, generator=g)).half().cuda(), (0.1 * torch.randn(C, generator=g)).half().cuda()
    W["t.proj_in.weight"], W["t.proj_in.bias"] = (torch.randn(C, C, generator=g) * C ** -0.5).half().cuda(), torch.zeros(C).half().cuda()
    B, Hh = 2, 16
    x = torch.randn(B, Hh, Hh, C, generator=g).half().cuda()
    ctx = torch.randn(B, 77, 1024, generator=g).half().cuda()
    kv = graphs.emit_cross_kv(engine, W, ctx, "t")
    kv = {k[len("t/"):] if k.startswith("t/") else k: v for k, v in kv.items()}
    old, old_min = engine.tblock, engine.tblock_min_rows
    try:
        engine.tblock, engine.tblock_min_rows = True, 0
        y1 = graphs.emit_transformer(engine, W, "t", x, kv, 5, 32).float()
        engine.tblock = False
        y0 = graphs.emit_transformer(engine, W, "t", x, kv, 5, 32).float()
    finally:
        engine.tblock, engine.tblock_min_rows = old, old_min
    assert rel_l2(y1, y0) < 5e-4, rel_l2(y1, y0)


def test_device_tape_packer_matches_the_torch_restatement(engine):
    """gn_pack_tblock_tape (the library's packer, what pack_state_dict uses on a ROCm device) against packing.pack_tblock_*_tape (torch ops; the
    layout tests/test_packing_cpu.py restates byte for byte): identical bytes for all three chains."""
    from genima_amd._lib import TBLOCK_FRONT, TBLOCK_MID, TBLOCK_TAIL

    g = torch.Generator().manual_seed(3)
    r16 = lambda *s: torch.randn(*s, generator=g).half()  # noqa: E731
    wo, bo, w1, w2, b2, wp, bp = r16(C, C), r16(C), r16(8 * C, C), r16(C, 4 * C), r16(C), r16(C, C), r16(C)
    c1, c2 = torch.randn(8 * C, generator=g), r16(8 * C)
    d = lambda t: t.cuda()  # noqa: E731
    tail = engine.pack_tblock_tape(TBLOCK_TAIL, d(wo), d(bo), d(w1), d(c1), d(c2), d(w2), d(b2), d(wp), d(bp))
    assert torch.equal(tail.cpu(), packing.pack_tblock_tail_tape(wo, bo, w1, c1, c2, w2, b2, wp, bp))
    mid = engine.pack_tblock_tape(TBLOCK_MID, d(wo), d(bo), d(wp), d(c1[:C]), d(c2[:C]))
    assert torch.equal(mid.cpu(), packing.pack_tblock_mid_tape(wo, bo, wp, c1[:C], c2[:C]))
    front = engine.pack_tblock_tape(TBLOCK_FRONT, d(wo), d(bo), d(w1[:3 * C]), d(c1[:3 * C].contiguous()), d(c2[:3 * C]))
    assert torch.equal(front.cpu(), packing.pack_tblock_front_tape(wo, bo, w1[:3 * C], c1[:3 * C], c2[:3 * C]))
