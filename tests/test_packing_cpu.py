"""-m "not gpu": the packed kernel layout is an invertible re-arrangement of the diffusers state dict (what the trainer's checkpoints rely on)."""
import torch

from genima_amd import configs, schema, weights
from genima_amd.packing import pack_state_dict, unpack_state_dict


def _roundtrip(sch, seed):
    sd = weights.synth_state_dict(sch, seed)
    packed = pack_state_dict(sd, "cpu", dtype=torch.float32)
    meta = packed.pop("__meta__")
    # the trainer drops the entries the fused tensors cover; the inverse must not need them
    for k in [k for k in packed if k.endswith((".attn1.to_q.weight", ".attn1.to_k.weight", ".time_emb_proj.weight", ".time_emb_proj.bias"))]:
        del packed[k]
    back = unpack_state_dict(packed, sch, meta.get("temb_slices"))
    assert list(back) == list(sch)
    for k in sch:
        assert torch.equal(back[k], sd[k]), k


def test_controlnet_and_unet_pack_unpack_roundtrip():
    fam = configs.family("tiny")
    _roundtrip(schema.controlnet_schema(fam["controlnet"]), 2)
    _roundtrip(schema.unet_schema(fam["unet"]), 1)


def test_layernorm_fold_entries_reproduce_layernorm_then_linear():
    """packing.fold_layernorms: rstd * (x @ ln_weight.T - mean * ln_c1) + ln_c2 == Linear(LayerNorm(x)) for every folded pair of a packed
    UNet and CLIP tower (the identity the HIP GEMM's gn_gemm_desc.ln_c1 mode evaluates), GEGLU row interleave and q | k | v concat included."""
    import torch.nn.functional as F

    fam = configs.family("tiny")
    for sch, seed in ((schema.unet_schema(fam["unet"]), 1), (schema.clip_text_schema(fam["text"]), 3)):
        sd = weights.synth_state_dict(sch, seed)
        for k in sd:  # non-trivial affine parameters
            if k.endswith(("norm1.weight", "norm2.weight", "norm3.weight")):
                sd[k] = 1.0 + 0.2 * torch.randn_like(sd[k])
            if k.endswith(("norm1.bias", "norm2.bias", "norm3.bias")):
                sd[k] = 0.1 * torch.randn_like(sd[k])
        packed = pack_state_dict(sd, "cpu")
        folded = [k[: -len(".ln_weight")] for k in packed if k.endswith(".ln_weight")]
        assert folded, "no LayerNorm was folded"
        for lin in folded:
            wg, c1, c2 = packed[lin + ".ln_weight"].float(), packed[lin + ".ln_c1"], packed[lin + ".ln_c2"].float()
            w = packed[lin + ".weight"].float()
            b = packed[lin + ".bias"].float() if lin + ".bias" in packed else 0.0
            ln = {".attn1.to_qkv": ".norm1", ".attn2.to_q": ".norm2", ".ff.net.0.proj": ".norm3", ".self_attn.qkv_proj": ".layer_norm1", ".mlp.fc1": ".layer_norm2"}
            suffix = next(s for s in ln if lin.endswith(s))
            pre = lin[: -len(suffix)] + ln[suffix]
            gamma, beta = packed[pre + ".weight"].float(), packed[pre + ".bias"].float()
            K = w.shape[1]
            x = torch.randn(37, K) * 0.8 + torch.randn(37, 1) * 2.0
            want = F.layer_norm(x, (K,), gamma, beta, 1e-5) @ w.t() + b
            mean, var = x.mean(1, keepdim=True), x.var(1, unbiased=False, keepdim=True)
            got = torch.rsqrt(var + 1e-5) * (x @ wg.t() - mean * c1[None, :]) + c2[None, :]
            assert c1.dtype == torch.float32
            err = float((got - want).abs().max()) / float(want.abs().max())
            assert err < 2e-3, (lin, err)  # f16 rounding of W * gamma and of c2


def test_upsample_phase_weights_are_the_collapsed_taps():
    import torch.nn.functional as F

    from genima_amd.packing import pack_upsample_phases

    """CPU-side identity in fp64: the four 2x2 phase convs on the source equal the 3x3 conv on the upsampled image, borders included."""
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(2, 5, 6, 8, generator=g).double(), torch.randn(16, 8, 3, 3, generator=g).double()
    ref = F.conv2d(F.interpolate(x.permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest"), w, padding=1)
    w4 = pack_upsample_phases(w, dtype=torch.float64)  # [4, 16, 4 * 8], K order (a, b, c)
    out = torch.zeros_like(ref)
    for dy in range(2):
        for dx in range(2):
            wk = w4[2 * dy + dx].view(16, 2, 2, 8).permute(0, 3, 1, 2)
            xp = F.pad(x.permute(0, 3, 1, 2), (1 - dx, dx, 1 - dy, dy))
            out[:, :, dy::2, dx::2] = F.conv2d(xp, wk)
    assert float((out - ref).abs().max()) < 1e-12




def test_tblock_tape_layout_byte_for_byte():
    """The weight tapes of the fused transformer-block chains (csrc/tblock.hip): every slot is the LDS image the kernel copies verbatim, so
    the layout is restated here byte by byte from the kernel's own addressing -- row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4) inside a
    [rows x 32] sub-tile (csrc/common.h lds_swz<64>), 20 KB slots in consumption order, the GEGLU chunk's c1 / c2 at byte 16384 of its last
    projection slot, the vector block behind the last slot."""
    import numpy as np

    from genima_amd import packing as P

    C = 320
    g = torch.Generator().manual_seed(0)
    r16 = lambda *s: torch.randn(*s, generator=g).half()  # noqa: E731
    wo, bo, w1, w2, b2, wp, bp = r16(C, C), r16(C), r16(8 * C, C), r16(C, 4 * C), r16(C), r16(C, C), r16(C)
    c1, c2 = torch.randn(8 * C, generator=g), r16(8 * C)
    tape = P.pack_tblock_tail_tape(wo, bo, w1, c1, c2, w2, b2, wp, bp).numpy()
    assert tape.dtype == np.uint8 and tape.size == 160 * 20480 + 3072

    def elem(slot, sub_off, row, k):  # the f16 the kernel's MFMA fragment read finds for (row, k) of a sub-tile at byte sub_off of a slot
        off = slot * 20480 + sub_off + row * 64 + (((k >> 3) ^ ((row >> 2) & 3)) << 4) + (k & 7) * 2
        return tape[off:off + 2].view(np.float16)[0]

    rng = np.random.default_rng(0)
    for _ in range(200):
        n, k = int(rng.integers(C)), int(rng.integers(C))
        assert elem(k // 32, 0, n, k % 32) == wo[n, k].numpy()                       # attn2.to_out: slots 0..9
        assert elem(150 + k // 32, 0, n, k % 32) == wp[n, k].numpy()                 # proj_out: slots 150..159
        ch, r = int(rng.integers(20)), int(rng.integers(128))
        assert elem(10 + 7 * ch + k // 64, 8192 * ((k % 64) // 32), r, k % 32) == w1[128 * ch + r, k].numpy()   # GEGLU projection chunk
        kk = int(rng.integers(64))
        assert elem(10 + 7 * ch + 5 + kk // 32, 0, n, kk % 32) == w2[n, 64 * ch + kk].numpy()                   # ff.net.2 chunk
        base = (10 + 7 * ch + 4) * 20480 + 16384
        assert tape[base + 4 * r:base + 4 * r + 4].view(np.float32)[0] == c1[128 * ch + r].numpy()
        assert tape[base + 512 + 2 * r:base + 512 + 2 * r + 2].view(np.float16)[0] == c2[128 * ch + r].numpy()
    vec = tape[160 * 20480:]
    assert (vec[:640].view(np.float16) == bo.numpy()).all() and (vec[640:1280].view(np.float16) == b2.numpy()).all()
    assert (vec[1280:1920].view(np.float16) == bp.numpy()).all() and not vec[1920:].any()
    mid = P.pack_tblock_mid_tape(wo, bo, wp, c1[:C], c2[:C]).numpy()
    assert mid.size == 20 * 20480 + 3072 and (mid[20 * 20480 + 640:20 * 20480 + 1920].view(np.float32) == c1[:C].numpy()).all()
    front = P.pack_tblock_front_tape(wo, bo, w1[:3 * C], c1[:3 * C], c2[:3 * C]).numpy()
    assert front.size == 40 * 20480 + 7168
    assert (front[40 * 20480 + 640:40 * 20480 + 640 + 3840].view(np.float32) == c1[:3 * C].numpy()).all()
