"""Weight repacking: diffusers/transformers-named fp32 state dicts -> the f16 device layouts libgenima_hip.so consumes.

  * Conv2d  OIHW  -> [Cout_pad8, KH*KW*Cin_pad8]  (tap-major, channel-minor: the implicit-GEMM K order); zero rows/cols in
    the padding so padded output channels are exact zeros and padded input channels are ignored.
  * Linear  [out, in] kept as is (K contiguous).
  * GEGLU   ``ff.net.0.proj`` [8C, C] -> alternating 32-row blocks [hidden | gate] (gn_gemm GN_ACT_GEGLU contract).
  * Self-attention ``to_q``/``to_k`` fused into one [2C, C] projection (``attn1.to_qk`` / ``self_attn.qk_proj``), and
    ``to_q``/``to_k``/``to_v`` into one [3C, C] projection (``attn1.to_qkv``: one two-destination launch, q | k row-major + V^T).
  * All ``time_emb_proj`` Linears of a network concatenated into one [sum Cout, temb] GEMM (``time_emb_proj_all``) whose
    output columns are sliced per ResNet block by pointer offset (gn_gemm ``ldshift``).
Explicit and caller-owned, as SURVEY.md section 8(b) asks: nothing is repacked behind the caller's back.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import torch


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pack_conv_weight(w: torch.Tensor, cin_splits=None, dtype=torch.float16) -> torch.Tensor:
    """OIHW fp32 -> [Cout_pad8, KH*KW*Cin_pad8] f16.  ``cin_splits=(C1, C2)`` keeps a virtual-concat boundary (each part
    must already be a multiple of 8)."""
    O, I, KH, KW = w.shape
    Ip, Op = _rup(I, 8), _rup(O, 8)
    p = torch.zeros((Op, KH, KW, Ip), dtype=torch.float32, device=w.device)
    p[:O, :, :, :I] = w.permute(0, 2, 3, 1)
    return p.reshape(Op, KH * KW * Ip).to(dtype).contiguous()


def pack_vec(b: torch.Tensor, n_pad: int, dtype=torch.float16) -> torch.Tensor:
    out = torch.zeros((n_pad,), dtype=torch.float32, device=b.device)
    out[: b.numel()] = b
    return out.to(dtype).contiguous()


def pack_geglu(w: torch.Tensor, b: torch.Tensor, dtype=torch.float16) -> Tuple[torch.Tensor, torch.Tensor]:
    n2 = w.shape[0]
    half = n2 // 2
    assert half % 32 == 0
    wh, wg = w[:half].reshape(half // 32, 32, -1), w[half:].reshape(half // 32, 32, -1)
    wp = torch.stack([wh, wg], dim=1).reshape(n2, -1)
    bh, bg = b[:half].reshape(half // 32, 32), b[half:].reshape(half // 32, 32)
    bp = torch.stack([bh, bg], dim=1).reshape(n2)
    return wp.to(dtype).contiguous(), bp.to(dtype).contiguous()


def _unpack_geglu(t: torch.Tensor) -> torch.Tensor:
    n2 = t.shape[0]
    blocks = t.reshape(n2 // 64, 2, 32, *t.shape[1:])
    return torch.cat([blocks[:, 0].reshape(n2 // 2, *t.shape[1:]), blocks[:, 1].reshape(n2 // 2, *t.shape[1:])], dim=0)


def unpack_state_dict(packed: Dict[str, torch.Tensor], schema: Dict[str, tuple], temb_slices=None) -> "OrderedDict[str, torch.Tensor]":
    """Inverse of ``pack_state_dict`` for the names of ``schema`` (diffusers layout, fp32): what ``save_pretrained`` of a network
    trained in the packed layout writes (diffusion/train_controlnet_genima.py:1486).  ``temb_slices``: the packer's
    ``__meta__['temb_slices']`` (needed when the per-ResNet ``time_emb_proj`` entries were dropped in favour of the fused one)."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in schema.items():
        shape = tuple(shape)
        if name not in packed:
            for q_or_k, fused, half in ((".attn1.to_q.weight", ".attn1.to_qk.weight", 0), (".attn1.to_k.weight", ".attn1.to_qk.weight", 1)):
                if name.endswith(q_or_k):
                    f = packed[name[: -len(q_or_k)] + fused]
                    n = f.shape[0] // 2
                    out[name] = f[half * n:(half + 1) * n].float().clone()
            if name.endswith(".time_emb_proj.weight") or name.endswith(".time_emb_proj.bias"):
                kind = name.rsplit(".", 1)[1]
                o, n = temb_slices[name[: -len(".time_emb_proj." + kind)]]
                out[name] = packed["time_emb_proj_all." + kind][o:o + n].float().clone()
            assert name in out, f"cannot reconstruct {name} from the packed tensors"
            continue
        t = packed[name].float()
        if len(shape) == 4:
            O, I, KH, KW = shape
            t = t.reshape(t.shape[0], KH, KW, -1)[:O, :, :, :I].permute(0, 3, 1, 2)
        elif len(shape) == 2:
            if name.endswith("ff.net.0.proj.weight"):
                t = _unpack_geglu(t)
            t = t[:, : shape[1]]
        elif len(shape) == 1:
            if name.endswith("ff.net.0.proj.bias"):
                t = _unpack_geglu(t)
            t = t[: shape[0]]
        out[name] = t.contiguous().clone()
        assert tuple(out[name].shape) == shape, (name, tuple(out[name].shape), shape)
    return out


def scale_vae_stream(sd: Dict[str, torch.Tensor], s: float) -> "OrderedDict[str, torch.Tensor]":
    """AutoencoderKL weights re-parametrised so that the RESIDUAL STREAM of the encoder and the decoder carries ``s`` times its values
    (s = 1/64: the stock SDXL VAE's stream exceeds f16's 65 504 -- diffusers' ``force_upcast`` runs it in fp32 for that reason).  Exact in
    real arithmetic: everything that WRITES the stream is scaled (conv_in, each ResnetBlock2D's conv2, the attention's to_out: weight and
    bias; convs that map the stream to the stream -- conv_shortcut, up / down samplers -- keep their weight and scale their bias), and
    everything that READS it is a GroupNorm, which is scale-invariant once its eps is multiplied by s^2 (graphs.emit_vae_* read
    ``__meta__["vae_stream_scale"]``).  Returns a new dict; tensors that do not change are shared."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in sd.items():
        head = k.startswith(("decoder.", "encoder."))
        both = head and (k.split(".", 1)[1].startswith("conv_in.") or ".conv2." in k or ".to_out.0." in k)
        bias_only = head and k.endswith(".bias") and (".conv_shortcut." in k or ".upsamplers." in k or ".downsamplers." in k)
        out[k] = v * s if (both or bias_only) else v
    return out


# (LayerNorm, consuming Linear) pairs whose LayerNorm is folded into the Linear's launch (gn_gemm_desc.ln_c1, csrc/gemm_common.h
# ln_fold_apply): diffusers BasicTransformerBlock norm1 -> attn1 q | k | v, norm2 -> attn2.to_q, norm3 -> the GEGLU projection;
# CLIPEncoderLayer layer_norm1 -> q | k | v, layer_norm2 -> fc1.
_LN_FOLDS = ((".norm1", ".attn1.to_qkv"), (".norm2", ".attn2.to_q"), (".norm3", ".ff.net.0.proj"),
             (".layer_norm1", ".self_attn.qkv_proj"), (".layer_norm2", ".mlp.fc1"))


def fold_layernorms(packed: Dict[str, torch.Tensor], hip=None) -> None:
    """For every pair of _LN_FOLDS present in a PACKED f16 dict add ``<linear>.ln_weight`` = W * gamma (f16, the packed row order of W:
    q | k | v concatenated, GEGLU interleaved), ``.ln_c1`` = row sums of that f16-rounded matrix (f32) and ``.ln_c2`` = W @ beta + b (f16).
    Linear(LayerNorm(x)) = rstd * (x @ ln_weight.T - mean * ln_c1) + ln_c2: the inference graphs then skip the LayerNorm launch."""
    for name in list(packed.keys()):
        for ln, lin in _LN_FOLDS:
            if not name.endswith(ln + ".weight"):
                continue
            base = name[: -len(ln + ".weight")]
            wn = base + lin + ".weight"
            if wn not in packed or packed[wn].dim() != 2:
                continue
            if packed[wn].shape[1] != packed[name].numel():
                continue
            bn = base + lin + ".bias"
            w16 = packed[wn]
            if hip is not None and w16.is_cuda and w16.dtype == torch.float16 and w16.is_contiguous():
                # on a ROCm device the fold runs as the library's own gn_pack_fold_layernorm (the entry point a non-Python host calls; no rocBLAS
                # gemv / Tensile kernel from torch in a process that only ever loads a checkpoint -- VERDICT r5 item 7)
                import ctypes as C
                from ._lib import check
                N, K = w16.shape
                g16, b16 = packed[name].to(w16.device, torch.float16).contiguous(), packed[base + ln + ".bias"].to(w16.device, torch.float16).contiguous()
                bias16 = packed[bn].to(w16.device, torch.float16).contiguous() if bn in packed else None
                wg = torch.empty_like(w16)
                c1, c2 = torch.empty(N, dtype=torch.float32, device=w16.device), torch.empty(N, dtype=torch.float16, device=w16.device)
                check(hip.lib.gn_pack_fold_layernorm(hip._ctx, C.c_void_p(w16.data_ptr()), C.c_void_p(g16.data_ptr()), C.c_void_p(b16.data_ptr()),
                                                     C.c_void_p(bias16.data_ptr()) if bias16 is not None else None, C.c_void_p(wg.data_ptr()),
                                                     C.c_void_p(c1.data_ptr()), C.c_void_p(c2.data_ptr()), N, K, w16.stride(0)), "gn_pack_fold_layernorm")
                hip._keepalive(g16, b16, bias16)
                packed[base + lin + ".ln_weight"], packed[base + lin + ".ln_c1"], packed[base + lin + ".ln_c2"] = wg, c1, c2
                continue
            w = packed[wn].float()
            gamma, beta = packed[name].float().to(w.device), packed[base + ln + ".bias"].float().to(w.device)
            wg = (w * gamma[None, :]).to(torch.float16)
            c2 = w @ beta
            if bn in packed:
                c2 = c2 + packed[bn].float().to(w.device)
            packed[base + lin + ".ln_weight"] = wg.contiguous()
            packed[base + lin + ".ln_c1"] = wg.float().sum(dim=1).contiguous()
            packed[base + lin + ".ln_c2"] = c2.to(torch.float16).contiguous()


# ---- weight tapes of the fused transformer-block chains (csrc/tblock.hip; include/genima_hip.h gn_tblock_desc) ------------------------------
TBLOCK_C, TBLOCK_SLOT, TBLOCK_VEC = 320, 20480, 3072


def _lds_image64(t: torch.Tensor) -> torch.Tensor:
    """[rows, 32] f16 -> the same bytes as the kernel's LDS sub-tile image: 64-byte rows whose four 16-byte chunks are XOR-swizzled by
    (row >> 2) & 3 (csrc/common.h lds_swz<64>): physical chunk pc of a row holds logical chunk pc ^ ((row >> 2) & 3)."""
    rows = t.shape[0]
    r = torch.arange(rows, device=t.device)
    idx = torch.arange(4, device=t.device)[None, :] ^ ((r >> 2) & 3)[:, None]
    return torch.gather(t.reshape(rows, 4, 8), 1, idx[:, :, None].expand(-1, -1, 8)).reshape(-1)


def _slot_bytes(*pieces: torch.Tensor) -> torch.Tensor:
    """pieces (any dtype, laid end to end) padded with zeros to one 20 KB slot, as bytes."""
    b = torch.cat([x.contiguous().reshape(-1).view(torch.uint8) for x in pieces])
    assert b.numel() <= TBLOCK_SLOT, b.numel()
    return torch.cat([b, torch.zeros(TBLOCK_SLOT - b.numel(), dtype=torch.uint8, device=b.device)])


def _nc_slots(w: torch.Tensor):
    """N = C Linear weight [320, 320] f16 -> ten slots, slot j = image of w[:, 32 j : 32 j + 32] (320 rows x 64 bytes)."""
    assert tuple(w.shape) == (TBLOCK_C, TBLOCK_C) and w.dtype == torch.float16, (w.shape, w.dtype)
    return [_slot_bytes(_lds_image64(w[:, 32 * j:32 * j + 32])) for j in range(TBLOCK_C // 32)]


def _vec_bytes(*pieces: torch.Tensor) -> torch.Tensor:
    b = torch.cat([x.contiguous().reshape(-1).view(torch.uint8) for x in pieces])
    assert b.numel() <= TBLOCK_VEC
    return torch.cat([b, torch.zeros(TBLOCK_VEC - b.numel(), dtype=torch.uint8, device=b.device)])


def pack_tblock_mid_tape(wo, bo, wq_ln, c1q, c2q) -> torch.Tensor:
    """Tape of GN_TBLOCK_MID: attn1.to_out.0 (weight, bias) then attn2.to_q with norm2 folded (``.ln_weight``, ``.ln_c1`` f32, ``.ln_c2``).
    Layout: 10 + 10 slots of 20 KB, then the 3 KB vector block  bo f16 [320] | c1 f32 [320] | c2 f16 [320]."""
    return torch.cat(_nc_slots(wo) + _nc_slots(wq_ln) + [_vec_bytes(bo.half(), c1q.float(), c2q.half())]).contiguous()


def pack_tblock_front_tape(w_in, b_in, wqkv_ln, c1, c2) -> torch.Tensor:
    """Tape of GN_TBLOCK_FRONT: proj_in (weight, bias), then attn1's q | k | v projection with norm1 folded (``attn1.to_qkv.ln_weight`` [960, 320],
    ``.ln_c1`` f32 [960], ``.ln_c2`` [960]).  Layout: 10 slots (proj_in) + 3 x 10 slots (q, k, v), then a 7 KB vector block
    b_in f16 [320] | c1 f32 [960] | c2 f16 [960]."""
    C = TBLOCK_C
    assert tuple(wqkv_ln.shape) == (3 * C, C), wqkv_ln.shape
    parts = _nc_slots(w_in)
    for g in range(3):
        parts += _nc_slots(wqkv_ln[g * C:(g + 1) * C].contiguous())
    vec = torch.cat([x.contiguous().reshape(-1).view(torch.uint8) for x in (b_in.half(), c1.float(), c2.half())])
    parts.append(torch.cat([vec, torch.zeros(7168 - vec.numel(), dtype=torch.uint8, device=vec.device)]))
    return torch.cat(parts).contiguous()


def pack_tblock_tail_tape(wo, bo, w1_ln, c1, c2, w2, b2, wp, bp) -> torch.Tensor:
    """Tape of GN_TBLOCK_TAIL: attn2.to_out.0; ff.net.0.proj with norm3 folded, in the packed GEGLU row order (32-row [hidden | gate]
    blocks) + its c1 (f32) / c2; ff.net.2; proj_out.  Layout: 10 slots (to_out); per 64-column chunk ch of the hidden dimension 5 slots of the
    projection -- slot js = images of w1_ln[128 ch : 128 ch + 128, 64 js + 32 sub : + 32], sub = 0, 1, the LAST one followed at byte 16384 by
    c1[128 ch : + 128] f32 and c2[128 ch : + 128] f16 -- and 2 slots of ff.net.2 (w2[:, 64 ch + 32 j : + 32]); 10 slots (proj_out); then the
    vector block  bo f16 [320] | b2 f16 [320] | bp f16 [320]."""
    C = TBLOCK_C
    assert tuple(w1_ln.shape) == (8 * C, C) and tuple(w2.shape) == (C, 4 * C), (w1_ln.shape, w2.shape)
    parts = _nc_slots(wo)
    for ch in range(4 * C // 64):
        rows = w1_ln[128 * ch:128 * ch + 128]
        for js in range(5):
            pieces = [_lds_image64(rows[:, 64 * js + 32 * sub:64 * js + 32 * sub + 32]) for sub in range(2)]
            if js == 4:
                pieces += [c1[128 * ch:128 * ch + 128].float(), c2[128 * ch:128 * ch + 128].half()]
            parts.append(_slot_bytes(*pieces))
        for j in range(2):
            parts.append(_slot_bytes(_lds_image64(w2[:, 64 * ch + 32 * j:64 * ch + 32 * j + 32])))
    parts += _nc_slots(wp)
    parts.append(_vec_bytes(bo.half(), b2.half(), bp.half()))
    return torch.cat(parts).contiguous()


def add_tblock_tapes(packed: Dict[str, torch.Tensor], hip=None) -> None:
    """For every single-block Transformer2DModel of width 320 in a PACKED f16 dict (after fold_layernorms) add ``<block>.tblock_mid.tape`` and
    ``<block>.tblock_tail.tape`` (uint8): graphs.emit_transformer then runs attn1.to_out .. attn2.to_q and attn2.to_out .. proj_out as two
    gn_tblock launches instead of six gn_gemm launches."""
    for name in list(packed.keys()):
        if not name.endswith(".transformer_blocks.0.ff.net.0.proj.ln_weight"):
            continue
        b = name[: -len(".ff.net.0.proj.ln_weight")]
        p = b[: -len(".transformer_blocks.0")]
        if p + ".transformer_blocks.1.norm1.weight" in packed or packed[name].shape[1] != TBLOCK_C:
            continue
        need = [b + ".attn1.to_out.0.weight", b + ".attn1.to_out.0.bias", b + ".attn2.to_q.ln_weight", b + ".attn2.to_out.0.weight",
                b + ".attn2.to_out.0.bias", b + ".ff.net.2.weight", b + ".ff.net.2.bias", p + ".proj_out.weight", p + ".proj_out.bias"]
        if any(k not in packed for k in need) or packed[p + ".proj_out.weight"].dim() != 2:
            continue
        dev = packed[name].device  # (a dict packed for a ROCm device holds its GEGLU tensors there already, the rest still on the host)
        g = lambda k: packed[k].to(dev)  # noqa: E731
        front = (b + ".attn1.to_qkv.ln_weight") in packed and (p + ".proj_in.weight") in packed and packed[p + ".proj_in.weight"].dim() == 2
        if hip is not None and dev.type == "cuda":
            # the library's own packer (gn_pack_tblock_tape: the entry point a non-Python host calls); bit-identical to the torch restatements above
            from ._lib import TBLOCK_FRONT, TBLOCK_MID, TBLOCK_TAIL
            if front:
                packed[p + ".tblock_front.tape"] = hip.pack_tblock_tape(TBLOCK_FRONT, g(p + ".proj_in.weight"), g(p + ".proj_in.bias"), g(b + ".attn1.to_qkv.ln_weight"),
                                                                        g(b + ".attn1.to_qkv.ln_c1"), g(b + ".attn1.to_qkv.ln_c2"))
            packed[b + ".tblock_mid.tape"] = hip.pack_tblock_tape(TBLOCK_MID, g(b + ".attn1.to_out.0.weight"), g(b + ".attn1.to_out.0.bias"), g(b + ".attn2.to_q.ln_weight"),
                                                                  g(b + ".attn2.to_q.ln_c1"), g(b + ".attn2.to_q.ln_c2"))
            packed[b + ".tblock_tail.tape"] = hip.pack_tblock_tape(TBLOCK_TAIL, g(b + ".attn2.to_out.0.weight"), g(b + ".attn2.to_out.0.bias"), g(name),
                                                                   g(b + ".ff.net.0.proj.ln_c1"), g(b + ".ff.net.0.proj.ln_c2"), g(b + ".ff.net.2.weight"),
                                                                   g(b + ".ff.net.2.bias"), g(p + ".proj_out.weight"), g(p + ".proj_out.bias"))
            hip.synchronize()
            continue
        if front:
            packed[p + ".tblock_front.tape"] = pack_tblock_front_tape(g(p + ".proj_in.weight"), g(p + ".proj_in.bias"), g(b + ".attn1.to_qkv.ln_weight"),
                                                                     g(b + ".attn1.to_qkv.ln_c1"), g(b + ".attn1.to_qkv.ln_c2"))
        packed[b + ".tblock_mid.tape"] = pack_tblock_mid_tape(g(b + ".attn1.to_out.0.weight"), g(b + ".attn1.to_out.0.bias"), g(b + ".attn2.to_q.ln_weight"),
                                                              g(b + ".attn2.to_q.ln_c1"), g(b + ".attn2.to_q.ln_c2"))
        packed[b + ".tblock_tail.tape"] = pack_tblock_tail_tape(g(b + ".attn2.to_out.0.weight"), g(b + ".attn2.to_out.0.bias"), g(name),
                                                                g(b + ".ff.net.0.proj.ln_c1"), g(b + ".ff.net.0.proj.ln_c2"), g(b + ".ff.net.2.weight"),
                                                                g(b + ".ff.net.2.bias"), g(p + ".proj_out.weight"), g(p + ".proj_out.bias"))


_UP_TAPS = (((0,), (1, 2)), ((0, 1), (2,)))  # [phase d][2x2 tap a] -> the 3x3 taps that land on source row / column (y - 1 + d + a)


def pack_upsample_phases(w: torch.Tensor, dtype=torch.float16) -> torch.Tensor:
    """3x3 conv that follows a nearest-2x upsample (diffusers Upsample2D) -> its four PHASE convs on the source pixels.
    Output pixel (2y + dy, 2x + dx) of conv3x3(upsample2x(src)) only sees the 2x2 source neighbourhood rows {y - 1 + dy, y + dy}, columns
    {x - 1 + dx, x + dx}: the three taps of a row (column) collapse onto two source rows (columns), so the phase's 2x2 weights are sums
    of the 3x3 ones -- 4 / 9 of the multiply-adds, exact in real arithmetic (the sums are taken in fp32 and rounded once).
    w: OIHW [Cout, Cin, 3, 3] -> [4 (phase = 2 dy + dx), round_up(Cout, 8), 4 * round_up(Cin, 8)] in the packed tap-major K order."""
    O, I, KH, KW = w.shape
    assert KH == 3 and KW == 3, w.shape
    acc_t = torch.float64 if w.dtype == torch.float64 else torch.float32
    w = w.detach().to(acc_t)
    Op, Ip = _rup(O, 8), _rup(I, 8)
    out = torch.zeros((4, Op, 4 * Ip), dtype=acc_t, device=w.device)
    for dy in range(2):
        for dx in range(2):
            for a in range(2):
                for b in range(2):
                    acc = torch.zeros((O, I), dtype=acc_t, device=w.device)
                    for ky in _UP_TAPS[dy][a]:
                        for kx in _UP_TAPS[dx][b]:
                            acc = acc + w[:, :, ky, kx]
                    t = (a * 2 + b) * Ip
                    out[2 * dy + dx, :O, t:t + I] = acc
    return out.to(dtype).contiguous()


def pack_state_dict(sd: Dict[str, torch.Tensor], device, dtype=torch.float16, up_phases: bool = True) -> "OrderedDict[str, torch.Tensor]":
    """Generic packer for UNet / ControlNet / VAE / CLIP-text state dicts (see module docstring for the derived entries).
    ``dtype=torch.float32`` gives the same layout for the fp32 master copy of a trainable network (training.TrainParams).
    ``up_phases=False`` skips the four-phase copies of the upsampler convs (``*.up4.weight``, 16 / 9 of the 3x3 weight's bytes): dicts
    handed to the trainers, whose tape only runs the fused-upsample 3x3 launch, do not need them.  The phase path sums taps in fp32 and
    rounds once, so it differs from the 3x3 launch by f16 rounding of the summed weights (6e-4 rel-L2, tests/test_upsample_phases_gpu.py)."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    temb_w, temb_b, temb_slices, off = [], [], OrderedDict(), 0
    # f16 copies for a ROCm device: the two layout shuffles run as the library's own gn_pack_* kernels on the uploaded tensors (the same
    # entry points a non-Python host calls; bit-identical to the torch restatements above, tests/test_kernels_gpu.py)
    hip = None
    if dtype == torch.float16 and torch.device(device).type == "cuda" and torch.cuda.is_available():
        from .engine import Engine

        hip = Engine(torch.device(device))
    for name, t in sd.items():
        t = t.detach().to(torch.float32)
        if t.dim() == 4:
            out[name] = hip.pack_conv_weight(t.to(hip.device)) if hip is not None else pack_conv_weight(t, dtype=dtype)
            if up_phases and dtype == torch.float16 and ".upsamplers." in name and tuple(t.shape[2:]) == (3, 3):
                # inference graphs run the nearest-2x + 3x3 conv as four 2x2 phase convs on the source pixels (4 / 9 of the work)
                out[name[: -len("weight")] + "up4.weight"] = pack_upsample_phases(t, dtype=dtype)
            bn = name[: -len("weight")] + "bias"
            if bn in sd:
                out[bn] = pack_vec(sd[bn].detach().float(), out[name].shape[0], dtype=dtype)
        elif t.dim() == 2:
            if name.endswith("ff.net.0.proj.weight"):
                bn = name[: -len("weight")] + "bias"
                if hip is not None:
                    wp, bp = hip.pack_geglu(t.to(hip.device), sd[bn].detach().float().to(hip.device))
                else:
                    wp, bp = pack_geglu(t, sd[bn].detach().float(), dtype=dtype)
                out[name], out[bn] = wp, bp
                continue
            if name.endswith("time_emb_proj.weight"):
                bn = name[: -len("weight")] + "bias"
                temb_w.append(t)
                temb_b.append(sd[bn].detach().float())
                temb_slices[name[: -len(".time_emb_proj.weight")]] = (off, t.shape[0])
                off += t.shape[0]
            if t.shape[1] % 8 != 0:  # pad the reduction dim
                tp = torch.zeros((t.shape[0], _rup(t.shape[1], 8)), device=t.device)
                tp[:, : t.shape[1]] = t
                t = tp
            out[name] = t.to(dtype).contiguous()
        elif t.dim() == 1:
            if name not in out:  # conv / geglu biases were handled with their weights
                out[name] = t.to(dtype).contiguous()
        else:
            out[name] = t.to(dtype).contiguous()
    # fused self-attention q|k projections
    for name in list(sd.keys()):
        for qn, kn, fused in ((".attn1.to_q.weight", ".attn1.to_k.weight", ".attn1.to_qk.weight"),
                              (".self_attn.q_proj.weight", ".self_attn.k_proj.weight", ".self_attn.qk_proj.weight")):
            if name.endswith(qn):
                base = name[: -len(qn)]
                out[base + fused] = torch.cat([sd[name], sd[base + kn]], dim=0).to(dtype).contiguous()
                qb, kb = name[: -len("weight")] + "bias", (base + kn)[: -len("weight")] + "bias"
                if qb in sd:
                    out[(base + fused)[: -len("weight")] + "bias"] = torch.cat([sd[qb], sd[kb]]).to(dtype).contiguous()
                if qn == ".self_attn.q_proj.weight" and base + ".self_attn.v_proj.weight" in sd:
                    # CLIP towers: q | k | v (+ their biases) as one two-destination launch too
                    vw = base + ".self_attn.v_proj.weight"
                    out[base + ".self_attn.qkv_proj.weight"] = torch.cat([sd[name], sd[base + kn], sd[vw]], dim=0).to(dtype).contiguous()
                    if qb in sd:
                        out[base + ".self_attn.qkv_proj.bias"] = torch.cat([sd[qb], sd[kb], sd[vw[: -len("weight")] + "bias"]]).to(dtype).contiguous()
                if qn == ".attn1.to_q.weight" and base + ".attn1.to_v.weight" in sd and qb not in sd:
                    # q | k | v as ONE launch (two-destination GEMM: q, k row-major + V^T): the inference graphs' self-attention
                    out[base + ".attn1.to_qkv.weight"] = torch.cat([sd[name], sd[base + kn], sd[base + ".attn1.to_v.weight"]], dim=0).to(dtype).contiguous()
    if dtype == torch.float16 and up_phases:
        # ResnetBlock2D with a conv_shortcut: conv2(h) + conv_shortcut(x) as ONE conv whose K axis carries the 1x1 weight behind the 3x3 one
        # (gn_gemm_desc.k_append) -- [Cout, 9 * C + Cx] and the two biases' sum (fp32 sum, one rounding).  Inference dicts only (as up4).
        for name in list(sd.keys()):
            if not name.endswith(".conv_shortcut.weight"):
                continue
            base = name[: -len(".conv_shortcut.weight")]
            w2, ws = out.get(base + ".conv2.weight"), out.get(name)
            if w2 is None or ws is None or sd[base + ".conv2.weight"].shape[2:] != (3, 3) or (w2.shape[1] // 9) % 64 != 0 or ws.shape[0] != w2.shape[0]:
                continue
            out[base + ".conv2sc.weight"] = torch.cat([w2, ws.to(w2.device)], dim=1).contiguous()
            b = sd[base + ".conv2.bias"].detach().float() + sd[base + ".conv_shortcut.bias"].detach().float()
            out[base + ".conv2sc.bias"] = pack_vec(b, w2.shape[0], dtype=dtype)
    if dtype == torch.float16 and up_phases:
        # Transformer2DModel with a Linear proj_out: behind its last BasicTransformerBlock, proj_out(ff.net.2(g) + h) + x has no nonlinearity between its
        # two Linears, so it is ONE GEMM over [g | h] (gn_gemm_desc.k_append, dense) with the weight [W_po W_ff2 | W_po] and the bias
        # W_po b_ff2 + b_po -- the products in fp32, one rounding to f16.  Inference dicts only.
        for name in list(sd.keys()):
            if not name.endswith(".proj_out.weight") or sd[name].dim() != 2:
                continue
            base = name[: -len(".proj_out.weight")]
            last = 0  # the LAST block's ff.net.2 is the Linear in front of proj_out (SD-2.x: one block per transformer; SDXL: 2 or 10)
            while (base + f".transformer_blocks.{last + 1}.norm1.weight") in sd:
                last += 1
            f2 = base + f".transformer_blocks.{last}.ff.net.2"
            if f2 + ".weight" not in sd or sd[f2 + ".weight"].shape[1] % 64 != 0:
                continue
            # (composed on the HOST in f64, once per checkpoint load: a 320 .. 1280-wide product -- and no Tensile f64 GEMM in a process that runs the hot path)
            wpo, wf2 = sd[name].detach().double().cpu(), sd[f2 + ".weight"].detach().double().cpu()
            dev = out[name].device
            out[base + ".ffo_pout.weight"] = torch.cat([wpo @ wf2, wpo], dim=1).to(dtype).contiguous().to(dev)
            b = wpo @ sd[f2 + ".bias"].detach().double().cpu() + sd[base + ".proj_out.bias"].detach().double().cpu()
            out[base + ".ffo_pout.bias"] = b.to(dtype).contiguous().to(dev)
    if dtype == torch.float16:
        fold_layernorms(out, hip)
        add_tblock_tapes(out, hip)
    meta = {}
    # every cross-attention layer's to_k | to_v stacked along N: the prompt's K / V projections of a whole network as ONE launch (graphs.emit_cross_kv;
    # 14 + 32 launches of M = 77 B rows otherwise).  Inference dicts; sites in dict order, (offset, C) per site in the meta.
    if dtype == torch.float16:
        sites = [n[: -len(".to_k.weight")] for n in out if n.endswith(".attn2.to_k.weight") and n[: -len("to_k.weight")] + "to_v.weight" in out]
        if sites and len({out[q + ".to_k.weight"].shape[1] for q in sites}) == 1:
            offs, o = OrderedDict(), 0
            for q in sites:
                c = out[q + ".to_k.weight"].shape[0]
                offs[q] = (o, c)
                o += 2 * c
            out["cross_kv_all.weight"] = torch.cat([t.to(out[sites[0] + ".to_k.weight"].device) for q in sites for t in (out[q + ".to_k.weight"], out[q + ".to_v.weight"])], dim=0).contiguous()
            meta["cross_kv"] = offs
    if temb_w:
        out["time_emb_proj_all.weight"] = torch.cat(temb_w, dim=0).to(dtype).contiguous()
        out["time_emb_proj_all.bias"] = torch.cat(temb_b, dim=0).to(dtype).contiguous()
        meta["temb_slices"] = temb_slices
        meta["temb_total"] = off
    dev = OrderedDict((k, v.to(device)) for k, v in out.items())
    if hip is not None:
        hip.synchronize()  # the packing kernels ran on this engine's stream: finished before any other engine reads the weights
    dev["__meta__"] = meta  # type: ignore[assignment]
    return dev
