"""Lowering of the Genima networks to libgenima_hip.so ops (through an ``Engine``), NHWC throughout.

Each ``emit_*`` function walks one architecture exactly as the reference's third-party modules execute it
(oracle/sd_torch.py is the CPU restatement with the same structure) and enqueues / records the
fused kernels:
  ResnetBlock2D      = GN+SiLU | conv3x3 (+bias +time shift) | GN+SiLU | conv3x3 (+bias +shortcut residual)   [4-5 launches]
  Transformer2DModel = GN | proj_in | LN | qk-proj | v-proj(transposed) | flash-attn | out-proj(+res) | LN | q-proj |
                       flash-attn(ctx K/V hoisted) | out-proj(+res) | LN | GEGLU-proj | ff-out(+res) | proj_out(+res)
  up-block concat    = virtual (two-source A operand / two-source GroupNorm), never materialised
  nearest-2x + conv  = one conv with the upsample folded into the gather
W is a ``packing.pack_state_dict`` result (device f16).  Names follow the diffusers checkpoint keys.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import os

import torch

from ._lib import ACT_GEGLU, ACT_GELU, ACT_NONE, ACT_QUICK_GELU, ACT_RELU, ACT_SILU, ACT_TANH3
from .engine import Engine, Norm


def _rup(x, m):
    return (x + m - 1) // m * m


def _heads(cfg, i):
    ahd = cfg["attention_head_dim"]
    return ahd[i] if isinstance(ahd, (list, tuple)) else ahd


# ------------------------------------------------------------------------------------------------ time embedding
def emit_added_cond(E: Engine, cfg, added) -> torch.Tensor:
    """SDXL ``text_time`` input of ``add_embedding``: cat(text_embeds [B, P] f16, sinusoid(time_ids [B, 6] f32)) -> [B, P + 6*dim]
    (diffusion/train_controlnet_sdxl_genima.py:1236-1262).  Recordable: the concat is two strided device copies."""
    text_embeds, time_ids = added
    B, P = text_embeds.shape
    dim = cfg["addition_time_embed_dim"]
    te = E.timestep_embedding(time_ids.reshape(-1), dim, cfg.get("flip_sin_to_cos", True), cfg.get("freq_shift", 0), name="a_sin")
    n = P + 6 * dim
    a = E.buf("a_cat", (B, n))
    E.copy4d(text_embeds, a, (1, 1, 1, B), (0, 0, 0, text_embeds.stride(0)), (0, 0, 0, n), P)
    E.copy4d(te, a[:, P:], (1, 1, 1, B), (0, 0, 0, 6 * dim), (0, 0, 0, n), 6 * dim)
    return a


def emit_time_shifts(E: Engine, W, cfg, t_dev: torch.Tensor, added=None) -> torch.Tensor:
    """t_dev f32 [B] -> all ResNet time shifts [B, temb_total] (one GEMM for every ``time_emb_proj``).
    ``added`` = (text_embeds, time_ids) for SDXL's ``addition_embed_type="text_time"``."""
    c0 = cfg["block_out_channels"][0]
    e = E.timestep_embedding(t_dev, c0, cfg.get("flip_sin_to_cos", True), cfg.get("freq_shift", 0), name="t_sin")
    e = E.linear(e, W["time_embedding.linear_1.weight"], W["time_embedding.linear_1.bias"], act=ACT_SILU, name="t_l1")
    if cfg.get("addition_embed_type") == "text_time":
        emb = E.linear(e, W["time_embedding.linear_2.weight"], W["time_embedding.linear_2.bias"], name="t_l2")
        a = E.linear(emit_added_cond(E, cfg, added), W["add_embedding.linear_1.weight"], W["add_embedding.linear_1.bias"], act=ACT_SILU, name="a_l1")
        emb = E.linear(a, W["add_embedding.linear_2.weight"], W["add_embedding.linear_2.bias"], residual=emb, name="a_l2")
        e = E.act(emb, ACT_SILU, name="t_act")
    else:
        # linear_2, then the SiLU every ResnetBlock2D applies to temb before its time_emb_proj
        e = E.linear(e, W["time_embedding.linear_2.weight"], W["time_embedding.linear_2.bias"], act=ACT_SILU, name="t_l2")
    return E.linear(e, W["time_emb_proj_all.weight"], W["time_emb_proj_all.bias"], name="t_shifts")


def _shift_for(W, shifts: Optional[torch.Tensor], prefix: str):
    if shifts is None:
        return None, 0
    off, n = W["__meta__"]["temb_slices"][prefix]
    return shifts[:, off:off + n], shifts.shape[1]


# ------------------------------------------------------------------------------------------------ blocks
def emit_resnet(E: Engine, W, p: str, x, x2, shifts, groups: int, eps: float, eps_in: Optional[float] = None):
    """eps_in: norm1's epsilon when the block input carries a scaled residual stream (packing.scale_vae_stream); default eps."""
    with E.scope(p):
        has_sc = (p + ".conv_shortcut.weight") in W
        # the 1x1 shortcut conv only reads the block input: where the program's side stream is idle (the UNet decoder, after the
        # ControlNet has been joined) AND the batch is too small to fill the chip, it runs there beside GroupNorm -> conv1 -> GroupNorm
        # ... unless it rides in conv2's K loop (one launch, no round trip of its output): gn_gemm_desc.k_append, packing `conv2sc` -- better
        # than the side stream at every batch size (single view 20.1 vs 20.8 ms, tiled B = 1 29.0 vs 29.9: profiles/r04_v6_side_free_ab.txt)
        c1w, c2w = W[p + ".conv1.weight"], W[p + ".conv2.weight"]

        def fuse(t, cout):  # GroupNorm-apply + SiLU inside the consuming conv's LDS patch (csrc/conv_gn.hip): the large, unconditioned convs
            return (getattr(E, "conv_gn", True) and t.dim() == 4 and t.shape[1] * t.shape[2] >= getattr(E, "conv_gn_min_hw", 0)
                    and E.conv2d_gn_supported(t, cout))

        # conv2 on the fused GroupNorm route (conv_gn.hip) takes the shortcut as its epilogue residual, so the two routes are exclusive: where conv2
        # will be fused the 1x1 shortcut stays its own launch (round 4 dropped it there: ADVICE r4, tests/test_conv_gn_gpu.py::test_vae_resnet_shortcut_routes)
        fuse2 = x.dim() == 4 and fuse(x[..., :1].expand(*x.shape[:3], c1w.shape[0]), c2w.shape[0])
        kapp = (has_sc and not fuse2 and getattr(E, "k_append", True) and (p + ".conv2sc.weight") in W and x.dim() == 4
                and (x2 is None or x.shape[-1] % 64 == 0) and x.shape[0] * x.shape[1] * x.shape[2] >= getattr(E, "k_append_min_rows", 0))
        side = has_sc and not kapp and getattr(E, "side_free", False) and E.record
        if kapp:
            sc = None
        elif has_sc:
            if side:
                E.fork()
            sc = E.conv2d(x, W[p + ".conv_shortcut.weight"], W[p + ".conv_shortcut.bias"], ksize=1, x2=x2, name="sc")
            if side:
                E.main()
        else:
            assert x2 is None
            sc = x
        sh, ld = _shift_for(W, shifts, p) if (p + ".time_emb_proj.weight") in W else (None, 0)

        if x2 is None and sh is None and fuse(x, c1w.shape[0]):
            st = E.groupnorm_stats(x, W[p + ".norm1.weight"], W[p + ".norm1.bias"], groups, eps if eps_in is None else eps_in, name="n1s")
            h = E.conv2d_gn(x, st, c1w, W[p + ".conv1.bias"], name="c1")
        else:
            # GroupNorm -> SiLU -> conv: through the GroupNorm bridge where this program wrote x (Engine.conv2d(norm=...): the statistics come out
            # of the producer, the conv normalises its own A tiles or one apply launch does), else the GroupNorm launch and the conv on its output
            n1 = Norm(W[p + ".norm1.weight"], W[p + ".norm1.bias"], groups, eps if eps_in is None else eps_in, ACT_SILU, "n1")
            h = E.conv2d(x, c1w, W[p + ".conv1.bias"], x2=x2, shift=sh, ldshift=ld, norm=n1, name="c1")
        if fuse2:
            assert not kapp and tuple(h.shape) == tuple(x.shape[:3]) + (c1w.shape[0],)
            st = E.groupnorm_stats(h, W[p + ".norm2.weight"], W[p + ".norm2.bias"], groups, eps, name="n2s")
            if side:
                E.join()
            return E.conv2d_gn(h, st, c2w, W[p + ".conv2.bias"], residual=sc, name="c2")
        n2 = Norm(W[p + ".norm2.weight"], W[p + ".norm2.bias"], groups, eps, ACT_SILU, "n2")
        if side:
            E.join()
        if kapp:  # (a concatenated block input -- the up blocks -- is appended as its two tensors)
            return E.conv2d(h, W[p + ".conv2sc.weight"], W[p + ".conv2sc.bias"], append=x, append2=x2, norm=n2, name="c2sc")
        return E.conv2d(h, c2w, W[p + ".conv2.bias"], residual=sc, norm=n2, name="c2")


def emit_cross_kv(E: Engine, W, ctx: torch.Tensor, tag: str) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    """K / V^T projections of the (constant) prompt states for every cross-attention layer, hoisted out of the step loop."""
    B, L, _ = ctx.shape
    kv = {}
    sites = [name[: -len(".to_k.weight")] for name in W if name.endswith(".attn2.to_k.weight")]
    offs = (W.get("__meta__") or {}).get("cross_kv") if hasattr(W, "get") else None
    if (offs and "cross_kv_all.weight" in W and getattr(E, "kv_all", os.environ.get("GN_KV_ALL", "1") != "0") and not E._fp8_weights
            and all(q in offs and offs[q][1] % 64 == 0 for q in sites)):
        # ONE launch for the whole network: [B, L, sum 2 C] = ctx @ [to_k ; to_v ; to_k ; ...]^T (packing `cross_kv_all`); a layer's K and V are column
        # slices of it (row stride = the full width), V row-major: the attention kernel transposes it out of its LDS tile (gn_attn_desc.v_rowmajor,
        # head dim 64 -- every SD / SDXL head), bit-identical to the V^T path at 77 keys
        with E.scope(tag):
            allkv = E.linear(ctx, W["cross_kv_all.weight"], name="kv_all")
        for q in sites:
            o, c = offs[q]
            kv[q] = (allkv[:, :, o:o + c], allkv[:, :, o + c:o + 2 * c], True)
        return kv
    pad = _rup(L, 64)
    with E.zero_pool(sum(B * W[p + ".to_v.weight"].shape[0] * pad + 64 for p in sites) if pad != L else 0):  # (eager engines: one fill for all V^T)
        for p in sites:
            with E.scope(tag + "/" + p):
                k = E.linear(ctx, W[p + ".to_k.weight"], name="k")
                vt = E.linear(ctx, W[p + ".to_v.weight"], transposed_out=True, rows_per_batch=L, pad_cols=pad, name="vt")
            kv[p] = (k, vt)
    return kv


def _cross_attention(E: Engine, q, kvp, heads: int):
    """attn2 against the hoisted prompt projections: (K, V^T) of emit_cross_kv's per-layer launches, or (K, V, True) = column slices of its one
    combined launch (V row-major)."""
    ck, cv = kvp[0], kvp[1]
    if len(kvp) > 2 and kvp[2]:
        return E.attention(q, ck, cv, heads, Nk=ck.shape[1], v_rowmajor=True, name="ca")
    return E.attention(q, ck, cv, heads, Nk=ck.shape[1], name="ca")


def _ln_fold(E: Engine, W, lin: str):
    """-> (ln_weight, ln_c1, ln_c2) of a Linear whose LayerNorm was folded at pack time, or None (no fold in the dict, the engine routes
    Linears through the fp8 MFMA -- its quantisation pass wants the normalised rows --, or GN_LN_FOLD=0)."""
    if lin + ".ln_weight" not in W or E._fp8_weights or not getattr(E, "ln_fold", True):
        return None
    return W[lin + ".ln_weight"], W[lin + ".ln_c1"], W[lin + ".ln_c2"]


def emit_transformer(E: Engine, W, p: str, x, kv, heads: int, groups: int):
    B, H, Wd, Cc = x.shape
    N = H * Wd
    with E.scope(p):
        # (E.tblock_any_fold: the trainer's frozen front switches the folded gn_gemm launches off -- their tune table is the inference
        # graphs' -- but keeps the chains, which carry their own folded weights in the tape)
        use_tb = (getattr(E, "tblock", True) and not E._fp8_weights and (getattr(E, "ln_fold", True) or getattr(E, "tblock_any_fold", False))
                  and B * N >= getattr(E, "tblock_min_rows", 0)
                  and f"{p}.transformer_blocks.0.tblock_tail.tape" in W and E.tblock_supported(B * N, Cc))
        front = None
        if use_tb and p + ".tblock_front.tape" in W and N % 128 == 0 and getattr(E, "tblock_front_on", True):
            # GroupNorm (from its statistics-only pass) + proj_in + norm1 -> q | k | v as ONE launch on rows resident in LDS (csrc/tblock.hip)
            st = E.groupnorm_stats(x, W[p + ".norm.weight"], W[p + ".norm.bias"], groups, 1e-6, name="gns")
            front = E.tblock_front(x.view(B, N, Cc), st, W[p + ".tblock_front.tape"], N, name="front")
            h = front[0]
        else:
            h = E.linear(x.view(B, N, Cc), W[p + ".proj_in.weight"], W[p + ".proj_in.bias"],
                         norm=Norm(W[p + ".norm.weight"], W[p + ".norm.bias"], groups, 1e-6, ACT_NONE, "gn"), name="pin")
        k = 0
        while f"{p}.transformer_blocks.{k}.norm1.weight" in W:
            b = f"{p}.transformer_blocks.{k}"
            with E.scope(f"tb{k}"):
                if front is not None and k == 0:
                    _, qk, vt = front
                    a = E.attention(qk[:, :, :Cc], qk[:, :, Cc:], vt, heads, name="sa")
                    h1, q = E.tblock_mid(a, h, W[b + ".tblock_mid.tape"], name="mid")
                    a = _cross_attention(E, q, kv[b + ".attn2"], heads)
                    out = E.tblock_tail(a, h1, x.view(B, N, Cc), W[b + ".tblock_tail.tape"], name="tail")
                    return out.view(B, H, Wd, Cc)
                # LayerNorm folded into the consuming Linear where the packed dict carries the folded weights (packing.fold_layernorms):
                # the Linear reads the raw rows and takes mean / rstd from its own K loop -- no LayerNorm launch, no round trip
                fold = _ln_fold(E, W, b + ".attn1.to_qkv")
                # V row-major (head dim 64): q | k | v is then ONE plain [.., 3C] launch and the attention kernel transposes V out of its LDS
                # tile (gn_attn_desc.v_rowmajor) -- no batch-transposed V^T epilogue (2-byte stores) on the projection
                vrow = getattr(E, "rowmajor_v", False) and Cc // heads == 64 and b + ".attn1.to_qkv.weight" in W
                if fold and vrow:
                    qkv = E.linear(h, fold[0], fold[2], ln_c1=fold[1], name="qkv")
                elif fold:
                    qk, vt = E.linear(h, fold[0], fold[2], ln_c1=fold[1], split_n=2 * Cc, rows_per_batch=N, pad_cols=_rup(N, 64), name="qk")
                else:
                    n = E.layernorm(h, W[b + ".norm1.weight"], W[b + ".norm1.bias"], name="ln1")
                    if vrow:
                        qkv = E.linear(n, W[b + ".attn1.to_qkv.weight"], name="qkv")
                    elif b + ".attn1.to_qkv.weight" in W and not (E._fp8_weights and b + ".attn1.to_qk.weight" in W):
                        # q | k | v in one two-destination launch (q, k row-major + V^T); with Linears routed to the fp8 MFMA the
                        # registered copies are to_qk / to_v (the two-destination epilogue has no fp8 form)
                        qk, vt = E.linear(n, W[b + ".attn1.to_qkv.weight"], split_n=2 * Cc, rows_per_batch=N, pad_cols=_rup(N, 64), name="qk")
                    else:
                        qk = E.linear(n, W[b + ".attn1.to_qk.weight"], name="qk")
                        vt = E.linear(n, W[b + ".attn1.to_v.weight"], transposed_out=True, rows_per_batch=N, pad_cols=_rup(N, 64), name="vt")
                if vrow:
                    a = E.attention(qkv[:, :, :Cc], qkv[:, :, Cc:2 * Cc], qkv[:, :, 2 * Cc:], heads, v_rowmajor=True, name="sa")
                else:
                    a = E.attention(qk[:, :, :Cc], qk[:, :, Cc:], vt, heads, name="sa")
                if use_tb and k == 0:
                    # attn1.to_out .. attn2.to_q and attn2.to_out .. proj_out as TWO launches that keep their rows of the residual stream in
                    # LDS and stream the weights from a tape (csrc/tblock.hip) instead of six gn_gemm launches
                    h1, q = E.tblock_mid(a, h, W[b + ".tblock_mid.tape"], name="mid")
                    a = _cross_attention(E, q, kv[b + ".attn2"], heads)
                    out = E.tblock_tail(a, h1, x.view(B, N, Cc), W[b + ".tblock_tail.tape"], name="tail")
                    return out.view(B, H, Wd, Cc)
                h = E.linear(a, W[b + ".attn1.to_out.0.weight"], W[b + ".attn1.to_out.0.bias"], residual=h, name="sao")
                fold = _ln_fold(E, W, b + ".attn2.to_q")
                if fold:
                    q = E.linear(h, fold[0], fold[2], ln_c1=fold[1], name="cq")
                else:
                    n = E.layernorm(h, W[b + ".norm2.weight"], W[b + ".norm2.bias"], name="ln2")
                    q = E.linear(n, W[b + ".attn2.to_q.weight"], name="cq")
                a = _cross_attention(E, q, kv[b + ".attn2"], heads)
                h = E.linear(a, W[b + ".attn2.to_out.0.weight"], W[b + ".attn2.to_out.0.bias"], residual=h, name="cao")
                fold = _ln_fold(E, W, b + ".ff.net.0.proj")
                if fold:
                    g = E.linear(h, fold[0], fold[2], ln_c1=fold[1], act=ACT_GEGLU, name="ffg")
                else:
                    n = E.layernorm(h, W[b + ".norm3.weight"], W[b + ".norm3.bias"], name="ln3")
                    g = E.linear(n, W[b + ".ff.net.0.proj.weight"], W[b + ".ff.net.0.proj.bias"], act=ACT_GEGLU, name="ffg")
                if (f"{p}.transformer_blocks.{k + 1}.norm1.weight" not in W and getattr(E, "k_append", True) and (p + ".ffo_pout.weight") in W
                        and not E._fp8_weights):
                    # the block's last Linear and the transformer's proj_out as one GEMM over [g | h] (packing `ffo_pout`, gn_gemm_desc.k_append)
                    out = E.linear(g, W[p + ".ffo_pout.weight"], W[p + ".ffo_pout.bias"], residual=x.view(B, N, Cc), append=h, name="ffpo")
                    return out.view(B, H, Wd, Cc)
                h = E.linear(g, W[b + ".ff.net.2.weight"], W[b + ".ff.net.2.bias"], residual=h, name="ffo")
            k += 1
        out = E.linear(h, W[p + ".proj_out.weight"], W[p + ".proj_out.bias"], residual=x.view(B, N, Cc), name="pout")
        return out.view(B, H, Wd, Cc)


def _emit_encoder(E: Engine, W, cfg, h, shifts, kv):
    G, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    skips = [h]
    nlev = len(cfg["block_out_channels"])
    for i, btype in enumerate(cfg["down_block_types"]):
        for j in range(cfg["layers_per_block"]):
            h = emit_resnet(E, W, f"down_blocks.{i}.resnets.{j}", h, None, shifts, G, eps)
            if btype == "CrossAttnDownBlock2D":
                h = emit_transformer(E, W, f"down_blocks.{i}.attentions.{j}", h, kv, _heads(cfg, i), G)
            skips.append(h)
        if i != nlev - 1:
            p = f"down_blocks.{i}.downsamplers.0.conv"
            h = E.conv2d(h, W[p + ".weight"], W[p + ".bias"], stride=2, name=p)
            skips.append(h)
    return h, skips


def _emit_mid(E: Engine, W, cfg, h, shifts, kv):
    G, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    h = emit_resnet(E, W, "mid_block.resnets.0", h, None, shifts, G, eps)
    h = emit_transformer(E, W, "mid_block.attentions.0", h, kv, _heads(cfg, len(cfg["block_out_channels"]) - 1), G)
    return emit_resnet(E, W, "mid_block.resnets.1", h, None, shifts, G, eps)


def _emit_upsample_conv(E: Engine, W, p: str, h: torch.Tensor) -> torch.Tensor:
    """Upsample2D = nearest 2x + 3x3 conv: four 2x2 phase convs on the source pixels where the packed dict carries the phase weights
    (packing.pack_upsample_phases: f16 inference dicts), else the 3x3 conv with the upsample fused into its gather."""
    # measured per shape (MI355X, B = 8 tiled call): 2.35 -> 1.42 ms and 2.09 -> 1.15 ms on the VAE's 256^2 -> 512^2 / 128^2 -> 256^2 upsamplers,
    # 1.15 -> 0.96 ms on the UNet's 32^2 -> 64^2 one; with the four phases as ONE launch also a gain at 2048 source rows (call 103.7 -> 103.1 ms)
    # and at the B = 1 call's 1024 rows, still a loss at 512 (the 3x3 launch with its K split wins there)
    rows = h.shape[0] * h.shape[1] * h.shape[2]
    if getattr(E, "up_phases", True) and (p + ".up4.weight") in W and rows >= getattr(E, "up_phases_min_rows", 1024):
        return E.conv2d_up2x(h, W[p + ".up4.weight"], W[p + ".bias"], name=p)
    return E.conv2d(h, W[p + ".weight"], W[p + ".bias"], upsample2x=True, name=p)


def emit_unet(E: Engine, W, cfg, x8: torch.Tensor, t_dev: torch.Tensor, kv, down_res: Optional[Sequence[torch.Tensor]] = None,
              mid_res: Optional[torch.Tensor] = None, added=None, before_residuals=None, shifts: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x8: scaled latents [B, H, W, 8] (channels >= in_channels zero).  Returns eps [B, H, W, 8] (first out_channels valid).
    ``added`` = (text_embeds, time_ids): SDXL added conditions.  ``before_residuals``: called once the encoder and mid block are
    emitted and before the first ControlNet residual is consumed (the pipeline joins the ControlNet's stream there)."""
    G, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    with E.scope("unet"):
        if shifts is None:  # (the pipeline computes the time shifts of ALL its steps in one pass and hands each step its rows)
            shifts = emit_time_shifts(E, W, cfg, t_dev, added)
        h = E.conv2d(x8, W["conv_in.weight"], W["conv_in.bias"], name="conv_in")
        h, skips = _emit_encoder(E, W, cfg, h, shifts, kv)
        h = _emit_mid(E, W, cfg, h, shifts, kv)
        if before_residuals is not None:
            before_residuals()
            E.side_free = x8.shape[0] * x8.shape[1] * x8.shape[2] <= getattr(E, "side_free_max_rows", 4096)  # the caller joined its side stream: free for the decoder at small batch
        # the residual adds only feed the decoder, so they sit after the mid block (same values; lets the encoder + mid overlap the ControlNet)
        if callable(down_res):  # emit_controlnet(defer_zero_convs=True): the zero convs run HERE with the skip as their residual operand (no add launch)
            skips, h = down_res(skips, h)
        elif down_res is not None and mid_res is not None and len(skips) < 16 and getattr(E, "add_multi_on", True):
            outs = E.add_multi(list(zip(skips, down_res)) + [(h, mid_res)], name="res_add")  # thirteen adds, one launch (gn_add_multi)
            skips, h = outs[:-1], outs[-1]
        else:
            if down_res is not None:
                skips = [E.add(s, r, name=f"skip_add{i}") for i, (s, r) in enumerate(zip(skips, down_res))]
            if mid_res is not None:
                h = E.add(h, mid_res, name="mid_add")
        nlev = len(cfg["block_out_channels"])
        for i, btype in enumerate(cfg["up_block_types"]):
            for j in range(cfg["layers_per_block"] + 1):
                h = emit_resnet(E, W, f"up_blocks.{i}.resnets.{j}", h, skips.pop(), shifts, G, eps)
                if btype == "CrossAttnUpBlock2D":
                    h = emit_transformer(E, W, f"up_blocks.{i}.attentions.{j}", h, kv, _heads(cfg, nlev - 1 - i), G)
            if i != nlev - 1:
                p = f"up_blocks.{i}.upsamplers.0.conv"
                h = _emit_upsample_conv(E, W, p, h)
        E.side_free = False
        return E.conv2d(h, W["conv_out.weight"], W["conv_out.bias"], norm=Norm(W["conv_norm_out.weight"], W["conv_norm_out.bias"], G, eps, ACT_SILU, "norm_out"),
                        name="conv_out")


def emit_controlnet_cond(E: Engine, W, cfg, cond8: torch.Tensor) -> torch.Tensor:
    """controlnet_cond_embedding: constant over the denoise loop, so the pipeline runs it once per call."""
    p = "controlnet_cond_embedding"
    with E.scope("cn_cond"):
        h = E.conv2d(cond8, W[p + ".conv_in.weight"], W[p + ".conv_in.bias"], act=ACT_SILU, name="in")
        n = len(cfg["conditioning_embedding_out_channels"]) - 1
        for i in range(n):
            h = E.conv2d(h, W[f"{p}.blocks.{2 * i}.weight"], W[f"{p}.blocks.{2 * i}.bias"], act=ACT_SILU, name=f"b{2 * i}")
            h = E.conv2d(h, W[f"{p}.blocks.{2 * i + 1}.weight"], W[f"{p}.blocks.{2 * i + 1}.bias"], act=ACT_SILU, stride=2, name=f"b{2 * i + 1}")
        return E.conv2d(h, W[p + ".conv_out.weight"], W[p + ".conv_out.bias"], name="out")


def emit_controlnet(E: Engine, W, cfg, x8, t_dev, kv, cond_emb: torch.Tensor, conditioning_scale: float = 1.0, added=None,
                    shifts: Optional[torch.Tensor] = None, defer_zero_convs: bool = False):
    """-> (list of down residuals (12 for SD-2.x, 9 for SDXL), mid residual), NHWC.
    ``defer_zero_convs``: -> (fn, None) instead; fn(unet_skips, unet_h) -> (skips, h) emits the zero convs (controlnet_down_blocks.* /
    controlnet_mid_block, 1x1) with the UNet's skip / mid tensor as their residual operand -- diffusers' ``sample + residual`` adds
    (UNet2DConditionModel.forward: down_block_res_samples + down_block_additional_residuals, mid_block_additional_residual) ride in the zero
    convs' epilogues -- and, while the side stream is free (a small batch, after the join), deals them over both streams."""
    with E.scope("cn"):
        if shifts is None:
            shifts = emit_time_shifts(E, W, cfg, t_dev, added)
        h = E.conv2d(x8, W["conv_in.weight"], W["conv_in.bias"], residual=cond_emb, name="conv_in")
        h, skips = _emit_encoder(E, W, cfg, h, shifts, kv)
        h = _emit_mid(E, W, cfg, h, shifts, kv)
        if defer_zero_convs:
            cn_skips, cn_h = skips, h

            def residuals(un_skips, un_h):
                jobs = [(f"controlnet_down_blocks.{i}", s, u) for i, (s, u) in enumerate(zip(cn_skips, un_skips))] + [("controlnet_mid_block", cn_h, un_h)]
                two = bool(getattr(E, "side_free", False)) and E.record and getattr(E, "zero_conv_split", True)
                order = [j for k, j in enumerate(jobs) if k % 2 == 1] + [j for k, j in enumerate(jobs) if k % 2 == 0] if two else jobs
                res = {}
                with E.scope("cn"):
                    if two:
                        E.fork()
                    for k, (p, s, u) in enumerate(order):
                        if two and k == len(jobs) // 2:
                            E.main()
                        res[p] = E.conv2d(s, W[p + ".weight"], W[p + ".bias"], ksize=1, out_scale=conditioning_scale, residual=u, name=p)
                    if two:
                        E.join()
                return [res[p] for p, _, _ in jobs[:-1]], res[jobs[-1][0]]
            return residuals, None
        outs = []
        for i, s in enumerate(skips):
            p = f"controlnet_down_blocks.{i}"
            outs.append(E.conv2d(s, W[p + ".weight"], W[p + ".bias"], ksize=1, out_scale=conditioning_scale, name=p))
        p = "controlnet_mid_block"
        mid = E.conv2d(h, W[p + ".weight"], W[p + ".bias"], ksize=1, out_scale=conditioning_scale, name=p)
        return outs, mid


# ------------------------------------------------------------------------------------------------ AutoencoderKL
def _vae_eps(W) -> float:
    """epsilon of the GroupNorms that read the VAE's residual stream: 1e-6 x (stream scale)^2 (packing.scale_vae_stream)."""
    meta = W.get("__meta__") or {}
    return 1e-6 * float(meta.get("vae_stream_scale", 1.0)) ** 2


def _emit_vae_attention(E: Engine, W, p: str, x, groups: int):
    """Single-head, d = C attention of the VAE mid block: QK^T and PV as plain MFMA GEMMs around a row softmax (the d=512
    head does not fit the flash kernel's register tile; it is 1.4 % of the decoder's FLOPs)."""
    B, H, Wd, Cc = x.shape
    N = H * Wd
    with E.scope(p):
        h = E.groupnorm(x, W[p + ".group_norm.weight"], W[p + ".group_norm.bias"], groups, _vae_eps(W), name="gn").view(B, N, Cc)
        q = E.linear(h, W[p + ".to_q.weight"], W[p + ".to_q.bias"], name="q")
        k = E.linear(h, W[p + ".to_k.weight"], W[p + ".to_k.bias"], name="k")
        Np = _rup(N, 64)
        vt = E.linear(h, W[p + ".to_v.weight"], W[p + ".to_v.bias"], transposed_out=True, rows_per_batch=N, pad_cols=Np, name="vt")
        a = E.buf("a", (B, N, Cc))
        s = E.buf("s", (N, Np), zero=True)
        for b in range(B):
            E.linear(q[b], k[b], out=s[:, :N] if Np == N else s[:, :N])
            E.softmax_rows(s[:, :N], float(Cc) ** -0.5)
            E.linear(s[:, :N], vt[b][:, :N], out=a[b])
        o = E.linear(a, W[p + ".to_out.0.weight"], W[p + ".to_out.0.bias"], residual=x.view(B, N, Cc), name="o")
        return o.view(B, H, Wd, Cc)


def _emit_vae_mid(E, W, p, h, G):
    h = emit_resnet(E, W, p + ".resnets.0", h, None, None, G, 1e-6, eps_in=_vae_eps(W))
    h = _emit_vae_attention(E, W, p + ".attentions.0", h, G)
    return emit_resnet(E, W, p + ".resnets.1", h, None, None, G, 1e-6, eps_in=_vae_eps(W))


def emit_vae_decode(E: Engine, W, cfg, z8: torch.Tensor) -> torch.Tensor:
    """z8: latents / scaling_factor, [B, h, w, 8] (channels >= 4 zero) -> image [B, 8h, 8w, 8] (first 3 channels valid)."""
    G = cfg["norm_num_groups"]
    n = len(cfg["block_out_channels"])
    with E.scope("vae_dec"):
        h = E.conv2d(z8, W["post_quant_conv.weight"], W["post_quant_conv.bias"], ksize=1, name="pq")
        h = E.conv2d(h, W["decoder.conv_in.weight"], W["decoder.conv_in.bias"], name="conv_in")
        h = _emit_vae_mid(E, W, "decoder.mid_block", h, G)
        for i in range(n):
            for j in range(cfg["layers_per_block"] + 1):
                h = emit_resnet(E, W, f"decoder.up_blocks.{i}.resnets.{j}", h, None, None, G, 1e-6, eps_in=_vae_eps(W))
            if i != n - 1:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                h = _emit_upsample_conv(E, W, p, h)
        cw = W["decoder.conv_out.weight"]
        if (getattr(E, "conv_gn", True) and h.shape[1] * h.shape[2] >= getattr(E, "conv_gn_min_hw", 0) and cw.shape[1] == 9 * h.shape[-1]
                and E.conv2d_gn_supported(h, cw.shape[0])):
            # conv_norm_out + SiLU on the conv's LDS patch (the narrow variant of csrc/conv_gn.hip: 3 of 8 padded output channels)
            st = E.groupnorm_stats(h, W["decoder.conv_norm_out.weight"], W["decoder.conv_norm_out.bias"], G, _vae_eps(W), name="norm_out_s")
            return E.conv2d_gn(h, st, cw, W["decoder.conv_out.bias"], name="conv_out")
        h = E.groupnorm(h, W["decoder.conv_norm_out.weight"], W["decoder.conv_norm_out.bias"], G, _vae_eps(W), act=ACT_SILU, name="norm_out")
        return E.conv2d(h, cw, W["decoder.conv_out.bias"], name="conv_out")


def emit_vae_encode_moments(E: Engine, W, cfg, x8: torch.Tensor) -> torch.Tensor:
    """x8: image in [-1, 1], [B, H, W, 8] -> moments [B, H/8, W/8, 8] = (mean[0:4], logvar[4:8]) before clamping."""
    G = cfg["norm_num_groups"]
    n = len(cfg["block_out_channels"])
    with E.scope("vae_enc"):
        h = E.conv2d(x8, W["encoder.conv_in.weight"], W["encoder.conv_in.bias"], name="conv_in")
        for i in range(n):
            for j in range(cfg["layers_per_block"]):
                h = emit_resnet(E, W, f"encoder.down_blocks.{i}.resnets.{j}", h, None, None, G, 1e-6, eps_in=_vae_eps(W))
            if i != n - 1:
                p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                h = E.conv2d(h, W[p + ".weight"], W[p + ".bias"], stride=2, pad=(0, 0, 1, 1), name=p)  # F.pad (0,1,0,1)
        h = _emit_vae_mid(E, W, "encoder.mid_block", h, G)
        h = E.groupnorm(h, W["encoder.conv_norm_out.weight"], W["encoder.conv_norm_out.bias"], G, _vae_eps(W), act=ACT_SILU, name="norm_out")
        h = E.conv2d(h, W["encoder.conv_out.weight"], W["encoder.conv_out.bias"], name="conv_out")
        return E.conv2d(h, W["quant_conv.weight"], W["quant_conv.bias"], ksize=1, name="quant")


# ------------------------------------------------------------------------------------------------ AutoencoderTiny (TAESD)
def _emit_tiny_block(E: Engine, W, p: str, x):
    """AutoencoderTinyBlock: relu(conv(relu(conv(relu(conv(x))))) + skip(x)); the skip is the identity at equal widths."""
    with E.scope(p):
        h = E.conv2d(x, W[p + ".conv.0.weight"], W[p + ".conv.0.bias"], act=ACT_RELU, name="c0")
        h = E.conv2d(h, W[p + ".conv.2.weight"], W[p + ".conv.2.bias"], act=ACT_RELU, name="c2")
        sk = E.conv2d(x, W[p + ".skip.weight"], None, ksize=1, name="skip") if (p + ".skip.weight") in W else x
        return E.conv2d(h, W[p + ".conv.4.weight"], W[p + ".conv.4.bias"], residual=sk, act=ACT_RELU, residual_before_act=True, name="c4")


def emit_taesd_decode(E: Engine, W, cfg, z8: torch.Tensor) -> torch.Tensor:
    """diffusers ``DecoderTiny.forward``: x = 3 tanh(x / 3); conv + relu; per stage {blocks, nearest-2x + bias-free conv}; last stage
    {block, conv to RGB}; x * 2 - 1 (the affine is folded into the last conv: out_scale 2 and the pre-shifted bias
    ``decoder.out_bias_shifted`` = bias - 0.5 that host.AutoencoderTiny packs).  z8 [B, h, w, 8] -> image [B, 8h, 8w, 8] in [-1, 1]."""
    nb = cfg["num_decoder_blocks"]
    with E.scope("taesd_dec"):
        h = E.act(z8, ACT_TANH3, name="clamp")
        h = E.conv2d(h, W["decoder.layers.0.weight"], W["decoder.layers.0.bias"], act=ACT_RELU, name="conv_in")
        idx = 2
        for i, n in enumerate(nb):
            final = i == len(nb) - 1
            for _ in range(n):
                h = _emit_tiny_block(E, W, f"decoder.layers.{idx}", h)
                idx += 1
            if not final:
                idx += 1
                h = E.conv2d(h, W[f"decoder.layers.{idx}.weight"], None, upsample2x=True, name=f"up{i}")
            else:
                h = E.conv2d(h, W[f"decoder.layers.{idx}.weight"], W["decoder.out_bias_shifted"], out_scale=2.0, name="conv_out")
            idx += 1
        return h


# ------------------------------------------------------------------------------------------------ CLIP text tower
def emit_clip_text(E: Engine, W, cfg, ids: torch.Tensor, hidden: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """ids int32 [B, L] -> last_hidden_state f16 [B, L, D] (after final_layer_norm)."""
    B, L = ids.shape
    heads = cfg["num_attention_heads"]
    eps = cfg.get("layer_norm_eps", 1e-5)
    D = cfg["hidden_size"]
    act = ACT_QUICK_GELU if cfg["hidden_act"] == "quick_gelu" else ACT_GELU
    nz = cfg["num_hidden_layers"] * (B * D * _rup(L, 64) + 64) if _rup(L, 64) != L else 0  # every layer's zero-padded V^T (eager engines: one fill)
    with E.scope("clip"), E.zero_pool(nz):
        x = E.embedding(ids, W["text_model.embeddings.token_embedding.weight"],
                        W["text_model.embeddings.position_embedding.weight"], name="emb")
        for i in range(cfg["num_hidden_layers"]):
            p = f"text_model.encoder.layers.{i}"
            with E.scope(f"l{i}"):
                fold = _ln_fold(E, W, p + ".self_attn.qkv_proj")
                if fold:  # LayerNorm folded into the q | k | v launch (packing.fold_layernorms)
                    qk, vt = E.linear(x, fold[0], fold[2], ln_c1=fold[1], ln_eps=eps, split_n=2 * D, rows_per_batch=L, pad_cols=_rup(L, 64), name="qk")
                else:
                    n = E.layernorm(x, W[p + ".layer_norm1.weight"], W[p + ".layer_norm1.bias"], eps, name="ln1")
                    if p + ".self_attn.qkv_proj.weight" in W:  # q | k | v in one two-destination launch
                        qk, vt = E.linear(n, W[p + ".self_attn.qkv_proj.weight"], W[p + ".self_attn.qkv_proj.bias"], split_n=2 * D,
                                          rows_per_batch=L, pad_cols=_rup(L, 64), name="qk")
                    else:
                        qk = E.linear(n, W[p + ".self_attn.qk_proj.weight"], W[p + ".self_attn.qk_proj.bias"], name="qk")
                        vt = E.linear(n, W[p + ".self_attn.v_proj.weight"], W[p + ".self_attn.v_proj.bias"], transposed_out=True,
                                      rows_per_batch=L, pad_cols=_rup(L, 64), name="vt")
                a = E.attention(qk[:, :, :D], qk[:, :, D:], vt, heads, causal=True, name="sa")
                x = E.linear(a, W[p + ".self_attn.out_proj.weight"], W[p + ".self_attn.out_proj.bias"], residual=x, name="o")
                fold = _ln_fold(E, W, p + ".mlp.fc1")
                if fold:
                    h = E.linear(x, fold[0], fold[2], ln_c1=fold[1], ln_eps=eps, act=act, name="fc1")
                else:
                    n = E.layernorm(x, W[p + ".layer_norm2.weight"], W[p + ".layer_norm2.bias"], eps, name="ln2")
                    h = E.linear(n, W[p + ".mlp.fc1.weight"], W[p + ".mlp.fc1.bias"], act=act, name="fc1")
                x = E.linear(h, W[p + ".mlp.fc2.weight"], W[p + ".mlp.fc2.bias"], residual=x, name="fc2")
                if hidden is not None:
                    hidden.append(x)
        return E.layernorm(x, W["text_model.final_layer_norm.weight"], W["text_model.final_layer_norm.bias"], eps, name="ln_f")


def emit_clip_text_sdxl(E: Engine, W, cfg, ids: torch.Tensor):
    """SDXL ``encode_prompt`` per tower (diffusion/train_controlnet_sdxl_genima.py:854-893): -> (hidden_states[-2] [B, L, D],
    ``text_embeds`` [B, projection_dim] = text_projection(final_layer_norm(last)[eot]) for the projection tower, else None)."""
    hidden: List[torch.Tensor] = []
    last = emit_clip_text(E, W, cfg, ids, hidden)
    pooled = None
    if "text_projection.weight" in W:
        with E.scope("clip_pool"):
            eot = E.argmax_rows(ids, name="eot")
            pooled = E.linear(E.gather_rows(last, eot, name="pooled"), W["text_projection.weight"], name="proj")
    return hidden[-2], pooled
