"""CPU oracle: fp32 PyTorch restatement of the SD-Turbo UNet / ControlNet / AutoencoderKL / CLIP text graphs.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product package
(``genima_amd``); only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
use it, as the checker / the timed CPU baseline.

PARITY UNPINNED for the diffusion networks: the arithmetic of this path lives in third-party packages
that are absent from /root/reference and from this image -- diffusers==0.29.0 (poetry.lock:595-596;
call sites controller/agent/sd_controlnet_agent.py:32-49, :67-76 and
diffusion/train_controlnet_genima.py:1038-1071, 1329-1388) -- and the reference holds no tests or
golden vectors.  This file restates the published architecture of the pinned
version (diffusers ``models/unets/unet_2d_condition.py``,
``unet_2d_blocks.py``, ``resnet.py``, ``attention.py``, ``attention_processor.py``,
``transformers/transformer_2d.py``, ``embeddings.py``, ``controlnet.py``,
``autoencoders/{autoencoder_kl,vae}.py``) and is pinned structurally: parameter counts reproduce the
public checkpoints exactly (tests/test_schema.py) and the CLIP text tower is checked against the
installed ``transformers`` ``CLIPTextModel`` (tests/golden/make_clip_golden.py).

All functions are functional over a diffusers-named fp32 state dict ``sd`` (OIHW convs, [out,in]
linears) and use NCHW tensors like the reference.  ``q`` is an optional storage-rounding hook
(``lambda t: t.half().float()``) applied where the HIP path stores fp16 activations, used to separate
kernel error from storage-rounding error in end-to-end comparisons; default identity = pure fp32.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
_id = lambda t: t  # noqa: E731


# ----------------------------------------------------------------------------- small helpers
def _w(sd, name):
    return sd[name + ".weight"]


def _b(sd, name):
    return sd.get(name + ".bias")


def linear(sd, name, x):
    return F.linear(x, _w(sd, name), _b(sd, name))


def conv(sd, name, x, stride=1, padding=1):
    return F.conv2d(x, _w(sd, name), _b(sd, name), stride=stride, padding=padding)


def group_norm(sd, name, x, groups, eps):
    return F.group_norm(x, groups, _w(sd, name), _b(sd, name), eps)


def layer_norm(sd, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), _w(sd, name), _b(sd, name), eps)


def timestep_embedding(t: Tensor, dim: int, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000.0) -> Tensor:
    """diffusers ``get_timestep_embedding`` (SURVEY Appendix A.1 'time' row)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    emb = t.to(torch.float32)[:, None] * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def time_embed(sd, cfg, t: Tensor, q=_id, added=None) -> Tensor:
    """``added`` = (text_embeds [B, P], time_ids [B, 6]) for SDXL's ``addition_embed_type="text_time"`` (SURVEY Appendix A.5;
    diffusion/train_controlnet_sdxl_genima.py:1236-1262): aug = add_embedding(cat(text_embeds, sinusoid(time_ids))), emb += aug."""
    c0 = cfg["block_out_channels"][0]
    flip, shift = cfg.get("flip_sin_to_cos", True), cfg.get("freq_shift", 0)
    e = q(timestep_embedding(t, c0, flip, shift))
    e = q(F.silu(linear(sd, "time_embedding.linear_1", e)))
    emb = q(linear(sd, "time_embedding.linear_2", e))
    if cfg.get("addition_embed_type") == "text_time":
        text_embeds, time_ids = added
        B = text_embeds.shape[0]
        te = q(timestep_embedding(time_ids.reshape(-1), cfg["addition_time_embed_dim"], flip, shift)).reshape(B, -1)
        a = q(torch.cat([text_embeds, te], dim=-1))
        a = q(F.silu(linear(sd, "add_embedding.linear_1", a)))
        emb = q(emb + linear(sd, "add_embedding.linear_2", a))
    return emb


# ----------------------------------------------------------------------------- blocks
def resnet_block(sd, p, x, temb, groups, eps, q=_id):
    """diffusers ``ResnetBlock2D`` with time_embedding_norm="default" (additive shift only)."""
    h = q(F.silu(group_norm(sd, p + ".norm1", x, groups, eps)))
    h = conv(sd, p + ".conv1", h)
    if temb is not None and (p + ".time_emb_proj.weight") in sd:
        h = h + linear(sd, p + ".time_emb_proj", q(F.silu(temb)))[:, :, None, None]
    h = q(h)
    h = q(F.silu(group_norm(sd, p + ".norm2", h, groups, eps)))
    h = conv(sd, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in sd:
        x = q(conv(sd, p + ".conv_shortcut", x, padding=0))
    return q(x + h)


def attention(qh: Tensor, kh: Tensor, vh: Tensor, heads: int, causal=False) -> Tensor:
    """softmax(q k^T / sqrt(d)) v over [B, N, C] tensors split into ``heads``."""
    B, Nq, C = qh.shape
    d = C // heads
    qh = qh.view(B, Nq, heads, d).transpose(1, 2)
    kh = kh.view(B, -1, heads, d).transpose(1, 2)
    vh = vh.view(B, -1, heads, d).transpose(1, 2)
    s = torch.matmul(qh, kh.transpose(-1, -2)) * (d ** -0.5)
    if causal:
        m = torch.full((Nq, kh.shape[2]), float("-inf")).triu(1)
        s = s + m
    o = torch.matmul(torch.softmax(s, dim=-1), vh)
    return o.transpose(1, 2).reshape(B, Nq, C)


def basic_transformer_block(sd, p, x, ctx, heads, q=_id):
    """diffusers ``BasicTransformerBlock``: self-attn, cross-attn, GEGLU feed-forward (pre-LN, residual)."""
    n = q(layer_norm(sd, p + ".norm1", x))
    a = q(attention(q(linear(sd, p + ".attn1.to_q", n)), q(linear(sd, p + ".attn1.to_k", n)),
                    q(linear(sd, p + ".attn1.to_v", n)), heads))
    x = q(x + linear(sd, p + ".attn1.to_out.0", a))
    n = q(layer_norm(sd, p + ".norm2", x))
    a = q(attention(q(linear(sd, p + ".attn2.to_q", n)), q(linear(sd, p + ".attn2.to_k", ctx)),
                    q(linear(sd, p + ".attn2.to_v", ctx)), heads))
    x = q(x + linear(sd, p + ".attn2.to_out.0", a))
    n = q(layer_norm(sd, p + ".norm3", x))
    hcat = linear(sd, p + ".ff.net.0.proj", n)
    hid, gate = hcat.chunk(2, dim=-1)
    g = q(hid * F.gelu(gate))  # GEGLU, erf gelu
    return q(x + linear(sd, p + ".ff.net.2", g))


def transformer2d(sd, p, x, ctx, heads, groups, q=_id):
    """diffusers ``Transformer2DModel`` (use_linear_projection=True; norm eps 1e-6)."""
    B, C, H, W = x.shape
    res = x
    h = q(group_norm(sd, p + ".norm", x, groups, 1e-6))
    h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    h = q(linear(sd, p + ".proj_in", h))
    k = 0
    while f"{p}.transformer_blocks.{k}.norm1.weight" in sd:
        h = basic_transformer_block(sd, f"{p}.transformer_blocks.{k}", h, ctx, heads, q)
        k += 1
    h = linear(sd, p + ".proj_out", h)
    h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return q(h + res)


def _heads(cfg, i):
    ahd = cfg["attention_head_dim"]
    return ahd[i] if isinstance(ahd, (list, tuple)) else ahd


def _encoder(sd, cfg, h, emb, ctx, q=_id):
    """Down blocks + (not mid).  Returns (h, skips)."""
    G, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    skips = [h]
    nlev = len(cfg["block_out_channels"])
    for i, btype in enumerate(cfg["down_block_types"]):
        for j in range(cfg["layers_per_block"]):
            h = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", h, emb, G, eps, q)
            if btype == "CrossAttnDownBlock2D":
                h = transformer2d(sd, f"down_blocks.{i}.attentions.{j}", h, ctx, _heads(cfg, i), G, q)
            skips.append(h)
        if i != nlev - 1:
            h = q(conv(sd, f"down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=1))
            skips.append(h)
    return h, skips


def _mid(sd, cfg, h, emb, ctx, q=_id):
    G, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    h = resnet_block(sd, "mid_block.resnets.0", h, emb, G, eps, q)
    h = transformer2d(sd, "mid_block.attentions.0", h, ctx, _heads(cfg, len(cfg["block_out_channels"]) - 1), G, q)
    return resnet_block(sd, "mid_block.resnets.1", h, emb, G, eps, q)


def unet_forward(sd, cfg, sample: Tensor, t: Tensor, ctx: Tensor,
                 down_residuals: Optional[Sequence[Tensor]] = None, mid_residual: Optional[Tensor] = None,
                 q=_id, added=None) -> Tensor:
    """``UNet2DConditionModel.forward(...).sample`` (call site diffusion/train_controlnet_genima.py:1377-1388)."""
    G, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    if t.dim() == 0:
        t = t[None].expand(sample.shape[0])
    emb = time_embed(sd, cfg, t, q, added)
    h = q(conv(sd, "conv_in", sample))
    h, skips = _encoder(sd, cfg, h, emb, ctx, q)
    if down_residuals is not None:
        skips = [q(s + r) for s, r in zip(skips, down_residuals)]
    h = _mid(sd, cfg, h, emb, ctx, q)
    if mid_residual is not None:
        h = q(h + mid_residual)
    nlev = len(cfg["block_out_channels"])
    rev_heads = [_heads(cfg, nlev - 1 - i) for i in range(nlev)]
    for i, btype in enumerate(cfg["up_block_types"]):
        for j in range(cfg["layers_per_block"] + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_block(sd, f"up_blocks.{i}.resnets.{j}", h, emb, G, eps, q)
            if btype == "CrossAttnUpBlock2D":
                h = transformer2d(sd, f"up_blocks.{i}.attentions.{j}", h, ctx, rev_heads[i], G, q)
        if i != nlev - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = q(conv(sd, f"up_blocks.{i}.upsamplers.0.conv", h))
    h = q(F.silu(group_norm(sd, "conv_norm_out", h, G, eps)))
    return q(conv(sd, "conv_out", h))


def controlnet_cond_embedding(sd, cfg, cond: Tensor, q=_id) -> Tensor:
    p = "controlnet_cond_embedding"
    h = q(F.silu(conv(sd, p + ".conv_in", cond)))
    n = len(cfg["conditioning_embedding_out_channels"]) - 1
    for i in range(n):
        h = q(F.silu(conv(sd, f"{p}.blocks.{2 * i}", h)))
        h = q(F.silu(conv(sd, f"{p}.blocks.{2 * i + 1}", h, stride=2)))
    return q(conv(sd, p + ".conv_out", h))


def controlnet_forward(sd, cfg, sample: Tensor, t: Tensor, ctx: Tensor, cond: Tensor,
                       conditioning_scale: float = 1.0, q=_id, added=None) -> Tuple[List[Tensor], Tensor]:
    """``ControlNetModel.forward(..., return_dict=False)`` -> (12 down residuals, mid residual)
    (call site diffusion/train_controlnet_genima.py:1368-1374)."""
    if t.dim() == 0:
        t = t[None].expand(sample.shape[0])
    emb = time_embed(sd, cfg, t, q, added)
    h = q(conv(sd, "conv_in", sample) + controlnet_cond_embedding(sd, cfg, cond, q))
    h, skips = _encoder(sd, cfg, h, emb, ctx, q)
    h = _mid(sd, cfg, h, emb, ctx, q)
    outs = [q(conv(sd, f"controlnet_down_blocks.{i}", s, padding=0) * conditioning_scale) for i, s in enumerate(skips)]
    mid = q(conv(sd, "controlnet_mid_block", h, padding=0) * conditioning_scale)
    return outs, mid


# ----------------------------------------------------------------------------- AutoencoderKL
def _vae_attention(sd, p, x, groups, q=_id):
    """diffusers VAE mid-block ``Attention`` (1 head, residual_connection=True, GroupNorm eps 1e-6)."""
    B, C, H, W = x.shape
    h = q(group_norm(sd, p + ".group_norm", x, groups, 1e-6))
    h = h.view(B, C, H * W).transpose(1, 2)
    a = q(attention(q(linear(sd, p + ".to_q", h)), q(linear(sd, p + ".to_k", h)), q(linear(sd, p + ".to_v", h)), 1))
    a = linear(sd, p + ".to_out.0", a)
    return q(a.transpose(1, 2).reshape(B, C, H, W) + x)


def _vae_mid(sd, p, h, G, q=_id):
    h = resnet_block(sd, p + ".resnets.0", h, None, G, 1e-6, q)
    h = _vae_attention(sd, p + ".attentions.0", h, G, q)
    return resnet_block(sd, p + ".resnets.1", h, None, G, 1e-6, q)


def vae_decode(sd, cfg, z: Tensor, q=_id) -> Tensor:
    """``AutoencoderKL.decode(z).sample``; caller passes ``latents / scaling_factor``."""
    G = cfg["norm_num_groups"]
    n = len(cfg["block_out_channels"])
    h = q(conv(sd, "post_quant_conv", z, padding=0))
    h = q(conv(sd, "decoder.conv_in", h))
    h = _vae_mid(sd, "decoder.mid_block", h, G, q)
    for i in range(n):
        for j in range(cfg["layers_per_block"] + 1):
            h = resnet_block(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, None, G, 1e-6, q)
        if i != n - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = q(conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", h))
    h = q(F.silu(group_norm(sd, "decoder.conv_norm_out", h, G, 1e-6)))
    return conv(sd, "decoder.conv_out", h)


def vae_encode_moments(sd, cfg, x: Tensor, q=_id) -> Tuple[Tensor, Tensor]:
    """``AutoencoderKL.encode(x).latent_dist`` -> (mean, logvar clamped to [-30, 20])
    (call site diffusion/train_controlnet_genima.py:1329-1332)."""
    G = cfg["norm_num_groups"]
    n = len(cfg["block_out_channels"])
    h = q(conv(sd, "encoder.conv_in", x))
    for i in range(n):
        for j in range(cfg["layers_per_block"]):
            h = resnet_block(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, None, G, 1e-6, q)
        if i != n - 1:
            h = F.pad(h, (0, 1, 0, 1))
            h = q(conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=0))
    h = _vae_mid(sd, "encoder.mid_block", h, G, q)
    h = q(F.silu(group_norm(sd, "encoder.conv_norm_out", h, G, 1e-6)))
    h = q(conv(sd, "encoder.conv_out", h))
    m = conv(sd, "quant_conv", h, padding=0)
    mean, logvar = m.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def _tiny_block(sd, p, x, q=_id):
    h = q(F.relu(conv(sd, p + ".conv.0", x)))
    h = q(F.relu(conv(sd, p + ".conv.2", h)))
    sk = F.conv2d(x, _w(sd, p + ".skip")) if (p + ".skip.weight") in sd else x
    return q(F.relu(conv(sd, p + ".conv.4", h) + sk))


def taesd_decode(sd, cfg, z: Tensor, q=_id) -> Tensor:
    """diffusers 0.29 ``AutoencoderTiny.decode`` = ``DecoderTiny.forward`` (models/autoencoders/vae.py): clamp by 3 tanh(x / 3), the
    ``nn.Sequential`` of convs / AutoencoderTinyBlock s / nearest-2x upsamplers, then ``x.mul(2).sub(1)``."""
    nb = cfg["num_decoder_blocks"]
    h = q(torch.tanh(z / 3.0) * 3.0)
    h = q(F.relu(conv(sd, "decoder.layers.0", h)))
    idx = 2
    for i, n in enumerate(nb):
        final = i == len(nb) - 1
        for _ in range(n):
            h = _tiny_block(sd, f"decoder.layers.{idx}", h, q)
            idx += 1
        if not final:
            idx += 1
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = q(F.conv2d(h, _w(sd, f"decoder.layers.{idx}"), None, padding=1))
        else:
            h = conv(sd, f"decoder.layers.{idx}", h)
        idx += 1
    return q(h * 2.0 - 1.0)


def vae_postprocess_u8(img: Tensor) -> Tensor:
    """``VaeImageProcessor.postprocess(output_type="pil")`` numerics: NCHW float -> NHWC uint8
    (SURVEY Appendix D.6).  ``round`` is round-half-to-even like numpy's."""
    x = (img / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float()
    return (x * 255).round().to(torch.uint8)


# ----------------------------------------------------------------------------- CLIP text tower
def clip_text_forward(sd, cfg, ids: Tensor, q=_id, hidden=None) -> Tensor:
    """transformers ``CLIPTextModel(ids)[0]`` = last_hidden_state after final_layer_norm
    (call site diffusion/train_controlnet_genima.py:1362)."""
    B, L = ids.shape
    heads = cfg["num_attention_heads"]
    eps = cfg.get("layer_norm_eps", 1e-5)
    x = sd["text_model.embeddings.token_embedding.weight"][ids] + \
        sd["text_model.embeddings.position_embedding.weight"][:L][None]
    x = q(x)
    for i in range(cfg["num_hidden_layers"]):
        p = f"text_model.encoder.layers.{i}"
        n = q(layer_norm(sd, p + ".layer_norm1", x, eps))
        a = q(attention(q(linear(sd, p + ".self_attn.q_proj", n)), q(linear(sd, p + ".self_attn.k_proj", n)),
                        q(linear(sd, p + ".self_attn.v_proj", n)), heads, causal=True))
        x = q(x + linear(sd, p + ".self_attn.out_proj", a))
        n = q(layer_norm(sd, p + ".layer_norm2", x, eps))
        h = linear(sd, p + ".mlp.fc1", n)
        if cfg["hidden_act"] == "quick_gelu":
            h = h * torch.sigmoid(1.702 * h)
        else:
            h = F.gelu(h)
        x = q(x + linear(sd, p + ".mlp.fc2", q(h)))
        if hidden is not None:
            hidden.append(x)
    return q(layer_norm(sd, "text_model.final_layer_norm", x, eps))


def clip_text_penultimate_and_pooled(sd, cfg, ids: Tensor, q=_id):
    """SDXL ``encode_prompt`` (diffusion/train_controlnet_sdxl_genima.py:854-893): ``text_encoder(ids, output_hidden_states=True)``
    -> (hidden_states[-2] = the input of the last encoder layer, no final LayerNorm;  for CLIPTextModelWithProjection also
    ``text_embeds`` = text_projection(final_layer_norm(last)[eot]), else None)."""
    hidden = []
    last = clip_text_forward(sd, cfg, ids, q, hidden)
    pooled = None
    if "text_projection.weight" in sd:
        eot = ids.argmax(dim=-1)
        pooled = q(last[torch.arange(last.shape[0]), eot] @ sd["text_projection.weight"].t())
    return hidden[-2], pooled


def clip_text_pooled_projection(sd, cfg, ids: Tensor, q=_id) -> Tensor:
    """openai-CLIP ``encode_text`` tail as run by hand in controller/method/genima_act.py:337-343:
    take the row at argmax(token id) (EOT) and multiply by ``text_projection``."""
    x = clip_text_forward(sd, cfg, ids, q)
    eot = ids.argmax(dim=-1)
    pooled = x[torch.arange(x.shape[0]), eot]
    return pooled @ sd["text_projection.weight"].t()
