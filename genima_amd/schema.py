"""State-dict schemas (parameter name -> shape) for the architectures on the Genima hot path.

Names and shapes follow the diffusers 0.29.0 / transformers 4.38.0 checkpoint layout the reference
loads (conv weights OIHW, Linear ``[out, in]``) so that a real ``diffusion_pytorch_model.safetensors``
maps 1:1 onto these keys (reference call sites: controller/agent/sd_controlnet_agent.py:32-42,
diffusion/train_controlnet_genima.py:1042-1071).  The schema *is* the architecture definition used by
both the HIP host classes (weight packing) and the tests (parameter-count pins: UNet 865.9 M,
ControlNet 364.2 M, VAE 83.7 M, CLIP-H text 340.4 M).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

Shape = Tuple[int, ...]
Schema = "OrderedDict[str, Shape]"


def _conv(s, name, cin, cout, k, bias=True):
    s[name + ".weight"] = (cout, cin, k, k)
    if bias:
        s[name + ".bias"] = (cout,)


def _linear(s, name, cin, cout, bias=True):
    s[name + ".weight"] = (cout, cin)
    if bias:
        s[name + ".bias"] = (cout,)


def _norm(s, name, c):
    s[name + ".weight"] = (c,)
    s[name + ".bias"] = (c,)


def _resnet(s, p, cin, cout, temb_dim):
    _norm(s, p + ".norm1", cin)
    _conv(s, p + ".conv1", cin, cout, 3)
    if temb_dim:
        _linear(s, p + ".time_emb_proj", temb_dim, cout)
    _norm(s, p + ".norm2", cout)
    _conv(s, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(s, p + ".conv_shortcut", cin, cout, 1)


def _transformer2d(s, p, c, ctx_dim, n_layers=1):
    _norm(s, p + ".norm", c)
    _linear(s, p + ".proj_in", c, c)
    for k in range(n_layers):
        b = f"{p}.transformer_blocks.{k}"
        _norm(s, b + ".norm1", c)
        for nm in ("to_q", "to_k", "to_v"):
            _linear(s, f"{b}.attn1.{nm}", c, c, bias=False)
        _linear(s, b + ".attn1.to_out.0", c, c)
        _norm(s, b + ".norm2", c)
        _linear(s, b + ".attn2.to_q", c, c, bias=False)
        _linear(s, b + ".attn2.to_k", ctx_dim, c, bias=False)
        _linear(s, b + ".attn2.to_v", ctx_dim, c, bias=False)
        _linear(s, b + ".attn2.to_out.0", c, c)
        _norm(s, b + ".norm3", c)
        _linear(s, b + ".ff.net.0.proj", c, 8 * c)
        _linear(s, b + ".ff.net.2", 4 * c, c)
    _linear(s, p + ".proj_out", c, c)


def transformer_layers(cfg, level: int) -> int:
    """``transformer_layers_per_block``: an int (SD-2.x: 1) or one entry per resolution level (SDXL: 1, 2, 10)."""
    t = cfg.get("transformer_layers_per_block", 1)
    return t[level] if isinstance(t, (list, tuple)) else t


def _unet_encoder(s, cfg):
    """conv_in + time (+ SDXL added-condition) embedding + down blocks + mid block (shared by UNet and ControlNet)."""
    boc = cfg["block_out_channels"]
    temb = boc[0] * 4
    ctx = cfg["cross_attention_dim"]
    _conv(s, "conv_in", cfg["in_channels"], boc[0], 3)
    _linear(s, "time_embedding.linear_1", boc[0], temb)
    _linear(s, "time_embedding.linear_2", temb, temb)
    if cfg.get("addition_embed_type") == "text_time":  # SDXL micro-conditioning
        _linear(s, "add_embedding.linear_1", cfg["projection_class_embeddings_input_dim"], temb)
        _linear(s, "add_embedding.linear_2", temb, temb)
    cout = boc[0]
    for i, btype in enumerate(cfg["down_block_types"]):
        cin, cout = cout, boc[i]
        for j in range(cfg["layers_per_block"]):
            _resnet(s, f"down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, temb)
            if btype == "CrossAttnDownBlock2D":
                _transformer2d(s, f"down_blocks.{i}.attentions.{j}", cout, ctx, transformer_layers(cfg, i))
        if i != len(boc) - 1:
            _conv(s, f"down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
    c = boc[-1]
    _resnet(s, "mid_block.resnets.0", c, c, temb)
    _transformer2d(s, "mid_block.attentions.0", c, ctx, transformer_layers(cfg, len(boc) - 1))
    _resnet(s, "mid_block.resnets.1", c, c, temb)


def unet_schema(cfg) -> "OrderedDict[str, Shape]":
    s: "OrderedDict[str, Shape]" = OrderedDict()
    _unet_encoder(s, cfg)
    boc = cfg["block_out_channels"]
    temb = boc[0] * 4
    ctx = cfg["cross_attention_dim"]
    rev = list(reversed(boc))
    n = cfg["layers_per_block"] + 1
    cout = rev[0]
    for i, btype in enumerate(cfg["up_block_types"]):
        prev, cout = cout, rev[i]
        cin = rev[min(i + 1, len(boc) - 1)]
        for j in range(n):
            skip = cin if j == n - 1 else cout
            rin = prev if j == 0 else cout
            _resnet(s, f"up_blocks.{i}.resnets.{j}", rin + skip, cout, temb)
            if btype == "CrossAttnUpBlock2D":
                _transformer2d(s, f"up_blocks.{i}.attentions.{j}", cout, ctx, transformer_layers(cfg, len(boc) - 1 - i))
        if i != len(boc) - 1:
            _conv(s, f"up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
    _norm(s, "conv_norm_out", boc[0])
    _conv(s, "conv_out", boc[0], cfg["out_channels"], 3)
    return s


def controlnet_skip_channels(cfg):
    """Channel count of each of the UNet skip tensors s0..s11."""
    boc = cfg["block_out_channels"]
    ch = [boc[0]]
    for i in range(len(boc)):
        ch += [boc[i]] * cfg["layers_per_block"]
        if i != len(boc) - 1:
            ch.append(boc[i])
    return ch


def controlnet_schema(cfg) -> "OrderedDict[str, Shape]":
    s: "OrderedDict[str, Shape]" = OrderedDict()
    _unet_encoder(s, cfg)
    boc = cfg["block_out_channels"]
    ce = cfg["conditioning_embedding_out_channels"]
    p = "controlnet_cond_embedding"
    _conv(s, p + ".conv_in", cfg["conditioning_channels"], ce[0], 3)
    for i in range(len(ce) - 1):
        _conv(s, f"{p}.blocks.{2 * i}", ce[i], ce[i], 3)
        _conv(s, f"{p}.blocks.{2 * i + 1}", ce[i], ce[i + 1], 3)  # stride 2
    _conv(s, p + ".conv_out", ce[-1], boc[0], 3)
    for i, c in enumerate(controlnet_skip_channels(cfg)):
        _conv(s, f"controlnet_down_blocks.{i}", c, c, 1)
    _conv(s, "controlnet_mid_block", boc[-1], boc[-1], 1)
    return s


def _vae_attn(s, p, c):
    _norm(s, p + ".group_norm", c)
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        _linear(s, f"{p}.{nm}", c, c)


def _vae_mid(s, p, c):
    _resnet(s, p + ".resnets.0", c, c, 0)
    _vae_attn(s, p + ".attentions.0", c)
    _resnet(s, p + ".resnets.1", c, c, 0)


def vae_schema(cfg, encoder=True, decoder=True) -> "OrderedDict[str, Shape]":
    s: "OrderedDict[str, Shape]" = OrderedDict()
    boc = cfg["block_out_channels"]
    lat = cfg["latent_channels"]
    L = cfg["layers_per_block"]
    if encoder:
        _conv(s, "encoder.conv_in", cfg["in_channels"], boc[0], 3)
        cout = boc[0]
        for i in range(len(boc)):
            cin, cout = cout, boc[i]
            for j in range(L):
                _resnet(s, f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, 0)
            if i != len(boc) - 1:
                _conv(s, f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
        _vae_mid(s, "encoder.mid_block", boc[-1])
        _norm(s, "encoder.conv_norm_out", boc[-1])
        _conv(s, "encoder.conv_out", boc[-1], 2 * lat, 3)
        _conv(s, "quant_conv", 2 * lat, 2 * lat, 1)
    if decoder:
        _conv(s, "post_quant_conv", lat, lat, 1)
        rev = list(reversed(boc))
        _conv(s, "decoder.conv_in", lat, rev[0], 3)
        _vae_mid(s, "decoder.mid_block", rev[0])
        cout = rev[0]
        for i in range(len(boc)):
            cin, cout = cout, rev[i]
            for j in range(L + 1):
                _resnet(s, f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, 0)
            if i != len(boc) - 1:
                _conv(s, f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
        _norm(s, "decoder.conv_norm_out", rev[-1])
        _conv(s, "decoder.conv_out", rev[-1], cfg["out_channels"], 3)
    return s


def _tiny_block(s, p, cin, cout):
    for k in (0, 2, 4):
        _conv(s, f"{p}.conv.{k}", cin if k == 0 else cout, cout, 3)
    if cin != cout:
        _conv(s, p + ".skip", cin, cout, 1, bias=False)


def taesd_schema(cfg, encoder=True, decoder=True) -> "OrderedDict[str, Shape]":
    """diffusers ``AutoencoderTiny`` (TAESD / TAESDXL; controller/agent/sd_controlnet_agent.py:45-49): ``nn.Sequential`` encoder /
    decoder whose entries are convs, ``AutoencoderTinyBlock`` s (conv.0 / conv.2 / conv.4 + optional 1x1 ``skip``) and parameter-free
    activations / upsamplers -- the sequential index of every entry is part of the key."""
    s: "OrderedDict[str, Shape]" = OrderedDict()
    lat = cfg["latent_channels"]
    if encoder:
        ch, nb = cfg["encoder_block_out_channels"], cfg["num_encoder_blocks"]
        idx, cin = 0, cfg["in_channels"]
        for i, n in enumerate(nb):
            if i == 0:
                _conv(s, f"encoder.layers.{idx}", cin, ch[i], 3)
            else:
                _conv(s, f"encoder.layers.{idx}", cin, ch[i], 3, bias=False)  # stride 2
            idx += 1
            for _ in range(n):
                _tiny_block(s, f"encoder.layers.{idx}", ch[i], ch[i])
                idx += 1
            cin = ch[i]
        _conv(s, f"encoder.layers.{idx}", cin, lat, 3)
    if decoder:
        ch, nb = cfg["decoder_block_out_channels"], cfg["num_decoder_blocks"]
        _conv(s, "decoder.layers.0", lat, ch[0], 3)
        idx = 2  # layers.1 is the activation
        for i, n in enumerate(nb):
            final = i == len(nb) - 1
            for _ in range(n):
                _tiny_block(s, f"decoder.layers.{idx}", ch[i], ch[i])
                idx += 1
            if not final:
                idx += 1  # nn.Upsample
            _conv(s, f"decoder.layers.{idx}", ch[i], cfg["out_channels"] if final else ch[i], 3, bias=final)
            idx += 1
    return s


def clip_text_schema(cfg) -> "OrderedDict[str, Shape]":
    s: "OrderedDict[str, Shape]" = OrderedDict()
    d, ff = cfg["hidden_size"], cfg["intermediate_size"]
    s["text_model.embeddings.token_embedding.weight"] = (cfg["vocab_size"], d)
    s["text_model.embeddings.position_embedding.weight"] = (cfg["max_position_embeddings"], d)
    for i in range(cfg["num_hidden_layers"]):
        p = f"text_model.encoder.layers.{i}"
        _norm(s, p + ".layer_norm1", d)
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            _linear(s, f"{p}.self_attn.{nm}", d, d)
        _norm(s, p + ".layer_norm2", d)
        _linear(s, p + ".mlp.fc1", d, ff)
        _linear(s, p + ".mlp.fc2", ff, d)
    _norm(s, "text_model.final_layer_norm", d)
    if cfg.get("projection_dim", 0):
        s["text_projection.weight"] = (cfg["projection_dim"], d)
    return s


def param_count(schema) -> int:
    n = 0
    for shp in schema.values():
        k = 1
        for d in shp:
            k *= d
        n += k
    return n
