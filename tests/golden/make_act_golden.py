"""Generate tests/golden/act_golden.npz: the ACT controller forward assembled from INDEPENDENT implementations of its parts -- the
installed ``transformers`` ``ResNetModel`` (basic layers, BatchNorm in eval mode = FrozenBatchNorm), ``DetrSinePositionEmbedding``,
``DetrEncoderLayer`` / ``DetrDecoderLayer`` (post-norm; positions added to q / k only; cross-attention keys = memory + pos) -- around
the reference-owned glue (ImageNet normalise, 2-layer state MLP, z = 0 prior, [latent, proprio] ++ image tokens with the views along
the width: controller/method/genima_act.py:27-92, :146-148, :165-214, :233-241).  RoboBase (the reference's implementation of this
path, an unpinned git dependency) is absent; it vendors the public DETR / ACT modules these HF classes also implement.
Language conditioning (FiLM + task token, [VERIFY] items of SURVEY.md Appendix E) is outside this pin: ``use_lang_cond`` is off here.
Run in the build container:    python tests/golden/make_act_golden.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import transformers.models.detr.modeling_detr as D  # noqa: E402
from transformers import DetrConfig, ResNetConfig, ResNetModel  # noqa: E402

from genima_amd import configs  # noqa: E402
from genima_amd.act import IMAGENET_MEAN, IMAGENET_STD, act_schema  # noqa: E402
from genima_amd import weights  # noqa: E402

cfg = dict(configs.TINY_ACT_POLICY, use_lang_cond=False)
d, heads = cfg["hidden_dim"], cfg["nheads"]
sd = weights.synth_state_dict(act_schema(cfg), seed=41)

# ---- ResNet-18 ----------------------------------------------------------------------------------------------------------------------
rc = ResNetConfig(embedding_size=64, hidden_sizes=[64, 128, 256, 512], depths=[2, 2, 2, 2], layer_type="basic", hidden_act="relu",
                  downsample_in_first_stage=False)
resnet = ResNetModel(rc).eval()
rm = {}


def bn_map(dst, src):
    for n in ("weight", "bias", "running_mean", "running_var"):
        rm[f"{dst}.{n}"] = sd[f"{src}.{n}"]


rm["embedder.embedder.convolution.weight"] = sd["backbone.conv1.weight"]
bn_map("embedder.embedder.normalization", "backbone.bn1")
for li in range(1, 5):
    for bi in range(2):
        s, t = f"backbone.layer{li}.{bi}", f"encoder.stages.{li - 1}.layers.{bi}"
        for k in (1, 2):
            rm[f"{t}.layer.{k - 1}.convolution.weight"] = sd[f"{s}.conv{k}.weight"]
            bn_map(f"{t}.layer.{k - 1}.normalization", f"{s}.bn{k}")
        if f"{s}.downsample.0.weight" in sd:
            rm[f"{t}.shortcut.convolution.weight"] = sd[f"{s}.downsample.0.weight"]
            bn_map(f"{t}.shortcut.normalization", f"{s}.downsample.1")
missing, unexpected = resnet.load_state_dict(rm, strict=False)
assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)

# ---- DETR layers --------------------------------------------------------------------------------------------------------------------
dc = DetrConfig(d_model=d, encoder_attention_heads=heads, decoder_attention_heads=heads, encoder_ffn_dim=cfg["dim_feedforward"],
                decoder_ffn_dim=cfg["dim_feedforward"], dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                activation_function="relu", use_timm_backbone=False, backbone_config=rc, backbone=None, use_pretrained_backbone=False)
dc._attn_implementation = "eager"


def attn_map(m, dst, src):
    W, b = sd[src + ".in_proj_weight"], sd[src + ".in_proj_bias"]
    for i, n in enumerate(("q_proj", "k_proj", "v_proj")):
        m[f"{dst}.{n}.weight"], m[f"{dst}.{n}.bias"] = W[i * d:(i + 1) * d], b[i * d:(i + 1) * d]
    m[f"{dst}.o_proj.weight"], m[f"{dst}.o_proj.bias"] = sd[src + ".out_proj.weight"], sd[src + ".out_proj.bias"]


def pair(m, dst, src):
    m[dst + ".weight"], m[dst + ".bias"] = sd[src + ".weight"], sd[src + ".bias"]


enc_layers, dec_layers = [], []
for i in range(cfg["enc_layers"]):
    p, m = f"transformer.encoder.layers.{i}", {}
    attn_map(m, "self_attn", p + ".self_attn")
    pair(m, "self_attn_layer_norm", p + ".norm1"); pair(m, "mlp.fc1", p + ".linear1"); pair(m, "mlp.fc2", p + ".linear2")
    pair(m, "final_layer_norm", p + ".norm2")
    layer = D.DetrEncoderLayer(dc).eval()
    layer.load_state_dict(m, strict=True)
    enc_layers.append(layer)
for i in range(cfg["dec_layers"]):
    p, m = f"transformer.decoder.layers.{i}", {}
    attn_map(m, "self_attn", p + ".self_attn"); attn_map(m, "encoder_attn", p + ".multihead_attn")
    pair(m, "self_attn_layer_norm", p + ".norm1"); pair(m, "encoder_attn_layer_norm", p + ".norm2")
    pair(m, "mlp.fc1", p + ".linear1"); pair(m, "mlp.fc2", p + ".linear2"); pair(m, "final_layer_norm", p + ".norm3")
    layer = D.DetrDecoderLayer(dc).eval()
    layer.load_state_dict(m, strict=True)
    dec_layers.append(layer)
dec_norm = nn.LayerNorm(d)
dec_norm.load_state_dict({"weight": sd["transformer.decoder.norm.weight"], "bias": sd["transformer.decoder.norm.bias"]})

# ---- inputs -------------------------------------------------------------------------------------------------------------------------
B, V, S = 2, cfg["num_views"], cfg["image_size"]
g = torch.Generator().manual_seed(9)
images = torch.randint(0, 256, (B, V, 3, S, S), generator=g, dtype=torch.uint8)
qpos = torch.randn(B, cfg["state_dim"], generator=g)

with torch.no_grad():
    mean = torch.tensor(IMAGENET_MEAN)[None, :, None, None]
    std = torch.tensor(IMAGENET_STD)[None, :, None, None]
    x = (images.float().flatten(0, 1) / 255.0 - mean) / std
    feat = resnet(pixel_values=x).last_hidden_state                                       # [B*V, 512, h, w]
    f = F.conv2d(feat, sd["input_proj.weight"], sd["input_proj.bias"])
    h, w = f.shape[-2:]
    try:
        pe = D.DetrSinePositionEmbedding(num_position_features=d // 2, normalize=True)
        pos_cam = pe(shape=(1, d, h, w), device=f.device, dtype=f.dtype, mask=torch.ones(1, h, w, dtype=torch.bool))
    except TypeError:
        pe = D.DetrSinePositionEmbedding(d // 2, normalize=True)
        pos_cam = pe(f[:1], torch.ones(1, h, w, dtype=torch.long))
    pos_cam = pos_cam.reshape(1, d, h, w)[0] if pos_cam.dim() != 3 or pos_cam.shape[0] != d else pos_cam
    fv = f.view(B, V, d, h, w).permute(0, 2, 3, 1, 4).reshape(B, d, h, V * w)               # views along the width
    pos_img = pos_cam.reshape(d, h, w).repeat(1, 1, V)
    src_img = fv.flatten(2).transpose(1, 2)
    pos_tok = pos_img.flatten(1).t()[None].expand(B, -1, -1)
    proprio = F.linear(F.linear(qpos, sd["input_proj_robot_state.0.weight"], sd["input_proj_robot_state.0.bias"]),
                       sd["input_proj_robot_state.2.weight"], sd["input_proj_robot_state.2.bias"])
    latent = F.linear(torch.zeros(B, cfg["latent_dim"]), sd["latent_out_proj.weight"], sd["latent_out_proj.bias"])
    src = torch.cat([latent[:, None], proprio[:, None], src_img], dim=1)
    pos = torch.cat([sd["additional_pos_embed.weight"][:2][None].expand(B, -1, -1), pos_tok], dim=1)
    for layer in enc_layers:
        src = layer(src, None, spatial_position_embeddings=pos)
    qe = sd["query_embed.weight"][None].expand(B, -1, -1)
    tgt = torch.zeros(B, cfg["num_queries"], d)
    for layer in dec_layers:
        tgt = layer(tgt, None, spatial_position_embeddings=pos, object_queries_position_embeddings=qe, encoder_hidden_states=src)
    hs = dec_norm(tgt)
    a_hat = F.linear(hs, sd["action_head.weight"], sd["action_head.bias"])
    is_pad = F.linear(hs, sd["is_pad_head.weight"], sd["is_pad_head.bias"])

out = {"images": images.numpy(), "qpos": qpos.numpy(), "resnet_features": feat.numpy().astype(np.float32),
       "pos_cam": pos_cam.reshape(d, h, w).numpy().astype(np.float32), "memory": src.numpy().astype(np.float32),
       "a_hat": a_hat.numpy().astype(np.float32), "is_pad_hat": is_pad.numpy().astype(np.float32), "seed": np.array(41)}
np.savez_compressed(os.path.join(HERE, "act_golden.npz"), **out)
print({k: v.shape for k, v in out.items()}, float(a_hat.abs().max()))
