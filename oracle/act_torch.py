"""CPU oracle: fp32 PyTorch restatement of the ACT controller forward the reference runs through RoboBase.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the arithmetic lives in RoboBase (unpinned git
dependency, README.md:40-46; call sites controller/method/genima_act.py:2-18, :27-92, :165-214) which is absent from
/root/reference and this image, and the reference has no tests.  This restates the public ACT / DETR semantics RoboBase vendors
(SURVEY.md Appendix E): ResNet-18 with FrozenBatchNorm per view -> 1x1 input_proj -> views concatenated along width -> sine
positional embedding -> post-norm DETR encoder(4)/decoder(6) over [latent, proprio, task] + image tokens with 20 learned queries
-> action / is_pad heads; plus the reference-owned pieces: ImageNet normalisation (genima_act.py:146-148, :188), the 2-layer
state MLP (:237-241), z = 0 prior at inference (:70-75).

Pinned: ResNet-18 + FrozenBN, the sine positions, the DETR encoder / decoder stack and the heads equal transformers' ``ResNetModel`` /
``DetrSinePositionEmbedding`` / ``DetrEncoderLayer`` / ``DetrDecoderLayer`` to 1e-6 (tests/golden/act_golden.npz, tests/test_golden_cpu.py).
UNPINNED ([VERIFY] against a real latest.pt): language conditioning -- FiLM after bn1 in the BasicBlocks of layer2..4 and a projected task
token, restated from the MT-ACT / RoboAgent lineage -- and the frame-stack ``projection_layer``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
_id = lambda t: t  # noqa: E731


def frozen_bn(sd, p, x, eps=1e-5):
    scale = sd[p + ".weight"] * (sd[p + ".running_var"] + eps).rsqrt()
    shift = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
    return x * scale[None, :, None, None] + shift[None, :, None, None]


def basic_block(sd, p, x, stride, q=_id, film=None):
    """torchvision BasicBlock with FrozenBatchNorm; ``film`` = (gamma, beta) [N, planes]: the MT-ACT ``resnet_film`` block applies
    out = (1 + gamma) * out + beta after bn1, before the ReLU."""
    idt = x
    h = frozen_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1))
    if film is not None:
        h = (1.0 + film[0][:, :, None, None]) * q(h) + film[1][:, :, None, None]
    h = q(F.relu(h))
    h = frozen_bn(sd, p + ".bn2", F.conv2d(h, sd[p + ".conv2.weight"], None, 1, 1))
    if (p + ".downsample.0.weight") in sd:
        idt = q(frozen_bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0)))
    return q(F.relu(h + idt))


def resnet18_features(sd, x, q=_id, task_emb=None, per_sample: int = 1):
    """``task_emb`` [B, lang_dim] with x = [B * per_sample, 3, H, W]: FiLM features per conditioned layer (layer2..4) come from
    ``backbone.film_fcs.j`` and are laid out [B, 2 (gamma, beta), blocks, planes] (MT-ACT ``_extract_film_features_for_layer``)."""
    p = "backbone"
    h = q(F.relu(frozen_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, 2, 3))))
    h = F.max_pool2d(h, 3, 2, 1)
    for j, (li, stride) in enumerate(((1, 1), (2, 2), (3, 2), (4, 2))):
        films = [None, None]
        if task_emb is not None and li >= 2 and f"{p}.film_fcs.{li - 2}.weight" in sd:
            ff = q(F.linear(task_emb, sd[f"{p}.film_fcs.{li - 2}.weight"], sd[f"{p}.film_fcs.{li - 2}.bias"]))
            ff = ff.view(ff.shape[0], 2, 2, -1).repeat_interleave(per_sample, dim=0)
            films = [(ff[:, 0, b], ff[:, 1, b]) for b in range(2)]
        h = basic_block(sd, f"{p}.layer{li}.0", h, stride, q, films[0])
        h = basic_block(sd, f"{p}.layer{li}.1", h, 1, q, films[1])
    return h


def sine_pos_embed(H, W, d, temperature=10000.0) -> torch.Tensor:
    """DETR PositionEmbeddingSine(normalize=True, scale=2*pi) for one camera's HxW map -> [d, H, W]."""
    npf = d // 2
    eps, scale = 1e-6, 2 * math.pi
    y = torch.arange(1, H + 1, dtype=torch.float32)[:, None].expand(H, W)
    x = torch.arange(1, W + 1, dtype=torch.float32)[None, :].expand(H, W)
    y = y / (H + eps) * scale
    x = x / (W + eps) * scale
    dim_t = temperature ** (2 * (torch.arange(npf, dtype=torch.float32) // 2) / npf)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).permute(2, 0, 1)


def mha(sd, p, q_in, k_in, v_in, heads, q=_id):
    """nn.MultiheadAttention (batch-first tensors [B, N, d]) with packed in_proj."""
    d = q_in.shape[-1]
    W, b = sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"]
    qq = q(F.linear(q_in, W[:d], b[:d]))
    kk = q(F.linear(k_in, W[d:2 * d], b[d:2 * d]))
    vv = q(F.linear(v_in, W[2 * d:], b[2 * d:]))
    B, Nq, _ = qq.shape
    hd = d // heads
    qh, kh, vh = (t.view(B, -1, heads, hd).transpose(1, 2) for t in (qq, kk, vv))
    a = torch.softmax(qh @ kh.transpose(-1, -2) * hd ** -0.5, dim=-1) @ vh
    a = q(a.transpose(1, 2).reshape(B, Nq, d))
    return F.linear(a, sd[p + ".out_proj.weight"], sd[p + ".out_proj.bias"])


def ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _nodrop(name, x):
    return x


def act_forward(sd, cfg, images_u8: torch.Tensor, qpos: torch.Tensor, task_emb: torch.Tensor = None, q=_id, latent_z: torch.Tensor = None,
                drop=_nodrop, images_float: torch.Tensor = None):
    """images_u8: [B, V, 3, H, W] values 0..255; qpos [B, state_dim]; task_emb [B, lang_dim] or None.
    -> (a_hat [B, num_queries, action_dim], is_pad_hat [B, num_queries, 1]).
    Training hooks: ``latent_z`` [B, latent_dim] replaces the z = 0 prior with the CVAE posterior sample (genima_act.py:57-68);
    ``drop(name, x)`` applies a dropout mask at the sites the device path supports (state MLP p = 0.3, genima_act.py:237-241; the DETR
    layers' residual / feed-forward dropouts p = 0.1 -- NOT nn.MultiheadAttention's attention-probability dropout);
    ``images_float`` = already augmented images on the 0..255 scale (then ``images_u8`` only gives the shape)."""
    B, V = images_u8.shape[:2]
    d, heads = cfg["hidden_dim"], cfg["nheads"]
    mean = torch.tensor(IMAGENET_MEAN)[None, :, None, None]
    std = torch.tensor(IMAGENET_STD)[None, :, None, None]
    src_img = images_float if images_float is not None else images_u8.float()
    x = q((src_img.flatten(0, 1) / 255.0 - mean) / std)
    lang = cfg.get("use_lang_cond") and task_emb is not None
    f = resnet18_features(sd, x, q, task_emb if lang else None, V)          # [B*V, 512, h, w]
    f = q(F.conv2d(f, sd["input_proj.weight"], sd["input_proj.bias"]))     # [B*V, d, h, w]
    h, w = f.shape[-2:]
    fs = cfg.get("frame_stack", 1)
    if fs > 1:  # image index = camera * fs + frame; the frames of a view are stacked on channels, then projection_layer (1x1)
        V //= fs
        f = f.view(B * V, fs * d, h, w)
        f = q(F.conv2d(f, sd["projection_layer.weight"], sd["projection_layer.bias"]))
    f = f.view(B, V, d, h, w).permute(0, 2, 3, 1, 4).reshape(B, d, h, V * w)   # views along width
    pos = sine_pos_embed(h, w, d).repeat(1, 1, V)                           # [d, h, V*w]
    src = f.flatten(2).transpose(1, 2)                                      # [B, N, d]
    pos = q(pos.flatten(1).t())[None].expand(B, -1, -1)
    proprio = F.linear(q(drop("state", F.linear(qpos, sd["input_proj_robot_state.0.weight"], sd["input_proj_robot_state.0.bias"]))),
                       sd["input_proj_robot_state.2.weight"], sd["input_proj_robot_state.2.bias"])
    z0 = torch.zeros(B, cfg["latent_dim"]) if latent_z is None else latent_z
    latent = F.linear(z0, sd["latent_out_proj.weight"], sd["latent_out_proj.bias"])
    extra = [latent, proprio]
    if cfg.get("use_lang_cond") and task_emb is not None:
        extra.append(F.linear(task_emb, sd["task_proj.weight"], sd["task_proj.bias"]))
    n_extra = len(extra)
    src = q(torch.cat([torch.stack(extra, dim=1), src], dim=1))
    pos = q(torch.cat([sd["additional_pos_embed.weight"][:n_extra][None].expand(B, -1, -1), pos], dim=1))
    for i in range(cfg["enc_layers"]):
        p = f"transformer.encoder.layers.{i}"
        qk = q(src + pos)
        src = q(ln(sd, p + ".norm1", src + drop(p + ".d1", mha(sd, p + ".self_attn", qk, qk, src, heads, q))))
        ff = F.linear(q(drop(p + ".df", F.relu(F.linear(src, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])))), sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
        src = q(ln(sd, p + ".norm2", src + drop(p + ".d2", ff)))
    memory = src
    mem_pos = q(memory + pos)
    nq = cfg["num_queries"]
    qpos_e = q(sd["query_embed.weight"])[None].expand(B, -1, -1)
    tgt = torch.zeros(B, nq, d)
    for i in range(cfg["dec_layers"]):
        p = f"transformer.decoder.layers.{i}"
        qk = q(tgt + qpos_e)
        tgt = q(ln(sd, p + ".norm1", tgt + drop(p + ".d1", mha(sd, p + ".self_attn", qk, qk, tgt, heads, q))))
        tgt = q(ln(sd, p + ".norm2", tgt + drop(p + ".d2", mha(sd, p + ".multihead_attn", q(tgt + qpos_e), mem_pos, memory, heads, q))))
        ff = F.linear(q(drop(p + ".df", F.relu(F.linear(tgt, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])))), sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
        tgt = q(ln(sd, p + ".norm3", tgt + drop(p + ".d3", ff)))
    hs = q(ln(sd, "transformer.decoder.norm", tgt))
    a_hat = F.linear(hs, sd["action_head.weight"], sd["action_head.bias"])
    is_pad = F.linear(hs, sd["is_pad_head.weight"], sd["is_pad_head.bias"])
    return a_hat, is_pad
