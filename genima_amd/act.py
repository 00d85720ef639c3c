"""ACT controller forward on libgenima_hip.so (reference controller/method/genima_act.py).

``GenimaACT`` keeps the plugin surface the eval loop uses (controller/eval_genima.py:55-64, :91-103, :243-247):
``.act(obs_dict_of_tensors, step, eval_mode) -> Tensor[B, num_queries, action_dim]``, ``.state_dict()`` /
``.load_state_dict(sd, strict=False)``, ``.train(bool)``, ``.encode_clip_text(tokens)``.  The forward -- ImageNet normalise ->
ResNet-18 (FrozenBatchNorm folded into the conv weights at pack time, ReLU and the identity add fused into the conv epilogue)
per view -> 1x1 input_proj -> views along width -> sine positions -> DETR encoder(4)/decoder(6) on the same MFMA GEMM /
flash-attention (head dim 32) / LayerNorm kernels as the diffusion path -> heads -- runs entirely in HIP; RoboBase itself is an
unpinned absent dependency, so the module wiring follows the public ACT/DETR semantics restated in oracle/act_torch.py (pinned against
transformers' ResNet / DETR layers: tests/golden/make_act_golden.py).  Language conditioning (``use_lang_cond: true``,
controller/cfgs/method/genima_act.yaml:39; ``encoder_model(image, task_emb)``, genima_act.py:190) follows the MT-ACT / RoboAgent lineage
RoboBase's ``ImageEncoderACT`` comes from: FiLM (gamma, beta from the 512-d CLIP embedding) after bn1 of every BasicBlock of layer2..4,
plus the projected embedding as a third extra encoder token; ``frame_stack > 1`` stacks the frames of a view on channels in front of
``projection_layer`` (genima_act.py:191-197).  Key names of the FiLM generators / token projection are [VERIFY] items of SURVEY.md
Appendix E until a real ``latest.pt`` is available: ``robobase_key_map`` lists the aliases accepted.
"""
from __future__ import annotations

import math
import re
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import configs, graphs, packing, schema, weights
from ._lib import ACT_NONE, ACT_RELU, GenimaHipError
from .engine import Engine
from .host import FrozenConfig

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
_RESNET18 = ((1, 64, 1), (2, 128, 2), (3, 256, 2), (4, 512, 2))


def act_schema(cfg) -> "OrderedDict[str, tuple]":
    s: "OrderedDict[str, tuple]" = OrderedDict()
    d, ff = cfg["hidden_dim"], cfg["dim_feedforward"]

    def bn(p, c):
        for n in ("weight", "bias", "running_mean", "running_var"):
            s[f"{p}.{n}"] = (c,)

    s["backbone.conv1.weight"] = (64, 3, 7, 7)
    bn("backbone.bn1", 64)
    cin = 64
    for li, c, stride in _RESNET18:
        for bi in range(2):
            p = f"backbone.layer{li}.{bi}"
            s[p + ".conv1.weight"] = (c, cin if bi == 0 else c, 3, 3)
            bn(p + ".bn1", c)
            s[p + ".conv2.weight"] = (c, c, 3, 3)
            bn(p + ".bn2", c)
            if bi == 0 and (stride != 1 or cin != c):
                s[p + ".downsample.0.weight"] = (c, cin, 1, 1)
                bn(p + ".downsample.1", c)
        cin = c
    s["input_proj.weight"], s["input_proj.bias"] = (d, 512, 1, 1), (d,)
    if cfg.get("use_lang_cond"):
        # FiLM generators of the language-conditioned ResNet (MT-ACT / RoboAgent ``resnet_film``: ``film_config.use_in_layers = [1, 2, 3]``,
        # i.e. layer2..layer4): one Linear per conditioned layer, lang_dim -> 2 (gamma, beta) x 2 blocks x planes
        for j, (li, c, _) in enumerate(_RESNET18[1:]):
            s[f"backbone.film_fcs.{j}.weight"], s[f"backbone.film_fcs.{j}.bias"] = (2 * 2 * c, cfg["lang_dim"]), (2 * 2 * c,)
    if cfg.get("frame_stack", 1) > 1:  # ACTPolicy.projection_layer (controller/method/genima_act.py:193-197): fs * hidden -> hidden, 1x1
        s["projection_layer.weight"], s["projection_layer.bias"] = (d, d * cfg["frame_stack"], 1, 1), (d,)

    def mha(p):
        s[p + ".in_proj_weight"], s[p + ".in_proj_bias"] = (3 * d, d), (3 * d,)
        s[p + ".out_proj.weight"], s[p + ".out_proj.bias"] = (d, d), (d,)

    def ffn(p):
        s[p + ".linear1.weight"], s[p + ".linear1.bias"] = (ff, d), (ff,)
        s[p + ".linear2.weight"], s[p + ".linear2.bias"] = (d, ff), (d,)

    def ln(p):
        s[p + ".weight"], s[p + ".bias"] = (d,), (d,)

    for i in range(cfg["enc_layers"]):
        p = f"transformer.encoder.layers.{i}"
        mha(p + ".self_attn"); ffn(p); ln(p + ".norm1"); ln(p + ".norm2")
    for i in range(cfg["dec_layers"]):
        p = f"transformer.decoder.layers.{i}"
        mha(p + ".self_attn"); mha(p + ".multihead_attn"); ffn(p); ln(p + ".norm1"); ln(p + ".norm2"); ln(p + ".norm3")
    ln("transformer.decoder.norm")
    s["query_embed.weight"] = (cfg["num_queries"], d)
    s["additional_pos_embed.weight"] = (3, d)
    s["input_proj_robot_state.0.weight"], s["input_proj_robot_state.0.bias"] = (d, cfg["state_dim"]), (d,)
    s["input_proj_robot_state.2.weight"], s["input_proj_robot_state.2.bias"] = (d, d), (d,)
    s["latent_out_proj.weight"], s["latent_out_proj.bias"] = (d, cfg["latent_dim"]), (d,)
    s["task_proj.weight"], s["task_proj.bias"] = (d, cfg["lang_dim"]), (d,)
    s["action_head.weight"], s["action_head.bias"] = (cfg["action_dim"], d), (cfg["action_dim"],)
    s["is_pad_head.weight"], s["is_pad_head.bias"] = (1, d), (1,)
    return s


def sine_pos_embed(H, W, d, temperature=10000.0) -> torch.Tensor:
    """DETR PositionEmbeddingSine(normalize=True, scale=2*pi) of one camera's map -> [H, W, d] (host constant)."""
    npf, eps, scale = d // 2, 1e-6, 2 * math.pi
    y = torch.arange(1, H + 1, dtype=torch.float32)[:, None].expand(H, W) / (H + eps) * scale
    x = torch.arange(1, W + 1, dtype=torch.float32)[None, :].expand(H, W) / (W + eps) * scale
    dim_t = temperature ** (2 * (torch.arange(npf, dtype=torch.float32) // 2) / npf)
    px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2)


def pack_act(sd: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    """FrozenBatchNorm folded into conv weight/bias; MHA in_proj split into q|k (fused) and v; the rest through the generic packer."""
    f32 = {k: v.detach().float() for k, v in sd.items()}
    folded: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in f32.items():
        m = re.match(r"(backbone\..*?)(conv\d|downsample\.0)\.weight$", k)
        if m:
            bnp = m.group(1) + ("bn" + m.group(2)[-1] if m.group(2).startswith("conv") else "downsample.1")
            scale = f32[bnp + ".weight"] * (f32[bnp + ".running_var"] + 1e-5).rsqrt()
            folded[k] = v * scale[:, None, None, None]
            folded[k[: -len("weight")] + "bias"] = f32[bnp + ".bias"] - f32[bnp + ".running_mean"] * scale
        elif ".bn" in k or "downsample.1." in k:
            continue
        elif k.endswith("in_proj_weight"):
            p, d = k[: -len("in_proj_weight")], v.shape[1]
            folded[p + "qk_proj.weight"], folded[p + "v_proj.weight"] = v[: 2 * d], v[2 * d:]
            b = f32[p + "in_proj_bias"]
            folded[p + "qk_proj.bias"], folded[p + "v_proj.bias"] = b[: 2 * d], b[2 * d:]
            folded[p + "q_proj.weight"], folded[p + "q_proj.bias"] = v[:d], b[:d]
            folded[p + "k_proj.weight"], folded[p + "k_proj.bias"] = v[d: 2 * d], b[d: 2 * d]
        elif k.endswith("in_proj_bias"):
            continue
        else:
            folded[k] = v
    W = packing.pack_state_dict(folded, device)
    for head in ("action_head", "is_pad_head"):  # pad output rows to a multiple of 8 (N % 4 == 0 for the GEMM)
        w, b = folded[head + ".weight"], folded[head + ".bias"]
        n8 = (w.shape[0] + 7) // 8 * 8
        wp = torch.zeros(n8, w.shape[1]); wp[: w.shape[0]] = w
        bp = torch.zeros(n8); bp[: b.shape[0]] = b
        W[head + ".weight"], W[head + ".bias"] = wp.half().to(device), bp.half().to(device)
    return W


def emit_act_forward(E: Engine, W, Wclip, cfg, ccfg, img_u8_nhwc: torch.Tensor, qpos: torch.Tensor,
                     tokens: Optional[torch.Tensor]):
    """img_u8_nhwc: uint8 [B, V, H, W, 3]; qpos f16 [B, state_dim_pad8]; tokens int32 [B, 77] or None.
    Every step is an Engine op (recordable: ``GenimaACT`` replays the whole controller forward from C++); the only torch calls
    are build-time constants (sine positions, zero / broadcast buffers).  -> (a_hat [B, nq, 8-padded], is_pad [B, nq, 8-padded])."""
    B, V, H, Wd, _ = img_u8_nhwc.shape
    d, heads = cfg["hidden_dim"], cfg["nheads"]
    dev = E.device
    with E.scope("act"):
        # ---- language conditioning: CLIP text tower -> EOT row -> text_projection -> task token ---------------------------
        task = None
        if cfg.get("use_lang_cond") and tokens is not None:
            xt = graphs.emit_clip_text(E, Wclip, ccfg, tokens)
            eot = E.argmax_rows(tokens, name="eot")
            pooled = E.gather_rows(xt, eot, name="pooled")
            task = E.linear(pooled, Wclip["text_projection.weight"], name="task_emb")
        film = None
        if task is not None and "backbone.film_fcs.0.weight" in W:
            film = {li: E.linear(task, W[f"backbone.film_fcs.{j}.weight"], W[f"backbone.film_fcs.{j}.bias"], name=f"film{li}")
                    for j, (li, _, _) in enumerate(_RESNET18[1:])}
        # ---- ResNet-18 per view (FrozenBN folded; ReLU / identity add in the conv epilogues) ---------------------------------
        x = E.image_normalize_u8(img_u8_nhwc.view(B * V, H, Wd, 3), IMAGENET_MEAN, IMAGENET_STD, 8, name="img")
        p = "backbone"
        h = E.conv2d(x, W[p + ".conv1.weight"], W[p + ".conv1.bias"], ksize=7, stride=2, act=ACT_RELU, name="c1")
        h = E.maxpool3x3s2(h, name="mp")
        for li, c, stride in _RESNET18:
            for bi in range(2):
                q = f"{p}.layer{li}.{bi}"
                st = stride if bi == 0 else 1
                if film is not None and li >= 2:
                    # relu((1 + gamma) * bn1(conv1(x)) + beta): the block's FiLM slice of this layer's generator output
                    # (layout [B, 2, blocks = 2, planes], MT-ACT resnet_film.py) -- the same (gamma, beta) for every view / frame of a sample
                    y = E.conv2d(h, W[q + ".conv1.weight"], W[q + ".conv1.bias"], stride=st, name=q + ".c1")
                    ff = film[li]
                    y = E.film(y, ff[:, bi * c:(bi + 1) * c], ff[:, (2 + bi) * c:(3 + bi) * c], V * y.shape[1] * y.shape[2], ACT_RELU,
                               out=y)
                else:
                    y = E.conv2d(h, W[q + ".conv1.weight"], W[q + ".conv1.bias"], stride=st, act=ACT_RELU, name=q + ".c1")
                idt = h
                if (q + ".downsample.0.weight") in W:
                    idt = E.conv2d(h, W[q + ".downsample.0.weight"], W[q + ".downsample.0.bias"], ksize=1, stride=st,
                                   pad=(0, 0, 0, 0), name=q + ".ds")
                h = E.conv2d(y, W[q + ".conv2.weight"], W[q + ".conv2.bias"], residual=idt, act=ACT_RELU, residual_before_act=True,
                             name=q + ".c2")
        f = E.conv2d(h, W["input_proj.weight"], W["input_proj.bias"], ksize=1, pad=(0, 0, 0, 0), name="input_proj")  # [B*V, fh, fw, d]
        fh, fw = f.shape[1], f.shape[2]
        fs = cfg.get("frame_stack", 1)
        if fs > 1:
            # image index = camera * fs + frame (controller/eval_genima.py:167-173): the frames of a view go side by side on the CHANNEL
            # axis and ``projection_layer`` (1x1 conv) brings fs * hidden back to hidden (genima_act.py:191-197)
            V //= fs
            stk = E.buf("fs_stack", (B * V, fh, fw, fs * d))
            E.copy4d(f, stk, (B * V, fs, fh * fw, 1), (fs * fh * fw * d, fh * fw * d, d, 0), (fh * fw * fs * d, d, fs * d, 0), d)
            f = E.conv2d(stk, W["projection_layer.weight"], W["projection_layer.bias"], ksize=1, pad=(0, 0, 0, 0), name="projection_layer")
        # ---- encoder sequence [latent, proprio, (task), image tokens]; views concatenated along WIDTH ----------------------
        n_extra = 3 if task is not None else 2
        n_img = fh * V * fw
        N = n_extra + n_img
        src = E.buf("src", (B, N, d))
        zeros = E.buf("zero_latent", (B, W["latent_out_proj.weight"].shape[1]), zero=True)
        E.linear(zeros, W["latent_out_proj.weight"], W["latent_out_proj.bias"], out=src[:, 0])  # z = 0 prior (genima_act.py:70-75)
        pr = E.linear(qpos, W["input_proj_robot_state.0.weight"], W["input_proj_robot_state.0.bias"], name="state0")
        E.linear(pr, W["input_proj_robot_state.2.weight"], W["input_proj_robot_state.2.bias"], out=src[:, 1])
        if task is not None:
            E.linear(task, W["task_proj.weight"], W["task_proj.bias"], out=src[:, 2])
        # f[(b*V + v), y, x, :] -> src[b, n_extra + y*(V*fw) + v*fw + x, :]   (index space (b, v, y, x), runs of d)
        E.copy4d(f, src[:, n_extra:], (B, V, fh, fw), (V * fh * fw * d, fh * fw * d, fw * d, d),
                 (N * d, fw * d, V * fw * d, d), d)
        pos = E.buf("pos", (B, N, d))
        if not getattr(pos, "_gn_init", False):  # build-time constant
            pos_img = sine_pos_embed(fh, fw, d).repeat(1, V, 1).reshape(n_img, d)
            pos.copy_(torch.cat([W["additional_pos_embed.weight"][:n_extra].float().cpu(), pos_img], dim=0).half()[None].expand(B, -1, -1))
            pos._gn_init = True

        def self_attn(pfx, x_qk, x_v, n, nm):
            qk = E.linear(x_qk, W[pfx + ".qk_proj.weight"], W[pfx + ".qk_proj.bias"], name=nm + "qk")
            vt = E.linear(x_v, W[pfx + ".v_proj.weight"], W[pfx + ".v_proj.bias"], transposed_out=True, rows_per_batch=n,
                          pad_cols=(n + 63) // 64 * 64, name=nm + "vt")
            return E.attention(qk[:, :, :d], qk[:, :, d:], vt, heads, Nk=n, name=nm + "a")

        for i in range(cfg["enc_layers"]):
            q = f"transformer.encoder.layers.{i}"
            nm = f"e{i}."
            a = self_attn(q + ".self_attn", E.add(src, pos, name=nm + "xp"), src, N, nm)
            s1 = E.linear(a, W[q + ".self_attn.out_proj.weight"], W[q + ".self_attn.out_proj.bias"], residual=src, name=nm + "o")
            src = E.layernorm(s1, W[q + ".norm1.weight"], W[q + ".norm1.bias"], name=nm + "n1")
            ffh = E.linear(src, W[q + ".linear1.weight"], W[q + ".linear1.bias"], act=ACT_RELU, name=nm + "f1")
            s2 = E.linear(ffh, W[q + ".linear2.weight"], W[q + ".linear2.bias"], residual=src, name=nm + "f2")
            src = E.layernorm(s2, W[q + ".norm2.weight"], W[q + ".norm2.bias"], name=nm + "n2")
        memory = src
        mem_pos = E.add(memory, pos, name="mem_pos")
        nq = cfg["num_queries"]
        qe = E.buf("query_pos", (B, nq, d))
        if not getattr(qe, "_gn_init", False):
            qe.copy_(W["query_embed.weight"][None].expand(B, -1, -1))
            qe._gn_init = True
        tgt = E.buf("tgt0", (B, nq, d), zero=True)
        for i in range(cfg["dec_layers"]):
            q = f"transformer.decoder.layers.{i}"
            nm = f"d{i}."
            a = self_attn(q + ".self_attn", E.add(tgt, qe, name=nm + "tq"), tgt, nq, nm)
            t1 = E.linear(a, W[q + ".self_attn.out_proj.weight"], W[q + ".self_attn.out_proj.bias"], residual=tgt, name=nm + "o")
            tgt = E.layernorm(t1, W[q + ".norm1.weight"], W[q + ".norm1.bias"], name=nm + "n1")
            m = q + ".multihead_attn"
            cq = E.linear(E.add(tgt, qe, name=nm + "tq2"), W[m + ".q_proj.weight"], W[m + ".q_proj.bias"], name=nm + "cq")
            ck = E.linear(mem_pos, W[m + ".k_proj.weight"], W[m + ".k_proj.bias"], name=nm + "ck")
            cvt = E.linear(memory, W[m + ".v_proj.weight"], W[m + ".v_proj.bias"], transposed_out=True, rows_per_batch=N,
                           pad_cols=(N + 63) // 64 * 64, name=nm + "cvt")
            a = E.attention(cq, ck, cvt, heads, Nk=N, name=nm + "ca")
            t2 = E.linear(a, W[m + ".out_proj.weight"], W[m + ".out_proj.bias"], residual=tgt, name=nm + "co")
            tgt = E.layernorm(t2, W[q + ".norm2.weight"], W[q + ".norm2.bias"], name=nm + "n2")
            ffh = E.linear(tgt, W[q + ".linear1.weight"], W[q + ".linear1.bias"], act=ACT_RELU, name=nm + "f1")
            t3 = E.linear(ffh, W[q + ".linear2.weight"], W[q + ".linear2.bias"], residual=tgt, name=nm + "f2")
            tgt = E.layernorm(t3, W[q + ".norm3.weight"], W[q + ".norm3.bias"], name=nm + "n3")
        hs = E.layernorm(tgt, W["transformer.decoder.norm.weight"], W["transformer.decoder.norm.bias"], name="dec_norm")
        a_hat = E.linear(hs, W["action_head.weight"], W["action_head.bias"], name="a_hat")
        is_pad = E.linear(hs, W["is_pad_head.weight"], W["is_pad_head.bias"], name="is_pad")
        return a_hat, is_pad, task


_ROBOBASE_PREFIXES = (
    # RoboBase ``ActBCAgent`` registers the same modules more than once: ``actor`` (GenimaACTPolicy: .actor_model, .encoder_model),
    # ``actor_model`` and ``encoder`` at the top level -- ``ckpt["agent"]`` carries each weight under every path (eval_genima.py:91-103)
    ("actor.encoder_model.backbone.0.body.", "backbone."), ("encoder.backbone.0.body.", "backbone."),   # DETR Joiner: backbone[0].body = ResNet
    ("actor.encoder_model.backbone.0.", "backbone."), ("encoder.backbone.0.", "backbone."),
    ("actor.encoder_model.", ""), ("encoder.", ""), ("actor.actor_model.", ""), ("actor_model.", ""), ("actor.", ""),
)
_ROBOBASE_ALIASES = {  # [VERIFY] names: the MT-ACT lineage calls the token projection ``proj_text_emb``
    "proj_text_emb.weight": "task_proj.weight", "proj_text_emb.bias": "task_proj.bias",
    "task_emb_proj.weight": "task_proj.weight", "task_emb_proj.bias": "task_proj.bias",
}
_CVAE_PREFIXES = ("encoder.layers.", "cls_embed.", "encoder_action_proj.", "encoder_joint_proj.", "latent_proj.")  # training-only weights


def robobase_key_map(key: str) -> Optional[str]:
    """RoboBase ``latest.pt`` ``ckpt["agent"]`` key -> this module's key (None: not a weight this package holds -- the ResNet's unused
    ``fc``, ``num_batches_tracked`` counters).  The CVAE posterior encoder (``actor.actor_model.encoder.layers.*`` / ``cls_embed`` /
    ``encoder_*_proj`` / ``latent_proj``) maps onto the training schema's names (act_training.act_train_schema)."""
    for src, dst in _ROBOBASE_PREFIXES:
        if key.startswith(src):
            k = dst + key[len(src):]
            k = _ROBOBASE_ALIASES.get(k, k)
            if k.startswith("backbone.fc.") or k.endswith("num_batches_tracked"):
                return None
            if k.startswith("encoder.layers.") and (src == "encoder." or src == "actor.encoder_model."):
                return None  # the image encoder has no ``layers``: only ``actor_model.encoder`` is the CVAE style encoder
            return k
    return _ROBOBASE_ALIASES.get(key, key)


def robobase_key_names(key: str, aliases: Optional[Dict[str, str]] = None):
    """This module's key -> EVERY path RoboBase's ``agent.state_dict()`` lists it under (the inverse of ``robobase_key_map``):
    ``GenimaACT`` registers ``self.encoder``, ``self.actor_model`` and ``self.actor`` (= GenimaACTPolicy holding both again as
    ``.encoder_model`` / ``.actor_model``; controller/method/genima_act.py:221-249), and ``nn.Module.state_dict()`` does not dedupe
    shared submodules -- the reference's gate (controller/eval_genima.py:94-100) walks all of them.  ``aliases`` renames leaf modules
    (own name -> checkpoint name, e.g. ``{"task_proj": "proj_text_emb"}``) for the [VERIFY] items of SURVEY.md Appendix E."""
    if aliases:
        head = key.split(".")[0]
        if head in aliases:
            key = aliases[head] + key[len(head):]
    if key.startswith("backbone."):
        body = "backbone.0.body." + key[len("backbone."):]
        return ["encoder." + body, "actor.encoder_model." + body]
    if key.startswith("input_proj."):
        return ["encoder." + key, "actor.encoder_model." + key]
    if key.startswith("projection_layer."):
        return ["actor." + key]
    return ["actor_model." + key, "actor.actor_model." + key]


_HF2OPENAI_CLIP = (("text_model.embeddings.token_embedding.weight", "token_embedding.weight"),
                   ("text_model.embeddings.position_embedding.weight", "positional_embedding"),
                   ("text_model.final_layer_norm.", "ln_final."), ("text_projection.weight", "text_projection"))


def clip_hf_to_openai(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    """transformers ``CLIPTextModelWithProjection`` names -> the openai ``clip`` package's (``clip.load("ViT-B/32")`` with ``.visual``
    deleted: controller/method/genima_act.py:314-346): q|k|v re-fused into ``attn.in_proj_*``, ``text_projection`` stored [width, proj]."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    layers = sorted({int(m.group(1)) for k in sd for m in [re.match(r"text_model\.encoder\.layers\.(\d+)\.", k)] if m})
    for src, dst in _HF2OPENAI_CLIP:
        for k, v in sd.items():
            if k == src:
                out[dst] = v.t().contiguous() if dst == "text_projection" else v
            elif src.endswith(".") and k.startswith(src):
                out[dst + k[len(src):]] = v
    for i in layers:
        p, q = f"text_model.encoder.layers.{i}.", f"transformer.resblocks.{i}."
        for wb in ("weight", "bias"):
            out[q + "attn.in_proj_" + wb] = torch.cat([sd[p + f"self_attn.{n}_proj.{wb}"] for n in "qkv"], dim=0)
            out[q + "attn.out_proj." + wb] = sd[p + "self_attn.out_proj." + wb]
            out[q + "ln_1." + wb], out[q + "ln_2." + wb] = sd[p + "layer_norm1." + wb], sd[p + "layer_norm2." + wb]
            out[q + "mlp.c_fc." + wb], out[q + "mlp.c_proj." + wb] = sd[p + "mlp.fc1." + wb], sd[p + "mlp.fc2." + wb]
    return out


def clip_openai_to_hf(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    """The inverse: an openai ``clip`` text-side state dict (keys optionally prefixed ``clip_model.``) -> transformers names."""
    sd = {(k[len("clip_model."):] if k.startswith("clip_model.") else k): v for k, v in sd.items()}
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    out["text_model.embeddings.token_embedding.weight"] = sd["token_embedding.weight"]
    out["text_model.embeddings.position_embedding.weight"] = sd["positional_embedding"]
    i = 0
    while f"transformer.resblocks.{i}.attn.in_proj_weight" in sd:
        p, q = f"text_model.encoder.layers.{i}.", f"transformer.resblocks.{i}."
        for wb in ("weight", "bias"):
            for n, part in zip("qkv", sd[q + "attn.in_proj_" + wb].chunk(3, dim=0)):
                out[p + f"self_attn.{n}_proj.{wb}"] = part.contiguous()
            out[p + "self_attn.out_proj." + wb] = sd[q + "attn.out_proj." + wb]
            out[p + "layer_norm1." + wb], out[p + "layer_norm2." + wb] = sd[q + "ln_1." + wb], sd[q + "ln_2." + wb]
            out[p + "mlp.fc1." + wb], out[p + "mlp.fc2." + wb] = sd[q + "mlp.c_fc." + wb], sd[q + "mlp.c_proj." + wb]
        i += 1
    out["text_model.final_layer_norm.weight"], out["text_model.final_layer_norm.bias"] = sd["ln_final.weight"], sd["ln_final.bias"]
    out["text_projection.weight"] = sd["text_projection"].t().contiguous()
    return out


class GenimaACT:
    """Controller plugin (``method._target_: method.genima_act.GenimaACT``, controller/cfgs/method/genima_act.yaml:3-4)."""

    def __init__(self, config: Optional[dict] = None, state_dict=None, clip_config: Optional[dict] = None, clip_state_dict=None,
                 device="cuda", seed: int = 0, key_aliases: Optional[Dict[str, str]] = None, **hydra_kwargs):
        self.key_aliases = dict(key_aliases or {})  # own leaf-module name -> checkpoint name ([VERIFY] items; settable from the method YAML)
        self._sd_train: "OrderedDict[str, torch.Tensor]" = OrderedDict()  # CVAE posterior encoder (training only), once loaded / trained
        self._clip_used = False
        self.config = FrozenConfig(dict(config or configs.ACT_POLICY))
        self.clip_config = FrozenConfig(dict(clip_config or configs.ACT_CLIP_TEXT))
        self._schema = act_schema(self.config)
        self._sd = OrderedDict((k, v.float()) for k, v in (state_dict or weights.synth_state_dict(self._schema, seed + 31)).items())
        self._clip_sd = clip_state_dict or weights.synth_state_dict(schema.clip_text_schema(self.clip_config), seed + 32)
        from .act_training import act_train_schema  # the CVAE posterior encoder is a registered module of the reference's agent too

        cvae = OrderedDict((k, v) for k, v in act_train_schema(self.config).items() if k not in self._schema)
        self._sd_train = OrderedDict((k, v.float()) for k, v in weights.synth_state_dict(cvae, seed + 33).items())
        self.device = torch.device(device)
        self.training = False
        self.W = self.Wclip = None
        self._progs = {}
        if self.device.type == "cuda" and torch.cuda.is_available():
            self.to(self.device)

    # ---- plugin surface -------------------------------------------------------------------------------------------
    def to(self, device):
        dev = torch.device(device)
        if dev.type != "cuda" or not torch.cuda.is_available():
            raise GenimaHipError("GenimaACT needs a ROCm device; there is no CPU fallback")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.W = pack_act(self._sd, self.device)
        self.Wclip = packing.pack_state_dict(self._clip_sd, self.device)
        self._progs = {}
        return self

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def _own_state(self) -> "OrderedDict[str, torch.Tensor]":
        """Forward weights (+ the CVAE posterior encoder once ``update`` ran or a checkpoint carried it), pulled lazily from the trainer."""
        tr = getattr(self, "_trainer", None)
        if tr is not None and getattr(self, "_stale_host", False):
            for k, v in tr.state_dict().items():
                (self._sd if k in self._sd else self._sd_train)[k] = v.detach().float().cpu()
            self._stale_host = False
        own = OrderedDict(self._sd)
        own.update(self._sd_train)
        return own

    def state_dict(self):
        """RoboBase's ``agent.state_dict()`` key set: every weight under each of its registrations (``encoder.*`` /
        ``actor.encoder_model.*``, ``actor_model.*`` / ``actor.actor_model.*``, ``actor.projection_layer.*``), so the reference's
        unchanged gate (controller/eval_genima.py:94-100: every non-clip key must be in ``ckpt["agent"]``) and ``save_snapshot``
        (controller/train_act.py:262-279) see the names a RoboBase checkpoint holds.  ``clip_model.*`` (openai names) appears once the
        text tower has been used, as in the reference (lazy ``clip.load`` on the first ``encode_clip_text``, genima_act.py:315-321)."""
        out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for k, v in self._own_state().items():
            for name in robobase_key_names(k, self.key_aliases):
                out[name] = v
        if self._clip_used:
            for k, v in clip_hf_to_openai(self._clip_sd).items():
                out["clip_model." + k] = v
        return out

    def load_state_dict(self, sd, strict: bool = False):
        """``agent.load_state_dict(ckpt["agent"], strict=False)`` (controller/eval_genima.py:91-103): RoboBase key families are mapped
        by ``robobase_key_map``; with ``strict=False`` unknown keys are reported, not fatal -- but a checkpoint that fills NONE of the
        forward's weights is an error (a silently random-initialised controller would act plausibly and wrongly)."""
        if getattr(self, "_trainer", None) is not None:
            # update() calls trained on the device: pull those weights to the host copy first, so that keys a partial checkpoint does not
            # carry keep their TRAINED values, and say that the optimizer state goes with the trainer
            import warnings

            self._own_state()
            warnings.warn("GenimaACT.load_state_dict: discarding the live ACT trainer (Adam moments, step count); keys missing from the "
                          "checkpoint keep their trained values")
        inv = {v: k for k, v in self.key_aliases.items()}
        new = {}
        for k, v in sd.items():
            if k.startswith("clip_model."):
                continue
            m = robobase_key_map(k)
            if m is not None and inv:
                head = m.split(".")[0]
                m = inv[head] + m[len(head):] if head in inv else m
            if m is not None and (m not in new or k.startswith("actor.")):  # duplicates carry the same tensor; prefer the actor.* path
                new[m] = v
        missing = [k for k in self._schema if k not in new]
        if len(missing) == len(self._schema):
            raise KeyError(f"no key of the state dict maps onto the ACT forward (first keys: {list(sd)[:4]})")
        if strict and missing:
            raise KeyError(f"missing keys: {missing[:4]}")
        for k in self._schema:
            if k in new:
                self._sd[k] = new[k].detach().float().cpu()
        unexpected = []
        for k, v in new.items():
            if k in self._schema:
                continue
            if k.startswith(_CVAE_PREFIXES):
                self._sd_train[k] = v.detach().float().cpu()
            else:
                unexpected.append(k)
        clip_keys = {k: v for k, v in sd.items() if k.startswith("clip_model.")}
        if clip_keys:  # a state dict taken after the first act() carries the text tower too (snapshots strip it)
            self._clip_sd = OrderedDict((k, v.detach().float().cpu()) for k, v in clip_openai_to_hf(clip_keys).items())
        self._trainer, self._stale_host = None, False  # a later update() starts from the loaded weights
        if self.W is not None:
            self.W = pack_act(self._sd, self.device)
            if clip_keys:
                self.Wclip = packing.pack_state_dict(self._clip_sd, self.device)
            self._progs = {}
        return missing, unexpected

    # ---- recorded forward programs (one per input shape), replayed from C++ --------------------------------------------------
    def _program(self, B, V, H, Wd, lang: bool):
        if self.W is None:
            raise GenimaHipError("GenimaACT is not on a ROCm device (no CPU fallback)")
        key = (B, V, H, Wd, lang)
        io = self._progs.get(key)
        if io is None:
            from types import SimpleNamespace

            from .engine import save_tune_table

            E = Engine(self.device, record=True)
            io = SimpleNamespace(engine=E)
            io.img = E.buf("in_img", (B, V, H, Wd, 3), dtype=torch.uint8, zero=True)
            sdim = (self.config["state_dim"] + 7) // 8 * 8
            io.qpos = E.buf("in_qpos", (B, sdim), zero=True)
            io.tokens = E.buf("in_tokens", (B, 77), dtype=torch.int32, zero=True) if lang else None
            io.a_hat, io.is_pad, io.task = emit_act_forward(E, self.W, self.Wclip, self.config, self.clip_config, io.img, io.qpos, io.tokens)
            save_tune_table()
            self._progs[key] = io
        return io

    def _run(self, img_u8_nhwc, qpos, tokens):
        B, V, H, Wd, _ = img_u8_nhwc.shape
        lang = bool(self.config.get("use_lang_cond")) and tokens is not None
        self._clip_used = self._clip_used or lang
        io = self._program(B, V, H, Wd, lang)
        io.img.copy_(img_u8_nhwc)
        io.qpos[:, : qpos.shape[1]].copy_(qpos.to(torch.float16))
        if lang:
            io.tokens.copy_(tokens.reshape(B, -1, tokens.shape[-1])[:, 0].to(torch.int32))  # text does not change across frames
        io.engine.use_stream(torch.cuda.current_stream(self.device))
        io.engine.run()
        return io

    def encode_clip_text(self, tokens: torch.Tensor):
        """tokens int [B, fs, 77] -> (task_emb f32 [B, projection_dim], None) (controller/method/genima_act.py:314-346)."""
        if self.W is None:
            raise GenimaHipError("GenimaACT is not on a ROCm device")
        self._clip_used = True
        E = Engine(self.device)
        tks = tokens.reshape(tokens.shape[0], -1, tokens.shape[-1])[:, 0].to(self.device, torch.int32).contiguous()
        x = graphs.emit_clip_text(E, self.Wclip, self.clip_config, tks)
        pooled = E.gather_rows(x, E.argmax_rows(tks))
        return E.linear(pooled, self.Wclip["text_projection.weight"]).to(torch.float32), x

    def act_tiled(self, tiled_u8: torch.Tensor, low_dim_state: torch.Tensor, lang_tokens: Optional[torch.Tensor]) -> torch.Tensor:
        """Device-resident fast path of eval_genima.py:224-247: the pipeline's uint8 tiled output [B, 2v, 2v, 3] is untiled on
        the device (crop order of controller/utils/misc.py:25-30 -> camera order of the tile) and fed straight to the policy."""
        B, H2, W2, _ = tiled_u8.shape
        v = H2 // 2
        img = torch.stack([tiled_u8[:, y:y + v, x:x + v] for (x, y) in ((0, 0), (v, 0), (0, v), (v, v))], dim=1)  # layout only
        io = self._run(img, low_dim_state.to(self.device).flatten(1), None if lang_tokens is None else lang_tokens.to(self.device))
        return io.a_hat[..., : self.config["action_dim"]]

    def update(self, replay_iter, step: int = 0, replay_buffer=None, **trainer_kw) -> Dict[str, float]:
        """``GenimaACT.update`` (controller/method/genima_act.py:348-422): one behaviour-cloning step on ``next(replay_iter)`` -- a dict
        with ``action`` [B, T, A], ``low_dim_state`` [B, fs, S], the ``*rgb*`` camera tensors [B, fs, 3, H, W] (``tp1`` keys ignored),
        ``lang_tokens`` [B, fs, 77] and ``reward``.  The trainer (act_training.ACTTrainer: CVAE posterior, loss, tape backward, two-group
        AdamW) is built on first use from this agent's weights; train-time augmentation runs when ``data_augmentation`` is on."""
        from .act_training import ACTTrainer, act_augment, act_train_schema

        if getattr(self, "_trainer", None) is None:
            sch = act_train_schema(self.config)
            if int(self.config.get("frame_stack", 1)) > 1:
                raise NotImplementedError("ACTTrainer does not apply projection_layer: frame_stack > 1 is inference-only here")
            sd = dict(self._sd)
            sd.update(self._sd_train)  # a checkpoint's CVAE posterior encoder, when one was loaded
            fresh = OrderedDict((k, v) for k, v in sch.items() if k not in sd)
            if fresh:
                sd.update(weights.synth_state_dict(fresh, 77))  # CVAE encoder: fresh init
            self._trainer = ACTTrainer(Engine(self.device), self.config, sd, self.clip_config, self.Wclip, **trainer_kw)
            self._aug_gen = torch.Generator().manual_seed(0)
        batch = next(replay_iter)
        batch = {k: torch.as_tensor(v) for k, v in batch.items()}
        qpos = batch["low_dim_state"].flatten(1).float()
        keys = [k for k in batch if re.match(r".*rgb(?!.*?tp1)", k) and "tp1" not in k]
        image = torch.stack([batch[k] for k in keys], dim=1)  # [B, V, fs, 3, H, W]
        B = image.shape[0]
        image = image.reshape(B, -1, 3, image.shape[-2], image.shape[-1]).to(self.device)
        img_u8 = (image if image.dtype == torch.uint8 else image.round().clamp(0, 255).to(torch.uint8)).permute(0, 1, 3, 4, 2).contiguous()
        task = None
        if self.config.get("use_lang_cond"):
            task, _ = self.encode_clip_text(batch["lang_tokens"])
        tr = self._trainer
        imgs = act_augment(tr.E, img_u8, self._aug_gen) if self.config.get("data_augmentation", True) else img_u8
        metrics = tr.update(imgs, qpos, task, batch["action"].float())
        if "reward" in batch:
            metrics["batch_reward"] = float(batch["reward"].float().mean())
        # the eval path reads the packed inference weights and state_dict() the host copies: both are refreshed lazily (act() /
        # _own_state()), not with a device sync + full-state D2H copy per training step
        self._dirty = self._stale_host = True
        return metrics

    def act(self, obs: Dict[str, torch.Tensor], step: int = 0, eval_mode: bool = True) -> torch.Tensor:
        """obs: {'<cam>_rgb': uint8/float [B, fs, 3, H, W], 'low_dim_state': f32 [B, fs, state], 'lang_tokens': int [B, fs, 77]}."""
        if getattr(self, "_dirty", False):  # weights moved by update(): re-pack once before acting
            self._own_state()
            self.W = pack_act(self._sd, self.device)
            self._progs, self._dirty = {}, False
        qpos = obs["low_dim_state"].to(self.device).flatten(1)
        rgb_keys = [k for k in obs if re.match(r"rgb.*|.*_rgb$", k)]  # obs-dict key order == RoboBase camera enumeration
        image = torch.stack([obs[k].to(self.device) for k in rgb_keys], dim=1)  # [B, V, fs, 3, H, W]
        B = image.shape[0]
        image = image.reshape(B, -1, 3, image.shape[-2], image.shape[-1])
        img_u8 = image.round().clamp(0, 255).to(torch.uint8) if image.dtype != torch.uint8 else image
        img_u8 = img_u8.permute(0, 1, 3, 4, 2).contiguous()
        toks = obs.get("lang_tokens") if self.config.get("use_lang_cond") else None
        io = self._run(img_u8, qpos, None if toks is None else toks.to(self.device))
        return io.a_hat[..., : self.config["action_dim"]].to(torch.float32)
