"""ACT controller update step on libgenima_hip.so.

Reference: ``GenimaACT.update`` (controller/method/genima_act.py:348-422) -> ``GenimaACTPolicy.forward`` with actions (:165-214) ->
``GenimaMVTransformer.forward`` training branch (:57-68: CVAE posterior -> reparametrize -> latent token) and ``calculate_loss``
(:94-139); optimizer = ``build_actor``'s AdamW with the backbone / rest parameter groups (:251-271, lr 5e-5 / lr_backbone 1e-5 /
weight_decay 1e-4 / no gradient clipping: controller/cfgs/method/genima_act.yaml:7-12).

Same execution model as the ControlNet fine-tune (training.py): an eager tape of hand-written backward ops over the MFMA GEMM /
attention / LayerNorm kernels, flat fp32 master + gradient + Adam moments, f16 compute with a loss scale.  What is trainable follows
DETR's backbone rule RoboBase inherits: ResNet ``conv1`` / ``layer1`` and every (Frozen)BatchNorm stay fixed; ``layer2..4``, the FiLM
generators and everything in the transformer train.  The frozen BatchNorms are applied as per-channel affines (gn_film) so the conv
weights train un-folded, exactly as in the reference.

Deviations, stated: (1) f16 compute where the reference trains in fp32; (2) of the DETR layers' dropouts the residual / feed-forward
ones and the state MLP's p = 0.3 are applied, ``nn.MultiheadAttention``'s attention-probability dropout is not (the flash kernel never
materialises the probabilities); (3) the CVAE encoder's module names are [VERIFY] items (RoboBase is absent).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import graphs, packing
from . import train_ops as T
from ._lib import ACT_NONE, ACT_RELU
from .act import IMAGENET_MEAN, IMAGENET_STD, _RESNET18, act_schema, pack_act, sine_pos_embed
from .engine import Engine
from .training import FrozenParams, Graph, TrainParams, Var

F16, F32 = torch.float16, torch.float32


def act_train_schema(cfg) -> "OrderedDict[str, tuple]":
    """``act_schema`` + the training-only CVAE posterior encoder (public ACT layout: cls_embed, encoder_action_proj, encoder_joint_proj,
    latent_proj and ``enc_layers`` more post-norm encoder layers under ``encoder.layers``)."""
    s = act_schema(cfg)
    d, ff, L = cfg["hidden_dim"], cfg["dim_feedforward"], cfg["latent_dim"]
    s["cls_embed.weight"] = (1, d)
    s["encoder_action_proj.weight"], s["encoder_action_proj.bias"] = (d, cfg["action_dim"]), (d,)
    s["encoder_joint_proj.weight"], s["encoder_joint_proj.bias"] = (d, cfg["state_dim"]), (d,)
    s["latent_proj.weight"], s["latent_proj.bias"] = (2 * L, d), (2 * L,)
    for i in range(cfg["enc_layers"]):
        p = f"encoder.layers.{i}"
        s[p + ".self_attn.in_proj_weight"], s[p + ".self_attn.in_proj_bias"] = (3 * d, d), (3 * d,)
        s[p + ".self_attn.out_proj.weight"], s[p + ".self_attn.out_proj.bias"] = (d, d), (d,)
        s[p + ".linear1.weight"], s[p + ".linear1.bias"] = (ff, d), (ff,)
        s[p + ".linear2.weight"], s[p + ".linear2.bias"] = (d, ff), (d,)
        for n in ("norm1", "norm2"):
            s[f"{p}.{n}.weight"], s[f"{p}.{n}.bias"] = (d,), (d,)
    return s


def trainable_names(schema) -> list:
    """Backbone group first (every trainable name containing "backbone": layer2..4 convs and the FiLM generators), then the rest --
    the two AdamW groups of build_actor are then two contiguous ranges of the flat buffers."""
    def frozen(k):
        return (".bn" in k or "downsample.1." in k or k.startswith("backbone.conv1") or k.startswith("backbone.bn1")
                or k.startswith("backbone.layer1.") or k.startswith("is_pad_head."))  # is_pad_hat is not part of the loss: grad None
    tr = [k for k in schema if not frozen(k)]
    return [k for k in tr if "backbone" in k] + [k for k in tr if "backbone" not in k]


def sinusoid_table(n: int, d: int) -> torch.Tensor:
    pos = torch.arange(n, dtype=torch.float64)[:, None]
    j = torch.arange(d, dtype=torch.float64)[None]
    ang = pos / torch.pow(torch.tensor(10000.0, dtype=torch.float64), 2 * torch.div(j, 2, rounding_mode="floor") / d)
    return torch.where((torch.arange(d) % 2 == 0)[None], ang.sin(), ang.cos()).float()


def _rup(x, m):
    return (x + m - 1) // m * m


class ACTTrainer:
    def __init__(self, E: Engine, cfg, state_dict: Dict[str, torch.Tensor], clip_cfg, clip_W, *, lr: float = 5e-5, lr_backbone: float = 1e-5,
                 weight_decay: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, loss_scale: float = 1024.0, dropout: float = 0.1,
                 state_dropout: float = 0.3, seed: int = 0):
        self.E, self.cfg, self.clip_cfg, self.clip_W = E, dict(cfg), clip_cfg, clip_W
        self.lr, self.lr_backbone, self.wd, self.betas, self.eps = lr, lr_backbone, weight_decay, betas, eps
        self.loss_scale, self.p_drop, self.p_state = float(loss_scale), dropout, state_dropout
        sch = act_train_schema(cfg)
        sd = {k: state_dict[k].detach().float() for k in sch}
        self.names = trainable_names(sch)
        self.cn = TrainParams(E, OrderedDict((k, sd[k]) for k in self.names))
        self.n_backbone = 0
        for k in self.names:
            if "backbone" in k:
                off, shape = self.cn.layout[k]
                n = 1
                for s_ in shape:
                    n *= s_
                self.n_backbone = max(self.n_backbone, _rup(off + n, 8))
        self._alias_mha()
        # frozen part: conv1 / layer1 with their BatchNorms folded (the inference packing), and every other BatchNorm as an affine
        self.frozen = FrozenParams(E, pack_act({k: v for k, v in sd.items() if k in act_schema(cfg)}, E.device))
        dev = E.device
        self.bn = {}
        for k in sch:
            if k.endswith(".running_var"):
                p = k[: -len(".running_var")]
                scale = sd[p + ".weight"] * (sd[p + ".running_var"] + 1e-5).rsqrt()
                shift = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
                self.bn[p] = ((scale - 1.0).to(F16)[None].contiguous().to(dev), shift.to(F16)[None].contiguous().to(dev))
        self._gen = torch.Generator(device=dev).manual_seed(seed)
        self.opt_step, self._ss, self._clip, self.last = 0, torch.zeros(1, dtype=F32, device=dev), torch.zeros(3, dtype=F32, device=dev), {}
        mean, std = torch.tensor(IMAGENET_MEAN), torch.tensor(IMAGENET_STD)
        g8, b8 = torch.zeros(1, 8), torch.zeros(1, 8)
        g8[0, :3], b8[0, :3] = 1.0 / std - 1.0, -mean / std
        g8[0, 3:] = -1.0  # padded channels stay exactly 0
        self._norm = (g8.to(F16).to(dev), b8.to(F16).to(dev))

    def _alias_mha(self):
        """``in_proj_weight`` [3d, d] is ONE parameter: q | k (fused), v, q, k are row ranges of it -- views into the flat buffers."""
        d = self.cfg["hidden_dim"]
        W, G = self.cn.W, self.cn.G
        for name in list(W):
            if isinstance(name, str) and name.endswith("in_proj_weight"):
                p = name[: -len("in_proj_weight")]
                for alias, (a, b) in (("qk_proj", (0, 2 * d)), ("v_proj", (2 * d, 3 * d)), ("q_proj", (0, d)), ("k_proj", (d, 2 * d))):
                    W[p + alias + ".weight"], G[p + alias + ".weight"] = W[name][a:b], G[name][a:b]
                    W[p + alias + ".bias"], G[p + alias + ".bias"] = W[p + "in_proj_bias"][a:b], G[p + "in_proj_bias"][a:b]

    # ------------------------------------------------------------------------------------------------------------------ forward pieces
    def _block(self, g: Graph, hv: Var, p: str, c: int, stride: int, film: Optional[Var], bi: int, rows_pf: int) -> Var:
        net, bn = self.cn, self.bn
        x_in = hv
        # (stride 2: the 3x3 conv strides itself; the 1x1 shortcut below sub-samples explicitly through copy4d)
        y = g.conv(net, hv, p + ".conv1.weight", None, stride=stride)
        rows_all = y.t.numel() // c
        y = g.film(y, *bn[p + ".bn1"], rows_all)
        if film is not None:
            ff = film.t
            y = g.film(y, ff[:, bi * c:(bi + 1) * c], ff[:, (2 + bi) * c:(3 + bi) * c], rows_pf * y.t.shape[1] * y.t.shape[2], ACT_RELU, feat=film)
        else:
            y = g.act(y, ACT_RELU)
        idt = x_in
        if (p + ".downsample.0.weight") in net.W:
            sub = self._subsample2(g, x_in) if stride == 2 else x_in
            idt = g.conv(net, sub, p + ".downsample.0.weight", None, ksize=1)
            idt = g.film(idt, *bn[p + ".downsample.1"], idt.t.numel() // c)
        z = g.conv(net, y, p + ".conv2.weight", None)
        z = g.film(z, *bn[p + ".bn2"], z.t.numel() // c)
        return g.act(g.add(z, idt), ACT_RELU)

    def _subsample2(self, g: Graph, x: Var) -> Var:
        E = g.E
        B, H, W, C = x.t.shape
        y = torch.empty((B, H // 2, W // 2, C), dtype=F16, device=E.device)
        E.copy4d(x.t, y, (1, B, H // 2, W // 2), (0, H * W * C, 2 * W * C, 2 * C), (0, (H // 2) * (W // 2) * C, (W // 2) * C, C), C)
        return g.custom(y, x.needs, lambda dy: g.acc(x, T.zero_upsample2x(E, dy.contiguous())))

    def _tokens(self, g: Graph, parts, Np: int) -> Var:
        """[B, n_i, d] Vars -> one zero-padded [B, Np, d] sequence (copies; the backward slices the gradient back)."""
        E = g.E
        B, d = parts[0].t.shape[0], parts[0].t.shape[-1]
        seq = torch.zeros((B, Np, d), dtype=F16, device=E.device)
        off = []
        o = 0
        for v in parts:
            n = v.t.numel() // (B * d)
            seq[:, o:o + n].copy_(v.t.reshape(B, n, d))
            off.append((o, n))
            o += n

        def bw(dy):
            for v, (a, n) in zip(parts, off):
                if v.needs:
                    g.acc(v, dy[:, a:a + n].reshape(v.t.shape).contiguous())
        return g.custom(seq, True, bw)

    def _enc_layer(self, g: Graph, p: str, src: Var, pos: Var, n_valid: int) -> Var:
        net, d, heads = self.cn, self.cfg["hidden_dim"], self.cfg["nheads"]
        xp = g.add(src, pos)
        qk = g.linear(net, xp, p + ".self_attn.qk_proj.weight", p + ".self_attn.qk_proj.bias")
        v = g.linear(net, src, p + ".self_attn.v_proj.weight", p + ".self_attn.v_proj.bias")
        a = g.attention(qk, 0, qk, d, v, heads, n_valid)
        o = g.dropout(g.linear(net, a, p + ".self_attn.out_proj.weight", p + ".self_attn.out_proj.bias"), self.p_drop, self._gen)
        src = g.layernorm(net, g.add(src, o), p + ".norm1.weight", p + ".norm1.bias")
        h = g.dropout(g.act(g.linear(net, src, p + ".linear1.weight", p + ".linear1.bias"), ACT_RELU), self.p_drop, self._gen)
        f = g.dropout(g.linear(net, h, p + ".linear2.weight", p + ".linear2.bias"), self.p_drop, self._gen)
        return g.layernorm(net, g.add(src, f), p + ".norm2.weight", p + ".norm2.bias")

    def _dec_layer(self, g: Graph, p: str, tgt: Var, qe: Var, memory: Var, mem_pos: Var, nq: int, n_mem: int) -> Var:
        net, d, heads = self.cn, self.cfg["hidden_dim"], self.cfg["nheads"]
        tq = g.add(tgt, qe)
        qk = g.linear(net, tq, p + ".self_attn.qk_proj.weight", p + ".self_attn.qk_proj.bias")
        v = g.linear(net, tgt, p + ".self_attn.v_proj.weight", p + ".self_attn.v_proj.bias")
        a = g.attention(qk, 0, qk, d, v, heads, nq)
        o = g.dropout(g.linear(net, a, p + ".self_attn.out_proj.weight", p + ".self_attn.out_proj.bias"), self.p_drop, self._gen)
        tgt = g.layernorm(net, g.add(tgt, o), p + ".norm1.weight", p + ".norm1.bias")
        m = p + ".multihead_attn"
        cq = g.linear(net, g.add(tgt, qe), m + ".q_proj.weight", m + ".q_proj.bias")
        ck = g.linear(net, mem_pos, m + ".k_proj.weight", m + ".k_proj.bias")
        cv = g.linear(net, memory, m + ".v_proj.weight", m + ".v_proj.bias")
        a = g.attention(cq, 0, ck, 0, cv, heads, n_mem)
        o = g.dropout(g.linear(net, a, m + ".out_proj.weight", m + ".out_proj.bias"), self.p_drop, self._gen)
        tgt = g.layernorm(net, g.add(tgt, o), p + ".norm2.weight", p + ".norm2.bias")
        h = g.dropout(g.act(g.linear(net, tgt, p + ".linear1.weight", p + ".linear1.bias"), ACT_RELU), self.p_drop, self._gen)
        f = g.dropout(g.linear(net, h, p + ".linear2.weight", p + ".linear2.bias"), self.p_drop, self._gen)
        return g.layernorm(net, g.add(tgt, f), p + ".norm3.weight", p + ".norm3.bias")

    # ------------------------------------------------------------------------------------------------------------------ the update
    def forward_backward(self, images: torch.Tensor, qpos: torch.Tensor, task_emb: Optional[torch.Tensor], actions: torch.Tensor,
                         eps: Optional[torch.Tensor] = None):
        """images: uint8 [B, V, H, W, 3] on the device, or already augmented f16 [B, V, H, W, 8] on the 0..1 scale; qpos f32 [B, state];
        task_emb f16/f32 [B, lang_dim] (CLIP pooled projection; frozen) or None; actions f32 [B, T >= num_queries, A]; eps f32
        [B, latent_dim] (drawn here when None).  Fills the flat gradient (loss-scaled) and returns out4 = (loss, l1, gripper, kl)."""
        E, cfg, net = self.E, self.cfg, self.cn
        dev = E.device
        g = Graph(E)
        B, V = images.shape[:2]
        d, Tq, L, A = cfg["hidden_dim"], cfg["num_queries"], cfg["latent_dim"], cfg["action_dim"]
        acts = actions[:, :Tq].to(dev, F32).contiguous()
        # ---- CVAE posterior: [CLS, proj(qpos), proj(actions)] + sinusoid table -> encoder -> latent_proj -> (mu, logvar) -> z
        Sp, Sv = _rup(Tq + 2, 8), Tq + 2
        qp16 = torch.zeros((B, _rup(cfg["state_dim"], 8)), dtype=F16, device=dev)
        qp16[:, : qpos.shape[1]] = qpos.to(dev)
        a16 = torch.zeros((B, Tq, _rup(A, 8)), dtype=F16, device=dev)
        a16[..., :A] = acts
        cls = net.W["cls_embed.weight"].view(1, 1, d).expand(B, 1, d).contiguous()

        def cls_bw(dy):
            T.colsum(E, dy.reshape(B, d).contiguous(), net.G["cls_embed.weight"].view(-1), 1, B, d, d)
        g._note(net, "cls_embed.weight")
        cls_v = g.custom(cls, True, cls_bw)
        qp_v = g.linear(net, Var(qp16, needs=False), "encoder_joint_proj.weight", "encoder_joint_proj.bias")
        ap_v = g.linear(net, Var(a16, needs=False), "encoder_action_proj.weight", "encoder_action_proj.bias")
        x = self._tokens(g, [cls_v.view(B, 1, d), qp_v.view(B, 1, d), ap_v], Sp)
        pos2 = torch.zeros((B, Sp, d), dtype=F16, device=dev)
        pos2[:, :Sv] = sinusoid_table(Sv, d).to(F16).to(dev)[None]
        pos2_v = Var(pos2, needs=False)
        for i in range(cfg["enc_layers"]):
            x = self._enc_layer(g, f"encoder.layers.{i}", x, pos2_v, Sv)
        cls_out = g.custom(x.t[:, 0].contiguous(), True, lambda dy: g.acc(x, _scatter_row(x.t, 0, dy)))
        info = g.linear(net, cls_out, "latent_proj.weight", "latent_proj.bias")  # [B, 2L]
        if eps is None:
            eps = torch.randn((B, L), generator=self._gen, device=dev, dtype=F32)
        eps = eps.to(dev, F32).contiguous()
        kl_w = float(cfg.get("kl_weight", 10.0))
        z = T.cvae_sample(E, info.t, eps, L, _rup(L, 8))
        z_v = g.custom(z, True, lambda dz: g.acc(info, T.cvae_bwd(E, info.t, eps, dz.contiguous(), L, self.loss_scale * kl_w / B)))
        # ---- image encoder: frozen conv1 / maxpool / layer1, trainable layer2..4 with FiLM
        Fz = self.frozen.W
        if images.dtype == torch.uint8:
            x8 = E.image_u8_to_f16(images.view(B * V, *images.shape[2:4], 3), 8, 1.0, 0.0)
        else:
            x8 = images.view(B * V, *images.shape[2:4], 8)
        x8 = E.film(x8, self._norm[0], self._norm[1], x8.numel() // 8)  # (v - mean) / std, padded channels 0
        h = E.conv2d(x8, Fz["backbone.conv1.weight"], Fz["backbone.conv1.bias"], ksize=7, stride=2, act=ACT_RELU)
        h = E.maxpool3x3s2(h)
        for bi in range(2):
            q = f"backbone.layer1.{bi}"
            y = E.conv2d(h, Fz[q + ".conv1.weight"], Fz[q + ".conv1.bias"], act=ACT_RELU)
            h = E.conv2d(y, Fz[q + ".conv2.weight"], Fz[q + ".conv2.bias"], residual=h, act=ACT_RELU, residual_before_act=True)
        hv = Var(h, needs=False)
        lang = bool(cfg.get("use_lang_cond")) and task_emb is not None
        task_v = None
        if lang:
            t16 = torch.zeros((B, _rup(cfg["lang_dim"], 8)), dtype=F16, device=dev)
            t16[:, : task_emb.shape[1]] = task_emb.to(dev)
            task_v = Var(t16, needs=False)
        for j, (li, c, stride) in enumerate(_RESNET18[1:]):
            film = g.linear(net, task_v, f"backbone.film_fcs.{j}.weight", f"backbone.film_fcs.{j}.bias") if lang else None
            for bi in range(2):
                hv = self._block(g, hv, f"backbone.layer{li}.{bi}", c, stride if bi == 0 else 1, film, bi, V)
        f = g.conv(net, hv, "input_proj.weight", "input_proj.bias", ksize=1)
        fh, fw = f.t.shape[1], f.t.shape[2]
        n_img = fh * V * fw
        img_tok = torch.empty((B, n_img, d), dtype=F16, device=dev)
        E.copy4d(f.t, img_tok, (B, V, fh, fw), (V * fh * fw * d, fh * fw * d, fw * d, d), (n_img * d, fw * d, V * fw * d, d), d)

        def img_bw(dy):
            df = torch.empty_like(f.t)
            E.copy4d(dy.contiguous(), df, (B, V, fh, fw), (n_img * d, fw * d, V * fw * d, d), (V * fh * fw * d, fh * fw * d, fw * d, d), d)
            g.acc(f, df)
        img_v = g.custom(img_tok, True, img_bw)
        # ---- [latent, proprio, (task)] ++ image tokens -> encoder -> decoder -> action head
        lat = g.linear(net, z_v, "latent_out_proj.weight", "latent_out_proj.bias")
        st = g.dropout(g.linear(net, Var(qp16, needs=False), "input_proj_robot_state.0.weight", "input_proj_robot_state.0.bias"), self.p_state, self._gen)
        prop = g.linear(net, st, "input_proj_robot_state.2.weight", "input_proj_robot_state.2.bias")
        extra = [lat.view(B, 1, d), prop.view(B, 1, d)]
        if lang:
            extra.append(g.linear(net, task_v, "task_proj.weight", "task_proj.bias").view(B, 1, d))
        n_extra = len(extra)
        N = n_extra + n_img
        Np = _rup(N, 8)
        src = self._tokens(g, extra + [img_v], Np)
        pos = torch.zeros((B, Np, d), dtype=F16, device=dev)
        pos_img = sine_pos_embed(fh, fw, d).repeat(1, V, 1).reshape(n_img, d)
        pos[:, :N] = torch.cat([net.W["additional_pos_embed.weight"][:n_extra].float().cpu(), pos_img], dim=0).to(F16).to(dev)[None]

        def pos_bw(dy):  # additional_pos_embed is a trainable embedding: its rows receive the batch-summed gradient
            T.colsum(E, dy[:, :n_extra].reshape(B, n_extra * d).contiguous(), net.G["additional_pos_embed.weight"].view(-1), 1, B, n_extra * d, n_extra * d)
        g._note(net, "additional_pos_embed.weight")
        pos_v = g.custom(pos, True, pos_bw)
        for i in range(cfg["enc_layers"]):
            src = self._enc_layer(g, f"transformer.encoder.layers.{i}", src, pos_v, N)
        memory = src
        mem_pos = g.add(memory, pos_v)
        Tp = _rup(Tq, 8)
        qe = torch.zeros((B, Tp, d), dtype=F16, device=dev)
        qe[:, :Tq] = net.W["query_embed.weight"][None]

        def qe_bw(dy):
            T.colsum(E, dy[:, :Tq].reshape(B, Tq * d).contiguous(), net.G["query_embed.weight"].view(-1), 1, B, Tq * d, Tq * d)
        g._note(net, "query_embed.weight")
        qe_v = g.custom(qe, True, qe_bw)
        tgt = Var(torch.zeros((B, Tp, d), dtype=F16, device=dev), needs=False)
        for i in range(cfg["dec_layers"]):
            tgt = self._dec_layer(g, f"transformer.decoder.layers.{i}", tgt, qe_v, memory, mem_pos, Tq, N)
        hs = g.layernorm(net, tgt, "transformer.decoder.norm.weight", "transformer.decoder.norm.bias")
        a_hat = g.linear(net, hs, "action_head.weight", "action_head.bias")  # [B, Tp, A] (A % 8 == 0)
        out4, d_a = T.act_loss(E, a_hat.t, acts, info.t, Tq, A, L, kl_w, self.loss_scale)
        a_hat.cell[0] = d_a
        g.backward()
        self.last["a_hat"] = a_hat.t[:, :Tq, :A]
        self.last["info"] = info.t
        return out4

    def optimizer_step(self):
        """AdamW over the two contiguous groups (backbone: lr_backbone; rest: lr), no clipping; a non-finite gradient skips the step and
        halves the loss scale (the f16 path's GradScaler; the fp32 reference has none)."""
        E, cn = self.E, self.cn
        inv = 1.0 / self.loss_scale
        T.sumsq(E, cn.grad, self._ss)
        T.clip_coef(E, self._ss, self._clip, 1e30, inv)  # max_norm = inf: coefficient 1, flag = non-finite gradient
        self.opt_step += 1
        nb, n = self.n_backbone, cn.numel
        for a, b, lr in ((0, nb, self.lr_backbone), (nb, n, self.lr)):
            if b > a:
                T.adamw(E, cn.master[a:b], cn.grad[a:b], cn.exp_avg[a:b], cn.exp_avg_sq[a:b], lr, self.betas[0], self.betas[1], self.eps, self.wd,
                        self.opt_step, self._clip, inv)
        cn.sync_half()
        self._alias_mha()
        cn.zero_grad()
        coef, norm, bad = self._clip.tolist()
        self.last["grad_norm"] = norm
        if bad:
            self.loss_scale *= 0.5
            self.opt_step -= 1
            return False
        return True

    def update(self, images, qpos, task_emb, actions, eps=None) -> Dict[str, float]:
        out4 = self.forward_backward(images, qpos, task_emb, actions, eps)
        self.optimizer_step()
        loss, l1, grip, kl = out4.tolist()
        return {"actor_loss": loss, "actor_l1_loss": l1, "actor_gripper_loss": grip, "actor_kl_loss": kl}

    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        """Trainable weights back in the checkpoint naming (fp32 masters un-packed)."""
        from .packing import unpack_state_dict
        sch = act_train_schema(self.cfg)
        return unpack_state_dict(self.cn.packed_master(), OrderedDict((k, sch[k]) for k in self.names))


def elastic_displacement(H: int, W: int, alpha: float = 80.0, sigma: float = 10.0, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """torchvision v2.ElasticTransform._get_params: per axis, uniform [-1, 1) noise, Gaussian blur (kernel int(8 sigma + 1) made odd,
    reflect padding) and a scale of alpha / size -- in the normalised [-1, 1] grid units, converted here to pixels (x (size - 1) / 2).
    Host work on one [H, W] field per call (the transform is called on the whole batch: one field for every image)."""
    import numpy as np
    from numpy.lib.stride_tricks import sliding_window_view

    k = int(8 * sigma + 1)
    k += (k % 2 == 0)
    half = (k - 1) * 0.5
    pdf = torch.exp(-0.5 * (torch.linspace(-half, half, k) / sigma) ** 2)  # (torchvision builds the kernel in f32)
    ker = (pdf / pdf.sum()).double().numpy()
    out = []
    for size, n in ((W, (H, W)), (H, (H, W))):
        # the draw stays torch's (the generator's stream is part of the recipe); the separable blur of ONE [H, W] field is host arithmetic
        # in f64 -- plain numpy: a sliding window along each axis against the kernel, no tensor-library convolution in the package
        f = (torch.rand([1, 1] + list(n), generator=generator) * 2 - 1).double().numpy()[0, 0]
        pad = k // 2
        f = np.pad(f, pad, mode="reflect")
        f = sliding_window_view(f, k, axis=1) @ ker   # along W
        f = sliding_window_view(f, k, axis=0) @ ker   # along H
        out.append(torch.from_numpy((f * alpha / size) * (size - 1) * 0.5))
    return torch.stack(out, dim=-1).float().contiguous()  # [H, W, 2] = (dx, dy) in pixels


def act_augment(E: Engine, images_u8: torch.Tensor, generator: Optional[torch.Generator] = None, p: float = 0.5, noise_std: float = 5.0):
    """``GenimaACTPolicy.aug_transforms`` (controller/method/genima_act.py:150-163) on uint8 [B, V, H, W, 3] device images:
    RandomApply(p)[ElasticTransform(80, 10)], RandomApply(p)[ColorJitter(0.2, 0.2, 0.1, 0.05)], RandomApply(p)[RandomCrop(size, padding
    = 4)] -- each called on the whole batch tensor, so ONE draw per call -- then AddGaussianNoise(0, 5.0) on the 0..255 scale
    (controller/utils/misc.py:50-65).  Returns f16 [B, V, H, W, 8] on the 0..1 scale (the trainer's float-image input).  The random
    draws use ``generator`` (a CPU generator) in torchvision's order; the pixel work runs in HIP kernels."""
    from . import augment as A
    from ._lib import check

    B, V, H, W, _ = images_u8.shape
    x = E.image_u8_to_f16(images_u8.view(B * V, H, W, 3), 8, 1.0, 0.0)
    if float(torch.rand(1, generator=generator)) < p:
        disp = elastic_displacement(H, W, generator=generator).to(E.device)
        y = torch.empty_like(x)
        check(E.lib.gn_warp_bilinear(E._ctx, x.data_ptr(), y.data_ptr(), disp.data_ptr(), B * V, H, W, 8), "gn_warp_bilinear")
        x = y
    if float(torch.rand(1, generator=generator)) < p:
        x = A.color_jitter(E, x, *A.draw_color_jitter(generator))
    if float(torch.rand(1, generator=generator)) < p:  # RandomCrop(padding=4): constant (zero) padding, then a crop back to H x W
        i = int(torch.randint(0, 9, size=(1,), generator=generator))
        j = int(torch.randint(0, 9, size=(1,), generator=generator))
        y = torch.zeros_like(x)  # out[r, c] = padded[r + i, c + j], padded = x at offset (4, 4): a shifted copy (layout only)
        r0, r1 = max(0, 4 - i), min(H, H + 4 - i)
        c0, c1 = max(0, 4 - j), min(W, W + 4 - j)
        y[:, r0:r1, c0:c1] = x[:, r0 + i - 4:r1 + i - 4, c0 + j - 4:c1 + j - 4]
        x = y
    noise = torch.zeros_like(x)
    noise[..., :3] = torch.randn((B * V, H, W, 3), device=E.device, dtype=torch.float32, generator=None).to(F16)
    one = torch.ones(B * V, dtype=F32, device=E.device)
    x = E.add_noise(x, noise, one, one * (noise_std / 255.0))
    return x.view(B, V, H, W, 8)


def _scatter_row(like: torch.Tensor, row: int, dy: torch.Tensor) -> torch.Tensor:
    out = torch.zeros_like(like)
    out[:, row] = dy
    return out
