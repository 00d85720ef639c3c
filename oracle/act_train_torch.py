"""CPU oracle of the ACT controller update: torch autograd over oracle/act_torch.py plus the CVAE posterior encoder and the loss.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates the reference's own training-side code:
  * ``GenimaMVTransformer.forward`` training branch (controller/method/genima_act.py:57-68): style_variable_encoder -> latent_proj ->
    (mu, logvar) -> reparametrize -> latent_out_proj;
  * ``calculate_loss`` (:94-139): masked mean L1 on the first A - 1 action dims, 0.05 x BCE-with-logits on the gripper dim,
    KL(mu, logvar) x kl_weight;
  * ``GenimaACT.update`` (:348-422): ``is_pad`` all False, AdamW with the backbone / rest parameter groups of ``build_actor`` (:251-271).
The CVAE encoder itself lives in RoboBase (absent): the public ACT layout is restated -- [CLS] ++ proj(qpos) ++ proj(actions) plus a fixed
sinusoid table through the same post-norm encoder layers -- PARITY UNPINNED for that part ([VERIFY]); the layer
arithmetic is the DETR layer pinned in act_torch.py.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from . import act_torch as OA

Tensor = torch.Tensor


def sinusoid_table(n: int, d: int) -> Tensor:
    """ACT ``get_sinusoid_encoding_table``: angle(pos, j) = pos / 10000^(2 (j // 2) / d); sin on even, cos on odd columns."""
    pos = torch.arange(n, dtype=torch.float64)[:, None]
    j = torch.arange(d, dtype=torch.float64)[None]
    ang = pos / torch.pow(torch.tensor(10000.0, dtype=torch.float64), 2 * torch.div(j, 2, rounding_mode="floor") / d)
    tab = torch.where((torch.arange(d) % 2 == 0)[None], ang.sin(), ang.cos())
    return tab.float()


def style_encoder(sd, cfg, qpos: Tensor, actions: Tensor, q=OA._id, drop=OA._nodrop) -> Tensor:
    """-> latent_info [B, 2 * latent_dim] = latent_proj(encoder([CLS, proj(qpos), proj(actions)])[CLS])."""
    B, T, _ = actions.shape
    d, heads = cfg["hidden_dim"], cfg["nheads"]
    cls = sd["cls_embed.weight"][None].expand(B, 1, d)
    qp = F.linear(qpos, sd["encoder_joint_proj.weight"], sd["encoder_joint_proj.bias"])[:, None]
    ap = F.linear(actions, sd["encoder_action_proj.weight"], sd["encoder_action_proj.bias"])
    x = q(torch.cat([cls, qp, ap], dim=1))
    pos = q(sinusoid_table(T + 2, d))[None].expand(B, -1, -1)
    for i in range(cfg["enc_layers"]):
        p = f"encoder.layers.{i}"
        qk = q(x + pos)
        x = q(OA.ln(sd, p + ".norm1", x + drop(p + ".d1", OA.mha(sd, p + ".self_attn", qk, qk, x, heads, q))))
        ff = F.linear(q(drop(p + ".df", F.relu(F.linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])))), sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
        x = q(OA.ln(sd, p + ".norm2", x + drop(p + ".d2", ff)))
    return q(F.linear(x[:, 0], sd["latent_proj.weight"], sd["latent_proj.bias"]))


def loss_fn(a_hat: Tensor, actions: Tensor, is_pad: Tensor, mu: Tensor, logvar: Tensor, kl_weight: float) -> Dict[str, Tensor]:
    """controller/method/genima_act.py:115-139 (kl_divergence of robobase/models/act/utils/misc.py: sum over dims, mean over batch)."""
    keep = (~is_pad).float()
    l1 = (F.l1_loss(actions[..., :-1], a_hat[..., :-1], reduction="none") * keep[..., None]).mean()
    grip = (F.binary_cross_entropy_with_logits(a_hat[..., -1], actions[..., -1], reduction="none") * 0.05 * keep).mean()
    kl = (-0.5 * (1 + logvar - mu.pow(2) - logvar.exp())).sum(1).mean(0)
    return {"l1": l1, "gripper_loss": grip, "kl": kl, "loss": l1 + grip + kl * kl_weight}


def update_forward_backward(sd: Dict[str, Tensor], cfg, images_u8: Tensor, qpos: Tensor, task_emb: Tensor, actions: Tensor, eps: Tensor,
                            trainable, q=OA._id, drop=OA._nodrop, images_float: Tensor = None) -> Tuple[Dict[str, Tensor], Dict[str, Tensor], Tensor]:
    """One ``update``: -> (loss dict, {name: gradient} for the names in ``trainable``, a_hat)."""
    params = {k: (v.detach().clone().requires_grad_(True) if k in trainable else v.detach()) for k, v in sd.items()}
    T = cfg["num_queries"]
    acts = actions[:, :T]
    is_pad = torch.zeros(acts.shape[:2], dtype=torch.bool)  # GenimaACT.update: is_pad = zeros_like(actions)[:, :, 0].bool()
    info = style_encoder(params, cfg, qpos, acts, q, drop)
    L = cfg["latent_dim"]
    mu, logvar = info[:, :L], info[:, L:]
    z = q(mu + (logvar / 2).exp() * eps)
    a_hat, _ = OA.act_forward(params, cfg, images_u8, qpos, task_emb, q, latent_z=z, drop=drop, images_float=images_float)
    out = loss_fn(a_hat, acts, is_pad, mu, logvar, cfg.get("kl_weight", 10.0))
    out["loss"].backward()
    grads = {k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in params.items() if k in trainable}
    return {k: v.detach() for k, v in out.items()}, grads, a_hat.detach()


def adamw_groups_step(sd, grads, backbone_names, lr: float, lr_backbone: float, weight_decay: float):
    """``build_actor``'s optimizer (genima_act.py:251-271): AdamW, two parameter groups, no gradient clipping (actor_grad_clip: null)."""
    ps = {k: torch.nn.Parameter(sd[k].detach().clone()) for k in grads}
    for k, p in ps.items():
        p.grad = grads[k].detach().clone()
    opt = torch.optim.AdamW([{"params": [p for k, p in ps.items() if k not in backbone_names]},
                             {"params": [p for k, p in ps.items() if k in backbone_names], "lr": lr_backbone}], lr=lr, weight_decay=weight_decay)
    opt.step()
    return {k: p.detach() for k, p in ps.items()}
