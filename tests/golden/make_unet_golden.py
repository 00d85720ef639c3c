"""Generate tests/golden/unet_golden.npz: UNet2DConditionModel + ControlNetModel forward and one fine-tune step at reduced width
through a SECOND, structurally different route than oracle/sd_torch.py.

diffusers 0.29.0 (the reference's implementation of this arithmetic: diffusion/train_controlnet_genima.py:1066-1071, :1368-1388;
controller/agent/sd_controlnet_agent.py:32-42) is absent from the image and from /root/reference, and nothing else installed
implements a ``UNet2DConditionModel``.  So this script rebuilds the two networks as ``torch.nn.Module`` trees whose attribute
names mirror the diffusers module hierarchy (``models/unets/unet_2d_condition.py``, ``unet_2d_blocks.py``, ``resnet.py``,
``attention.py``, ``transformers/transformer_2d.py``, ``controlnet.py``) and loads the seeded synthetic weights with
``load_state_dict(strict=True)`` -- which independently checks every key name and shape of genima_amd/schema.py -- and computes with
torch's own modules: ``nn.GroupNorm`` / ``nn.Conv2d`` / ``nn.LayerNorm`` / ``nn.Linear``, ``F.scaled_dot_product_attention`` (the
kernel diffusers' ``AttnProcessor2_0`` calls, not a hand-written softmax), torch autograd + ``clip_grad_norm_`` + ``optim.AdamW`` for the
step.  It shares no code with oracle/sd_torch.py (functional, dict-of-tensors) -- but both were written by the same builder from the
same published architecture, so this is a consistency pin, not a pin against diffusers itself; DESIGN.md section 4 says so.
Run in the build container:    python tests/golden/make_unet_golden.py
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from genima_amd import configs, schema, weights  # noqa: E402
from inputs import pattern_u8  # noqa: E402


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        return (self.conv_shortcut(x) if hasattr(self, "conv_shortcut") else x) + h


class Attention(nn.Module):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads = heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_k = nn.Linear(ctx_dim, dim, bias=False)
        self.to_v = nn.Linear(ctx_dim, dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        B, N, C = x.shape
        q, k, v = self.to_q(x), self.to_k(ctx), self.to_v(ctx)
        sp = lambda t: t.view(B, -1, self.heads, C // self.heads).transpose(1, 2)  # noqa: E731
        o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v))
        return self.to_out[0](o.transpose(1, 2).reshape(B, N, C))


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, 2 * inner)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.attn1 = Attention(dim, dim, heads)
        self.attn2 = Attention(dim, ctx_dim, heads)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), ctx)
        return x + self.ff(self.norm3(x))


class Transformer2DModel(nn.Module):
    def __init__(self, dim, ctx_dim, heads, groups, layers=1):
        super().__init__()
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, ctx_dim, heads) for _ in range(layers)])
        self.proj_out = nn.Linear(dim, dim)

    def forward(self, x, ctx):
        B, C, H, W = x.shape
        h = self.proj_in(self.norm(x).permute(0, 2, 3, 1).reshape(B, H * W, C))
        for blk in self.transformer_blocks:
            h = blk(h, ctx)
        return self.proj_out(h).reshape(B, H, W, C).permute(0, 3, 1, 2) + x


class Sampler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)


class Block(nn.Module):
    """CrossAttnDownBlock2D / DownBlock2D / CrossAttnUpBlock2D / UpBlock2D (attentions / samplers optional)."""

    def __init__(self, chans, temb, groups, eps, attn=None, down=False, up=False):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ci, co, temb, groups, eps) for ci, co in chans])
        if attn is not None:
            self.attentions = nn.ModuleList([Transformer2DModel(co, attn[0], attn[1], groups) for _, co in chans])
        if down:
            self.downsamplers = nn.ModuleList([Sampler(chans[-1][1])])
        if up:
            self.upsamplers = nn.ModuleList([Sampler(chans[-1][1])])


class Mid(nn.Module):
    def __init__(self, c, temb, groups, eps, ctx_dim, heads):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(c, ctx_dim, heads, groups)])

    def forward(self, h, temb, ctx):
        return self.resnets[1](self.attentions[0](self.resnets[0](h, temb), ctx), temb)


class TimeEmb(nn.Module):
    def __init__(self, c0, temb):
        super().__init__()
        self.linear_1, self.linear_2 = nn.Linear(c0, temb), nn.Linear(temb, temb)

    def forward(self, t, c0):
        half = c0 // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        ang = t.float()[:, None] * freqs[None]
        return self.linear_2(F.silu(self.linear_1(torch.cat([ang.cos(), ang.sin()], dim=-1))))  # flip_sin_to_cos: cos first


class EncoderMixin(nn.Module):
    def build_encoder(self, cfg):
        boc, L = cfg["block_out_channels"], cfg["layers_per_block"]
        G, eps, ctx = cfg["norm_num_groups"], cfg["norm_eps"], cfg["cross_attention_dim"]
        temb = boc[0] * 4
        self.c0 = boc[0]
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        self.time_embedding = TimeEmb(boc[0], temb)
        self.down_blocks = nn.ModuleList()
        cout = boc[0]
        for i, bt in enumerate(cfg["down_block_types"]):
            cin, cout = cout, boc[i]
            chans = [(cin if j == 0 else cout, cout) for j in range(L)]
            self.down_blocks.append(Block(chans, temb, G, eps, (ctx, cfg["attention_head_dim"][i]) if bt.startswith("CrossAttn") else None,
                                          down=i != len(boc) - 1))
        self.mid_block = Mid(boc[-1], temb, G, eps, ctx, cfg["attention_head_dim"][-1])

    def encode(self, h, temb, ctx):
        skips = [h]
        for blk in self.down_blocks:
            for j, r in enumerate(blk.resnets):
                h = r(h, temb)
                if hasattr(blk, "attentions"):
                    h = blk.attentions[j](h, ctx)
                skips.append(h)
            if hasattr(blk, "downsamplers"):
                h = F.conv2d(h, blk.downsamplers[0].conv.weight, blk.downsamplers[0].conv.bias, stride=2, padding=1)
                skips.append(h)
        return h, skips


class UNet(EncoderMixin):
    def __init__(self, cfg):
        super().__init__()
        self.build_encoder(cfg)
        boc, L = cfg["block_out_channels"], cfg["layers_per_block"]
        G, eps, ctx = cfg["norm_num_groups"], cfg["norm_eps"], cfg["cross_attention_dim"]
        temb, rev = boc[0] * 4, list(reversed(boc))
        rheads = list(reversed(cfg["attention_head_dim"]))
        self.up_blocks = nn.ModuleList()
        cout = rev[0]
        for i, bt in enumerate(cfg["up_block_types"]):
            prev, cout = cout, rev[i]
            cin = rev[min(i + 1, len(boc) - 1)]
            chans = [((prev if j == 0 else cout) + (cin if j == L else cout), cout) for j in range(L + 1)]
            self.up_blocks.append(Block(chans, temb, G, eps, (ctx, rheads[i]) if bt.startswith("CrossAttn") else None, up=i != len(boc) - 1))
        self.conv_norm_out = nn.GroupNorm(G, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)

    def forward(self, x, t, ctx, down_res=None, mid_res=None):
        temb = self.time_embedding(t, self.c0)
        h, skips = self.encode(self.conv_in(x), temb, ctx)
        h = self.mid_block(h, temb, ctx)
        if down_res is not None:
            skips = [s + r for s, r in zip(skips, down_res)]
            h = h + mid_res
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                h = r(torch.cat([h, skips.pop()], dim=1), temb)
                if hasattr(blk, "attentions"):
                    h = blk.attentions[j](h, ctx)
            if hasattr(blk, "upsamplers"):
                h = blk.upsamplers[0].conv(F.interpolate(h, scale_factor=2.0, mode="nearest"))
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class CondEmbedding(nn.Module):
    def __init__(self, cin, ce, cout):
        super().__init__()
        self.conv_in = nn.Conv2d(cin, ce[0], 3, padding=1)
        self.blocks = nn.ModuleList()
        for i in range(len(ce) - 1):
            self.blocks.append(nn.Conv2d(ce[i], ce[i], 3, padding=1))
            self.blocks.append(nn.Conv2d(ce[i], ce[i + 1], 3, padding=1, stride=2))
        self.conv_out = nn.Conv2d(ce[-1], cout, 3, padding=1)

    def forward(self, c):
        h = F.silu(self.conv_in(c))
        for b in self.blocks:
            h = F.silu(b(h))
        return self.conv_out(h)


class ControlNet(EncoderMixin):
    def __init__(self, cfg):
        super().__init__()
        self.build_encoder(cfg)
        self.controlnet_cond_embedding = CondEmbedding(cfg["conditioning_channels"], cfg["conditioning_embedding_out_channels"],
                                                       cfg["block_out_channels"][0])
        self.controlnet_down_blocks = nn.ModuleList([nn.Conv2d(c, c, 1) for c in schema.controlnet_skip_channels(cfg)])
        self.controlnet_mid_block = nn.Conv2d(cfg["block_out_channels"][-1], cfg["block_out_channels"][-1], 1)

    def forward(self, x, t, ctx, cond):
        temb = self.time_embedding(t, self.c0)
        h, skips = self.encode(self.conv_in(x) + self.controlnet_cond_embedding(cond), temb, ctx)
        h = self.mid_block(h, temb, ctx)
        return [z(s) for z, s in zip(self.controlnet_down_blocks, skips)], self.controlnet_mid_block(h)


if __name__ == "__main__":
    fam = configs.family("tiny")
    ucfg, ccfg = fam["unet"], fam["controlnet"]
    usd = weights.synth_state_dict(schema.unet_schema(ucfg), seed=21)
    csd = weights.synth_state_dict(schema.controlnet_schema(ccfg), seed=22)
    unet, cn = UNet(ucfg).eval(), ControlNet(ccfg)
    unet.load_state_dict(usd, strict=True)   # every diffusers key of the schema must exist here with the same shape
    cn.load_state_dict(csd, strict=True)
    unet.requires_grad_(False)

    g = torch.Generator().manual_seed(23)
    B, hw = 2, 16
    # inputs are f16-representable (the HIP path takes f16 tensors / uint8 images) and stored compactly
    x = torch.randn(B, 4, hw, hw, generator=g).half().float()
    t = torch.tensor([999.0, 199.0])
    ctx = (torch.randn(B, 77, ucfg["cross_attention_dim"], generator=g) * 0.5).half().float()
    cond = (torch.from_numpy(pattern_u8((B, 3, 8 * hw, 8 * hw), salt=1)).float() / 255.0).half().float()
    with torch.no_grad():
        down, mid = cn(x, t, ctx, cond)
        eps = unet(x, t, ctx, down, mid)
        eps_plain = unet(x, t, ctx)
    out = {"x": x.half().numpy(), "t": t.numpy(), "ctx": ctx.half().numpy(), "eps": eps.numpy(), "eps_no_controlnet": eps_plain.numpy(),
           "down0": down[0].numpy(), "down11": down[-1].numpy(), "mid": mid.numpy(),
           "down_sums": np.array([[float(d.double().sum()), float((d.double() ** 2).sum())] for d in down]),
           "seeds": np.array([21, 22, 23])}

    # ---- one fine-tune step (diffusion/train_controlnet_genima.py:1359-1408) through autograd over the module route --------------
    hw2 = 32  # the HIP attention backward wants token counts in multiples of 8: 32 x 32 latents leave 4 x 4 at the deepest level
    lat = (torch.randn(B, 4, hw2, hw2, generator=g) * 0.8).half().float()
    noise = torch.randn(B, 4, hw2, hw2, generator=g).half().float()
    cond = (torch.from_numpy(pattern_u8((B, 3, 8 * hw2, 8 * hw2), salt=2)).float() / 255.0).half().float()
    tt = torch.tensor([301.0, 744.0])
    sa, s1 = torch.tensor([0.8, 0.5]), torch.tensor([0.6, 0.866])
    noisy = sa.view(-1, 1, 1, 1) * lat + s1.view(-1, 1, 1, 1) * noise
    cn.train()
    d2, m2 = cn(noisy, tt, ctx, cond)
    pred = unet(noisy, tt, ctx, d2, m2)
    loss = F.mse_loss(pred.float(), noise.float(), reduction="mean")
    loss.backward()
    names = [n for n, _ in cn.named_parameters()]
    gnorms = np.array([float(p.grad.double().norm()) for _, p in cn.named_parameters()])
    total = float(torch.nn.utils.clip_grad_norm_(cn.parameters(), 1.0))
    keep = ["controlnet_down_blocks.3.weight", "conv_in.weight", "controlnet_cond_embedding.conv_in.weight",
            "down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k.weight", "mid_block.resnets.1.time_emb_proj.bias"]
    before = {k: dict(cn.named_parameters())[k].detach().clone() for k in keep}
    grads = {k: dict(cn.named_parameters())[k].grad.detach().clone() for k in keep}  # after the clip (scaled by 1/total)
    torch.optim.AdamW(cn.parameters(), lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8).step()
    out.update(train_latents=lat.half().numpy(), train_noise=noise.half().numpy(), train_t=tt.numpy(), train_sqrt_ac=sa.numpy(), train_sqrt_1mac=s1.numpy(),
               train_loss=np.array(float(loss.detach())), train_pred=pred.detach().numpy(), train_grad_norm=np.array(total),
               train_grad_names=np.array(names), train_grad_norms=gnorms)
    for k in keep:
        out["train_clipped_grad/" + k] = grads[k].numpy()
        out["train_update/" + k] = (dict(cn.named_parameters())[k].detach() - before[k]).numpy()
    np.savez_compressed(os.path.join(HERE, "unet_golden.npz"), **{k: (v.astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float64 and k not in ("down_sums", "train_grad_norms") else v) for k, v in out.items()})
    print({k: getattr(v, "shape", None) for k, v in out.items() if not k.startswith("train_clipped")}, float(loss), total)
