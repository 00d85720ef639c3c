"""The CPU oracle (oracle/sd_torch.py, oracle/act_torch.py, oracle/train_torch.py) against the golden vectors under tests/golden/:

  vae_golden.npz   AutoencoderKL encode / decode  <- transformers' Janus VQ-VAE blocks (an independent implementation of the taming /
                   SD VAE ResnetBlock, AttnBlock, Up / Downsample, MidBlock); tests/golden/make_vae_golden.py
  act_golden.npz   ACT controller forward         <- transformers' ResNetModel + DETR encoder / decoder layers + sine positions
                   (independent implementations of the public modules RoboBase vendors); tests/golden/make_act_golden.py
  unet_golden.npz  UNet + ControlNet forward and one fine-tune step <- a second, nn.Module-based route on torch's own SDPA /
                   GroupNorm / autograd / AdamW, ``load_state_dict(strict=True)`` on the diffusers key names; a CONSISTENCY pin only
                   (no installed package implements UNet2DConditionModel); tests/golden/make_unet_golden.py
(clip_text_golden.npz / tiling_golden.npz / scheduler_tables.json: tests/test_cpu_oracle.py)."""
import os

import numpy as np
import torch

from genima_amd import configs, schema, weights
from genima_amd.act import act_schema
from oracle import act_torch as OA
from oracle import sd_torch as O
from oracle import train_torch as OT

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def pattern_u8(shape, salt):
    import sys
    sys.path.insert(0, GOLD)
    from inputs import pattern_u8 as f
    return f(shape, salt)


def test_vae_oracle_matches_janus_vqvae_blocks():
    g = np.load(os.path.join(GOLD, "vae_golden.npz"))
    cfg = configs.TINY_VAE
    sd = weights.synth_state_dict(schema.vae_schema(cfg), seed=int(g["seed"]))
    with torch.no_grad():
        img = O.vae_decode(sd, cfg, torch.from_numpy(g["z"]))
        mean, logvar = O.vae_encode_moments(sd, cfg, torch.from_numpy(g["x"]))
    assert _rel(img, torch.from_numpy(g["decoded"])) < 1e-5
    assert _rel(torch.cat([mean, logvar], 1), torch.from_numpy(g["moments"])) < 1e-5


def test_act_oracle_matches_hf_resnet_and_detr():
    g = np.load(os.path.join(GOLD, "act_golden.npz"))
    cfg = dict(configs.TINY_ACT_POLICY, use_lang_cond=False)
    sd = weights.synth_state_dict(act_schema(cfg), seed=int(g["seed"]))
    images = torch.from_numpy(g["images"])
    with torch.no_grad():
        mean = torch.tensor(OA.IMAGENET_MEAN)[None, :, None, None]
        std = torch.tensor(OA.IMAGENET_STD)[None, :, None, None]
        feat = OA.resnet18_features(sd, (images.float().flatten(0, 1) / 255.0 - mean) / std)
        a_hat, is_pad = OA.act_forward(sd, cfg, images, torch.from_numpy(g["qpos"]), None)
    assert _rel(feat, torch.from_numpy(g["resnet_features"])) < 1e-5
    assert float((OA.sine_pos_embed(2, 2, cfg["hidden_dim"]) - torch.from_numpy(g["pos_cam"])).abs().max()) < 1e-6
    assert _rel(a_hat, torch.from_numpy(g["a_hat"])) < 1e-5 and _rel(is_pad, torch.from_numpy(g["is_pad_hat"])) < 1e-5


def _unet_setup():
    g = np.load(os.path.join(GOLD, "unet_golden.npz"))
    fam = configs.family("tiny")
    usd = weights.synth_state_dict(schema.unet_schema(fam["unet"]), seed=int(g["seeds"][0]))
    csd = weights.synth_state_dict(schema.controlnet_schema(fam["controlnet"]), seed=int(g["seeds"][1]))
    T = lambda k: torch.from_numpy(g[k]).float()  # noqa: E731
    cond = lambda hw, salt: (torch.from_numpy(pattern_u8((2, 3, 8 * hw, 8 * hw), salt)).float() / 255.0).half().float()  # noqa: E731
    return g, fam, usd, csd, T, cond


def test_unet_controlnet_oracle_matches_module_route():
    g, fam, usd, csd, T, cond = _unet_setup()
    with torch.no_grad():
        down, mid = O.controlnet_forward(csd, fam["controlnet"], T("x"), T("t"), T("ctx"), cond(16, 1))
        eps = O.unet_forward(usd, fam["unet"], T("x"), T("t"), T("ctx"), down, mid)
        eps0 = O.unet_forward(usd, fam["unet"], T("x"), T("t"), T("ctx"))
    assert len(down) == 12
    assert _rel(eps, T("eps")) < 2e-5 and _rel(eps0, T("eps_no_controlnet")) < 2e-5
    assert _rel(down[0], T("down0")) < 2e-5 and _rel(down[-1], T("down11")) < 2e-5 and _rel(mid, T("mid")) < 2e-5
    sums = np.array([[float(d.double().sum()), float((d.double() ** 2).sum())] for d in down])
    assert np.abs(sums[:, 1] / g["down_sums"][:, 1] - 1).max() < 1e-4  # every one of the 12 residuals (sum of squares)


def test_train_step_oracle_matches_module_route():
    g, fam, usd, csd, T, cond = _unet_setup()
    loss, grads, pred = OT.train_forward_backward(usd, csd, fam["unet"], fam["controlnet"], T("train_latents"), T("train_noise"), T("train_t"),
                                                  T("train_sqrt_ac"), T("train_sqrt_1mac"), T("ctx"), cond(32, 2))
    assert abs(float(loss) - float(g["train_loss"])) < 1e-5 * float(g["train_loss"]) and _rel(pred, T("train_pred")) < 2e-5
    names = [str(n) for n in g["train_grad_names"]]
    assert sorted(names) == sorted(grads)  # the module route's parameter names ARE the diffusers keys of the schema
    gn = np.array([float(grads[n].double().norm()) for n in names])
    ref = g["train_grad_norms"]
    total = float(np.sqrt((ref ** 2).sum()))
    assert np.abs(gn - ref).max() < 2e-5 * total and abs(float(np.sqrt((gn ** 2).sum())) - float(g["train_grad_norm"])) < 1e-5 * total
    new, norm = OT.adamw_step(csd, grads, 1e-5)
    assert abs(norm - float(g["train_grad_norm"])) < 1e-5 * total
    for k in [k[len("train_update/"):] for k in g.files if k.startswith("train_update/")]:
        clipped = grads[k] / max(norm, 1.0)  # clip_grad_norm_(max_norm=1): scale by 1 / (norm + 1e-6) only when norm > 1
        assert _rel(clipped, T("train_clipped_grad/" + k)) < 5e-5, k
        assert _rel(new[k] - csd[k], T("train_update/" + k)) < 2e-3, k  # Adam's first step ~ lr * sign(g): tiny gradients amplify
