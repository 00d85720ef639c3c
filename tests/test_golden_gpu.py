"""-m gpu: the HIP path against the committed golden vectors (tests/golden/*.npz; provenance in tests/test_golden_cpu.py): the
AutoencoderKL against transformers' Janus VQ-VAE blocks, the ACT controller against transformers' ResNet + DETR layers, UNet +
ControlNet + the fine-tune step against the nn.Module route.  Weights are the goldens' seeds rounded to f16 (what the device
holds), inputs are f16-representable, so the only difference left is f16 storage of activations: the bars are the network bars
of test_models_gpu.py (rel-L2 <= 3e-3) and of test_training_gpu.py."""
import os
import sys

import numpy as np
import pytest
import torch

from genima_amd import configs, schema, weights
from util import rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
from inputs import pattern_u8  # noqa: E402


def _load(name):
    return np.load(os.path.join(GOLD, name))


def test_vae_hip_vs_janus_golden():
    from genima_amd.host import AutoencoderKL

    g = _load("vae_golden.npz")
    cfg = configs.TINY_VAE
    sd = weights.synth_state_dict(schema.vae_schema(cfg), seed=int(g["seed"]))  # fp32 masters; the module rounds to f16 when packing
    vae = AutoencoderKL(cfg, sd).to("cuda")
    img = vae.decode(torch.from_numpy(g["z"]).half()).sample.float().cpu()
    e = rel_l2(img, torch.from_numpy(g["decoded"]))
    dist = vae.encode(torch.from_numpy(g["x"]).half()).latent_dist
    mom = torch.cat([dist.mean, dist.logvar], 1).float().cpu()
    e2 = rel_l2(mom, torch.from_numpy(g["moments"]))
    print(f"VAE decode vs Janus golden {e:.2e}, encode moments {e2:.2e}")
    assert e < 3e-3 and e2 < 3e-3


def test_act_hip_vs_hf_golden():
    from genima_amd.act import GenimaACT, act_schema

    g = _load("act_golden.npz")
    cfg = dict(configs.TINY_ACT_POLICY, use_lang_cond=False)
    sd = weights.synth_state_dict(act_schema(cfg), seed=int(g["seed"]))
    agent = GenimaACT(cfg, sd, configs.TINY_ACT_CLIP_TEXT, None, device="cuda")
    images = torch.from_numpy(g["images"])  # [B, V, 3, S, S] uint8
    cams = ["left_shoulder", "right_shoulder", "front", "wrist"]
    obs = {f"{c}_rgb": images[:, i:i + 1] for i, c in enumerate(cams)}
    obs["low_dim_state"] = torch.from_numpy(g["qpos"])[:, None]
    a = agent.act(obs, step=0, eval_mode=True).cpu()
    e = rel_l2(a, torch.from_numpy(g["a_hat"]))
    print(f"ACT a_hat vs HF ResNet + DETR golden {e:.2e}")
    assert a.shape == (2, 20, 8) and e < 4e-3


def _nets():
    from genima_amd.host import ControlNetModel, UNet2DConditionModel

    g = _load("unet_golden.npz")
    fam = configs.family("tiny")
    usd = weights.synth_state_dict(schema.unet_schema(fam["unet"]), seed=int(g["seeds"][0]))
    csd = weights.synth_state_dict(schema.controlnet_schema(fam["controlnet"]), seed=int(g["seeds"][1]))
    return g, fam, usd, csd, UNet2DConditionModel(fam["unet"], usd).to("cuda"), ControlNetModel(fam["controlnet"], csd).to("cuda")


def _cond(hw, salt):
    return (torch.from_numpy(pattern_u8((2, 3, 8 * hw, 8 * hw), salt)).float() / 255.0).half()


def test_unet_controlnet_hip_vs_module_golden():
    g, fam, usd, csd, unet, cn = _nets()
    T = lambda k: torch.from_numpy(g[k])  # noqa: E731
    x, t, ctx = T("x").half(), T("t").float(), T("ctx").half()
    down, mid = cn(x, t, encoder_hidden_states=ctx, controlnet_cond=_cond(16, 1), return_dict=False)
    eps = unet(x, t, encoder_hidden_states=ctx, down_block_additional_residuals=down, mid_block_additional_residual=mid).sample
    eps0 = unet(x, t, encoder_hidden_states=ctx).sample
    errs = {"eps": rel_l2(eps, T("eps")), "eps_no_cn": rel_l2(eps0, T("eps_no_controlnet")), "down0": rel_l2(down[0], T("down0")),
            "down11": rel_l2(down[-1], T("down11")), "mid": rel_l2(mid, T("mid"))}
    print("UNet / ControlNet vs module-route golden:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert len(down) == 12 and all(v < 3e-3 for v in errs.values()), errs
    ss = np.array([float((d.double() ** 2).sum()) for d in down])
    assert np.abs(ss / g["down_sums"][:, 1] - 1).max() < 1e-2  # all 12 residuals, by energy


def test_train_step_hip_vs_module_golden():
    from genima_amd.engine import Engine
    from genima_amd.host import nchw_to_nhwc
    from genima_amd.packing import pack_state_dict
    from genima_amd.training import ControlNetTrainer

    g, fam, usd, csd, unet, _ = _nets()
    T = lambda k: torch.from_numpy(g[k])  # noqa: E731
    S = 4096.0
    tr = ControlNetTrainer(Engine("cuda:0"), fam["unet"], fam["controlnet"], unet.W, csd, lr=1e-5, loss_scale=S)
    dev = lambda x: x.cuda()  # noqa: E731
    args = (dev(nchw_to_nhwc(T("train_latents").half(), 8)), dev(nchw_to_nhwc(T("train_noise").half(), 8)), dev(T("train_t").float()),
            dev(T("train_sqrt_ac").float()), dev(T("train_sqrt_1mac").float()), dev(T("ctx").half()), dev(nchw_to_nhwc(_cond(32, 2), 8)))
    loss = float(tr.forward_backward(*args).cpu())
    pred = tr.last["pred"][..., :4].permute(0, 3, 1, 2).float().cpu()
    ref_loss = float(g["train_loss"])
    e_pred = rel_l2(pred, T("train_pred"))
    # per-tensor gradient norms in the diffusers naming: un-pack the flat gradient exactly as save_pretrained un-packs the weights
    from genima_amd.packing import unpack_state_dict
    packed = {n: (tr.cn.G[n].float() / S) for n in tr.cn.layout}
    grads = unpack_state_dict(packed, schema.controlnet_schema(fam["controlnet"]), tr.cn.temb_slices)
    names = [str(n) for n in g["train_grad_names"]]
    gn = np.array([float(grads[n].double().norm()) for n in names])
    ref = g["train_grad_norms"]
    total_ref = float(g["train_grad_norm"])
    total = float(np.sqrt((gn ** 2).sum()))
    big = ref > 1e-2 * total_ref
    worst = float(np.abs(gn[big] / ref[big] - 1).max())
    print(f"train step vs module-route golden: loss {loss:.6f} / {ref_loss:.6f}, pred rel-L2 {e_pred:.2e}, |g| {total:.5f} / {total_ref:.5f}, "
          f"worst per-tensor norm deviation (tensors >= 1 % of |g|) {worst:.2e}")
    assert abs(loss - ref_loss) <= 2e-3 * ref_loss and e_pred < 1e-2
    assert abs(total - total_ref) <= 1e-2 * total_ref and worst < 3e-2
    for k in [k[len("train_clipped_grad/"):] for k in g.files if k.startswith("train_clipped_grad/")]:
        want = T("train_clipped_grad/" + k).float() * max(total_ref, 1.0)  # the fixture stores the gradient after clip_grad_norm_(1.0)
        if float(want.norm()) > 1e-3 * total_ref:
            assert rel_l2(grads[k].cpu(), want) < 3e-2, k
    tr.optimizer_step()
    tr.update_scale()
    assert abs(tr.last["grad_norm"] - total_ref) <= 1e-2 * total_ref
