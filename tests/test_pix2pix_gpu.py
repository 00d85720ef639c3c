"""-m gpu: the InstructPix2Pix family (controller/agent/sd_pix2pix_agent.py,
diffusion/train_instruct_pix2pix_genima.py) on the tiny family with an 8-channel conv_in, against oracle/pix2pix_torch.py.

Tolerances: the networks are held to the bars of test_models_gpu.py / test_training_gpu.py (same kernels, same depth); index / mask
logic (conditioning dropout, EMA schedule, checkpoint layout) is exact."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from genima_amd import configs, schema, weights
from genima_amd.engine import Engine
from genima_amd.host import nchw_to_nhwc
from genima_amd.packing import pack_state_dict
from genima_amd.pix2pix import InstructPix2PixTrainer, StableDiffusionInstructPix2PixPipeline, ema_decay_at, expand_conv_in
from genima_amd.scheduler import DDPMScheduler
from oracle import pix2pix_torch as OP
from oracle import sd_torch as O
from util import q16, rel_l2

pytestmark = pytest.mark.gpu
FAM = configs.family("tiny-pix2pix")


def _r16(sd):
    return weights.round_to(sd, torch.float16)


def _pipe(seed=20):
    pipe = StableDiffusionInstructPix2PixPipeline.from_synthetic(FAM, seed=seed)
    for m in (pipe.vae, pipe.text_encoder, pipe.unet):
        m.load_state_dict(_r16(m.state_dict()))
    return pipe.to("cuda")


@pytest.mark.parametrize("guidance", [0.0, 2.5])
def test_pix2pix_pipeline_vs_oracle(guidance):
    pipe = _pipe()
    B, steps, R = 2, 3, 128
    img_u8 = torch.from_numpy(weights.counter_bytes(5, "p2p", B * R * R * 3).reshape(B, R, R, 3))
    ids = pipe.encode_ids(["tiled perspectives of a robot arm executing 'open the box'"] * B)
    neg = pipe.encode_ids(["monochrome, lowres, bad anatomy, worst quality, low quality"] * B)
    lat = q16(torch.randn(B, 4, R // 8, R // 8, generator=torch.Generator().manual_seed(2)))
    out = pipe(prompt_ids=ids, image=img_u8, negative_prompt=["monochrome, lowres, bad anatomy, worst quality, low quality"] * B,
               num_inference_steps=steps, guidance_scale=guidance, image_guidance_scale=1.5, latents=lat.half(), output_type="np")
    u8 = out.images
    assert u8.shape == (B, R, R, 3) and u8.dtype == np.uint8
    io = pipe.program(B, R, R, steps, (guidance, 1.5) if guidance > 1 else None)
    lat_hip = io.latents.permute(0, 3, 1, 2).float().cpu()
    img01 = q16(img_u8.permute(0, 3, 1, 2).float() / 255.0)
    args = (pipe.unet.state_dict(), pipe.unet.config, pipe.vae.state_dict(), pipe.vae.config, pipe.text_encoder.state_dict(),
            pipe.text_encoder.config, configs.SD_TURBO_SCHEDULER, ids.long(), neg.long(), img01, lat, steps, guidance, 1.5)
    with torch.no_grad():
        x16, im16 = OP.pipeline(*args, q=q16)
        x32, im32 = OP.pipeline(*args)
    e16, e32, eref = rel_l2(lat_hip, x16), rel_l2(lat_hip, x32), rel_l2(x16, x32)
    ref_u8 = O.vae_postprocess_u8(im16).numpy()
    d = np.abs(u8.astype(np.int32) - ref_u8.astype(np.int32))
    print(f"pix2pix pipeline (guidance {guidance}): latents vs f16-storage oracle {e16:.2e}, vs fp32 {e32:.2e} (oracle16 vs 32 {eref:.2e}); "
          f"uint8 mean|diff| {d.mean():.3f} max {d.max()}")
    assert e32 <= max(1.5 * eref + 5e-4, 3e-3) and e16 < 6e-3
    assert d.mean() < 0.6 and (d > 3).mean() < 1e-2
    out2 = pipe(prompt_ids=ids, image=img_u8, negative_prompt=["x"] * B if guidance > 1 else None, num_inference_steps=steps,
                guidance_scale=guidance, latents=lat.half())
    assert out2[0][0].size == (R, R)
    if guidance <= 1:  # same inputs -> bit-identical replay
        assert np.array_equal(np.asarray(out2.images[1]), u8[1])


def _train_setup(B=2, seed=0):
    ucfg = FAM["unet"]
    base = weights.synth_state_dict(schema.unet_schema(dict(ucfg, in_channels=4)), 1)
    usd = expand_conv_in(base, 8)
    assert usd["conv_in.weight"].shape[1] == 8 and float(usd["conv_in.weight"][:, 4:].abs().max()) == 0.0
    assert torch.equal(usd["conv_in.weight"][:, :4], base["conv_in.weight"])
    # the added channels start at zero in the reference; give them weight here so their gradient path is checked on live values
    usd["conv_in.weight"][:, 4:] = weights.synth_state_dict({"w": tuple(usd["conv_in.weight"][:, 4:].shape)}, 9)["w"] * 0.5
    usd = _r16(usd)
    g = torch.Generator().manual_seed(seed)
    lat, noise = q16(torch.randn(B, 4, 32, 32, generator=g)), q16(torch.randn(B, 4, 32, 32, generator=g))
    ctx, emb = q16(torch.randn(B, 77, 128, generator=g)), q16(torch.randn(B, 4, 32, 32, generator=g))
    t = torch.tensor([801, 399][:B])
    sa, s1 = DDPMScheduler().add_noise_coeffs(t)
    return ucfg, usd, lat, noise, ctx, emb, t, sa, s1


def test_pix2pix_train_step_vs_autograd_oracle():
    ucfg, usd, lat, noise, ctx, emb, t, sa, s1 = _train_setup()
    E = Engine("cuda:0")
    S = 4096.0
    tr = InstructPix2PixTrainer(E, ucfg, usd, lr=1e-4, loss_scale=S, use_ema=True)
    dev = lambda x: x.cuda()  # noqa: E731
    args = (dev(nchw_to_nhwc(lat, 8).half()), dev(nchw_to_nhwc(noise, 8).half()), dev(t.float()), dev(sa), dev(s1), dev(ctx.half()),
            dev(nchw_to_nhwc(emb, 8).half()))
    loss = float(tr.forward_backward(*args).cpu())
    pred = tr.last["pred"][..., :4].permute(0, 3, 1, 2).float().cpu()
    layout = list(tr.cn.layout)
    g_hip = {n: (tr.cn.G[n].float() / S).cpu() for n in layout}
    l32, g32, p32 = OP.train_forward_backward(usd, ucfg, lat, noise, t.float(), sa, s1, ctx, emb)
    l16, g16, p16 = OP.train_forward_backward(usd, ucfg, lat, noise, t.float(), sa, s1, ctx, emb, q=q16)
    P32, P16 = pack_state_dict(g32, "cpu", dtype=torch.float32), pack_state_dict(g16, "cpu", dtype=torch.float32)
    flat = lambda d: torch.cat([d[n].reshape(-1).float().cpu() for n in layout])  # noqa: E731
    f_hip, f32_, f16_ = flat(g_hip), flat(P32), flat(P16)
    gnorm = float(f32_.norm())
    e_pred, e_pref, e_all, e_ref = rel_l2(pred, p32), rel_l2(p16, p32), rel_l2(f_hip, f32_), rel_l2(f16_, f32_)
    print(f"pix2pix step: loss hip {loss:.6f} oracle {float(l32):.6f}; pred rel-L2 {e_pred:.2e} (f16 oracle {e_pref:.2e}); flat gradient "
          f"|g| {gnorm:.3e} rel-L2 {e_all:.2e} (f16 oracle {e_ref:.2e})")
    assert abs(loss - float(l32)) <= 2e-3 * float(l32)
    assert e_pred <= min(1e-2, 1.5 * e_pref + 5e-4)
    assert torch.isfinite(f_hip).all() and e_all <= min(1e-2, 1.5 * e_ref + 2e-3)
    worst = sorted(((rel_l2(g_hip[n], P32[n].float()), n) for n in layout if float(P32[n].float().norm()) >= 1e-3 * gnorm), reverse=True)
    print("worst per-tensor gradient errors:", [(f"{e:.2e}", n) for e, n in worst[:4]])
    assert worst[0][0] <= 2e-2, worst[:4]
    # the decoder's concat / upsample convs and the 8-channel conv_in (all 8 input channels) receive weight gradients
    for n in ("conv_in.weight", "up_blocks.0.resnets.0.conv1.weight", "up_blocks.0.upsamplers.0.conv.weight", "up_blocks.1.resnets.0.conv_shortcut.weight"):
        assert n in g_hip and float(g_hip[n].abs().max()) > 0 and rel_l2(g_hip[n], P32[n].float()) < 2e-2, n
    dead = [n for n in layout if float(P32[n].abs().max()) > 1e-7 * gnorm and float(g_hip[n].abs().max()) == 0.0]
    assert not dead, dead

    # ---- optimizer + EMA: EMAModel's first step copies the parameters (decay 0), later ones follow (1 + n) / (10 + n)
    assert [ema_decay_at(k) for k in (1, 2, 3)] == [OP.ema_decay(k) for k in (1, 2, 3)] == [0.0, 2 / 11, 3 / 12]
    tr.optimizer_step(); tr.update_scale()
    tr.cn.zero_grad()
    shadow0 = tr.ema.clone()
    for k in (1, 2, 3):
        tr.step(*args)
        want = OP.ema_step(shadow0.cpu(), tr.cn.master.cpu(), k)
        assert torch.allclose(tr.ema.cpu(), want, rtol=0, atol=1e-7), k
        shadow0 = tr.ema.clone()
    assert tr.ema_steps == 3 and not torch.equal(tr.ema, tr.cn.master)


def test_pix2pix_train_step_from_batch_dropout_and_checkpoints(tmp_path):
    """The whole step body from a collated batch (VAE encode sample / mode, CLIP, conditioning dropout), the checkpoint layout
    (``checkpoint-N/unet`` + ``unet_ema``) and the agent that loads it (controller/agent/sd_pix2pix_agent.py:18-41)."""
    from genima_amd.agent import SDPix2PixAgent
    from genima_amd.pipeline import HashTokenizer

    ucfg, usd, *_ = _train_setup()
    E = Engine("cuda:0")
    synth = lambda sch, s: weights.synth_state_dict(sch, s)  # noqa: E731
    vae_W, text_W = pack_state_dict(synth(schema.vae_schema(FAM["vae"]), 3), "cuda"), pack_state_dict(synth(schema.clip_text_schema(FAM["text"]), 4), "cuda")
    tok = HashTokenizer(FAM["text"]["vocab_size"])

    def trainer(cdp):
        tr = InstructPix2PixTrainer(E, ucfg, usd, lr=1e-4, use_ema=True, conditioning_dropout_prob=cdp)
        tr.attach_frozen(FAM["vae"], vae_W, FAM["text"], text_W, DDPMScheduler(), seed=5)
        tr.set_null_prompt(tok([""], return_tensors="pt").input_ids)
        return tr

    B, R = 4, 256
    g = torch.Generator().manual_seed(3)
    batch = dict(original_pixel_values=torch.rand(B, 3, R, R, generator=g) * 2 - 1, edited_pixel_values=torch.rand(B, 3, R, R, generator=g) * 2 - 1,
                 input_ids=tok(["open the box"] * B, return_tensors="pt").input_ids)
    tr = trainer(0.05)
    # conditioning dropout, exact against the reference's masks (p = 0.25 puts the four samples in the four regimes)
    tr.cdp = 0.25
    ctx = q16(torch.randn(B, 77, FAM["text"]["hidden_size"], generator=g)).cuda().half()
    mom = q16(torch.randn(B, 32, 32, 8, generator=g)).cuda().half()
    rp = torch.tensor([0.1, 0.3, 0.6, 0.9])
    c2, m2 = tr.apply_conditioning_dropout(ctx, mom, rp)
    null = tr._null_ctx.float().cpu()
    wc, wm = OP.conditioning_dropout(ctx.float().cpu(), null.expand(B, -1, -1), mom.float().cpu().permute(0, 3, 1, 2), rp, 0.25)
    assert torch.equal(c2.float().cpu(), wc) and torch.equal(m2.float().cpu().permute(0, 3, 1, 2), wm)
    assert torch.equal(c2[0].float().cpu(), null[0]) and torch.equal(c2[2], ctx[2]) and float(m2[1].abs().max()) == 0 and torch.equal(m2[0], mom[0])
    tr.cdp = 0.05
    losses = [float(tr.train_step(batch)) for _ in range(3)]
    assert all(np.isfinite(losses)) and tr.opt_step >= 2 and tr.ema_steps == 3
    out = str(tmp_path / "run")
    d = tr.save_state(out, 3)
    assert sorted(os.listdir(d)) == ["ema_flat.safetensors", "optimizer_flat.safetensors", "unet", "unet_ema"]
    m3, e3 = tr.cn.master.clone(), tr.ema.clone()
    la = [float(tr.train_step(batch)) for _ in range(2)]
    tr2 = trainer(0.05)
    assert tr2.load_state(d) == 3 and tr2.ema_steps == 3 and tr2.opt_step == tr.opt_step - 2
    assert torch.equal(tr2.cn.master, m3) and torch.equal(tr2.ema, e3), "unet / unet_ema must round-trip bit for bit"
    # the agent resolves <diffusion_ckpt>/checkpoint-<max>/unet by natural sort and runs the pipeline with the fine-tuned 8-channel UNet
    os.makedirs(os.path.join(out, "checkpoint-10"))
    tr.save_pretrained(os.path.join(out, "checkpoint-10", "unet"))
    cfg = SimpleNamespace(sd_ckpt="synthetic:tiny-pix2pix", diffusion_ckpt=out, device="cuda", image_resolution=128, torch_compile=False)
    agent = SDPix2PixAgent(cfg)
    sd = tr.controlnet_state_dict()
    assert all(torch.equal(agent.pipe.unet.state_dict()[k], sd[k].cpu()) for k in sd) and agent.pipe.unet.config["in_channels"] == 8
    from PIL import Image
    img = Image.fromarray(weights.counter_bytes(7, "obs", 128 * 128 * 3).reshape(128, 128, 3))
    res = agent.infer(prompts=["tiled perspectives of a robot arm executing 'open box'"], images=[img], negative_prompts=["lowres"],
                      num_inference_steps=2, guidance_scale=0.0, generator=[torch.Generator().manual_seed(2)])
    assert res[0][0].size == (128, 128)
    with pytest.raises(FileNotFoundError):
        SDPix2PixAgent(SimpleNamespace(sd_ckpt="synthetic:tiny-pix2pix", diffusion_ckpt=str(tmp_path / "nowhere"), device="cuda",
                                       image_resolution=128, torch_compile=False))
    assert len(la) == 2
