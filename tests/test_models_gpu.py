"""-m gpu: network-level parity of the HIP host classes (tiny family: same topology as SD-Turbo at reduced width) against the
CPU oracle on the same seeded synthetic weights / inputs.  Two comparisons per network:
  * vs the oracle with f16 storage rounding emulated at the same points (isolates kernel error: tight),
  * vs the pure fp32 oracle (includes f16 storage error accumulated through the depth of the network: looser)."""
import numpy as np
import pytest
import torch

from genima_amd import configs, schema, weights
from genima_amd.host import AutoencoderKL, CLIPTextModel, ControlNetModel, UNet2DConditionModel
from oracle import sd_torch as O
from util import q16, rel_l2

pytestmark = pytest.mark.gpu

FAM = configs.family("tiny")


def _r16(sd):
    return weights.round_to(sd, torch.float16)


def _inputs(B=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = q16(torch.randn(B, 4, 16, 16, generator=g))
    ctx = q16(torch.randn(B, 77, 128, generator=g))
    cond = q16(torch.rand(B, 3, 128, 128, generator=g))
    t = torch.tensor([999.0, 399.0][:B])
    return x, ctx, cond, t


def _report(name, y, ref16, ref32, tol16, tol32):
    """Whole-network bar.  Individual kernels meet 1e-3 (test_kernels_gpu.py); through ~100 f16-stored layers the storage
    rounding itself accumulates to ~1.5e-3 of independent noise, which any f16 implementation (the reference's included)
    carries.  So: (a) the HIP output must be as close to the fp32 oracle as the f16-storage oracle is (x1.5 + 5e-4), and
    (b) within tol16 of the f16-storage oracle (two f16 pipelines with different rounding points)."""
    e16, e32, eref = rel_l2(y, ref16), rel_l2(y, ref32), rel_l2(ref16, ref32)
    print(f"{name}: rel-L2 vs f16-storage oracle {e16:.2e}, vs fp32 oracle {e32:.2e} (f16-storage oracle vs fp32: {eref:.2e})")
    assert torch.isfinite(y.float()).all()
    assert e16 <= tol16, f"{name}: {e16:.3e} > {tol16:.1e} (f16-storage oracle)"
    assert e32 <= min(tol32, 1.5 * eref + 5e-4), f"{name}: {e32:.3e} vs fp32 oracle; f16-storage oracle itself is at {eref:.3e}"


def test_clip_text():
    cfg = FAM["text"]
    sd = _r16(weights.synth_state_dict(schema.clip_text_schema(cfg), 4))
    m = CLIPTextModel(cfg, sd).to("cuda")
    V = cfg["vocab_size"]
    ids = torch.zeros(2, 77, dtype=torch.int64)
    ids[0, :14] = torch.tensor([V - 2] + [320 + i for i in range(12)] + [V - 1])
    ids[1, :5] = torch.tensor([V - 2, 7, 8, 9, V - 1])
    y = m(ids)[0].float().cpu()
    with torch.no_grad():
        _report("clip", y, O.clip_text_forward(sd, cfg, ids, q16), O.clip_text_forward(sd, cfg, ids), 1e-3, 3e-3)


def test_unet_and_controlnet():
    ucfg, ccfg = FAM["unet"], FAM["controlnet"]
    usd = _r16(weights.synth_state_dict(schema.unet_schema(ucfg), 1))
    csd = _r16(weights.synth_state_dict(schema.controlnet_schema(ccfg), 2))
    unet, cn = UNet2DConditionModel(ucfg, usd).to("cuda"), ControlNetModel(ccfg, csd).to("cuda")
    x, ctx, cond, t = _inputs()
    down, mid = cn(x.half(), t, ctx.half(), cond.half(), return_dict=False)
    with torch.no_grad():
        d16, m16 = O.controlnet_forward(csd, ccfg, x, t, ctx, cond, q=q16)
        d32, m32 = O.controlnet_forward(csd, ccfg, x, t, ctx, cond)
    for i, (a, b, c) in enumerate(zip(down, d16, d32)):
        _report(f"controlnet down[{i}]", a.float().cpu(), b, c, 3e-3, 1e-2)
    _report("controlnet mid", mid.float().cpu(), m16, m32, 3e-3, 1e-2)
    # UNet fed with the ORACLE's residuals so the two networks are checked independently
    eps = unet(x.half(), t, ctx.half(), [d.half() for d in d16], m16.half()).sample
    with torch.no_grad():
        e16 = O.unet_forward(usd, ucfg, x, t, ctx, [q16(d) for d in d16], q16(m16), q=q16)
        e32 = O.unet_forward(usd, ucfg, x, t, ctx, [q16(d) for d in d16], q16(m16))
    _report("unet eps", eps.float().cpu(), e16, e32, 3e-3, 1e-2)
    # no-residual path
    eps0 = unet(x.half(), t, ctx.half()).sample
    with torch.no_grad():
        _report("unet eps (no controlnet)", eps0.float().cpu(), O.unet_forward(usd, ucfg, x, t, ctx, q=q16),
                O.unet_forward(usd, ucfg, x, t, ctx), 3e-3, 1e-2)


def test_controlnet_from_unet_is_identity_on_unet():
    """from_unet: zero convs -> residuals are exactly zero and encoder weights are the UNet's."""
    unet = UNet2DConditionModel.from_config(FAM["unet"], 1)
    cn = ControlNetModel.from_unet(unet).to("cuda")
    x, ctx, cond, t = _inputs(1)
    down, mid = cn(x.half(), t[:1], ctx.half(), cond.half(), return_dict=False)
    assert all(float(d.abs().max()) == 0.0 for d in down) and float(mid.abs().max()) == 0.0
    assert torch.equal(cn.state_dict()["down_blocks.1.resnets.0.conv1.weight"], unet.state_dict()["down_blocks.1.resnets.0.conv1.weight"])


def test_vae_decode_and_encode():
    cfg = FAM["vae"]
    sd = _r16(weights.synth_state_dict(schema.vae_schema(cfg), 3))
    vae = AutoencoderKL(cfg, sd).to("cuda")
    g = torch.Generator().manual_seed(3)
    z = q16(torch.randn(2, 4, 16, 16, generator=g))
    img = vae.decode(z.half()).sample.float().cpu()
    with torch.no_grad():
        _report("vae decode", img, O.vae_decode(sd, cfg, z, q16), O.vae_decode(sd, cfg, z), 3e-3, 1e-2)
    x = q16(torch.rand(2, 3, 128, 128, generator=g) * 2 - 1)
    dist = vae.encode(x.half()).latent_dist
    with torch.no_grad():
        m16, lv16 = O.vae_encode_moments(sd, cfg, x, q16)
        m32, lv32 = O.vae_encode_moments(sd, cfg, x)
    _report("vae encode mean", dist.mean.cpu(), m16, m32, 3e-3, 1e-2)
    _report("vae encode logvar", dist.logvar.cpu(), lv16, lv32, 3e-3, 1e-2)


def test_vae_stream_scaling_is_exact_and_survives_an_out_of_range_stream():
    """AutoencoderKL.enable_stream_scaling (the f16 answer to diffusers' force_upcast / upcast_vae, packing.scale_vae_stream): (a) on an
    ordinary VAE the scaled-stream decode / encode equal the fp32 oracle ON THE ORIGINAL WEIGHTS as well as the plain path does -- the
    re-parametrisation changes nothing but rounding; (b) on a VAE whose residual stream leaves f16's range (conv_in x 3e4, the stock SDXL
    VAE's failure) the plain f16 decode is not finite, the scaled one still matches the fp32 oracle."""
    cfg = FAM["vae"]
    sd = _r16(weights.synth_state_dict(schema.vae_schema(cfg), 3))
    g = torch.Generator().manual_seed(5)
    z = q16(torch.randn(2, 4, 16, 16, generator=g))
    x = q16(torch.rand(2, 3, 128, 128, generator=g) * 2 - 1)
    vae = AutoencoderKL(cfg, sd).to("cuda").enable_stream_scaling()
    assert vae.W["__meta__"]["vae_stream_scale"] == 1.0 / 64.0
    with torch.no_grad():
        _report("vae decode, scaled stream", vae.decode(z.half()).sample.float().cpu(), O.vae_decode(sd, cfg, z, q16), O.vae_decode(sd, cfg, z), 3e-3, 1e-2)
        m16, _ = O.vae_encode_moments(sd, cfg, x, q16)
        m32, _ = O.vae_encode_moments(sd, cfg, x)
    _report("vae encode mean, scaled stream", vae.encode(x.half()).latent_dist.mean.cpu(), m16, m32, 3e-3, 1e-2)
    big = dict(sd)
    for k in ("decoder.conv_in.weight", "decoder.conv_in.bias"):
        big[k] = sd[k] * 3.0e4
    with torch.no_grad():
        ref = O.vae_decode(big, cfg, z)
    plain = AutoencoderKL(cfg, big).to("cuda").decode(z.half()).sample.float().cpu()
    assert not torch.isfinite(plain).all() or rel_l2(plain, ref) > 0.1, "the test VAE was meant to overflow plain f16"
    scaled = AutoencoderKL(cfg, big).to("cuda").enable_stream_scaling().decode(z.half()).sample.float().cpu()
    assert torch.isfinite(scaled).all()
    e = rel_l2(scaled, ref)
    print(f"vae decode, stream ~1e5, scaled by 1/64: rel-L2 vs fp32 oracle {e:.2e}")
    assert e <= 5e-3, e


def _oracle_pipeline(pipe, ids, img_u8, latents, steps, q, guidance=None, neg_ids=None):
    """Appendix D loop on the CPU oracle with the same weights.  guidance: diffusers' classifier-free guidance -- both networks on the
    negative and on the positive prompt, noise_pred = uncond + guidance * (text - uncond)."""
    from oracle import scheduler as OS

    sd_t, sd_c, sd_u, sd_v = (m.state_dict() for m in (pipe.text_encoder, pipe.controlnet, pipe.unet, pipe.vae))
    ts, sig, init = OS.euler_set_timesteps(configs.SD_TURBO_SCHEDULER, steps)
    ctx = O.clip_text_forward(sd_t, pipe.text_encoder.config, ids, q)
    nctx = O.clip_text_forward(sd_t, pipe.text_encoder.config, neg_ids, q) if guidance else None
    cond = q(img_u8.permute(0, 3, 1, 2).float() / 255.0)
    x = q(latents * init)
    for i in range(steps):
        xs = q(x / float((sig[i] ** 2 + 1) ** 0.5))
        t = torch.full((x.shape[0],), float(ts[i]))
        down, mid = O.controlnet_forward(sd_c, pipe.controlnet.config, xs, t, ctx, cond, q=q)
        eps = O.unet_forward(sd_u, pipe.unet.config, xs, t, ctx, down, mid, q=q)
        if guidance:
            down, mid = O.controlnet_forward(sd_c, pipe.controlnet.config, xs, t, nctx, cond, q=q)
            eps_u = O.unet_forward(sd_u, pipe.unet.config, xs, t, nctx, down, mid, q=q)
            eps = q(eps_u + guidance * (eps - eps_u))
        x = q(torch.from_numpy(OS.euler_step(eps.numpy(), float(sig[i]), float(sig[i + 1]), x.numpy())))
    img = O.vae_decode(sd_v, pipe.vae.config, q(x / pipe.vae.config["scaling_factor"]), q)
    return x, img


def test_pipeline_classifier_free_guidance():
    """guidance_scale > 1 (diffusers' do_classifier_free_guidance; the reference's configs run 0.0): negative | positive prompt rows through
    the ControlNet and the UNet, eps = uncond + g (text - uncond), against the oracle loop; guidance 0.0 afterwards still replays the
    single-batch program."""
    from genima_amd.pipeline import StableDiffusionControlNetPipeline

    pipe = StableDiffusionControlNetPipeline.from_synthetic(FAM, seed=21)
    for m in (pipe.vae, pipe.text_encoder, pipe.unet, pipe.controlnet):
        m.load_state_dict(_r16(m.state_dict()))
    cn = pipe.controlnet.state_dict()  # the synthetic ControlNet's zero convs are zero: give the residuals something to say
    for k in cn:
        if k.startswith(("controlnet_down_blocks", "controlnet_mid_block")) and k.endswith("weight"):
            cn[k] = q16(torch.randn(cn[k].shape, generator=torch.Generator().manual_seed(len(k))) * 0.05)
    pipe.controlnet.load_state_dict(cn)
    pipe.to("cuda")
    B, steps, g_scale = 2, 3, 3.0
    img_u8 = torch.from_numpy(weights.counter_bytes(3, "ctrl", B * 128 * 128 * 3).reshape(B, 128, 128, 3))
    prompts, negs = ["tiled perspectives of a robot arm executing 'open the box'"] * B, ["blurry"] * B
    ids, nids = pipe.encode_ids(prompts), pipe.encode_ids(negs)
    lat = q16(torch.randn(B, 4, 16, 16, generator=torch.Generator().manual_seed(4)))
    out = pipe(prompt=prompts, negative_prompt="blurry", image=img_u8, num_inference_steps=steps, guidance_scale=g_scale, latents=lat.half(), output_type="latent")
    with torch.no_grad():
        x16, _ = _oracle_pipeline(pipe, ids, img_u8, lat, steps, q16, guidance=g_scale, neg_ids=nids)
        x32, _ = _oracle_pipeline(pipe, ids, img_u8, lat, steps, lambda t: t, guidance=g_scale, neg_ids=nids)
        x_plain, _ = _oracle_pipeline(pipe, ids, img_u8, lat, steps, lambda t: t)
    _report("pipeline latents, guidance 3", out.images.float().cpu(), x16, x32, 6e-3, 3e-2)
    assert rel_l2(x32, x_plain) > 5e-2, "the guided and the unguided trajectories should differ for this test to mean anything"
    out0 = pipe(prompt=prompts, image=img_u8, num_inference_steps=steps, guidance_scale=0.0, latents=lat.half(), output_type="latent")
    assert rel_l2(out0.images.float().cpu(), x_plain) < 6e-3


@pytest.mark.parametrize("graph", [False, True])
def test_pipeline_end_to_end(graph):
    from genima_amd.pipeline import StableDiffusionControlNetPipeline

    pipe = StableDiffusionControlNetPipeline.from_synthetic(FAM, seed=20)
    for m in (pipe.vae, pipe.text_encoder, pipe.unet, pipe.controlnet):  # f16-representable master weights
        m.load_state_dict(_r16(m.state_dict()))
    pipe.to("cuda")
    pipe.enable_hip_graph(graph)
    B, steps = 2, 5
    img_u8 = torch.from_numpy(weights.counter_bytes(3, "ctrl", B * 128 * 128 * 3).reshape(B, 128, 128, 3))
    ids = pipe.encode_ids(["tiled perspectives of a robot arm executing 'open the box'"] * B)
    g = torch.Generator().manual_seed(2)
    lat = q16(torch.randn(B, 4, 16, 16, generator=g))
    out = pipe(prompt_ids=ids, image=img_u8, num_inference_steps=steps, guidance_scale=0.0, latents=lat.half(), output_type="np")
    u8 = out.images
    assert u8.shape == (B, 128, 128, 3) and u8.dtype == np.uint8
    assert pipe.scheduler.timesteps.to(torch.int64).tolist() == [999, 799, 599, 399, 199]  # bit-exact indices
    lat_hip = pipe.program(B, 128, 128, steps).latents.permute(0, 3, 1, 2).float().cpu()
    with torch.no_grad():
        x16, img16 = _oracle_pipeline(pipe, ids, img_u8, lat, steps, q16)
        x32, img32 = _oracle_pipeline(pipe, ids, img_u8, lat, steps, lambda t: t)
    _report("pipeline latents", lat_hip, x16, x32, 5e-3, 3e-2)
    ref_u8 = O.vae_postprocess_u8(img16).numpy()
    d = np.abs(u8.astype(np.int32) - ref_u8.astype(np.int32))
    print(f"pipeline uint8: max |diff| {d.max()}, mean |diff| {d.mean():.4f}, >1 LSB: {(d > 1).mean():.2e}")
    assert d.mean() < 0.5 and (d > 2).mean() < 1e-2
    # PIL surface + determinism of a second call (same injected latents)
    out2 = pipe(prompt_ids=ids, image=img_u8, num_inference_steps=steps, guidance_scale=0.0, latents=lat.half())
    assert out2[0][0].size == (128, 128) and np.array_equal(np.asarray(out2.images[1]), u8[1])


def test_zero_convs_fused_behind_the_join_match_the_add_launch():
    """The ControlNet's zero convs with the UNet's skip / mid tensor as residual operand (default; dealt over both streams at small batch) against the
    round-5 route (13 zero convs on the ControlNet's stream + one gn_add_multi launch): same latents within two f16 roundings of the skip tensors,
    no add launch in the program, every zero conv carries a residual.  Reference: UNet2DConditionModel.forward's
    down_block_res_samples + down_block_additional_residuals / mid_block_additional_residual, called from `self.pipe(...)`
    (controller/agent/sd_controlnet_agent.py:67-76)."""
    from genima_amd.pipeline import StableDiffusionControlNetPipeline

    B, steps = 2, 2
    img_u8 = torch.from_numpy(weights.counter_bytes(3, "ctrl", B * 128 * 128 * 3).reshape(B, 128, 128, 3))
    lat = q16(torch.randn(B, 4, 16, 16, generator=torch.Generator().manual_seed(2)))
    lats, kinds = [], []
    for fused, split in ((False, True), (True, False), (True, True)):
        pipe = StableDiffusionControlNetPipeline.from_synthetic(FAM, seed=20)
        pipe.to("cuda")
        pipe.zero_convs_fused = fused
        ids = pipe.encode_ids(["open the box"] * B)
        prog = None
        import genima_amd.engine as eng
        old = eng.Engine.__init__

        def init(self, *a, _old=old, _split=split, **k):
            _old(self, *a, **k)
            self.zero_conv_split = _split
        eng.Engine.__init__ = init
        try:
            pipe(prompt_ids=ids, image=img_u8, num_inference_steps=steps, guidance_scale=0.0, latents=lat.half(), output_type="np")
            prog = pipe.program(B, 128, 128, steps)
        finally:
            eng.Engine.__init__ = old
        lats.append(prog.latents.float().cpu().clone())
        kinds.append([m["kind"] for m in prog.engine.meta])
    assert "add_multi" in kinds[0] and "add_multi" not in kinds[1] and "add_multi" not in kinds[2]
    assert kinds[2].count("stream") > kinds[1].count("stream"), "the split route forks the side stream once more per step"
    assert torch.equal(lats[1], lats[2]), "dealing the zero convs over two streams changes no value"
    assert rel_l2(lats[1], lats[0]) < 2e-3, rel_l2(lats[1], lats[0])
