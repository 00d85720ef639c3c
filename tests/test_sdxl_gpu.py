"""-m gpu: the SDXL-Turbo family deltas (diffusion/train_controlnet_sdxl_genima.py) on the tiny-xl configs
(same topology: 3 levels, no attention at level 0, 1/2/3 transformer layers per block (1/2/10 at full size), text_time added conditions,
two CLIP towers with penultimate-hidden-state context and pooled projection) against the CPU oracle: forward of both towers / ControlNet /
UNet through the host classes, then the fine-tune step's loss and gradients.  Bars as in test_models_gpu.py / test_training_gpu.py."""
import pytest
import torch

from genima_amd import configs, schema, weights
from genima_amd.engine import Engine
from genima_amd.host import CLIPTextModel, CLIPTextModelWithProjection, ControlNetModel, UNet2DConditionModel, nchw_to_nhwc
from genima_amd.packing import pack_state_dict
from genima_amd.scheduler import DDPMScheduler
from genima_amd.training import ControlNetTrainer
from oracle import sd_torch as O
from oracle import train_torch as OT
from util import q16, rel_l2

pytestmark = pytest.mark.gpu

FAM = configs.family("tiny-xl")


def _r16(sd):
    return weights.round_to(sd, torch.float16)


def _ids(V, B=2):
    ids = torch.zeros(B, 77, dtype=torch.int64)
    ids[0, :14] = torch.tensor([V - 2] + [320 + i for i in range(12)] + [V - 1])
    ids[1, :5] = torch.tensor([V - 2, 7, 8, 9, V - 1])
    return ids[:B]


def _close(name, y, ref16, ref32, tol16=3e-3, tol32=1e-2):
    e16, e32, eref = rel_l2(y, ref16), rel_l2(y, ref32), rel_l2(ref16, ref32)
    print(f"{name}: rel-L2 vs f16-storage oracle {e16:.2e}, vs fp32 oracle {e32:.2e} (f16-storage oracle vs fp32: {eref:.2e})")
    assert torch.isfinite(y.float()).all()
    assert e16 <= tol16 and e32 <= min(tol32, 1.5 * eref + 5e-4), name


def test_sdxl_text_towers():
    for key, cls, seed in (("text", CLIPTextModel, 5), ("text_2", CLIPTextModelWithProjection, 6)):
        cfg = FAM[key]
        sd = _r16(weights.synth_state_dict(schema.clip_text_schema(cfg), seed))
        ids = _ids(cfg["vocab_size"])
        out = cls(cfg, sd).to("cuda")(ids, output_hidden_states=True)
        with torch.no_grad():
            p16, e16 = O.clip_text_penultimate_and_pooled(sd, cfg, ids, q16)
            p32, e32 = O.clip_text_penultimate_and_pooled(sd, cfg, ids)
        _close(f"{key} hidden_states[-2]", out[-1][-2].float().cpu(), p16, p32, 1e-3, 3e-3)
        if e32 is not None:
            _close(f"{key} text_embeds", out[0].float().cpu(), e16, e32, 2e-3, 5e-3)


def _inputs(B=2, seed=0):
    ucfg = FAM["unet"]
    g = torch.Generator().manual_seed(seed)
    lat = q16(torch.randn(B, 4, 32, 32, generator=g))
    noise = q16(torch.randn(B, 4, 32, 32, generator=g))
    ctx = q16(torch.randn(B, 77, ucfg["cross_attention_dim"], generator=g))
    cond = q16(torch.rand(B, 3, 256, 256, generator=g))
    text_embeds = q16(torch.randn(B, ucfg["projection_class_embeddings_input_dim"] - 6 * ucfg["addition_time_embed_dim"], generator=g))
    time_ids = torch.tensor([[256.0, 256.0, 0.0, 0.0, 256.0, 256.0]] * B)  # compute_embeddings: original size, crop, target size
    t = torch.tensor([801, 399][:B])
    return lat, noise, ctx, cond, (text_embeds, time_ids), t


def test_sdxl_unet_and_controlnet_forward():
    ucfg, ccfg = FAM["unet"], FAM["controlnet"]
    usd = _r16(weights.synth_state_dict(schema.unet_schema(ucfg), 1))
    csd = _r16(weights.synth_state_dict(schema.controlnet_schema(ccfg), 2))
    unet, cn = UNet2DConditionModel(ucfg, usd).to("cuda"), ControlNetModel(ccfg, csd).to("cuda")
    x, _, ctx, cond, added, t = _inputs()
    tf = t.float()
    kw = dict(added_cond_kwargs={"text_embeds": added[0], "time_ids": added[1]})
    down, mid = cn(x.half(), tf, ctx.half(), cond.half(), return_dict=False, **kw)
    assert len(down) == 9  # SDXL: 3 levels -> 9 skips
    with torch.no_grad():
        d16, m16 = O.controlnet_forward(csd, ccfg, x, tf, ctx, cond, q=q16, added=added)
        d32, m32 = O.controlnet_forward(csd, ccfg, x, tf, ctx, cond, added=added)
    for i, (a, b, c) in enumerate(zip(down, d16, d32)):
        _close(f"xl controlnet down[{i}]", a.float().cpu(), b, c)
    _close("xl controlnet mid", mid.float().cpu(), m16, m32)
    eps = unet(x.half(), tf, ctx.half(), [d.half() for d in d16], m16.half(), **kw).sample
    with torch.no_grad():
        e16 = O.unet_forward(usd, ucfg, x, tf, ctx, [q16(d) for d in d16], q16(m16), q=q16, added=added)
        e32 = O.unet_forward(usd, ucfg, x, tf, ctx, [q16(d) for d in d16], q16(m16), added=added)
    _close("xl unet eps", eps.float().cpu(), e16, e32)


def test_sdxl_pipeline_end_to_end():
    """StableDiffusionXLControlNetPipeline (two towers -> context + added conditions, EulerAncestral with per-step noise) against the
    oracle restating the same chain; the per-step noise comes from the same CPU generator on both sides."""
    import numpy as np

    from genima_amd.pipeline import StableDiffusionXLControlNetPipeline

    pipe = StableDiffusionXLControlNetPipeline.from_synthetic(FAM, seed=30)
    for m in (pipe.vae, pipe.text_encoder, pipe.text_encoder_2, pipe.unet, pipe.controlnet):
        m.load_state_dict(_r16(m.state_dict()))
    pipe.to("cuda")
    B, steps, R = 2, 3, 256
    img_u8 = torch.from_numpy(weights.counter_bytes(5, "xl_ctrl", B * R * R * 3).reshape(B, R, R, 3))
    ids = _ids(FAM["text"]["vocab_size"])
    lat = q16(torch.randn(B, 4, R // 8, R // 8, generator=torch.Generator().manual_seed(2)))
    out = pipe(prompt_ids=ids, prompt_ids_2=ids, image=img_u8, num_inference_steps=steps, guidance_scale=0.0, latents=lat.half(),
               generator=torch.Generator().manual_seed(11), output_type="np")
    assert out.images.shape == (B, R, R, 3) and out.images.dtype == np.uint8
    lat_hip = pipe.program(B, R, R, steps).latents.permute(0, 3, 1, 2).float().cpu()

    def oracle(q):
        g = torch.Generator().manual_seed(11)
        sch = pipe.scheduler
        sch.set_timesteps(steps)
        tl, tg = pipe.text_encoder.state_dict(), pipe.text_encoder_2.state_dict()
        pl, _ = O.clip_text_penultimate_and_pooled(tl, FAM["text"], ids, q)
        pg, pooled = O.clip_text_penultimate_and_pooled(tg, FAM["text_2"], ids, q)
        ctx = torch.cat([pl, pg], -1)
        added = (pooled, torch.tensor([[float(R), float(R), 0.0, 0.0, float(R), float(R)]] * B))
        cond = img_u8.permute(0, 3, 1, 2).float() / 255.0
        x = q(lat * sch.init_noise_sigma)
        usd, csd = pipe.unet.state_dict(), pipe.controlnet.state_dict()
        for i in range(steps):
            t = torch.full((B,), float(sch.timesteps[i]))
            xin = q(x * sch.input_scale(i))
            down, mid = O.controlnet_forward(csd, FAM["controlnet"], xin, t, ctx, q(cond), q=q, added=added)
            eps = O.unet_forward(usd, FAM["unet"], xin, t, ctx, down, mid, q=q, added=added)
            s_down, s_up = sch.ancestral_sigmas(i)
            noise = torch.randn(B, 4, R // 8, R // 8, generator=g, dtype=torch.float16).float()
            x = q(x + eps * (s_down - float(sch.sigmas[i])))
            if s_up > 0:
                x = q(x + noise * s_up)
        return x

    with torch.no_grad():
        x16, x32 = oracle(q16), oracle(lambda t: t)
    e16, e32, eref = rel_l2(lat_hip, x16), rel_l2(lat_hip, x32), rel_l2(x16, x32)
    print(f"xl pipeline latents: rel-L2 vs f16-storage oracle {e16:.2e}, vs fp32 oracle {e32:.2e} (f16-storage oracle vs fp32: {eref:.2e})")
    assert e16 <= 5e-3 and e32 <= min(3e-2, 1.5 * eref + 1e-3)


def test_sdxl_controlnet_train_step():
    ucfg, ccfg = FAM["unet"], FAM["controlnet"]
    usd = _r16(weights.synth_state_dict(schema.unet_schema(ucfg), 1))
    csd = _r16(weights.synth_state_dict(schema.controlnet_schema(ccfg), 2))
    lat, noise, ctx, cond, added, t = _inputs()
    sa, s1 = DDPMScheduler().add_noise_coeffs(t)
    S = 4096.0
    tr = ControlNetTrainer(Engine("cuda:0"), ucfg, ccfg, pack_state_dict(usd, "cuda"), csd, lr=1e-4, loss_scale=S)
    dev = lambda x: x.cuda()  # noqa: E731
    loss = float(tr.forward_backward(dev(nchw_to_nhwc(lat, 8).half()), dev(nchw_to_nhwc(noise, 8).half()), dev(t.float()), dev(sa), dev(s1),
                                     dev(ctx.half()), dev(nchw_to_nhwc(cond, 8).half()), added=(dev(added[0].half()), dev(added[1]))).cpu())
    layout = list(tr.cn.layout)
    assert any(n.startswith("add_embedding.") for n in layout)
    g_hip = torch.cat([(tr.cn.G[n].float() / S).reshape(-1).cpu() for n in layout])
    l32, g32, _ = OT.train_forward_backward(usd, csd, ucfg, ccfg, lat, noise, t.float(), sa, s1, ctx, cond, added=added)
    l16, g16, _ = OT.train_forward_backward(usd, csd, ucfg, ccfg, lat, noise, t.float(), sa, s1, ctx, cond, q=q16, added=added)
    P32, P16 = pack_state_dict(g32, "cpu", dtype=torch.float32), pack_state_dict(g16, "cpu", dtype=torch.float32)
    f32_, f16_ = (torch.cat([P[n].reshape(-1).float() for n in layout]) for P in (P32, P16))
    e, eref = rel_l2(g_hip, f32_), rel_l2(f16_, f32_)
    print(f"xl loss hip {loss:.6f} oracle {float(l32):.6f}; flat gradient rel-L2 vs fp32 oracle {e:.2e} (f16-storage oracle: {eref:.2e})")
    assert abs(loss - float(l32)) <= 2e-3 * float(l32)
    assert e <= min(1e-2, 1.5 * eref + 2e-3)
    gn = float(f32_.double().norm())
    for n in layout:
        if n.startswith("add_embedding."):
            assert float((tr.cn.G[n].float().cpu() / S - P32[n]).double().norm()) <= 2e-2 * float(P32[n].double().norm()) + 1e-4 * gn, n


def test_sdxl_fp8_frozen_step_every_routed_linear_matches_the_fp8_oracle():
    """BASELINE configs[4] ("fp8 MFMA"): with ``enable_fp8_frozen()`` the frozen UNet's transformer Linears run on the fp8 MFMA inside the
    real train step.  Every Linear the step routes there is captured (inputs and output) and checked against the CPU restatement of the
    scheme (oracle/fp8_torch.py: row-wise e4m3 quantisation of activations and weights, exact products, f32 accumulate, dequantise,
    epilogue); then the whole step is compared with the f16 step: the loss and the ControlNet gradient move by e4m3-sized amounts only."""
    from genima_amd._lib import ACT_GEGLU, ACT_NONE
    from oracle import fp8_torch as F8

    ucfg, ccfg = FAM["unet"], FAM["controlnet"]
    usd = _r16(weights.synth_state_dict(schema.unet_schema(ucfg), 1))
    csd = _r16(weights.synth_state_dict(schema.controlnet_schema(ccfg), 2))
    lat, noise, ctx, cond, added, t = _inputs()
    sa, s1 = DDPMScheduler().add_noise_coeffs(t)
    S = 4096.0
    dev = lambda x: x.cuda()  # noqa: E731
    args = (dev(nchw_to_nhwc(lat, 8).half()), dev(nchw_to_nhwc(noise, 8).half()), dev(t.float()), dev(sa), dev(s1), dev(ctx.half()),
            dev(nchw_to_nhwc(cond, 8).half()))
    kw = dict(added=(dev(added[0].half()), dev(added[1])))
    unet_W = pack_state_dict(usd, "cuda")

    def run(fp8):
        E = Engine("cuda:0")
        tr = ControlNetTrainer(E, ucfg, ccfg, unet_W, csd, lr=1e-4, loss_scale=S)
        calls = []
        if fp8:
            E.fp8_min_rows = 1  # the tiny family's GEMMs have 32 .. 2048 rows: route them all (the product default skips < 1024)
            assert tr.enable_fp8_frozen() > 20
            inner = E.linear_fp8

            def spy(xq, xs, wq, ws, bias=None, *, act=ACT_NONE, residual=None, out=None, name=None):
                y = inner(xq, xs, wq, ws, bias, act=act, residual=residual, out=out, name=name)
                calls.append((xq, xs, wq, ws, bias, act, residual, y))
                return y
            E.linear_fp8 = spy
        loss = float(tr.forward_backward(*args, **kw).cpu())
        g = torch.cat([(tr.cn.G[n].float() / S).reshape(-1) for n in tr.cn.layout]).cpu()
        return loss, g, calls, E

    l16, g16, _, _ = run(False)
    l8, g8, calls, E = run(True)
    assert len(calls) >= 40, len(calls)
    # the quantised operands the GEMM consumed ARE the oracle's quantisation of some f16 tensor; check the GEMM + epilogue on them
    import torch.nn.functional as Fn
    worst, checked = 0.0, 0
    for xq, xs, wq, ws, bias, act, residual, y in calls[:: max(1, len(calls) // 24)]:
        K = wq.shape[1]
        xd = xq.view(-1, K).cpu().view(torch.float8_e4m3fn).float() * xs[: xq.numel() // K].cpu()[:, None]
        wd = wq.cpu().view(torch.float8_e4m3fn).float() * ws[: wq.shape[0]].cpu()[:, None]
        ref = (xd.double() @ wd.double().t()).float()
        if bias is not None:
            ref = ref + bias.float().cpu()
        if act == ACT_GEGLU:  # packed rows alternate 32-row [hidden | gate] blocks
            r = ref.view(ref.shape[0], -1, 2, 32)
            ref = (r[:, :, 0] * Fn.gelu(r[:, :, 1])).reshape(ref.shape[0], -1)
        else:
            assert act == ACT_NONE
        if residual is not None:
            ref = ref + residual.float().cpu().view(ref.shape)
        worst = max(worst, rel_l2(y.float().cpu().view(ref.shape), ref))
        checked += 1
    # and the quantiser itself, on one activation: bytes and scales bit-exact against the oracle
    xprobe = (torch.randn(300, 128, generator=torch.Generator().manual_seed(3)) * 2).half()
    q_dev, s_dev = E.quantize_fp8(xprobe.cuda())
    q_ref, s_ref = F8.quantize_rows(xprobe)
    assert torch.equal(q_dev.cpu()[:, :128].view(torch.float8_e4m3fn).float(), q_ref.float()) and torch.equal(s_dev.cpu()[:300], s_ref)
    e_g = rel_l2(g8, g16)
    print(f"tiny-xl fp8-frozen step: {len(calls)} Linears on the fp8 MFMA ({checked} checked vs the oracle, worst rel-L2 {worst:.2e}); "
          f"loss {l8:.6f} vs f16 {l16:.6f}; ControlNet gradient rel-L2 fp8 vs f16 {e_g:.2e}")
    assert worst < 2e-3
    assert abs(l8 - l16) <= 2e-2 * l16 and 1e-4 < e_g < 0.2  # e4m3 carries 3 mantissa bits: percent-level, and not identical
