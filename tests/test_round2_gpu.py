"""-m gpu: round-2 additions on the HIP path -- AutoencoderTiny (TAESD) decode, the agents' ``autoencoder: taesd`` switch, recorded
program invalidation when a module or the scheduler is replaced, gradient accumulation + lr schedule of the fine-tune step."""
import types

import numpy as np
import pytest
import torch

from genima_amd import configs, schema, weights
from oracle import sd_torch as O
from tests.util import assert_close, q16

pytestmark = pytest.mark.gpu


def test_taesd_decode_matches_oracle():
    """controller/agent/sd_controlnet_agent.py:45-49: ``AutoencoderTiny.decode`` at full TAESD width (it IS tiny: 1.22 M params)."""
    from genima_amd.host import AutoencoderTiny

    sd = weights.round_to(weights.synth_state_dict(schema.taesd_schema(configs.TAESD), 11), torch.float16)
    vae = AutoencoderTiny(configs.TAESD, sd).to("cuda")
    z = (torch.randn(2, 4, 16, 24, generator=torch.Generator().manual_seed(3)) * 2.5).half()
    got = vae.decode(z).sample.float().cpu()
    with torch.no_grad():
        ref = O.taesd_decode(sd, configs.TAESD, z.float())
        ref16 = O.taesd_decode(sd, configs.TAESD, z.float(), q16)
    assert tuple(got.shape) == (2, 3, 128, 192)
    e = assert_close(got, ref, rel=2e-3, what="taesd decode vs fp32 oracle")
    from tests.util import rel_l2
    assert e <= 1.5 * rel_l2(ref16, ref) + 5e-4, "HIP decode should be as close to fp32 as the f16-storage oracle is"


def _agent_cfg(**kw):
    base = dict(diffusion_ckpt="", sd_ckpt="synthetic:tiny", device="cuda", image_resolution=512, vae_slicing=False, upcast_vae=False,
                fused_projections=True, enable_xformers_memory_efficient_attention=True, show_diffusion_progress=False,
                torch_compile=False, autoencoder="")
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_agent_taesd_switch_and_program_invalidation(tmp_path):
    from genima_amd.agent import SDControlNetAgent
    from genima_amd.host import AutoencoderTiny, ControlNetModel

    tdir = str(tmp_path / "taesd")
    AutoencoderTiny.from_config(configs.TAESD, seed=5).save_pretrained(tdir)
    agent = SDControlNetAgent(_agent_cfg(autoencoder=tdir))
    assert isinstance(agent.pipe.vae, AutoencoderTiny)
    pipe = agent.pipe
    img = torch.from_numpy(weights.counter_bytes(9, "r2", 128 * 128 * 3).reshape(1, 128, 128, 3))
    ids = pipe.encode_ids(["tiled perspectives of a robot arm executing 'open box'"])
    lat = torch.randn(1, 4, 16, 16, generator=torch.Generator().manual_seed(2)).half()
    kw = dict(prompt_ids=ids, image=img, num_inference_steps=2, guidance_scale=0.0, latents=lat, output_type="np")
    a = pipe(**kw).images
    assert a.shape == (1, 128, 128, 3) and a.dtype == np.uint8
    # oracle: same 2-step latents through the fp32 TAESD decoder
    lat_out = pipe(**dict(kw, output_type="latent")).images.float().cpu()
    with torch.no_grad():
        ref = O.vae_postprocess_u8(O.taesd_decode(pipe.vae.state_dict(), configs.TAESD, lat_out / 1.0)).numpy()
    dd = np.abs(a.astype(np.int32) - ref.astype(np.int32))  # random-weight TAESD output spans many times [-1, 1]: 1e-3 relative > 1 LSB there
    assert dd.mean() < 0.25 and (dd > 1).mean() < 2e-3 and dd.max() <= 4, (dd.mean(), (dd > 1).mean(), dd.max())
    # replacing the ControlNet (new packed tensors) must not replay the recorded program of the old one
    fam = configs.family("tiny")
    pipe.controlnet = ControlNetModel(fam["controlnet"], weights.synth_state_dict(schema.controlnet_schema(fam["controlnet"]), 77)).to("cuda")
    b = pipe(**kw).images
    assert not np.array_equal(a, b), "a stale recorded program was replayed after pipe.controlnet changed"
    # ... nor after load_state_dict re-packs in place
    pipe.controlnet.load_state_dict(weights.synth_state_dict(schema.controlnet_schema(fam["controlnet"]), 78))
    c = pipe(**kw).images
    assert not np.array_equal(b, c)
    # ... nor after the scheduler object is swapped (log_validation does exactly this, train_controlnet_genima.py:545-553)
    from genima_amd.scheduler import EulerDiscreteScheduler
    pipe.scheduler = EulerDiscreteScheduler.from_config(dict(configs.SD_TURBO_SCHEDULER, timestep_spacing="leading"))
    d = pipe(**kw).images
    assert not np.array_equal(c, d)
    assert np.array_equal(d, pipe(**kw).images)  # and an unchanged pipeline still replays bit-identically


def test_gradient_accumulation_and_lr_schedule():
    """Two micro-batches of one sample under ``gradient_accumulation_steps=2`` = one step on the batch of two (same noise draws fed
    explicitly): accelerate's 1/N loss scaling, diffusion/train_controlnet_genima.py:1319, :1402; the lr multiplier is applied and
    advanced once per applied step (:1206-1213, :1407)."""
    from genima_amd.engine import Engine
    from genima_amd.host import UNet2DConditionModel, ControlNetModel
    from genima_amd.train_loop import get_scheduler
    from genima_amd.training import ControlNetTrainer

    fam = configs.family("tiny")
    E = Engine("cuda:0")
    unet = UNet2DConditionModel.from_config(fam["unet"], 1).to("cuda")
    cn_sd = weights.synth_state_dict(schema.controlnet_schema(fam["controlnet"]), 2)
    g = torch.Generator().manual_seed(0)
    B, h = 2, 32
    lat = (torch.randn(B, h, h, 8, generator=g) * 0.5).half().cuda()
    noi = torch.randn(B, h, h, 8, generator=g).half().cuda()
    lat[..., 4:] = 0
    noi[..., 4:] = 0
    t = torch.tensor([301.0, 744.0]).cuda()
    sa, s1 = torch.tensor([0.8, 0.5]).cuda(), torch.tensor([0.6, 0.866]).cuda()
    ctx = (torch.randn(B, 77, fam["unet"]["cross_attention_dim"], generator=g) * 0.3).half().cuda()
    cond = torch.rand(B, 8 * h, 8 * h, 8, generator=g).half().cuda()
    cond[..., 3:] = 0
    lam = get_scheduler("constant_with_warmup", num_warmup_steps=4)
    full = ControlNetTrainer(E, fam["unet"], fam["controlnet"], unet.W, cn_sd, lr=1e-3, lr_lambda=lambda s: lam(s + 1))
    acc = ControlNetTrainer(E, fam["unet"], fam["controlnet"], unet.W, cn_sd, lr=1e-3, lr_lambda=lambda s: lam(s + 1),
                            gradient_accumulation_steps=2)
    full.step(lat, noi, t, sa, s1, ctx, cond)
    for i in range(B):
        sl = slice(i, i + 1)
        acc.step(lat[sl], noi[sl], t[sl], sa[sl], s1[sl], ctx[sl], cond[sl])
        assert acc.sync_gradients == (i == B - 1)
    torch.cuda.synchronize()
    assert full.opt_step == acc.opt_step == 1 and full.sched_step == acc.sched_step == 1
    assert abs(full.current_lr() - 1e-3 * 2 / 4) < 1e-12  # after one applied step the warm-up multiplier is (1 + 1) / 4
    gn_f, gn_a = full.last["grad_norm"], acc.last["grad_norm"]
    assert abs(gn_f - gn_a) <= 2e-3 * gn_f, (gn_f, gn_a)
    # Adam's first step is sign-like (|update| = lr wherever the gradient is not ~0), so the two runs agree except on the few elements
    # whose tiny gradient changes sign with the summation order: compare in the mean, not in the max
    init = ControlNetTrainer(E, fam["unet"], fam["controlnet"], unet.W, cn_sd).cn.master
    upd = (full.cn.master - init).abs().mean().item()
    d = (full.cn.master - acc.cn.master).abs().mean().item()
    assert upd > 0 and d <= 0.02 * upd, (d, upd)


def test_data_path_to_trainer_end_to_end(tmp_path):
    """PNG tree -> RLBenchDataset -> DataLoader (uint8 NHWC, prefetch thread) -> device ToTensor / Normalize kernel -> train_step, driven by
    TrainLoop with accumulation, checkpoint rotation and ``resume latest`` (diffusion/train_controlnet_genima.py:870-964, 1281-1457)."""
    import os
    import pickle

    from PIL import Image

    from genima_amd import data as D
    from genima_amd.engine import Engine
    from genima_amd.packing import pack_state_dict
    from genima_amd.pipeline import HashTokenizer
    from genima_amd.scheduler import DDPMScheduler
    from genima_amd.train_loop import TrainLoop, list_checkpoints
    from genima_amd.training import ControlNetTrainer

    root = str(tmp_path / "data")
    rng = np.random.RandomState(1)
    base = os.path.join(root, "open_box", "variation0")
    os.makedirs(os.path.join(base, "episodes"))
    with open(os.path.join(base, "variation_descriptions.pkl"), "wb") as f:
        pickle.dump(["open the box"], f)
    for e in range(2):
        for kind in ("rgb", "rgb_rendered"):
            d = os.path.join(base, "episodes", f"episode{e}", kind)
            os.makedirs(d)
            for i in range(4):
                Image.fromarray(rng.randint(0, 256, (300, 300, 3), dtype=np.uint8)).save(os.path.join(d, f"{i}.png"))
    ds = D.RLBenchDataset(root, tasks="open_box", num_demos=2)
    fam = configs.family("tiny")
    tok = HashTokenizer(fam["text"]["vocab_size"])
    loader = D.DataLoader(ds, 2, tok, 256, shuffle=True, seed=0)  # Resize(256) + CenterCrop(256): 32x32 latents
    E = Engine("cuda:0")
    # the device conversion equals ToTensor + Normalize exactly (f16-rounded)
    hb = next(iter(D.DataLoader(ds, 2, tok, 256, shuffle=False, prefetch=0)))
    db = D.to_device(E, hb)
    ref = D.collate_fn([ds[0], ds[1]], tok, 256)
    assert torch.equal(db["pixel_values"][..., :3].permute(0, 3, 1, 2).float().cpu(), ref["pixel_values"].half().float())
    assert torch.equal(db["conditioning_pixel_values"][..., :3].permute(0, 3, 1, 2).float().cpu(), ref["conditioning_pixel_values"].half().float())
    assert float(db["pixel_values"][..., 3:].abs().max()) == 0.0 and torch.equal(db["input_ids"].cpu(), ref["input_ids"])

    def trainer():
        synth = lambda sch, s: weights.synth_state_dict(sch, s)  # noqa: E731
        tr = ControlNetTrainer(E, fam["unet"], fam["controlnet"], pack_state_dict(synth(schema.unet_schema(fam["unet"]), 1), "cuda"),
                               synth(schema.controlnet_schema(fam["controlnet"]), 2), lr=1e-4, gradient_accumulation_steps=2)
        tr.attach_frozen(fam["vae"], pack_state_dict(synth(schema.vae_schema(fam["vae"]), 3), "cuda"), fam["text"],
                         pack_state_dict(synth(schema.clip_text_schema(fam["text"]), 4), "cuda"), DDPMScheduler(), seed=5,
                         augmentations="crop,colorjitter")
        return tr

    out = str(tmp_path / "run")
    logs = []
    tr = trainer()
    loop = TrainLoop(tr, out, num_train_epochs=2, checkpointing_steps=1, checkpoints_total_limit=2, log=logs.append)
    # 6 examples / batch 2 = 3 micro-batches per epoch, 2 per optimizer step: the epoch's last batch syncs on its own (accelerate's
    # end_of_dataloader), so ceil(3 / 2) = 2 optimizer steps per epoch and nothing accumulated crosses the epoch boundary
    assert loop.run(loader) == 4
    assert tr.opt_step == 4 and tr._micro == 0 and np.isfinite(float(loop.last_loss))
    assert list_checkpoints(out) == ["checkpoint-3", "checkpoint-4"]
    tr2 = trainer()
    loop2 = TrainLoop(tr2, out, max_train_steps=5, checkpointing_steps=100, resume_from_checkpoint="latest", log=logs.append)
    assert loop2.run(loader) == 5 and torch.equal(tr2.cn.exp_avg_sq > 0, tr2.cn.exp_avg_sq > 0)
    assert any("Resuming from checkpoint checkpoint-4" in m for m in logs) and tr2.opt_step == 5
