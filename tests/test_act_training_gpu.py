"""-m gpu: the ACT controller update (genima_amd/act_training.py; controller/method/genima_act.py:27-139, :348-422) against torch autograd
over the CPU oracle (oracle/act_train_torch.py): the four loss terms, every trainable gradient, then the two-group AdamW step.
Dropout is off for the comparison (its masks are drawn on the device); a second run checks the dropout path statistically."""
import pytest
import torch

from genima_amd import configs, packing, schema, weights
from genima_amd.act_training import ACTTrainer, act_train_schema, trainable_names
from genima_amd.engine import Engine
from oracle import act_train_torch as OT
from oracle import sd_torch as O
from util import q16, rel_l2

pytestmark = pytest.mark.gpu


def _setup(seed=0):
    cfg, ccfg = dict(configs.TINY_ACT_POLICY, kl_weight=10.0), configs.TINY_ACT_CLIP_TEXT
    sd = weights.round_to(weights.synth_state_dict(act_train_schema(cfg), 61), torch.float16)
    for k in sd:  # BatchNorm statistics away from the identity so the affine path is exercised
        if k.endswith("running_var"):
            sd[k] = sd[k].abs() + 0.5
    g = torch.Generator().manual_seed(seed)
    B, V, S = 2, cfg["num_views"], cfg["image_size"]
    images = torch.randint(0, 256, (B, V, 3, S, S), generator=g, dtype=torch.uint8)
    qpos = q16(torch.randn(B, cfg["state_dim"], generator=g))
    task = q16(torch.randn(B, cfg["lang_dim"], generator=g) * 0.5)
    actions = q16(torch.randn(B, cfg["num_queries"], cfg["action_dim"], generator=g))
    actions[..., -1] = (actions[..., -1] > 0).float()  # gripper open / closed targets
    eps = torch.randn(B, cfg["latent_dim"], generator=g)
    return cfg, ccfg, sd, images, qpos, task, actions, eps


def test_act_update_matches_autograd_oracle():
    cfg, ccfg, sd, images, qpos, task, actions, eps = _setup()
    E = Engine("cuda:0")
    S = 256.0
    tr = ACTTrainer(E, cfg, sd, ccfg, None, loss_scale=S, dropout=0.0, state_dropout=0.0, lr=5e-5, lr_backbone=1e-5, weight_decay=1e-4)
    names = tr.names
    assert names[0].startswith("backbone.layer2") and not any(n.startswith("backbone.layer1") or ".bn" in n or "is_pad_head" in n for n in names)
    out4 = tr.forward_backward(images.permute(0, 1, 3, 4, 2).contiguous().cuda(), qpos, task, actions, eps).cpu()
    packed = {n: (tr.cn.G[n].float() / S).cpu() for n in tr.cn.layout}
    sch = act_train_schema(cfg)
    from collections import OrderedDict
    g_hip = packing.unpack_state_dict(packed, OrderedDict((k, sch[k]) for k in names))
    l32, g32, a32 = OT.update_forward_backward(sd, cfg, images, qpos, task, actions, eps, set(names))
    l16, g16, _ = OT.update_forward_backward(sd, cfg, images, qpos, task, actions, eps, set(names), q=q16)
    print("losses hip", [f"{v:.5f}" for v in out4.tolist()], "oracle", [f"{float(l32[k]):.5f}" for k in ("loss", "l1", "gripper_loss", "kl")])
    for got, k in zip(out4.tolist(), ("loss", "l1", "gripper_loss", "kl")):
        assert abs(got - float(l32[k])) <= 3e-3 * abs(float(l32[k])) + 1e-5, k
    e_a = rel_l2(tr.last["a_hat"].float().cpu(), a32)
    flat = lambda gd: torch.cat([gd[n].reshape(-1).float() for n in names])  # noqa: E731
    f_hip, f32_, f16_ = flat(g_hip), flat(g32), flat(g16)
    e_all, e_ref = rel_l2(f_hip, f32_), rel_l2(f16_, f32_)
    gn = float(f32_.norm())
    worst = sorted(((rel_l2(g_hip[n], g32[n]), n) for n in names if float(g32[n].norm()) > 1e-2 * gn), reverse=True)[:4]
    print(f"a_hat rel-L2 {e_a:.2e}; flat gradient ({f32_.numel()} elements, |g| {gn:.3e}) rel-L2 vs fp32 oracle {e_all:.2e} "
          f"(f16-storage oracle: {e_ref:.2e}); worst tensors {[(f'{e:.1e}', n) for e, n in worst]}")
    dead = [n for n in names if float(g_hip[n].abs().max()) == 0.0 and float(g32[n].abs().max()) > 0]
    assert not dead, dead[:5]
    assert e_a < 5e-3 and e_all <= max(1.5 * e_ref + 2e-3, 1e-2)
    # two-group AdamW (backbone lr 1e-5, rest 5e-5, weight decay 1e-4, no clipping)
    assert tr.optimizer_step()
    new = tr.state_dict()
    bb = {n for n in names if "backbone" in n}
    ref = OT.adamw_groups_step(sd, g32, bb, 5e-5, 1e-5, 1e-4)
    for n in ("backbone.layer4.1.conv2.weight", "transformer.decoder.layers.1.multihead_attn.in_proj_weight", "action_head.weight", "cls_embed.weight"):
        step_hip, step_ref = new[n].cpu() - sd[n], ref[n] - sd[n]
        lr = 1e-5 if n in bb else 5e-5
        assert float(step_ref.abs().max()) <= 1.01 * lr * (1 + 1e-4 * float(sd[n].abs().max())) + 1e-12
        agree = float(((step_hip.sign() == step_ref.sign()) | (step_ref.abs() < 0.5 * lr)).float().mean())
        assert agree > 0.97 and abs(float(step_hip.abs().mean()) / float(step_ref.abs().mean()) - 1) < 0.05, (n, agree)


def test_act_update_with_dropout_and_uint8_pipeline_trains():
    cfg, ccfg, sd, images, qpos, task, actions, _ = _setup(1)
    tr = ACTTrainer(Engine("cuda:0"), cfg, sd, ccfg, None, loss_scale=256.0, lr=2e-3, lr_backbone=2e-4, seed=3)  # large lr: a visible descent
    img = images.permute(0, 1, 3, 4, 2).contiguous().cuda()
    losses = [tr.update(img, qpos, task, actions)["actor_l1_loss"] for _ in range(12)]
    print("l1 over 12 updates with dropout:", [f"{v:.4f}" for v in losses])
    assert all(v == v for v in losses) and min(losses[-3:]) < 0.9 * losses[0] and tr.opt_step == 12


def test_agent_update_surface_and_augmentations():
    """``GenimaACT.update(replay_iter, step, replay_buffer)`` from a RoboBase-shaped replay batch, with the train-time augmentation
    pipeline (elastic warp / colour jitter / zero-padded crop / Gaussian noise); then ``act`` sees the updated weights."""
    import numpy as np

    from genima_amd.act import GenimaACT, act_schema
    from genima_amd.act_training import act_augment, elastic_displacement

    cfg, ccfg = dict(configs.TINY_ACT_POLICY, data_augmentation=True), configs.TINY_ACT_CLIP_TEXT
    agent = GenimaACT(cfg, None, ccfg, None, device="cuda", seed=4)
    B, S, Tq = 2, cfg["image_size"], cfg["num_queries"]
    g = torch.Generator().manual_seed(8)
    cams = ["left_shoulder", "right_shoulder", "front", "wrist"]
    Vc = ccfg["vocab_size"]
    toks = np.zeros((B, 1, 77), dtype=np.int32)
    toks[:, 0, :5] = [Vc - 2, 3, 4, 5, Vc - 1]
    batch = {f"{c}_rgb": torch.randint(0, 256, (B, 1, 3, S, S), generator=g, dtype=torch.uint8).numpy() for c in cams}
    batch.update({f"{c}_rgb_tp1": batch[f"{c}_rgb"] for c in cams})  # next-step observations must be ignored (rgb(?!.*?tp1))
    batch.update(low_dim_state=torch.randn(B, 1, cfg["state_dim"], generator=g).numpy(), lang_tokens=toks, reward=np.ones((B,), np.float32),
                 action=torch.rand(B, Tq, cfg["action_dim"], generator=g).numpy())
    obs = {k: torch.as_tensor(v) for k, v in batch.items() if "tp1" not in k and k not in ("action", "reward")}
    before = agent.act(obs).cpu()
    m = [agent.update(iter([batch]), i, lr=1e-3, lr_backbone=1e-4) for i in range(3)]
    assert set(m[0]) == {"actor_loss", "actor_l1_loss", "actor_gripper_loss", "actor_kl_loss", "batch_reward"} and m[0]["batch_reward"] == 1.0
    assert all(np.isfinite(list(x.values())).all() for x in m)
    after = agent.act(obs).cpu()
    assert not torch.equal(before, after), "act() must use the weights update() moved"
    # the augmentation pieces: identity displacement leaves the image alone; the field has the published scale (alpha / size * smoothing)
    E = agent._trainer.E
    img = torch.randint(0, 256, (1, 2, 64, 64, 3), generator=g, dtype=torch.uint8).cuda()
    disp = elastic_displacement(64, 64, generator=torch.Generator().manual_seed(1))
    assert tuple(disp.shape) == (64, 64, 2) and 0.01 < float(disp.abs().mean()) < 10.0
    out = act_augment(E, img, torch.Generator().manual_seed(5))
    assert out.shape == (1, 2, 64, 64, 8) and out.dtype == torch.float16 and float(out[..., 3:].abs().max()) == 0.0
    clean = act_augment(E, img, torch.Generator().manual_seed(5), p=0.0, noise_std=0.0)
    assert torch.equal(clean[..., :3].float().cpu(), (img.float() / 255.0).half().float().cpu())
    # the checkpoint seam after training (controller/train_act.py:262-279 -> controller/eval_genima.py:91-103): the snapshot carries the
    # trained weights under RoboBase's key paths and a fresh agent that passes the reference's gate acts identically
    import tempfile

    from genima_amd.harness import load_controller_ckpt, save_snapshot

    with tempfile.TemporaryDirectory() as td:
        payload = save_snapshot(agent, td + "/snapshots/exp/latest.pt", cfg={}, epoch=3, num_iters=3)
        assert not any("clip_model" in k for k in payload["agent"]) and "actor.actor_model.encoder.layers.0.linear1.weight" in payload["agent"]
        fresh = GenimaACT(cfg, None, ccfg, agent._clip_sd, device="cuda", seed=123)
        load_controller_ckpt(fresh, td + "/snapshots/exp/latest.pt")
    assert torch.equal(fresh.act(obs).cpu(), after)
