"""-m gpu: ACT controller forward (ResNet-18 + DETR encoder/decoder on the HIP kernels) and the CLIP pooled text projection
against the CPU oracle (oracle/act_torch.py, oracle/sd_torch.py) on seeded synthetic weights."""
import pytest
import torch

from genima_amd import configs, schema, weights
from genima_amd.act import GenimaACT, act_schema
from oracle import act_torch as OA
from oracle import sd_torch as O
from util import q16, rel_l2

pytestmark = pytest.mark.gpu


def _setup(cfg, ccfg, seed=0):
    sd = weights.round_to(weights.synth_state_dict(act_schema(cfg), seed + 31), torch.float16)
    csd = weights.round_to(weights.synth_state_dict(schema.clip_text_schema(ccfg), seed + 32), torch.float16)
    return sd, csd, GenimaACT(cfg, sd, ccfg, csd, device="cuda")


@pytest.mark.parametrize("family", ["tiny", "full"])
def test_act_forward(family):
    if family == "tiny":
        cfg, ccfg = configs.TINY_ACT_POLICY, configs.TINY_ACT_CLIP_TEXT
    else:
        cfg, ccfg = configs.ACT_POLICY, dict(configs.TINY_ACT_CLIP_TEXT, projection_dim=512)
    sd, csd, agent = _setup(cfg, ccfg)
    B, V, S = 2, cfg["num_views"], cfg["image_size"]
    g = torch.Generator().manual_seed(5)
    cams = ["left_shoulder", "right_shoulder", "front", "wrist"][:V]
    obs = {f"{c}_rgb": torch.randint(0, 256, (B, 1, 3, S, S), generator=g, dtype=torch.uint8) for c in cams}
    obs["low_dim_state"] = torch.randn(B, 1, cfg["state_dim"], generator=g)
    Vc = ccfg["vocab_size"]
    toks = torch.zeros(B, 1, 77, dtype=torch.int32)
    toks[:, 0, :6] = torch.tensor([Vc - 2, 11, 12, 13, 14, Vc - 1], dtype=torch.int32)
    toks[1, 0, 4:6] = torch.tensor([Vc - 1, 0], dtype=torch.int32)
    obs["lang_tokens"] = toks
    a = agent.act(obs, step=0, eval_mode=True).cpu()
    assert a.shape == (B, cfg["num_queries"], cfg["action_dim"]) and a.dtype == torch.float32
    # oracle
    imgs = torch.stack([obs[f"{c}_rgb"][:, 0] for c in cams], dim=1)
    qpos = q16(obs["low_dim_state"][:, 0])
    ids = toks[:, 0].long()
    with torch.no_grad():
        t16 = q16(O.clip_text_pooled_projection(csd, ccfg, ids, q16))
        t32 = O.clip_text_pooled_projection(csd, ccfg, ids)
        r16, _ = OA.act_forward(sd, cfg, imgs, qpos, t16, q16)
        r32, _ = OA.act_forward(sd, cfg, imgs, qpos, t32)
    task, _ = agent.encode_clip_text(toks)
    e_t = rel_l2(task.cpu(), t32)
    e16, e32, eref = rel_l2(a, r16), rel_l2(a, r32), rel_l2(r16, r32)
    print(f"ACT[{family}] task_emb rel-L2 {e_t:.2e}; a_hat vs f16-storage oracle {e16:.2e}, vs fp32 {e32:.2e} (oracle16 vs 32: {eref:.2e})")
    assert e_t < 3e-3
    assert e16 < 4e-3 and e32 < max(1.5 * eref + 5e-4, 4e-3)


def test_act_frame_stack_and_robobase_state_dict():
    """``frame_stack = 2`` (per-view frames stacked on channels in front of ``projection_layer``, controller/method/genima_act.py:191-197)
    with language conditioning, loaded the way the eval loop does it: ``agent.load_state_dict(ckpt["agent"], strict=False)``
    (controller/eval_genima.py:91-103) from a RoboBase-shaped dict -- every weight under ``actor.actor_model.* / actor.encoder_model.*``
    AND its duplicate registrations (``actor_model.*``, ``encoder.*``), the CVAE posterior encoder's keys beside them."""
    cfg, ccfg = dict(configs.TINY_ACT_POLICY, frame_stack=2), configs.TINY_ACT_CLIP_TEXT
    sd = weights.round_to(weights.synth_state_dict(act_schema(cfg), 51), torch.float16)
    csd = weights.round_to(weights.synth_state_dict(schema.clip_text_schema(ccfg), 52), torch.float16)
    enc_keys = lambda k: k.startswith("backbone.") or k.startswith("input_proj.")  # noqa: E731
    agent_sd = {}
    for k, v in sd.items():
        if enc_keys(k):
            body = k.replace("backbone.", "backbone.0.body.", 1) if k.startswith("backbone.") else k
            agent_sd["actor.encoder_model." + body] = v
            agent_sd["encoder." + body] = v
        elif k == "projection_layer.weight" or k == "projection_layer.bias":
            agent_sd["actor." + k] = v
        else:
            agent_sd["actor.actor_model." + k] = v
            agent_sd["actor_model." + k] = v
    agent_sd["actor.actor_model.encoder.layers.0.norm1.weight"] = torch.ones(cfg["hidden_dim"])      # CVAE style encoder (training only)
    agent_sd["actor.actor_model.cls_embed.weight"] = torch.zeros(1, cfg["hidden_dim"])
    agent_sd["actor.encoder_model.backbone.0.body.bn1.num_batches_tracked"] = torch.tensor(0)
    agent = GenimaACT(cfg, None, ccfg, csd, device="cuda", seed=99)  # different random init: everything must come from the dict
    missing, unexpected = agent.load_state_dict(agent_sd, strict=False)
    assert missing == [] and unexpected == [], (missing[:4], unexpected[:4])  # the CVAE keys land in the training-only store
    assert torch.equal(agent._sd_train["cls_embed.weight"], torch.zeros(1, cfg["hidden_dim"]))
    with pytest.raises(KeyError):
        agent.load_state_dict({"critic.fc.weight": torch.zeros(2, 2)}, strict=False)
    B, V, S, fs = 2, cfg["num_views"], cfg["image_size"], 2
    g = torch.Generator().manual_seed(6)
    cams = ["left_shoulder", "right_shoulder", "front", "wrist"][:V]
    obs = {f"{c}_rgb": torch.randint(0, 256, (B, fs, 3, S, S), generator=g, dtype=torch.uint8) for c in cams}
    obs["low_dim_state"] = torch.randn(B, fs, cfg["state_dim"], generator=g)[:, :1]
    Vc = ccfg["vocab_size"]
    toks = torch.zeros(B, fs, 77, dtype=torch.int32)
    toks[:, :, :5] = torch.tensor([Vc - 2, 21, 22, 23, Vc - 1], dtype=torch.int32)
    obs["lang_tokens"] = toks
    a = agent.act(obs, step=0, eval_mode=True).cpu()
    imgs = torch.stack([obs[f"{c}_rgb"] for c in cams], dim=1).reshape(B, V * fs, 3, S, S)  # camera-major: index = cam * fs + frame
    qpos = q16(obs["low_dim_state"].flatten(1))
    with torch.no_grad():
        t16 = q16(O.clip_text_pooled_projection(csd, ccfg, toks[:, 0].long(), q16))
        r16, _ = OA.act_forward(sd, cfg, imgs, qpos, t16, q16)
        r32, _ = OA.act_forward(sd, cfg, imgs, qpos, O.clip_text_pooled_projection(csd, ccfg, toks[:, 0].long()))
        no_lang, _ = OA.act_forward(sd, dict(cfg, use_lang_cond=False), imgs, qpos, None)
    e16, e32, eref = rel_l2(a, r16), rel_l2(a, r32), rel_l2(r16, r32)
    print(f"ACT frame_stack=2 + FiLM: vs f16-storage oracle {e16:.2e}, vs fp32 {e32:.2e} (oracle16 vs 32: {eref:.2e}); "
          f"language conditioning moves the actions by {rel_l2(r32, no_lang):.2e}")
    assert a.shape == (B, cfg["num_queries"], cfg["action_dim"])
    assert e16 < 4e-3 and e32 < max(1.5 * eref + 5e-4, 4e-3) and rel_l2(r32, no_lang) > 1e-2
