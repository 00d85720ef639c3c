"""-m "not gpu": the ACT controller's checkpoint contract in the reference's direction:
``GenimaACT.state_dict()`` lists RoboBase's key paths, the reference's unchanged gate passes on a RoboBase-shaped checkpoint, and the
``save_snapshot`` payload round-trips."""
import os
from collections import OrderedDict

import pytest
import torch

from genima_amd import configs, schema, weights
from genima_amd.act import GenimaACT, act_schema, clip_hf_to_openai, clip_openai_to_hf, robobase_key_map, robobase_key_names
from genima_amd.act_training import act_train_schema
from genima_amd.harness import load_controller_ckpt, save_snapshot


def _robobase_agent_state(cfg, seed):
    """What ``ActBCAgent.state_dict()`` of the reference holds: nn.Module does not dedupe shared submodules, so every weight sits under
    ``encoder.* / actor.encoder_model.*`` or ``actor_model.* / actor.actor_model.*`` (controller/method/genima_act.py:221-249), the ResNet
    under the DETR Joiner's ``backbone.0.body``; plus things this package does not hold (``fc`` of the torchvision ResNet)."""
    sd = weights.synth_state_dict(act_train_schema(cfg), seed)
    out = OrderedDict()
    for k, v in sd.items():
        if k.startswith("backbone."):
            for p in ("encoder.", "actor.encoder_model."):
                out[p + "backbone.0.body." + k[len("backbone."):]] = v
        elif k.startswith("input_proj."):
            out["encoder." + k] = out["actor.encoder_model." + k] = v
        elif k.startswith("projection_layer."):
            out["actor." + k] = v
        else:
            out["actor_model." + k] = out["actor.actor_model." + k] = v
    return sd, out


def test_key_names_are_the_inverse_of_the_key_map():
    cfg = dict(configs.TINY_ACT_POLICY, frame_stack=2)
    for k in act_train_schema(cfg):
        names = robobase_key_names(k)
        assert len(names) == (1 if k.startswith("projection_layer.") else 2)
        assert all(robobase_key_map(n) == k for n in names), (k, names)
    assert robobase_key_names("task_proj.weight", {"task_proj": "proj_text_emb"}) == ["actor_model.proj_text_emb.weight", "actor.actor_model.proj_text_emb.weight"]
    assert robobase_key_map("encoder.backbone.0.body.fc.weight") is None
    assert robobase_key_map("actor.encoder_model.backbone.0.body.bn1.num_batches_tracked") is None


@pytest.mark.parametrize("frame_stack", [1, 2])
def test_reference_gate_and_snapshot_round_trip(tmp_path, frame_stack):
    cfg, ccfg = dict(configs.TINY_ACT_POLICY, frame_stack=frame_stack), configs.TINY_ACT_CLIP_TEXT
    own, agent_sd = _robobase_agent_state(cfg, 71)
    ckpt = tmp_path / "snapshots" / "exp" / "latest.pt"
    os.makedirs(ckpt.parent)
    torch.save({"cfg": {"x": 1}, "_epoch": 3, "_num_iters": 40, "agent": agent_sd}, ckpt)   # train_act.py:262-279's payload

    agent = GenimaACT(cfg, None, ccfg, None, device="cpu", seed=5)   # host-side object: no device work in this test
    keys = list(agent.state_dict().keys())
    assert set(keys) == set(agent_sd.keys()), sorted(set(keys) ^ set(agent_sd.keys()))[:6]
    assert not any("clip" in k for k in keys)  # the reference loads clip lazily on the first act()
    # --- controller/eval_genima.py:91-103, replayed literally ---
    checkpoint = torch.load(ckpt, map_location="cpu", weights_only=False)
    missing_keys = [k for k in agent.state_dict().keys() if k not in checkpoint["agent"].keys() and "clip" not in k]
    assert missing_keys == []
    missing, unexpected = agent.load_state_dict(checkpoint["agent"], strict=False)
    assert missing == [] and unexpected == []
    for k, v in own.items():
        got = agent._sd[k] if k in agent._sd else agent._sd_train[k]
        assert torch.equal(got, v.float()), k
    # the helper that restates the gate raises on a checkpoint that lacks a key, as the reference does
    bad = dict(agent_sd)
    del bad["actor.actor_model.action_head.weight"]
    torch.save({"agent": bad}, tmp_path / "bad.pt")
    with pytest.raises(ValueError, match="Missing keys in controller checkpoint"):
        load_controller_ckpt(agent, tmp_path / "bad.pt")

    # --- controller/train_act.py:262-279: snapshot payload; clip_model keys stripped even after the text tower was used ---
    agent._clip_used = True
    assert any(k.startswith("clip_model.") for k in agent.state_dict())
    out = tmp_path / "snapshots" / "exp" / "100.pt"
    save_snapshot(agent, out, cfg={"experiment_name": "exp"}, epoch=100, num_iters=1234)
    payload = torch.load(out, weights_only=False)
    assert sorted(payload) == ["_epoch", "_num_iters", "agent", "cfg"] and payload["_epoch"] == 100 and payload["_num_iters"] == 1234
    assert set(payload["agent"]) == set(agent_sd) and all(torch.equal(payload["agent"][k], agent_sd[k].float()) for k in agent_sd)
    fresh = GenimaACT(cfg, None, ccfg, None, device="cpu", seed=9)
    load_controller_ckpt(fresh, out)
    assert all(torch.equal(fresh._sd[k], agent._sd[k]) for k in agent._sd)


def test_clip_openai_names_round_trip():
    ccfg = configs.TINY_ACT_CLIP_TEXT
    hf = weights.synth_state_dict(schema.clip_text_schema(ccfg), 3)
    oa = clip_hf_to_openai(hf)
    L, d = ccfg["num_hidden_layers"], ccfg["hidden_size"]
    assert oa["transformer.resblocks.0.attn.in_proj_weight"].shape == (3 * d, d) and oa["text_projection"].shape == (d, ccfg["projection_dim"])
    assert f"transformer.resblocks.{L - 1}.mlp.c_proj.bias" in oa and "positional_embedding" in oa
    back = clip_openai_to_hf({"clip_model." + k: v for k, v in oa.items()})
    assert set(back) == set(hf) and all(torch.equal(back[k], hf[k]) for k in hf)
