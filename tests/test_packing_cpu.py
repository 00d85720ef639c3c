"""-m "not gpu": the packed kernel layout is an invertible re-arrangement of the diffusers state dict (what the trainer's checkpoints rely on)."""
import torch

from genima_amd import configs, schema, weights
from genima_amd.packing import pack_state_dict, unpack_state_dict


def _roundtrip(sch, seed):
    sd = weights.synth_state_dict(sch, seed)
    packed = pack_state_dict(sd, "cpu", dtype=torch.float32)
    meta = packed.pop("__meta__")
    # the trainer drops the entries the fused tensors cover; the inverse must not need them
    for k in [k for k in packed if k.endswith((".attn1.to_q.weight", ".attn1.to_k.weight", ".time_emb_proj.weight", ".time_emb_proj.bias"))]:
        del packed[k]
    back = unpack_state_dict(packed, sch, meta.get("temb_slices"))
    assert list(back) == list(sch)
    for k in sch:
        assert torch.equal(back[k], sd[k]), k


def test_controlnet_and_unet_pack_unpack_roundtrip():
    fam = configs.family("tiny")
    _roundtrip(schema.controlnet_schema(fam["controlnet"]), 2)
    _roundtrip(schema.unet_schema(fam["unet"]), 1)
