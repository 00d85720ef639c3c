"""Generate tests/golden/clip_text_golden.npz with the INSTALLED third-party ``transformers`` CLIPTextModel.

This is the one piece of the hot path whose pinned implementation family (transformers; the reference
pins 4.38.0, poetry.lock:3071-3072, call site diffusion/train_controlnet_genima.py:1042-1047, :1362) is
importable in the build container, so it pins the oracle's CLIP text tower to a real implementation:
same seeded synthetic weights -> HF forward -> stored ``last_hidden_state``.  Run in the build container:
    python tests/golden/make_clip_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from transformers import CLIPTextConfig, CLIPTextModel  # noqa: E402

from genima_amd import configs, schema, weights  # noqa: E402

out = {}
for tag, cfg in (("gelu", configs.TINY_TEXT), ("quick_gelu", dict(configs.TINY_ACT_CLIP_TEXT, projection_dim=0))):
    sd = weights.synth_state_dict(schema.clip_text_schema(cfg), seed=11)
    hf_cfg = CLIPTextConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"],
                            intermediate_size=cfg["intermediate_size"], num_hidden_layers=cfg["num_hidden_layers"],
                            num_attention_heads=cfg["num_attention_heads"], max_position_embeddings=77,
                            hidden_act=cfg["hidden_act"], layer_norm_eps=cfg["layer_norm_eps"],
                            bos_token_id=cfg["vocab_size"] - 2, eos_token_id=cfg["vocab_size"] - 1, pad_token_id=0)
    m = CLIPTextModel(hf_cfg).eval()
    own = set(m.state_dict().keys())
    if not any(k.startswith("text_model.") for k in own):  # transformers >= 5 dropped the prefix
        sd = {k[len("text_model."):]: v for k, v in sd.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
    V = cfg["vocab_size"]
    ids = np.zeros((2, 77), dtype=np.int64)
    ids[0, :14] = [V - 2] + [320 + i for i in range(12)] + [V - 1]
    ids[1, :9] = [V - 2] + [400 + 3 * i for i in range(7)] + [V - 1]
    with torch.no_grad():
        y = m(torch.from_numpy(ids))[0].numpy()
    out[f"{tag}_ids"] = ids
    out[f"{tag}_last_hidden_state"] = y.astype(np.float32)
np.savez_compressed(os.path.join(HERE, "clip_text_golden.npz"), **out)
print({k: v.shape for k, v in out.items()})
