"""Simulator-free restatement of ONE control step of the reference's evaluation loop (controller/eval_genima.py:162-248), the
counterpart of BASELINE.json configs[0] (SURVEY.md section 8a row a1 / 8d "Config 1"): observation dict -> PIL views -> prompt ->
2x2 tiling -> ``diffusion_agent.infer`` -> untile -> observation overwrite -> ``controller_agent.act`` -> actions.

RLBench / CoppeliaSim / Hydra stay out of scope; what this module pins is the data contract between the simulator loop and the two
plugins, so the agents of this package can be exercised (and the reference's loop body checked against them) without a simulator:
  * views are gathered camera-major, frame-minor (``rgbs[cam * num_frames + t]``, :167-173) and tiled per frame;
  * one prompt per frame, ``f"tiled perspectives of a robot arm executing '{goal}'"`` (:178);
  * the SAME generator object is passed once per tiled image (``generator * len(tiled_images)``, :130-135, :209);
  * the pipeline output is indexed ``[0]`` for the list of images (:215, :225) and untiled with the agent's
    ``transform_to_half_resolution`` (:226-230);
  * the four camera keys ``wrist / front / right_shoulder / left_shoulder`` are overwritten by name (:231-234), every observation
    entry gets a leading batch axis on the device (:237-240), and ``act(...)[0]`` is the ``[queries, action_dim]`` plan (:243-248).
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch

from .tiling import tile_images, untile_images

NEGATIVE_PROMPT = "monochrome, lowres, bad anatomy, worst quality, low quality"
OVERWRITTEN_CAMERAS = ("wrist", "front", "right_shoulder", "left_shoulder")


def make_prompt(goal: str) -> str:
    return f"tiled perspectives of a robot arm executing '{goal}'"


def control_step(diffusion_agent, controller_agent, obs: Dict[str, np.ndarray], goal: str, cameras: Sequence[str], num_frames: int,
                 generator: List[torch.Generator], num_diffusion_steps: int, guidance_scale: float, device, episode_step: int = 0):
    """obs: ``{'<cam>_rgb': uint8 [fs, 3, 256, 256], 'low_dim_state': f32 [fs, S], 'lang_tokens': int [fs, 1|.., 77], ...}`` as the
    RoboBase env wrapper hands it over.  Returns ``(actions np.float32 [queries, action_dim], obs_after, tiled_in, tiled_out)``;
    ``obs_after`` holds the device tensors the controller saw (camera images replaced by the generated joint-target views)."""
    from PIL import Image

    rgbs = [Image.fromarray(np.transpose(obs[f"{cam}_rgb"][t], (1, 2, 0))) for cam in cameras for t in range(num_frames)]
    prompts = [make_prompt(goal)] * num_frames
    negative = [NEGATIVE_PROMPT] * num_frames
    tiled_in = tile_images(rgbs, num_frames)
    with torch.inference_mode():
        out = diffusion_agent.infer(images=tiled_in, prompts=prompts, negative_prompts=negative,
                                    num_inference_steps=num_diffusion_steps, guidance_scale=guidance_scale,
                                    generator=generator * len(tiled_in))
        tiled_out = out[0]
        untiled = untile_images(tiled_out, cameras, diffusion_agent.transform_to_half_resolution)
        obs = dict(obs)
        for cam in OVERWRITTEN_CAMERAS:
            obs[f"{cam}_rgb"] = untiled[cam]
        obs_dev = {k: torch.from_numpy(np.asarray(v)).to(device).unsqueeze(0) for k, v in obs.items()}
        actions = controller_agent.act(obs_dev, step=episode_step, eval_mode=True)[0]
    return actions.detach().float().cpu().numpy(), obs_dev, tiled_in, tiled_out


def load_controller_ckpt(controller_agent, checkpoint_path, device="cpu"):
    """``Workspace.load_controller_ckpt`` (controller/eval_genima.py:91-103), statement for statement: the gate that every non-clip key
    of ``controller_agent.state_dict()`` is present in ``checkpoint["agent"]``, then ``load_state_dict(..., strict=False)``."""
    checkpoint = torch.load(checkpoint_path, map_location=device, weights_only=False)
    missing_keys = [k for k in controller_agent.state_dict().keys() if k not in checkpoint["agent"].keys() and "clip" not in k]
    if len(missing_keys) > 0:
        raise ValueError(f"Missing keys in controller checkpoint: {missing_keys}")
    controller_agent.load_state_dict(checkpoint["agent"], strict=False)
    return checkpoint


def save_snapshot(controller_agent, path, cfg=None, epoch: int = 0, num_iters: int = 0):
    """``ControllerWorkspace.save_snapshot`` (controller/train_act.py:262-279): ``{"cfg", "_epoch", "_num_iters", "agent"}`` with the
    ``clip_model`` keys filtered out of the agent's state dict, written with ``torch.save``."""
    import os

    state_dict = {k: v for k, v in controller_agent.state_dict().items() if "clip_model" not in k}
    payload = {"cfg": cfg, "_epoch": epoch, "_num_iters": num_iters, "agent": state_dict}
    os.makedirs(os.path.dirname(os.path.abspath(str(path))), exist_ok=True)
    with open(path, "wb") as f:
        torch.save(payload, f)
    return payload
