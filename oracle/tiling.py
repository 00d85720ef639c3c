"""CPU oracle (numpy): 2x2 camera tiling / untiling, restating controller/utils/misc.py:6-47.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pinned bit-exactly by golden vectors produced by
importing the reference's own ``controller/utils/misc.py`` (tests/golden/make_tiling_golden.py).
Images are uint8 HWC arrays; PIL ``paste((x, y))`` places a tile at column x, row y.
"""
from __future__ import annotations

import numpy as np

# (x, y) paste offsets in camera order -- controller/utils/misc.py:13-16
TILE_ORIGINS = ((0, 0), (256, 0), (0, 256), (256, 256))


def tile_images(rgbs, num_frames: int, view: int = 256):
    """rgbs: list of 4*num_frames HWC uint8 [view,view,3], camera-major (rgbs[cam*num_frames+t])."""
    out = []
    for t in range(num_frames):
        canvas = np.zeros((2 * view, 2 * view, 3), dtype=np.uint8)
        for cam, (x, y) in enumerate(TILE_ORIGINS):
            x, y = x * view // 256, y * view // 256
            canvas[y:y + view, x:x + view] = rgbs[cam * num_frames + t]
        out.append(canvas)
    return out


def untile_images(gen_images, cameras, view: int = 256):
    """gen_images: list of HWC uint8 [2v,2v,3] -> dict cam -> uint8 [frames,3,v,v]
    (crop order controller/utils/misc.py:25-30; the half-resolution Resize+CenterCrop of
    controller/agent/diffusion_agent.py:55-62 is the identity on a v x v crop)."""
    out = {c: [] for c in cameras}
    for img in gen_images:
        for cam_idx, cam in enumerate(cameras):
            x, y = TILE_ORIGINS[cam_idx]
            x, y = x * view // 256, y * view // 256
            crop = img[y:y + view, x:x + view]
            out[cam].append(np.transpose(crop[None], (0, 3, 1, 2)))
    return {c: np.concatenate(v, axis=0) for c, v in out.items()}
