"""Generate tests/golden/tiling_golden.npz by IMPORTING the reference's own controller/utils/misc.py.

Run in the build container only (``/root/reference`` does not exist on the GPU box):
    python tests/golden/make_tiling_golden.py
Inputs are seeded synthetic camera frames from genima_amd.weights.counter_bytes; outputs are what the
reference's ``tile_images`` / ``untile_images`` return (controller/utils/misc.py:6-47), stored as data.
"""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, "/root/reference/controller")

from genima_amd.weights import counter_bytes  # noqa: E402
from utils.misc import tile_images, untile_images  # noqa: E402  (the reference module)

CAMS = ["wrist", "front", "right_shoulder", "left_shoulder"]
num_frames = 2
rgbs = [counter_bytes(7, f"cam{c}_t{t}", 256 * 256 * 3).reshape(256, 256, 3)
        for c in range(4) for t in range(num_frames)]
tiled = tile_images([Image.fromarray(a) for a in rgbs], num_frames)
tiled_np = np.stack([np.array(t) for t in tiled])
unt = untile_images(tiled, CAMS, lambda im: im)  # Resize(256)+CenterCrop(256) is the identity on a 256 crop
out = {
    "num_frames": np.array(num_frames),
    "tiled_small": tiled_np[:, 248:264, 248:264].copy(),       # the seam region, stored raw
    "tiled_sha256": np.frombuffer(hashlib.sha256(tiled_np.tobytes()).digest(), dtype=np.uint8),
}
for c in CAMS:
    out[f"untiled_sha256_{c}"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(unt[c]).tobytes()).digest(), dtype=np.uint8)
    out[f"untiled_shape_{c}"] = np.array(unt[c].shape)
np.savez_compressed(os.path.join(HERE, "tiling_golden.npz"), **out)
print("wrote tiling_golden.npz", tiled_np.shape, {c: unt[c].shape for c in CAMS})
