"""2x2 camera tiling / untiling (pure layout) with the reference's semantics (controller/utils/misc.py:6-47).

``tile_images`` / ``untile_images`` keep the reference signatures (PIL in / PIL out, numpy NCHW per camera) so
controller/eval_genima.py:186 and :224-228 can call them unchanged; ``tile_u8`` / ``untile_u8`` are the array forms the
batched pipeline uses directly.
"""
from __future__ import annotations

import numpy as np

CROP_ORDER = ((0, 0, 256, 256), (256, 0, 512, 256), (0, 256, 256, 512), (256, 256, 512, 512))  # (l, t, r, b)


def tile_u8(rgbs, num_frames: int) -> np.ndarray:
    """rgbs: sequence of 4*num_frames uint8 HWC arrays [256,256,3], camera-major -> uint8 [num_frames, 512, 512, 3]."""
    out = np.zeros((num_frames, 512, 512, 3), dtype=np.uint8)
    for t in range(num_frames):
        for cam, (l, tp, r, b) in enumerate(CROP_ORDER):
            a = np.asarray(rgbs[cam * num_frames + t], dtype=np.uint8)
            assert a.shape == (256, 256, 3), "For tiling, image sizes must be 256x256"
            out[t, tp:b, l:r] = a
    return out


def untile_u8(tiled: np.ndarray, cameras) -> dict:
    """tiled uint8 [frames, 512, 512, 3] -> {camera: uint8 [frames, 3, 256, 256]}."""
    assert tiled.shape[1:] == (512, 512, 3), "For untiling, image sizes must be 512x512"
    out = {}
    for cam_idx, cam in enumerate(cameras):
        l, tp, r, b = CROP_ORDER[cam_idx]
        out[cam] = np.ascontiguousarray(np.transpose(tiled[:, tp:b, l:r], (0, 3, 1, 2)))
    return out


def tile_images(rgbs, num_frames):
    from PIL import Image

    assert isinstance(rgbs[0], Image.Image), "Images must be PIL Images"
    assert rgbs[0].size == (256, 256), "For tiling, image sizes must be 256x256"
    arr = tile_u8([np.asarray(im.convert("RGB")) for im in rgbs], num_frames)
    return [Image.fromarray(a) for a in arr]


def untile_images(gen_images, cameras, resize_transform):
    assert gen_images[0].size == (512, 512), "For untiling, image sizes must be 512x512"
    untiled = {c: [] for c in cameras}
    for img in gen_images:
        for cam_idx, cam in enumerate(cameras):
            g = resize_transform(img.crop(CROP_ORDER[cam_idx]))
            untiled[cam].append(np.transpose(np.expand_dims(np.array(g), axis=0), (0, 3, 1, 2)))
    return {c: np.concatenate(v, axis=0) for c, v in untiled.items()}
