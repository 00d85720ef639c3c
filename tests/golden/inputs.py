"""Deterministic synthetic inputs shared by the golden generators (tests/golden/make_*.py) and the tests that replay them, so
incompressible random images need not be stored in the fixtures."""
import numpy as np


def pattern_u8(shape, salt: int = 0) -> np.ndarray:
    """uint8 [B, C, H, W]: diagonal ramps + a coarse checker + a per-(b, c) offset -- structured like a rendered scene (flat areas,
    edges), a pure function of the indices."""
    B, C, H, W = shape
    b, c, y, x = np.meshgrid(np.arange(B), np.arange(C), np.arange(H), np.arange(W), indexing="ij")
    v = 3 * x + 5 * y + 53 * c + 101 * b + 17 * salt + 64 * (((x >> 4) + (y >> 4) + c) & 1) + ((x * y) >> 5)
    return (v & 255).astype(np.uint8)
