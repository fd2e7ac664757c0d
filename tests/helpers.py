"""Shared synthetic inputs and comparison helpers for the parity tests."""
import numpy as np


def rand_image(rows, cols, seed=1234):
    """i.i.d. uniform [0,1) f32, default_rng(seed)."""
    return np.random.default_rng(seed).random((rows, cols), dtype=np.float32)


def smooth_image(rows, cols):
    """three oriented sinusoids + a step edge: well-conditioned dominant orientation"""
    y, x = np.mgrid[0:rows, 0:cols].astype(np.float32)
    img = (0.5 + 0.2 * np.sin(0.21 * x + 0.05 * y) + 0.15 * np.sin(0.07 * x - 0.19 * y)
           + 0.1 * np.cos(0.13 * (x + y)))
    img = img + 0.25 * (x > 0.6 * cols)
    return img.astype(np.float32)


def angle_diff(a, b, period):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)) % period
    return np.minimum(d, period - d)


# sizes the reference semantics make interesting: 1x1, thinner than the halo, odd, ragged
EDGE_SHAPES = [(1, 1), (1, 7), (7, 1), (2, 2), (3, 5), (4, 4), (5, 3), (9, 9), (8, 64), (17, 31),
               (64, 48), (33, 65), (70, 129), (185, 256)]
