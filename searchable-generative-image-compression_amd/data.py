"""Deterministic synthetic inputs: there is no dataset offline.  `synth_images` gives
"natural-like" low-pass noise in [-1,1] (box-blurred uniform noise mixed with a little white noise) from
numpy's counter-based Philox stream, so every machine produces the same bytes."""
import numpy as np
import torch


def _box_blur(a, k):
    """separable box blur with edge replication, a: (..., H, W)"""
    pad = k // 2
    for axis in (-2, -1):
        ap = np.concatenate([np.repeat(np.take(a, [0], axis=axis), pad, axis=axis), a,
                             np.repeat(np.take(a, [-1], axis=axis), pad, axis=axis)], axis=axis)
        c = np.cumsum(ap, axis=axis, dtype=np.float64)
        z = np.zeros_like(np.take(c, [0], axis=axis))
        c = np.concatenate([z, c], axis=axis)
        n = a.shape[axis]
        hi = np.take(c, np.arange(k, k + n), axis=axis)
        lo = np.take(c, np.arange(0, n), axis=axis)
        a = ((hi - lo) / k).astype(np.float32)
    return a


def synth_images(B, H, W, seed=0, k=9):
    r = np.random.Generator(np.random.Philox(key=[0x5EED, seed]))
    base = r.random((B, 3, H, W), dtype=np.float32)
    smooth = _box_blur(base, k)
    smooth = (smooth - smooth.min()) / max(float(smooth.max() - smooth.min()), 1e-6)
    x = 0.85 * smooth + 0.15 * r.random((B, 3, H, W), dtype=np.float32)
    return torch.from_numpy((2.0 * x - 1.0).astype(np.float32)).clamp_(-1.0, 1.0)
