"""CPU restatement (test infrastructure) of the resize step of the reference's input transform, row N2 of SURVEY.md 8f.

The reference resizes with `T.Resize(img_size, T.InterpolationMode.BICUBIC)` on a PIL image (strhub/data/module.py:77,
applied at read.py:41-43 and in the LMDB dataset, strhub/data/dataset.py:132-148), i.e. Pillow's `Image.resize(..., BICUBIC)`:
the third-party routine ImagingResample (Pillow, src/libImaging/Resample.c — not vendored in /root/reference; the
version installed in this container pins the behaviour, see tests/test_resize.py).  Published algorithm, restated:

  * separable; horizontal pass first, its result stored as uint8, then the vertical pass (a pass whose size does not
    change is skipped);
  * per output index xx: center = (xx + 0.5) * scale, support = 2 * max(scale, 1), taps x in
    [int(center - support + 0.5), int(center + support + 0.5)) clipped to the image, weights bicubic(a = -0.5) of
    (x - center + 0.5) / max(scale, 1) normalised to sum 1 in float64, then fixed point with 22 fractional bits
    (round half away from zero);
  * pixel = clip8((2^21 + sum_x in[x] * k[x]) >> 22).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


def coefficients(in_size: int, out_size: int):
    """(bounds [out, 2] = (first tap, tap count), fixed-point weights [out, ksize] int32) of one pass."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One resampling pass over `axis` of a uint8 [H, W, C] array."""
    in_size = img.shape[axis]
    if in_size == out_size:
        return img
    bounds, kk = coefficients(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[xx, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [H, W, 3] -> uint8 [out_h, out_w, 3], bit-exact with Pillow's Image.resize((out_w, out_h), BICUBIC)."""
    assert img.dtype == np.uint8 and img.ndim == 3
    return _pass(_pass(img, out_w, 1), out_h, 0)
