"""Oracle (test infrastructure only): per-frame paste-back of a swapped crop into its original frame.

Restates ``REFace/scripts/VFace_inference_batch.py:597-636`` in numpy.  The reference does this frame by frame on the host
with Pillow (``Image.fromarray(..).resize(.., BILINEAR)``, ``Image.transform(.., PERSPECTIVE, coeffs, BILINEAR)``,
``alpha_composite``) and torchvision (``ToTensor``, ``Normalize``, ``Resize`` on a tensor = ``F.interpolate(mode="bilinear",
align_corners=False)``).  Pillow is a third-party dependency of the reference (``REFace/environment.yml:175`` pins
``pillow==9.5.0``; ``requirements.txt:24`` says 9.0.1) and is not under ``/root/reference``: the functions below restate Pillow's
published algorithms (``src/libImaging/Resample.c``: ``precompute_coeffs`` / ``normalize_coeffs_8bpc`` / the two 8-bit passes;
``Geometry.c``: ``perspective_transform`` / ``bilinear_filter32RGB``; ``AlphaComposite.c``), and parity is PINNED by comparing
them with the Pillow installed in this image (12.2.0) on the reference's own call sequence --
``tests/test_oracle_golden.py::test_paste_*``.  The 8-bit bilinear resampling, the perspective transform and the alpha
composite are the same algorithms in Pillow 9.x and 12.2 (the fixed-point 8bpc resampler dates from 3.x, the pixel-centre
convention of ``Image.transform`` from 5.x), so pinning against 12.2.0 pins the reference's 9.5.0 behaviour too.

Arithmetic type of the quantisation (:597-608): ``to_u8`` / ``clamp01`` compute in float32 -- the reference under
``--precision full``.  Under its default ``--precision autocast`` the decoded tensor is float16 and torch / numpy round every
operation to float16; ``to_u8_half`` / ``clamp01_half`` restate that (pixels can differ by one between the two).

Nothing here is imported by the product path.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: 8-bit samples, 2 guard bits


def to_u8(x01: np.ndarray) -> np.ndarray:
    """``255. * x`` in float32, then ``astype(np.uint8)`` (truncation) -- VFace_inference_batch.py:606-608."""
    return (np.float32(255.0) * x01.astype(np.float32)).astype(np.uint8)


def clamp01(x: np.ndarray) -> np.ndarray:
    """``torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0)`` in float32 -- :597."""
    return np.clip((x.astype(np.float32) + np.float32(1.0)) / np.float32(2.0), np.float32(0.0), np.float32(1.0))


def clamp01_half(x: np.ndarray) -> np.ndarray:
    """:597 on a float16 tensor (``--precision autocast``): ``x + 1.0`` and ``/ 2.0`` each rounded to float16."""
    h = x.astype(np.float16)
    return np.clip(((h + np.float16(1.0)).astype(np.float16) / np.float16(2.0)).astype(np.float16), np.float16(0.0), np.float16(1.0))


def to_u8_half(x01: np.ndarray) -> np.ndarray:
    """:606-608 on the float16 array: ``255. * x`` stays float16 (a Python scalar does not widen a numpy array), then truncation."""
    return (np.float16(255.0) * x01.astype(np.float16)).astype(np.float16).astype(np.uint8)


def resample_coeffs(in_size: int, out_size: int):
    """Pillow's ``precompute_coeffs`` (bilinear filter, support 1) + ``normalize_coeffs_8bpc``: per output sample the first
    input index, the tap count and ``ksize`` fixed-point weights (22 fractional bits)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        bounds[xx] = (xmin, xmax)
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
    return bounds, kk


def _pass(img: np.ndarray, bounds, kk, axis: int) -> np.ndarray:
    """One 8-bit pass of ``ImagingResampleHorizontal_8bpc`` / ``Vertical``: ss = 2^21 + sum(sample * k); clip8(ss >> 22)."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for xx in range(bounds.shape[0]):
        x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(n):
            acc += src[x0 + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``Image.resize((out_w, out_h), Image.BILINEAR)`` of an RGB uint8 [H, W, 3] image: horizontal pass, then vertical
    (``ImagingResampleInner``), each skipped when that size does not change."""
    h, w = img.shape[:2]
    if out_w != w:
        img = _pass(img, *resample_coeffs(w, out_w), axis=1)
    if out_h != h:
        img = _pass(img, *resample_coeffs(h, out_h), axis=0)
    return img


def perspective_paste(swapped: np.ndarray, background: np.ndarray, coeffs) -> np.ndarray:
    """``projected = swapped.convert('RGBA') (alpha 255) .transform(background.size, PERSPECTIVE, coeffs, BILINEAR)`` followed
    by ``background.convert('RGBA').alpha_composite(projected)`` (:627-633), returned as RGB.  Pillow maps the CENTRE of output
    pixel (x, y) through the eight coefficients in double precision; a source point outside [0, w) x [0, h) leaves the pixel
    transparent (the background shows), inside it the four neighbours (clamped at the border) are blended in double and
    truncated to 8 bits, alpha stays 255 -- so the composite is a per-pixel select."""
    a = [float(v) for v in coeffs]
    H, W = background.shape[:2]
    sh, sw = swapped.shape[:2]
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64) + 0.5, np.arange(W, dtype=np.float64) + 0.5, indexing="ij")
    den = a[6] * xs + a[7] * ys + 1
    xin = (a[0] * xs + a[1] * ys + a[2]) / den
    yin = (a[3] * xs + a[4] * ys + a[5]) / den
    inside = (xin >= 0.0) & (xin < sw) & (yin >= 0.0) & (yin < sh)       # (a NaN compares false: outside)
    xin = np.where(inside, xin, 0.5) - 0.5
    yin = np.where(inside, yin, 0.5) - 0.5
    x = np.where(xin < 0.0, np.floor(xin), np.trunc(xin)).astype(np.int64)
    y = np.where(yin < 0.0, np.floor(yin), np.trunc(yin)).astype(np.int64)
    dx, dy = (xin - x)[..., None], (yin - y)[..., None]
    x0, x1 = np.clip(x, 0, sw - 1), np.clip(x + 1, 0, sw - 1)
    yc = np.clip(y, 0, sh - 1)
    s = swapped.astype(np.float64)
    v1 = s[yc, x0] + (s[yc, x1] - s[yc, x0]) * dx
    has2 = ((y + 1 >= 0) & (y + 1 < sh))[..., None]
    y1 = np.clip(y + 1, 0, sh - 1)
    v2 = np.where(has2, s[y1, x0] + (s[y1, x1] - s[y1, x0]) * dx, v1)
    v = (v1 + (v2 - v1) * dy).astype(np.uint8)
    return np.where(inside[..., None], v, background)


def normalise_frame(frame_u8: np.ndarray) -> np.ndarray:
    """``ToTensor`` + ``Normalize(0.5, 0.5)`` (:48-56): [H, W, 3] uint8 -> [3, H, W] float32 in [-1, 1]."""
    t = frame_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    return (t - np.float32(0.5)) / np.float32(0.5)


def resize_bilinear_f32(t: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """``transforms.Resize([H, W])`` on a TENSOR (torchvision 0.14: no antialias) = ``F.interpolate(mode='bilinear',
    align_corners=False)`` (:612): float32 arithmetic as ATen's ``upsample_bilinear2d`` CPU/CUDA kernels do it."""
    c, h, w = t.shape
    f32 = np.float32

    def idx(out_n, in_n):
        scale = f32(in_n) / f32(out_n)
        src = np.maximum(scale * (np.arange(out_n, dtype=np.float32) + f32(0.5)) - f32(0.5), f32(0.0)).astype(np.float32)
        i0 = np.minimum(src.astype(np.int64), in_n - 1)
        i1 = np.minimum(i0 + 1, in_n - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, (f32(1.0) - l1).astype(np.float32), l1

    y0, y1, hy0, hy1 = idx(out_h, h)
    x0, x1, wx0, wx1 = idx(out_w, w)
    t = t.astype(np.float32)
    top = wx0[None, None, :] * t[:, y0][:, :, x0] + wx1[None, None, :] * t[:, y0][:, :, x1]
    bot = wx0[None, None, :] * t[:, y1][:, :, x0] + wx1[None, None, :] * t[:, y1][:, :, x1]
    return (hy0[None, :, None] * top + hy1[None, :, None] * bot).astype(np.float32)
