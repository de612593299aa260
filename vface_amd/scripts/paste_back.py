"""Paste-back of the swapped crops into their original frames, on the GPU.

Mirrors the per-frame block of ``REFace/scripts/VFace_inference_batch.py:597-636``, which runs on the host with numpy, Pillow
and torchvision -- every decoded frame crosses PCIe as fp32 and the background frame crosses it three times.  Here a batch of
frames stays in HBM and each step is one launch of ``csrc/paste.hip`` (8-bit results bit-identical to Pillow's):

    reference (per frame i)                                                       here (whole batch)
    ----------------------------------------------------------------------------  ---------------------------------------------
    x = clamp((decode(z) + 1) / 2, 0, 1); (255. * x).astype(uint8)          :597-608   hip.frame_to_u8
    Image.fromarray(..).resize((1024, 1024), BILINEAR)                      :608       hip.resample_u8 (x pass, y pass)
    get_tensor()(orig) ; transforms.Resize([H, W])                          :611-612   hip.frame_normalise_resize
    encode_first_stage -> get_first_stage_encoding -> decode_first_stage    :615-617   the VAE engine (vae_engine.py)
    clamp, uint8, .resize((orig.shape[1], orig.shape[2]), BILINEAR)         :618-621   hip.frame_to_u8 + hip.resample_u8
    swapped.putalpha(255); .transform(orig.size, PERSPECTIVE, coeffs, BILINEAR);
    background.alpha_composite(projected)                                   :627-633   hip.perspective_paste

Only the tap tables of the resampling (Pillow ``precompute_coeffs`` / ``normalize_coeffs_8bpc``: a few KB per size pair) are
computed on the host, once per (in, out) size, and cached on the device.  There is no CPU fallback: without the HIP library
every call raises.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch

from .. import hip

PRECISION_BITS = 32 - 8 - 2      # Pillow Resample.c: 8-bit samples, 2 guard bits -> 22 fractional bits


def resample_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Tap tables of ``Image.resize(.., Image.BILINEAR)`` from ``in_size`` to ``out_size`` samples: ``bounds`` int32
    [out_size, 2] = (first input sample, taps) and ``kk`` int32 [out_size, ksize] = 22-bit fixed-point weights, as Pillow's
    ``precompute_coeffs`` (triangle filter, support = max(scale, 1), window centred on (o + 0.5) * scale, weights normalised to
    sum 1 in double) and ``normalize_coeffs_8bpc`` (round half away from zero) produce them."""
    if in_size <= 0 or out_size <= 0:
        raise ValueError("resample_coeffs: sizes must be positive")
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = filterscale            # bilinear: filter support 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    inv = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    w = np.zeros((out_size, ksize), np.float64)
    total = np.zeros(out_size, np.float64)
    for x in range(ksize):           # sequential accumulation, tap by tap, as the C loop sums it
        a = np.abs((x + xmin - center + 0.5) * inv)
        col = np.where((x < xmax) & (a < 1.0), 1.0 - a, 0.0)
        w[:, x] = col
        total = total + col
    nz = total != 0.0
    w[nz] = w[nz] / total[nz, None]
    w[np.arange(ksize)[None, :] >= xmax[:, None]] = 0.0
    fixed = w * float(1 << PRECISION_BITS)
    kk = np.where(w < 0, np.trunc(-0.5 + fixed), np.trunc(0.5 + fixed)).astype(np.int32)
    return np.stack([xmin, xmax], 1).astype(np.int32), kk


class PasteBack:
    """Batch paste-back on one device.  ``encode_decode`` is the background round trip of :615-617 -- a callable
    ``[F, 3, H, W] fp32 in [-1, 1] -> decoded [F, 3, H, W]`` (``PasteBack.vae_round_trip(model)`` builds it from a
    ``LatentDiffusion`` with a first stage); ``None`` pastes over the untouched original frame instead (no VAE built).
    ``half_arithmetic``: quantise float16 decoder outputs with float16 arithmetic, as the reference's default
    ``--precision autocast`` run does (``hip.frame_to_u8``); the default is the fp32 arithmetic of ``--precision full``."""

    def __init__(self, H: int = 512, W: int = 512, canvas: int = 1024, device="cuda:0",
                 encode_decode: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, half_arithmetic: bool = False):
        self.H, self.W, self.canvas = H, W, canvas
        self.half_arithmetic = half_arithmetic
        self.device = torch.device(device)
        self.encode_decode = encode_decode
        self._tables: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor]] = {}

    @staticmethod
    def vae_round_trip(model) -> Callable[[torch.Tensor], torch.Tensor]:
        def run(x):
            z = model.get_first_stage_encoding(model.encode_first_stage(x))      # :615-616
            return model.decode_first_stage(z)                                    # :617
        return run

    def _table(self, in_size: int, out_size: int):
        key = (in_size, out_size)
        if key not in self._tables:
            b, k = resample_coeffs(in_size, out_size)
            self._tables[key] = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device))
        return self._tables[key]

    def resize_u8(self, frames: torch.Tensor, out_w: int, out_h: int) -> torch.Tensor:
        """``Image.resize((out_w, out_h), Image.BILINEAR)`` of uint8 [F, H, W, 3] frames: x pass, then y pass (each skipped when
        that size is unchanged, as Pillow does)."""
        _, h, w, _ = frames.shape
        if out_w != w:
            frames = hip.resample_u8(frames, out_w, 0, *self._table(w, out_w))
        if out_h != h:
            frames = hip.resample_u8(frames, out_h, 1, *self._table(h, out_h))
        return frames

    def swapped_canvas(self, x_samples: torch.Tensor) -> torch.Tensor:
        """Decoded crops [F, 3, H, W] in [-1, 1] -> the 1024 x 1024 uint8 images the reference saves and projects (:597-608)."""
        return self.resize_u8(self._to_u8(x_samples), self.canvas, self.canvas)

    def _to_u8(self, x: torch.Tensor) -> torch.Tensor:
        return hip.frame_to_u8(x, half_arithmetic=self.half_arithmetic and x.dtype == torch.float16)

    def background(self, frames_u8: torch.Tensor) -> torch.Tensor:
        """The frame the crop is pasted over: the original after the encode / decode round trip "to get the consistent output for
        background" (:610-623).  NOTE :623 passes ``(shape[1], shape[2])`` = (height, width) of the original to ``Image.resize``,
        which takes (width, height): the result only matches the original's size -- and ``alpha_composite`` only accepts it --
        for SQUARE frames; as there, a non-square frame raises ValueError("images do not match")."""
        _, Ho, Wo, _ = frames_u8.shape
        if self.encode_decode is None:
            return frames_u8.clone()
        x = hip.frame_normalise_resize(frames_u8, self.H, self.W)
        rec = self.encode_decode(x)
        out = self.resize_u8(self._to_u8(rec), Ho, Wo)      # (width, height) := (orig height, orig width), :623
        if tuple(out.shape[1:3]) != (Ho, Wo):
            raise ValueError("images do not match")
        return out

    def paste(self, x_samples: torch.Tensor, frames_u8: torch.Tensor, inv_transforms) -> torch.Tensor:
        """``x_samples`` decoded crops [F, 3, H, W]; ``frames_u8`` the original frames uint8 [F, Ho, Wo, 3] (device);
        ``inv_transforms`` [F, 8] the rows of ``inv_transforms_all`` (:625).  Returns the pasted frames uint8 [F, Ho, Wo, 3]."""
        if frames_u8.device != self.device or x_samples.device != self.device:
            raise hip.VFaceHipError("paste-back runs on the GPU: frames and samples must be device tensors")
        F_ = x_samples.shape[0]
        co = torch.as_tensor(np.asarray(inv_transforms, dtype=np.float64).reshape(F_, 8)).to(self.device)
        crop = self.swapped_canvas(x_samples)
        frame = self.background(frames_u8.contiguous())
        return hip.perspective_paste(crop, frame, co)
