"""Deterministic synthetic weights and inputs for the VFace denoising path.

There are no checkpoints on the GPU box (``last.ckpt`` needs network), so every
parity and bench run uses a portable, name-keyed fill: the tensor stored under a
given ``state_dict`` key is a pure function of (key, shape, seed).  The same fill
is applied to the reference modules when golden vectors are generated
(``tests/golden/make_golden.py``) and to this package's modules on the GPU box,
so weights never have to be committed.

Scales are variance preserving (uniform with std ``1/sqrt(fan_in)``) so activations stay
O(1) through the 860 M-parameter UNet in fp16, and the convolutions the reference
zero-initialises (``zero_module``: ``openaimodel.py:229-231,824``, ``attention.py:272``)
are filled like any other so residual branches are visible to parity tests.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Mapping, Tuple

import numpy as np
import torch


def _rng(key: str, seed: int) -> np.random.Generator:
    h = zlib.crc32(key.encode("utf-8")) & 0xFFFFFFFF
    return np.random.Generator(np.random.PCG64([h, seed & 0xFFFFFFFF]))


def synth_tensor(key: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    """fp32 tensor for state-dict entry ``key`` of ``shape`` (see module docstring)."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    u = _rng(key, seed).random(n, dtype=np.float32) * 2.0 - 1.0  # U(-1, 1)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "bias":
        u *= 0.05
    elif len(shape) <= 1:  # norm gains
        u = 1.0 + 0.1 * u
    else:
        fan_in = int(np.prod(shape[1:]))
        u *= math.sqrt(3.0 / fan_in)
    return torch.from_numpy(u.reshape(shape))


def synth_state_dict(shapes: Mapping[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(k, tuple(s), seed) for k, s in shapes.items()}


@torch.no_grad()
def fill_module_(module: torch.nn.Module, seed: int = 0, prefix: str = "") -> torch.nn.Module:
    """In-place deterministic fill of every parameter/buffer of ``module`` keyed by its state-dict name."""
    for k, v in module.state_dict().items():
        if not torch.is_floating_point(v):
            continue
        v.copy_(synth_tensor(prefix + k, tuple(v.shape), seed).to(v.dtype))
    return module


def synth_normal(tag: str, shape: Iterable[int], seed: int = 42) -> torch.Tensor:
    """Seeded N(0,1) fp32 tensor (latents / conditioning), portable across hosts."""
    shape = tuple(int(s) for s in shape)
    g = _rng(tag, seed)
    return torch.from_numpy(g.standard_normal(int(np.prod(shape)), dtype=np.float32).reshape(shape))


def synth_flow(num_pairs: int, h: int, w: int, seed: int = 42) -> torch.Tensor:
    """Smooth synthetic optical flow at latent resolution, ``[num_pairs, 2, h, w]`` fp32:
    per-pair constant translation in [-2, 2] px plus a 0.5 px sinusoid; channel 0 = dx, 1 = dy
    (``temporal_flow.py:40-53``)."""
    g = _rng("flow", seed)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    out = np.zeros((max(num_pairs, 0), 2, h, w), np.float32)
    for i in range(num_pairs):
        tx, ty = g.uniform(-2.0, 2.0, 2).astype(np.float32)
        ph = np.float32(g.uniform(0, 2 * math.pi))
        out[i, 0] = tx + 0.5 * np.sin(2 * math.pi * ys / h + ph)
        out[i, 1] = ty + 0.5 * np.cos(2 * math.pi * xs / w + ph)
    return torch.from_numpy(out)


def synth_mask(f: int, h: int, w: int) -> torch.Tensor:
    """Centred ellipse in {0,1}, ``[f,1,h,w]`` (stand-in for the face-parsing inpaint mask)."""
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    m = (((ys - (h - 1) / 2) / (0.38 * h)) ** 2 + ((xs - (w - 1) / 2) / (0.30 * w)) ** 2 <= 1.0).astype(np.float32)
    return torch.from_numpy(m)[None, None].repeat(f, 1, 1, 1)
