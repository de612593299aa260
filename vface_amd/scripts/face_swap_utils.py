"""``combine_fft_high_low`` (``REFace/scripts/face_swap_utils.py:425-464``) and ``mix_source_and_target``
(``:189-199``) as stand-alone GPU functions.

Inside the UNet these never run as separate ops: FSAI is folded into the q,k projection weights
(``vface_amd.packing.fold_fsai``).  The stand-alone form applies the same linear map as one MFMA GEMM over
``[q2 | q1]`` (K = 2d) with the two d x d band matrices; inputs and matrices are rounded to the
16-bit compute type, the result is returned in fp32 like the reference's.
"""
from __future__ import annotations

import math

import torch

from .. import hip

_BAND_CACHE = {}


def _band_matrices(d: int, split_ratio: float, device, dtype) -> torch.Tensor:
    """Wt [d, 2d] = [A_lo^T | A_hi^T] so that out = [q2 | q1] @ Wt^T."""
    key = (d, round(float(split_ratio), 9), str(device), dtype)
    if key not in _BAND_CACHE:
        s = int(d * split_ratio)
        eye = torch.eye(d, dtype=torch.float64)
        f = torch.fft.fft(eye, dim=-1)

        def band(lo, hi):
            keep = torch.zeros_like(f)
            keep[:, lo:hi] = f[:, lo:hi]
            return torch.fft.ifft(keep, dim=-1).real  # row j = response to a unit input in channel j -> A[j, :]

        a_lo, a_hi = band(0, s), band(s, d)
        _BAND_CACHE[key] = torch.cat([a_lo.t(), a_hi.t()], 1).to(device=device, dtype=dtype).contiguous()
    return _BAND_CACHE[key]


def combine_fft_high_low(q1: torch.Tensor, q2: torch.Tensor, split_ratio: float = 0.5,
                         compute_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """High-frequency bins [int(d*ratio), d) from ``q1``, low-frequency bins from ``q2``; ``[..., d]`` fp32 out."""
    if not (q1.is_cuda and q2.is_cuda):
        raise hip.VFaceHipError("combine_fft_high_low needs CUDA tensors: no CPU fallback on the VFace path")
    shape, d = q1.shape, q1.shape[-1]
    if d % 64:
        raise hip.VFaceHipError("combine_fft_high_low: channel count must be a multiple of 64")
    dev = q1.device

    def to16(t):
        t = t.reshape(-1, d).contiguous()
        if t.dtype == compute_dtype:
            return t
        o = torch.empty(t.shape, dtype=compute_dtype, device=dev)
        hip.cast_f32(t.float(), o)
        return o

    a, b = to16(q2), to16(q1)
    out = torch.empty(a.shape[0], d, dtype=torch.float32, device=dev)
    hip.gemm(a, _band_matrices(d, split_ratio, dev, compute_dtype), out, M=a.shape[0], N=d, K=2 * d, lda=d, ldc=d,
             ldw=2 * d, a2=b, lda2=d, k1=d, flags=hip.EPI_OUT_F32)
    return out.reshape(shape)
