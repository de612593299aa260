"""Oracle (test infrastructure only): the VFace attention hook.

Restates ``REFace/ldm/models/pnp_utils.py:57-339`` (``register_spa_attn_injection`` and the replacement
``forward`` it installs on every ``attn1``) and the fusion helpers it calls from
``REFace/scripts/face_swap_utils.py`` and ``REFace/scripts/temporal_flow.py``.

The reference patches ``module.forward`` with a closure; the oracle is functional, so the "registry" is a
dict ``{attn1 module name -> HookCfg}`` that :func:`register_spa_attn_injection` updates with the same
arguments and the same selection rule (per-group ordinal ``block_indices``; a module that is not selected
keeps whatever closure it had, pnp_utils.py:292-304).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import flow as oflow


@dataclass
class HookCfg:
    switch_on: bool = True
    chunks: int = 3
    fusion: str = "replace"
    flow: Optional[Sequence[torch.Tensor]] = None
    split_ratio_fft: float = 0.8
    alpha: float = 0.8


# --------------------------------------------------------------------------- fusion helpers
def combine_fft_high_low(q1: torch.Tensor, q2: torch.Tensor, split_ratio: float = 0.5) -> torch.Tensor:
    """face_swap_utils.py:425-464.  1-D complex FFT along the channel axis; bins [0, int(d*ratio)) from
    ``q2`` (own branch), bins [int(d*ratio), d) from ``q1`` (structure branch); real part of the inverse.
    The splice is not Hermitian symmetric; the discarded imaginary part is non-zero."""
    q1 = q1.float()
    q2 = q2.float()
    f1 = torch.fft.fft(q1, dim=-1)
    f2 = torch.fft.fft(q2, dim=-1)
    d = q1.size(-1)
    s = int(d * split_ratio)
    comb = torch.zeros_like(f1)
    comb[..., :s] = f2[..., :s]
    comb[..., s:] = f1[..., s:]
    return torch.fft.ifft(comb, dim=-1).real.to(torch.float32)


def fsai_matrices(d: int, split_ratio: float, dtype=torch.float64):
    """The same map as :func:`combine_fft_high_low` written as two real d x d matrices:
    ``out = q2 @ A_lo + q1 @ A_hi`` with ``A_lo + A_hi = I``.  ``A[j, i] = Re(1/d * sum_{k in band}
    exp(2*pi*1j*k*(i-j)/d))`` (row index = input channel j, column = output channel i)."""
    s = int(d * split_ratio)
    idx = torch.arange(d, dtype=torch.float64)
    diff = idx.view(1, d) - idx.view(d, 1)  # i - j
    k_lo = torch.arange(0, s, dtype=torch.float64)
    k_hi = torch.arange(s, d, dtype=torch.float64)

    def band(ks):
        if ks.numel() == 0:
            return torch.zeros(d, d, dtype=torch.float64)
        ang = 2.0 * math.pi * diff.unsqueeze(-1) * ks / d
        return torch.cos(ang).sum(-1) / d

    return band(k_lo).to(dtype), band(k_hi).to(dtype)


def temporal_attention(x: torch.Tensor, window_size: int = 5, sigma: float = 1.0) -> torch.Tensor:
    """pnp_utils.py:59-90.  Gaussian-weighted mean over the frame axis, renormalised at clip ends."""
    T = x.shape[0]
    pad = window_size // 2
    out = torch.zeros_like(x)
    offs = torch.arange(-pad, pad + 1, dtype=torch.float32)
    g = torch.exp(-0.5 * (offs / sigma) ** 2)
    g = g / g.sum()
    for t in range(T):
        acc = 0.0
        wt = 0.0
        for i, o in enumerate(offs):
            j = t + int(o.item())
            if 0 <= j < T:
                acc = acc + g[i] * x[j]
                wt = wt + g[i]
        out[t] = acc / wt
    return out


def adain_fusion_for_attn(a: torch.Tensor, b: torch.Tensor, alpha: float = 0.71, normalized: bool = True):
    """face_swap_utils.py:372-389.  Per-token AdaIN over the channel axis; with ``normalized`` the result is
    divided by the GLOBAL std of the fused tensor (all frames, tokens, channels; unbiased)."""
    mean_a = a.mean(dim=-1, keepdim=True)
    std_a = a.std(dim=-1, keepdim=True)
    mean_b = b.mean(dim=-1, keepdim=True)
    std_b = b.std(dim=-1, keepdim=True)
    fused = (a - mean_a) / (std_a + 1e-5) * std_b + mean_b
    if normalized:
        return fused / (fused.std() + 1e-5)
    return alpha * fused


def mix_source_and_target(target: torch.Tensor, source: torch.Tensor, alpha: float = 0.5):
    """face_swap_utils.py:189-199."""
    return (1 - alpha) * source + alpha * target


# --------------------------------------------------------------------------- hooked attention
def apply_fusion(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, cfg: HookCfg, spatial_hw=None):
    """In-place edit of freshly projected q, k (and v for ``fft_vfixed``), ``[B, n, d]`` with
    ``B = chunks * c`` laid out ``[uncond ; cond ; recon]`` (pnp_utils.py:129-262)."""
    B = q.shape[0]
    c = B // cfg.chunks
    if not cfg.switch_on:
        return q, k, v
    if cfg.chunks == 3:
        f = cfg.fusion
        if f == "replace":  # :133-143
            q[c:2 * c] = q[:c]; k[c:2 * c] = k[:c]
            q[2 * c:] = q[:c]; k[2 * c:] = k[:c]
        elif f == "temporal":  # :145-154
            t1 = temporal_attention(q[:c]); t2 = temporal_attention(k[:c])
            q[c:2 * c] = t1; k[c:2 * c] = t2
            q[2 * c:] = t1; k[2 * c:] = t2
        elif f == "adaIn":  # :155-160
            q[c:2 * c] = adain_fusion_for_attn(q[:c], q[c:2 * c], alpha=0.9)
            k[c:2 * c] = adain_fusion_for_attn(k[:c], k[c:2 * c], alpha=0.9)
            q[2 * c:] = adain_fusion_for_attn(q[:c], q[2 * c:], alpha=0.9)
            k[2 * c:] = adain_fusion_for_attn(k[:c], k[2 * c:], alpha=0.9)
        elif f == "mix":  # :161-166
            q[c:2 * c] = mix_source_and_target(q[:c], q[c:2 * c], alpha=0.5)
            k[c:2 * c] = mix_source_and_target(k[:c], k[c:2 * c], alpha=0.5)
            q[2 * c:] = mix_source_and_target(q[:c], q[2 * c:], alpha=0.5)
            k[2 * c:] = mix_source_and_target(k[:c], k[2 * c:], alpha=0.5)
        elif f in ("fft", "flow_fix", "fft_vfixed"):  # :169-256
            r = 0.8 if f == "fft_vfixed" else cfg.split_ratio_fft
            q[c:2 * c] = combine_fft_high_low(q[:c], q[c:2 * c], split_ratio=r)
            k[c:2 * c] = combine_fft_high_low(k[:c], k[c:2 * c], split_ratio=r)
            q[2 * c:] = combine_fft_high_low(q[:c], q[2 * c:], split_ratio=r)
            k[2 * c:] = combine_fft_high_low(k[:c], k[2 * c:], split_ratio=r)
            if f == "flow_fix" and cfg.flow is not None and _flow_gate(q.shape[1], spatial_hw):
                h, w = spatial_hw if spatial_hw is not None else (64, 64)
                for t in (q, k):  # :206-218; chunk 1 only, already FSAI'd
                    m = t[c:2 * c].reshape(c, h, w, -1).permute(0, 3, 1, 2)
                    m = oflow.align_by_flow(m, cfg.flow, cfg.alpha)
                    t[c:2 * c] = m.permute(0, 2, 3, 1).reshape(c, h * w, -1)
            if f == "fft_vfixed":  # :255-256
                v[c:2 * c] = v[c].repeat(c, 1, 1)
                v[2 * c:] = v[2 * c].repeat(c, 1, 1)
        # any other fusion string: no edit (falls through every elif in the reference)
    elif cfg.chunks == 2:  # :259-262
        q[c:] = q[:c]; k[c:] = k[:c]
    return q, k, v


def _flow_gate(n: int, spatial_hw) -> bool:
    """pnp_utils.py:201 fires only for n == 4096 (64x64).  The build generalises the gate to "the map the
    flow was supplied for" (SURVEY §7): with ``spatial_hw`` given, fire when n == h*w of the level-0 map."""
    if spatial_hw is None:
        return n == 4096
    return n == spatial_hw[0] * spatial_hw[1]


def _ident(t):
    return t


def attention(x, wq, wk, wv, wo, bo, heads: int, context=None, cfg: Optional[HookCfg] = None,
              level0_hw=None, rnd: Callable[[torch.Tensor], torch.Tensor] = _ident):
    """``CrossAttention.forward`` (attention.py:179-221) and its hooked replacement
    (pnp_utils.py:94-287).  ``rnd`` marks the reference's fp16 rounding points under CUDA autocast
    (identity in the default fp32 oracle)."""
    x = rnd(x)  # LayerNorm output is fp32 under autocast; nn.Linear casts it (and the weights) to fp16
    ctx = x if context is None else rnd(context)
    q = rnd(x @ wq.t())
    k = rnd(ctx @ wk.t())
    v = rnd(ctx @ wv.t())
    if cfg is not None:
        q, k, v = apply_fusion(q, k, v, cfg, level0_hw)
        q, k, v = rnd(q), rnd(k), rnd(v)
    B, n, d = q.shape
    dh = d // heads
    scale = dh ** -0.5

    def split(t):
        return t.reshape(B, t.shape[1], heads, dh).permute(0, 2, 1, 3)

    qh, kh, vh = split(q), split(k), split(v)
    if rnd is _ident:
        # fp32 oracle: the same softmax(q k^T * scale) v without materialising [B*h, n, n] (minutes -> seconds on CPU)
        out = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh, scale=scale)
        out = out.permute(0, 2, 1, 3).reshape(B, n, d)
        return out @ wo.t() + bo
    sim = rnd(rnd(qh @ kh.transpose(-1, -2)) * scale)
    attn = rnd(sim.float().softmax(dim=-1))  # autocast: softmax in fp32, result cast for the fp16 bmm
    out = rnd(attn @ vh)
    out = out.permute(0, 2, 1, 3).reshape(B, n, d)
    return rnd(out @ wo.t() + bo)


# --------------------------------------------------------------------------- registry
def register_spa_attn_injection(registry: Dict[str, HookCfg], names_by_group: Dict[str, List[str]],
                                injection_schedule=None, switch_on=True, input_blocks=False, output_blocks=True,
                                middle_block=False, attn_component="attn1", chunks=3, flow=None,
                                block_indices=None, fusion="replace", split_ratio_fft=0.8, alpha=0.8):
    """Same arguments and selection rule as pnp_utils.py:57,289-339.  ``names_by_group`` maps
    ``'input_blocks' | 'middle_block' | 'output_blocks'`` to that group's ``attn1`` module names in
    ``named_modules`` order (pnp_utils.py:33-40)."""
    cfg = HookCfg(switch_on, chunks, fusion, flow, split_ratio_fft, alpha)
    for enabled, group in ((input_blocks, "input_blocks"), (output_blocks, "output_blocks"),
                           (middle_block, "middle_block")):
        if not enabled:
            continue
        names = [n for n in names_by_group[group] if n.endswith(attn_component)]
        for i, name in enumerate(names):
            if block_indices is None or i in block_indices:
                registry[name] = cfg
    return registry
