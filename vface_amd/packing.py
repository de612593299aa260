"""Weight packing for the HIP kernels (host side, done once per module / hook configuration).

Everything here is a re-layout or an exact linear refactoring of the reference's fp32 parameters,
computed in fp64 and handed to the device in the 16-bit compute type:

* ``pack_conv3x3``   nn.Conv2d weight [Cout, Cin, 3, 3] -> [Cout, (ky, kx, Cin_pad)] for the implicit GEMM.
* ``pack_geglu``     GEGLU.proj rows (attention.py:40-44: first half value, second half gate) interleaved in
                     16-row blocks so the GEMM epilogue finds value and gate of a channel in the same lane.
* ``pack_qkv``       to_q | to_k | to_v stacked into one [3d, d] matrix (one GEMM, attention.py:161,171-172).
* ``fold_fsai``      frequency-spectrum attention interpolation folded into the projection:
                     ``combine_fft_high_low`` (face_swap_utils.py:425-464) is linear, so for chunk c >= 1
                     q_new = q_c A_lo + q_0 A_hi = x_c (Wq^T A_lo) + x_0 (Wq^T A_hi).
* ``fold_mix``       the same mechanism for ``mix_source_and_target`` (face_swap_utils.py:189-199).
"""
from __future__ import annotations

from typing import Optional

import torch


def pack_conv3x3(w: torch.Tensor, cin_pad: int | None = None) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> [Cout, K = 9*Cin_pad], the K order the implicit-GEMM kernel walks:
    * Cin % 64 == 0: (64-channel chunk, tap, channel-in-chunk) -- the 9 taps of a chunk are adjacent K tiles, so a
      pixel's 128-byte channel chunk is re-read 9 times within 9 tiles (L2 hits) instead of once per Cin sweep;
    * otherwise (the 9 -> 320 input conv): (tap, channel), channels zero-padded to a multiple of 8."""
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    cp = cin_pad if cin_pad is not None else (cin + 7) // 8 * 8
    taps = torch.zeros(cout, 9, cp, dtype=w.dtype)
    taps[..., :cin] = w.permute(0, 2, 3, 1).reshape(cout, 9, cin)
    if cp % 64 == 0:
        taps = taps.reshape(cout, 9, cp // 64, 64).permute(0, 2, 1, 3)
    return taps.reshape(cout, 9 * cp).contiguous()


def pack_conv_window(w: torch.Tensor, cin_pad: int | None = None) -> torch.Tensor:
    """[Cout, Cin, KH, KW] -> [Cout, KH*KW*Cin_pad] in the K order of ``pack_conv3x3`` for any window up to 3x3."""
    cout, cin, kh, kw = w.shape
    cp = cin_pad if cin_pad is not None else (cin + 7) // 8 * 8
    taps = torch.zeros(cout, kh * kw, cp, dtype=w.dtype)
    taps[..., :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    if cp % 64 == 0:
        taps = taps.reshape(cout, kh * kw, cp // 64, 64).permute(0, 2, 1, 3)
    return taps.reshape(cout, kh * kw * cp).contiguous()


def pack_upsample_phases(w: torch.Tensor) -> torch.Tensor:
    """conv3x3(nearest_upsample2x(x)) by output parity (openaimodel.py:108-118): pixel (2i+py, 2j+px) sees source rows
    {i-1, i} (py = 0) or {i, i+1} (py = 1) -- likewise columns -- so the 3x3 kernel collapses, exactly, into one 2x2 kernel
    per phase whose taps are sums of the original ones (summed here in fp64).  -> [4 (2*py+px), Cout, 4*Cin_pad] packed."""
    w64 = w.double()
    rows = {0: ([0], [1, 2]), 1: ([0, 1], [2])}     # parity -> (taps folded onto the first source row, onto the second)
    out = []
    for py in (0, 1):
        for px in (0, 1):
            k = torch.zeros(w.shape[0], w.shape[1], 2, 2, dtype=torch.float64)
            for ty, kys in enumerate(rows[py]):
                for tx, kxs in enumerate(rows[px]):
                    k[:, :, ty, tx] = w64[:, :, kys][:, :, :, kxs].sum(dim=(2, 3))
            out.append(pack_conv_window(k.float()))
    return torch.stack(out).contiguous()


def pack_conv1x1(w: torch.Tensor) -> torch.Tensor:
    return w.reshape(w.shape[0], w.shape[1]).contiguous()


def pack_geglu(w: torch.Tensor, b: torch.Tensor):
    two_inner, d = w.shape
    inner = two_inner // 2
    assert inner % 16 == 0, "GEGLU inner width must be a multiple of 16"
    val, gate = w[:inner].reshape(inner // 16, 16, d), w[inner:].reshape(inner // 16, 16, d)
    wp = torch.stack([val, gate], 1).reshape(two_inner, d).contiguous()
    bv, bg = b[:inner].reshape(inner // 16, 16), b[inner:].reshape(inner // 16, 16)
    bp = torch.stack([bv, bg], 1).reshape(two_inner).contiguous()
    return wp, bp


def ffn_w2_perm(inner: int) -> torch.Tensor:
    """Column order of ``ff.net[2].weight`` for the fused FeedForward kernel (csrc/ffn.hip): inside every aligned block of 16
    hidden units, position ``8 h + j`` holds original unit ``8 (j >> 2) + 4 h + (j & 3)`` -- the k order in which the 16-bit
    roundings of a 32 x 32 fp32 accumulator tile's registers 0..7 (rows ``(reg & 3) + 8 (reg >> 2) + 4 h`` of lane half h) form
    ONE k16 MFMA B operand without any lane movement (guide 3, "an accumulator tile as the next MFMA's operand")."""
    assert inner % 16 == 0
    j = torch.arange(8)
    h = torch.arange(2)
    within = (8 * (j[None, :] >> 2) + 4 * h[:, None] + (j[None, :] & 3)).reshape(16)       # [8 h + j]
    return (torch.arange(0, inner, 16)[:, None] + within[None, :]).reshape(inner)


def pack_ffn_w2(w2: torch.Tensor) -> torch.Tensor:
    """``ff.net[2].weight`` [C, 4C] with its columns in ``ffn_w2_perm`` order."""
    return w2[:, ffn_w2_perm(w2.shape[1])].contiguous()


def pack_st_front(w_in: torch.Tensor, w_proj: torch.Tensor) -> torch.Tensor:
    """Weights of the fused SpatialTransformer front (csrc/stfront.hip): ``proj_in`` rows ``[C, C]`` (its k index is the channel,
    as the GroupNorm'd activations are read), then the attn1 projection rows ``[NQ, C]`` with their k columns in ``ffn_w2_perm``
    order -- the order in which the LayerNorm'd accumulator tile of ``proj_in`` becomes that GEMM's B operand."""
    c = w_in.shape[0]
    assert w_in.shape == (c, c) and w_proj.shape[1] == c
    return torch.cat([w_in, w_proj[:, ffn_w2_perm(c)]], 0).contiguous()


def pack_attn_out_ffn(w_o: torch.Tensor, w1_geglu: torch.Tensor, w_proj_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weight stream of the fused block TAIL (csrc/ffn.hip, PRE form): ``attn1.to_out[0]`` rows ``[C, C]`` (k = channel of the
    attention output), then the GEGLU-interleaved ``ff.net[0]`` rows ``[8C, C]`` (``pack_geglu``) with their k columns in
    ``ffn_w2_perm`` order -- the order in which the LayerNorm'd accumulator tile of the out-projection is that GEMM's B operand --
    and, for the POST form, the SpatialTransformer's ``proj_out`` rows ``[C, C]`` with the same column order (their operand is the
    FeedForward's rounded accumulator tile)."""
    c = w_o.shape[0]
    assert w_o.shape == (c, c) and w1_geglu.shape == (8 * c, c)
    perm = ffn_w2_perm(c)
    parts = [w_o, w1_geglu[:, perm]]
    if w_proj_out is not None:
        assert w_proj_out.shape == (c, c)
        parts.append(w_proj_out[:, perm])
    return torch.cat(parts, 0).contiguous()


def pack_qkv(wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor) -> torch.Tensor:
    return torch.cat([wq, wk, wv], 0).contiguous()


def _band_filter_rows(m: torch.Tensor, lo: int, hi: int) -> torch.Tensor:
    """Re(ifft(mask[lo:hi] * fft(row))) for every row of ``m`` (fp64)."""
    f = torch.fft.fft(m.double(), dim=-1)
    keep = torch.zeros_like(f)
    keep[..., lo:hi] = f[..., lo:hi]
    return torch.fft.ifft(keep, dim=-1).real


def fold_fsai(wq: torch.Tensor, wk: torch.Tensor, split_ratio: float) -> torch.Tensor:
    """[2d, 2d] fp32: rows = fused q outputs then fused k outputs; columns = [own x | chunk-0 x].
    Bins [0, int(d*ratio)) come from the own branch, the rest from chunk 0 (face_swap_utils.py:447-457)."""
    d = wq.shape[0]
    s = int(d * split_ratio)
    blocks = []
    for w in (wq, wk):
        wt = w.double().t()  # [d_in, d_out]: row k is the response of every output channel to input k
        own = _band_filter_rows(wt, 0, s).t()      # (Wq^T A_lo)^T  -> [d_out, d_in]
        struct = _band_filter_rows(wt, s, d).t()   # (Wq^T A_hi)^T
        blocks.append(torch.cat([own, struct], 1))
    return torch.cat(blocks, 0).float().contiguous()


def fold_mix(wq: torch.Tensor, wk: torch.Tensor, alpha: float) -> torch.Tensor:
    """mix_source_and_target(target=chunk 0, source=own, alpha): (1-alpha)*own + alpha*chunk0."""
    blocks = [torch.cat([(1.0 - alpha) * w.double(), alpha * w.double()], 1) for w in (wq, wk)]
    return torch.cat(blocks, 0).float().contiguous()
