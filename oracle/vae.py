"""Oracle (test infrastructure only): the first-stage KL-VAE.

Restates ``REFace/ldm/modules/diffusionmodules/model.py`` -- ``Encoder`` (:368-459), ``Decoder`` (:462-568),
``ResnetBlock`` (:82-141, ``temb`` is None on this path), ``AttnBlock`` (:150-202), ``Downsample`` (:60-79),
``Upsample`` (:42-57), ``Normalize`` (:38-39, GroupNorm 32, eps 1e-6), ``nonlinearity`` (:33-35) -- and
``AutoencoderKL.encode / decode`` (``ldm/models/autoencoder.py:323-333``) with
``DiagonalGaussianDistribution`` (``ldm/modules/distributions/distributions.py:24-37,61-62``), as plain functions over a
state dict with the reference's key names (``encoder.down.0.block.0.norm1.weight`` ...).  fp32, CPU.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Sequence, Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class VAESpec:
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    in_channels: int = 3
    out_ch: int = 3
    z_channels: int = 4
    embed_dim: int = 4
    attn_resolutions: Tuple[int, ...] = ()   # project_ffhq.yaml:75 -- only the two mid AttnBlocks exist
    resolution: int = 256


FFHQ_VAE = VAESpec()


def _gn_swish(sd, pre, x):
    h = F.group_norm(x, 32, sd[pre + ".weight"], sd[pre + ".bias"], eps=1e-6)   # model.py:38-39
    return h * torch.sigmoid(h)                                                  # :33-35


def _conv(sd, pre, x, stride=1, padding=1):
    return F.conv2d(x, sd[pre + ".weight"], sd[pre + ".bias"], stride=stride, padding=padding)


def resnet_block(sd, pre, x):
    """model.py:118-141 with temb None and dropout 0."""
    h = _conv(sd, pre + ".conv1", _gn_swish(sd, pre + ".norm1", x))
    h = _conv(sd, pre + ".conv2", _gn_swish(sd, pre + ".norm2", h))
    if pre + ".nin_shortcut.weight" in sd:
        x = _conv(sd, pre + ".nin_shortcut", x, padding=0)
    elif pre + ".conv_shortcut.weight" in sd:
        x = _conv(sd, pre + ".conv_shortcut", x)
    return x + h


def attn_block(sd, pre, x):
    """model.py:176-202: one head over all c channels, scale c**-0.5."""
    h = F.group_norm(x, 32, sd[pre + ".norm.weight"], sd[pre + ".norm.bias"], eps=1e-6)
    q, k, v = (_conv(sd, f"{pre}.{n}", h, padding=0) for n in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)
    k = k.reshape(b, c, hh * ww)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)
    v = v.reshape(b, c, hh * ww)
    o = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, pre + ".proj_out", o, padding=0)


def encoder(sd, spec: VAESpec, x, pre="encoder"):
    """model.py:434-459."""
    h = _conv(sd, pre + ".conv_in", x)
    nres = len(spec.ch_mult)
    for lvl in range(nres):
        for blk in range(spec.num_res_blocks):
            h = resnet_block(sd, f"{pre}.down.{lvl}.block.{blk}", h)
            if f"{pre}.down.{lvl}.attn.{blk}.q.weight" in sd:
                h = attn_block(sd, f"{pre}.down.{lvl}.attn.{blk}", h)
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1))                                     # :72-77 asymmetric padding, then stride 2
            h = _conv(sd, f"{pre}.down.{lvl}.downsample.conv", h, stride=2, padding=0)
    h = resnet_block(sd, pre + ".mid.block_1", h)
    h = attn_block(sd, pre + ".mid.attn_1", h)
    h = resnet_block(sd, pre + ".mid.block_2", h)
    return _conv(sd, pre + ".conv_out", _gn_swish(sd, pre + ".norm_out", h))


def decoder(sd, spec: VAESpec, z, pre="decoder"):
    """model.py:534-568."""
    h = _conv(sd, pre + ".conv_in", z)
    h = resnet_block(sd, pre + ".mid.block_1", h)
    h = attn_block(sd, pre + ".mid.attn_1", h)
    h = resnet_block(sd, pre + ".mid.block_2", h)
    nres = len(spec.ch_mult)
    for lvl in reversed(range(nres)):
        for blk in range(spec.num_res_blocks + 1):
            h = resnet_block(sd, f"{pre}.up.{lvl}.block.{blk}", h)
            if f"{pre}.up.{lvl}.attn.{blk}.q.weight" in sd:
                h = attn_block(sd, f"{pre}.up.{lvl}.attn.{blk}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")         # :55-58
            h = _conv(sd, f"{pre}.up.{lvl}.upsample.conv", h)
    return _conv(sd, pre + ".conv_out", _gn_swish(sd, pre + ".norm_out", h))


def encode_moments(sd, spec: VAESpec, x):
    """AutoencoderKL.encode up to the Gaussian's parameters (autoencoder.py:323-327): [B, 2*embed_dim, h, w]."""
    return _conv(sd, "quant_conv", encoder(sd, spec, x), padding=0)


def sample(moments, noise=None, scale_factor=1.0):
    """distributions.py:24-37 (``sample``) / :61-62 (``mode`` when noise is None), times ddpm.py's scale_factor."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    if noise is None:
        return mean * scale_factor
    logvar = torch.clamp(logvar, -30.0, 20.0)
    return (mean + torch.exp(0.5 * logvar) * noise) * scale_factor


def decode(sd, spec: VAESpec, z):
    """AutoencoderKL.decode (autoencoder.py:329-333)."""
    return decoder(sd, spec, _conv(sd, "post_quant_conv", z, padding=0))


def param_shapes(spec: VAESpec) -> Dict[str, Sequence[int]]:
    """State-dict keys and shapes of AutoencoderKL's encoder / decoder / quant convs for ``spec``."""
    out: Dict[str, Sequence[int]] = {}

    def conv(pre, cin, cout, k):
        out[pre + ".weight"] = (cout, cin, k, k); out[pre + ".bias"] = (cout,)

    def norm(pre, c):
        out[pre + ".weight"] = (c,); out[pre + ".bias"] = (c,)

    def res(pre, cin, cout):
        norm(pre + ".norm1", cin); conv(pre + ".conv1", cin, cout, 3)
        norm(pre + ".norm2", cout); conv(pre + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(pre + ".nin_shortcut", cin, cout, 1)

    def attn(pre, c):
        norm(pre + ".norm", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(f"{pre}.{n}", c, c, 1)

    ch, nres = spec.ch, len(spec.ch_mult)
    conv("encoder.conv_in", spec.in_channels, ch, 3)
    in_mult = (1,) + tuple(spec.ch_mult)
    bi = ch
    for lvl in range(nres):
        bi, bo = ch * in_mult[lvl], ch * spec.ch_mult[lvl]
        for blk in range(spec.num_res_blocks):
            res(f"encoder.down.{lvl}.block.{blk}", bi, bo)
            bi = bo
        if lvl != nres - 1:
            conv(f"encoder.down.{lvl}.downsample.conv", bi, bi, 3)
    res("encoder.mid.block_1", bi, bi); attn("encoder.mid.attn_1", bi); res("encoder.mid.block_2", bi, bi)
    norm("encoder.norm_out", bi); conv("encoder.conv_out", bi, 2 * spec.z_channels, 3)
    bi = ch * spec.ch_mult[-1]
    conv("decoder.conv_in", spec.z_channels, bi, 3)
    res("decoder.mid.block_1", bi, bi); attn("decoder.mid.attn_1", bi); res("decoder.mid.block_2", bi, bi)
    for lvl in reversed(range(nres)):
        bo = ch * spec.ch_mult[lvl]
        for blk in range(spec.num_res_blocks + 1):
            res(f"decoder.up.{lvl}.block.{blk}", bi, bo)
            bi = bo
        if lvl != 0:
            conv(f"decoder.up.{lvl}.upsample.conv", bi, bi, 3)
    norm("decoder.norm_out", bi); conv("decoder.conv_out", bi, spec.out_ch, 3)
    conv("quant_conv", 2 * spec.z_channels, 2 * spec.embed_dim, 1)
    conv("post_quant_conv", spec.embed_dim, spec.z_channels, 1)
    return out
