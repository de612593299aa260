"""Drop-in surface of ``REFace/ldm/modules/attention.py`` for the VFace hot path.

Same class names, constructor signatures, attributes (``heads``, ``dim_head``, ``scale``, ``to_q`` ...) and
state-dict keys as the reference (attention.py:37-64, 152-289), so ``last.ckpt`` loads unchanged and
``register_spa_attn_injection`` finds the same ``attn1`` modules in the same order.  The modules own
parameters; every ``forward`` runs hand-written gfx950 kernels (``vface_amd.engine`` / ``vface_amd.module_exec`` /
``vface_amd.hip``); calling them with CPU tensors raises.
"""
from __future__ import annotations

import torch
from torch import nn

from ... import hip


def exists(val):
    return val is not None


def default(val, d):
    return val if val is not None else (d() if callable(d) else d)


def Normalize(in_channels):
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class GEGLU(nn.Module):
    """attention.py:37-45: ``proj`` maps dim_in -> 2*dim_out; forward = value * gelu(gate)."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        from ...module_exec import geglu_forward
        return geglu_forward(self, x)


class FeedForward(nn.Module):
    """attention.py:47-64 with ``glu=True`` (the only form the UNet builds)."""

    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        inner = int(dim * mult)
        dim_out = default(dim_out, dim)
        if not glu:
            raise NotImplementedError("the VFace UNet uses gated feed-forward only (attention.py:228)")
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out))

    def forward(self, x):
        from ...module_exec import feedforward_forward
        return feedforward_forward(self, x)


class CrossAttention(nn.Module):
    """attention.py:152-221.  ``forward`` handles what the path issues: self-attention (``context=None``,
    optionally hooked) and attention over a single context token; ``mask`` is not used on this path."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0., sep_head_att=False):
        super().__init__()
        inner = dim_head * heads
        context_dim = default(context_dim, query_dim)
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.head_splits = [6, 2]
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))
        self._pack = None

    # -- standalone execution (module-level drop-in; the UNet engine fuses these calls instead)
    def _packed(self, dtype):
        from ... import packing
        ver = tuple(p._version for p in self.parameters()) + (dtype,)
        if self._pack is None or self._pack["ver"] != ver:
            dev = self.to_q.weight.device
            w16 = lambda t: t.detach().to(device=dev, dtype=dtype).contiguous()
            self._pack = {"ver": ver, "wlin": {},
                          "wqkv": w16(packing.pack_qkv(self.to_q.weight, self.to_k.weight, self.to_v.weight))
                          if self.to_k.weight.shape == self.to_q.weight.shape else None,
                          "wq": w16(self.to_q.weight), "wk": w16(self.to_k.weight), "wv": w16(self.to_v.weight),
                          "wo": w16(self.to_out[0].weight), "bo": self.to_out[0].bias.detach().float().contiguous()}
        return self._pack

    def forward(self, x, context=None, mask=None, _cfg=None):
        if mask is not None:
            raise NotImplementedError("attention masks are not used on the VFace path (pnp_utils.py:276-280 is dead)")
        if not x.is_cuda:
            raise hip.VFaceHipError("CrossAttention.forward needs CUDA tensors: no CPU fallback on the VFace path")
        from ...engine import attn_module_forward
        return attn_module_forward(self, x, context, _cfg)


class BasicTransformerBlock(nn.Module):
    """attention.py:224-243.  Inside the UNet the block runs as part of ``UNetEngine._st``; called directly,
    ``forward(x [B, n, d], context [B, 1, ctx])`` runs the same kernel sequence (``vface_amd.module_exec``)."""

    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True,
                 sep_head_att=False):
        super().__init__()
        self.attn1 = CrossAttention(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head,
                                    dropout=dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.norm3 = nn.LayerNorm(dim)
        self.checkpoint = checkpoint

    def forward(self, x, context=None):
        from ...module_exec import transformer_block_forward
        return transformer_block_forward(self, x, context)


class SpatialTransformer(nn.Module):
    """attention.py:246-289.  Inside the UNet the layer runs through ``UNetEngine._st``; called directly,
    ``forward(x [N, C, H, W], context [N, 1, ctx])`` runs the same kernel sequence (``vface_amd.module_exec``)."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, sep_head_att=False,
                 head_splits=None):
        super().__init__()
        if depth != 1:
            raise NotImplementedError("transformer_depth != 1 is not part of the VFace configuration")
        self.in_channels = in_channels
        inner = n_heads * d_head
        self.norm = Normalize(in_channels)
        self.proj_in = nn.Conv2d(in_channels, inner, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, context=None):
        from ...module_exec import spatial_transformer_forward
        return spatial_transformer_forward(self, x, context)
