"""Oracle (test infrastructure only): the ldm ``UNetModel`` forward as plain functions over a state dict.

Restates ``REFace/ldm/modules/diffusionmodules/openaimodel.py:528-907`` (construction order, which fixes the
state-dict key names and the ``attn1`` ordinals) and ``:860-907`` (forward), ``ResBlock._forward``
(``:255-275``), ``Downsample``/``Upsample`` (``:134-160``/``:91-119``), ``SpatialTransformer`` and
``BasicTransformerBlock`` (``REFace/ldm/modules/attention.py:224-289``), ``GEGLU``/``FeedForward``
(``:37-64``), ``GroupNorm32`` and ``timestep_embedding`` (``REFace/ldm/modules/diffusionmodules/util.py:214-216,
151-171``), for the configuration of ``project_ffhq.yaml:33-56`` (``use_spatial_transformer``,
``legacy: false``, ``transformer_depth 1``, no class conditioning).

fp32 everywhere by default.  ``half=torch.float16`` inserts the reference's CUDA-autocast rounding points
(SURVEY §3 precision map) so the fp16 HIP path can be compared at matching precision.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import hooks as ohooks


@dataclass
class UNetSpec:
    in_channels: int = 9
    model_channels: int = 320
    out_channels: int = 4
    num_res_blocks: int = 2
    attention_resolutions: Tuple[int, ...] = (4, 2, 1)
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_heads: int = 8
    context_dim: int = 768


# layer descriptors: (kind, key prefix, params)
@dataclass
class Layer:
    kind: str  # conv | res | st | down | up
    prefix: str
    cin: int = 0
    cout: int = 0
    heads: int = 0


def topology(spec: UNetSpec) -> Dict[str, List[List[Layer]]]:
    """Block structure in construction order (openaimodel.py:668-824)."""
    mc = spec.model_channels
    inp: List[List[Layer]] = [[Layer("conv", "input_blocks.0.0", spec.in_channels, mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, mult in enumerate(spec.channel_mult):
        for _ in range(spec.num_res_blocks):
            i = len(inp)
            layers = [Layer("res", f"input_blocks.{i}.0", ch, mult * mc)]
            ch = mult * mc
            if ds in spec.attention_resolutions:
                layers.append(Layer("st", f"input_blocks.{i}.1", ch, ch, spec.num_heads))
            inp.append(layers)
            chans.append(ch)
        if level != len(spec.channel_mult) - 1:
            i = len(inp)
            inp.append([Layer("down", f"input_blocks.{i}.0", ch, ch)])
            chans.append(ch)
            ds *= 2
    mid = [[Layer("res", "middle_block.0", ch, ch), Layer("st", "middle_block.1", ch, ch, spec.num_heads),
            Layer("res", "middle_block.2", ch, ch)]]
    out: List[List[Layer]] = []
    for level, mult in list(enumerate(spec.channel_mult))[::-1]:
        for i in range(spec.num_res_blocks + 1):
            ich = chans.pop()
            j = len(out)
            layers = [Layer("res", f"output_blocks.{j}.0", ch + ich, mc * mult)]
            ch = mc * mult
            if ds in spec.attention_resolutions:
                layers.append(Layer("st", f"output_blocks.{j}.{len(layers)}", ch, ch, spec.num_heads))
            if level and i == spec.num_res_blocks:
                layers.append(Layer("up", f"output_blocks.{j}.{len(layers)}", ch, ch))
                ds //= 2
            out.append(layers)
    return {"input_blocks": inp, "middle_block": mid, "output_blocks": out}


def attn1_names(spec: UNetSpec) -> Dict[str, List[str]]:
    """``attn1`` module names per group, in ``named_modules`` order (pnp_utils.py:33-40,290,307,324), with
    the group prefix kept so names are unique keys."""
    topo = topology(spec)
    return {g: [f"{l.prefix}.transformer_blocks.0.attn1" for blk in topo[g] for l in blk if l.kind == "st"]
            for g in topo}


def param_shapes(spec: UNetSpec) -> Dict[str, Tuple[int, ...]]:
    """Every state-dict key of the reference UNetModel for ``spec`` with its shape."""
    s: Dict[str, Tuple[int, ...]] = {}
    mc = spec.model_channels
    te = 4 * mc
    s["time_embed.0.weight"] = (te, mc); s["time_embed.0.bias"] = (te,)
    s["time_embed.2.weight"] = (te, te); s["time_embed.2.bias"] = (te,)

    def conv(p, ci, co, k):
        s[p + ".weight"] = (co, ci, k, k); s[p + ".bias"] = (co,)

    def norm(p, c):
        s[p + ".weight"] = (c,); s[p + ".bias"] = (c,)

    def lin(p, ci, co, bias=True):
        s[p + ".weight"] = (co, ci)
        if bias:
            s[p + ".bias"] = (co,)

    topo = topology(spec)
    for g in ("input_blocks", "middle_block", "output_blocks"):
        for blk in topo[g]:
            for l in blk:
                p = l.prefix
                if l.kind == "conv":
                    conv(p, l.cin, l.cout, 3)
                elif l.kind == "down":
                    conv(p + ".op", l.cin, l.cout, 3)
                elif l.kind == "up":
                    conv(p + ".conv", l.cin, l.cout, 3)
                elif l.kind == "res":
                    norm(p + ".in_layers.0", l.cin); conv(p + ".in_layers.2", l.cin, l.cout, 3)
                    lin(p + ".emb_layers.1", te, l.cout)
                    norm(p + ".out_layers.0", l.cout); conv(p + ".out_layers.3", l.cout, l.cout, 3)
                    if l.cin != l.cout:
                        conv(p + ".skip_connection", l.cin, l.cout, 1)
                elif l.kind == "st":
                    c = l.cin
                    norm(p + ".norm", c); conv(p + ".proj_in", c, c, 1)
                    t = p + ".transformer_blocks.0"
                    for a, cd in (("attn1", c), ("attn2", spec.context_dim)):
                        lin(f"{t}.{a}.to_q", c, c, False); lin(f"{t}.{a}.to_k", cd, c, False)
                        lin(f"{t}.{a}.to_v", cd, c, False); lin(f"{t}.{a}.to_out.0", c, c)
                    lin(t + ".ff.net.0.proj", c, 8 * c); lin(t + ".ff.net.2", 4 * c, c)
                    for nn_ in ("norm1", "norm2", "norm3"):
                        norm(f"{t}.{nn_}", c)
                    conv(p + ".proj_out", c, c, 1)
    norm("out.0", mc)
    conv("out.2", mc, spec.out_channels, 3)
    return s


def timestep_embedding(t: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    """util.py:151-171."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class _Ctx:
    def __init__(self, sd, half):
        self.half = half
        if half is None:
            self.rnd = ohooks._ident
            self.sd = {k: v.float() for k, v in sd.items()}
        else:
            self.rnd = lambda t: t.to(half).float()
            self.sd = {k: v.float() for k, v in sd.items()}
            self.sdh = {k: v.to(half).float() for k, v in sd.items()}

    def w(self, key, lowp=True):
        """weight as seen by an autocast fp16 op (lowp) or an fp32-list op."""
        if self.half is not None and lowp:
            return self.sdh[key]
        return self.sd[key]


def _gn32(c: _Ctx, x, p, eps=1e-5):
    # GroupNorm32 (util.py:214-216): fp32 math, result cast back to the input dtype (fp16 under autocast)
    return c.rnd(F.group_norm(x, 32, c.w(p + ".weight", False), c.w(p + ".bias", False), eps))


def _silu(c: _Ctx, x):
    return c.rnd(F.silu(x))


def _conv(c: _Ctx, x, p, stride=1, pad=1):
    return c.rnd(F.conv2d(c.rnd(x), c.w(p + ".weight"), c.w(p + ".bias"), stride=stride, padding=pad))


def _linear(c: _Ctx, x, p, bias=True):
    return c.rnd(F.linear(c.rnd(x), c.w(p + ".weight"), c.w(p + ".bias") if bias else None))


def _res(c: _Ctx, l: Layer, x, emb):
    p = l.prefix
    h = _conv(c, _silu(c, _gn32(c, x, p + ".in_layers.0")), p + ".in_layers.2")
    e = _linear(c, _silu(c, emb), p + ".emb_layers.1")
    h = c.rnd(h + e[:, :, None, None])
    h = _conv(c, _silu(c, _gn32(c, h, p + ".out_layers.0")), p + ".out_layers.3")
    skip = x if l.cin == l.cout else _conv(c, x, p + ".skip_connection", pad=0)
    return c.rnd(skip + h)


def _st(c: _Ctx, l: Layer, x, context, registry, level0_hw):
    p = l.prefix
    B, C, H, W = x.shape
    x_in = x
    # nn.GroupNorm(eps=1e-6) is on autocast's fp32 list: fp32 output, cast to fp16 by the 1x1 conv
    h = F.group_norm(x, 32, c.w(p + ".norm.weight", False), c.w(p + ".norm.bias", False), 1e-6)
    h = _conv(c, h, p + ".proj_in", pad=0)
    t = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    tp = p + ".transformer_blocks.0"

    def ln(v, q):
        return F.layer_norm(v, (C,), c.w(f"{tp}.{q}.weight", False), c.w(f"{tp}.{q}.bias", False), 1e-5)

    def attn(v, a, ctx, cfg):
        return ohooks.attention(v, c.w(f"{tp}.{a}.to_q.weight"), c.w(f"{tp}.{a}.to_k.weight"),
                                c.w(f"{tp}.{a}.to_v.weight"), c.w(f"{tp}.{a}.to_out.0.weight"),
                                c.w(f"{tp}.{a}.to_out.0.bias"), l.heads, ctx, cfg, level0_hw, c.rnd)

    cfg = registry.get(f"{tp}.attn1") if registry else None
    t = c.rnd(attn(ln(t, "norm1"), "attn1", None, cfg) + t)
    t = c.rnd(attn(ln(t, "norm2"), "attn2", context, None) + t)
    g = _linear(c, ln(t, "norm3"), tp + ".ff.net.0.proj")
    a, gate = g.chunk(2, dim=-1)
    g = c.rnd(a * c.rnd(F.gelu(gate)))
    t = c.rnd(_linear(c, g, tp + ".ff.net.2") + t)
    h = t.reshape(B, H, W, C).permute(0, 3, 1, 2)
    h = _conv(c, h, p + ".proj_out", pad=0)
    return c.rnd(h + x_in)


def unet_forward(sd: Dict[str, torch.Tensor], spec: UNetSpec, x: torch.Tensor, timesteps: torch.Tensor,
                 context: torch.Tensor, registry: Optional[Dict[str, ohooks.HookCfg]] = None,
                 half: Optional[torch.dtype] = None, return_trace: bool = False):
    """openaimodel.py:860-907.  ``x`` ``[N, in_channels, H, W]`` fp32, ``timesteps`` ``[N]``, ``context``
    ``[N, 1, context_dim]``; returns eps ``[N, out_channels, H, W]`` fp32."""
    c = _Ctx(sd, half)
    topo = topology(spec)
    level0_hw = (x.shape[2], x.shape[3])
    temb = timestep_embedding(timesteps, spec.model_channels)
    emb = _linear(c, _silu_plain(c, _linear(c, temb, "time_embed.0")), "time_embed.2")
    trace = {}

    def run(blk, h):
        for l in blk:
            if l.kind == "conv":
                h = _conv(c, h, l.prefix)
            elif l.kind == "res":
                h = _res(c, l, h, emb)
            elif l.kind == "st":
                h = _st(c, l, h, context, registry, level0_hw)
            elif l.kind == "down":
                h = _conv(c, h, l.prefix + ".op", stride=2)
            elif l.kind == "up":
                h = _conv(c, F.interpolate(h, scale_factor=2, mode="nearest"), l.prefix + ".conv")
            if return_trace:
                trace[l.prefix] = h
        return h

    h = x.float()
    hs = []
    for blk in topo["input_blocks"]:
        h = run(blk, h)
        hs.append(h)
    h = run(topo["middle_block"][0], h)
    for blk in topo["output_blocks"]:
        h = torch.cat([h, hs.pop()], dim=1)
        h = run(blk, h)
    h = _conv(c, _silu(c, _gn32(c, h, "out.0")), "out.2")
    h = h.float()
    return (h, trace) if return_trace else h


def _silu_plain(c: _Ctx, x):
    return c.rnd(F.silu(x))
