"""Stand-alone ``forward`` of the container modules of the drop-in surface (SURVEY 8b "module surface"):
``ResBlock`` / ``Downsample`` / ``Upsample`` / ``TimestepEmbedSequential`` (openaimodel.py:74-160, 255-275),
``SpatialTransformer`` / ``BasicTransformerBlock`` / ``FeedForward`` / ``GEGLU`` (attention.py:37-64, 239-243, 278-289).

Inside ``UNetModel.forward`` these layers are executed by ``UNetEngine`` as one fused kernel sequence (stacked
embedding projections, in-place concatenation, ...); a caller that invokes a sub-module directly gets the same
per-layer kernel sequence through the same engine methods, on tensors in the reference's layouts: ``[N, C, H, W]``
images, ``[B, n, d]`` token matrices, ``[N, emb]`` embeddings.  Results come back in the input's floating-point type.
There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import hip
from .engine import Act, UNetEngine


def _engine_for(module, x: torch.Tensor) -> UNetEngine:
    """One parameter-version-checked engine + packed weights per module instance."""
    if not x.is_cuda:
        raise hip.VFaceHipError(f"{type(module).__name__}.forward needs CUDA tensors: no CPU fallback on the VFace path")
    dt = x.dtype if x.dtype in (torch.float16, torch.bfloat16) else torch.float16
    ver = tuple(p._version for p in module.parameters()) + (dt, str(x.device))
    cache = module.__dict__.get("_vface_exec")
    if cache is None or cache["ver"] != ver:
        cache = {"ver": ver, "eng": UNetEngine(None, dt, device=x.device), "packed": {}}
        module.__dict__["_vface_exec"] = cache
    return cache["eng"], cache["packed"]


def _sd(module):
    return {k: v.detach() for k, v in module.state_dict().items()}


def _to_act(eng: UNetEngine, x: torch.Tensor) -> Act:
    """NCHW -> NHWC 16-bit (+ the fp32 carrier when the input is fp32 and the residual stream is on)."""
    N, C, H, W = x.shape
    cpad = (C + 7) // 8 * 8
    t = eng._new(N * H * W, cpad)
    hip.nchw_to_nhwc(x.float().contiguous(), t, N=N, C_=C, hw=H * W, cpad=cpad)
    t32 = None
    if eng.stream32 and x.dtype == torch.float32 and C % 8 == 0:
        t32 = x.permute(0, 2, 3, 1).reshape(N * H * W, C).contiguous()
    return Act(t, N, H, W, None, t32)


def _from_act(a: Act, like: torch.Tensor) -> torch.Tensor:
    src = a.t32 if a.t32 is not None else a.t
    C = src.shape[1]
    return src.reshape(a.N, a.H, a.W, C).permute(0, 3, 1, 2).to(like.dtype).contiguous()


def _emb_rowbias(eng: UNetEngine, sd, emb: torch.Tensor) -> torch.Tensor:
    """emb_layers = SiLU -> Linear (openaimodel.py:218-224): fp32 [N, cout] row bias for the first conv's epilogue."""
    N = emb.shape[0]
    e16 = eng._new(N, emb.shape[1])
    hip.silu(emb.float().contiguous(), e16)
    w = eng.pack_lin(sd, "emb_layers.1")   # (small; re-packed per call)
    out = eng._new(N, w["w"].shape[0], torch.float32)
    eng._gemm(e16, w, out, flags=hip.EPI_OUT_F32)
    return out


def resblock_forward(mod, x: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    eng, P = _engine_for(mod, x)
    sd = _sd(mod)
    if "res" not in P:
        P["res"] = eng.pack_res({("." + k): v for k, v in sd.items()}, "")
    p = dict(P["res"])
    rb = _emb_rowbias(eng, sd, emb)
    p["emb_slice"] = (0, rb.shape[1])
    y = eng._res(_to_act(eng, x), p, rb, None)
    return _from_act(y, x)


def conv_forward(mod, x: torch.Tensor, which: str) -> torch.Tensor:
    """Downsample.op (stride 2) / Upsample.conv (after nearest x2)."""
    eng, P = _engine_for(mod, x)
    if "conv" not in P:
        sd = {("." + k): v for k, v in _sd(mod).items()}
        P["conv"] = eng.pack_up(sd, ".conv") if which == "up" else eng.pack_conv3(sd, ".op")
    a = _to_act(eng, x)
    y = eng._conv(a, P["conv"], None, stride=2 if which == "down" else 1, upsample=which == "up")
    return _from_act(y, x)


def _a2vec(eng: UNetEngine, sd, t: str, context: Optional[torch.Tensor], N: int, c: int, P: dict) -> torch.Tensor:
    """The single-token cross-attention's contribution to_out(to_v(ctx)) as an fp32 [N, c] row bias."""
    if context is None:
        raise ValueError("context (cross-attention conditioning) is required: the reference's attn2 with context=None is "
                         "self-attention, which is not part of the VFace configuration")
    ctx = context.reshape(context.shape[0], -1)
    if context.dim() != 3 or context.shape[1] != 1:
        raise hip.VFaceHipError(f"context must be [N, 1, context_dim] (single token, SURVEY F11); got {tuple(context.shape)}")
    c16 = eng._new(N, ctx.shape[1])
    hip.cast_f32(ctx.float().contiguous(), c16)
    v = eng._new(N, c)
    if "a2" not in P:
        P["a2"] = (eng.pack_lin(sd, t + ".attn2.to_v", bias=False), eng.pack_lin(sd, t + ".attn2.to_out.0"))
    eng._gemm(c16, P["a2"][0], v)
    out = eng._new(N, c, torch.float32)
    eng._gemm(v, P["a2"][1], out, flags=hip.EPI_OUT_F32)
    return out


def spatial_transformer_forward(mod, x: torch.Tensor, context: Optional[torch.Tensor]) -> torch.Tensor:
    eng, P = _engine_for(mod, x)
    sd = {("." + k): v for k, v in _sd(mod).items()}
    if "st" not in P:
        P["st"] = eng.pack_st(sd, "")
    p = dict(P["st"])
    a = _to_act(eng, x)
    a2 = _a2vec(eng, sd, ".transformer_blocks.0", context, a.N, p["c"], P)
    p["a2_slice"] = (0, p["c"])
    y = eng._st(a, p, mod, a2, None)
    return _from_act(y, x)


def transformer_block_forward(mod, x: torch.Tensor, context: Optional[torch.Tensor]) -> torch.Tensor:
    """``x`` [B, n, d] tokens."""
    eng, P = _engine_for(mod, x)
    sd = {("." + k): v for k, v in _sd(mod).items()}
    if "blk" not in P:
        P["blk"] = eng.pack_block(sd, "")
    p = P["blk"]
    B, n, d = x.shape
    if eng.stream32 and d % 8 == 0:
        t0 = x.reshape(B * n, d).float().contiguous()
    else:
        t0 = eng._new(B * n, d)
        hip.cast_f32(x.reshape(B * n, d).float().contiguous(), t0)
    a2 = _a2vec(eng, sd, "", context, B, d, P)
    t2, t2_32 = eng._block(t0, p, mod.attn1, a2, B, n, None, want32=True)
    return t2_32.reshape(B, n, d).to(x.dtype)


def feedforward_forward(mod, x: torch.Tensor) -> torch.Tensor:
    """FeedForward (GEGLU -> Dropout(0) -> Linear), ``x`` [..., d]."""
    from . import packing
    eng, P = _engine_for(mod, x)
    if "ff" not in P:
        sd = _sd(mod)
        w, b = packing.pack_geglu(sd["net.0.proj.weight"].float().cpu(), sd["net.0.proj.bias"].float().cpu())
        P["ff"] = ({"w": eng._w16(w), "b": eng._f32(b)}, eng.pack_lin(sd, "net.2"))
    ff1, ff2 = P["ff"]
    d = x.shape[-1]
    x16 = eng._new(x.numel() // d, d)
    hip.cast_f32(x.reshape(-1, d).float().contiguous(), x16)
    hmid = eng._new(x16.shape[0], ff1["w"].shape[0] // 2)
    hip.gemm(x16, ff1["w"], hmid, M=x16.shape[0], N=ff1["w"].shape[0], K=d, lda=d, ldc=hmid.shape[1], bias=ff1["b"],
             flags=hip.EPI_GEGLU)
    out = eng._new(x16.shape[0], ff2["w"].shape[0], torch.float32)
    eng._gemm(hmid, ff2, out, flags=hip.EPI_OUT_F32)
    return out.reshape(*x.shape[:-1], out.shape[1]).to(x.dtype)


def geglu_forward(mod, x: torch.Tensor) -> torch.Tensor:
    from . import packing
    eng, P = _engine_for(mod, x)
    if "g" not in P:
        sd = _sd(mod)
        w, b = packing.pack_geglu(sd["proj.weight"].float().cpu(), sd["proj.bias"].float().cpu())
        P["g"] = {"w": eng._w16(w), "b": eng._f32(b)}
    g = P["g"]
    d = x.shape[-1]
    x16 = eng._new(x.numel() // d, d)
    hip.cast_f32(x.reshape(-1, d).float().contiguous(), x16)
    out = eng._new(x16.shape[0], g["w"].shape[0] // 2)
    hip.gemm(x16, g["w"], out, M=x16.shape[0], N=g["w"].shape[0], K=d, lda=d, ldc=out.shape[1], bias=g["b"], flags=hip.EPI_GEGLU)
    return out.reshape(*x.shape[:-1], out.shape[1]).to(x.dtype)
