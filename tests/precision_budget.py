#!/usr/bin/env python3
"""Precision budget of the 16-bit HIP path, emulated on the CPU (test infrastructure, not a test: not collected).

Re-runs the oracle's UNet with the HIP path's own rounding points (one rounding per kernel output, fp32
accumulate / norm statistics / softmax) under switches, so the share of the whole-UNet error that comes from
(a) fp16 weights, (b) the 16-bit residual stream, (c) the 16-bit branch activations can be read before any kernel is
changed:

    python tests/precision_budget.py [--full] [--mode plain]

Everything is compared against the fp32 oracle run (which `tests/test_oracle_golden.py` pins to the reference).
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import hooks as ohooks  # noqa: E402
from oracle import unet as ounet  # noqa: E402
from vface_amd.utils import synth  # noqa: E402


class Emu:
    """rounding switches: w16 (weights), act16 (branch activations = MFMA operands), stream16 (residual carriers)."""

    def __init__(self, sd, w16=True, act16=True, stream16=True, half=torch.float16, w16_filter=None, gn_in16=False, gn_in16_filter=None):
        self.half = half
        self.gn_in16 = gn_in16     # a ResBlock's in_layers GroupNorm normalises the 16-bit copy of its input (statistics stay fp32) ..
        self.gn_in16_filter = gn_in16_filter      # .. in the ResBlocks whose state-dict prefix this predicate accepts (None: all)
        q = lambda t: t.to(half).float()
        self.a = q if act16 else (lambda t: t)
        self.s = q if stream16 else (lambda t: t)
        self.si = self.s           # rounding of a transformer block's INTERIOR running sums (t0, t1); default: as the stream
        self.si_min_c = None       # .. applied only in blocks of at least this many channels (None: all)
        self.sd = {k: v.float() for k, v in sd.items()}
        self.sdh = {k: (q(v.float()) if (w16 and (w16_filter is None or w16_filter(k))) else v.float()) for k, v in sd.items()}

    def w(self, k):
        return self.sdh[k]

    def f(self, k):
        return self.sd[k]


def conv(c, x16, p, stride=1, pad=1):
    return F.conv2d(x16, c.w(p + ".weight"), c.f(p + ".bias"), stride=stride, padding=pad)


def res(c, l, x, emb_all):
    p = l.prefix
    if c.gn_in16 and (c.gn_in16_filter is None or c.gn_in16_filter(p)):   # statistics from the fp32 values (producer epilogue), normalisation applied to the rounded copy
        xr = c.a(x)
        N, C = x.shape[:2]
        xg = x.reshape(N, 32, -1)
        mu, var = xg.mean(-1), xg.var(-1, unbiased=False)
        y = ((xr.reshape(N, 32, -1) - mu[..., None]) * torch.rsqrt(var[..., None] + 1e-5)).reshape(x.shape)
        y = y * c.f(p + ".in_layers.0.weight")[None, :, None, None] + c.f(p + ".in_layers.0.bias")[None, :, None, None]
        a = c.a(F.silu(y))
    else:
        a = c.a(F.silu(F.group_norm(x, 32, c.f(p + ".in_layers.0.weight"), c.f(p + ".in_layers.0.bias"), 1e-5)))
    e = F.linear(emb_all, c.w(p + ".emb_layers.1.weight"), c.f(p + ".emb_layers.1.bias"))   # fp32 row bias
    h1 = c.a(conv(c, a, p + ".in_layers.2") + e[:, :, None, None])
    a2 = c.a(F.silu(F.group_norm(h1, 32, c.f(p + ".out_layers.0.weight"), c.f(p + ".out_layers.0.bias"), 1e-5)))
    h2 = conv(c, a2, p + ".out_layers.3")
    skip = x if l.cin == l.cout else conv(c, c.a(x), p + ".skip_connection", pad=0)
    return c.s(h2 + skip)


def st(c, l, x, context, registry, hw0):
    p = l.prefix
    B, C, H, W = x.shape
    g = c.a(F.group_norm(x, 32, c.f(p + ".norm.weight"), c.f(p + ".norm.bias"), 1e-6))
    si = c.si if (c.si_min_c is None or C >= c.si_min_c) else c.s      # (interior sums rounded only in blocks at least this wide)
    t0 = si(conv(c, g, p + ".proj_in", pad=0)).permute(0, 2, 3, 1).reshape(B, H * W, C)
    tp = p + ".transformer_blocks.0"
    ln = lambda v, q: c.a(F.layer_norm(v, (C,), c.f(f"{tp}.{q}.weight"), c.f(f"{tp}.{q}.bias"), 1e-5))
    cfg = registry.get(f"{tp}.attn1") if registry else None
    # attn1 (kernel rounding points: q,k,v 16-bit; scores/softmax fp32; P 16-bit; O 16-bit; out-proj fp32 sum)
    xl = ln(t0, "norm1")
    q = c.a(xl @ c.w(f"{tp}.attn1.to_q.weight").t()); k = c.a(xl @ c.w(f"{tp}.attn1.to_k.weight").t())
    v = c.a(xl @ c.w(f"{tp}.attn1.to_v.weight").t())
    if cfg is not None:
        q, k, v = ohooks.apply_fusion(q, k, v, cfg, hw0)
        q, k, v = c.a(q), c.a(k), c.a(v)
    dh = C // l.heads
    sp = lambda t: t.reshape(B, t.shape[1], l.heads, dh).permute(0, 2, 1, 3)
    if c.a(torch.tensor(1.0 + 2 ** -12)).item() == 1.0 + 2 ** -12:
        o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), scale=dh ** -0.5)
    else:
        qs = c.a(sp(q) * (dh ** -0.5))
        o = torch.empty_like(qs)
        for b0 in range(B):  # materialised per sample: P is rounded to 16 bits before PV
            P = (qs[b0] @ sp(k)[b0].transpose(-1, -2)).softmax(-1)
            o[b0] = c.a(P) @ sp(v)[b0]   # (kernel: unnormalised P rounded, denominator from the rounded P: same order of error)
    o = c.a(o.permute(0, 2, 1, 3).reshape(B, H * W, C))
    # attn2 on a single token == to_out(to_v(ctx)) broadcast, fp32 row bias
    a2 = F.linear(c.a(F.linear(c.a(context.reshape(B, -1)), c.w(f"{tp}.attn2.to_v.weight"))),
                  c.w(f"{tp}.attn2.to_out.0.weight"), c.f(f"{tp}.attn2.to_out.0.bias"))
    t1 = si(F.linear(o, c.w(f"{tp}.attn1.to_out.0.weight"), c.f(f"{tp}.attn1.to_out.0.bias")) + a2[:, None, :] + t0)
    gg = F.linear(ln(t1, "norm3"), c.w(tp + ".ff.net.0.proj.weight"), c.f(tp + ".ff.net.0.proj.bias"))
    aa, gate = gg.chunk(2, dim=-1)
    ff = c.a(aa * F.gelu(gate))
    t2 = c.s(F.linear(ff, c.w(tp + ".ff.net.2.weight"), c.f(tp + ".ff.net.2.bias")) + t1)
    h = t2.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return c.s(conv(c, c.a(h), p + ".proj_out", pad=0) + x)


def forward(c, spec, x, timesteps, context, registry=None):
    topo = ounet.topology(spec)
    hw0 = (x.shape[2], x.shape[3])
    temb = c.a(ounet.timestep_embedding(timesteps, spec.model_channels))
    e0 = c.a(F.silu(F.linear(temb, c.w("time_embed.0.weight"), c.f("time_embed.0.bias"))))
    emb = c.a(F.silu(F.linear(e0, c.w("time_embed.2.weight"), c.f("time_embed.2.bias"))))

    def run(blk, h):
        for l in blk:
            if l.kind == "conv":
                h = c.s(conv(c, c.a(h), l.prefix))
            elif l.kind == "res":
                h = res(c, l, h, emb)
            elif l.kind == "st":
                h = st(c, l, h, context, registry, hw0)
            elif l.kind == "down":
                h = c.s(conv(c, c.a(h), l.prefix + ".op", stride=2))
            elif l.kind == "up":
                h = c.s(conv(c, F.interpolate(c.a(h), scale_factor=2, mode="nearest"), l.prefix + ".conv"))
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
    a = c.a(F.silu(F.group_norm(h, 32, c.f("out.0.weight"), c.f("out.0.bias"), 1e-5)))
    return conv(c, a, "out.2")


def rel(a, b):
    return ((a - b).norm() / b.norm()).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="the 859.5M configuration (else model_channels 64)")
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--fusion", default="none")
    ap.add_argument("--hw", type=int, default=64)
    ap.add_argument("--bf16", action="store_true")
    a = ap.parse_args()
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    spec = ounet.UNetSpec() if a.full else ounet.UNetSpec(model_channels=64)
    sd = synth.synth_state_dict(ounet.param_shapes(spec), seed=0)
    B = 3 * a.frames
    x = synth.synth_normal("pb.x", (B, 9, a.hw, a.hw))
    ctx = synth.synth_normal("pb.ctx", (B, 1, 768))
    t = torch.full((B,), 481, dtype=torch.long)
    reg = {}
    if a.fusion != "none":
        ohooks.register_spa_attn_injection(reg, ounet.attn1_names(spec), 1, switch_on=True, input_blocks=True,
                                           middle_block=False, output_blocks=False, chunks=3, fusion=a.fusion)
    half = torch.bfloat16 if a.bf16 else torch.float16
    with torch.no_grad():
        t0 = time.time()
        ref = ounet.unet_forward(sd, spec, x, t, ctx, reg)
        print(f"fp32 oracle: {time.time() - t0:.1f} s", flush=True)
        chk = forward(Emu(sd, False, False, False), spec, x, t, ctx, reg)
        print(f"emulator with every switch off vs oracle: {rel(chk, ref):.2e}", flush=True)
        def inner16(**kw):
            e = Emu(sd, half=half, **kw)
            e.si = lambda t: t.to(half).float()
            return e
        y = forward(inner16(w16=True, act16=True, stream16=False), spec, x, t, ctx, reg)
        print(f"{'fp32 main stream, 16-bit block-interior sums (t0, t1)':55s} {rel(y, ref):.3e}", flush=True)
        rows = [
            ("reference autocast rounding points (oracle half=)", None),
            ("HIP path today: w16 + act16 + stream16", dict(w16=True, act16=True, stream16=True)),
            ("fp32 residual stream: w16 + act16", dict(w16=True, act16=True, stream16=False)),
            ("  + ResBlock in-GN fused into conv1 (reads the 16-bit copy)", dict(w16=True, act16=True, stream16=False, gn_in16=True)),
            ("weights only", dict(w16=True, act16=False, stream16=False)),
            ("activations only (16-bit stream)", dict(w16=False, act16=True, stream16=True)),
            ("activations only (fp32 stream)", dict(w16=False, act16=True, stream16=False)),
            ("stream only", dict(w16=False, act16=False, stream16=True)),
            ("conv weights only", dict(w16=True, act16=False, stream16=False,
                                       w16_filter=lambda k: k.endswith("weight") and ("in_layers.2" in k or "out_layers.3" in k or ".op." in k or ".conv." in k or k.startswith("input_blocks.0.0") or k.startswith("out.2")))),
            ("transformer weights only", dict(w16=True, act16=False, stream16=False, w16_filter=lambda k: "transformer_blocks" in k or "proj_in" in k or "proj_out" in k)),
        ]
        for name, kw in rows:
            t0 = time.time()
            if kw is None:
                y = ounet.unet_forward(sd, spec, x, t, ctx, reg, half=half)
            else:
                y = forward(Emu(sd, half=half, **kw), spec, x, t, ctx, reg)
            print(f"{name:55s} {rel(y, ref):.3e}   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
