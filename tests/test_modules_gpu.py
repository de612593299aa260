"""The container modules of the drop-in surface called DIRECTLY (SURVEY 8b "module surface": a reference-side caller may
invoke `SpatialTransformer(x, ctx)`, `ResBlock(x, emb)`, ... attention.py:278-289, openaimodel.py:243-275): every `forward`
runs the HIP kernel sequence of `vface_amd.module_exec` and is compared with the oracle's restatement of the same layer."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import unet as ounet
from vface_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1.2e-3   # a few 16-bit kernels in sequence (per-kernel bound 1e-3, tests/test_kernels_gpu.py)


def _filled(mod, prefix):
    synth.fill_module_(mod, seed=0, prefix=prefix)
    sd = {prefix + k: v.clone() for k, v in mod.state_dict().items()}
    return mod.to(DEV), sd


def test_resblock_down_up_forward():
    from vface_amd.ldm.modules.diffusionmodules.openaimodel import Downsample, ResBlock, TimestepEmbedSequential, Upsample
    for cin, cout in ((64, 128), (128, 128)):
        mod, sd = _filled(ResBlock(cin, 256, 0.0, out_channels=cout), "r.")
        x = synth.synth_normal("mod.res.x", (3, cin, 16, 16))
        emb = synth.synth_normal("mod.res.emb", (3, 256))
        got = mod(x.to(DEV), emb.to(DEV)).float().cpu()
        ref = ounet._res(ounet._Ctx(sd, None), ounet.Layer("res", "r", cin, cout), x, emb)
        assert got.shape == ref.shape and rel_l2(got, ref) < TOL, (cin, cout, rel_l2(got, ref))
    for cls, key, which in ((Downsample, "op", "down"), (Upsample, "conv", "up")):
        mod, sd = _filled(cls(64, True, out_channels=64), "c.")
        x = synth.synth_normal("mod.conv.x", (2, 64, 16, 16))
        got = mod(x.to(DEV)).float().cpu()
        xin = F.interpolate(x, scale_factor=2, mode="nearest") if which == "up" else x
        ref = F.conv2d(xin, sd[f"c.{key}.weight"], sd[f"c.{key}.bias"], stride=2 if which == "down" else 1, padding=1)
        assert rel_l2(got, ref) < TOL, which
    # the reference's type dispatch of (emb, context) (openaimodel.py:74-88)
    seq = TimestepEmbedSequential(ResBlock(64, 256, 0.0, out_channels=64), Downsample(64, True, out_channels=64))
    seq, sd = _filled(seq, "s.")
    x = synth.synth_normal("mod.seq.x", (2, 64, 16, 16))
    emb = synth.synth_normal("mod.seq.emb", (2, 256))
    got = seq(x.to(DEV), emb.to(DEV), None).float().cpu()
    h = ounet._res(ounet._Ctx(sd, None), ounet.Layer("res", "s.0", 64, 64), x, emb)
    ref = F.conv2d(h, sd["s.1.op.weight"], sd["s.1.op.bias"], stride=2, padding=1)
    assert rel_l2(got, ref) < TOL


def test_spatial_transformer_block_ff_geglu_forward():
    from vface_amd.ldm.modules.attention import SpatialTransformer
    c, heads = 128, 8
    mod, sd = _filled(SpatialTransformer(c, heads, c // heads, depth=1, context_dim=768), "t.")
    x = synth.synth_normal("mod.st.x", (3, c, 16, 16))
    ctx = synth.synth_normal("mod.st.ctx", (3, 1, 768))
    got = mod(x.to(DEV), ctx.to(DEV)).float().cpu()
    ref = ounet._st(ounet._Ctx(sd, None), ounet.Layer("st", "t", c, c, heads), x, ctx, None, (16, 16))
    assert rel_l2(got, ref) < TOL, rel_l2(got, ref)
    # the block alone on a token matrix
    blk = mod.transformer_blocks[0]
    tok = synth.synth_normal("mod.blk.x", (3, 256, c))
    got = blk(tok.to(DEV), ctx.to(DEV)).float().cpu()
    tp = "t.transformer_blocks.0"
    cc = ounet._Ctx(sd, None)
    ln = lambda v, q: F.layer_norm(v, (c,), sd[f"{tp}.{q}.weight"], sd[f"{tp}.{q}.bias"], 1e-5)
    from oracle import hooks as ohooks
    att = lambda v, a, cx: ohooks.attention(v, sd[f"{tp}.{a}.to_q.weight"], sd[f"{tp}.{a}.to_k.weight"], sd[f"{tp}.{a}.to_v.weight"],
                                           sd[f"{tp}.{a}.to_out.0.weight"], sd[f"{tp}.{a}.to_out.0.bias"], heads, cx, None, None)
    t = att(ln(tok, "norm1"), "attn1", None) + tok
    t = att(ln(t, "norm2"), "attn2", ctx) + t
    g = F.linear(ln(t, "norm3"), sd[tp + ".ff.net.0.proj.weight"], sd[tp + ".ff.net.0.proj.bias"])
    a, gate = g.chunk(2, dim=-1)
    ref = F.linear(a * F.gelu(gate), sd[tp + ".ff.net.2.weight"], sd[tp + ".ff.net.2.bias"]) + t
    assert rel_l2(got, ref) < TOL, rel_l2(got, ref)
    # FeedForward / GEGLU
    ffx = synth.synth_normal("mod.ff.x", (2, 100, c))
    g = F.linear(ffx, sd[tp + ".ff.net.0.proj.weight"], sd[tp + ".ff.net.0.proj.bias"])
    a, gate = g.chunk(2, dim=-1)
    assert rel_l2(blk.ff.net[0](ffx.to(DEV)).float().cpu(), a * F.gelu(gate)) < TOL
    assert rel_l2(blk.ff(ffx.to(DEV)).float().cpu(),
                  F.linear(a * F.gelu(gate), sd[tp + ".ff.net.2.weight"], sd[tp + ".ff.net.2.bias"])) < TOL
    from vface_amd import hip
    with pytest.raises(hip.VFaceHipError):
        mod(x, ctx)          # CPU tensors: no fallback
