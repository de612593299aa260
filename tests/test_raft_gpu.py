"""The RAFT-shaped flow producer on the GPU against oracle/raft.py -- a restatement of the PUBLISHED raft_large
architecture (torchvision is third-party and absent: parity with the reference's flow values is unpinned, see oracle/raft.py).
Synthetic weights; the comparisons are stage by stage so a failure names its layer."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from oracle import raft as oraft
from vface_amd.utils import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def sd():
    s = synth.synth_state_dict(oraft.param_shapes(), seed=0)
    # a flow head that moves: the variance-preserving fill gives ~0.2 px per update, make it a few px over the run
    s["update_block.flow_head.conv2.weight"] = s["update_block.flow_head.conv2.weight"] * 4.0
    return s


@pytest.fixture(scope="module")
def eng(sd):
    from vface_amd.raft import RaftEngine
    return RaftEngine(sd, torch.float16, DEV)


def _nchw(tok, n, h, w):
    return tok.float().reshape(n, h, w, -1).permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("enc", ["feature_encoder", "context_encoder"])
def test_encoder_matches_oracle(sd, eng, enc):
    """7x7 stride-2 stem through the explicit window matrix, three residual stages (stride 1 / 2 / 2, strided 1x1 shortcuts),
    InstanceNorm (feature) or folded BatchNorm (context), 1x1 head: [N, 3, 96, 128] -> [N, 256, 12, 16]."""
    x = synth.synth_normal(f"raft.{enc}.x", (2, 3, 96, 128)).clamp(-1, 1)      # (the encoders alone accept any multiple of 8)
    ref = oraft.encoder(sd, enc, x)
    got = _nchw(eng._encoder(enc, eng._tokens8(x.to(DEV)), 2, 96, 128), 2, 12, 16)
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < 3e-3          # 15 rounded fp16 layers


def test_correlation_pyramid_lookup_matches_oracle(eng):
    """All-pairs correlation (GEMM, fp32 out), avg-pool pyramid and the 4 x 81 bilinear window lookup at fractional, partly
    out-of-range coordinates, against corr_pyramid / corr_lookup of the oracle on the same 16-bit features."""
    from vface_amd import hip
    B, h, w = 2, 16, 24
    fm1 = synth.synth_normal("raft.fm1", (B, 256, h, w)).half().float()
    fm2 = synth.synth_normal("raft.fm2", (B, 256, h, w)).half().float()
    flow = synth.synth_normal("raft.lookup.flow", (B, 2, h, w)) * 3.0
    flow[0, :, 0, 0] = torch.tensor([-20.0, 30.0])          # a window entirely outside the map
    pyr = oraft.corr_pyramid(fm1, fm2)
    ref = oraft.corr_lookup(pyr, oraft.coords_grid(B, h, w) + flow)
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(B * h * w, -1).contiguous()
    f1, f2 = tok(fm1).half().to(DEV), tok(fm2).half().to(DEV)
    hw = h * w
    vols = [torch.empty(B * hw, h, w, dtype=torch.float32, device=DEV)]
    for b in range(B):
        hip.gemm(f1[b * hw:(b + 1) * hw], f2[b * hw:(b + 1) * hw], vols[0].view(B * hw, hw)[b * hw:(b + 1) * hw], M=hw, N=hw, K=256,
                 lda=256, ldc=hw, flags=hip.EPI_OUT_F32, split_k=False)
    for _ in range(3):
        ph, pw = vols[-1].shape[1:]
        nxt = torch.empty(B * hw, ph // 2, pw // 2, dtype=torch.float32, device=DEV)
        hip.avgpool2_f32(vols[-1], nxt, R=B * hw, h=ph, w=pw)
        vols.append(nxt)
    for l in range(4):
        assert (vols[l].cpu() / 16.0 - pyr[l][:, 0]).abs().max() < 2e-5 * pyr[0].abs().max()
    out = torch.zeros(B * hw, 328, dtype=torch.float16, device=DEV)
    hip.corr_lookup(vols, tok(flow).to(DEV), out, h=h, w=w, scale=1.0 / 16.0)
    got = _nchw(out[:, :324], B, h, w)
    assert (got - ref).abs().max() < 2e-3 * max(1.0, ref.abs().max().item())       # fp16 storage of O(1) values
    assert out[:, 324:].abs().max() == 0
    # the out-of-range windows read zeros (levels 0-2; at level 3 the coordinate / 8 is back inside the map)
    # (the oracle normalises to [-1, 1] and back, as torchvision's grid_sample helper does: 1e-8-size weights at the -1 border)
    assert got[0, :243, 0, 0].abs().max() == 0 and ref[0, :243, 0, 0].abs().max() < 1e-6 and ref[0, 243:, 0, 0].abs().max() > 1e-3


def test_convex_upsample_matches_oracle(sd):
    from vface_amd import hip
    B, h, w = 2, 8, 10
    hidden = synth.synth_normal("raft.up.h", (B, 128, h, w))
    flow = synth.synth_normal("raft.up.flow", (B, 2, h, w)) * 2.0
    mask = 0.25 * oraft._conv(sd, "mask_predictor.conv", F.relu(oraft._conv(sd, "mask_predictor.convrelu.0", hidden)))
    ref = oraft.upsample_flow(sd, hidden, flow)
    tok = lambda t: t.permute(0, 2, 3, 1).reshape(B * h * w, -1).contiguous()
    got = hip.convex_upsample((tok(mask) / 0.25).to(DEV), tok(flow).to(DEV), B=B, h=h, w=w, mult=0.25).cpu()
    assert got.shape == ref.shape and (got - ref).abs().max() < 1e-4


@pytest.mark.parametrize("H,W,iters", [(128, 128, 4), (128, 192, 6)])
def test_flow_matches_oracle(sd, eng, H, W, iters):
    """The whole network: encoders, correlation, `iters` ConvGRU updates with fp32 hidden state and flow, convex upsampling.
    Bounded per update on the 1/8-resolution flow and on the final field, in pixels and relative to the flow's own size."""
    B = 2
    img = synth.synth_normal("raft.flow.img", (B + 1, 3, H, W)).clamp(-1, 1)
    ref_up, ref_low = oraft.raft_forward(sd, img[1:], img[:-1], iters, all_low_res=True)
    up, low = eng.flow(img[1:].to(DEV), img[:-1].to(DEV), iters, return_low_res=True)
    mag = ref_low[-1].abs().mean().item()
    assert mag > 0.3, "the synthetic network should move by a sizeable fraction of a cell"
    e_low = (low.cpu() - ref_low[-1]).abs().max().item()
    e_up = (up.cpu() - ref_up).abs().max().item()
    print(f"\nraft {H}x{W} {iters} updates: |flow| mean {mag:.3f} cells, max err low-res {e_low:.2e} cells, upsampled {e_up:.2e} px, "
          f"rel-L2 {rel_l2(up.cpu(), ref_up):.2e}")
    assert up.shape == (B, 2, H, W) and torch.isfinite(up).all()
    assert rel_l2(up.cpu(), ref_up) < 1e-2 and e_low < 0.05 * max(1.0, mag)


def test_return_flow_has_the_references_interface(sd):
    """`return_flow(video)` (temporal_flow.py:163-188): B - 1 flows of [1, 2, H, W], flow i from (video[i + 1], video[i]); 20
    updates; the batched run agrees with a pair-by-pair run (nothing mixes samples; the GEMM launches may tile a different
    batch differently, so agreement is to rounding, not bit for bit)."""
    from vface_amd.raft import RAFT
    from vface_amd.scripts import temporal_flow as tf
    model = RAFT()
    model.load_state_dict(sd)
    model = model.to(DEV)
    video = synth.synth_normal("raft.video", (3, 3, 128, 128)).clamp(-1, 1).to(DEV)
    flows = tf.return_flow(video, model)
    assert len(flows) == 2 and all(f.shape == (1, 2, 128, 128) for f in flows)
    one = tf.compute_flow(video[2:3], video[1:2], model)
    assert (one - flows[1]).abs().max().item() < 0.05 * max(1.0, flows[1].abs().max().item())
    with pytest.raises(Exception):
        tf.return_flow(video.cpu(), model)
    tf.set_flow_model(model)
    assert torch.equal(tf.return_flow(video)[0], flows[0])
    tf.set_flow_model(None)


def test_flow_bf16_compute_and_input_validation(sd):
    """bf16 operands (3 fewer mantissa bits: a looser bound), and the shapes the engine refuses."""
    from vface_amd import hip
    from vface_amd.raft import RaftEngine
    eng16 = RaftEngine(sd, torch.bfloat16, DEV)
    img = synth.synth_normal("raft.flow.img", (3, 3, 128, 128)).clamp(-1, 1)
    ref = oraft.raft_forward(sd, img[1:], img[:-1], 3)
    up = eng16.flow(img[1:].to(DEV), img[:-1].to(DEV), 3)
    assert torch.isfinite(up).all() and rel_l2(up.cpu(), ref) < 5e-2
    with pytest.raises(hip.VFaceHipError):
        eng16.flow(img[1:, :, :64, :64].to(DEV), img[:-1, :, :64, :64].to(DEV), 1)      # below 128 x 128
    with pytest.raises(hip.VFaceHipError):
        eng16.flow(img[1:, :, :, :124].to(DEV), img[:-1, :, :, :124].to(DEV), 1)       # not a multiple of 8
    with pytest.raises(hip.VFaceHipError):
        eng16.flow(img[1:], img[:-1], 1)                                                # host tensors
