"""RAFT-shaped optical-flow producer on the HIP kernels.

``REFace/scripts/temporal_flow.py:27-38`` builds ``torchvision.models.optical_flow.raft_large(pretrained=True)`` and
``compute_flow`` (:33-38) runs it with ``num_flow_updates=20``, keeping the last prediction.  This module is that network for
the MI355X: ``RAFT`` is a parameter container whose ``state_dict`` has the keys of torchvision's ``raft_large`` (so its published
weights load with ``load_state_dict``), ``RaftEngine`` executes it --

  * every convolution on ``vface_conv3x3`` (3x3, stride 1 / 2) or ``vface_im2col`` + ``vface_gemm`` (7x7, 1x5, 5x1, strided 1x1),
    fp16 / bf16 operands, fp32 accumulation; the context encoder's BatchNorm (eval) is folded into its convolutions on the host;
  * the all-pairs correlation as one ``vface_gemm`` per frame pair (fmap1 x fmap2^T, fp32 out), pyramid by ``vface_avgpool2_f32``;
  * InstanceNorm / ReLU / tanh / residual adds, the ConvGRU gates (fp32 master hidden state), the correlation lookup, the flow
    update and the convex upsampling on the glue kernels of ``csrc/raft.hip``.

All frame pairs of a clip are one batch (M = pairs * h/8 * w/8 rows), chunked only by the correlation volume's HBM budget.

**Parity unpinned**: torchvision is third-party, not under the reference tree and not installed here, and its weights are not
available offline.  The architecture is restated from the published model (parameter count 5 257 536, as torchvision documents
for ``raft_large``) and the engine is tested against ``oracle/raft.py`` on synthetic weights; the state-dict key names are the
ones that architecture defines and are equally unverified.  No CPU fallback: without the HIP library every call raises.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import hip, packing

ENC_LAYERS = (64, 64, 96, 128, 256)
ENC_STRIDES = (2, 1, 2, 2)
CORR_LEVELS, CORR_RADIUS = 4, 4
HIDDEN = 128


class _Conv(nn.Module):
    def __init__(self, cin, cout, kh, kw=None):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin, kh, kw or kh))
        self.bias = nn.Parameter(torch.zeros(cout))


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))


def _cna(cin, cout, k, bn):      # Conv2dNormActivation: (conv, norm, activation) -- InstanceNorm / ReLU hold no parameters
    return nn.Sequential(_Conv(cin, cout, k), _BN(cout) if bn else nn.Identity())


class _Block(nn.Module):
    def __init__(self, cin, cout, stride, bn):
        super().__init__()
        self.convnormrelu1 = _cna(cin, cout, 3, bn)
        self.convnormrelu2 = _cna(cout, cout, 3, bn)
        if stride != 1:
            self.downsample = _cna(cin, cout, 1, bn)


class _Encoder(nn.Module):
    def __init__(self, bn):
        super().__init__()
        self.convnormrelu = _cna(3, ENC_LAYERS[0], 7, bn)
        cin = ENC_LAYERS[0]
        for li, (cout, stride) in enumerate(zip(ENC_LAYERS[1:4], ENC_STRIDES[1:]), start=1):
            setattr(self, f"layer{li}", nn.Sequential(_Block(cin, cout, stride, bn), _Block(cout, cout, 1, bn)))
            cin = cout
        self.conv = _Conv(ENC_LAYERS[3], ENC_LAYERS[4], 1)


class _Seq1(nn.Sequential):       # Conv2dNormActivation without a norm: ("0" = conv)
    def __init__(self, cin, cout, k):
        super().__init__(_Conv(cin, cout, k))


class _MotionEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.convcorr1 = _Seq1(CORR_LEVELS * (2 * CORR_RADIUS + 1) ** 2, 256, 1)
        self.convcorr2 = _Seq1(256, 192, 3)
        self.convflow1 = _Seq1(2, 128, 7)
        self.convflow2 = _Seq1(128, 64, 3)
        self.conv = _Seq1(192 + 64, 126, 3)


class _ConvGRU(nn.Module):
    def __init__(self, kh, kw):
        super().__init__()
        for n in ("convz", "convr", "convq"):
            setattr(self, n, _Conv(HIDDEN + 256, HIDDEN, kh, kw))


class _Recurrent(nn.Module):
    def __init__(self):
        super().__init__()
        self.convgru1 = _ConvGRU(1, 5)
        self.convgru2 = _ConvGRU(5, 1)


class _FlowHead(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = _Conv(HIDDEN, 256, 3)
        self.conv2 = _Conv(256, 2, 3)


class _UpdateBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.motion_encoder = _MotionEncoder()
        self.recurrent_block = _Recurrent()
        self.flow_head = _FlowHead()


class _MaskPredictor(nn.Module):
    def __init__(self):
        super().__init__()
        self.convrelu = _Seq1(HIDDEN, 256, 3)
        self.conv = _Conv(256, 8 * 8 * 9, 1)


class RAFT(nn.Module):
    """Parameter container with ``raft_large``'s state-dict layout; ``forward(image1, image2, num_flow_updates)`` returns the
    list the reference indexes with ``[-1]`` -- only the last prediction is materialised (the earlier ones are discarded there)."""

    def __init__(self, compute_dtype: torch.dtype = torch.float16):
        super().__init__()
        self.feature_encoder = _Encoder(bn=False)
        self.context_encoder = _Encoder(bn=True)
        self.update_block = _UpdateBlock()
        self.mask_predictor = _MaskPredictor()
        self.compute_dtype = compute_dtype
        self._engine: Optional[RaftEngine] = None

    @property
    def engine(self) -> "RaftEngine":
        if self._engine is None:
            self._engine = RaftEngine(self.state_dict(), self.compute_dtype, next(self.parameters()).device)
        return self._engine

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def forward(self, image1: torch.Tensor, image2: torch.Tensor, num_flow_updates: int = 12) -> List[torch.Tensor]:
        return [self.engine.flow(image1, image2, num_flow_updates)]


class RaftEngine:
    """Executes the network on device buffers.  ``sd``: a ``raft_large``-layout state dict (any float dtype, any device)."""

    def __init__(self, sd: Dict[str, torch.Tensor], dtype: torch.dtype = torch.float16, device="cuda:0",
                 corr_budget_bytes: int = 8 << 30):
        hip.load()      # no CPU fallback: fail here if the library is missing
        self.dtype, self.dev = dtype, torch.device(device)
        self.corr_budget = corr_budget_bytes
        sd = {k: v.detach().float().cpu() for k, v in sd.items()}
        self.P: Dict[str, dict] = {}
        for enc, bn in (("feature_encoder", False), ("context_encoder", True)):
            self._add(sd, f"{enc}.convnormrelu.0", f"{enc}.convnormrelu.1" if bn else None, cin_pad=8)
            for li in (1, 2, 3):
                for bi in (0, 1):
                    b = f"{enc}.layer{li}.{bi}"
                    self._add(sd, f"{b}.convnormrelu1.0", f"{b}.convnormrelu1.1" if bn else None)
                    self._add(sd, f"{b}.convnormrelu2.0", f"{b}.convnormrelu2.1" if bn else None)
                    if f"{b}.downsample.0.weight" in sd:
                        self._add(sd, f"{b}.downsample.0", f"{b}.downsample.1" if bn else None)
            self._add(sd, f"{enc}.conv")
        m = "update_block.motion_encoder"
        self._add(sd, f"{m}.convcorr1.0", cin_pad=328)
        self._add(sd, f"{m}.convcorr2.0")
        self._add(sd, f"{m}.convflow1.0", cin_pad=8)
        self._add(sd, f"{m}.convflow2.0")
        self._add(sd, f"{m}.conv.0", cout_pad=128)
        for g in ("convgru1", "convgru2"):
            p = f"update_block.recurrent_block.{g}"
            wz, wr = sd[f"{p}.convz.weight"], sd[f"{p}.convr.weight"]
            self._add({"zr.weight": torch.cat([wz, wr], 0), "zr.bias": torch.cat([sd[f"{p}.convz.bias"], sd[f"{p}.convr.bias"]])}, "zr",
                      store=f"{p}.zr")
            self._add(sd, f"{p}.convq")
        self._add(sd, "update_block.flow_head.conv1")
        self._add(sd, "update_block.flow_head.conv2", cout_pad=8)
        self._add(sd, "mask_predictor.convrelu.0")
        self._add(sd, "mask_predictor.conv")

    # ---- weights --------------------------------------------------------------------------------------------------------
    def _add(self, sd, name, bn: Optional[str] = None, cin_pad: Optional[int] = None, cout_pad: Optional[int] = None,
             store: Optional[str] = None):
        w, b = sd[name + ".weight"].clone(), sd[name + ".bias"].clone()
        if bn is not None:      # eval-mode BatchNorm folded into the convolution (an exact refactoring in real arithmetic)
            scale = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + 1e-5)
            w = w * scale[:, None, None, None]
            b = (b - sd[bn + ".running_mean"]) * scale + sd[bn + ".bias"]
        cout, cin, kh, kw = w.shape
        if cout_pad is not None and cout_pad > cout:
            w = torch.cat([w, torch.zeros(cout_pad - cout, cin, kh, kw)], 0)
            b = torch.cat([b, torch.zeros(cout_pad - cout)])
        if (kh, kw) == (3, 3):
            wp, kind = packing.pack_conv3x3(w, cin_pad), "conv3"
        else:       # explicit window matrix / plain GEMM: K order (tap, channel), channels zero-padded
            cp = cin_pad if cin_pad is not None else (cin + 7) // 8 * 8
            t = torch.zeros(w.shape[0], kh * kw, cp)
            t[..., :cin] = w.permute(0, 2, 3, 1).reshape(w.shape[0], kh * kw, cin)
            wp, kind = t.reshape(w.shape[0], kh * kw * cp).contiguous(), "gemm"
        self.P[store or name] = {"w": wp.to(self.dev, self.dtype), "b": b.to(self.dev), "kh": kh, "kw": kw, "cout": w.shape[0],
                                 "cin": cin, "kind": kind}

    # ---- building blocks -------------------------------------------------------------------------------------------------
    def _buf(self, rows, cols, dtype=None):
        return torch.empty(rows, cols, dtype=dtype or self.dtype, device=self.dev)

    def _conv3(self, name, x, out, *, nimg, H, W, cin, ldx, stride=1, out32=False):
        p = self.P[name]
        hip.conv3x3(x, p["w"], out, nimg=nimg, H=H, W=W, cin=cin, cout=p["cout"], ldx=ldx, ldy=out.stride(0), stride=stride,
                    bias=p["b"], flags=hip.EPI_OUT_F32 if out32 else 0)

    def _window(self, name, x, out, *, nimg, H, W, C_, ldx, stride=1, out32=False):
        """kh x kw convolution, 'same' padding, through the explicit window matrix (a 1x1 stride-1 window needs none)."""
        p = self.P[name]
        kh, kw = p["kh"], p["kw"]
        OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
        M, K = nimg * OH * OW, kh * kw * C_
        if kh == kw == 1 and stride == 1:
            a, lda = x, ldx
        else:
            a = self._buf(M, K)
            hip.im2col(x, a, nimg=nimg, H=H, W=W, C_=C_, kh=kh, kw=kw, stride=stride, pad_y=(kh - 1) // 2, pad_x=(kw - 1) // 2, ldx=ldx)
            lda = K
        hip.gemm(a, p["w"], out, M=M, N=p["cout"], K=K, lda=lda, ldc=out.stride(0), bias=p["b"],
                 flags=hip.EPI_OUT_F32 if out32 else 0)

    def _norm_act(self, x, *, nimg, hw, C_, act, inst, residual=None, y=None):
        stats = hip.channel_stats(x, nimg=nimg, hw=hw, C_=C_) if inst else None
        hip.channel_norm_act(x, y if y is not None else x, M=nimg * hw, hw=hw, C_=C_, act=act, stats=stats, residual=residual)

    def _encoder(self, enc: str, x8: torch.Tensor, nimg: int, H: int, W: int) -> torch.Tensor:
        inst = enc == "feature_encoder"
        H0, W0 = H, W
        H, W_ = (H0 - 1) // ENC_STRIDES[0] + 1, (W0 - 1) // ENC_STRIDES[0] + 1
        x = self._buf(nimg * H * W_, ENC_LAYERS[0])
        self._window(f"{enc}.convnormrelu.0", x8, x, nimg=nimg, H=H0, W=W0, C_=8, ldx=8, stride=ENC_STRIDES[0])      # 7x7, stride 2
        self._norm_act(x, nimg=nimg, hw=H * W_, C_=ENC_LAYERS[0], act=hip.ACT_RELU, inst=inst)
        cin = ENC_LAYERS[0]
        for li, (cout, stride) in enumerate(zip(ENC_LAYERS[1:4], ENC_STRIDES[1:]), start=1):
            for bi in (0, 1):
                b = f"{enc}.layer{li}.{bi}"
                st = stride if bi == 0 else 1
                c0 = cin if bi == 0 else cout
                OH, OW = (H - 1) // st + 1, (W_ - 1) // st + 1
                t = self._buf(nimg * OH * OW, cout)
                self._conv3(f"{b}.convnormrelu1.0", x, t, nimg=nimg, H=H, W=W_, cin=c0, ldx=c0, stride=st)
                self._norm_act(t, nimg=nimg, hw=OH * OW, C_=cout, act=hip.ACT_RELU, inst=inst)
                t2 = self._buf(nimg * OH * OW, cout)
                self._conv3(f"{b}.convnormrelu2.0", t, t2, nimg=nimg, H=OH, W=OW, cin=cout, ldx=cout)
                self._norm_act(t2, nimg=nimg, hw=OH * OW, C_=cout, act=hip.ACT_RELU, inst=inst)
                if st != 1:      # relu(norm(conv1x1 stride 2 (x)) + y)
                    s = self._buf(nimg * OH * OW, cout)
                    self._window(f"{b}.downsample.0", x, s, nimg=nimg, H=H, W=W_, C_=c0, ldx=c0, stride=st)
                    self._norm_act(s, nimg=nimg, hw=OH * OW, C_=cout, act=hip.ACT_RELU, inst=inst, residual=t2)
                    x = s
                else:            # relu(x + y)
                    self._norm_act(x, nimg=nimg, hw=OH * OW, C_=cout, act=hip.ACT_RELU, inst=False, residual=t2, y=t2)
                    x = t2
                H, W_ = OH, OW
            cin = cout
        out = self._buf(nimg * H * W_, ENC_LAYERS[4])
        self._window(f"{enc}.conv", x, out, nimg=nimg, H=H, W=W_, C_=ENC_LAYERS[3], ldx=ENC_LAYERS[3])
        return out

    def _tokens8(self, img: torch.Tensor) -> torch.Tensor:
        N, C_, H, W = img.shape
        out = self._buf(N * H * W, 8)
        hip.nchw_to_nhwc(img.float().contiguous(), out, N=N, C_=C_, hw=H * W, cpad=8)
        return out

    # ---- the network -----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def flow(self, image1: torch.Tensor, image2: torch.Tensor, num_flow_updates: int = 20, return_low_res: bool = False):
        """[B, 3, H, W] x 2 in [-1, 1] -> flow [B, 2, H, W] fp32 (the last prediction).  H, W multiples of 8 and >= 128: the
        coarsest of the four pyramid levels must keep 2 x 2 samples (a 1-sample level divides by zero in grid_sample's
        normalisation -- torchvision refuses such inputs too)."""
        if not (image1.is_cuda and image2.is_cuda):
            raise hip.VFaceHipError("the flow producer runs on the GPU: images must be device tensors (no CPU fallback)")
        B, _, H, W = image1.shape
        if image1.shape != image2.shape or image1.shape[1] != 3 or H % 8 or W % 8 or min(H, W) < 128:
            raise hip.VFaceHipError(f"flow: two [B, 3, H, W] batches, H and W multiples of 8 and >= 128; got {tuple(image1.shape)}, {tuple(image2.shape)}")
        h, w = H // 8, W // 8
        per_pair = int(h * w) ** 2 * 4 * 4 // 3 + 1
        chunk = max(1, min(B, self.corr_budget // per_pair))
        outs, lows = [], []
        for b0 in range(0, B, chunk):
            up, low = self._flow_chunk(image1[b0:b0 + chunk], image2[b0:b0 + chunk], num_flow_updates)
            outs.append(up)
            lows.append(low)
        up = torch.cat(outs, 0)
        return (up, torch.cat(lows, 0)) if return_low_res else up

    def _flow_chunk(self, image1, image2, iters):
        B, _, H, W = image1.shape
        h, w = H // 8, W // 8
        hw, M = h * w, B * h * w
        dt = self.dtype
        fm = self._encoder("feature_encoder", self._tokens8(torch.cat([image1, image2], 0)), 2 * B, H, W)      # [2B hw, 256]
        # all-pairs correlation: one GEMM per pair, fp32 out (the 1 / sqrt(256) is applied in the lookup: exact, a power of two)
        vols = [torch.empty(M, h, w, dtype=torch.float32, device=self.dev)]
        v0 = vols[0].view(M, hw)
        for b in range(B):
            hip.gemm(fm[b * hw:(b + 1) * hw], fm[(B + b) * hw:(B + b + 1) * hw], v0[b * hw:(b + 1) * hw], M=hw, N=hw, K=256, lda=256,
                     ldc=hw, flags=hip.EPI_OUT_F32, split_k=False)
        for _ in range(CORR_LEVELS - 1):
            ph, pw = vols[-1].shape[1:]
            nxt = torch.empty(M, ph // 2, pw // 2, dtype=torch.float32, device=self.dev)
            hip.avgpool2_f32(vols[-1], nxt, R=M, h=ph, w=pw)
            vols.append(nxt)
        del fm
        ctx = self._encoder("context_encoder", self._tokens8(image1), B, H, W)                                # [M, 256]
        # hx = [h | context | motion (126) | flow (2)], rhx = [r * h | the same x]: the ConvGRU inputs
        hx, rhx = self._buf(M, 384), self._buf(M, 384)
        h32 = torch.empty(M, HIDDEN, dtype=torch.float32, device=self.dev)
        hip.channel_norm_act(ctx, hx, M=M, hw=hw, C_=HIDDEN, act=hip.ACT_TANH, y32=h32)
        hip.channel_norm_act(ctx[:, HIDDEN:], hx[:, HIDDEN:], M=M, hw=hw, C_=HIDDEN, act=hip.ACT_RELU, ldx=256, ldy=384)
        hip.copy2d(hx[:, HIDDEN:], rhx[:, HIDDEN:], rows=M, cols=HIDDEN, ld_src=384, ld_dst=384)
        flow32 = torch.zeros(M, 2, dtype=torch.float32, device=self.dev)
        flow8 = torch.zeros(M, 8, dtype=dt, device=self.dev)
        corr = torch.zeros(M, 328, dtype=dt, device=self.dev)          # 324 features + 4 zero columns (K % 8)
        c1, cf, f1 = self._buf(M, 256), self._buf(M, 256), self._buf(M, 128)
        zr, z, q = self._buf(M, 256), self._buf(M, HIDDEN), self._buf(M, HIDDEN)
        fh = self._buf(M, 256)
        delta = torch.empty(M, 8, dtype=torch.float32, device=self.dev)
        me, rb = "update_block.motion_encoder", "update_block.recurrent_block"
        relu = lambda t, C_, ld=None: hip.channel_norm_act(t, t, M=M, hw=hw, C_=C_, act=hip.ACT_RELU, ldx=ld, ldy=ld)
        for _ in range(iters):
            hip.corr_lookup(vols, flow32, corr, h=h, w=w, scale=1.0 / math.sqrt(256.0))
            self._window(f"{me}.convcorr1.0", corr, c1, nimg=B, H=h, W=w, C_=328, ldx=328)
            relu(c1, 256)
            self._conv3(f"{me}.convcorr2.0", c1, cf, nimg=B, H=h, W=w, cin=256, ldx=256)                       # -> cf[:, :192]
            self._window(f"{me}.convflow1.0", flow8, f1, nimg=B, H=h, W=w, C_=8, ldx=8)
            relu(f1, 128)
            self._conv3(f"{me}.convflow2.0", f1, cf[:, 192:], nimg=B, H=h, W=w, cin=128, ldx=128)             # -> cf[:, 192:]
            relu(cf, 256)
            self._conv3(f"{me}.conv.0", cf, hx[:, 256:], nimg=B, H=h, W=w, cin=256, ldx=256)                   # -> hx[:, 256:384]
            relu(hx[:, 256:], 128, 384)
            hip.flow_update(flow32, None, [(hx[:, 382:], 384)], dtype=dt)                                     # cat[.., flow]
            hip.copy2d(hx[:, 256:], rhx[:, 256:], rows=M, cols=128, ld_src=384, ld_dst=384)
            for g in ("convgru1", "convgru2"):
                self._window(f"{rb}.{g}.zr", hx, zr, nimg=B, H=h, W=w, C_=384, ldx=384)
                hip.gru_gate(zr, h32, z, rhx, M=M, hidden=HIDDEN, ldrh=384)
                self._window(f"{rb}.{g}.convq", rhx, q, nimg=B, H=h, W=w, C_=384, ldx=384)
                hip.gru_update(q, z, h32, hx, 384, None, 0, M=M, hidden=HIDDEN)
            self._conv3("update_block.flow_head.conv1", hx, fh, nimg=B, H=h, W=w, cin=HIDDEN, ldx=384)
            relu(fh, 256)
            self._conv3("update_block.flow_head.conv2", fh, delta, nimg=B, H=h, W=w, cin=256, ldx=256, out32=True)
            hip.flow_update(flow32, delta, [(flow8, 8)], dtype=dt)
        # convex upsampling of the last flow only (the reference keeps flow_predictions[-1])
        self._conv3("mask_predictor.convrelu.0", hx, fh, nimg=B, H=h, W=w, cin=HIDDEN, ldx=384)
        relu(fh, 256)
        mask = torch.empty(M, 576, dtype=torch.float32, device=self.dev)
        self._window("mask_predictor.conv", fh, mask, nimg=B, H=h, W=w, C_=256, ldx=256, out32=True)
        up = hip.convex_upsample(mask, flow32, B=B, h=h, w=w, mult=0.25)
        low = flow32.view(B, h, w, 2).permute(0, 3, 1, 2).contiguous()
        return up, low
