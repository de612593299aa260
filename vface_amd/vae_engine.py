"""Kernel sequencing for the first-stage KL-VAE: ``AutoencoderKL.encode / decode``
(``REFace/ldm/models/autoencoder.py:323-333``) over ``Encoder`` / ``Decoder``
(``ldm/modules/diffusionmodules/model.py:368-568``) with the kernels of the UNet path:

* every 3x3 convolution is ``vface_conv3x3`` (``Downsample`` :72-77 = trailing zero padding + stride 2, ``Upsample`` :55-58 =
  the fused nearest x2), with the producer-side column statistics feeding the next GroupNorm (eps 1e-6, swish);
* ``AttnBlock`` (:176-202) has ONE head of all 512 channels, beyond what the streaming attention kernel keeps in LDS, and
  runs once per frame, not per step: scores = ``vface_gemm`` (fp32 out) per image, ``vface_softmax_rows``, then
  ``P @ V`` as a GEMM against V^T, which is produced directly as ``Wv @ h^T`` (the value bias moves to the output column
  bias: the rows of P sum to one);
* ``quant_conv`` (1x1 after ``conv_out``, no nonlinearity between) is folded into ``conv_out`` in fp64; ``post_quant_conv``
  keeps its own tiny GEMM (its bias does not commute with ``conv_in``'s zero padding); ``1/scale_factor`` is not applied
  here (``decode_first_stage`` does that, ddpm.py:1284).

Activations are NHWC 16-bit ``[F*H*W, C]`` as in the UNet engine; frames are processed in chunks so every tensor view
stays below the 4 GiB the buffer descriptors address.  No CPU path."""
from __future__ import annotations

from typing import Optional

import torch

from . import hip, packing
from .engine import Act, _phase_form_pays
from .ldm.modules.distributions.distributions import DiagonalGaussianDistribution


class VAEEngine:
    def __init__(self, vae, dtype: torch.dtype = torch.float16, max_frames: int = 8):
        self.vae = vae
        self.dtype = dtype
        self.max_frames = max_frames
        self._packed = None
        hip.load()

    @property
    def device(self):
        return next(self.vae.parameters()).device

    # ------------------------------------------------------------------ weights
    def _w16(self, t):
        return t.detach().to(device=self.device, dtype=self.dtype).contiguous()

    def _f32(self, t):
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def pack(self):
        if not next(self.vae.parameters()).is_cuda:
            raise hip.VFaceHipError("AutoencoderKL parameters are not on the GPU: the VFace path has no CPU fallback")
        sd = {k: v.detach().double().cpu() for k, v in self.vae.state_dict().items()}
        P = {}

        def conv3(pre, w=None, b=None, cout_pad=None):
            w = sd[pre + ".weight"] if w is None else w
            b = sd[pre + ".bias"] if b is None else b
            cout, cin = w.shape[0], w.shape[1]
            if cout_pad is not None and cout_pad > cout:   # GEMM N must be a multiple of 4: zero output channels
                w = torch.cat([w, torch.zeros(cout_pad - cout, *w.shape[1:], dtype=w.dtype)], 0)
                b = torch.cat([b, torch.zeros(cout_pad - cout, dtype=b.dtype)])
            return {"w": self._w16(packing.pack_conv3x3(w.float())), "b": self._f32(b), "cin": cin,
                    "cinp": (cin + 7) // 8 * 8, "cout": w.shape[0], "cout_true": cout}

        def lin(pre, rows=slice(None)):
            w = sd[pre + ".weight"]
            return {"w": self._w16(w.reshape(w.shape[0], w.shape[1])[rows]), "b": self._f32(sd[pre + ".bias"][rows])}

        def gn(pre):
            return (self._f32(sd[pre + ".weight"]), self._f32(sd[pre + ".bias"]))

        def res(pre):
            d = {"norm1": gn(pre + ".norm1"), "conv1": conv3(pre + ".conv1"), "norm2": gn(pre + ".norm2"),
                 "conv2": conv3(pre + ".conv2")}
            if pre + ".nin_shortcut.weight" in sd:
                d["nin"] = lin(pre + ".nin_shortcut")
            return d

        def attn(pre):
            wq, wk = sd[pre + ".q.weight"], sd[pre + ".k.weight"]
            c = wq.shape[0]
            return {"norm": gn(pre + ".norm"), "c": c,
                    "qk": {"w": self._w16(torch.cat([wq.reshape(c, c), wk.reshape(c, c)], 0)),
                           "b": self._f32(torch.cat([sd[pre + ".q.bias"], sd[pre + ".k.bias"]]))},
                    "wv": self._w16(sd[pre + ".v.weight"].reshape(c, c)), "bv": self._f32(sd[pre + ".v.bias"]),
                    "proj": lin(pre + ".proj_out")}

        enc, dec = self.vae.encoder, self.vae.decoder
        P["enc.conv_in"] = conv3("encoder.conv_in")
        for lvl in range(enc.num_resolutions):
            for blk in range(enc.num_res_blocks):
                P[f"enc.down.{lvl}.{blk}"] = res(f"encoder.down.{lvl}.block.{blk}")
            if lvl != enc.num_resolutions - 1:
                P[f"enc.down.{lvl}.ds"] = conv3(f"encoder.down.{lvl}.downsample.conv")
        P["enc.mid1"], P["enc.attn"], P["enc.mid2"] = res("encoder.mid.block_1"), attn("encoder.mid.attn_1"), res("encoder.mid.block_2")
        P["enc.norm_out"] = gn("encoder.norm_out")
        # quant_conv o conv_out, folded in fp64 (both linear, nothing between them: autoencoder.py:324-325)
        wq = sd["quant_conv.weight"].reshape(sd["quant_conv.weight"].shape[0], -1)
        wc, bc = sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"]
        P["enc.conv_out"] = conv3("", w=torch.einsum("om,mikl->oikl", wq, wc), b=wq @ bc + sd["quant_conv.bias"])
        zc, ed = dec.z_channels, self.vae.embed_dim
        wpq = torch.zeros(8, 8, dtype=torch.float64); wpq[:zc, :ed] = sd["post_quant_conv.weight"].reshape(zc, ed)
        bpq = torch.zeros(8, dtype=torch.float64); bpq[:zc] = sd["post_quant_conv.bias"]
        P["dec.post_quant"] = {"w": self._w16(wpq), "b": self._f32(bpq)}
        P["dec.conv_in"] = conv3("decoder.conv_in")
        P["dec.mid1"], P["dec.attn"], P["dec.mid2"] = res("decoder.mid.block_1"), attn("decoder.mid.attn_1"), res("decoder.mid.block_2")
        for lvl in range(dec.num_resolutions):
            for blk in range(dec.num_res_blocks + 1):
                P[f"dec.up.{lvl}.{blk}"] = res(f"decoder.up.{lvl}.block.{blk}")
            if lvl != 0:
                P[f"dec.up.{lvl}.us"] = conv3(f"decoder.up.{lvl}.upsample.conv")
                P[f"dec.up.{lvl}.us"]["phases"] = self._w16(packing.pack_upsample_phases(
                    sd[f"decoder.up.{lvl}.upsample.conv.weight"].float()))
        P["dec.norm_out"] = gn("decoder.norm_out")
        P["dec.conv_out"] = conv3("decoder.conv_out", cout_pad=4)
        self._packed = P

    def _ensure_packed(self):
        if self._packed is None:
            self.pack()

    # ------------------------------------------------------------------ building blocks
    def _new(self, rows, cols, dtype=None):
        return torch.empty(rows, cols, dtype=dtype or self.dtype, device=self.device)

    def _cs(self, rows, cols, hw):
        if hw % 64 or cols % 8:
            return None
        return torch.empty(rows // 64, cols, 2, dtype=torch.float32, device=self.device)

    def _gn(self, x: Act, gn, silu: bool) -> Act:
        if x.cs is not None:
            st = hip.groupnorm_stats_from_cols(x.cs, nimg=x.N, hw=x.hw, C_=x.C, eps=1e-6)
        else:
            st = hip.groupnorm_stats(x.t, nimg=x.N, hw=x.hw, C_=x.C, ldx=x.ld, eps=1e-6)
        y = self._new(x.M, x.C)
        hip.groupnorm_apply(x.t, st, gn[0], gn[1], y, nimg=x.N, hw=x.hw, C_=x.C, ldx=x.ld, ldy=x.C, silu=silu)
        return Act(y, x.N, x.H, x.W)

    def _conv(self, x: Act, w: dict, stride=1, upsample=False, trailing_pad=False, residual=None, out_f32=False) -> Act:
        VH, VW = (2 * x.H, 2 * x.W) if upsample else (x.H, x.W)
        OH, OW = (VH - 1) // stride + 1, (VW - 1) // stride + 1
        rows = x.N * OH * OW
        out = self._new(rows, w["cout"], torch.float32 if out_f32 else None)
        cs = None if out_f32 else self._cs(rows, w["cout"], OH * OW)
        assert x.C == w["cinp"], (x.C, w["cinp"])
        flags = (hip.EPI_OUT_F32 if out_f32 else 0) | (hip.CONV_PAD_TRAILING if trailing_pad else 0)
        if upsample and "phases" in w and residual is None and not out_f32 and (x.H * x.W) % 64 == 0 \
                and _phase_form_pays(x.H * x.W, w["cout"]):
            # Upsample (model.py:55-58) as four parity-phase 2x2 convs with pre-summed taps: 4/9 of the multiply-adds
            hip.upsample2x_conv3x3(x.t, w["phases"], out, nimg=x.N, H=x.H, W=x.W, cin=w["cinp"], cout=w["cout"], ldx=x.ld,
                                   ldy=out.stride(0), bias=w["b"], colstats=cs)
            return Act(out, x.N, OH, OW, cs)
        hip.conv3x3(x.t, w["w"], out, nimg=x.N, H=x.H, W=x.W, cin=w["cinp"], cout=w["cout"], ldx=x.ld, ldy=out.stride(0),
                    stride=stride, upsample=upsample, bias=w["b"], residual=residual,
                    ldr=residual.stride(0) if residual is not None else 0, flags=flags, colstats=cs)
        return Act(out, x.N, OH, OW, cs)

    def _res(self, x: Act, p: dict) -> Act:
        """ResnetBlock.forward (model.py:118-141), temb None."""
        h = self._conv(self._gn(x, p["norm1"], True), p["conv1"])
        h = self._gn(h, p["norm2"], True)
        if "nin" in p:
            skip = self._new(x.M, p["conv2"]["cout"])
            hip.gemm(x.t, p["nin"]["w"], skip, M=x.M, N=skip.shape[1], K=x.C, lda=x.ld, ldc=skip.shape[1], bias=p["nin"]["b"],
                     rows_per_sample=x.hw)
        else:
            skip = x.t
        return self._conv(h, p["conv2"], residual=skip)

    def _attn(self, x: Act, p: dict) -> Act:
        """AttnBlock.forward (model.py:176-202)."""
        c, n = p["c"], x.hw
        g = self._gn(x, p["norm"], False)
        qk = self._new(x.M, 2 * c)
        hip.gemm(g.t, p["qk"]["w"], qk, M=x.M, N=2 * c, K=c, lda=c, ldc=2 * c, bias=p["qk"]["b"])
        att = self._new(x.M, c)
        scores = torch.empty(n, n, dtype=torch.float32, device=self.device)
        prob = self._new(n, n)
        vt = self._new(c, n)
        for b in range(x.N):
            rows = slice(b * n, (b + 1) * n)
            hip.gemm(qk[rows], qk[rows, c:], scores, M=n, N=n, K=c, lda=2 * c, ldw=2 * c, ldc=n, flags=hip.EPI_OUT_F32)
            hip.softmax_rows(scores, prob, M=n, N=n, scale=float(int(c) ** -0.5))
            hip.gemm(p["wv"], g.t[rows], vt, M=c, N=n, K=c, lda=c, ldw=c, ldc=n)              # V^T = Wv h^T
            hip.gemm(prob, vt, att[rows], M=n, N=c, K=n, lda=n, ldw=n, ldc=c, bias=p["bv"])   # P V + bv (rows of P sum to 1)
        out = self._new(x.M, c)
        cs = self._cs(x.M, c, n)
        hip.gemm(att, p["proj"]["w"], out, M=x.M, N=c, K=c, lda=c, ldc=c, bias=p["proj"]["b"], residual=x.t, ldr=x.ld,
                 colstats=cs, rows_per_sample=n)
        return Act(out, x.N, x.H, x.W, cs)

    # ------------------------------------------------------------------ encode / decode
    def _chunks(self, F):
        return [(f0, min(F, f0 + self.max_frames)) for f0 in range(0, F, self.max_frames)]

    def encode(self, x: torch.Tensor) -> DiagonalGaussianDistribution:
        self._ensure_packed()
        if not x.is_cuda:
            raise hip.VFaceHipError("input is not on the GPU: the VFace path has no CPU fallback")
        P, enc = self._packed, self.vae.encoder
        F, C, H, W = x.shape
        if C != enc.in_channels or H % (1 << (enc.num_resolutions - 1)) or W % (1 << (enc.num_resolutions - 1)):
            raise hip.VFaceHipError(f"encode expects [F, {enc.in_channels}, H, W] with H, W multiples of "
                                    f"{1 << (enc.num_resolutions - 1)}; got {tuple(x.shape)}")
        zc = enc.z_channels
        h, w = H >> (enc.num_resolutions - 1), W >> (enc.num_resolutions - 1)
        moments = torch.empty(F * h * w, P["enc.conv_out"]["cout"], dtype=torch.float32, device=self.device)
        for f0, f1 in self._chunks(F):
            n = f1 - f0
            xin = self._new(n * H * W, 8)
            hip.nchw_to_nhwc(x[f0:f1].float().contiguous(), xin, N=n, C_=C, hw=H * W, cpad=8)
            a = self._conv(Act(xin, n, H, W), P["enc.conv_in"])
            for lvl in range(enc.num_resolutions):
                for blk in range(enc.num_res_blocks):
                    a = self._res(a, P[f"enc.down.{lvl}.{blk}"])
                if lvl != enc.num_resolutions - 1:
                    a = self._conv(a, P[f"enc.down.{lvl}.ds"], stride=2, trailing_pad=True)
            a = self._res(a, P["enc.mid1"])
            a = self._attn(a, P["enc.attn"])
            a = self._res(a, P["enc.mid2"])
            a = self._conv(self._gn(a, P["enc.norm_out"], True), P["enc.conv_out"], out_f32=True)
            moments[f0 * h * w:f1 * h * w] = a.t
        return DiagonalGaussianDistribution(moments, F, h, w, zc)

    def decode(self, z: torch.Tensor) -> torch.Tensor:
        self._ensure_packed()
        if not z.is_cuda:
            raise hip.VFaceHipError("latents are not on the GPU: the VFace path has no CPU fallback")
        P, dec = self._packed, self.vae.decoder
        F, C, h, w = z.shape
        if C != self.vae.embed_dim:
            raise hip.VFaceHipError(f"decode expects [F, {self.vae.embed_dim}, h, w]; got {tuple(z.shape)}")
        up = 1 << (dec.num_resolutions - 1)
        out = torch.empty(F, dec.out_ch, h * up, w * up, dtype=torch.float32, device=self.device)
        for f0, f1 in self._chunks(F):
            n = f1 - f0
            zin = self._new(n * h * w, 8)
            hip.nchw_to_nhwc(z[f0:f1].float().contiguous(), zin, N=n, C_=C, hw=h * w, cpad=8)
            z8 = self._new(n * h * w, 8)
            hip.gemm(zin, P["dec.post_quant"]["w"], z8, M=n * h * w, N=8, K=8, lda=8, ldc=8, bias=P["dec.post_quant"]["b"])
            a = self._conv(Act(z8, n, h, w), P["dec.conv_in"])
            a = self._res(a, P["dec.mid1"])
            a = self._attn(a, P["dec.attn"])
            a = self._res(a, P["dec.mid2"])
            for lvl in reversed(range(dec.num_resolutions)):
                for blk in range(dec.num_res_blocks + 1):
                    a = self._res(a, P[f"dec.up.{lvl}.{blk}"])
                if lvl != 0:
                    a = self._conv(a, P[f"dec.up.{lvl}.us"], upsample=True)
            a = self._conv(self._gn(a, P["dec.norm_out"], True), P["dec.conv_out"], out_f32=True)
            hip.nhwc_to_nchw_f32(a.t, out[f0:f1], N=n, C_=dec.out_ch, hw=a.H * a.W, ldx=a.t.stride(0))
        return out
