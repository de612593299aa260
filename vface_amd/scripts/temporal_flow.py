"""``warp_image`` / ``align_by_flow`` / ``return_flow`` (``REFace/scripts/temporal_flow.py:40-53, 163-188, 222-237``) as
stand-alone GPU functions on NCHW tensors.  ``return_flow`` runs the RAFT-shaped network of ``vface_amd/raft.py`` (torchvision's
``raft_large`` is third-party and absent here: parity of the flow VALUES is unpinned, see that module).

Both run ``vface_flow_warp``: the coordinate arithmetic follows the reference's fp32 operation order exactly, so
the integer gather indices are the reference's bit for bit; values are blended in fp32 and stored in the 16-bit
compute type.  Inside the UNet the warp runs directly on the token-major fused q|k buffer instead.
"""
from __future__ import annotations

from typing import Sequence, Union

import torch

from .. import hip
from ..engine import _dev_flow


def _tokens(x: torch.Tensor, dt) -> torch.Tensor:
    F_, C, H, W = x.shape
    cp = (C + 7) // 8 * 8
    out = torch.empty(F_ * H * W, cp, dtype=dt, device=x.device)
    hip.nchw_to_nhwc(x.float().contiguous(), out, N=F_, C_=C, hw=H * W, cpad=cp)
    return out


def _nchw(tok: torch.Tensor, F_, C, H, W) -> torch.Tensor:
    return tok.reshape(F_, H, W, -1)[..., :C].permute(0, 3, 1, 2)


def align_by_flow(x_prev: torch.Tensor = None, flow: Union[Sequence[torch.Tensor], torch.Tensor] = None,
                  alpha: float = 0.5, compute_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """frame i+1 <- alpha*x[i+1] + (1-alpha)*warp(x[i], flow[i]), reading the unmodified source; frame 0 unchanged."""
    if not x_prev.is_cuda:
        raise hip.VFaceHipError("align_by_flow needs CUDA tensors: no CPU fallback on the VFace path")
    F_, C, H, W = x_prev.shape
    fl = _dev_flow(flow, x_prev.device)
    src = _tokens(x_prev, compute_dtype)
    dst = torch.empty_like(src)
    cp = src.shape[1]
    hip.flow_warp(src, dst, fl, F=F_, h=H, w=W, C_=cp, ld_src=cp, fs_src=H * W * cp, ld_dst=cp, fs_dst=H * W * cp,
                  alpha=alpha)
    return _nchw(dst, F_, C, H, W).to(x_prev.dtype)


def warp_image(img: torch.Tensor, flow: torch.Tensor, compute_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """``img`` [B,C,H,W] sampled at (x+dx, y+dy), bilinear, border padding, align_corners=True."""
    if not img.is_cuda:
        raise hip.VFaceHipError("warp_image needs CUDA tensors: no CPU fallback on the VFace path")
    B, C, H, W = img.shape
    fl = flow.to(device=img.device, dtype=torch.float32).reshape(B, 2, H, W).contiguous()
    src = _tokens(img, compute_dtype)
    cp = src.shape[1]
    out = torch.empty_like(src)
    pair_src = torch.empty(2 * H * W, cp, dtype=compute_dtype, device=img.device)
    pair_dst = torch.empty_like(pair_src)
    for b in range(B):  # frame 1 of each pair is a dummy: alpha = 0 makes it the pure warp of frame 0
        pair_src[:H * W] = src[b * H * W:(b + 1) * H * W]
        pair_src[H * W:] = 0
        hip.flow_warp(pair_src, pair_dst, fl[b:b + 1], F=2, h=H, w=W, C_=cp, ld_src=cp, fs_src=H * W * cp, ld_dst=cp,
                      fs_dst=H * W * cp, alpha=0.0)
        out[b * H * W:(b + 1) * H * W] = pair_dst[H * W:]
    return _nchw(out, B, C, H, W).to(img.dtype)


_raft_model = None


def set_flow_model(model) -> None:
    """Install the flow network ``return_flow`` uses: a ``vface_amd.raft.RAFT`` (e.g. with torchvision's ``raft_large`` weights
    loaded through ``load_state_dict``) on the GPU.  The reference builds its model at import time (:27-28)."""
    global _raft_model
    _raft_model = model


@torch.no_grad()
def compute_flow(img1: torch.Tensor, img2: torch.Tensor, model) -> torch.Tensor:
    """``model(img1, img2, num_flow_updates=20)[-1]`` (:33-38): [B, 2, H, W]."""
    return model(img1, img2, num_flow_updates=20)[-1]


@torch.no_grad()
def return_flow(video: torch.Tensor, model=None):
    """``video`` [B, 3, H, W] -> list of B - 1 flows [1, 2, H, W], flow i = ``compute_flow(video[i + 1], video[i])`` (:163-188).
    The reference loops over the pairs; here all pairs are one batch through the network (same values per pair: nothing in
    the network mixes samples), split back into the list the caller indexes."""
    model = model if model is not None else _raft_model
    if model is None:
        raise hip.VFaceHipError("return_flow: no flow model installed (set_flow_model(RAFT().to('cuda')) with weights loaded)")
    if not video.is_cuda:
        raise hip.VFaceHipError("return_flow needs CUDA tensors: no CPU fallback on the VFace path")
    if video.shape[0] < 2:
        return []
    flows = compute_flow(video[1:], video[:-1], model)
    return [flows[i:i + 1] for i in range(flows.shape[0])]
