"""Oracle (test infrastructure only): flow-guided temporal smoothing of Q/K maps.

Restates ``REFace/scripts/temporal_flow.py:40-53`` (``warp_image``) and ``:222-237`` (``align_by_flow``)
with the bilinear sampler written out explicitly (ATen ``grid_sampler_2d``, ``align_corners=True``,
``padding_mode='border'``, bilinear) so the integer gather indices are observable.  All arithmetic is
fp32 in the reference's operation order; the index rule is what the HIP kernel must match bit-exactly.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch


def sample_coords(flow: torch.Tensor, cuda_recip_div: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Un-normalised, border-clamped sampling coordinates (ix, iy), each ``[H,W]`` fp32, for one flow
    field ``[2,H,W]`` (channel 0 = dx, 1 = dy, in pixels of the map being warped).

    temporal_flow.py:43-49: ``vgrid = grid + flow``; ``g = 2.0*v/max(W-1,1) - 1.0``.
    ATen GridSampler (align_corners=True): ``ix = ((g + 1) / 2) * (W - 1)``; border: clamp to [0, W-1].

    ``cuda_recip_div``: the tensor / python-scalar division ``2.0*v / max(W-1, 1)`` is a true division in CPU ATen (where
    the golden vectors were made) but CUDA ATen's ``div`` by a scalar multiplies by the fp32 reciprocal
    (``a * (1 / b)``, BinaryDivTrueKernel.cu): the coordinate differs in its last ulp for most inputs and ``floor`` differs
    next to integers.  This flag EMULATES that form (the reference cannot be run on its native device here).
    """
    assert flow.dtype == torch.float32 and flow.dim() == 3 and flow.shape[0] == 2
    _, H, W = flow.shape
    xs = torch.arange(W, dtype=torch.float32).view(1, W).expand(H, W)
    ys = torch.arange(H, dtype=torch.float32).view(H, 1).expand(H, W)
    vx = xs + flow[0]
    vy = ys + flow[1]
    if cuda_recip_div:
        rw = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(float(max(W - 1, 1)), dtype=torch.float32)
        rh = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(float(max(H - 1, 1)), dtype=torch.float32)
        gx = (2.0 * vx) * rw - 1.0
        gy = (2.0 * vy) * rh - 1.0
    else:
        gx = 2.0 * vx / float(max(W - 1, 1)) - 1.0
        gy = 2.0 * vy / float(max(H - 1, 1)) - 1.0
    ix = ((gx + 1.0) / 2.0) * float(W - 1)
    iy = ((gy + 1.0) / 2.0) * float(H - 1)
    ix = torch.clamp(ix, 0.0, float(W - 1))
    iy = torch.clamp(iy, 0.0, float(H - 1))
    return ix, iy


def gather_indices(flow: torch.Tensor, cuda_recip_div: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Integer north-west corner (x0, y0) of the bilinear footprint, int32 ``[H,W]`` each."""
    ix, iy = sample_coords(flow, cuda_recip_div)
    return torch.floor(ix).to(torch.int32), torch.floor(iy).to(torch.int32)


def warp_image(img: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    """``img`` ``[C,H,W]`` fp32 sampled at (x+dx, y+dy): bilinear, border clamp (temporal_flow.py:40-53)."""
    C, H, W = img.shape
    ix, iy = sample_coords(flow)
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    wx1 = ix - x0
    wy1 = iy - y0
    wx0 = 1.0 - wx1
    wy0 = 1.0 - wy1
    x0i = x0.long()
    y0i = y0.long()
    x1i = x0i + 1
    y1i = y0i + 1
    # corners outside the image contribute zero (ATen within_bounds); after border clamping this only
    # happens for x1 == W / y1 == H, whose weight is exactly 0.
    vx1 = (x1i <= W - 1).to(img.dtype)
    vy1 = (y1i <= H - 1).to(img.dtype)
    x1c = x1i.clamp(max=W - 1)
    y1c = y1i.clamp(max=H - 1)
    flat = img.reshape(C, H * W)

    def at(yy, xx):
        return flat[:, (yy * W + xx).reshape(-1)].reshape(C, H, W)

    out = at(y0i, x0i) * (wx0 * wy0)
    out = out + at(y0i, x1c) * (wx1 * wy0 * vx1)
    out = out + at(y1c, x0i) * (wx0 * wy1 * vy1)
    out = out + at(y1c, x1c) * (wx1 * wy1 * vx1 * vy1)
    return out


def align_by_flow(x: torch.Tensor, flow: Sequence[torch.Tensor], alpha: float) -> torch.Tensor:
    """temporal_flow.py:222-237.  ``x`` ``[F,C,H,W]``; ``flow[i]`` ``[1,2,H,W]`` or ``[2,H,W]``.
    Frame 0 unchanged; frame i+1 = alpha*x[i+1] + (1-alpha)*warp(x[i], flow[i]) reading the
    UNMODIFIED source (not recurrent).  Warp in fp32 (grid_sample is on autocast's fp32 list);
    the result takes ``x``'s dtype on store (temporal_flow.py:229,234-235)."""
    out = x.clone()
    for i in range(x.shape[0] - 1):
        f = flow[i]
        f = f.reshape(2, f.shape[-2], f.shape[-1]).float()
        warped = warp_image(x[i].float(), f)
        # natural torch promotion, as in the reference: alpha*x keeps x's dtype (fp16 under autocast),
        # (1-alpha)*warped is fp32, the sum is fp32 and is rounded to x's dtype on store.
        out[i + 1] = (alpha * x[i + 1] + (1.0 - alpha) * warped).to(x.dtype)
    return out


def flow_to_latent(flow_px: torch.Tensor, factor: int = 8) -> torch.Tensor:
    """the reference computes RAFT flow between 512 x 512 frames (``temporal_flow.py:163-188`` ``return_flow``,
    call site ``VFace_inference_batch.py:550-553``) and hands it, unresized, to a warp of the 64 x 64 attention maps, where
    ``grid + flow`` fails on the shape mismatch -- the resample it needs is not defined anywhere in the
    reference.  This build defines it as the area mean over each ``factor x factor`` block of pixels, divided by ``factor``
    (a displacement of d pixels is d / factor latent cells): ``[P, 2, H, W]`` -> ``[P, 2, H/factor, W/factor]``, fp32.
    (RAFT itself is third-party ``torchvision==0.14.1`` weights, absent here: parity of the flow VALUES is unpinned.)"""
    assert flow_px.dim() == 4 and flow_px.shape[1] == 2 and flow_px.shape[2] % factor == 0 and flow_px.shape[3] % factor == 0
    return torch.nn.functional.avg_pool2d(flow_px.float(), factor) / float(factor)
