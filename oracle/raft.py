"""Oracle (test infrastructure only): the optical-flow producer behind ``return_flow``.

``REFace/scripts/temporal_flow.py:27-38, 163-188`` calls ``torchvision.models.optical_flow.raft_large(pretrained=True)`` with
``num_flow_updates=20`` on consecutive frames (``compute_flow(frame2, frame1)``: image1 = frame i + 1, image2 = frame i) and
keeps the last prediction.  torchvision (pinned 0.14.1 by the reference's environment) is a third-party dependency that is NOT
under ``/root/reference`` and NOT installed in this image, and its pretrained weights are not available offline:

    **PARITY UNPINNED.**  This file restates the published RAFT-large architecture of torchvision 0.14
    (``torchvision/models/optical_flow/raft.py``: FeatureEncoder / ResidualBlock / CorrBlock / MotionEncoder / ConvGRU /
    RecurrentBlock / FlowHead / MaskPredictor / ``upsample_flow``, and ``_utils.py``: ``grid_sample``, ``make_coords_grid``)
    from its paper (Teed & Deng, ECCV 2020) and the library's documented structure, in plain fp32 torch.  Neither the module
    code nor a golden vector of it could be checked here; the state-dict key names are the ones that architecture defines and
    are equally unverified.  The HIP path is tested against THIS restatement on synthetic weights.

Nothing here is imported by the product path.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

ENC_LAYERS = (64, 64, 96, 128, 256)
ENC_STRIDES = (2, 1, 2, 2)
CORR_LEVELS, CORR_RADIUS = 4, 4
HIDDEN = 128


def param_shapes() -> Dict[str, Tuple[int, ...]]:
    """State-dict keys and shapes of ``raft_large`` (conv weights [out, in, kh, kw]; BatchNorm of the context encoder with its
    running statistics; InstanceNorm of the feature encoder has no parameters)."""
    s: Dict[str, Tuple[int, ...]] = {}

    def conv(name, cin, cout, kh, kw=None):
        s[name + ".weight"] = (cout, cin, kh, kw or kh)
        s[name + ".bias"] = (cout,)

    def bn(name, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            s[f"{name}.{k}"] = (c,)

    for enc, has_bn in (("feature_encoder", False), ("context_encoder", True)):
        conv(f"{enc}.convnormrelu.0", 3, ENC_LAYERS[0], 7)
        if has_bn:
            bn(f"{enc}.convnormrelu.1", ENC_LAYERS[0])
        cin = ENC_LAYERS[0]
        for li, (cout, stride) in enumerate(zip(ENC_LAYERS[1:4], ENC_STRIDES[1:]), start=1):
            for bi in range(2):
                b = f"{enc}.layer{li}.{bi}"
                c0 = cin if bi == 0 else cout
                conv(f"{b}.convnormrelu1.0", c0, cout, 3)
                conv(f"{b}.convnormrelu2.0", cout, cout, 3)
                if has_bn:
                    bn(f"{b}.convnormrelu1.1", cout)
                    bn(f"{b}.convnormrelu2.1", cout)
                if bi == 0 and stride != 1:
                    conv(f"{b}.downsample.0", c0, cout, 1)
                    if has_bn:
                        bn(f"{b}.downsample.1", cout)
            cin = cout
        conv(f"{enc}.conv", ENC_LAYERS[3], ENC_LAYERS[4], 1)
    ncorr = CORR_LEVELS * (2 * CORR_RADIUS + 1) ** 2
    m = "update_block.motion_encoder"
    conv(f"{m}.convcorr1.0", ncorr, 256, 1)
    conv(f"{m}.convcorr2.0", 256, 192, 3)
    conv(f"{m}.convflow1.0", 2, 128, 7)
    conv(f"{m}.convflow2.0", 128, 64, 3)
    conv(f"{m}.conv.0", 192 + 64, 126, 3)
    for g, (kh, kw) in (("convgru1", (1, 5)), ("convgru2", (5, 1))):
        for c in ("convz", "convr", "convq"):
            conv(f"update_block.recurrent_block.{g}.{c}", HIDDEN + 256, HIDDEN, kh, kw)
    conv("update_block.flow_head.conv1", HIDDEN, 256, 3)
    conv("update_block.flow_head.conv2", 256, 2, 3)
    conv("mask_predictor.convrelu.0", HIDDEN, 256, 3)
    conv("mask_predictor.conv", 256, 8 * 8 * 9, 1)
    return s


def _conv(sd, name, x, stride=1, padding=None):
    w = sd[name + ".weight"]
    if padding is None:
        padding = ((w.shape[2] - 1) // 2, (w.shape[3] - 1) // 2)
    return F.conv2d(x, w, sd[name + ".bias"], stride=stride, padding=padding)


def _norm(sd, name, x, kind):
    if kind == "instance":
        return F.instance_norm(x, eps=1e-5)
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                        training=False, eps=1e-5)


def encoder(sd, enc: str, x: torch.Tensor) -> torch.Tensor:
    kind = "instance" if enc == "feature_encoder" else "batch"
    x = F.relu(_norm(sd, f"{enc}.convnormrelu.1", _conv(sd, f"{enc}.convnormrelu.0", x, stride=ENC_STRIDES[0]), kind))
    for li, stride in enumerate(ENC_STRIDES[1:], start=1):
        for bi in range(2):
            b = f"{enc}.layer{li}.{bi}"
            st = stride if bi == 0 else 1
            y = F.relu(_norm(sd, f"{b}.convnormrelu1.1", _conv(sd, f"{b}.convnormrelu1.0", x, stride=st), kind))
            y = F.relu(_norm(sd, f"{b}.convnormrelu2.1", _conv(sd, f"{b}.convnormrelu2.0", y), kind))
            if st != 1:
                x = _norm(sd, f"{b}.downsample.1", _conv(sd, f"{b}.downsample.0", x, stride=st), kind)
            x = F.relu(x + y)
    return _conv(sd, f"{enc}.conv", x)


def coords_grid(B: int, h: int, w: int, device=None) -> torch.Tensor:
    ys, xs = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].repeat(B, 1, 1, 1)      # channel 0 = x, channel 1 = y


def corr_pyramid(fmap1: torch.Tensor, fmap2: torch.Tensor) -> List[torch.Tensor]:
    B, C, h, w = fmap1.shape
    corr = torch.matmul(fmap1.reshape(B, C, h * w).transpose(1, 2), fmap2.reshape(B, C, h * w)) / math.sqrt(C)
    corr = corr.reshape(B * h * w, 1, h, w)
    pyr = [corr]
    for _ in range(CORR_LEVELS - 1):
        corr = F.avg_pool2d(corr, kernel_size=2, stride=2)
        pyr.append(corr)
    return pyr


def _grid_sample_abs(img, grid_abs):
    h, w = img.shape[-2:]
    xg, yg = grid_abs.split([1, 1], dim=-1)
    xg = 2 * xg / (w - 1) - 1
    if h > 1:
        yg = 2 * yg / (h - 1) - 1
    return F.grid_sample(img, torch.cat([xg, yg], -1), mode="bilinear", align_corners=True)


def corr_lookup(pyr: List[torch.Tensor], coords: torch.Tensor) -> torch.Tensor:
    """[B, 4 * 81, h, w]: per level a (2r+1)^2 window of bilinear samples around coords / 2^level.  The window offsets are
    ``stack(meshgrid(d, d, indexing='ij'), -1)`` ADDED to (x, y): entry (i, j) samples at (x + d_i, y + d_j)."""
    r = CORR_RADIUS
    side = 2 * r + 1
    d = torch.linspace(-r, r, side, device=coords.device)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), -1).view(1, side, side, 2)
    B, _, h, w = coords.shape
    cen = coords.permute(0, 2, 3, 1).reshape(B * h * w, 1, 1, 2)
    out = []
    for vol in pyr:
        out.append(_grid_sample_abs(vol, cen + delta).view(B, h, w, -1))
        cen = cen / 2
    return torch.cat(out, -1).permute(0, 3, 1, 2).contiguous()


def update_block(sd, hidden, context, corr_features, flow):
    m = "update_block.motion_encoder"
    corr = F.relu(_conv(sd, f"{m}.convcorr1.0", corr_features))
    corr = F.relu(_conv(sd, f"{m}.convcorr2.0", corr))
    fl = F.relu(_conv(sd, f"{m}.convflow1.0", flow))
    fl = F.relu(_conv(sd, f"{m}.convflow2.0", fl))
    mo = F.relu(_conv(sd, f"{m}.conv.0", torch.cat([corr, fl], 1)))
    x = torch.cat([context, mo, flow], 1)
    for g in ("convgru1", "convgru2"):
        p = f"update_block.recurrent_block.{g}"
        hx = torch.cat([hidden, x], 1)
        z = torch.sigmoid(_conv(sd, f"{p}.convz", hx))
        r = torch.sigmoid(_conv(sd, f"{p}.convr", hx))
        q = torch.tanh(_conv(sd, f"{p}.convq", torch.cat([r * hidden, x], 1)))
        hidden = (1 - z) * hidden + z * q
    delta = _conv(sd, "update_block.flow_head.conv2", F.relu(_conv(sd, "update_block.flow_head.conv1", hidden)))
    return hidden, delta


def upsample_flow(sd, hidden, flow, factor: int = 8):
    mask = 0.25 * _conv(sd, "mask_predictor.conv", F.relu(_conv(sd, "mask_predictor.convrelu.0", hidden)))
    B, _, h, w = flow.shape
    mask = torch.softmax(mask.view(B, 1, 9, factor, factor, h, w), dim=2)
    up = F.unfold(factor * flow, kernel_size=3, padding=1).view(B, 2, 9, 1, 1, h, w)
    up = torch.sum(mask * up, dim=2)
    return up.permute(0, 1, 4, 2, 5, 3).reshape(B, 2, h * factor, w * factor)


@torch.no_grad()
def raft_forward(sd, image1: torch.Tensor, image2: torch.Tensor, num_flow_updates: int = 20, all_low_res: bool = False):
    """Last flow prediction [B, 2, H, W] (what ``compute_flow`` returns); ``all_low_res`` also returns the 1/8-resolution flow
    after every update (for tests)."""
    B = image1.shape[0]
    fmaps = encoder(sd, "feature_encoder", torch.cat([image1, image2], 0))
    pyr = corr_pyramid(fmaps[:B], fmaps[B:])
    ctx = encoder(sd, "context_encoder", image1)
    hidden, context = torch.tanh(ctx[:, :HIDDEN]), F.relu(ctx[:, HIDDEN:])
    h, w = fmaps.shape[-2:]
    coords0 = coords_grid(B, h, w, image1.device)
    coords1 = coords0.clone()
    lows = []
    for _ in range(num_flow_updates):
        corr_features = corr_lookup(pyr, coords1)
        hidden, delta = update_block(sd, hidden, context, corr_features, coords1 - coords0)
        coords1 = coords1 + delta
        lows.append(coords1 - coords0)
    up = upsample_flow(sd, hidden, coords1 - coords0)
    return (up, lows) if all_low_res else up


def return_flow(sd, video: torch.Tensor, num_flow_updates: int = 20) -> List[torch.Tensor]:
    """temporal_flow.py:163-188: one [1, 2, H, W] flow per consecutive pair, ``compute_flow(video[i + 1], video[i])``."""
    return [raft_forward(sd, video[i + 1:i + 2], video[i:i + 1], num_flow_updates) for i in range(video.shape[0] - 1)]
