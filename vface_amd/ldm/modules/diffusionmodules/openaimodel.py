"""Drop-in surface of ``REFace/ldm/modules/diffusionmodules/openaimodel.py`` for the VFace hot path:
``UNetModel`` with the reference's constructor (openaimodel.py:558-588), module tree and state-dict keys
(``input_blocks.i.j...``, ``middle_block.j...``, ``output_blocks.i.j...``, ``time_embed``, ``out``), for the
spatial-transformer configuration of ``project_ffhq.yaml:33-56``.

``UNetModel.forward`` (openaimodel.py:860-907) runs on MI355X through
``vface_amd.engine.UNetEngine``; there is no CPU path.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
from torch import nn

from .... import hip
from ..attention import SpatialTransformer
from .util import conv_nd, linear, normalization


class TimestepBlock(nn.Module):
    """Marker: blocks that take the timestep embedding (openaimodel.py:61-71)."""


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """openaimodel.py:74-88.  Inside ``UNetModel.forward`` the engine dispatches by layer type; called directly, the same
    type dispatch of ``(emb, context)`` as the reference."""

    def forward(self, x, emb, context=None):
        for layer in self:
            if isinstance(layer, TimestepBlock):
                x = layer(x, emb)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            else:
                x = layer(x)
        return x


class Upsample(nn.Module):
    """openaimodel.py:91-119: nearest x2 then 3x3 conv (fused into one implicit-GEMM kernel)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        if not use_conv:
            raise NotImplementedError("conv_resample=False is not part of the VFace configuration")
        self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=padding)

    def forward(self, x):
        from ....module_exec import conv_forward
        return conv_forward(self, x, "up")


class Downsample(nn.Module):
    """openaimodel.py:134-160: 3x3 conv, stride 2, padding 1."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels, self.use_conv, self.dims = channels, out_channels or channels, use_conv, dims
        if not use_conv:
            raise NotImplementedError("conv_resample=False is not part of the VFace configuration")
        self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)

    def forward(self, x):
        from ....module_exec import conv_forward
        return conv_forward(self, x, "down")


class ResBlock(TimestepBlock):
    """openaimodel.py:163-275 (plain residual block: no up/down, no scale-shift norm)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False,
                 use_scale_shift_norm=False, dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        if use_scale_shift_norm or up or down or use_conv:
            raise NotImplementedError("resblock_updown / scale-shift / conv skip are not part of the VFace configuration")
        self.channels, self.emb_channels, self.dropout = channels, emb_channels, dropout
        self.out_channels = out_channels or channels
        self.use_checkpoint = use_checkpoint
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(),
                                       conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1))
        self.skip_connection = (nn.Identity() if self.out_channels == channels
                                else conv_nd(dims, channels, self.out_channels, 1))

    def forward(self, x, emb):
        """openaimodel.py:243-275 on ``x`` [N, C, H, W], ``emb`` [N, emb_channels]."""
        from ....module_exec import resblock_forward
        return resblock_forward(self, x, emb)


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True,
                 add_conv_in_front_of_unet=False, sep_head_att=False, land_mark_id_seperate_layers=False,
                 head_splits=None, compute_dtype=torch.float16):
        super().__init__()
        unsupported = [("use_spatial_transformer", not use_spatial_transformer), ("num_classes", num_classes is not None),
                       ("resblock_updown", resblock_updown), ("use_scale_shift_norm", use_scale_shift_norm),
                       ("n_embed", n_embed is not None), ("add_conv_in_front_of_unet", add_conv_in_front_of_unet),
                       ("sep_head_att", sep_head_att), ("land_mark_id_seperate_layers", land_mark_id_seperate_layers),
                       ("dims", dims != 2), ("conv_resample", not conv_resample)]
        bad = [n for n, b in unsupported if b]
        if bad:
            raise NotImplementedError(f"UNetModel options outside the VFace configuration (project_ffhq.yaml:33-56): {bad}")
        if context_dim is None:
            raise ValueError("context_dim is required with use_spatial_transformer")
        if isinstance(context_dim, (list, tuple)) or type(context_dim).__name__ == "ListConfig":
            context_dim = list(context_dim)[0]
        if num_heads == -1 and num_head_channels == -1:
            raise ValueError("Either num_heads or num_head_channels has to be set")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        self.image_size, self.in_channels, self.model_channels, self.out_channels = image_size, in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, tuple(attention_resolutions)
        self.dropout, self.channel_mult, self.conv_resample = dropout, tuple(channel_mult), conv_resample
        self.num_classes, self.use_checkpoint = num_classes, use_checkpoint
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.num_heads, self.num_head_channels, self.num_heads_upsample = num_heads, num_head_channels, num_heads_upsample
        self.context_dim = context_dim
        self.compute_dtype = compute_dtype

        def heads_for(ch):
            if num_head_channels == -1:
                h, dh = num_heads, ch // num_heads
            else:
                h, dh = ch // num_head_channels, num_head_channels
            if legacy:
                dh = ch // h
            return h, dh

        def transformer(ch):
            h, dh = heads_for(ch)
            return SpatialTransformer(ch, h, dh, depth=transformer_depth, context_dim=context_dim)

        ted = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, ted), nn.SiLU(), linear(ted, ted))
        self.input_blocks = nn.ModuleList(
            [TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        skip_chans = [model_channels]
        ch, ds = model_channels, 1
        for level, mult in enumerate(self.channel_mult):
            for _ in range(num_res_blocks):
                layers = [ResBlock(ch, ted, dropout, out_channels=mult * model_channels, dims=dims,
                                   use_checkpoint=use_checkpoint)]
                ch = mult * model_channels
                if ds in self.attention_resolutions:
                    layers.append(transformer(ch))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                skip_chans.append(ch)
            if level != len(self.channel_mult) - 1:
                self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, dims=dims, out_channels=ch)))
                skip_chans.append(ch)
                ds *= 2
        self.middle_block = TimestepEmbedSequential(
            ResBlock(ch, ted, dropout, dims=dims, use_checkpoint=use_checkpoint), transformer(ch),
            ResBlock(ch, ted, dropout, dims=dims, use_checkpoint=use_checkpoint))
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(self.channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = skip_chans.pop()
                layers = [ResBlock(ch + ich, ted, dropout, out_channels=model_channels * mult, dims=dims,
                                   use_checkpoint=use_checkpoint)]
                ch = model_channels * mult
                if ds in self.attention_resolutions:
                    layers.append(transformer(ch))
                if level and i == num_res_blocks:
                    layers.append(Upsample(ch, conv_resample, dims=dims, out_channels=ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), conv_nd(dims, model_channels, out_channels, 3, padding=1))
        self._engine = None

    # ------------------------------------------------------------------ structure tables for the engine
    @staticmethod
    def _kind(layer):
        if isinstance(layer, ResBlock):
            return "res"
        if isinstance(layer, SpatialTransformer):
            return "st"
        if isinstance(layer, Downsample):
            return "down"
        if isinstance(layer, Upsample):
            return "up"
        if isinstance(layer, nn.Conv2d):
            return "conv"
        raise TypeError(type(layer))

    def block_table(self):
        def blk(prefix, seq):
            return [(self._kind(l), f"{prefix}.{j}", l) for j, l in enumerate(seq)]
        ins = [blk(f"input_blocks.{i}", b) for i, b in enumerate(self.input_blocks)]
        mid = blk("middle_block", self.middle_block)
        outs = [blk(f"output_blocks.{i}", b) for i, b in enumerate(self.output_blocks)]
        return ins, mid, outs

    def layer_table(self) -> List[Tuple[str, str, nn.Module]]:
        ins, mid, outs = self.block_table()
        return [l for b in ins for l in b] + mid + [l for b in outs for l in b]

    @staticmethod
    def block_out_channels(block) -> int:
        kind, _, l = block[-1]
        if kind == "res":
            return l.out_channels
        if kind == "st":
            return l.in_channels
        if kind in ("down", "up"):
            return l.out_channels
        return l.out_channels  # conv

    # ------------------------------------------------------------------ execution
    @property
    def engine(self):
        if self._engine is None or self._engine.dtype != self.compute_dtype:
            from ....engine import UNetEngine
            object.__setattr__(self, "_engine", UNetEngine(self, self.compute_dtype))
        return self._engine

    def forward(self, x, timesteps=None, context=None, y=None, return_features=False, **kwargs):
        """openaimodel.py:860-907.  ``x`` [N, in_channels, H, W], ``timesteps`` [N], ``context`` [N, 1, 768]."""
        if y is not None or return_features:
            raise NotImplementedError("class conditioning / return_features are not part of the VFace path")
        if context is None:
            raise ValueError("context (cross-attention conditioning) is required")
        if not x.is_cuda:
            raise hip.VFaceHipError("UNetModel.forward needs CUDA tensors: the VFace hot path has no CPU fallback")
        return self.engine.forward(x, timesteps, context)
