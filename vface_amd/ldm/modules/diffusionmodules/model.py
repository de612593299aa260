"""Parameter containers for the first-stage KL-VAE with the module tree -- and therefore the state-dict keys -- of
``REFace/ldm/modules/diffusionmodules/model.py`` (``Encoder`` :368-459, ``Decoder`` :462-568, ``ResnetBlock`` :82-141,
``AttnBlock`` :150-202, ``Downsample`` :60-79, ``Upsample`` :42-57), so ``first_stage_model.*`` of ``last.ckpt`` loads
unchanged.  There is no CPU forward here: ``vface_amd/vae_engine.py`` sequences the HIP kernels."""
from __future__ import annotations

from torch import nn


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


class _NoForward(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is a parameter container; the VAE runs through vface_amd.vae_engine")


class Upsample(_NoForward):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)


class Downsample(_NoForward):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, 3, 2, 0)


class ResnetBlock(_NoForward):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        if conv_shortcut or temb_channels > 0 or dropout != 0.0:
            raise NotImplementedError("the VAE uses ResnetBlock without temb, dropout or a 3x3 shortcut (model.py:401-404)")
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)


class AttnBlock(_NoForward):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)


def _check(attn_resolutions, resamp_with_conv, attn_type, use_linear_attn):
    if len(attn_resolutions) or not resamp_with_conv or attn_type != "vanilla" or use_linear_attn:
        raise NotImplementedError("first_stage_config of project_ffhq.yaml:57-78: attn_resolutions [], conv resampling, "
                                  "vanilla mid attention")


class Encoder(_NoForward):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        _check(attn_resolutions, resamp_with_conv, attn_type, use_linear_attn)
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.z_channels, self.double_z = resolution, in_channels, z_channels, double_z
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            down = nn.Module()
            down.block, down.attn = nn.ModuleList(), nn.ModuleList()
            for _ in range(num_res_blocks):
                down.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, 1, 1)


class Decoder(_NoForward):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", **ignorekwargs):
        super().__init__()
        _check(attn_resolutions, resamp_with_conv, attn_type, use_linear_attn)
        if give_pre_end or tanh_out:
            raise NotImplementedError("give_pre_end / tanh_out are not used by first_stage_config")
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.out_ch, self.z_channels = resolution, out_ch, z_channels
        block_in = ch * ch_mult[self.num_resolutions - 1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block_out = ch * ch_mult[i_level]
            up = nn.Module()
            up.block, up.attn = nn.ModuleList(), nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                up.block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
            self.up.insert(0, up)   # prepend: index = resolution level, as the reference (:526)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)
