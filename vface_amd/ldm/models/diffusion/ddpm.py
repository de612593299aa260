"""The thin slice of ``REFace/ldm/models/diffusion/ddpm.py`` that sits on the hot path:
``LatentDiffusion.apply_model`` (ddpm.py:1519-1617) and ``DiffusionWrapper.forward`` (:2231-2257) for
``conditioning_key: crossattn``, plus the schedule buffers of ``DDPM.register_schedule`` the sampler reads.
The first-stage VAE is optional: ``first_stage_config`` builds ``AutoencoderKL`` under ``first_stage_model``
and ``encode_first_stage`` / ``get_first_stage_encoding`` / ``decode_first_stage`` (:1402, :850-857, :1277-1284) work as in
the reference.  Conditioning encoders, losses and training (the other ~2200 lines) are out of scope.

State-dict keys of the UNet are ``model.diffusion_model.*`` as in ``last.ckpt`` so
``load_state_dict(ckpt["state_dict"], strict=False)`` (VFace_inference_batch.py:118-135) fills it unchanged.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from ...modules.diffusionmodules.openaimodel import UNetModel
from ...modules.diffusionmodules.util import make_beta_schedule


class DiffusionWrapper(nn.Module):
    def __init__(self, unet: UNetModel, conditioning_key="crossattn"):
        super().__init__()
        if conditioning_key != "crossattn":
            raise NotImplementedError("the VFace configuration uses conditioning_key='crossattn' (project_ffhq.yaml:13)")
        self.diffusion_model = unet
        self.conditioning_key = conditioning_key

    def forward(self, x, t, c_concat=None, c_crossattn=None):
        cc = c_crossattn[0] if len(c_crossattn) == 1 else torch.cat(c_crossattn, 1)
        return self.diffusion_model(x, t, context=cc)


class LatentDiffusion(nn.Module):
    def __init__(self, unet_config: dict, timesteps=1000, linear_start=0.00085, linear_end=0.012,
                 beta_schedule="linear", scale_factor=0.18215, parameterization="eps", first_stage_config=None):
        super().__init__()
        self.model = DiffusionWrapper(UNetModel(**unet_config))
        if first_stage_config is not None:
            from ..autoencoder import AutoencoderKL
            self.first_stage_model = AutoencoderKL(**first_stage_config)
        self.parameterization = parameterization
        self.scale_factor = scale_factor
        self.num_timesteps = int(timesteps)
        betas = make_beta_schedule(beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end)
        ac = np.cumprod(1.0 - betas, axis=0)
        f32 = lambda a: torch.tensor(a, dtype=torch.float32)
        self.register_buffer("betas", f32(betas), persistent=False)
        self.register_buffer("alphas_cumprod", f32(ac), persistent=False)
        self.register_buffer("alphas_cumprod_prev", f32(np.append(1.0, ac[:-1])), persistent=False)

    @property
    def device(self):
        return self.betas.device

    @property
    def unet(self) -> UNetModel:
        return self.model.diffusion_model

    # ---- first stage (ddpm.py:1402-1420, 850-857, 1277-1300; the patch-split branches are not configured) ----
    def encode_first_stage(self, x):
        return self.first_stage_model.encode(x)

    def get_first_stage_encoding(self, encoder_posterior, noise=None):
        from ...modules.distributions.distributions import DiagonalGaussianDistribution
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            return encoder_posterior.sample(noise, scale=self.scale_factor)   # scale_factor * sample(), one kernel
        if isinstance(encoder_posterior, torch.Tensor):
            return self.scale_factor * encoder_posterior
        raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")

    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        if predict_cids:
            raise NotImplementedError("predict_cids belongs to VQ first stages; the VFace configuration uses AutoencoderKL")
        return self.first_stage_model.decode(1. / self.scale_factor * z)

    def apply_model(self, x_noisy, t, cond):
        if isinstance(cond, dict):
            return self.model(x_noisy, t, **cond)
        if not isinstance(cond, list):
            cond = [cond]
        return self.model(x_noisy, t, c_crossattn=cond)


# project_ffhq.yaml:33-56
FFHQ_UNET_CONFIG = dict(image_size=32, in_channels=9, out_channels=4, model_channels=320,
                        attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8,
                        use_spatial_transformer=True, transformer_depth=1, context_dim=768, use_checkpoint=True,
                        legacy=False, add_conv_in_front_of_unet=False)
