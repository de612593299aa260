"""``AutoencoderKL`` of ``REFace/ldm/models/autoencoder.py:285-335``: same constructor arguments that matter
(``ddconfig``, ``embed_dim``), same state-dict keys (``encoder.* decoder.* quant_conv.* post_quant_conv.*``), same
``encode`` -> posterior / ``decode`` surface; the computation is ``vface_amd.vae_engine.VAEEngine`` on the GPU.
Training, losses, logging and the Lightning base class are out of scope."""
from __future__ import annotations

import torch
from torch import nn

from ..modules.diffusionmodules.model import Decoder, Encoder
from ..modules.distributions.distributions import DiagonalGaussianDistribution  # noqa: F401  (re-export, as the reference)


class AutoencoderKL(nn.Module):
    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=(), image_key="image",
                 colorize_nlabels=None, monitor=None, compute_dtype=torch.float16):
        super().__init__()
        assert ddconfig["double_z"]
        self.image_key = image_key
        self.encoder = Encoder(**ddconfig)
        self.decoder = Decoder(**ddconfig)
        self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self.embed_dim = embed_dim
        self.compute_dtype = compute_dtype
        self._engine = None
        if ckpt_path is not None:
            sd = torch.load(ckpt_path, map_location="cpu")["state_dict"]
            sd = {k: v for k, v in sd.items() if not any(k.startswith(ik) for ik in ignore_keys)}
            self.load_state_dict(sd, strict=False)

    @property
    def engine(self):
        if self._engine is None:
            from ...vae_engine import VAEEngine
            self._engine = VAEEngine(self, self.compute_dtype)
        return self._engine

    def encode(self, x):
        """x: [F, 3, H, W] in [-1, 1] on the GPU -> posterior (autoencoder.py:323-327)."""
        return self.engine.encode(x)

    def decode(self, z):
        """z: [F, embed_dim, h, w] (unscaled) -> [F, 3, 8h, 8w] fp32 (autoencoder.py:329-333)."""
        return self.engine.decode(z)

    def forward(self, input, sample_posterior=True):
        posterior = self.encode(input)
        z = posterior.sample() if sample_posterior else posterior.mode()
        return self.decode(z), posterior


# project_ffhq.yaml:57-78
FFHQ_VAE_CONFIG = dict(embed_dim=4, ddconfig=dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128,
                                                  ch_mult=[1, 2, 4, 4], num_res_blocks=2, attn_resolutions=[], dropout=0.0))
