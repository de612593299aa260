"""``DiagonalGaussianDistribution`` of ``REFace/ldm/modules/distributions/distributions.py:24-62`` as the object
``AutoencoderKL.encode`` returns: it keeps the encoder's moments on the device in the layout the HIP kernels produced them
(fp32 NHWC ``[F*h*w, 2*zc]``) and draws / takes the mode through ``vface_vae_sample``."""
from __future__ import annotations

import torch

from .... import hip


class DiagonalGaussianDistribution:
    def __init__(self, moments_nhwc: torch.Tensor, F: int, h: int, w: int, zc: int, deterministic: bool = False):
        self.parameters = moments_nhwc   # fp32 [F*h*w, >= 2*zc]: mean | logvar
        self.F, self.h, self.w, self.zc = F, h, w, zc
        self.deterministic = deterministic

    def _draw(self, noise, scale):
        z = torch.empty(self.F, self.zc, self.h, self.w, dtype=torch.float32, device=self.parameters.device)
        hip.vae_sample(self.parameters, noise, z, F=self.F, hw=self.h * self.w, zc=self.zc, scale=scale)
        return z

    def sample(self, noise: torch.Tensor = None, scale: float = 1.0):
        """mean + std * N(0, 1) (:35-37).  The reference draws with ``torch.randn(shape)`` on the CPU generator and moves
        it to the device; pass ``noise`` to reproduce a given draw."""
        if self.deterministic:
            return self.mode(scale)
        if noise is None:
            noise = torch.randn(self.F, self.zc, self.h, self.w).to(self.parameters.device)
        return self._draw(noise.to(device=self.parameters.device, dtype=torch.float32).contiguous(), scale)

    def mode(self, scale: float = 1.0):
        return self._draw(None, scale)

    @property
    def mean(self):
        return self.mode()
