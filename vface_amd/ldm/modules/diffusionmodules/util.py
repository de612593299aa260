"""Host-side helpers of ``REFace/ldm/modules/diffusionmodules/util.py`` that the VFace path uses:
the beta / DDIM schedules (util.py:21-74) and parameter-container layers.  Schedules are float64 numpy on
the host, exactly as in the reference; tensors' math runs in HIP kernels (``vface_amd.hip``)."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """util.py:21-43 ('linear' is linear in sqrt(beta); the only schedule project_ffhq.yaml uses)."""
    if schedule == "linear":
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2
    if schedule == "sqrt_linear":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64)
    if schedule == "sqrt":
        return np.linspace(linear_start, linear_end, n_timestep, dtype=np.float64) ** 0.5
    raise ValueError(f"schedule '{schedule}' unknown.")


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """util.py:46-61."""
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        steps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        steps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps_out = steps + 1  # "add one to get the final alpha values right"
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """util.py:64-74.  ``alphacums``: fp32 values (numpy or CPU tensor) indexed by timestep."""
    ac = np.asarray(alphacums, dtype=np.float32)
    alphas = ac[ddim_timesteps]
    alphas_prev = np.asarray([ac[0]] + ac[ddim_timesteps[:-1]].tolist(), dtype=np.float64)
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in the following sigma_t schedule "
              f"for ddim sampler {sigmas}")
    return sigmas, alphas, alphas_prev


class GroupNorm32(nn.GroupNorm):
    """util.py:214-216 (parameter container; fp32 statistics are the HIP kernel's)."""


def normalization(channels):
    return GroupNorm32(32, channels)


def conv_nd(dims, *args, **kwargs):
    if dims != 2:
        raise ValueError(f"unsupported dimensions: {dims} (the VFace UNet is 2-D)")
    return nn.Conv2d(*args, **kwargs)


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


def noise_like(shape, device, repeat=False):
    """util.py:264-267."""
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)
