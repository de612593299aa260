"""Oracle (test infrastructure only): the DDIM sampler of the VFace path.

Restates ``REFace/ldm/models/diffusion/ddim_w_inv.py``: ``make_schedule`` (``:155-184``), the sampling loop
with its per-step hook registration (``ddim_sampling`` ``:254-355``), ``p_sample_ddim_with_inverse``
(``:621-738``) and ``ddim_invert`` (``:360-490``); plus ``make_beta_schedule('linear')``,
``make_ddim_timesteps('uniform')`` and ``make_ddim_sampling_parameters``
(``REFace/ldm/modules/diffusionmodules/util.py:21-74``) and the ``alphas_cumprod`` buffers of
``DDPM.register_schedule`` (``REFace/ldm/models/diffusion/ddpm.py``, linear_start 0.00085, linear_end 0.012,
``project_ffhq.yaml:5-6``).

The per-step recon latents the reference ``torch.load``s from disk inside the loop (``:22-26,628``) are
passed as a mapping ``{timestep -> [F,4,h,w]}``.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Mapping, Optional, Sequence

import numpy as np
import torch

from . import hooks as ohooks
from . import unet as ounet


def alphas_cumprod(timesteps: int = 1000, linear_start: float = 0.00085, linear_end: float = 0.012) -> np.ndarray:
    """util.py:21-25 ('linear' schedule is linear in sqrt(beta)), float64."""
    betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas, axis=0)


def ddim_timesteps(S: int, T: int = 1000) -> np.ndarray:
    """util.py:46-61, 'uniform': range(0, T, T//S) + 1."""
    c = T // S
    return np.asarray(list(range(0, T, c))) + 1


class Schedule:
    """ddim_w_inv.py:155-184 for eta given; arrays indexed by DDIM index (ascending time)."""

    def __init__(self, S: int, eta: float = 0.0, T: int = 1000):
        ac = alphas_cumprod(T)
        self.alphas_cumprod = torch.from_numpy(ac).to(torch.float32)  # to_torch: fp32 (:161)
        self.timesteps = ddim_timesteps(S, T)
        # make_ddim_sampling_parameters is fed the fp32 tensor on CPU (:175): index it as the reference does
        acf = self.alphas_cumprod
        a = acf[self.timesteps]
        a_prev = torch.tensor([acf[0].item()] + acf[self.timesteps[:-1]].tolist(), dtype=torch.float64).numpy()
        a = a.numpy()
        self.sigmas = eta * np.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
        self.alphas = a
        self.alphas_prev = a_prev
        self.sqrt_one_minus_alphas = np.sqrt(1.0 - a)


def cfg_combine(e_u, e_c, e_r, s: float):
    """ddim_w_inv.py:666-667 (recon branch formula reproduced as written)."""
    return e_u + s * (e_c - e_u), e_r + s * (e_r - e_u)


def ddim_update(x, e_t, a_t: float, a_prev: float, sigma_t: float, sqrt_1m_at: float, noise=None):
    """ddim_w_inv.py:677-700: pred_x0, dir_xt, x_prev (fp32; python scalars become fp32 via torch.full)."""
    a_t = torch.tensor(a_t, dtype=torch.float32)
    a_prev = torch.tensor(a_prev, dtype=torch.float32)
    sigma_t = torch.tensor(sigma_t, dtype=torch.float32)
    s1 = torch.tensor(sqrt_1m_at, dtype=torch.float32)
    pred_x0 = (x - s1 * e_t) / a_t.sqrt()
    dir_xt = (1.0 - a_prev - sigma_t ** 2).sqrt() * e_t
    n = sigma_t * (noise if noise is not None else torch.zeros_like(x))
    return a_prev.sqrt() * pred_x0 + dir_xt + n, pred_x0


def sample(apply_model: Callable, spec_names: Dict[str, List[str]], S: int, x_T: torch.Tensor,
           cond: torch.Tensor, uncond: torch.Tensor, target_cond: torch.Tensor,
           inv_latents: Mapping[int, torch.Tensor], inpaint_image: torch.Tensor, inpaint_mask: torch.Tensor,
           scale: float = 3.0, eta: float = 0.0, flow=None, steps_limit: Optional[int] = None,
           hook_schedule: str = "shipped", fusion: str = "flow_fix", split_ratio_fft: float = 0.8,
           alpha: float = 0.8):
    """ddim_sampling (:254-355) + p_sample_ddim_with_inverse (:621-738).

    ``apply_model(x_in[3F,9,h,w], t_in[3F], c_in[3F,1,768], registry) -> eps[3F,4,h,w]``.
    ``hook_schedule='shipped'`` reproduces :303,:305 — every step: all attn1 OFF, then the input-block attn1
    modules ON with ``fusion`` (default flow_fix).  ``'none'`` leaves every hook off (inversion)."""
    sch = Schedule(S, eta)
    img = x_T
    F_ = x_T.shape[0]
    time_range = np.flip(sch.timesteps)
    total = len(time_range)
    registry: Dict[str, ohooks.HookCfg] = {}
    trace = []
    for i, step in enumerate(time_range):
        if steps_limit is not None and i >= steps_limit:
            break
        ohooks.register_spa_attn_injection(registry, spec_names, 1, switch_on=False, input_blocks=True,
                                           middle_block=True, output_blocks=True, flow=flow, chunks=3,
                                           block_indices=list(range(9)), fusion="flow_fix")
        if hook_schedule == "shipped":
            ohooks.register_spa_attn_injection(registry, spec_names, 1, switch_on=True, input_blocks=True,
                                               middle_block=False, output_blocks=False, flow=flow, chunks=3,
                                               block_indices=list(range(9)), fusion=fusion,
                                               split_ratio_fft=split_ratio_fft, alpha=alpha)
        index = total - i - 1
        t = torch.full((F_,), int(step), dtype=torch.long)
        inv_t = inv_latents[int(step)]
        x9 = torch.cat([img, inpaint_image, inpaint_mask], dim=1)
        r9 = torch.cat([inv_t, inpaint_image, inpaint_mask], dim=1)
        x_in = torch.cat([x9, x9, r9], dim=0)  # [uncond ; cond ; recon] (:654-655)
        t_in = torch.cat([t] * 3)
        c_in = torch.cat([uncond, cond, target_cond], dim=0)  # (:661-662)
        e_u, e_c, e_r = apply_model(x_in, t_in, c_in, registry).chunk(3)
        e_t, _e_rec = cfg_combine(e_u, e_c, e_r, scale)
        noise = torch.randn_like(img) if eta != 0.0 else None
        img, pred_x0 = ddim_update(img, e_t, float(sch.alphas[index]), float(sch.alphas_prev[index]),
                                   float(sch.sigmas[index]), float(sch.sqrt_one_minus_alphas[index]), noise)
        trace.append(img)
    return img, trace


def invert(apply_model: Callable, S: int, x0: torch.Tensor, cond: torch.Tensor, inpaint_image: torch.Tensor,
           inpaint_mask: torch.Tensor, batch_size: int, steps_limit: Optional[int] = None):
    """ddim_invert (:360-490), hooks off, no CFG (:426-427); returns {timestep -> target half} as the reference
    saves them (:464,:483-486).  Update (:449): with ``cur = max(0, step - 1000//S)``,
    ``x <- (x - sqrt(1-a_cur) e) * sqrt(a_step)/sqrt(a_cur) + sqrt(1-a_step) e``."""
    sch = Schedule(S, 0.0)
    ac = sch.alphas_cumprod
    x = x0
    b = x.shape[0]
    out: Dict[int, torch.Tensor] = {}
    n = len(sch.timesteps)
    for i, step in enumerate(sch.timesteps):
        if steps_limit is not None and i >= steps_limit:
            break
        t = torch.full((b,), int(step), dtype=torch.long)
        x9 = torch.cat([x, inpaint_image, inpaint_mask], dim=1)
        e_t = apply_model(x9, t, cond, None)
        a_next = ac[int(step)]
        cur = max(0, int(step) - (1000 // n))
        a_cur = ac[cur]
        x = (x9[:, :4] - (1 - a_cur).sqrt() * e_t) * a_next.sqrt() / a_cur.sqrt() + (1 - a_next).sqrt() * e_t
        out[int(step)] = x[:batch_size].clone()
    return x, out
