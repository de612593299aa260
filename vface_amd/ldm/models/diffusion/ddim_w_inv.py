"""``DDIMSampler`` of the VFace path with the reference's public surface
(``REFace/ldm/models/diffusion/ddim_w_inv.py``): ``make_schedule`` (:155-184), ``sample`` (:186-252),
``ddim_sampling`` (:254-355), ``p_sample_ddim_with_inverse`` (:621-738), ``ddim_invert`` (:360-490).

What differs is where the work happens, not what is computed:

* the 3-way batch ``[uncond ; cond ; recon]`` (:654-662) is written by one packing kernel straight into the
  NHWC 16-bit layout the UNet's first convolution reads;
* guidance + x0 prediction + x_{t-1} (:666-700) is one fused kernel on the UNet's fp32 output;
* the per-step recon latents the reference ``torch.load``s from disk inside the loop (:22-26,628) may also be
  handed over as a mapping ``{timestep: tensor}`` and are then kept resident in HBM (a directory of
  ``ddim_latents_{t}.pt`` files still works);
* the hook schedule hard-coded at :289-305 (every step: all ``attn1`` off, then the input-block ``attn1`` on with
  ``fusion="flow_fix"``, split 0.8, alpha 0.8) is the default ``hook_plan`` and can be replaced, e.g. by
  ``HookPlan(fusion="replace")`` for structure injection only.

Exact dead-branch elimination, opt-in (``sampler.drop_dead_branches = True``; results are bit-identical, tests/test_unet_gpu.py):

* sampling: the *recon* third of the batch is a pure sink -- ``e_t_recon`` only feeds ``x_prev_recon`` (:667,703-707), which
  ``p_sample_ddim_with_inverse`` drops (:738 returns ``x_prev, pred_x0``), and every hook mode writes INTO chunk 2, never reads
  from it (pnp_utils.py:136-142,195-199,255-256) -- so the UNet runs on ``[uncond ; cond]`` (2F samples) and the per-step
  recon latents are not even loaded;
* inversion: the *source* half of the 2F batch is never used -- only ``nosie[:batch_size]`` is saved (:464-486) and the
  entry point re-loads ``x_noisy`` from the saved latents (VFace_inference_batch.py:531-543) -- so it runs on the target half.

Reference quirks kept on purpose: chunk 0 (the structure source) is the unconditional branch; the
recon branch guidance is ``e_r + s (e_r - e_u)`` (:667); the RNG is drawn even when ``eta == 0`` (:697,702).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, Mapping, Optional, Sequence, Union

import numpy as np
import torch

from .... import hip
from ....engine import Act, _dev_flow
from ...modules.diffusionmodules.util import make_ddim_sampling_parameters, make_ddim_timesteps, noise_like
from ..pnp_utils import register_spa_attn_injection


def load_ddim_latents_at_t(t, ddim_latents_path):
    """ddim_w_inv.py:22-26."""
    path = os.path.join(ddim_latents_path, f"ddim_latents_{t}.pt")
    assert os.path.exists(path), f"Missing latents at t {t} path {path}"
    return torch.load(path)


@dataclass
class HookPlan:
    """Which attn1 modules are switched on each step and how (the arguments of the second
    ``register_spa_attn_injection`` call at ddim_w_inv.py:305)."""
    fusion: str = "flow_fix"
    input_blocks: bool = True
    middle_block: bool = False
    output_blocks: bool = False
    block_indices: Optional[Sequence[int]] = tuple(range(9))
    split_ratio_fft: float = 0.8
    alpha: float = 0.8
    chunks: int = 3
    enabled: bool = True


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self.hook_plan = HookPlan()
        # what to do with a flow field at pixel resolution (what the reference's own script produces): None =
        # raise like the reference's warp_image does; "area" = bring it to the latent map with vface_flow_to_latent
        self.flow_resample = None
        # exact dead-branch elimination (module docstring): sampling without the recon third, inversion without the source half
        self.drop_dead_branches = False
        # chunks 0 and 1 of the batch this sampler assembles are the same x at the same t (:632-655: `x_in = cat([x, x, inv_t])`)
        # and only diverge at the first attn2: their common prefix runs once (UNetEngine._shared_block).  Exact for chunks 0 and 2;
        # chunk 1 under `fft` takes q0, k0 where the reference's FFT round trip returns them up to fp32 rounding.
        # VFACE_SHARE_PREFIX=0 / `share_prefix = False`: every chunk computed on its own (A/B switch)
        self.share_prefix = os.environ.get("VFACE_SHARE_PREFIX", "1") != "0"

    def register_buffer(self, name, attr):
        # the reference forces .to("cuda") here (:149-153); buffers follow the model's device instead
        if isinstance(attr, torch.Tensor) and attr.device != self.model.device:
            attr = attr.to(self.model.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        ac = self.model.alphas_cumprod
        assert ac.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        self.register_buffer('betas', self.model.betas.clone().detach().float())
        self.register_buffer('alphas_cumprod', ac.clone().detach().float())
        self._alphas_cumprod_host = ac.detach().float().cpu().numpy()   # scalars for the update kernels: no per-step sync
        self.register_buffer('alphas_cumprod_prev', self.model.alphas_cumprod_prev.clone().detach().float())
        sig, a, ap = make_ddim_sampling_parameters(alphacums=ac.detach().float().cpu().numpy(),
                                                   ddim_timesteps=self.ddim_timesteps, eta=ddim_eta, verbose=verbose)
        # host-side scalars: one value per step is handed to the update kernel
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = np.asarray(sig), np.asarray(a), np.asarray(ap)
        self.ddim_sqrt_one_minus_alphas = np.sqrt(1. - self.ddim_alphas)

    # ------------------------------------------------------------------ sampling
    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, target_conditioning=None, inverse_results_dir=None,
               callback=None, normals_sequence=None, img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None,
               temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None, verbose=True, flow=None,
               x_T=None, log_every_t=100, unconditional_guidance_scale=1., unconditional_conditioning=None,
               src_im=None, tar=None, **kwargs):
        if conditioning is not None and not isinstance(conditioning, dict) and conditioning.shape[0] != batch_size:
            print(f"Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}")
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        return self.ddim_sampling(conditioning, (batch_size, C, H, W), target_conditioning=target_conditioning,
                                  inverse_results_dir=inverse_results_dir, callback=callback,
                                  img_callback=img_callback, quantize_denoised=quantize_x0, mask=mask, x0=x0,
                                  ddim_use_original_steps=False, noise_dropout=noise_dropout, temperature=temperature,
                                  score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, x_T=x_T,
                                  flow=flow, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, **kwargs)

    def _register_step_hooks(self, flow):
        """ddim_w_inv.py:303,305."""
        hp = self.hook_plan
        register_spa_attn_injection(self, 1, switch_on=False, input_blocks=True, middle_block=True, output_blocks=True,
                                    attn_component="attn1", flow=flow, chunks=3, block_indices=list(range(9)),
                                    fusion="flow_fix", split_ratio_fft=0.8, alpha=0.8)
        if hp.enabled:
            register_spa_attn_injection(self, 1, switch_on=True, input_blocks=hp.input_blocks,
                                        middle_block=hp.middle_block, output_blocks=hp.output_blocks,
                                        attn_component="attn1", flow=flow, chunks=hp.chunks,
                                        block_indices=None if hp.block_indices is None else list(hp.block_indices),
                                        fusion=hp.fusion, split_ratio_fft=hp.split_ratio_fft, alpha=hp.alpha)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, target_conditioning=None, inverse_results_dir=None, x_T=None,
                      ddim_use_original_steps=False, callback=None, timesteps=None, quantize_denoised=False, mask=None,
                      x0=None, img_callback=None, log_every_t=100, temperature=1., noise_dropout=0.,
                      score_corrector=None, flow=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                      unconditional_conditioning=None, src_im=None, max_steps=None, **kwargs):
        state, steps = self._sampling_plan(cond, shape, target_conditioning=target_conditioning, inverse_results_dir=inverse_results_dir,
                                           x_T=x_T, ddim_use_original_steps=ddim_use_original_steps, callback=callback,
                                           quantize_denoised=quantize_denoised, mask=mask, x0=x0, img_callback=img_callback,
                                           log_every_t=log_every_t, temperature=temperature, noise_dropout=noise_dropout,
                                           score_corrector=score_corrector, flow=flow, unconditional_guidance_scale=unconditional_guidance_scale,
                                           unconditional_conditioning=unconditional_conditioning, max_steps=max_steps, **kwargs)
        for _ in steps:
            pass
        return state["img"], state["intermediates"]

    def _sampling_plan(self, cond, shape, target_conditioning=None, inverse_results_dir=None, x_T=None,
                       ddim_use_original_steps=False, callback=None, quantize_denoised=False, mask=None, x0=None,
                       img_callback=None, log_every_t=100, temperature=1., noise_dropout=0., score_corrector=None, flow=None,
                       unconditional_guidance_scale=1., unconditional_conditioning=None, max_steps=None, **kwargs):
        """The sampling loop (ddim_w_inv.py:254-355) as ``(state, generator)``: every ``next()`` enqueues ONE DDIM step (hook
        registration :303,305 included) and updates ``state["img"]``; ``ddim_sampling`` drives it to the end, and
        ``sample_while_inverting`` interleaves it with the next batch's inversion loop."""
        if ddim_use_original_steps or mask is not None or quantize_denoised or score_corrector is not None:
            raise NotImplementedError("original-step sampling / masks / quantisation / score correctors are not used "
                                      "by scripts/VFace_inference_batch.py")
        if target_conditioning is None:
            raise NotImplementedError("the VFace path always samples with the inverted-target branch (:313-328)")
        device = self.model.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        img = img.contiguous()
        timesteps = self.ddim_timesteps
        state = {"img": img, "intermediates": {'x_inter': [img], 'pred_x0': [img]}}
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        flow_dev = _dev_flow(flow, device)
        if flow_dev is not None and tuple(flow_dev.shape[-2:]) != tuple(shape[-2:]) and self.flow_resample == "area":
            fh, fw = flow_dev.shape[-2] // shape[-2], flow_dev.shape[-1] // shape[-1]
            if fh == fw and fh >= 1 and (fh * shape[-2], fw * shape[-1]) == tuple(flow_dev.shape[-2:]):
                flow_dev = hip.flow_to_latent(flow_dev, fh)
        if flow_dev is not None and tuple(flow_dev.shape[-2:]) != tuple(shape[-2:]):
            # the shipped script feeds 512x512 RAFT flow to a 64x64 map and dies in warp_image
            raise RuntimeError(f"The size of the flow field {tuple(flow_dev.shape[-2:])} must match the latent map "
                               f"{tuple(shape[-2:])} (temporal_flow.py:43: grid + flow)")
        self._register_step_hooks(flow_dev)  # the pre-loop registrations (:289,291) are overwritten before any UNet call

        def steps():
            img = state["img"]
            for i, step in enumerate(time_range):
                if max_steps is not None and i >= max_steps:
                    break
                self._register_step_hooks(flow_dev)
                index = total_steps - i - 1
                ts = torch.full((b,), int(step), device=device, dtype=torch.long)
                img, pred_x0 = self.p_sample_ddim_with_inverse(
                    img, cond, ts, index=index, target_conditioning=target_conditioning,
                    inverse_results_dir=inverse_results_dir, temperature=temperature, noise_dropout=noise_dropout,
                    unconditional_guidance_scale=unconditional_guidance_scale, flow=flow_dev,
                    unconditional_conditioning=unconditional_conditioning, **kwargs)
                if callback:
                    callback(i)
                if img_callback:
                    img_callback(pred_x0, i)
                if index % log_every_t == 0 or index == total_steps - 1:
                    state["intermediates"]['x_inter'].append(img)
                    state["intermediates"]['pred_x0'].append(pred_x0)
                state["img"] = img
                yield i
        return state, steps()

    def _inv_latent(self, t: int, src, device):
        if isinstance(src, Mapping):
            v = src[t]
        elif hasattr(src, "get_latent"):
            v = src.get_latent(t)
        else:
            v = load_ddim_latents_at_t(t, src)
        return v.to(device=device, dtype=torch.float32).contiguous()

    @torch.no_grad()
    def p_sample_ddim_with_inverse(self, x, c, t, index, target_conditioning=None, inverse_results_dir=None,
                                   repeat_noise=False, src_start=None, use_original_steps=False,
                                   quantize_denoised=False, temperature=1., noise_dropout=0., score_corrector=None,
                                   corrector_kwargs=None, unconditional_guidance_scale=1., flow=None,
                                   unconditional_conditioning=None, **kwargs):
        if 'test_model_kwargs' in kwargs:
            kw = kwargs['test_model_kwargs']
            inpaint, mask = kw['inpaint_image'], kw['inpaint_mask']
        elif 'rest' in kwargs:
            inpaint, mask = kwargs['rest'][:, :4], kwargs['rest'][:, 4:5]
        else:
            raise Exception("kwargs must contain either 'test_model_kwargs' or 'rest' key")
        if unconditional_conditioning is None or unconditional_guidance_scale == 1.:
            raise NotImplementedError("the VFace path samples with classifier-free guidance (scale 3.0, :654-667)")
        if src_start is not None or use_original_steps or quantize_denoised or score_corrector is not None:
            raise NotImplementedError("options unused by the VFace entry point")
        device = x.device
        F_, C, H, W = x.shape
        unet = self.model.model.diffusion_model
        eng = unet.engine
        # t[0] == ddim_timesteps[index] by construction of the loop (ddim_w_inv.py:299-300); reading it from the host
        # schedule avoids a device->host sync every step (the reference's t[0].item() stalls the launch queue)
        if hasattr(self, "ddim_timesteps") and 0 <= index < len(self.ddim_timesteps):
            t_host = int(self.ddim_timesteps[index])
        else:
            t_host = int(t[0].item()) if torch.is_tensor(t) else int(t)
        drop = bool(getattr(self, "drop_dead_branches", False))
        nb = 2 if drop else 3
        # (without the recon third its per-step latent is not needed at all: in the file-backed pipeline that is a disk read less)
        inv_t = None if drop else self._inv_latent(t_host, inverse_results_dir, device)
        f32 = lambda v: v.to(device=device, dtype=torch.float32).contiguous()
        x_in = torch.empty(nb * F_ * H * W, 16, dtype=eng.dtype, device=device)
        hip.pack_unet_input(f32(x), inv_t, f32(inpaint), f32(mask), x_in, F=F_, h=H, w=W, cpad=16)
        t_in = torch.cat([t] * nb)
        # [uncond ; cond ; recon] (:654-667).  The three parts are the same tensors at every step of a clip: build the
        # concatenation once so the UNet engine sees one context object (and keeps its context-only projections)
        parts = (unconditional_conditioning, c) + (() if drop else (target_conditioning,))
        cc = getattr(self, "_c_in_cache", None)
        if cc is not None and len(cc[0]) == len(parts) and all(a is b for a, b in zip(cc[0], parts)) and \
                cc[1] == tuple(p._version for p in parts):
            c_in = cc[2]
        else:
            c_in = torch.cat(list(parts), dim=0)
            self._c_in_cache = (parts, tuple(p._version for p in parts), c_in)
        saved_live, saved_share = eng.live_chunks, eng.share_prefix
        eng.live_chunks = 2 if drop else None      # the hooks still say chunks = 3: the batch holds the first two of them
        # x_in / t_in above ARE [x ; x ; inv_t] / [t ; t ; t]: chunks 0 and 1 are the same input up to the first attn2
        eng.share_prefix = bool(self.share_prefix)
        try:
            eps = eng.step_forward_nhwc(Act(x_in, nb * F_, H, W), t_in, c_in)  # fp32 [nb F*HW, 4]
        finally:
            eng.live_chunks, eng.share_prefix = saved_live, saved_share
        a_t, a_prev = float(self.ddim_alphas[index]), float(self.ddim_alphas_prev[index])
        sigma_t, s1m = float(self.ddim_sigmas[index]), float(self.ddim_sqrt_one_minus_alphas[index])
        noise = noise_like(x.shape, device, repeat_noise) * temperature  # drawn even when sigma_t == 0 (:697)
        noise_like(x.shape, device, repeat_noise)                        # the recon twin's draw (:702)
        if noise_dropout > 0.:
            raise NotImplementedError("noise_dropout is not used by the VFace entry point")
        x_prev = torch.empty_like(x, dtype=torch.float32)
        pred_x0 = torch.empty_like(x, dtype=torch.float32)
        hip.ddim_step(eps, f32(x), inv_t, x_prev, F=F_, C_=C, hw=H * W, lde=eps.stride(0),
                      scale=float(unconditional_guidance_scale), a_t=a_t, a_prev=a_prev, sigma_t=sigma_t,
                      sqrt_one_minus_at=s1m, pred_x0=pred_x0, noise=noise if sigma_t != 0.0 else None,
                      single_branch=2 if drop else False)
        return x_prev, pred_x0

    # ------------------------------------------------------------------ inversion
    @torch.no_grad()
    def ddim_invert(self, x, cond, S, shape, eta=0., unconditional_guidance_scale=1., unconditional_conditioning=None,
                    inverse_dir=None, batch_size=6, src_lm=None, tar_lm=None, max_steps=None, **kwargs):
        """ddim_w_inv.py:360-490: hooks off, batch 2F = [target ; source], no guidance (the entry point passes
        ``unconditional_conditioning=None``); stores the target half per step -- into ``inverse_dir`` if it is a
        path (``ddim_latents_{t}.pt``, as the reference) or into it if it is a dict (device resident)."""
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=False)
        state, steps = self._invert_plan(x, cond, shape, unconditional_guidance_scale=unconditional_guidance_scale,
                                         unconditional_conditioning=unconditional_conditioning, inverse_dir=inverse_dir,
                                         batch_size=batch_size, max_steps=max_steps, **kwargs)
        for _ in steps:
            pass
        return state["x"], state["intermediates"]

    def _invert_plan(self, x, cond, shape, unconditional_guidance_scale=1., unconditional_conditioning=None, inverse_dir=None,
                     batch_size=6, max_steps=None, interleaved=False, **kwargs):
        """The inversion loop as ``(state, generator)`` on the schedule ``make_schedule`` has set: the hooks are switched off once in
        front of the loop (:389), every ``next()`` enqueues ONE inversion step and stores the target half's latent.
        ``interleaved`` (``sample_while_inverting``): a sampling loop runs between this one's steps and switches its own hooks on, so
        they are switched off again at EVERY step."""
        if unconditional_conditioning is not None and unconditional_guidance_scale != 1.:
            raise NotImplementedError("guided inversion is not used by the VFace entry point (:540)")
        device = x.device
        if getattr(self, "drop_dead_branches", False) and x.shape[0] > batch_size:
            # only the first `batch_size` samples (the target half) are ever saved (:464-486) and nothing else of this loop is
            # used by the entry point: run those alone (hooks are off, the kernels are batch-invariant: the same bits)
            x, cond = x[:batch_size], cond[:batch_size]
            kw0 = kwargs.get('test_model_kwargs')
            if kw0 is not None:
                kwargs = dict(kwargs, test_model_kwargs={k: (v[:batch_size] if torch.is_tensor(v) else v) for k, v in kw0.items()})
        b = x.shape[0]
        timesteps = self.ddim_timesteps
        kw = kwargs.get('test_model_kwargs')
        if kw is None:
            raise Exception("ddim_invert needs test_model_kwargs (inpaint_image, inpaint_mask)")
        f32 = lambda v: v.to(device=device, dtype=torch.float32).contiguous()
        inpaint, mask = f32(kw['inpaint_image']), f32(kw['inpaint_mask'])
        unet = self.model.model.diffusion_model
        eng = unet.engine
        _, C, H, W = x.shape
        ac = self._alphas_cumprod_host
        state = {"x": f32(x), "intermediates": {'x_inter': [x]}}

        def hooks_off():
            register_spa_attn_injection(self, 1, switch_on=False, input_blocks=True, middle_block=True, output_blocks=True,
                                        attn_component="attn1", chunks=3)

        def steps():
            x = state["x"]
            if not interleaved:
                hooks_off()
            for i, step in enumerate(timesteps):
                if max_steps is not None and i >= max_steps:
                    break
                if interleaved:
                    hooks_off()
                ts = torch.full((b,), int(step), device=device, dtype=torch.long)
                x9 = torch.empty(b * H * W, 16, dtype=eng.dtype, device=device)
                hip.nchw_to_nhwc(torch.cat([x, inpaint, mask], 1).contiguous(), x9, N=b, C_=9, hw=H * W, cpad=16)
                eps = eng.step_forward_nhwc(Act(x9, b, H, W), ts, cond)
                a_next = float(ac[int(step)])
                cur = max(0, int(step) - (1000 // len(timesteps)))
                a_cur = float(ac[cur])
                # x <- (x - sqrt(1-a_cur) e) * sqrt(a_next)/sqrt(a_cur) + sqrt(1-a_next) e   (:449), as the DDIM update
                # kernel with a_t = a_cur, a_prev = a_next, scale = 0 on an eps laid out [e ; e ; e]
                x_new = torch.empty_like(x)
                hip.ddim_step(eps, x, None, x_new, F=b, C_=C, hw=H * W, lde=eps.stride(0), scale=0.0, a_t=a_cur,
                              a_prev=a_next, sigma_t=0.0, sqrt_one_minus_at=float(np.sqrt(np.float32(1.0) - np.float32(a_cur))),
                              single_branch=True)
                x = x_new
                state["intermediates"]['x_inter'].append(x)
                save = x[:batch_size].detach().clone()
                if isinstance(inverse_dir, dict):
                    inverse_dir[int(step)] = save
                elif inverse_dir is not None:
                    torch.save(save, os.path.join(inverse_dir, f"ddim_latents_{step}.pt"))
                state["x"] = x
                yield i
        return state, steps()

    # ------------------------------------------------------------------ inversion of the NEXT batch beside sampling of this one
    @torch.no_grad()
    def sample_while_inverting(self, sample_kwargs: dict, invert_kwargs: dict):
        """``sample(**sample_kwargs)`` of one batch and ``ddim_invert(**invert_kwargs)`` of the NEXT batch, step by step on two HIP
        streams at once.  The entry point's batches are independent (VFace_inference_batch.py:413, 529-553: a batch's inversion only
        feeds its own sampling), so the two 50-step loops of consecutive batches can share the chip: one launch sequence each
        (the engine's own two-stream split is off meanwhile), hooks switched per step (on for the sampling forward, off for the
        inversion forward), results bit-identical to running the loops one after the other (tests).  Returns
        ``((samples, intermediates), (x_inverted, intermediates))``."""
        sk, ik = dict(sample_kwargs), dict(invert_kwargs)
        S, eta = sk.pop("S"), sk.pop("eta", 0.)
        if ik.pop("S") != S or ik.pop("eta", 0.) != eta:
            raise ValueError("sample_while_inverting: both loops walk ONE schedule (the same S and eta)")
        # arguments `sample` / `ddim_invert` accept and do not act on (:186-252, :360-372) may be dropped; one that WOULD act is an
        # error here rather than a silent difference from the sequential order (`callback` / `img_callback` are passed on)
        for k, unused in (("normals_sequence", None), ("quantize_x0", False), ("corrector_kwargs", None), ("verbose", None), ("tar", None),
                          ("src_im", None)):
            v = sk.pop(k, unused)
            if k in ("quantize_x0",) and v:
                raise NotImplementedError(f"sample_while_inverting: {k}={v!r} is not part of the VFace path")
        for k in ("src_lm", "tar_lm"):
            ik.pop(k, None)
        batch_size, (C, H, W) = sk.pop("batch_size"), sk.pop("shape")
        ik.pop("shape", None)
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=False)
        st_s, g_s = self._sampling_plan(sk.pop("conditioning"), (batch_size, C, H, W), **sk)
        st_i, g_i = self._invert_plan(ik.pop("x"), ik.pop("cond"), None, interleaved=True, **ik)
        cur = torch.cuda.current_stream()
        eng = self.model.model.diffusion_model.engine
        if getattr(self, "_pipe_streams", None) is None:
            # a pair whose launches really overlap (HIP maps streams onto a few hardware queues; two streams on one queue run their
            # launches one after the other -- seen in round 6: 1.41 s instead of 1.07 s for a pipelined batch): the engine's probe,
            # a spin kernel on each stream timed together and alone, up to three pairs
            self._pipe_streams = tuple(eng._concurrent_stream_pair())
        saved_split = eng.split_streams
        eng.split_streams = 1          # the two loops ARE the two launch sequences
        for s in self._pipe_streams:
            s.wait_stream(cur)
        try:
            live = [True, True]
            while any(live):
                for n, (g, s) in enumerate(zip((g_s, g_i), self._pipe_streams)):
                    if live[n]:
                        # (its own split-K scratch: a scratch is used in order by the launches of ONE stream)
                        with torch.cuda.stream(s), hip.workspace_domain(3 + n):
                            try:
                                next(g)
                            except StopIteration:
                                live[n] = False
        finally:
            eng.split_streams = saved_split
            for s in self._pipe_streams:
                cur.wait_stream(s)
        for t in [st_s["img"], st_i["x"]] + (list(ik["inverse_dir"].values()) if isinstance(ik.get("inverse_dir"), dict) else []):
            t.record_stream(cur)       # (allocated on a side stream, consumed on the caller's from here on)
        return (st_s["img"], st_s["intermediates"]), (st_i["x"], st_i["intermediates"])
