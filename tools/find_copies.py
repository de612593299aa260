#!/usr/bin/env python3
"""Diagnostic: which Python call sites issue the per-step device copies (aten::copy_ / cat) of a DDIM step."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vface_amd.utils import synth
from vface_amd.ldm.models.diffusion.ddpm import LatentDiffusion, FFHQ_UNET_CONFIG
from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler, HookPlan

dev = "cuda"
ldm = LatentDiffusion(dict(FFHQ_UNET_CONFIG, compute_dtype=torch.float16))
synth.fill_module_(ldm.unet, seed=0)
ldm = ldm.to(dev)
sampler = DDIMSampler(ldm)
sampler.make_schedule(50, ddim_eta=0.0, verbose=False)
sampler.hook_plan = HookPlan(fusion="replace")
F_, h = 2, 64
st = lambda s_, shape: torch.stack([synth.synth_normal(f"fc.{s_}.{f}", shape) for f in range(F_)]).to(dev)
x = st("x", (4, h, h)); c, uc, tc = st("c", (1, 768)), st("uc", (1, 768)), st("tc", (1, 768))
inp, mask = st("inp", (4, h, h)), synth.synth_mask(F_, h, h).to(dev)
steps = [int(s) for s in sampler.ddim_timesteps[::-1]]
inv = {s_: st(f"inv{s_}", (4, h, h)) for s_ in steps}
def one(i):
    sampler._register_step_hooks(None)
    ts = torch.full((F_,), steps[i], device=dev, dtype=torch.long)
    return sampler.p_sample_ddim_with_inverse(x, c, ts, index=len(steps) - 1 - i, target_conditioning=tc, inverse_results_dir=inv,
                                              unconditional_guidance_scale=3.0, flow=None, unconditional_conditioning=uc,
                                              test_model_kwargs={"inpaint_image": inp, "inpaint_mask": mask})
with torch.no_grad():
    one(0); one(1)
    sites = collections.Counter()
    orig_copy, orig_cat, orig_to, orig_contig, orig_clone = torch.Tensor.copy_, torch.cat, torch.Tensor.to, torch.Tensor.contiguous, torch.Tensor.clone
    def site():
        for fr in reversed(traceback.extract_stack()[:-2]):
            if "vface_amd" in fr.filename or "bench" in fr.filename:
                return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line}"
        return "?"
    def wrap(name, fn, real_copy=lambda a, k, out: True):
        def f(*a, **k):
            out = fn(*a, **k)
            if real_copy(a, k, out):
                sites[(name, site())] += 1
            return out
        return f
    torch.Tensor.copy_ = wrap("copy_", orig_copy)
    torch.cat = wrap("cat", orig_cat)
    torch.Tensor.to = wrap("to", orig_to, lambda a, k, out: out.data_ptr() != a[0].data_ptr())
    torch.Tensor.contiguous = wrap("contiguous", orig_contig, lambda a, k, out: out.data_ptr() != a[0].data_ptr())
    torch.Tensor.clone = wrap("clone", orig_clone)
    one(2)
    torch.cuda.synchronize()
for (name, s_), n in sites.most_common(40):
    print(f"{n:4d} {name:10s} {s_}")
