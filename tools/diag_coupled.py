#!/usr/bin/env python3
"""flow_fix on two coupled launch streams (parallel.StreamShard) against one launch sequence: final latents after k DDIM steps,
per-frame difference pattern.  usage (GPU box): python tools/diag_coupled.py [--frames 16] [--steps 1 3] [--reps 2]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def pattern(a, b):
    d = (a.double() - b.double()).abs()
    per = [(i, int((d[i] > 0).sum()), float(d[i].max())) for i in range(a.shape[0]) if bool((d[i] > 0).any())]
    return "; ".join(f"f{i}: {n} el, max {m:.2e}" for i, n, m in per) or "equal"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--steps", type=int, nargs="+", default=[1, 3])
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--res", type=int, default=512)
    a = ap.parse_args()
    from vface_amd import hip
    from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler, HookPlan
    from vface_amd.ldm.models.diffusion.ddpm import FFHQ_UNET_CONFIG, LatentDiffusion
    from vface_amd.utils import synth
    dev = torch.device("cuda", 0)
    hip.load()
    ldm = LatentDiffusion(dict(FFHQ_UNET_CONFIG, compute_dtype=torch.float16))
    synth.fill_module_(ldm.unet, seed=0)
    ldm = ldm.to(dev)
    sampler = DDIMSampler(ldm)
    sampler.make_schedule(50, ddim_eta=0.0, verbose=False)
    eng = ldm.unet.engine
    F_, h = a.frames, a.res // 8
    sampler.flow_gate = "reference" if h == 64 else "flow_hw"
    sampler.hook_plan = HookPlan(fusion="flow_fix")
    stack = lambda s_, shape: torch.stack([synth.synth_normal(f"bench.{s_}.{f}", shape) for f in range(F_)]).to(dev)
    x_T = stack("xT", (4, h, h))
    c, uc, tc = stack("c", (1, 768)), stack("uc", (1, 768)), stack("tc", (1, 768))
    inp = stack("inp", (4, h, h)) * 0.18215
    mask = synth.synth_mask(F_, h, h).to(dev)
    steps = [int(s) for s in sampler.ddim_timesteps[::-1]]
    inv = {s_: stack(f"inv{s_}", (4, h, h)) for s_ in steps}
    flow = synth.synth_flow(F_ - 1, h, h).to(dev)

    def run(nsteps, streams):
        eng.split_streams = streams
        with torch.no_grad():
            img, _ = sampler.sample(S=50, batch_size=F_, shape=[4, h, h], conditioning=c, target_conditioning=tc, inverse_results_dir=inv,
                                    verbose=False, unconditional_guidance_scale=3.0, unconditional_conditioning=uc, eta=0.0, x_T=x_T,
                                    flow=flow, test_model_kwargs={"inpaint_image": inp, "inpaint_mask": mask}, max_steps=nsteps)
        torch.cuda.synchronize()
        return img.clone()

    for ns in a.steps:
        ref = run(ns, 1)
        ref2 = run(ns, 1)
        print(f"steps={ns}: one sequence twice: {pattern(ref2, ref)}", flush=True)
        for r in range(a.reps):
            got = run(ns, 2)
            print(f"steps={ns} rep {r}: two coupled streams vs one: {pattern(got, ref)}", flush=True)


if __name__ == "__main__":
    main()
