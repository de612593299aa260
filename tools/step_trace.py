#!/usr/bin/env python3
"""Launch-by-launch trace of ONE hooked-UNet forward (the F-frame batch of bench.py's default workload), kernel by kernel with
HIP events around every C-ABI call: order, shape, microseconds, TFLOP/s (GEMM / conv / attention) or GB/s (the rest, by the
bytes of the tensors handed over).  Where the step's time goes, call by call -- what by_family in bench.py aggregates.

usage (GPU box): python tools/step_trace.py [--frames 8] [--fusion replace] [--reps 5] > gpurun_out/step_trace.txt
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--fusion", default="replace")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--reps", type=int, default=5)
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
    eng.use_graph = False
    eng.decompose_attn1 = True
    F_, h = a.frames, a.res // 8
    sampler.flow_gate = "reference" if h == 64 else "flow_hw"
    sampler.hook_plan = HookPlan(fusion=a.fusion, enabled=a.fusion != "none")
    stack = lambda s_, shape: torch.stack([synth.synth_normal(f"bench.{s_}.{f}", shape) for f in range(F_)]).to(dev)
    x_T = stack("xT", (4, h, h))
    c, uc, tc = stack("c", (1, 768)), stack("uc", (1, 768)), stack("tc", (1, 768))
    inp = stack("inp", (4, h, h)) * 0.18215
    mask = synth.synth_mask(F_, h, h).to(dev)
    steps = [int(s) for s in sampler.ddim_timesteps[::-1]]
    inv = {s_: stack(f"inv{s_}", (4, h, h)) for s_ in steps}
    flow = synth.synth_flow(F_ - 1, h, h).to(dev) if a.fusion == "flow_fix" else None
    kw = {"inpaint_image": inp, "inpaint_mask": mask}

    def one_step(img, i):
        s_ = steps[i % len(steps)]
        sampler._register_step_hooks(flow)
        ts = torch.full((F_,), s_, device=dev, dtype=torch.long)
        img, _ = sampler.p_sample_ddim_with_inverse(img, c, ts, index=len(steps) - 1 - (i % len(steps)), target_conditioning=tc,
                                                    inverse_results_dir=inv, unconditional_guidance_scale=3.0, flow=flow,
                                                    unconditional_conditioning=uc, test_model_kwargs=kw)
        return img

    names = ["gemm", "conv3x3", "conv3x3_plus_1x1", "upsample2x_conv3x3", "attention", "layernorm", "groupnorm_apply",
             "groupnorm_stats_from_cols", "groupnorm_stats", "ffn_fused", "attn_out_ffn_fused", "attn_out_ffn_proj_fused", "st_front", "gn_silu_conv3x3_small", "linear_small", "flow_warp", "silu", "cast_f32", "timestep_embedding",
             "pack_unet_input", "ddim_step", "groupnorm_coeffs_from_cols", "nchw_to_nhwc", "nhwc_to_nchw_f32", "copy2d"]
    rec, on = [], [False]

    def wrap(name):
        orig = getattr(hip, name, None)
        if orig is None:
            return

        def f(*args, **kwargs):
            if not on[0]:
                return orig(*args, **kwargs)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*args, **kwargs)
            e1.record()
            tens = [t for t in list(args) + list(kwargs.values()) if torch.is_tensor(t)]
            ints = {k: v for k, v in kwargs.items() if isinstance(v, (int, bool)) and not isinstance(v, torch.Tensor)}
            extra = [k for k, v in kwargs.items() if torch.is_tensor(v)]
            rec.append((name, ints, extra, e0, e1, args, kwargs))
            return r
        setattr(hip, name, f)
    for n in names:
        wrap(n)

    def flops_bytes(name, ints, args, kwargs):
        g = ints.get
        if name == "gemm":
            fl = 2.0 * g("M") * g("N") * g("K")
            by = g("M") * g("K") * 2 + (g("M") * g("N") * 2 if args[2] is not None else 0)
            for k in ("residual32", "out32"):
                if kwargs.get(k) is not None:
                    by += g("M") * g("N") * 4 // (2 if (ints.get("flags", 0) & 1) else 1)
            if kwargs.get("residual") is not None:
                by += g("M") * g("N") * 2
            return fl, by
        if name in ("conv3x3", "conv3x3_plus_1x1", "upsample2x_conv3x3"):
            H, W, s, up = g("H"), g("W"), g("stride", 1), g("upsample", False)
            if name == "upsample2x_conv3x3":
                OH, OW, taps = 2 * H, 2 * W, 4
            else:
                VH, VW = (2 * H, 2 * W) if up else (H, W)
                OH, OW, taps = (VH - 1) // s + 1, (VW - 1) // s + 1, 9
            Mo = g("nimg") * OH * OW
            fl = 2.0 * Mo * g("cout") * (taps * g("cin") + g("c2", 0))
            by = g("nimg") * H * W * (g("cin") + g("c2", 0)) * 2 + Mo * g("cout") * 2
            if kwargs.get("out32") is not None:
                by += Mo * g("cout") * 4
            if kwargs.get("residual32") is not None:
                by += Mo * g("cout") * 4
            return fl, by
        if name == "attention":
            vs = max(g("v_sets", 1), 1)
            fl = 4.0 * g("n") * g("nk") * g("dh") * g("heads") * g("B") * vs
            by = g("B") * g("heads") * g("dh") * 2 * (g("n") * (1 + vs) + g("nk") * (1 + vs))
            return fl, by
        if name == "st_front":
            M, C, NQ, rf, lo = g("M"), g("C_"), g("NQ"), g("rows_full"), g("nq_lo", 0)
            return 2.0 * M * C * C + 2.0 * C * (rf * NQ + (M - rf) * (NQ - lo)), M * C * 8 + (rf * NQ + (M - rf) * (NQ - lo)) * 2
        if name == "attn_out_ffn_proj_fused":
            return 28.0 * g("M") * g("C_") ** 2, g("M") * g("C_") * 14
        if name == "attn_out_ffn_fused":
            return 26.0 * g("M") * g("C_") ** 2, g("M") * g("C_") * 8
        if name == "ffn_fused":
            return 24.0 * g("M") * g("C_") ** 2, g("M") * g("C_") * 6
        by = 0
        for t in list(args) + list(kwargs.values()):
            if torch.is_tensor(t) and t.numel() > 4096:
                by += t.numel() * t.element_size() if t.is_contiguous() else t.shape[0] * t.shape[-1] * t.element_size()
        return 0.0, by

    with torch.no_grad():
        img = x_T
        for i in range(2):
            img = one_step(img, i)
        torch.cuda.synchronize()
        acc = None
        for rep in range(a.reps):
            rec.clear()
            on[0] = True
            img = one_step(x_T, rep)
            on[0] = False
            torch.cuda.synchronize()
            us = [e0.elapsed_time(e1) * 1e3 for (_, _, _, e0, e1, _, _) in rec]
            acc = us if acc is None else [min(p, q) for p, q in zip(acc, us)]
    tot = sum(acc)
    print(f"# {len(rec)} C-ABI calls per step, {tot / 1e3:.3f} ms of event-bracketed time (min over {a.reps} reps), "
          f"F={F_} fusion={a.fusion} res={a.res}")
    print(f"{'#':>4s} {'call':26s} {'us':>8s} {'TFLOP/s':>8s} {'GB/s':>7s}  shape")
    agg = {}
    for i, ((name, ints, extra, _, _, args, kwargs), u) in enumerate(zip(rec, acc)):
        fl, by = flops_bytes(name, ints, args, kwargs)
        shape = " ".join(f"{k}={int(v)}" for k, v in ints.items() if k not in ("lda", "ldc", "ldw", "ldx", "ldy", "ldo", "ldr",
                                                                               "ldq", "ldk", "ldv", "bsq", "bsk", "bsv", "bso",
                                                                               "lda2", "ldx2", "split_k"))
        print(f"{i:4d} {name:26s} {u:8.1f} {fl / u / 1e6 if fl else 0:8.0f} {by / u / 1e3 if by else 0:7.0f}  {shape} {'+' + ','.join(extra) if extra else ''}")
        k = (name, shape.split(" flags")[0] if name != "gemm" else f"M={ints.get('M')} N={ints.get('N')} K={ints.get('K')} "
             f"{'geglu ' if ints.get('flags', 0) & 1 else ''}{'+' + ','.join(e for e in extra if e in ('residual32', 'out32', 'residual', 'colstats', 'a2'))}")
        d = agg.setdefault(k, [0, 0.0, 0.0, 0.0])
        d[0] += 1; d[1] += u; d[2] += fl; d[3] += by
    print("\n# aggregated by (call, shape), sorted by total time")
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{d[1]:9.1f} us  x{d[0]:3d}  {d[1] / d[0]:8.1f} us/call  {d[2] / d[1] / 1e6 if d[2] else 0:6.0f} TFLOP/s {d[3] / d[1] / 1e3:6.0f} GB/s  {k[0]} {k[1]}")


if __name__ == "__main__":
    main()
