#!/usr/bin/env python3
"""Feasibility probe: does running the DDIM step's UNet forward as TWO half-batches (frames 0..F/2-1 and F/2..F-1 of every chunk) on
two HIP streams at once beat one full-batch forward?  (One workgroup per CU kernels run their HBM phases in lock-step; two
independent launch streams interleave MFMA-bound and HBM-bound kernels.)  Times hipGraph replays:
  full        one graph, 3F samples
  halves_seq  two graphs of 3F/2 samples, one stream, back to back
  halves_par  the same two graphs on two streams, concurrently
usage (GPU box): python tools/two_stream_probe.py [--frames 8] [--fusion replace]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--fusion", default="replace")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cut", type=int, default=0, help="also time an UNEVEN two-way split: frames [0, cut) and [cut, F)")
    ap.add_argument("--sweep", action="store_true", help="the engine's own two-stream path (split_streams 1 vs 2) over batch sizes, "
                                                         "unhooked and with the chosen fusion")
    a = ap.parse_args()
    from vface_amd import hip
    from vface_amd.engine import Act
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
    F_, h = a.frames, 64
    sampler.hook_plan = HookPlan(fusion=a.fusion, enabled=a.fusion != "none")
    stack = lambda s_, shape: torch.stack([synth.synth_normal(f"bench.{s_}.{f}", shape) for f in range(F_)]).to(dev)
    x_T = stack("xT", (4, h, h))
    c, uc, tc = stack("c", (1, 768)), stack("uc", (1, 768)), stack("tc", (1, 768))
    inp = stack("inp", (4, h, h)) * 0.18215
    mask = synth.synth_mask(F_, h, h).to(dev)
    steps = [int(s) for s in sampler.ddim_timesteps[::-1]]
    inv = {s_: stack(f"inv{s_}", (4, h, h)) for s_ in steps}
    kw = {"inpaint_image": inp, "inpaint_mask": mask}
    grabbed = {}
    orig = eng.step_forward_nhwc

    def grab(x, ts, ctx):
        grabbed["args"] = (x.t.clone(), x.N, x.H, x.W, ts.clone(), ctx.clone())
        return orig(x, ts, ctx)
    eng.step_forward_nhwc = grab
    with torch.no_grad():
        sampler._register_step_hooks(None)
        ts = torch.full((F_,), steps[0], device=dev, dtype=torch.long)
        sampler.p_sample_ddim_with_inverse(x_T, c, ts, index=len(steps) - 1, target_conditioning=tc, inverse_results_dir=inv,
                                           unconditional_guidance_scale=3.0, flow=None, unconditional_conditioning=uc, test_model_kwargs=kw)
    eng.step_forward_nhwc = orig
    xt, N, H, W, tsN, ctx = grabbed["args"]
    if a.sweep:
        def t_ms(fn):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / a.iters
        with torch.no_grad():
            for hooked in (False, True):
                sampler.hook_plan = HookPlan(fusion=a.fusion, enabled=hooked)
                sampler._register_step_hooks(None)
                for n in ((8, 12, 16, 24, 32, 48) if not hooked else (12, 24, 48)):
                    rep = (n + N - 1) // N
                    xs = xt.reshape(N, -1).repeat(rep, 1)[:n].reshape(n * H * W, -1).contiguous()
                    tn, cn = tsN.repeat(rep)[:n].contiguous(), ctx.repeat(rep, 1, 1)[:n].contiguous()
                    res = {}
                    for streams in (1, 2):
                        eng.split_streams, eng._graphs, eng._split_state = streams, {}, {}
                        out = eng.step_forward_nhwc(Act(xs, n, H, W), tn, cn).clone()
                        res[streams] = (t_ms(lambda: eng.step_forward_nhwc(Act(xs, n, H, W), tn, cn)), out, bool(eng._graph_failed),
                                        any(k[0] != "plan" for k in eng._split_state))
                    print(f"{'hooked ' + a.fusion if hooked else 'unhooked':16s} N={n:3d}: one sequence {res[1][0]:7.3f} ms   two streams {res[2][0]:7.3f} ms "
                          f"({res[2][0] / res[1][0]:.3f})  split={res[2][3]} equal={torch.equal(res[1][1], res[2][1])} graph_failed={res[1][2] or res[2][2]}", flush=True)
        return
    chunks = N // F_
    hw, C = H * W, xt.shape[1]
    def parts(k):
        out, per = [], F_ // k
        for j in range(k):
            idx = torch.tensor([ch * F_ + f for ch in range(chunks) for f in range(j * per, (j + 1) * per)], device=dev)
            xs = xt.reshape(N, hw * C).index_select(0, idx).reshape(len(idx) * hw, C).contiguous()
            out.append((Act(xs, len(idx), H, W), tsN.index_select(0, idx).contiguous(), ctx.index_select(0, idx).contiguous(), idx))
        return out
    halves = parts(2)
    uneven = None
    if a.cut:
        uneven = []
        for lo, hi in ((0, a.cut), (a.cut, F_)):
            idx = torch.tensor([ch * F_ + f for ch in range(chunks) for f in range(lo, hi)], device=dev)
            xs = xt.reshape(N, hw * C).index_select(0, idx).reshape(len(idx) * hw, C).contiguous()
            uneven.append((Act(xs, len(idx), H, W), tsN.index_select(0, idx).contiguous(), ctx.index_select(0, idx).contiguous(), idx))
    quarters = parts(4) if F_ % 4 == 0 else None
    full = (Act(xt, N, H, W), tsN, ctx)
    s0 = torch.cuda.current_stream()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ss = [torch.cuda.Stream() for _ in range(4)]

    def run_full():
        return eng.step_forward_nhwc(*full)

    def run_seq():
        return [eng.step_forward_nhwc(*hv[:3]).clone() for hv in halves]

    def run_par(pp=None, join=True):
        pp = halves if pp is None else pp
        outs = []
        for k, (s, hv) in enumerate(zip(ss, pp)):
            if join:
                s.wait_stream(s0)
            with torch.cuda.stream(s), hip.workspace_domain(k + 1):
                outs.append(eng.step_forward_nhwc(*hv[:3]))
        if join:
            for s in ss[:len(pp)]:
                s0.wait_stream(s)
        return outs

    def free_running(pp):
        # every part replays its own graph `iters` times on its own stream, no joins in between (what two independent DDIM loops do)
        run_par(pp); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in ss[:len(pp)]:
            s.wait_stream(s0)
        for _ in range(a.iters):
            run_par(pp, join=False)
        for s in ss[:len(pp)]:
            s0.wait_stream(s)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    def time_ms(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters
    def spin_overlap(pair):
        def timed(streams):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s0)
            for s in streams:
                s.wait_stream(s0)
                with torch.cuda.stream(s):
                    torch.cuda._sleep(300_000)
            for s in streams:
                s0.wait_stream(s)
            e1.record(s0)
            e1.synchronize()
            return e0.elapsed_time(e1)
        timed(pair)
        return timed(pair) / timed(pair[:1])
    print(f"spin kernels on the probe's first two streams: together / alone = {spin_overlap(ss[:2]):.2f} (1 = they overlap, 2 = one queue)", flush=True)
    with torch.no_grad():
        ref = run_full().clone()
        t_full = time_ms(run_full)
        seq = run_seq()
        t_seq = time_ms(run_seq)
        par = [o.clone() for o in run_par()]
        t_par = time_ms(run_par)
        ok_seq = all(torch.equal(o, ref.reshape(N, -1).index_select(0, hv[3]).reshape(o.shape)) for o, hv in zip(seq, halves))
        ok_par = all(torch.equal(o, ref.reshape(N, -1).index_select(0, hv[3]).reshape(o.shape)) for o, hv in zip(par, halves))
        t_free = free_running(halves)
        t_q = t_qf = float("nan")
        if quarters is not None:
            qo = [o.clone() for o in run_par(quarters)]
            ok_q = all(torch.equal(o, ref.reshape(N, -1).index_select(0, hv[3]).reshape(o.shape)) for o, hv in zip(qo, quarters))
            t_q = time_ms(lambda: run_par(quarters))
            t_qf = free_running(quarters)
        if uneven is not None:
            uo = [o.clone() for o in run_par(uneven)]
            ok_u = all(torch.equal(o, ref.reshape(N, -1).index_select(0, hv[3]).reshape(o.shape)) for o, hv in zip(uo, uneven))
            print(f"uneven split {a.cut}/{F_ - a.cut}: {time_ms(lambda: run_par(uneven)):.3f} ms (bit-identical: {ok_u})", flush=True)
        t_full2 = time_ms(run_full)
    print(f"F={F_} fusion={a.fusion}: full {t_full:.3f} / {t_full2:.3f} ms   halves_seq {t_seq:.3f} ms (bit-identical to full: {ok_seq})   "
          f"halves_par {t_par:.3f} ms (bit-identical: {ok_par})   halves free-running {t_free:.3f} ms   "
          f"quarters_par {t_q:.3f} ms (bit-identical: {ok_q if quarters is not None else None})   quarters free-running {t_qf:.3f} ms")


if __name__ == "__main__":
    main()
