#!/usr/bin/env python3
"""Benchmark of the VFace per-frame DDIM denoising hot path on MI355X.

A "step" is ONE DDIM step of the hot path over one batch of synthetic input: pack [uncond ; cond ; recon]
(3F samples) -> hooked UNet forward -> guidance + x_{t-1} update, for the F frames this rank owns.
`value` = swapped frames/s for the whole job at 50 DDIM steps per frame = (F * n_gpus) / (50 * s_per_step).

Default workload (N = 1): BASELINE.json configs[1] -- an 8-frame 512x512 clip (latent 64x64), 50-step DDIM
schedule, structure attention injection only (fusion "replace" on the input-block attn1 modules), synthetic
latents / conditioning / weights (name-keyed deterministic fill; there are no checkpoints on the GPU box).
Inputs are resident in HBM when the timed region starts.  With --gpus N every rank runs the same per-GPU
workload on its own frames (weak scaling); `--fusion flow_fix` adds the FSAI + flow path, whose one-neighbour
boundary exchange runs over RCCL.

Also reported on the same JSON line:
  roofline     -- the dominant kernel (implicit-GEMM 3x3 conv): algorithmic FLOPs per launch / mean launch
                  duration, timed with HIP events on the launch stream inside the timed region, against the
                  dense 16-bit MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md).
  cpu_baseline -- the CPU oracle (torch fp32 restatement of the reference, pinned to reference-generated
                  golden vectors) timed on this box's host cores on a bounded sample: F=1 (batch 3) UNet
                  forwards at 64x64 with the same hook mode, extrapolated to 50 steps per frame.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"


def log(msg):
    if int(os.environ.get("RANK", 0)) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=8, help="frames per GPU")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--fusion", default="replace", help="replace | fft | flow_fix | none")
    ap.add_argument("--inv-steps", type=int, default=3, help="DDIM-inversion steps timed after the run (0 = skip)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-forwards", type=int, default=3)
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1 only: skip the extra workloads (config 3: 32 frames + fft; shipped schedule: 16 frames + flow_fix)")
    ap.add_argument("--extra-steps", type=int, default=10)
    ap.add_argument("--eager", action="store_true",
                    help="timed region launches kernel by kernel (default: the UNet forward of a step replayed from a hipGraph, "
                         "falling back to kernel-by-kernel launches by itself where a graph cannot be captured)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the environment: start N ranks (one per GPU) with
    torch.distributed.run as a CHILD process and pass its exit code on.  This parent never touches the GPU (no HIP call,
    no torch.cuda.is_available()); rank 0 of the children prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"self-launch: {' '.join(cmd)}")
    return subprocess.call(cmd, env=env)


class ConvTimer:
    """HIP-event timing of every convolution launch inside the timed region, by kernel: `patch3` = conv_patch_kernel<3x3>
    (the dominant kernel of the step: conv.hip), `patch2` = its 2x2 parity-phase form (four launches per call), `patch8x8` =
    its 8x8 form (four images per workgroup; the time includes the split-K reduce pass), `im2col` = gemm.hip's implicit GEMM
    (stride 2, the 9->320 and 320->4 convolutions)."""

    def __init__(self):
        self.cls = {k: {"events": [], "flops": 0.0, "launches": 0} for k in ("patch3", "patch2", "patch8x8", "im2col")}
        self.on = False

    def _timed(self, key, flops, launches, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        c = self.cls[key]
        c["events"].append((e0, e1)); c["flops"] += flops; c["launches"] += launches

    def wrap(self, hip):
        timer = self
        orig = hip.conv3x3

        def conv3x3(x, wt, out, *, nimg, H, W, cin, cout, stride=1, upsample=False, **kw):
            call = lambda: orig(x, wt, out, nimg=nimg, H=H, W=W, cin=cin, cout=cout, stride=stride, upsample=upsample, **kw)
            if not timer.on:
                return call()
            VH, VW = (2 * H, 2 * W) if upsample else (H, W)
            OH, OW = (VH - 1) // stride + 1, (VW - 1) // stride + 1
            patch = 0 if (kw.get("flags", 0) & hip.EPI_OUT_F32) else hip.conv_uses_patch_kernel(H, W, cin, cout, 3, stride, upsample, kw.get("flags", 0))
            timer._timed({1: "patch3", 2: "patch8x8"}.get(patch, "im2col"), 2.0 * nimg * OH * OW * cout * 9 * cin, 1, call)
        hip.conv3x3 = conv3x3
        orig_up = hip.upsample2x_conv3x3

        def upsample2x_conv3x3(x, wt, out, *, nimg, H, W, cin, cout, **kw):
            # four parity-phase launches of a 2x2 window on the low-resolution input: FLOPs counted as executed (4 taps),
            # not as the 9-tap form they replace
            call = lambda: orig_up(x, wt, out, nimg=nimg, H=H, W=W, cin=cin, cout=cout, **kw)
            if not timer.on:
                return call()
            patch = hip.conv_uses_patch_kernel(H, W, cin, cout, 2, 1, False, kw.get("flags", 0)) == 1
            timer._timed("patch2" if patch else "im2col", 4 * 2.0 * nimg * H * W * cout * 4 * cin, 4, call)
        hip.upsample2x_conv3x3 = upsample2x_conv3x3
        orig_p1 = hip.conv3x3_plus_1x1

        def conv3x3_plus_1x1(x, x2, wt, out, *, nimg, H, W, cin, c2, cout, **kw):
            call = lambda: orig_p1(x, x2, wt, out, nimg=nimg, H=H, W=W, cin=cin, c2=c2, cout=cout, **kw)
            if not timer.on:
                return call()
            patch = hip.conv_uses_patch_kernel(H, W, cin, cout, 3, 1, False, kw.get("flags", 0)) if c2 % 64 == 0 else 0
            timer._timed({1: "patch3", 2: "patch8x8"}.get(patch, "im2col"), 2.0 * nimg * H * W * cout * (9 * cin + c2), 1, call)
        hip.conv3x3_plus_1x1 = conv3x3_plus_1x1

    def summary(self):
        out = {}
        for k, c in self.cls.items():
            ms = sum(a.elapsed_time(b) for a, b in c["events"])
            out[k] = {"launches": c["launches"], "ms": ms, "flops": c["flops"],
                      "tflops": c["flops"] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                      "mean_launch_us": ms * 1e3 / max(c["launches"], 1)}
        return out


def usable_cores():
    """Host cores this process may really use: the affinity mask, capped by the cgroup CPU quota when there is one and
    by the GPU box's documented share of 16 cores per GPU otherwise (its affinity mask shows all 256 host threads:
    256 torch threads on a 16-core share ran a forward in 177 s instead of ~5 s).  VFACE_CPU_THREADS overrides."""
    if os.environ.get("VFACE_CPU_THREADS"):
        return max(1, int(os.environ["VFACE_CPU_THREADS"]))
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, min(n, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return min(n, 16)


def cpu_baseline(n_forwards, ddim_steps):
    """Oracle (kind 'port') timed on the host cores, as BASELINE.md 4 plans it: full-size UNet, 64x64 latent, the shipped
    hook schedule (flow_fix on the input-block attn1), F = 2 frames (batch 6) so the flow warp really runs, 1 warm-up +
    `n_forwards` (>= 3) timed forwards on every core of the affinity mask."""
    from oracle import hooks as ohooks
    from oracle import unet as ounet
    from vface_amd.utils import synth
    cores = usable_cores()
    torch.set_num_threads(cores)
    n_forwards = max(3, n_forwards)
    log(f"cpu_baseline: oracle on {cores} threads ...")
    F_ = 2
    spec = ounet.UNetSpec()
    sd = synth.synth_state_dict(ounet.param_shapes(spec), seed=0)
    x = synth.synth_normal("bench.cpu.x", (3 * F_, 9, 64, 64))
    ctx = synth.synth_normal("bench.cpu.ctx", (3 * F_, 1, 768))
    t = torch.full((3 * F_,), 481, dtype=torch.long)
    flow = [f[None] for f in synth.synth_flow(F_ - 1, 64, 64)]
    reg = {}
    ohooks.register_spa_attn_injection(reg, ounet.attn1_names(spec), 1, switch_on=True, input_blocks=True,
                                       middle_block=False, output_blocks=False, chunks=3, flow=flow,
                                       block_indices=list(range(9)), fusion="flow_fix", split_ratio_fft=0.8, alpha=0.8)
    with torch.no_grad():
        t0 = time.time()
        ounet.unet_forward(sd, spec, x, t, ctx, reg)  # warm-up
        log(f"cpu_baseline: warm-up forward {time.time() - t0:.1f} s")
        t0 = time.time()
        for _ in range(n_forwards):
            ounet.unet_forward(sd, spec, x, t, ctx, reg)
        dt = (time.time() - t0) / n_forwards
    return {"value": F_ / (ddim_steps * dt), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_forwards} timed hooked-UNet forwards (+1 warm-up) of the CPU oracle, F={F_} (batch {3 * F_}), "
                      f"64x64 latent, fp32, fusion=flow_fix on the input-block attn1 (the shipped schedule); {dt:.2f} s per "
                      f"forward, x{ddim_steps} steps per clip of {F_} frames"}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback on the product path)"
    # VFACE_BENCH_REHEARSE=1: every rank on cuda:0 with the gloo backend -- to rehearse the N > 1 control flow on a
    # one-GPU box (the numbers of such a run mean nothing).  The driver's multi-GPU runs never set it.
    rehearse = os.environ.get("VFACE_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist, backend = None, None
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # "nccl" IS RCCL on ROCm
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
        backend = dist.get_backend()

    from vface_amd import hip
    from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler, HookPlan
    from vface_amd.ldm.models.diffusion.ddpm import FFHQ_UNET_CONFIG, LatentDiffusion
    from vface_amd.parallel import FrameShard
    from vface_amd.utils import synth

    hip.load()
    log(f"building the 859.5M-parameter UNet with synthetic weights (rank {rank}/{world}) ...")
    timer = ConvTimer()
    timer.wrap(hip)
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    h = a.res // 8
    ldm = LatentDiffusion(dict(FFHQ_UNET_CONFIG, compute_dtype=dt))
    synth.fill_module_(ldm.unet, seed=0)
    ldm = ldm.to(dev)
    sampler = DDIMSampler(ldm)
    if h != 64:
        sampler.flow_gate = "flow_hw"   # the reference's gate (4096 tokens) only ever fires at 512 x 512
    sampler.make_schedule(a.ddim_steps, ddim_eta=0.0, verbose=False)
    steps = [int(s) for s in sampler.ddim_timesteps[::-1]]

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def run_workload(F_, fusion, n_steps, n_warm, time_convs, inv_steps, graph=False):
        """W untimed + exactly K timed DDIM steps of an F-frame clip per GPU; returns per-rank wall seconds etc.
        graph=True: the UNet forward of a step replayed from a hipGraph (UNetEngine.step_forward_nhwc) -- no per-launch
        events can be recorded inside a graph, so the headline run (which carries the live roofline timing) launches kernel
        by kernel and the graph run is reported beside it."""
        sampler.hook_plan = HookPlan(fusion=fusion, enabled=fusion != "none")
        ldm.unet.engine.use_graph, ldm.unet.engine._graphs = bool(graph), {}
        shard = FrameShard(rank, world, F_ * world, dist)
        g0 = shard.first  # global index of this rank's first frame
        tag = lambda s_, f: f"bench.{s_}.{g0 + f}"
        stack = lambda s_, shape: torch.stack([synth.synth_normal(tag(s_, f), shape) for f in range(F_)]).to(dev)
        x_T = stack("xT", (4, h, h))
        c, uc, tc = stack("c", (1, 768)), stack("uc", (1, 768)), stack("tc", (1, 768))
        inp = stack("inp", (4, h, h)) * 0.18215
        mask = synth.synth_mask(F_, h, h).to(dev)
        inv = {s_: stack(f"inv{s_}", (4, h, h)) for s_ in steps}  # device-resident recon latents
        flow = None
        eng = ldm.unet.engine
        eng.halo_exchange, eng.halo_flow = None, None
        if fusion == "flow_fix":
            gflow = synth.synth_flow(F_ * world - 1, h, h)  # one field per consecutive global frame pair
            flow = shard.local_flow(gflow).to(dev)
            shard.install(eng, gflow, dev)
        kw = {"inpaint_image": inp, "inpaint_mask": mask}

        def one_step(img, i):
            s_ = steps[i % len(steps)]
            sampler._register_step_hooks(flow)
            ts = torch.full((F_,), s_, device=dev, dtype=torch.long)
            img, _ = sampler.p_sample_ddim_with_inverse(img, c, ts, index=len(steps) - 1 - (i % len(steps)),
                                                        target_conditioning=tc, inverse_results_dir=inv,
                                                        unconditional_guidance_scale=3.0, flow=flow,
                                                        unconditional_conditioning=uc, test_model_kwargs=kw)
            return img

        with torch.no_grad():
            img = x_T
            for i in range(n_warm):
                img = one_step(img, i)
            img = x_T
            fence()
            timer.on = time_convs
            t0 = time.perf_counter()
            for i in range(n_steps):
                img = one_step(img, i)
            t_enq = time.perf_counter() - t0      # host time to enqueue the steps (the GPU runs behind)
            fence()
            el = time.perf_counter() - t0
            timer.on = False
        assert torch.isfinite(img).all(), "non-finite latents"
        # DDIM inversion (ddim_w_inv.py:360-490; SURVEY 8d asks for it separately): hooks off, batch 2F = [target ; source],
        # no guidance -- 50 such steps per clip precede sampling unless the latents are cached.  Timed outside the step loop.
        inv_ms = None
        if inv_steps > 0:
            with torch.no_grad():
                x2 = torch.cat([x_T, stack("xsrc", (4, h, h))])
                c2 = torch.cat([c, tc])
                kw2 = {"inpaint_image": torch.cat([inp, inp]), "inpaint_mask": torch.cat([mask, mask])}
                store = {}
                sampler.ddim_invert(x2, c2, a.ddim_steps, (4, h, h), inverse_dir=store, batch_size=F_, max_steps=1,
                                    test_model_kwargs=kw2)
                fence()
                t1 = time.perf_counter()
                sampler.ddim_invert(x2, c2, a.ddim_steps, (4, h, h), inverse_dir=store, batch_size=F_, max_steps=inv_steps,
                                    test_model_kwargs=kw2)
                fence()
                inv_ms = (time.perf_counter() - t1) / inv_steps * 1e3
            sampler.make_schedule(a.ddim_steps, ddim_eta=0.0, verbose=False)
        if dist is not None:
            tt = torch.tensor([el], device="cpu" if rehearse else dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return {"ms_step": el / n_steps * 1e3, "enqueue_ms": t_enq / n_steps * 1e3, "inv_ms": inv_ms, "elapsed": el}

    F_ = a.frames
    log("weights resident; warm-up ...")
    # Timed region: W + K DDIM steps with NO instrumentation inside -- the UNet forward of a step replayed from a hipGraph
    # (one host call instead of ~1200; VERDICT r1 #9/#10: no per-launch event records in the region `value` is timed over).
    # The per-launch HIP events behind `roofline` are recorded in a second, kernel-by-kernel pass of the same W + K steps
    # right after it (events cannot be recorded inside a captured graph); that pass's own step time is reported beside.
    r = run_workload(F_, a.fusion, a.steps, a.warmup, False, a.inv_steps, graph=not a.eager)
    graphed = ldm.unet.engine.use_graph and len(ldm.unet.engine._graphs) > 0
    launch_mode = ("hipGraph replay of the UNet forward of each step (UNetEngine.step_forward_nhwc)" if graphed else
                   "kernel by kernel" + ("" if a.eager else " (no graph captured: frame-sharded exchange inside the forward, or capture failed)"))
    ms_step, inv_ms = r["ms_step"], r["inv_ms"]
    if inv_ms is not None:
        log(f"inversion: {inv_ms:.2f} ms/step (2F = {2 * F_} unhooked sample-forwards)")
    log(f"timed {a.steps} steps: {ms_step:.2f} ms/step (host enqueue {r['enqueue_ms']:.2f} ms/step; {launch_mode})")
    log("instrumented pass (kernel by kernel, HIP events around every convolution launch) ...")
    ri = run_workload(F_, a.fusion, a.steps, a.warmup, True, 0, graph=False)
    el = ri["elapsed"]
    log(f"  {ri['ms_step']:.2f} ms/step (host enqueue {ri['enqueue_ms']:.2f} ms/step)")
    fps = (F_ * world) / (a.ddim_steps * ms_step / 1e3)
    conv = timer.summary()
    conv_ms = sum(c["ms"] for c in conv.values())
    conv_flops = sum(c["flops"] for c in conv.values())
    unet_gflop = {64: 796.94, 96: 2137.52, 32: 176.34}.get(h)   # BASELINE.md 2, per sample-forward

    # the other single-GPU BASELINE configurations, same process and build, outside the timed region of the headline run
    extras = []
    if world == 1 and not a.no_extras and h == 64:
        for name, f2, fus in (("BASELINE configs[2]: 32-frame 512x512 clip + frequency-spectrum attention interpolation", 32, "fft"),
                              ("shipped hook schedule (ddim_w_inv.py:303-305), config 4's per-GPU share: 16 frames + flow_fix", 16, "flow_fix")):
            if f2 == F_ and fus == a.fusion:
                continue
            log(f"extra workload: {f2} frames, fusion={fus} ...")
            e = run_workload(f2, fus, a.extra_steps, 2, False, 0, graph=not a.eager)
            eg = ldm.unet.engine.use_graph and len(ldm.unet.engine._graphs) > 0
            extras.append({"workload": name, "frames_per_gpu": f2, "fusion": fus, "steps": a.extra_steps, "warmup": 2,
                           "launch": "hipGraph replay" if eg else "kernel by kernel",
                           "ms_per_step": e["ms_step"], "host_enqueue_ms_per_step": e["enqueue_ms"],
                           "frames_per_s": f2 / (a.ddim_steps * e["ms_step"] / 1e3),
                           "unet_algorithmic_tflops": 3 * f2 * unet_gflop * 1e9 / (e["ms_step"] * 1e-3) / 1e12})
            log(f"  {e['ms_step']:.2f} ms/step = {extras[-1]['frames_per_s']:.2f} frames/s")
    ldm.unet.engine.use_graph, ldm.unet.engine._graphs = False, {}     # (frees the captured graphs' activation pools)
    if rank == 0:
        dom = conv["patch3"] if conv["patch3"]["launches"] else max(conv.values(), key=lambda c: c["ms"])
        achieved = dom["tflops"]
        all_conv = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        unet_tflops = 3 * F_ * unet_gflop * 1e9 / (ms_step * 1e-3) / 1e12 if unet_gflop else None
        # HBM-side bytes per conv launch are NOT measured in this run: they come from separate rocprofv3 --pmc passes of this
        # same command (tools/traffic.sh: 2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md "HBM");
        # the committed summary is quoted only for the workload it was collected on, otherwise null.
        traffic, traffic_src = None, None
        prof = os.path.join(ROOT, "profiles")
        cands = sorted(f for f in os.listdir(prof) if f.endswith("_hbm_traffic.json")) if os.path.isdir(prof) else []
        if cands and F_ == 8 and h == 64 and a.fusion == "replace" and a.dtype == "fp16":
            tf = os.path.join(prof, cands[-1])
            tj = json.load(open(tf))
            key = "_conv_patch3" if "_conv_patch3" in tj else next((k for k in tj if k.startswith("conv_patch_kernel") and ", 3, 3" in k), "_conv_all")
            traffic = tj[key]["hbm_bytes_per_launch"]
            traffic_src = (f"QUOTED from profiles/{cands[-1]} [{key}] (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                           "command on the build it names), not measured in this run")
        out = {
            "metric": "swapped frames/sec at 512x512, 50-step DDIM", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if a.dtype == "fp16" else "bf16",
            "data": "synthetic",
            "config": {"workload": f"{F_}-frame {a.res}x{a.res} clip per GPU, {a.ddim_steps}-step DDIM, hooked REFace "
                                   f"UNet (859.5M params), attn1 fusion={a.fusion} on input blocks, CFG scale 3.0, "
                                   f"batch [uncond;cond;recon] = {3 * F_} samples per step",
                       "frames_per_gpu": F_, "latent": [h, h], "fusion": a.fusion,
                       "world_size": world, "backend": backend, "launch": launch_mode,
                       "unet_algorithmic_tflops_per_gpu": unet_tflops,
                       "host_enqueue_ms_per_step": r["enqueue_ms"],
                       # north_star also asks for the rate as a fraction of the attention-GEMM roofline: the attn1 QKV
                       # projections + QK^T + PV are 160.9 GFLOP per sample-forward at 64x64 = 24.1 TFLOP per
                       # swapped frame; at the 2.5 PFLOP/s dense peak that alone would allow 103.6 frames/s per GPU
                       "attention_gemm_roofline_frac": (fps / world) * 24.135e12 / (MFMA_PEAK_TFLOPS * 1e12) if h == 64 else None},
            "inversion": None if inv_ms is None else {
                "ms_per_step": inv_ms, "steps_timed": a.inv_steps,
                "note": "DDIM inversion step (hooks off, batch 2F, no guidance), outside the timed region; `value` is "
                        "sampling only, as BASELINE's metric",
                "frames_per_s_sampling_plus_inversion": (F_ * world) / (a.ddim_steps * (ms_step + inv_ms) / 1e3)},
            "extra": extras,
            "instrumented_pass": {"launch": "kernel by kernel, HIP events around every convolution launch (what `roofline` is computed from)",
                                  "ms_per_step": ri["ms_step"], "host_enqueue_ms_per_step": ri["enqueue_ms"], "steps": a.steps, "warmup": a.warmup},
            # the dominant kernel of the step (largest share of kernel time in profiles/*_kernel_stats.csv): conv_patch_kernel<3,3>,
            # the patch-staged stride-1 3x3 convolution (incl. the launches that carry a ResBlock's fused 1x1 shortcut)
            "roofline": {"bound": "mfma", "kernel": "conv_patch_kernel<T, NT, 3, 3> (conv.hip: patch-staged 3x3 convolution)",
                         "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "launches": dom["launches"], "mean_launch_us": dom["mean_launch_us"],
                         "share_of_step_time": dom["ms"] / (el * 1e3),
                         "timing": "HIP events on the launch stream around every launch of this kernel, recorded in the instrumented pass: the same "
                                   "W + K steps launched kernel by kernel right after the timed region (events cannot be recorded inside a "
                                   "captured graph); share_of_step_time is relative to that pass",
                         "note": "peak is the datasheet 2.5 PFLOP/s; an MFMA-only loop of this tile shape sustains 1.5 PFLOP/s on "
                                 "this chip (it clocks ~1.6 GHz under matrix load: DESIGN.md 4, profiles/r02_a_conv_patch_ablations.txt)",
                         "all_conv_launches": {"tflops": all_conv, "frac": all_conv / MFMA_PEAK_TFLOPS, "ms": conv_ms,
                                               "share_of_step_time": conv_ms / (el * 1e3),
                                               "by_kernel": {k: {kk: vv for kk, vv in c.items() if kk != "flops"} for k, c in conv.items()}}},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_forwards, a.ddim_steps)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
