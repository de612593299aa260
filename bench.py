#!/usr/bin/env python3
"""Benchmark of the VFace per-frame DDIM denoising hot path on MI355X.

A "step" is ONE DDIM step of the hot path over one batch of synthetic input: pack [uncond ; cond ; recon]
(3F samples) -> hooked UNet forward -> guidance + x_{t-1} update, for the F frames this rank owns.
`value` = swapped frames/s for the whole job at 50 DDIM steps per frame = (F * n_gpus) / (50 * s_per_step).

Default workload (N = 1): BASELINE.json configs[2], the largest configuration of BASELINE's list that one GPU runs -- a
32-frame 512x512 clip (latent 64x64), 50-step DDIM schedule, frequency-spectrum attention interpolation (fusion "fft" on the
input-block attn1 modules), 96 samples per step -- on synthetic latents / conditioning / weights (name-keyed deterministic
fill; there are no checkpoints on the GPU box).  configs[1] (8 frames, "replace": the headline of rounds 1-4) and the
16-frame flow_fix share are timed in the same process as `extra`.
Inputs are resident in HBM when the timed region starts.  With --gpus N every rank runs the same per-GPU
workload on its own frames (weak scaling); `--fusion flow_fix` adds the FSAI + flow path, whose one-neighbour
boundary exchange runs over RCCL.

With --gpus N > 1 the defaults change to what north_star asks the multi-GPU run to measure: the shipped hook
schedule (`--fusion flow_fix`) at BASELINE config 4's per-GPU share (16 frames per GPU), so the one-neighbour halo
exchange over RCCL is inside the timed region, and the line carries `exchange: {mode, bytes_per_step,
wait_ms_per_step}`, `ranks_seen` (an all-reduce of 1 over the group) and `scaling_anchor`: the like-for-like
single-GPU figure of that workload is `python bench.py --gpus 1 --fusion flow_fix --frames 16` (= the flow_fix extra of the
default N = 1 line), not the default N = 1 `value` (configs[2]: fft, 32 frames).  Every wait on a peer is bounded
(VFACE_EXCHANGE_TIMEOUT_S, default 120 s): a rank whose neighbour never sends exits non-zero naming it.

Also reported on the same JSON line:
  roofline     -- the kernel family with the largest share of the step (by HIP events on the launch stream around
                  every launch, recorded in an instrumented kernel-by-kernel pass of the same W + K steps right after
                  the timed region): FLOPs per launch / mean launch duration against the dense 16-bit MFMA peak
                  (2.5 PFLOP/s, MI355X_MICROARCH.md), with `by_family` = {gemm, conv, attention, norm} so that the
                  whole-UNet figure can be recomputed from its parts.
  cpu_baseline -- the CPU oracle (torch fp32 restatement of the reference, pinned to reference-generated
                  golden vectors) timed on this box's host cores on a bounded sample: 3 timed (+1 warm-up) hooked-UNet
                  forwards of an F=2 clip (batch 6) at 64x64 under the SAME hook mode as the headline (--fusion),
                  extrapolated to 50 steps per clip.
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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU (default: 32 at --gpus 1 = BASELINE configs[2]; "
                                                              "16 at --gpus N > 1 = config 4's per-GPU share)")
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--fusion", default=None, help="replace | fft | flow_fix | none (default: fft at --gpus 1; flow_fix -- the "
                                                   "shipped schedule, ddim_w_inv.py:303-305, with its halo exchange -- at --gpus N > 1)")
    ap.add_argument("--exchange", default="p2p", choices=["p2p", "allgather"], help="halo exchange form at --gpus N > 1")
    ap.add_argument("--inv-steps", type=int, default=3, help="DDIM-inversion steps timed after the run (0 = skip)")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-forwards", type=int, default=3)
    ap.add_argument("--no-extras", action="store_true",
                    help="N = 1 only: skip the extra workloads (configs[1]: 8 frames + replace; shipped schedule: 16 frames + flow_fix; 768x768)")
    ap.add_argument("--extra-steps", type=int, default=10)
    ap.add_argument("--streams", type=int, default=None, choices=[1, 2],
                    help="launch sequences of a graph-replayed forward (default: the engine's, VFACE_STREAMS or 2: two frame halves on "
                         "two HIP streams where no hook couples frames; 1 = one sequence, what the rocprofv3 summaries under profiles/ run)")
    ap.add_argument("--eager", action="store_true",
                    help="timed region launches kernel by kernel (default: the UNet forward of a step replayed from a hipGraph, "
                         "falling back to kernel-by-kernel launches by itself where a graph cannot be captured)")
    return resolve_defaults(ap.parse_args(argv))


def resolve_defaults(a):
    """Workload defaults by GPU count.  One GPU: BASELINE configs[2] -- 32 frames, frequency-spectrum attention interpolation
    ("fft"), the largest single-GPU configuration of BASELINE's list.  N > 1: the shipped schedule, "flow_fix" (REFace/ldm/models/diffusion/ddim_w_inv.py:303-305), 16 frames per
    GPU (config 4's share) -- the only workload whose data path has an exchange step (temporal_flow.py:222-237: frame i+1 reads
    frame i), so a scaling curve from it measures the RCCL halo exchange north_star names."""
    if a.fusion is None:
        a.fusion = "fft" if a.gpus == 1 else "flow_fix"
    if a.frames is None:
        a.frames = 32 if a.gpus == 1 else 16
    return a


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


class FamilyTimer:
    """HIP-event timing (on the launch stream) of every launch of the instrumented pass, by kernel family:
      conv       by kernel: `patch3` = conv_patch_kernel<3x3> (conv.hip), `patch2` = its 2x2 parity-phase form (four launches
                 per call), `patch8x8` = its 8x8 form (four images per workgroup; the time includes the split-K reduce pass),
                 `im2col` = gemm.hip's implicit GEMM (stride 2); `in16` = inconv.hip (the 9 -> 320 input convolution, K = 144 in one
                 MFMA pass, bound by its stores: FLOPs as executed over the 16 stored channels); `out_fused` = outconv.hip (GroupNorm + SiLU +
                 the 320->4 convolution in one launch, vector dot products: priced at its ALGORITHMIC 2 M 4 9 Cin FLOPs);
      gemm       gemm_kernel<T, MODE_PLAIN, ..>: every Linear / 1x1 conv, keyed by shape and epilogue form: `ff1_MxNxK` = the GEGLU projections,
                 `MxNxK+rb+r32+o32+cs+a2` the rest (row bias, fp32 residual rows in, fp32 carrier out, column statistics, dual-source K);
                 launches that qualify run gemm_big.hip's 256 x 320 tile; FLOPs = 2 M N K as executed (incl. the folded FSAI K = 2d);
                 `ffn_fused` = ffn_fused_kernel (ffn.hip): LayerNorm + both FeedForward GEMMs + residual of a level-0 block;
                 `attn_out_ffn_fused` = the same kernel with attn1's out-projection, attn2's row bias and the residual in front;
                 `attn_out_ffn_proj_fused` = ... and the SpatialTransformer's proj_out + input residual + column statistics behind;
                 `st_front` = st_front_kernel (stfront.hip): GroupNorm-apply + proj_in + LayerNorm + attn1 projection of a level-0 block;
                 `linear_small` = linear_small_kernel: the time-embedding chain on 3F rows;
      attention  attn_kernel<T, DH, ..> by head dim; FLOPs = the ALGORITHMIC 4 n nk dh per (output sample, head) --
                 the shared-score form executes fewer;
      norm       layernorm / groupnorm apply+finalize: HBM-bound, reported in GB/s of algorithmic bytes."""

    def __init__(self):
        self.cls = {}
        self.on = False

    def _timed(self, fam, key, flops, launches, fn, nbytes=0.0):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        c = self.cls.setdefault((fam, key), {"events": [], "flops": 0.0, "launches": 0, "bytes": 0.0})
        c["events"].append((e0, e1)); c["flops"] += flops; c["launches"] += launches; c["bytes"] += nbytes

    def reset(self):
        self.cls = {}

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
            kind = {1: "patch3", 2: "patch8x8"}.get(patch, "im2col")
            if cin == 16 and stride == 1 and not upsample and not (OH * OW) % 64 and not OW % 16 and not cout % 80:
                kind = "in16"      # (vf_conv_in16_ok's geometry: the UNet's input convolution on inconv.hip)
            timer._timed("conv", kind, 2.0 * nimg * OH * OW * cout * 9 * cin, 1, call)
        hip.conv3x3 = conv3x3
        orig_up = hip.upsample2x_conv3x3

        def upsample2x_conv3x3(x, wt, out, *, nimg, H, W, cin, cout, **kw):
            # four parity-phase launches of a 2x2 window on the low-resolution input: FLOPs counted as executed (4 taps),
            # not as the 9-tap form they replace
            call = lambda: orig_up(x, wt, out, nimg=nimg, H=H, W=W, cin=cin, cout=cout, **kw)
            if not timer.on:
                return call()
            patch = hip.conv_uses_patch_kernel(H, W, cin, cout, 2, 1, False, kw.get("flags", 0)) == 1
            timer._timed("conv", "patch2" if patch else "im2col", 4 * 2.0 * nimg * H * W * cout * 4 * cin, 4, call)
        hip.upsample2x_conv3x3 = upsample2x_conv3x3
        orig_p1 = hip.conv3x3_plus_1x1

        def conv3x3_plus_1x1(x, x2, wt, out, *, nimg, H, W, cin, c2, cout, **kw):
            call = lambda: orig_p1(x, x2, wt, out, nimg=nimg, H=H, W=W, cin=cin, c2=c2, cout=cout, **kw)
            if not timer.on:
                return call()
            patch = hip.conv_uses_patch_kernel(H, W, cin, cout, 3, 1, False, kw.get("flags", 0)) if c2 % 64 == 0 else 0
            timer._timed("conv", {1: "patch3", 2: "patch8x8"}.get(patch, "im2col"), 2.0 * nimg * H * W * cout * (9 * cin + c2), 1, call)
        hip.conv3x3_plus_1x1 = conv3x3_plus_1x1
        orig_out = hip.gn_silu_conv3x3_small

        def gn_silu_conv3x3_small(x, gn_ab, wt, bias, out, *, nimg, H, W, cin, cout):
            # the UNet's out layer (GroupNorm + SiLU + conv3x3 to 4 channels) in one launch: algorithmic FLOPs 2 M cout 9 cin
            call = lambda: orig_out(x, gn_ab, wt, bias, out, nimg=nimg, H=H, W=W, cin=cin, cout=cout)
            if not timer.on:
                return call()
            timer._timed("conv", "out_fused", 2.0 * nimg * H * W * cout * 9 * cin, 1, call)
        hip.gn_silu_conv3x3_small = gn_silu_conv3x3_small
        orig_gemm = hip.gemm

        def gemm(a, wt, out, *, M, N, K, **kw):
            call = lambda: orig_gemm(a, wt, out, M=M, N=N, K=K, **kw)
            if not timer.on:
                return call()
            # by shape and epilogue form (what tools/step_trace.py prints per call): `ff1` = the GEGLU projections; the rest as
            # M x N x K + the operands of its epilogue (r32 = fp32 residual rows in, o32 = fp32 carrier out, cs = column statistics,
            # a2 = dual-source K: the hook's folded linear fusions)
            if kw.get("flags", 0) & hip.EPI_GEGLU:
                key = f"ff1_{M}x{N}x{K}"
            else:
                key = f"{M}x{N}x{K}" + "".join(t for t, k in (("+rb", "rowbias"), ("+r32", "residual32"), ("+o32", "out32"), ("+cs", "colstats"),
                                                            ("+a2", "a2")) if kw.get(k) is not None)
            timer._timed("gemm", key, 2.0 * M * N * K, 1, call)
        hip.gemm = gemm
        orig_ffn = hip.ffn_fused

        def ffn_fused(x32, gamma, beta, w1, b1, w2p, b2, out16, *, M, C_, **kw):
            # LayerNorm + ff.net[0] (GEGLU, N = 8 C) + ff.net[2] (K = 4 C) + residual in one launch: 2 M (8 C C + 4 C C) FLOPs
            call = lambda: orig_ffn(x32, gamma, beta, w1, b1, w2p, b2, out16, M=M, C_=C_, **kw)
            if not timer.on:
                return call()
            timer._timed("gemm", "ffn_fused", 24.0 * M * C_ * C_, 1, call)
        hip.ffn_fused = ffn_fused
        orig_tail = hip.attn_out_ffn_fused

        def attn_out_ffn_fused(att, resid32, rowbias, wo_w1, bo, gamma, beta, b1, w2p, b2, out16, *, M, C_, **kw):
            # attn1's out-projection (2 M C C) in front of the fused FeedForward (24 M C C), one launch
            call = lambda: orig_tail(att, resid32, rowbias, wo_w1, bo, gamma, beta, b1, w2p, b2, out16, M=M, C_=C_, **kw)
            if not timer.on:
                return call()
            timer._timed("gemm", "attn_out_ffn_fused", 26.0 * M * C_ * C_, 1, call)
        hip.attn_out_ffn_fused = attn_out_ffn_fused
        orig_ls = hip.linear_small

        def linear_small(a, wt, bias, out, *, M, N, K, **kw):
            # the time-embedding chain's few-row Linear layers (linear_small.hip); FLOPs 2 M N K as executed
            call = lambda: orig_ls(a, wt, bias, out, M=M, N=N, K=K, **kw)
            if not timer.on:
                return call()
            timer._timed("gemm", "linear_small", 2.0 * M * N * K, 1, call)
        hip.linear_small = linear_small
        orig_tailp = hip.attn_out_ffn_proj_fused

        def attn_out_ffn_proj_fused(att, resid32, rowbias, w_stream, bo, gamma, beta, b1, w2p, b2, b_po, x_in, out16, out32, colstats, *,
                                    M, C_, **kw):
            # ... and the SpatialTransformer's proj_out (2 M C C) behind it, still one launch
            call = lambda: orig_tailp(att, resid32, rowbias, w_stream, bo, gamma, beta, b1, w2p, b2, b_po, x_in, out16, out32, colstats,
                                      M=M, C_=C_, **kw)
            if not timer.on:
                return call()
            timer._timed("gemm", "attn_out_ffn_proj_fused", 28.0 * M * C_ * C_, 1, call)
        hip.attn_out_ffn_proj_fused = attn_out_ffn_proj_fused
        orig_front = hip.st_front

        def st_front(x32, gn_ab, wcat, b_in, gamma, beta, t0, qkv, *, M, C_, hw, NQ, rows_full, nq_lo=0, **kw):
            # GroupNorm-apply + proj_in + LayerNorm + the attn1 projection of a level-0 SpatialTransformer in one launch:
            # 2 M C C (proj_in) + 2 C (rows_full NQ + (M - rows_full) (NQ - nq_lo)) FLOPs as executed
            call = lambda: orig_front(x32, gn_ab, wcat, b_in, gamma, beta, t0, qkv, M=M, C_=C_, hw=hw, NQ=NQ, rows_full=rows_full,
                                      nq_lo=nq_lo, **kw)
            if not timer.on:
                return call()
            timer._timed("gemm", "st_front", 2.0 * M * C_ * C_ + 2.0 * C_ * (rows_full * NQ + (M - rows_full) * (NQ - nq_lo)), 1, call)
        hip.st_front = st_front
        orig_attn = hip.attention

        def attention(q, k, v, out, *, B, heads, n, nk, dh, v_sets=1, **kw):
            call = lambda: orig_attn(q, k, v, out, B=B, heads=heads, n=n, nk=nk, dh=dh, v_sets=v_sets, **kw)
            if not timer.on:
                return call()
            timer._timed("attention", f"dh{dh}" + (f"_shared{v_sets}" if v_sets > 1 else ""),
                         4.0 * n * nk * dh * heads * B * max(v_sets, 1), 1, call)
        hip.attention = attention
        orig_ln = hip.layernorm

        def layernorm(x, gamma, beta, out, *, M, C_, **kw):
            call = lambda: orig_ln(x, gamma, beta, out, M=M, C_=C_, **kw)
            if not timer.on:
                return call()
            timer._timed("norm", "layernorm", 0.0, 1, call, nbytes=float(M) * C_ * (x.element_size() + out.element_size()))
        hip.layernorm = layernorm
        orig_gna = hip.groupnorm_apply

        def groupnorm_apply(x, stats, gamma, beta, out, *, nimg, hw, C_, **kw):
            call = lambda: orig_gna(x, stats, gamma, beta, out, nimg=nimg, hw=hw, C_=C_, **kw)
            if not timer.on:
                return call()
            timer._timed("norm", "groupnorm_apply", 0.0, 1, call, nbytes=float(nimg) * hw * C_ * (x.element_size() + out.element_size()))
        hip.groupnorm_apply = groupnorm_apply
        orig_gnc = hip.groupnorm_stats_from_cols

        def groupnorm_stats_from_cols(colstats, *, nimg, hw, C_, **kw):
            if not timer.on:
                return orig_gnc(colstats, nimg=nimg, hw=hw, C_=C_, **kw)
            box = []
            timer._timed("norm", "groupnorm_finalize", 0.0, 1, lambda: box.append(orig_gnc(colstats, nimg=nimg, hw=hw, C_=C_, **kw)),
                         nbytes=float(nimg) * (hw // 64) * C_ * 8)
            return box[0]
        hip.groupnorm_stats_from_cols = groupnorm_stats_from_cols
        orig_gco = hip.groupnorm_coeffs_from_cols

        def groupnorm_coeffs_from_cols(colstats, gamma, beta, *, nimg, hw, C_, **kw):
            if not timer.on:
                return orig_gco(colstats, gamma, beta, nimg=nimg, hw=hw, C_=C_, **kw)
            box = []
            timer._timed("norm", "groupnorm_finalize", 0.0, 1, lambda: box.append(orig_gco(colstats, gamma, beta, nimg=nimg, hw=hw, C_=C_, **kw)),
                         nbytes=float(nimg) * (hw // 64) * C_ * 8)
            return box[0]
        hip.groupnorm_coeffs_from_cols = groupnorm_coeffs_from_cols
        orig_fw = hip.flow_warp

        def flow_warp(src, dst, flow, *, F, h, w, C_, **kw):
            call = lambda: orig_fw(src, dst, flow, F=F, h=h, w=w, C_=C_, **kw)
            if not timer.on:
                return call()
            timer._timed("norm", "flow_warp", 0.0, 1, call, nbytes=float(F) * h * w * C_ * 2 * 6)
        hip.flow_warp = flow_warp

    def summary(self):
        out = {}
        for (fam, k), c in self.cls.items():
            ms = sum(a.elapsed_time(b) for a, b in c["events"])
            out.setdefault(fam, {})[k] = {"launches": c["launches"], "ms": ms, "flops": c["flops"], "bytes": c["bytes"],
                                          "tflops": c["flops"] / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                                          "gbps": c["bytes"] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
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


def cpu_baseline(n_forwards, ddim_steps, fusion):
    """Oracle (kind 'port') timed on the host cores, as BASELINE.md 4 plans it: full-size UNet, 64x64 latent, the hook mode
    of the headline beside it (`fusion` on the input-block attn1), F = 2 frames (batch 6: flow_fix then really warps a frame),
    1 warm-up + `n_forwards` (>= 3) timed forwards on every core of the affinity mask."""
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
    flow = [f[None] for f in synth.synth_flow(F_ - 1, 64, 64)] if fusion == "flow_fix" else None
    reg = {}
    if fusion != "none":
        ohooks.register_spa_attn_injection(reg, ounet.attn1_names(spec), 1, switch_on=True, input_blocks=True,
                                           middle_block=False, output_blocks=False, chunks=3, flow=flow,
                                           block_indices=list(range(9)), fusion=fusion, split_ratio_fft=0.8, alpha=0.8)
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
                      f"64x64 latent, fp32, fusion={fusion} on the input-block attn1 (the headline's hook mode); {dt:.2f} s per "
                      f"forward, x{ddim_steps} steps per clip of {F_} frames"}


def traffic_from_profiles(prefixes, workload_ok):
    """HBM-side bytes per launch of the kernels whose names start with one of `prefixes`, QUOTED from the newest committed
    profiles/*_hbm_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command: tools/traffic.sh;
    2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md "HBM") -- never measured in this run, and only
    when that summary was collected on the SAME kernel sources this run is built from (`_build.source_sha16`): a stale
    summary yields null and says why."""
    from vface_amd.utils.buildinfo import source_sha16
    prof = os.path.join(ROOT, "profiles")
    cands = sorted(f for f in os.listdir(prof) if f.endswith("_hbm_traffic.json")) if os.path.isdir(prof) else []
    if not cands:
        return None, "no profiles/*_hbm_traffic.json"
    if not workload_ok:
        return None, "the committed traffic summaries are of the default workload (32 frames, 512x512, fft, fp16) only"
    tj = json.load(open(os.path.join(prof, cands[-1])))
    have, want = (tj.get("_build") or {}).get("source_sha16"), source_sha16()
    if have != want:
        return None, (f"profiles/{cands[-1]} was collected on kernel sources {have or 'unrecorded'}, this run is built from "
                      f"{want}: not quoted (re-run tools/traffic.sh)")
    sel = [v for k, v in tj.items() if not k.startswith("_") and any(k.startswith(pf) for pf in prefixes)]
    n = sum(v["launches"] for v in sel)
    if not n:
        return None, f"profiles/{cands[-1]} has no kernel named {prefixes}"
    return (sum(v["hbm_bytes_per_launch"] * v["launches"] for v in sel) / n,
            f"QUOTED from profiles/{cands[-1]} {list(prefixes)} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
            f"command on kernel sources {want}), not measured in this run")


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
    dist, backend, ranks_seen = None, None, 1
    if world > 1:
        import torch.distributed as dist
        # every collective / point-to-point wait of the run is bounded (VFACE_EXCHANGE_TIMEOUT_S, default 120 s): under RCCL the
        # process group's watchdog aborts a rank whose peer never shows up, under gloo FrameShard's waits raise ExchangeTimeout
        from vface_amd.parallel import process_group_timeout
        if rehearse:
            dist.init_process_group("gloo", timeout=process_group_timeout())
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=process_group_timeout())   # "nccl" IS RCCL on ROCm
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
        backend = dist.get_backend()
        one = torch.ones(1, dtype=torch.int32, device="cpu" if rehearse else dev)
        dist.all_reduce(one)                       # every rank really is in the group: reported as `ranks_seen`
        ranks_seen = int(one.item())
        assert ranks_seen == a.gpus, (ranks_seen, a.gpus)

    from vface_amd import hip
    from vface_amd.ldm.models.diffusion.ddim_w_inv import DDIMSampler, HookPlan
    from vface_amd.ldm.models.diffusion.ddpm import FFHQ_UNET_CONFIG, LatentDiffusion
    from vface_amd.parallel import FrameShard
    from vface_amd.utils import synth

    hip.load()
    log(f"building the 859.5M-parameter UNet with synthetic weights (rank {rank}/{world}) ...")
    timer = FamilyTimer()
    timer.wrap(hip)
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    ldm = LatentDiffusion(dict(FFHQ_UNET_CONFIG, compute_dtype=dt))
    synth.fill_module_(ldm.unet, seed=0)
    ldm = ldm.to(dev)
    sampler = DDIMSampler(ldm)
    sampler.make_schedule(a.ddim_steps, ddim_eta=0.0, verbose=False)
    steps = [int(s) for s in sampler.ddim_timesteps[::-1]]
    eng = ldm.unet.engine
    if a.streams is not None:
        eng.split_streams = a.streams
    UNET_GFLOP = {64: 796.94, 96: 2137.52, 32: 176.34}   # BASELINE.md 2 / SURVEY 8d, per sample-forward, by latent size

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def run_workload(F_, fusion, n_steps, n_warm, instrument, inv_steps, graph=True, res=None, drop=False):
        """W untimed + exactly K timed DDIM steps of an F-frame clip per GPU at resolution `res`; returns per-rank wall
        seconds etc.  graph=True: the UNet forward of a step replayed from a hipGraph (UNetEngine.step_forward_nhwc: the
        product default) -- no per-launch events can be recorded inside a graph, so `instrument=True` (events around every
        launch, by family) goes with graph=False."""
        h = (res or a.res) // 8
        # the reference's flow gate (4096 tokens) only ever fires at 512 x 512: other resolutions use the generalised one
        sampler.flow_gate = "reference" if h == 64 else "flow_hw"
        sampler.hook_plan = HookPlan(fusion=fusion, enabled=fusion != "none")
        sampler.drop_dead_branches = bool(drop)     # (extras only: the headline always runs the full 3F batch)
        eng.use_graph, eng._graphs, eng._graph_failed = bool(graph), {}, set()
        eng._split_state = {}
        eng.decompose_attn1 = bool(instrument)
        shard = FrameShard(rank, world, F_ * world, dist, mode=a.exchange)
        g0 = shard.first  # global index of this rank's first frame
        tag = lambda s_, f: f"bench.{s_}.{g0 + f}"
        stack = lambda s_, shape: torch.stack([synth.synth_normal(tag(s_, f), shape) for f in range(F_)]).to(dev)
        x_T = stack("xT", (4, h, h))
        c, uc, tc = stack("c", (1, 768)), stack("uc", (1, 768)), stack("tc", (1, 768))
        inp = stack("inp", (4, h, h)) * 0.18215
        mask = synth.synth_mask(F_, h, h).to(dev)
        inv = {s_: stack(f"inv{s_}", (4, h, h)) for s_ in steps}  # device-resident recon latents
        flow = None
        eng.halo_exchange, eng.halo_flow = None, None
        eng.exchange_events = [] if (world > 1 and fusion == "flow_fix") else None
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
            # (graph mode: capture happens in the first step, the two-stream overlap check in the second, and a fall-back to one
            # launch sequence would capture again in the third -- none of that may land in the timed region, however small --warmup is)
            for i in range(max(n_warm, 3 if graph else 0)):
                img = one_step(img, i)
            img = x_T
            fence()
            if eng.exchange_events is not None:
                eng.exchange_events.clear()
            timer.on = instrument
            t0 = time.perf_counter()
            done_clips = []
            for i in range(n_steps):
                if i and i % len(steps) == 0:
                    # K beyond one schedule: a new clip's start latents -- the synthetic UNet iterated through the schedule again and
                    # again on its own output leaves fp32 range.  A tensor hand-over, nothing enqueued; the finished clip's latents are
                    # kept so that the bit comparison between the timed region and the kernel-by-kernel pass covers every step
                    done_clips.append(img)
                    img = x_T
                img = one_step(img, i)
            t_enq = time.perf_counter() - t0      # host time to enqueue the steps (the GPU runs behind)
            fence()
            el = time.perf_counter() - t0
            timer.on = False
        assert torch.isfinite(img).all(), "non-finite latents"
        graphed = bool(eng.use_graph and eng._graphs)
        nseg = max((len(g["segments"]) for g in eng._graphs.values()), default=0)
        exch = None
        if world > 1 and fusion == "flow_fix":
            wait_ms = sum(e0.elapsed_time(e1) for e0, e1 in eng.exchange_events) / max(n_steps, 1)
            # level-0 maps: n = h*h tokens, d = model_channels = 320 -> one [n, 2d] 16-bit slab per hooked layer (2 of them)
            exch = {"mode": a.exchange, "bytes_per_step": shard.slab_bytes_per_step(h * h, 320),
                    "wait_ms_per_step": wait_ms, "exchanges_per_step": len(eng.exchange_events) / max(n_steps, 1),
                    "note": "bytes this rank sends per DDIM step (one [n, 2d] fp16 slab per hooked level-0 layer, one hop down the "
                            "chain; none from the last rank); wait = HIP events around the finish_exchange calls on rank 0's "
                            "successor-facing stream, i.e. how long the launch stream stalls for the neighbour's slab"}
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
        split2 = graphed and any(k[0] != "plan" and "streams" in v for k, v in eng._split_state.items() if isinstance(v, dict))
        # (a coupled configuration whose single sequence measured faster than its two coupled halves stays on one: UNetEngine.split_timing)
        tkey = ("coupled" if fusion == "flow_fix" else "free", (2 if drop else 3) * F_, h, h)
        chose_one = tkey in eng._split_off
        if chose_one:
            split2, nseg = False, 1          # (the halves' segmented graphs are still cached; the steps replay the whole batch's single graph)
        two = split2 and (nseg <= 1 or dist is None)
        two_text = (": the frames of the batch as two halves (every chunk's frames [0, F/2) and [F/2, F)), each half its own graph, on two "
                    "HIP streams at once, joined every step (engine._step_forward_split; bit-identical to the single launch sequence)")
        if two and nseg > 1:
            two_text += (f"; the halves are coupled through one frame (flow_fix: half 0 hands its last frame's fused q|k to half 1 through a "
                         f"slot + event, parallel.StreamShard), each half a chain of {nseg} graph segments cut at the hand-overs")
        if chose_one:
            t2, t1 = eng.split_timing[tkey]
            two_text = (f": ONE launch sequence -- the engine timed this configuration both ways on its second step ({t2:.2f} ms as two "
                        f"frame halves on two streams, {t1:.2f} ms as one sequence) and kept the faster")
        launch = ("kernel by kernel" if not graphed else
                  ("hipGraph replay of the UNet forward of each step (UNetEngine.step_forward_nhwc)" + (two_text if (two or chose_one) else "")
                   if (nseg <= 1 or two) else
                   f"hipGraph replay in {nseg} segments cut at the halo exchanges, the RCCL send / recv / wait calls issued from the host "
                   "between them (engine._GraphSegments)"))
        eng.exchange_events = None
        sampler.drop_dead_branches = False
        return {"final_latents": torch.cat([t.detach() for t in done_clips] + [img.detach()]).clone(),
                "ms_step": el / n_steps * 1e3, "enqueue_ms": t_enq / n_steps * 1e3, "inv_ms": inv_ms, "elapsed": el,
                "launch": launch, "exchange": exch, "h": h}

    F_ = a.frames
    h = a.res // 8
    log("weights resident; warm-up ...")
    # Timed region: W + K DDIM steps with NO instrumentation inside -- the UNet forward of a step replayed from a hipGraph,
    # the product default (one host call instead of ~1200).  The per-launch HIP events behind `roofline` are recorded in a
    # second, kernel-by-kernel pass of the same W + K steps right after it (events cannot be recorded inside a captured
    # graph); that pass's own step time is reported beside.
    r = run_workload(F_, a.fusion, a.steps, a.warmup, False, a.inv_steps, graph=not a.eager)
    launch_mode = r["launch"] + ("" if (a.eager or r["launch"] != "kernel by kernel") else " (no graph captured: capture failed)")
    ms_step, inv_ms = r["ms_step"], r["inv_ms"]
    if inv_ms is not None:
        log(f"inversion: {inv_ms:.2f} ms/step (2F = {2 * F_} unhooked sample-forwards)")
    log(f"timed {a.steps} steps: {ms_step:.2f} ms/step (host enqueue {r['enqueue_ms']:.2f} ms/step; {launch_mode})")
    log("instrumented pass (kernel by kernel, HIP events around every GEMM / conv / attention / norm launch) ...")
    timer.reset()
    ri = run_workload(F_, a.fusion, a.steps, a.warmup, True, 0, graph=False)
    el = ri["elapsed"]
    log(f"  {ri['ms_step']:.2f} ms/step (host enqueue {ri['enqueue_ms']:.2f} ms/step)")
    # the timed region (hipGraph replay, two launch streams where they apply) and this pass (kernel by kernel, one launch sequence)
    # walked the same K steps from the same x_T: their latents must be the same bits
    same_bits = bool(torch.equal(r["final_latents"], ri["final_latents"]))
    log(f"  latents after {a.steps} steps: timed region {'==' if same_bits else '!='} kernel-by-kernel pass")
    fps = (F_ * world) / (a.ddim_steps * ms_step / 1e3)
    fam = timer.summary()
    unet_gflop = UNET_GFLOP.get(h)

    # the other single-GPU BASELINE configurations, same process and build, outside the timed region of the headline run
    extras = []
    if world == 1 and not a.no_extras and a.res == 512:
        for name, f2, fus, res2 in (
                ("BASELINE configs[1]: 8-frame 512x512 clip, structure attention injection only (the headline of rounds 1-4)", 8, "replace", 512),
                ("BASELINE configs[2]: 32-frame 512x512 clip + frequency-spectrum attention interpolation", 32, "fft", 512),
                ("shipped hook schedule (ddim_w_inv.py:303-305), config 4's per-GPU share: 16 frames + flow_fix", 16, "flow_fix", 512),
                ("shipped hook schedule (ddim_w_inv.py:303-305) at the headline's size: 32 frames + flow_fix", 32, "flow_fix", 512),
                ("BASELINE configs[4]'s per-GPU share: 32 frames at 768x768 (96x96 latents, n = 9216 tokens at level 0), all three "
                 "modules (flow_fix, flow gate generalised to the flow field's h*w: the reference's n == 4096 gate never fires "
                 "at this size)", 32, "flow_fix", 768)):
            if f2 == F_ and fus == a.fusion and res2 == a.res:
                continue
            log(f"extra workload: {f2} frames at {res2}x{res2}, fusion={fus} ...")
            e = run_workload(f2, fus, a.extra_steps if res2 == 512 else max(3, a.extra_steps // 2), 2, False, 0, graph=not a.eager, res=res2)
            same2 = None
            if fus == "flow_fix" and res2 == 512 and not a.eager and eng.split_streams == 2 and "two halves" in e["launch"]:
                # flow_fix on two launch streams runs its halves as coupled in-process shards (parallel.StreamShard): every bench run
                # re-checks that hand-over against the single launch sequence, bit for bit, on this workload
                eng.split_streams = 1
                e1 = run_workload(f2, fus, a.extra_steps, 2, False, 0, graph=True, res=res2)
                eng.split_streams = 2
                same2 = bool(torch.equal(e["final_latents"], e1["final_latents"]))
                log(f"  two launch streams {'==' if same2 else '!='} one launch sequence after {a.extra_steps} steps (one sequence: {e1['ms_step']:.2f} ms/step)")
                if not same2:
                    # fail closed: a two-stream result that is not the single sequence's bits is not reported as a timing --
                    # this workload's line is the one-sequence run, and the engine stays on one launch sequence for what follows
                    e, eng.split_streams = e1, 1
                    log("  MISMATCH: reporting the one-sequence timing; the engine stays on one launch sequence")
            extras.append({"workload": name, "frames_per_gpu": f2, "fusion": fus, "res": res2, "latent": [e["h"], e["h"]],
                           "launch_streams": 2 if "two halves" in e["launch"] else 1, "two_streams_bits_equal_one_sequence": same2,
                           # ms of one step as two launch sequences (frame halves on two streams) / as one, timed by the engine on the second
                           # step of this configuration; it keeps the faster form (one sequence only if it wins by more than 0.5 %)
                           "engine_two_vs_one_sequence_ms": eng.split_timing.get(("coupled" if fus == "flow_fix" else "free", 3 * f2, e["h"], e["h"])),
                           "steps": a.extra_steps if res2 == 512 else max(3, a.extra_steps // 2), "warmup": 2,
                           "launch": "hipGraph replay" if e["launch"] != "kernel by kernel" else "kernel by kernel",
                           "ms_per_step": e["ms_step"], "host_enqueue_ms_per_step": e["enqueue_ms"],
                           "frames_per_s": f2 / (a.ddim_steps * e["ms_step"] / 1e3),
                           "unet_algorithmic_tflops": 3 * f2 * UNET_GFLOP[e["h"]] * 1e9 / (e["ms_step"] * 1e-3) / 1e12})
            log(f"  {e['ms_step']:.2f} ms/step = {extras[-1]['frames_per_s']:.2f} frames/s "
                f"({extras[-1]['unet_algorithmic_tflops']:.0f} TFLOP/s algorithmic)")
    # exact dead-branch elimination, reported BESIDE the headline and never as `value`: BASELINE's metric
    # defines a swapped frame as 50 x 3 sample-forwards; the sampler drops the recon branch's x_prev and the inversion's source
    # half, so the same outputs (bit-identical: tests/test_unet_gpu.py::test_drop_recon_is_bit_identical) need 2 + 1 of 3 + 2
    dead = None
    if world == 1 and not a.no_extras and a.res == 512:
        log(f"extra: dead-branch elimination ({F_} frames, fusion={a.fusion}: sampling on [uncond;cond], inversion on the target half) ...")
        e = run_workload(F_, a.fusion, a.extra_steps, 2, False, a.inv_steps, graph=not a.eager, drop=True)
        dead = {"note": "sampler.drop_dead_branches: sampling on [uncond ; cond] (2F samples per step: the recon third's x_prev is dropped "
                        "by the sampler and no hook mode reads chunk 2), inversion on the target half (F samples: only it is saved); "
                        "outputs bit-identical to the full batches; NOT the metric's definition of a frame (3 sample-forwards per step)",
                "frames_per_gpu": F_, "fusion": a.fusion, "steps": a.extra_steps, "sampling_ms_per_step": e["ms_step"],
                "sampling_frames_per_s": F_ / (a.ddim_steps * e["ms_step"] / 1e3), "inversion_ms_per_step": e["inv_ms"],
                "frames_per_s_sampling_plus_inversion": None if e["inv_ms"] is None else F_ / (a.ddim_steps * (e["ms_step"] + e["inv_ms"]) / 1e3),
                "full_batch_sampling_ms_per_step": ms_step, "full_batch_inversion_ms_per_step": inv_ms}
        log(f"  sampling {e['ms_step']:.2f} ms/step (full batch {ms_step:.2f}), inversion "
            f"{e['inv_ms'] if e['inv_ms'] is not None else float('nan'):.2f} ms/step (full batch {inv_ms if inv_ms is not None else float('nan'):.2f})")
    eng.use_graph, eng._graphs = False, {}     # (frees the captured graphs' activation pools)
    # the widened pipeline end to end, outside the metric: VAE encode -> RAFT-shaped flow -> 50-step DDIM inversion
    # -> 50-step sampling with the shipped flow_fix schedule -> VAE decode -> paste-back, through the CLI's own code path
    e2e = None
    if world == 1 and not a.no_extras and a.res == 512 and a.dtype == "fp16":
        try:
            from vface_amd.scripts import VFace_inference_batch as cli
            log("extra: end-to-end pipeline (encode, flow, inversion, sampling, decode, paste-back), 2 x 8 frames ...")
            opt = cli.build_parser().parse_args(["--synthetic", "--with_vae", "--raft_flow", "--paste_back", "--skip_save", "--n_frames", "16",
                                                 "--n_samples", "8", "--fusion", "flow_fix", "--ddim_steps", str(a.ddim_steps),
                                                 "--Base_dir", "/tmp/vface_bench_e2e"])
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):      # (the CLI path prints its progress: stdout carries the ONE JSON line)
                res_ = cli.run_synthetic(opt)
            st = res_["batches"][-1]["stage_seconds"]        # the second batch: graphs captured, caches warm
            tot = sum(st.values())
            opt2 = cli.build_parser().parse_args(["--synthetic", "--with_vae", "--raft_flow", "--paste_back", "--skip_save", "--n_frames", "16",
                                                  "--n_samples", "8", "--fusion", "flow_fix", "--ddim_steps", str(a.ddim_steps),
                                                  "--Base_dir", "/tmp/vface_bench_e2e", "--drop_dead_branches"])
            with contextlib.redirect_stdout(sys.stderr):
                st2 = cli.run_synthetic(opt2)["batches"][-1]["stage_seconds"]
            # ... and with the next batch's inversion beside this batch's sampling (--pipeline_inversion): three batches,
            # the MIDDLE one is the steady state (its wall time: from the previous batch's last frame to its own)
            opt3 = cli.build_parser().parse_args(["--synthetic", "--with_vae", "--raft_flow", "--paste_back", "--skip_save", "--n_frames", "24",
                                                  "--n_samples", "8", "--fusion", "flow_fix", "--ddim_steps", str(a.ddim_steps),
                                                  "--Base_dir", "/tmp/vface_bench_e2e", "--drop_dead_branches", "--pipeline_inversion"])
            with contextlib.redirect_stdout(sys.stderr):
                mid = cli.run_synthetic(opt3)["batches"][1]
            e2e = {"workload": "8 frames 512x512 -> 1024x1024 pasted frames: VAE encode, flow (7 pairs, 20 updates), 50-step inversion (2F "
                               "samples), 50-step sampling (flow_fix), VAE decode, paste-back incl. the background's VAE round trip; "
                               "synthetic weights and frames; conditioning encoders, face alignment and video I/O are not part of it",
                   "stage_seconds": st, "seconds_per_8_frames": tot, "frames_per_s": 8.0 / tot,
                   "with_drop_dead_branches": {"stage_seconds": st2, "seconds_per_8_frames": sum(st2.values()),
                                               "frames_per_s": 8.0 / sum(st2.values()),
                                               "note": "the same pipeline with --drop_dead_branches (bit-identical frames)",
                                               "pipelined_inversion": {
                                                   "stage_seconds": mid["stage_seconds"], "seconds_per_8_frames": sum(mid["stage_seconds"].values()),
                                                   "frames_per_s": 8.0 / sum(mid["stage_seconds"].values()),
                                                   "wall_seconds_incl_host_generation_of_the_synthetic_frames": mid["batch_wall_seconds"],
                                                   "note": "+ --pipeline_inversion: batch k + 1's DDIM inversion runs beside batch k's sampling on two "
                                                           "HIP streams (DDIMSampler.sample_while_inverting; frames bit-identical to the sequential "
                                                           "order: tests/test_unet_gpu.py); the middle batch of three, the sum of its stages as "
                                                           "above (the next batch's encoding and flow are stages of this one: they run before it "
                                                           "samples), each stage's wall time with the GPU drained on both sides"}}}
            log(f"  {tot:.2f} s per 8 frames = {8.0 / tot:.2f} frames/s end to end: " + ", ".join(f"{k} {v * 1e3:.0f} ms" for k, v in st.items()))
            log(f"  with --drop_dead_branches {8.0 / sum(st2.values()):.2f} frames/s; + --pipeline_inversion {8.0 / sum(mid['stage_seconds'].values()):.2f} frames/s")
        except Exception as ex:       # an extra never costs the metric line
            e2e = {"error": f"{type(ex).__name__}: {ex}"}
            log(f"  end-to-end extra failed: {e2e['error']}")
    if rank == 0:
        step_ms_i = el * 1e3                     # the instrumented pass, all K steps
        unet_tflops = 3 * F_ * unet_gflop * 1e9 / (ms_step * 1e-3) / 1e12 if unet_gflop else None

        def fam_total(name):
            d = fam.get(name, {})
            ms = sum(c["ms"] for c in d.values())
            fl = sum(c["flops"] for c in d.values())
            by = sum(c["bytes"] for c in d.values())
            n = sum(c["launches"] for c in d.values())
            o = {"launches": n, "ms": ms, "ms_per_step": ms / a.steps, "share_of_step_time": ms / step_ms_i if step_ms_i else 0.0,
                 "by_kernel": {k: {kk: vv for kk, vv in c.items() if kk not in ("flops", "bytes") and not (kk == "gbps" and not c["bytes"])
                                   and not (kk == "tflops" and not c["flops"])} for k, c in sorted(d.items())}}
            if fl:
                o["tflops"] = fl / (ms * 1e-3) / 1e12 if ms else 0.0
                o["frac"] = o["tflops"] / MFMA_PEAK_TFLOPS
                o["tflop_per_step"] = fl / a.steps / 1e12
            if by:
                o["gbps"] = by / (ms * 1e-3) / 1e9 if ms else 0.0
                o["frac_of_8TBps"] = o["gbps"] / 8000.0
            return o

        by_family = {k: fam_total(k) for k in ("gemm", "conv", "attention", "norm")}
        by_family["gemm"]["kernel"] = "gemm_kernel<T, MODE_PLAIN, NT, DB, PERSIST, RM> (gemm.hip): every Linear / 1x1 conv, + ffn_fused_kernel<T, C> (ffn.hip): the level-0 FeedForward in one launch + st_front_kernel<T, C> (stfront.hip): GroupNorm-apply, proj_in, LayerNorm and the attn1 projection of a level-0 block in one launch; FLOPs as executed (2 M N K)"
        by_family["conv"]["kernel"] = "conv_patch_kernel<T, NT, KH, KW, ..> (conv.hip) + gemm_kernel<T, MODE_CONV_*> (im2col); FLOPs as executed"
        by_family["attention"]["kernel"] = "attn_kernel<T, DH, QT, G, LAZY> (attention.hip); algorithmic FLOPs 4 n nk dh per (output sample, head)"
        by_family["norm"]["kernel"] = "layernorm_kernel, gn_apply_kernel, gn_finalize_cols, flow_warp_kernel (pointwise.hip): HBM-bound, algorithmic bytes"
        covered = sum(by_family[k]["ms"] for k in by_family)
        # north_star's "fraction of the attention-GEMM roofline", as a KERNEL figure: the launches that compute attn1's
        # q|k|v projections and softmax(QK^T)V -- the attention kernels, the plain / dual-source (FSAI-folded) projection GEMMs of the 640- /
        # 1280-channel blocks, and the level-0 front kernel (which also holds proj_in: its FLOPs as executed are counted with it) -- FLOPs
        # as executed over their own event-timed duration, against the dense MFMA peak
        def _is_qkv(key):
            m = key.split("+")[0].split("x")
            return len(m) == 3 and all(v.isdigit() for v in m) and (int(m[1]) == 3 * int(m[2]) or "+a2" in key)
        a1 = [c for k, c in fam.get("gemm", {}).items() if k == "st_front" or _is_qkv(k)] + list(fam.get("attention", {}).values())
        a1_ms, a1_fl = sum(c["ms"] for c in a1), sum(c["flops"] for c in a1)
        attn1_kernel = None if not a1_ms else {
            "tflops": a1_fl / (a1_ms * 1e-3) / 1e12, "frac": a1_fl / (a1_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "ms_per_step": a1_ms / a.steps,
            "tflop_per_step": a1_fl / a.steps / 1e12,
            "kernels": "attn_kernel (all head dims) + the attn1 projection GEMMs (M x 3d x d and the dual-source FSAI-folded M x 2d x 2d) + "
                       "st_front_kernel (level 0: GroupNorm-apply, proj_in, LayerNorm and the q|k|v projection in one launch, all of its FLOPs)"}
        mfma_fams = ("gemm", "conv", "attention")
        dom = max(mfma_fams, key=lambda k: by_family[k]["ms"])
        d = by_family[dom]
        prefixes = {"gemm": ("gemm_kernel<F16, 0,", "gemm_kernel<BF16, 0,", "ffn_fused_kernel", "st_front_kernel"), "conv": ("conv_patch_kernel", "gemm_kernel<F16, 1,", "gemm_kernel<F16, 2,"),
                    "attention": ("attn_kernel",)}[dom]
        traffic, traffic_src = traffic_from_profiles(prefixes, F_ == 32 and h == 64 and a.fusion == "fft" and a.dtype == "fp16" and world == 1)
        baseline_cfg = {(8, "replace", 512): "BASELINE configs[1]: ", (32, "fft", 512): "BASELINE configs[2] (the largest single-GPU configuration): ",
                        (16, "flow_fix", 512): "BASELINE configs[3]'s per-GPU share (64 frames / 4 GPUs): ",
                        (32, "flow_fix", 768): "BASELINE configs[4]'s per-GPU share (256 frames / 8 GPUs): "}.get((F_, a.fusion, a.res), "")
        out = {
            "metric": "swapped frames/sec at 512x512, 50-step DDIM", "value": fps, "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if a.dtype == "fp16" else "bf16",
            "data": "synthetic",
            "config": {"workload": baseline_cfg + f"{F_}-frame {a.res}x{a.res} clip per GPU, {a.ddim_steps}-step DDIM, hooked REFace "
                                   f"UNet (859.5M params), attn1 fusion={a.fusion} on input blocks, CFG scale 3.0, "
                                   f"batch [uncond;cond;recon] = {3 * F_} samples per step",
                       "frames_per_gpu": F_, "latent": [h, h], "fusion": a.fusion,
                       "world_size": world, "backend": backend, "launch": launch_mode,
                       "launch_streams": 2 if "two halves" in launch_mode else 1,
                       # step time / (half A + half B) of the engine's one measured split step: ~0.5 = the halves ran at once
                       "launch_stream_overlap": eng.split_overlap,
                       # torch.equal of the latents after the K timed steps (graph replay, two streams) and after the same K steps
                       # launched kernel by kernel in one sequence (the instrumented pass)
                       "timed_region_bits_equal_kernel_by_kernel": same_bits,
                       # the engine's own check on the second split step of each kind of split ("free" / "coupled" halves): eps of the two
                       # launch sequences == eps of one sequence over the whole batch (UNetEngine.split_checked); and the same comparison
                       # over `extra_steps` DDIM steps of every two-stream flow_fix extra below (None: no such extra ran)
                       "two_sequence_selfcheck": dict(eng.split_checked),
                       # [ms as two launch sequences, ms as one] of the headline configuration, timed by the engine on its second step
                       "engine_two_vs_one_sequence_ms": eng.split_timing.get(("coupled" if a.fusion == "flow_fix" else "free", 3 * F_, h, h)),
                       "two_streams_bits_equal_one_sequence": (None if not any(x.get("two_streams_bits_equal_one_sequence") is not None for x in extras)
                                                               else all(x["two_streams_bits_equal_one_sequence"] for x in extras
                                                                        if x.get("two_streams_bits_equal_one_sequence") is not None)),
                       "unet_algorithmic_tflops_per_gpu": unet_tflops,
                       "unet_algorithmic_frac_of_mfma_peak": unet_tflops / MFMA_PEAK_TFLOPS if unet_tflops else None,
                       "host_enqueue_ms_per_step": r["enqueue_ms"],
                       # north_star also asks for the rate as a fraction of the attention-GEMM roofline: the attn1 QKV
                       # projections + QK^T + PV are 160.9 GFLOP per sample-forward at 64x64 = 24.1 TFLOP per
                       # swapped frame; at the 2.5 PFLOP/s dense peak that alone would allow 103.6 frames/s per GPU
                       "attention_gemm_roofline_frac": (fps / world) * 24.135e12 / (MFMA_PEAK_TFLOPS * 1e12) if h == 64 else None,
                       # ... the figure above is the attention-GEMM SHARE of the whole step's rate; this one is the roofline fraction of the
                       # attn1 kernels themselves: (projection + attention FLOPs as executed) / their own event-timed ms / peak
                       "attn1_kernel_roofline_frac": attn1_kernel["frac"] if attn1_kernel else None,
                       "attn1_kernel_roofline": attn1_kernel},
            "exchange": r["exchange"],
            "ranks_seen": ranks_seen,
            # the like-for-like single-GPU figure of an N > 1 line: the N = 1 DEFAULT is BASELINE configs[2] (32 frames, fft),
            # the N > 1 default the shipped schedule at config 4's share (16 frames per GPU, flow_fix)
            "scaling_anchor": None if world == 1 else (
                f"N=1 anchor of this workload: `python bench.py --gpus 1 --fusion {a.fusion} --frames {F_}` (also reported by the "
                f"default N=1 run among `extra` when fusion=flow_fix, frames=16); the default N=1 `value` is configs[2] "
                f"(fft, 32 frames) and is NOT the same workload"),
            "inversion": None if inv_ms is None else {
                "ms_per_step": inv_ms, "steps_timed": a.inv_steps,
                "note": "DDIM inversion step (hooks off, batch 2F, no guidance), outside the timed region; `value` is "
                        "sampling only, as BASELINE's metric",
                "frames_per_s_sampling_plus_inversion": (F_ * world) / (a.ddim_steps * (ms_step + inv_ms) / 1e3)},
            "extra": extras,
            "dead_branch_elimination": dead,
            "end_to_end": e2e,
            "instrumented_pass": {"launch": "kernel by kernel, ONE launch sequence over the whole batch, HIP events around every GEMM / convolution / "
                                            "attention / norm launch (what `roofline` is computed from; vface_attn1_forward's launches issued call "
                                            "by call, bit-identical).  With launch_streams = 2 the timed region runs the same kernels as two "
                                            "half-batch sequences at once: its step time is below this pass's sum of launch durations by what the "
                                            "overlap wins (profiles/r04_n)",
                                  "ms_per_step": ri["ms_step"], "host_enqueue_ms_per_step": ri["enqueue_ms"], "steps": a.steps, "warmup": a.warmup,
                                  "event_covered_share": covered / step_ms_i if step_ms_i else None},
            # the kernel family with the largest share of the step's kernel time (by family the plain GEMM, not the
            # convolution, is the largest and the furthest below the roofline); the others are in by_family
            "roofline": {"bound": "mfma", "kernel": d["kernel"], "family": dom,
                         "achieved": d["tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": d["tflops"] / MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "launches": d["launches"], "mean_launch_us": d["ms"] * 1e3 / max(d["launches"], 1),
                         "share_of_step_time": d["share_of_step_time"],
                         "timing": "HIP events on the launch stream around every launch of this family, recorded in the instrumented pass: the same "
                                   "W + K steps launched kernel by kernel right after the timed region (events cannot be recorded inside a "
                                   "captured graph); share_of_step_time is relative to that pass",
                         "note": "peak is the datasheet 2.5 PFLOP/s; an MFMA-only loop sustains ~1.5 PFLOP/s on this chip (it clocks ~1.6 GHz "
                                 "under matrix load: DESIGN.md 4).  whole-UNet check: sum over by_family of ms_per_step = the kernel time of a step; "
                                 "config.unet_algorithmic_tflops_per_gpu = 3F x 796.94 GFLOP / ms_per_step",
                         "by_family": by_family},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.cpu_forwards, a.ddim_steps, a.fusion)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except Exception as ex:
        from vface_amd.parallel import ExchangeTimeout
        if isinstance(ex, ExchangeTimeout):
            # a peer never showed up: say which, and leave with a non-zero code NOW (destroy_process_group would wait for it too)
            print(f"[bench] FATAL rank {os.environ.get('RANK', 0)}: {ex}", file=sys.stderr, flush=True)
            os._exit(3)
        raise
