"""Device-resident execution of the hooked ldm UNet on MI355X.

The nn.Modules in ``vface_amd.ldm`` only own parameters (with the reference's state-dict names); this
engine walks them and issues hand-written HIP kernels (``vface_amd/hip.py`` -> ``libvface_hip.so``):

* activations live in HBM as token-major / NHWC 16-bit matrices ``[N*H*W, C]`` -- the layout in which the
  reference's ``b c h w <-> b (h w) c`` rearranges (attention.py:284,287) are no-ops, every conv is an
  implicit GEMM over contiguous channels, and ``th.cat([h, hs.pop()], 1)`` (openaimodel.py:898) is two
  producers writing disjoint column ranges of one buffer;
* per-sample additive terms (the time-embedding projection of every ResBlock, and the single-token
  cross-attention, which reduces to ``to_out(to_v(ctx))`` broadcast over tokens -- SURVEY F11) are computed
  once per forward as small GEMMs and folded into GEMM epilogues as a row bias;
* the attn1 hook (pnp_utils.py:94-287) is executed from its configuration, not from the closure: "replace"
  is an index map inside the attention kernel, FSAI / mix are folded into the q,k projection weights, and
  the flow warp is one gather kernel on chunk 1's fused q|k.

There is no CPU path here.
"""
from __future__ import annotations

import os

import math

import numpy as np
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import hip, packing


@dataclass
class HookCfg:
    """Captured arguments of ``register_spa_attn_injection`` (pnp_utils.py:57) for one attn1 module."""
    switch_on: bool = True
    chunks: int = 3
    fusion: str = "replace"
    flow: Optional[torch.Tensor] = None  # [F-1, 2, h, w] fp32 on the device, or None
    split_ratio_fft: float = 0.8
    alpha: float = 0.8
    # which attention maps the flow smoothing applies to: "reference" = exactly pnp_utils.py:201 (`q.shape[1] == 4096`,
    # reshaped to 64 x 64: the 512 x 512 level-0 maps and nothing else -- at any other resolution the reference silently
    # skips the warp); "flow_hw" = the level whose token count equals h*w of the supplied flow field (identical at
    # 512 x 512; what a 768 x 768 clip -- BASELINE config 5 -- needs for the module to act at all)
    flow_gate: str = "reference"


class Act:
    """A 2-D view ``[rows, C]`` (stride ``(ld, 1)``) of a 16-bit device buffer, with its image geometry and,
    when its producer emitted them, the per-64-row-slice column statistics ``cs`` ``[rows/64, C, 2]`` (fp32 view)
    from which a following GroupNorm takes mean / rstd without re-reading the tensor.

    ``t32`` (optional) is the same activation as the fp32 residual-stream carrier: the un-rounded sum its producer's
    epilogue formed.  Residual adds and normalisations read it; only matrix-core operands read the 16-bit ``t``,
    which may then be absent (None) when no consumer needs it."""
    __slots__ = ("t", "N", "H", "W", "cs", "t32")

    def __init__(self, t: Optional[torch.Tensor], N: int, H: int, W: int, cs: Optional[torch.Tensor] = None,
                 t32: Optional[torch.Tensor] = None):
        for v in (t, t32):
            assert v is None or (v.dim() == 2 and v.stride(1) == 1)
        assert t is not None or t32 is not None
        self.t, self.N, self.H, self.W, self.cs, self.t32 = t, N, H, W, cs, t32

    @property
    def any(self):
        return self.t if self.t is not None else self.t32

    @property
    def C(self):
        return self.any.shape[1]

    @property
    def ld(self):
        return self.t.stride(0)

    @property
    def M(self):
        return self.any.shape[0]

    @property
    def src(self):
        """What a normalisation reads: the fp32 carrier when there is one."""
        return self.t32 if self.t32 is not None else self.t

    @property
    def hw(self):
        return self.H * self.W


def _dev_flow(flow, device) -> Optional[torch.Tensor]:
    """Accept the reference's ``list of [1,2,h,w]`` or a stacked tensor; return [F-1,2,h,w] fp32 on device."""
    if flow is None:
        return None
    if isinstance(flow, (list, tuple)):
        if len(flow) == 0:
            return None
        flow = torch.cat([f.reshape(1, 2, f.shape[-2], f.shape[-1]) for f in flow], 0)
    return flow.to(device=device, dtype=torch.float32).contiguous()


def flow_gate_hw(cfg: HookCfg, n: int, flow_hw):
    """The (h, w) of the maps to warp if the flow smoothing fires for an attention over ``n`` tokens, else None.
    ``flow_hw``: spatial size of the flow field in use (the local fields, or the clip's when frames are sharded)."""
    if flow_hw is None:
        return None
    h, w = int(flow_hw[0]), int(flow_hw[1])
    if cfg.flow_gate == "reference":
        if n != 4096:
            return None
        if (h, w) != (64, 64):   # the reference reshapes to 64 x 64 and adds the flow to a 64 x 64 grid (temporal_flow.py:43)
            raise RuntimeError(f"The size of the flow field ({h}, {w}) must match the 64 x 64 attention map "
                               "(pnp_utils.py:201-207, temporal_flow.py:43)")
        return 64, 64
    if cfg.flow_gate != "flow_hw":
        raise ValueError(f"flow_gate must be 'reference' or 'flow_hw', not {cfg.flow_gate!r}")
    return (h, w) if n == h * w else None


def plan_fusion(cfg: Optional[HookCfg], N: int, n: int, clip_flow_hw=None, live: Optional[int] = None) -> dict:
    """Map a hook configuration onto the kernels' mechanisms (pnp_utils.py:129-262).
    Returns fusion code, chunks, which folded weight to use, and the flow / v-broadcast options.  ``clip_flow_hw``: the
    flow field's (h, w) when frames are sharded (a one-frame shard has no local field but still takes part in the
    boundary exchange): ``warp_hw`` is then set even when ``flow`` is None.
    ``live``: the batch holds only the first ``live`` of the hook's ``chunks`` chunks (the sampler left the recon third out:
    every hook mode edits chunk k >= 1 from chunk 0 and chunk k alone, pnp_utils.py:133-262, so the chunks that ARE there
    compute exactly what they compute in the full batch); the plan's ``chunks`` is then ``live``, its fusion still the one the
    hook's own ``chunks`` selects."""
    pl = {"fusion": hip.FUSION_NONE, "chunks": 1, "wlin": None, "flow": None, "alpha": 0.8, "v_fixed": False,
          "staged": None, "warp_hw": None, "hook_chunks": 1}
    if cfg is None or not cfg.switch_on:
        return pl
    chunks = cfg.chunks
    if chunks not in (2, 3):
        return pl  # the reference edits nothing for other values
    there = chunks if live is None else live
    if not 1 <= there <= chunks:
        raise hip.VFaceHipError(f"hooked attn1: {there} live chunks of chunks={chunks}")
    if N % there:
        raise hip.VFaceHipError(f"hooked attn1: batch {N} is not divisible by its {there} chunk(s) (hook chunks={chunks})")
    pl["chunks"], pl["hook_chunks"] = there, chunks
    f = cfg.fusion
    if chunks == 2 or f == "replace":
        pl["fusion"] = hip.FUSION_REPLACE
    elif f in ("fft", "flow_fix", "fft_vfixed"):
        pl["fusion"] = hip.FUSION_LINEAR
        pl["wlin"] = ("fsai", 0.8 if f == "fft_vfixed" else cfg.split_ratio_fft)
        pl["v_fixed"] = f == "fft_vfixed"
        if f == "flow_fix":
            fhw = tuple(cfg.flow.shape[-2:]) if cfg.flow is not None else clip_flow_hw
            hw = flow_gate_hw(cfg, n, fhw)
            if hw is not None:
                nf = cfg.flow.shape[0] if cfg.flow is not None else 0
                if nf != N // there - 1:
                    raise RuntimeError(f"flow has {nf} fields for {N // there} frames "
                                       "(align_by_flow needs F-1, temporal_flow.py:231-233)")
                pl["flow"], pl["alpha"], pl["warp_hw"] = cfg.flow, cfg.alpha, hw
    elif f == "mix":
        pl["fusion"], pl["wlin"] = hip.FUSION_LINEAR, ("mix", 0.5)
    elif f in ("temporal", "adaIn"):
        pl["staged"] = f  # edits that are not a sample map or a folded weight: separate kernels on the qkv buffer
    else:
        pl["chunks"] = 1  # unknown fusion strings edit nothing in the reference
    if pl["chunks"] == 1 and pl["fusion"] != hip.FUSION_NONE:
        pl["fusion"], pl["wlin"], pl["flow"], pl["warp_hw"], pl["v_fixed"] = hip.FUSION_NONE, None, None, None, False   # chunk 0 alone: unedited
    return pl


def staged_attn1(x16: torch.Tensor, wqkv, wo, bo, out, *, B, n, d, heads, mode, rowbias=None, residual=None,
                 residual32=None, out32=None, chunks: int = 3):
    """Hooked attn1 for fusion modes that edit q,k with their own kernels ("temporal", "adaIn"; pnp_utils.py:145-160):
    full projection -> edit chunk 1 / chunk 2 q,k in the qkv buffer -> attention -> out-projection."""
    dev, dt = x16.device, x16.dtype
    c = B // chunks          # (chunks < 3: the batch came without its last chunk(s), plan_fusion `live`)
    Fn = c * n
    qkv = torch.empty(B * n, 3 * d, dtype=dt, device=dev)
    hip.gemm(x16, wqkv, qkv, M=B * n, N=3 * d, K=x16.shape[1], lda=x16.stride(0), ldc=3 * d)
    if mode == "temporal" and chunks > 1:
        hip.temporal_gauss(qkv, qkv[Fn:], qkv[2 * Fn:] if chunks > 2 else None, F=c, n=n, C_=2 * d, ld_src=3 * d, fs_src=n * 3 * d,
                           ld_dst=3 * d, fs_dst=n * 3 * d)
    elif mode == "temporal":
        pass
    elif mode == "adaIn":
        for col in (0, d):  # q then k
            for ch in range(1, chunks):
                own = qkv[ch * Fn:(ch + 1) * Fn, col:col + d]
                hip.adain_fusion(qkv[:Fn, col:col + d], own, own, rows=Fn, C_=d, lda=3 * d, ldb=3 * d, ldd=3 * d)
    else:
        raise ValueError(mode)
    att = torch.empty(B * n, d, dtype=dt, device=dev)
    hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=B, heads=heads, n=n, nk=n, dh=d // heads, ldq=3 * d, ldk=3 * d,
                  ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d, bso=n * d, scale=(d // heads) ** -0.5)
    o = out if out is not None else out32
    hip.gemm(att, wo, out, M=B * n, N=o.shape[1], K=d, lda=d, ldc=out.stride(0) if out is not None else 0, bias=bo,
             rowbias=rowbias, rows_per_sample=n, residual=residual, ldr=residual.stride(0) if residual is not None else 0,
             residual32=residual32, out32=out32)
    return o


COMPUTE_DTYPE = torch.float16  # module-level default for standalone module calls


def attn_module_forward(mod, x: torch.Tensor, context: Optional[torch.Tensor], cfg: Optional[HookCfg]):
    """``CrossAttention.forward`` / the hooked closure for a stand-alone module call on ``[B, n, d]`` CUDA
    tensors (attention.py:179-221, pnp_utils.py:94-287).  Returns the 16-bit result, as autocast does."""
    from . import packing
    dt = x.dtype if x.dtype in (torch.float16, torch.bfloat16) else COMPUTE_DTYPE
    pk = mod._packed(dt)
    B, n, d_in = x.shape
    d = mod.heads * mod.dim_head
    dev = x.device

    def to16(t):
        t = t.contiguous()
        if t.dtype == dt:
            return t
        o = torch.empty(t.shape, dtype=dt, device=dev)
        hip.cast_f32(t.float(), o)
        return o

    x16 = to16(x).reshape(B * n, d_in)
    out = torch.empty(B * n, mod.to_out[0].weight.shape[0], dtype=dt, device=dev)
    if context is None:
        pl = plan_fusion(cfg, B, n)
        if pl["staged"]:
            staged_attn1(x16, pk["wqkv"], pk["wo"], pk["bo"], out, B=B, n=n, d=d, heads=mod.heads, mode=pl["staged"])
            return out.reshape(B, n, -1)
        chunks = pl["chunks"]
        wlin = None
        if pl["wlin"]:
            key = (pl["wlin"][0], round(float(pl["wlin"][1]), 9))
            if key not in pk["wlin"]:
                wq, wk = mod.to_q.weight.detach().float().cpu(), mod.to_k.weight.detach().float().cpu()
                w = packing.fold_fsai(wq, wk, key[1]) if key[0] == "fsai" else packing.fold_mix(wq, wk, key[1])
                pk["wlin"][key] = w.to(device=dev, dtype=dt).contiguous()
            wlin = pk["wlin"][key]
        idx = torch.arange(B, dtype=torch.int32)
        c = B // chunks
        qk_map = (idx % c).to(dev) if pl["fusion"] == hip.FUSION_REPLACE else None
        v_map = torch.where(idx < c, idx, (idx // c) * c).to(dev) if pl["v_fixed"] else None
        ws = torch.empty(hip.attn1_workspace_bytes(B, n, d, chunks), dtype=torch.uint8, device=dev)
        flow = pl["flow"]
        hw = pl["warp_hw"] or (0, 0)
        hip.attn1_forward(x16, pk["wqkv"], wlin, pk["wo"], pk["bo"], out, B=B, n=n, d=d, heads=mod.heads,
                          chunks=chunks, fusion=pl["fusion"], ldx=d_in, ldo=out.shape[1], workspace=ws,
                          v_fixed=pl["v_fixed"], flow=flow, h=hw[0], w=hw[1], alpha=pl["alpha"], qk_map=qk_map,
                          v_map=v_map)
    else:
        m = context.shape[1]
        c16 = to16(context).reshape(B * m, context.shape[2])
        q = torch.empty(B * n, d, dtype=dt, device=dev)
        k = torch.empty(B * m, d, dtype=dt, device=dev)
        v = torch.empty(B * m, d, dtype=dt, device=dev)
        hip.gemm(x16, pk["wq"], q, M=B * n, N=d, K=d_in, lda=d_in, ldc=d)
        hip.gemm(c16, pk["wk"], k, M=B * m, N=d, K=c16.shape[1], lda=c16.shape[1], ldc=d)
        hip.gemm(c16, pk["wv"], v, M=B * m, N=d, K=c16.shape[1], lda=c16.shape[1], ldc=d)
        att = torch.empty(B * n, d, dtype=dt, device=dev)
        hip.attention(q, k, v, att, B=B, heads=mod.heads, n=n, nk=m, dh=mod.dim_head, ldq=d, ldk=d, ldv=d, bsq=n * d,
                      bsk=m * d, bsv=m * d, ldo=d, bso=n * d, scale=mod.scale)
        hip.gemm(att, pk["wo"], out, M=B * n, N=out.shape[1], K=d, lda=d, ldc=out.shape[1], bias=pk["bo"])
    return out.reshape(B, n, -1)


def _phase_form_pays(hw_in: int, cout: int) -> bool:
    """The four parity-phase launches of an upsampling conv each cover hw_in rows per sample: below ~200 tiles per launch the
    9-tap form on the upsampled grid fills the chip better.  The tile count is taken at a NOMINAL batch, so the choice -- and the
    bits -- do not depend on the batch: 48 samples since round 6 (one launch stream's half of the 32-frame headline, a rank's 16-frame
    share at N > 1), like the convolution kernels' own rules.  At 48 the 8x8 -> 16x16 upsampling at 1280 channels takes the phase form
    too (measured, profiles/r05_m / r06_j: 369 vs 434 us at 48 samples, 399 vs 829 at 96 -- and 369 vs 224 at 24, which the 8-frame
    clip pays); same-box A/B on the headline: 76.00 -> 75.42 ms/step.  VFACE_PHASE_NOMINAL24=1: the rule of rounds 2-5 (A/B)."""
    if os.environ.get("VFACE_NO_PHASE_UPSAMPLE") == "1":   # A/B switch for measurements
        return False
    if os.environ.get("VFACE_PHASE_NOMINAL24") == "1":
        return (24 * hw_in // 128) * (cout // 128) >= 400
    return (48 * hw_in // 128) * (cout // 128) >= 200


class _GraphSegments:
    """Capture of one UNet forward as a chain of hipGraphs sharing ONE memory pool, cut wherever the forward calls the
    frame-shard exchange (``start_exchange`` / ``finish_exchange``): those two calls are host-issued RCCL operations and run
    BETWEEN segment replays, on the same stream, in the captured order.  Installed as the engine's ``halo_exchange`` while
    capturing.  Buffers the captured launches exchange with the host calls have fixed addresses: the slab to send is a view
    of a pool tensor; the slab received lands in a persistent ``recv`` buffer owned by this object (the inner exchange
    receives straight into it where it can -- RCCL p2p -- otherwise its result is copied there).
    An unsharded forward is the degenerate case: one segment, no host call."""

    def __init__(self, engine, inner):
        self.engine, self.inner = engine, inner
        self.stream = torch.cuda.Stream()
        self.pool = torch.cuda.graph_pool_handle()
        self.segments: list = []      # [(CUDAGraph, host_op | None)]
        self.keep: list = []          # tensors the host ops refer to
        self.cur = None
        self.index = 0                # ordinal of the next exchange inside the forward (stated by the engine: FrameShard.set_index)
        self.n_started = 0            # exchanges the capture pass has started for real ..
        self.pending = None           # .. and the state dict of one that was started and not yet finished
        # what the engine reads from its exchange object
        if inner is not None:
            self.rank, self.world, self.first, self.count = inner.rank, inner.world, inner.first, inner.count

    def begin(self):
        self.cur = torch.cuda.CUDAGraph()
        # (thread_local: a collective backend's watchdog thread may poll events while this thread captures)
        self.cur.capture_begin(pool=self.pool, capture_error_mode="thread_local")

    def end(self, host_op):
        # A segment in which the forward launched nothing (two exchange calls back to back: half 0 of a coupled split has
        # nothing to wait for) is not kept -- torch says so with a warning at capture_end; replaying it would be a no-op launch
        # per step.  Its host call, if any, still runs at its place in the chain.
        import warnings
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            self.cur.capture_end()
        empty = any("Graph is empty" in str(w.message) for w in caught)
        for w in caught:
            if "Graph is empty" not in str(w.message):
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        if not empty or host_op is not None:
            self.segments.append((None if empty else self.cur, host_op))
        self.cur = None

    def abort(self):
        """End a capture that will not be used.  ``capture_end`` must run on the CAPTURING stream: the exception that brings us
        here has already unwound the ``with torch.cuda.stream(..)`` block, and ending the capture from another stream is a
        fatal HIP error (process abort), not a Python exception."""
        if self.cur is not None:
            try:
                with torch.cuda.stream(self.stream):
                    self.cur.capture_end()
            except Exception:
                pass
            self.cur = None

    # ---- the exchange interface (parallel.FrameShard), as seen by UNetEngine._attn1_sharded while capturing
    def set_index(self, k: int) -> None:
        self.index = int(k)

    def start_exchange(self, tail: torch.Tensor):
        inner, state, k = self.inner, {}, self.index
        recv = torch.empty_like(tail) if inner.rank > 0 else None      # (allocated in the shared pool: lives with the graphs)
        self.keep += [tail, recv, state]

        def op():
            if hasattr(inner, "set_index"):
                inner.set_index(k)                 # (the replayed host call states the same ordinal as the captured forward did)
            state["h"] = inner.start_exchange(tail, recv=recv)
        self.end(op)
        op()                       # the capture pass exchanges for real too (garbage slabs): the ranks' calls stay paired
        self.n_started += 1
        self.pending = state
        self.begin()
        return (state, recv)

    def finish_exchange(self, handle):
        state, recv = handle
        inner, eng = self.inner, self.engine
        if os.environ.get("VFACE_TEST_FAIL_CAPTURE") == "mid":      # (test hook: die with one exchange started, not finished)
            raise RuntimeError("injected capture failure (VFACE_TEST_FAIL_CAPTURE)")

        def op():
            ev = eng.exchange_events
            if ev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            halo = inner.finish_exchange(state.pop("h"))
            if halo is not None and recv is not None and halo.data_ptr() != recv.data_ptr():
                recv.copy_(halo)
            if ev is not None:
                e1.record()
                ev.append((e0, e1))
        self.end(op)
        op()
        self.pending = None
        self.begin()
        return recv


class _CountingExchange:
    """The frame-shard exchange seen by the warm-up forward of a capture: passes every call through and records the slabs a
    forward sends (shapes only), so that a capture pass that dies half-way can finish its paired exchanges (``FrameShard.drain``)."""

    def __init__(self, inner):
        self.inner, self.tails = inner, []
        self.rank, self.world, self.first, self.count = inner.rank, inner.world, inner.first, inner.count

    def set_index(self, k: int) -> None:
        if hasattr(self.inner, "set_index"):
            self.inner.set_index(k)

    def start_exchange(self, tail, recv=None):
        self.tails.append(tail)
        return self.inner.start_exchange(tail, recv=recv) if recv is not None else self.inner.start_exchange(tail)

    def finish_exchange(self, handle):
        return self.inner.finish_exchange(handle)


class UNetEngine:
    """Packed weights + kernel sequencing for one ``UNetModel``."""

    def __init__(self, unet, dtype: torch.dtype = torch.float16, device=None):
        """``unet=None``: an engine for stand-alone sub-modules (``vface_amd.module_exec``); ``device`` is then required."""
        self.unet = unet
        self.dtype = dtype
        self._device = torch.device(device) if device is not None else None
        self._packed: Dict[str, dict] = {}
        self._maps: Dict[tuple, torch.Tensor] = {}
        self._version = None
        # multi-GPU: a parallel.FrameShard (start_exchange / finish_exchange), installed by FrameShard.install
        self.halo_exchange = None
        self.halo_flow: Optional[torch.Tensor] = None  # flow from the previous rank's last frame into our frame 0
        self.halo_hw = None                            # (h, w) of the clip's flow fields (set with halo_exchange)
        self.exchange_events = None                    # a list: (start, end) HIP events around every finish_exchange (bench.py)
        # norm3 + FeedForward of the level-0 transformer blocks (C = 320; also the 64- / 128-channel test models) as one
        # activation-stationary kernel (csrc/ffn.hip).  VFACE_FUSE_FFN=0: the three-kernel path (A/B switch).
        self.fuse_ffn = os.environ.get("VFACE_FUSE_FFN", "1") != "0"
        # GroupNorm-apply -> proj_in -> LayerNorm -> attn1 projection of the level-0 SpatialTransformers (C = 320; also the 64- /
        # 128-channel test models) as one activation-stationary kernel (csrc/stfront.hip): four launches and four HBM round trips
        # of the token matrix less per block.  VFACE_FUSE_FRONT=0: the separate launches (A/B switch).
        self.fuse_front = os.environ.get("VFACE_FUSE_FRONT", "1") != "0"
        # attn1's out-projection (+ attn2's row bias + residual) in front of the fused FeedForward, one launch: the block's running
        # sum after attention never exists in HBM (csrc/ffn.hip, PRE form).  VFACE_FUSE_TAIL=0: GEMM + fused FeedForward (A/B).
        self.fuse_tail = os.environ.get("VFACE_FUSE_TAIL", "1") != "0"
        # ... and the SpatialTransformer's proj_out + x_in + column statistics behind it, still one launch (ffn.hip POST form)
        self.fuse_post = os.environ.get("VFACE_FUSE_POST", "1") != "0"
        # the time-embedding chain (timestep embedding, time_embed, every ResBlock's emb_layers) as three few-row launches
        self.fuse_temb = os.environ.get("VFACE_FUSE_TEMB", "1") != "0"
        # the UNet's `out` layer (GroupNorm -> SiLU -> conv3x3 to 4 channels) as one launch (csrc/outconv.hip).  VFACE_FUSE_OUT=0: A/B
        self.fuse_out = os.environ.get("VFACE_FUSE_OUT", "1") != "0"
        # the concat buffers' fp32 carrier over all columns, read by the output blocks' GroupNorm (the layout up to round 4); default: the
        # carrier covers the skip columns only and the concat GroupNorm reads the 16-bit copy (forward_nhwc).  VFACE_CONCAT32=1: A/B
        self.concat32 = os.environ.get("VFACE_CONCAT32", "0") == "1"
        self.interior16 = os.environ.get("VFACE_INTERIOR16", "1") != "0"      # (see _st; VFACE_INTERIOR16=0: fp32 interior sums, A/B)
        self._front_supported: Dict[tuple, bool] = {}
        self._ffn_supported: Dict[tuple, bool] = {}
        self.decompose_attn1 = False                   # bench.py's instrumented pass: vface_attn1_forward's launches call by call
        # the batch holds only the first `live_chunks` chunks of the hooks' three (the sampler's dead-branch elimination leaves
        # the recon third out: DDIMSampler.drop_dead_branches); None = every chunk is there
        self.live_chunks: Optional[int] = None
        # The batch is the sampler's own [x ; x ; inv_t] with t repeated (ddim_w_inv.py:632-655): chunks 0 and 1 enter the UNet
        # with IDENTICAL inputs and diverge only where attn2's row bias (the context) is first added.  The sampler states it per
        # call (DDIMSampler.p_sample_ddim_with_inverse; never true for a batch handed in through apply_model): the first
        # ResBlock and the first SpatialTransformer's front -- and, where the hook leaves chunk 1's q,k equal to chunk 0's
        # (no hook, `replace`, `fft`), its attention -- then run on 2F samples, chunk 0 reading chunk 1's rows (_shared_block).
        self.share_prefix = False
        # fp32 residual stream: residual sums are carried between kernels in fp32, 16-bit copies exist only
        # where a matrix-core operand needs them.  VFACE_STREAM32=0 restores the all-16-bit activations (A/B switch).
        self.stream32 = os.environ.get("VFACE_STREAM32", "1") != "0"
        # GroupNorm-apply + SiLU of a ResBlock fused into the patch-staged convolution's operand path (openaimodel.py:201-205,
        # 225-232 `GroupNorm32 -> SiLU -> conv`): "both" = in_layers and out_layers; "out" = out_layers only (the in_layers
        # normalisation then reads the fp32 carrier in its own pass instead of the 16-bit copy); "off" (default) = separate
        # gn_apply passes.  Exact (bit-identical to the separate pass on the same input) but MEASURED SLOWER on this kernel:
        # the ~70 vector instructions per 1-KiB patch piece sit in the K-tile period's critical path -- 28.22 vs 27.49 ms per
        # DDIM step (conv 9.46 vs 7.77 ms, gn_apply 0 vs 1.0 ms), DESIGN 4 -- so it is opt-in.
        self.fuse_gn = os.environ.get("VFACE_FUSE_GN", "off")
        # hipGraph replay of the UNet forward of a DDIM step (step_forward_nhwc): capture once per (batch, resolution, hook
        # configuration, context shape) and replay -- the default since round 3 (bit-equal to kernel-by-kernel launches, one
        # host call per step instead of ~1200; what bench.py times); VFACE_GRAPH=0 launches kernel by kernel.  The C ABI is
        # allocation-free and stream-ordered, so the captured graph is exactly the eager launch sequence.  Every cached graph
        # pins a private pool with the activations of one forward (~0.45 GB per frame at 512 x 512): the cache is bounded by
        # BYTES (VFACE_GRAPH_GB, default 64 of the 288 GB) and by count, least recently used first; a forward whose pool alone
        # exceeds the budget is not cached and runs kernel by kernel.
        self.use_graph = os.environ.get("VFACE_GRAPH", "1") != "0"
        self._graphs: "Dict[tuple, dict]" = {}
        self._graph_failed: set = set()       # keys whose capture failed: they run kernel by kernel, other keys still capture
        self.graph_capacity = 12        # (a split configuration holds three graphs: two halves and, for its self-check and timing, the whole batch)
        self.graph_budget_bytes = int(float(os.environ.get("VFACE_GRAPH_GB", "64")) * (1 << 30))
        # Two launch streams (VFACE_STREAMS=2, the default; 1 = one stream): a graph-replayed forward whose frames are not coupled
        # across the split (no hook, or replace / fft / mix: every edit stays inside a frame's own chunks) runs as two half-batches
        # -- frames [0, F/2) and [F/2, F) of every chunk -- on two HIP streams at once.  The kernels that own a whole CU per
        # workgroup run their HBM phases in lock-step across the chip; two independent launch sequences put one
        # half's HBM-bound launches beside the other's matrix-bound ones.  Every kernel is batch-invariant, so the halves' results
        # are the full batch's bit for bit (tests); measured 1-4.5 % per step depending on the box (profiles/r04_n).
        self.split_streams = int(os.environ.get("VFACE_STREAMS", "2"))
        self._split_state: "Dict[tuple, dict]" = {}
        self._split_pair = None
        self._split_verified = False
        self.split_overlap = None      # step time / (half A + half B) of the measured split step: ~0.5 = the halves ran at once
        # Self-check of the two-sequence form: the SECOND split step of every split configuration kind ("free": halves
        # that never read each other; "coupled": flow_fix's halves handing one frame over, parallel.StreamShard) is also run as ONE
        # launch sequence over the whole batch and the two eps compared bit for bit (one extra forward per kind and engine).  A
        # mismatch -- round 4 / 5: a kernel of one half mis-executing beside the other half's attention waves -- makes the engine
        # return the single sequence's result and stay on one launch sequence for good.  VFACE_SPLIT_SELFCHECK=0 skips it.
        self.split_selfcheck = os.environ.get("VFACE_SPLIT_SELFCHECK", "1") != "0"
        self.split_checked: Dict[str, bool] = {}       # kind -> the halves' eps equalled the single sequence's
        self.split_timing: Dict[tuple, tuple] = {}     # (kind, samples, H, W) -> (ms of a two-sequence step, ms of a one-sequence step)
        self._split_off: set = set()                   # ... the configurations that stay on one launch sequence because it measured faster
        hip.load()

    # ------------------------------------------------------------------ weights
    @property
    def device(self):
        return self._device if self._device is not None else next(self.unet.parameters()).device

    def _w16(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(device=self.device, dtype=self.dtype).contiguous()

    def _f32(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().to(device=self.device, dtype=torch.float32).contiguous()

    def pack(self):
        """(Re)build every packed device weight from the module parameters."""
        u = self.unet
        if not next(u.parameters()).is_cuda:
            raise hip.VFaceHipError("UNetModel parameters are not on the GPU: the VFace path has no CPU fallback")
        P = self._packed = {}
        sd = {k: v.detach() for k, v in u.state_dict().items()}
        cpu = lambda k: sd[k].float().cpu()

        conv3 = lambda prefix: self.pack_conv3(sd, prefix)
        lin = lambda prefix, bias=True, conv=False: self.pack_lin(sd, prefix, bias, conv)

        P["time_embed.0"] = lin("time_embed.0")
        P["time_embed.2"] = lin("time_embed.2")
        emb_w, emb_b, off = [], [], 0
        vcat_w, voff = [], 0
        for kind, prefix, mod in u.layer_table():
            if kind == "conv":
                P[prefix] = conv3(prefix)
            elif kind == "down":
                P[prefix] = conv3(prefix + ".op")
            elif kind == "up":
                P[prefix] = self.pack_up(sd, prefix + ".conv")
            elif kind == "res":
                d = self.pack_res(sd, prefix)
                cout = d["conv1"]["cout"]
                emb_w.append(sd[prefix + ".emb_layers.1.weight"].float())
                emb_b.append(sd[prefix + ".emb_layers.1.bias"].float())
                d["emb_slice"] = (off, off + cout)
                off += cout
                P[prefix] = d
            elif kind == "st":
                d = self.pack_st(sd, prefix)
                d["a2_slice"] = (voff, voff + d["c"])
                vcat_w.append(sd[prefix + ".transformer_blocks.0.attn2.to_v.weight"].float())
                voff += d["c"]
                P[prefix] = d
        P["emb_all"] = {"w": self._w16(torch.cat(emb_w, 0)), "b": self._f32(torch.cat(emb_b, 0)), "n": off}
        P["a2_v_all"] = {"w": self._w16(torch.cat(vcat_w, 0)), "n": voff}
        P["out.gn"] = (self._f32(sd["out.0.weight"]), self._f32(sd["out.0.bias"]))
        P["out.conv"] = conv3("out.2")
        self._version = self._param_version()

    # ---- per-layer packers (also used for stand-alone sub-modules, vface_amd/module_exec.py); `sd`: name -> tensor
    def pack_conv3(self, sd, prefix):
        w = sd[prefix + ".weight"]
        d = {"w": self._w16(packing.pack_conv3x3(w.detach().float().cpu())), "b": self._f32(sd[prefix + ".bias"]),
             "cin": w.shape[1], "cinp": (w.shape[1] + 7) // 8 * 8, "cout": w.shape[0]}
        return d

    def pack_up(self, sd, prefix):
        d = self.pack_conv3(sd, prefix)
        # nearest x2 + conv3x3 = four parity-phase 2x2 convs with pre-summed taps (4/9 of the multiply-adds)
        d["phases"] = self._w16(packing.pack_upsample_phases(sd[prefix + ".weight"].detach().float().cpu()))
        return d

    def pack_lin(self, sd, prefix, bias=True, conv=False):
        w = sd[prefix + ".weight"]
        w = w.reshape(w.shape[0], w.shape[1]) if conv else w
        return {"w": self._w16(w), "b": self._f32(sd[prefix + ".bias"]) if bias else None}

    def pack_res(self, sd, prefix):
        d = {"in_gn": (self._f32(sd[prefix + ".in_layers.0.weight"]), self._f32(sd[prefix + ".in_layers.0.bias"])),
             "conv1": self.pack_conv3(sd, prefix + ".in_layers.2"),
             "out_gn": (self._f32(sd[prefix + ".out_layers.0.weight"]), self._f32(sd[prefix + ".out_layers.0.bias"])),
             "conv2": self.pack_conv3(sd, prefix + ".out_layers.3")}
        if (prefix + ".skip_connection.weight") in sd:
            d["skip"] = self.pack_lin(sd, prefix + ".skip_connection", conv=True)
            wsk = sd[prefix + ".skip_connection.weight"]
            if d["conv2"]["cinp"] % 64 == 0 and wsk.shape[1] % 64 == 0:
                # second conv + 1x1 shortcut in one K loop (vface_conv3x3_plus_1x1): weights side by side, biases summed
                d["conv2_skip"] = dict(d["conv2"], c2=wsk.shape[1],
                                       w=torch.cat([d["conv2"]["w"], d["skip"]["w"]], 1).contiguous(),
                                       b=(d["conv2"]["b"] + d["skip"]["b"]).contiguous())
        return d

    def pack_block(self, sd, t):
        """BasicTransformerBlock at state-dict prefix ``t``: attn1 / ff / norms (+ attn2's out projection; its to_v is
        stacked with the other blocks' by the UNet packer)."""
        cpu = lambda k: sd[k].detach().float().cpu()
        ffw, ffb = packing.pack_geglu(cpu(t + ".ff.net.0.proj.weight"), cpu(t + ".ff.net.0.proj.bias"))
        return {"ln1": (self._f32(sd[t + ".norm1.weight"]), self._f32(sd[t + ".norm1.bias"])),
                "ln3": (self._f32(sd[t + ".norm3.weight"]), self._f32(sd[t + ".norm3.bias"])),
                "wqkv": self._w16(packing.pack_qkv(sd[t + ".attn1.to_q.weight"], sd[t + ".attn1.to_k.weight"],
                                                  sd[t + ".attn1.to_v.weight"])),
                "wo": self.pack_lin(sd, t + ".attn1.to_out.0"),
                "ff1": {"w": self._w16(ffw), "b": self._f32(ffb)}, "ff2": self.pack_lin(sd, t + ".ff.net.2"),
                # fused FeedForward (csrc/ffn.hip): ff.net[2] in the order GEMM 1's accumulators hand the hidden units over --
                # packed only for the widths that kernel takes (a C = 1280 block would hold 13 MB of dead copy)
                "ff2p": (self._w16(packing.pack_ffn_w2(cpu(t + ".ff.net.2.weight")))
                         if self.fuse_ffn and hip.ffn_fused_width_supported(sd[t + ".norm1.weight"].shape[0]) else None),
                # the same kernel with attn1's out-projection in front (vface_attn_out_ffn_fused): [to_out ; ff.net[0] (k permuted)]
                "tail_w": (self._w16(packing.pack_attn_out_ffn(cpu(t + ".attn1.to_out.0.weight"), ffw))
                           if self.fuse_ffn and self.fuse_tail and hip.ffn_fused_width_supported(sd[t + ".norm1.weight"].shape[0])
                           else None),
                "a2_out": self.pack_lin(sd, t + ".attn2.to_out.0"), "c": sd[t + ".norm1.weight"].shape[0],
                "wlin": {}, "attn1_name": t + ".attn1",
                "qk_src": (sd[t + ".attn1.to_q.weight"], sd[t + ".attn1.to_k.weight"])}

    def pack_st(self, sd, prefix):
        d = self.pack_block(sd, prefix + ".transformer_blocks.0")
        d.update({"gn": (self._f32(sd[prefix + ".norm.weight"]), self._f32(sd[prefix + ".norm.bias"])),
                  "proj_in": self.pack_lin(sd, prefix + ".proj_in", conv=True),
                  "proj_out": self.pack_lin(sd, prefix + ".proj_out", conv=True)})
        c = d["c"]
        d["front_w"] = None
        if self.fuse_front and hip.st_front_supported(128, c, 128):      # (the widths csrc/stfront.hip takes)
            t = prefix + ".transformer_blocks.0.attn1"
            w_in = sd[prefix + ".proj_in.weight"].detach().float().cpu().reshape(c, c)
            w_p = packing.pack_qkv(sd[t + ".to_q.weight"], sd[t + ".to_k.weight"], sd[t + ".to_v.weight"]).detach().float().cpu()
            d["front_w"] = self._w16(packing.pack_st_front(w_in, w_p))
        d["tail_post"] = False
        if d.get("tail_w") is not None and self.fuse_post:
            # proj_out's rows behind the tail's weight stream (vface_attn_out_ffn_proj_fused), k columns in the stream's order
            w_po = sd[prefix + ".proj_out.weight"].detach().float().cpu().reshape(c, c)
            d["tail_w"] = torch.cat([d["tail_w"], self._w16(w_po[:, packing.ffn_w2_perm(c)])], 0).contiguous()
            d["tail_post"] = True
        return d

    def _param_version(self):
        return tuple(p._version for p in self.unet.parameters())

    def _ensure_packed(self):
        if not self._packed or self._version != self._param_version():
            self.pack()

    def _wlin(self, st: dict, kind: str, param: float) -> torch.Tensor:
        key = (kind, round(float(param), 9))
        if key not in st["wlin"]:
            wq, wk = (t.detach().float().cpu() for t in st["qk_src"])
            w = packing.fold_fsai(wq, wk, param) if kind == "fsai" else packing.fold_mix(wq, wk, param)
            st["wlin"][key] = self._w16(w)
        return st["wlin"][key]

    def _map(self, kind: str, B: int, c: int) -> torch.Tensor:
        key = (kind, B, c)
        if key not in self._maps:
            if kind == "qk_replace":
                m = torch.arange(B, dtype=torch.int32) % c
            elif kind in ("share_qk", "share_v"):
                # _st_front_shared with a warp: slot 0 = chunk 1's warped q|k, slot 1 = chunk 0's q|k and the v of chunks 0, 1
                idx = torch.arange(B, dtype=torch.int32)
                m = torch.where(idx < c, idx + c, idx)            # chunk 0 reads slot 1 (q|k and v)
                if kind == "share_qk":
                    m = torch.where((idx >= c) & (idx < 2 * c), idx - c, m)      # chunk 1's q|k: slot 0 (its v stays in slot 1 = itself)
            else:  # v_fixed: chunk 0 identity, chunk k >= 1 -> its first frame
                idx = torch.arange(B, dtype=torch.int32)
                m = torch.where(idx < c, idx, (idx // c) * c)
            self._maps[key] = m.to(self.device)
        return self._maps[key]

    # ------------------------------------------------------------------ primitive steps
    def _new(self, rows: int, cols: int, dtype=None) -> torch.Tensor:
        return torch.empty(rows, cols, dtype=dtype or self.dtype, device=self.device)

    def _new_target(self, rows: int, cols: int, hw: int, need16: bool = True):
        """A fresh output ``(16-bit buffer | None, column statistics | None, fp32 carrier | None)``: statistics when the
        image size allows (hw % 64 == 0); the carrier when the fp32 residual stream is on (and then the 16-bit copy only
        if a matrix-core operand will read it)."""
        s32 = self.stream32 and cols % 8 == 0
        return (self._new(rows, cols) if (need16 or not s32) else None, self._new_cs(rows, cols, hw),
                self._new(rows, cols, torch.float32) if s32 else None)

    def _new_cs(self, rows: int, cols: int, hw: int) -> Optional[torch.Tensor]:
        if hw % 64 or cols % 4:
            return None
        return torch.empty(rows // 64, cols, 2, dtype=torch.float32, device=self.device)

    def _gemm(self, a: torch.Tensor, w: dict, out: Optional[torch.Tensor], hw: int = 0, **kw):
        K = a.shape[1]
        if hw > 1 and "rowbias" not in kw:
            kw["rows_per_sample"] = hw   # split-K decided per sample: a frame's bits do not depend on its batch
        hip.gemm(a, w["w"], out, M=a.shape[0], N=w["w"].shape[0], K=K, lda=a.stride(0),
                 ldc=out.stride(0) if out is not None else 0, ldw=w["w"].shape[1], bias=w.get("b"), **kw)

    @staticmethod
    def _resid(x) -> dict:
        """Residual operand of an epilogue: the fp32 carrier when the tensor has one."""
        if isinstance(x, Act):
            if x.t32 is not None:
                return {"residual32": x.t32}
            return {"residual": x.t, "ldr": x.t.stride(0)}
        if x.dtype == torch.float32:
            return {"residual32": x}
        return {"residual": x, "ldr": x.stride(0)}

    def _gn(self, x: Act, gn, eps: float, silu: bool) -> Act:
        # (the fp32 carrier where there is one.  Reading the 16-bit copy in EVERY ResBlock -- 2 B per element instead of 4 -- was
        #  measured in round 5: -0.1 ms of an 83 ms step for +0.6 % of the error budget, profiles/r05_h: not taken; the concat
        #  GroupNorms, whose carrier had no other reader, do read 16 bits: forward_nhwc)
        src = x.src
        if x.cs is not None:
            st = hip.groupnorm_stats_from_cols(x.cs, nimg=x.N, hw=x.hw, C_=x.C, eps=eps)
        else:
            st = hip.groupnorm_stats(src, nimg=x.N, hw=x.hw, C_=x.C, ldx=src.stride(0), eps=eps)
        y = self._new(x.M, x.C)
        hip.groupnorm_apply(src, st, gn[0], gn[1], y, nimg=x.N, hw=x.hw, C_=x.C, ldx=src.stride(0), ldy=x.C, silu=silu)
        return Act(y, x.N, x.H, x.W)

    def _conv(self, x: Act, w: dict, tgt, stride=1, upsample=False, rowbias=None, residual=None, out_f32=False,
              stream=True, gn_ab=None) -> Act:
        """``tgt``: None (allocate) or ``(16-bit out view | None, colstats view | None, fp32 carrier view | None)``.
        ``residual``: an ``Act`` / tensor added in the epilogue.  ``stream=False``: a branch activation (consumed by one
        GEMM / GroupNorm only): 16-bit output, no fp32 carrier."""
        VH, VW = (2 * x.H, 2 * x.W) if upsample else (x.H, x.W)
        OH, OW = (VH - 1) // stride + 1, (VW - 1) // stride + 1
        if tgt is None:
            if out_f32:
                out, cs, o32 = self._new(x.N * OH * OW, w["cout"], torch.float32), None, None
            elif not stream:
                out, cs, o32 = self._new(x.N * OH * OW, w["cout"]), self._new_cs(x.N * OH * OW, w["cout"], OH * OW), None
            else:
                out, cs, o32 = self._new_target(x.N * OH * OW, w["cout"], OH * OW)
        else:
            out, cs, o32 = tgt
        assert x.C == w["cinp"], (x.C, w["cinp"])
        ldy = out.stride(0) if out is not None else 0
        if upsample and "phases" in w and _phase_form_pays(x.H * x.W, w["cout"]) and residual is None and not out_f32 \
                and (cs is None or (x.H * x.W) % 64 == 0):
            hip.upsample2x_conv3x3(x.t, w["phases"], out, nimg=x.N, H=x.H, W=x.W, cin=w["cinp"], cout=w["cout"], ldx=x.ld,
                                   ldy=ldy, bias=w["b"], rowbias=rowbias, colstats=cs, out32=o32)
            return Act(out, x.N, OH, OW, cs, o32)
        hip.conv3x3(x.t, w["w"], out, nimg=x.N, H=x.H, W=x.W, cin=w["cinp"], cout=w["cout"], ldx=x.ld,
                    ldy=ldy, stride=stride, upsample=upsample, bias=w["b"], rowbias=rowbias,
                    flags=hip.EPI_OUT_F32 if out_f32 else 0, colstats=cs, out32=o32, gn_ab=gn_ab, gn_silu=gn_ab is not None,
                    **(self._resid(residual) if residual is not None else {}))
        return Act(out, x.N, OH, OW, cs, o32)

    def _gn_fusable(self, x: Act, w: dict, which: str) -> bool:
        """Can GroupNorm-apply + SiLU of ``x`` ride in the operand path of the 3x3 convolution ``w``?  Needs producer-side
        column statistics, a 16-bit copy of ``x`` and a launch that runs the patch-staged kernel."""
        mode = self.fuse_gn
        if w["cout"] % 128:               # the fused form exists in the 128-wide tile only (the 160-wide one spilled: not built since round 6)
            return False
        if mode.endswith("128"):          # (the spelling of rounds 4-5, when "out" / "both" also took the spilled 160-wide form)
            mode = mode[:-3]
        if mode == "off" or (which == "in" and mode != "both"):
            return False
        return x.cs is not None and x.t is not None and x.C == w["cinp"] and \
            hip.conv_uses_patch_kernel(x.H, x.W, w["cinp"], w["cout"], 3, 1, False) == 1

    def _res(self, x: Act, p: dict, emb_all: torch.Tensor, out) -> Act:
        """ResBlock._forward (openaimodel.py:255-275), non-updown, no scale-shift."""
        a, b = p["emb_slice"]
        if self._gn_fusable(x, p["conv1"], "in"):
            ab = hip.groupnorm_coeffs_from_cols(x.cs, p["in_gn"][0], p["in_gn"][1], nimg=x.N, hw=x.hw, C_=x.C, eps=1e-5)
            h = self._conv(x, p["conv1"], None, rowbias=emb_all[:, a:b], stream=False, gn_ab=ab)
        else:
            h = self._gn(x, p["in_gn"], 1e-5, True)
            h = self._conv(h, p["conv1"], None, rowbias=emb_all[:, a:b], stream=False)
        gn2 = None
        if self._gn_fusable(h, p["conv2"], "out"):
            gn2 = hip.groupnorm_coeffs_from_cols(h.cs, p["out_gn"][0], p["out_gn"][1], nimg=h.N, hw=h.hw, C_=h.C, eps=1e-5)
        else:
            h = self._gn(h, p["out_gn"], 1e-5, True)
        if "conv2_skip" in p and os.environ.get("VFACE_NO_SKIP_FUSION") != "1":   # (env: A/B switch for measurements)
            w = p["conv2_skip"]
            o, cs, o32 = self._new_target(x.M, w["cout"], x.hw) if out is None else out
            hip.conv3x3_plus_1x1(h.t, x.t, w["w"], o, nimg=x.N, H=x.H, W=x.W, cin=w["cinp"], c2=w["c2"], cout=w["cout"],
                                 ldx=h.ld, ldx2=x.ld, ldy=o.stride(0) if o is not None else 0, bias=w["b"], colstats=cs,
                                 out32=o32, gn_ab=gn2, gn_silu=gn2 is not None)
            return Act(o, x.N, x.H, x.W, cs, o32)
        if "skip" in p:
            if self.stream32 and p["conv2"]["cout"] % 8 == 0:
                skip = self._new(x.M, p["conv2"]["cout"], torch.float32)
                self._gemm(x.t, p["skip"], None, hw=x.H * x.W, out32=skip)
            else:
                skip = self._new(x.M, p["conv2"]["cout"])
                self._gemm(x.t, p["skip"], skip, hw=x.H * x.W)
        else:
            skip = x
        return self._conv(h, p["conv2"], out, residual=skip, gn_ab=gn2)

    def _attn1(self, xln: torch.Tensor, resid: torch.Tensor, p: dict, cfg: Optional[HookCfg], a2vec: torch.Tensor,
               N: int, n: int, heads: int, hw) -> torch.Tensor:
        """``resid`` 16-bit -> 16-bit result; ``resid`` fp32 (the residual stream) -> fp32 result, no 16-bit copy."""
        d = p["c"]
        s32 = resid.dtype == torch.float32
        out = None if s32 else self._new(N * n, d)
        out32 = self._new(N * n, d, torch.float32) if s32 else None
        res_kw = {"residual32": resid, "out32": out32} if s32 else {"residual": resid, "ldr": resid.stride(0)}
        pl = plan_fusion(cfg, N, n, self.halo_hw if self.halo_exchange is not None else None, self.live_chunks)
        if pl["staged"]:
            if self.halo_exchange is not None:
                raise NotImplementedError(f"fusion={pl['staged']!r} couples frames beyond one neighbour (temporal: +-2 "
                                          "frames; adaIn: a global std) and is not sharded across GPUs")
            kw = {"residual32": resid, "out32": out32} if s32 else {"residual": resid}
            return staged_attn1(xln, p["wqkv"], p["wo"]["w"], p["wo"]["b"], out, B=N, n=n, d=d, heads=heads,
                                mode=pl["staged"], rowbias=a2vec, chunks=pl["chunks"], **kw)
        fusion, chunks, flow, alpha, v_fixed = pl["fusion"], pl["chunks"], pl["flow"], pl["alpha"], pl["v_fixed"]
        wlin = self._wlin(p, *pl["wlin"]) if pl["wlin"] else None
        qk_map = self._map("qk_replace", N, N // chunks) if fusion == hip.FUSION_REPLACE else None
        v_map = self._map("v_fixed", N, N // chunks) if v_fixed else None
        ws = torch.empty(hip.attn1_workspace_bytes(N, n, d, chunks), dtype=torch.uint8, device=self.device)
        hw = pl["warp_hw"]
        if hw is not None and self.halo_exchange is not None:
            # every rank of a sharded clip takes part in the boundary exchange, a one-frame shard (no local field) too
            return self._attn1_sharded(xln, res_kw, p, wlin, a2vec, N, n, heads, flow, hw, alpha, out, chunks)
        if self.decompose_attn1 or pl["hook_chunks"] != chunks:
            # (a batch without its last chunk: the shared-score attention must run the FULL hook's instantiation with fewer live
            # sets to keep the bits of the full batch -- vface_attn1_forward only knows the chunks it is handed)
            self._attn1_decomposed(xln, p, wlin, a2vec, N, n, heads, chunks, fusion, v_fixed, flow if hw is not None else None,
                                   hw, alpha, qk_map, v_map, out, res_kw, pl["hook_chunks"])
            return out32 if s32 else out
        hip.attn1_forward(xln, p["wqkv"], wlin, p["wo"]["w"], p["wo"]["b"], out, B=N, n=n, d=d, heads=heads,
                          chunks=chunks, fusion=fusion, ldx=xln.stride(0), ldo=d, workspace=ws,
                          rowbias=a2vec, v_fixed=v_fixed, flow=flow if hw is not None else None,
                          h=hw[0] if hw else 0, w=hw[1] if hw else 0,
                          alpha=alpha, qk_map=qk_map, v_map=v_map, **res_kw)
        return out32 if s32 else out

    def _attn1_decomposed(self, xln, p, wlin, a2vec, N, n, heads, chunks, fusion, v_fixed, flow, hw, alpha, qk_map, v_map,
                          out, res_kw, hook_chunks=None):
        """The launch sequence of ``vface_attn1_forward`` (capi.cpp) issued call by call from here -- the same kernels with the
        same parameters in the same order, hence the same bits (tests/test_kernels_gpu.py) -- so that ``bench.py``'s
        instrumented pass can put HIP events around the projections and the attention kernel separately."""
        d = p["c"]
        F_ = N // chunks
        Fn = F_ * n
        ldx = xln.stride(0)
        qkv = self._new(N * n, 3 * d)
        g = lambda a, w, o, M, Nn, K, **kw: hip.gemm(a, w, o, M=M, N=Nn, K=K, lda=ldx, ldc=o.stride(0), ldw=w.stride(0),
                                                      split_k=False, **kw)
        if fusion == hip.FUSION_NONE:
            g(xln, p["wqkv"], qkv, N * n, 3 * d, d)
        else:
            g(xln, p["wqkv"], qkv, Fn, 3 * d, d)
            g(xln[Fn:], p["wqkv"][2 * d:], qkv[Fn:, 2 * d:], N * n - Fn, d, d)
            if fusion == hip.FUSION_LINEAR:
                T = self._new(Fn, 2 * d) if flow is not None else None
                for c in range(1, chunks):
                    dst = T if (flow is not None and c == 1) else qkv[c * Fn:(c + 1) * Fn, :2 * d]
                    g(xln[c * Fn:], wlin, dst, Fn, 2 * d, 2 * d, a2=xln, lda2=ldx, k1=d)
                if flow is not None:
                    hip.flow_warp(T, qkv[Fn:2 * Fn, :2 * d], flow, F=F_, h=hw[0], w=hw[1], C_=2 * d, ld_src=2 * d,
                                  fs_src=n * 2 * d, ld_dst=3 * d, fs_dst=n * 3 * d, alpha=alpha)
        att = self._new(N * n, d)
        kw = dict(heads=heads, n=n, nk=n, dh=d // heads, ldq=3 * d, ldk=3 * d, ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d,
                  bsv=n * 3 * d, ldo=d, bso=n * d,
                  scale=float(np.float32(1.0) / np.sqrt(np.float32(d // heads))))   # fp32 arithmetic, as capi.cpp computes it
        hc = hook_chunks or chunks
        if fusion == hip.FUSION_REPLACE and chunks > 1 and hip.load().vface_attention_shared_scores_supported(d // heads, hc):
            hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=F_, v_map=v_map if v_fixed else None, v_sets=hc,
                          v_sets_live=chunks, set_stride=F_, **kw)
        else:
            hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=N, qk_map=qk_map if fusion == hip.FUSION_REPLACE else None,
                          v_map=v_map if v_fixed else None, **kw)
        o = out if out is not None else res_kw["out32"]
        hip.gemm(att, p["wo"]["w"], out, M=N * n, N=d, K=d, lda=d, ldc=out.stride(0) if out is not None else 0,
                 bias=p["wo"]["b"], rowbias=a2vec, rows_per_sample=n, split_k=False, **res_kw)
        return o

    def _attn1_sharded(self, xln, res_kw, p, wlin, a2vec, N, n, heads, flow, hw, alpha, out, chunks=3):
        """flow_fix with frames sharded across ranks: the same kernels as vface_attn1_forward, sequenced here
        so the one-neighbour boundary exchange sits between the fused projection and the warp."""
        d = p["c"]
        F_ = N // chunks
        Fn = F_ * n
        qkv = self._new(N * n, 3 * d)
        T = self._new(Fn, 2 * d)
        def fused(c, dst):
            hip.gemm(xln[c * Fn:], wlin, dst, M=Fn, N=2 * d, K=2 * d, lda=xln.stride(0), ldc=dst.stride(0), ldw=2 * d,
                     a2=xln, lda2=xln.stride(0), k1=d)

        fused(1, T)
        # my last frame's fused q|k goes to the next rank; the previous rank's arrives while chunk 0 / chunk 2 /
        # the v projections below are computed
        handle = self._halo_start(T[(F_ - 1) * n:])
        hip.gemm(xln, p["wqkv"], qkv, M=Fn, N=3 * d, K=d, lda=xln.stride(0), ldc=3 * d)
        hip.gemm(xln[Fn:], p["wqkv"][2 * d:], qkv[Fn:, 2 * d:], M=N * n - Fn, N=d, K=d, lda=xln.stride(0), ldc=3 * d)
        for ch in range(2, chunks):
            fused(ch, qkv[ch * Fn:(ch + 1) * Fn, :2 * d])
        ev = self.exchange_events if not isinstance(self.halo_exchange, _GraphSegments) else None
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        halo = self.halo_exchange.finish_exchange(handle)
        if ev is not None:
            e1.record()
            ev.append((e0, e1))
        dst = qkv[Fn:2 * Fn, :2 * d]
        hip.flow_warp(T, dst, flow, F=F_, h=hw[0], w=hw[1], C_=2 * d, ld_src=2 * d, fs_src=n * 2 * d,
                      ld_dst=3 * d, fs_dst=n * 3 * d, alpha=alpha, prev=halo, ld_prev=2 * d,
                      flow_prev=self.halo_flow if halo is not None else None)
        att = self._new(N * n, d)
        hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=N, heads=heads, n=n, nk=n, dh=d // heads, ldq=3 * d,
                      ldk=3 * d, ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d, bsv=n * 3 * d, ldo=d, bso=n * d,
                      scale=(d // heads) ** -0.5)
        hip.gemm(att, p["wo"]["w"], out, M=N * n, N=d, K=d, lda=d, ldc=d, bias=p["wo"]["b"], rowbias=a2vec,
                 rows_per_sample=n, **res_kw)
        return res_kw.get("out32") if out is None else out

    def _block(self, t0: torch.Tensor, p: dict, attn1, a2vec: torch.Tensor, N: int, n: int, hw, want32: bool = False):
        """BasicTransformerBlock._forward (attention.py:239-243) on the block's running sum ``t0`` ``[N*n, c]`` (fp32 when the
        residual stream is on, else 16-bit): returns its last value as the 16-bit operand of the next projection (and, if
        ``want32``, as fp32 too).  ``a2vec``: the single-token cross-attention's contribution, fp32 ``[N, c]``."""
        c, M = p["c"], t0.shape[0]
        ln = self._new(M, c)
        hip.layernorm(t0, p["ln1"][0], p["ln1"][1], ln, M=M, C_=c, ldx=c, ldy=c)
        cfg = getattr(attn1, "_vface_cfg", None)
        fw = attn1.__dict__.get("forward")
        if fw is not None and not getattr(fw, "_vface", False):
            raise hip.VFaceHipError("attn1.forward was replaced by a closure this engine does not know; use "
                                    "vface_amd.ldm.models.pnp_utils.register_spa_attn_injection")
        t1 = self._attn1(ln, t0, p, cfg, a2vec, N, n, attn1.heads, hw)
        return self._ffn(t1, p, n, want32)

    def _ffn(self, t1: torch.Tensor, p: dict, n: int, want32: bool = False):
        """``x + ff(norm3(x))`` (attention.py:243) on the block's running sum ``t1`` (fp32 with the residual stream, else 16-bit)."""
        c, M = p["c"], t1.shape[0]
        t2 = self._new(M, c)
        t2_32 = self._new(M, c, torch.float32) if want32 else None
        if self.fuse_ffn and t1.dtype == torch.float32 and p["ff2p"] is not None and self._ffn_ok(M, c):
            # norm3 -> ff.net[0] (GEGLU) -> ff.net[2] -> + x in ONE launch (csrc/ffn.hip): the [M, 4c] hidden matrix never exists
            hip.ffn_fused(t1, p["ln3"][0], p["ln3"][1], p["ff1"]["w"], p["ff1"]["b"], p["ff2p"], p["ff2"]["b"], t2, M=M, C_=c,
                          out32=t2_32)
            return (t2, t2_32) if want32 else t2
        ln = self._new(M, c)
        hip.layernorm(t1, p["ln3"][0], p["ln3"][1], ln, M=M, C_=c, ldx=c, ldy=c)
        ff = self._new(M, 4 * c)
        hip.gemm(ln, p["ff1"]["w"], ff, M=M, N=8 * c, K=c, lda=c, ldc=4 * c, bias=p["ff1"]["b"], flags=hip.EPI_GEGLU)
        self._gemm(ff, p["ff2"], t2, hw=n, out32=t2_32, **self._resid(t1))
        return (t2, t2_32) if want32 else t2

    def _st_front(self, x: Act, p: dict, attn1, a2vec: torch.Tensor, post=None):
        """The SpatialTransformer up to and including its transformer block, with the FRONT -- GroupNorm-apply, proj_in,
        LayerNorm (norm1) and attn1's projection -- as ONE launch (csrc/stfront.hip) instead of four: needs the fp32 carrier
        and the producer's column statistics of ``x`` and a width the kernel takes.  Returns the block's last running sum
        (16-bit, proj_out's operand), or None when this layer does not qualify (the caller then runs the separate launches), or
        True when ``post = (out16 | None, colstats | None, out32)`` was given and the tail launch also ran proj_out + ``x`` into it.
        What follows the front is the launch sequence of ``vface_attn1_forward`` (capi.cpp) minus its first two GEMMs: the
        dual-source projections of the hook's linear fusions (they read the LayerNorm output the front also writes then), the flow
        warp -- with the boundary exchange between chunk 1's fused projection and the warp when frames are sharded --, the
        attention kernel, the out-projection into the fp32 stream, then norm3 + FeedForward."""
        c, N, n, M = p["c"], x.N, x.hw, x.M
        if not self.fuse_front or p.get("front_w") is None or x.t32 is None or x.cs is None:
            return None
        key = (M, c, n)
        ok = self._front_supported.get(key)
        if ok is None:
            ok = self._front_supported[key] = bool(hip.st_front_supported(M, c, n))
        if not ok:
            return None
        cfg = getattr(attn1, "_vface_cfg", None)
        fw = attn1.__dict__.get("forward")
        if fw is not None and not getattr(fw, "_vface", False):
            raise hip.VFaceHipError("attn1.forward was replaced by a closure this engine does not know; use "
                                    "vface_amd.ldm.models.pnp_utils.register_spa_attn_injection")
        pl = plan_fusion(cfg, N, n, self.halo_hw if self.halo_exchange is not None else None, self.live_chunks)
        if pl["staged"]:
            return None      # "temporal" / "adaIn" edit a full q,k,v buffer with their own kernels
        d, heads = c, attn1.heads
        fusion, chunks, flow, alpha, v_fixed = pl["fusion"], pl["chunks"], pl["flow"], pl["alpha"], pl["v_fixed"]
        F_ = N // chunks
        Fn = F_ * n
        hw = pl["warp_hw"]
        sharded = hw is not None and self.halo_exchange is not None
        if hw is None:
            flow = None
        ab = hip.groupnorm_coeffs_from_cols(x.cs, p["gn"][0], p["gn"][1], nimg=N, hw=n, C_=c, eps=1e-6)
        t0 = self._new(M, c, torch.float32)
        qkv = self._new(M, 3 * d)
        ln = self._new(M, c) if fusion == hip.FUSION_LINEAR else None
        hip.st_front(x.t32, ab, p["front_w"], p["proj_in"]["b"], p["ln1"][0], p["ln1"][1], t0, qkv, M=M, C_=c, hw=n, NQ=3 * d,
                     rows_full=M if fusion == hip.FUSION_NONE else Fn, nq_lo=0 if fusion == hip.FUSION_NONE else 2 * d, ln=ln)
        if fusion == hip.FUSION_LINEAR:
            wlin = self._wlin(p, *pl["wlin"])
            ldl = ln.stride(0)

            def fused(ch, dst):
                hip.gemm(ln[ch * Fn:], wlin, dst, M=Fn, N=2 * d, K=2 * d, lda=ldl, ldc=dst.stride(0), ldw=2 * d, a2=ln, lda2=ldl, k1=d,
                         split_k=False)
            T = self._new(Fn, 2 * d) if (flow is not None or sharded) else None
            halo = None
            if sharded:
                fused(1, T)
                handle = self._halo_start(T[(F_ - 1) * n:])
                for ch in range(2, chunks):
                    fused(ch, qkv[ch * Fn:(ch + 1) * Fn, :2 * d])
                ev = self.exchange_events if not isinstance(self.halo_exchange, _GraphSegments) else None
                if ev is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                halo = self.halo_exchange.finish_exchange(handle)
                if ev is not None:
                    e1.record()
                    ev.append((e0, e1))
            else:
                for ch in range(1, chunks):
                    fused(ch, T if (T is not None and ch == 1) else qkv[ch * Fn:(ch + 1) * Fn, :2 * d])
            if T is not None:
                hip.flow_warp(T, qkv[Fn:2 * Fn, :2 * d], flow, F=F_, h=hw[0], w=hw[1], C_=2 * d, ld_src=2 * d, fs_src=n * 2 * d,
                              ld_dst=3 * d, fs_dst=n * 3 * d, alpha=alpha, prev=halo, ld_prev=2 * d,
                              flow_prev=self.halo_flow if halo is not None else None)
        att = self._new(M, d)
        kw = dict(heads=heads, n=n, nk=n, dh=d // heads, ldq=3 * d, ldk=3 * d, ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d,
                  bsv=n * 3 * d, ldo=d, bso=n * d,
                  scale=float(np.float32(1.0) / np.sqrt(np.float32(d // heads))))   # fp32 arithmetic, as capi.cpp computes it
        v_map = self._map("v_fixed", N, F_) if v_fixed else None
        if fusion == hip.FUSION_REPLACE and chunks > 1 and \
                hip.load().vface_attention_shared_scores_supported(d // heads, pl["hook_chunks"]):
            hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=F_, v_map=v_map, v_sets=pl["hook_chunks"], v_sets_live=chunks,
                          set_stride=F_, **kw)
        else:
            hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=N,
                          qk_map=self._map("qk_replace", N, F_) if fusion == hip.FUSION_REPLACE else None, v_map=v_map, **kw)
        if self.fuse_tail and p.get("tail_w") is not None and n % 128 == 0 and self._ffn_ok(M, c):
            # to_out + bias + attn2's row bias + residual -> norm3 -> FeedForward -> + x in ONE launch: t1 never exists in HBM
            if self.fuse_post and p.get("tail_post") and post is not None and (post[0] is not None or post[2] is not None):
                hip.attn_out_ffn_proj_fused(att, t0, a2vec, p["tail_w"], p["wo"]["b"], p["ln3"][0], p["ln3"][1], p["ff1"]["b"], p["ff2p"],
                                            p["ff2"]["b"], p["proj_out"]["b"], x.t32, post[0], post[2], post[1], M=M, C_=c,
                                            rows_per_sample=n)
                return True
            t2 = self._new(M, c)
            hip.attn_out_ffn_fused(att, t0, a2vec, p["tail_w"], p["wo"]["b"], p["ln3"][0], p["ln3"][1], p["ff1"]["b"], p["ff2p"],
                                   p["ff2"]["b"], t2, M=M, C_=c, rows_per_sample=n)
            return t2
        t1 = self._new(M, c, torch.float32)
        hip.gemm(att, p["wo"]["w"], None, M=M, N=d, K=d, lda=d, ldc=0, bias=p["wo"]["b"], rowbias=a2vec, rows_per_sample=n,
                 split_k=False, residual32=t0, out32=t1)
        return self._ffn(t1, p, n)

    # ------------------------------------------------------------------ chunks 0 and 1 of the sampler's batch share their prefix
    def _share_ok(self, block, h: Act) -> bool:
        """Can input block 1 -- ``[ResBlock, SpatialTransformer]`` -- of a ``[x ; x ; inv_t]`` batch run its chunk-0 / chunk-1
        prefix once (``_shared_block``)?  Needs the whole batch (no dead-branch elimination), the fused front and the fused
        tail with ``proj_out`` behind it (the launches whose operands can be handed over as row ranges), and a hook mode whose
        chunk-1 edit is either the identity on identical inputs (none, ``replace``, ``fft``, ``mix``) or the flow warp."""
        L = 3 if self.live_chunks is None else self.live_chunks      # chunks in the batch: 3, or 2 = [uncond ; cond] (dead-branch elimination)
        if len(block) != 2 or block[0][0] != "res" or block[1][0] != "st" or L not in (2, 3):
            return False
        if h.N % L or h.t32 is None or h.cs is None or h.hw % 128 or not self.stream32:
            return False
        P = self._packed
        pr, p = P[block[0][1]], P[block[1][1]]
        c, n, F_ = p["c"], h.hw, h.N // L
        if pr["conv2"]["cout"] != c or c % 8:
            return False
        if not (self.fuse_front and self.fuse_ffn and self.fuse_tail and self.fuse_post) or p.get("front_w") is None or \
                p.get("tail_w") is None or not p.get("tail_post"):
            return False
        key = ((L - 1) * F_ * n, c, n)
        ok = self._front_supported.get(key)
        if ok is None:
            ok = self._front_supported[key] = bool(hip.st_front_supported(key[0], c, n))
        if not ok or not self._ffn_ok(F_ * n, c) or not self._ffn_ok((L - 1) * F_ * n, c):
            return False
        attn1 = block[1][2].transformer_blocks[0].attn1
        cfg = getattr(attn1, "_vface_cfg", None)
        if cfg is not None and cfg.switch_on and cfg.chunks != 3:
            return False
        try:
            pl = plan_fusion(cfg, h.N, n, self.halo_hw if self.halo_exchange is not None else None, self.live_chunks)
        except Exception:
            return False          # (the whole-batch path raises it where the caller expects it)
        if pl["staged"] or pl["v_fixed"]:
            return False
        if L == 2:
            # [uncond ; cond] alone: shared only where chunk 1 IS chunk 0 at this layer (no hook, fft, mix) -- the one case whose
            # chunk-1 bits the three-chunk shared form changes, so that dropping the recon third keeps ITS bits; replace / flow_fix
            # are bit-identical shared or not, and stay on the whole-batch launches here
            return pl["fusion"] == hip.FUSION_NONE or (pl["fusion"] == hip.FUSION_LINEAR and pl["warp_hw"] is None)
        return pl["fusion"] in (hip.FUSION_NONE, hip.FUSION_REPLACE, hip.FUSION_LINEAR)

    def _shared_block(self, block, h: Act, out, emb_all: torch.Tensor, a2_all: torch.Tensor) -> Act:
        """Input block 1 of the sampler's ``[uncond ; cond ; recon]`` batch (ddim_w_inv.py:632-655: ``x_in = cat([x, x, inv_t])``,
        ``t_in = cat([t] * 3)``): the ResBlock and everything of the SpatialTransformer in front of attn2's row bias see the same
        numbers for chunks 0 and 1, and every kernel here is batch-invariant -- so they run on the LAST 2F samples (a contiguous
        row range of the 3F-sample buffers) and chunk 0 reads chunk 1's rows.  ``h``: input block 0's output over all 3F samples
        (the skip connection needs it whole).  Chunks 0 and 2 come out bit-identical to the whole-batch launches; chunk 1 too
        under ``flow_fix`` / ``replace`` / no hook; under ``fft`` / ``mix`` its q,k ARE chunk 0's (what the reference's
        ``combine_fft_high_low(q0, q1)`` returns for q1 = q0 up to its FFT's fp32 rounding, face_swap_utils.py:425-464) instead of
        the folded-weight projection of the same rows."""
        P = self._packed
        (_, pre_r, _), (_, pre_s, mod) = block
        L = 3 if self.live_chunks is None else self.live_chunks
        F_ = h.N // L
        Fn = F_ * h.hw
        hv = Act(h.t[Fn:] if h.t is not None else None, (L - 1) * F_, h.H, h.W, h.cs[Fn // 64:], h.t32[Fn:])
        co = P[pre_r]["conv2"]["cout"]
        r = self._res(hv, P[pre_r], emb_all[F_:], self._new_target(hv.M, co, hv.hw, need16=False))
        p = P[pre_s]
        a, b = p["a2_slice"]
        self._st_front_shared(r, p, mod.transformer_blocks[0].attn1, a2_all[:, a:b], out, F_, L)
        return Act(out[0], h.N, h.H, h.W, out[1], out[2])

    def _st_front_shared(self, x: Act, p: dict, attn1, a2vec: torch.Tensor, post, F_: int, L: int = 3) -> None:
        """``_st_front`` for ``_shared_block``: ``x`` holds the 2F samples [A ; C] -- A stands for chunks 0 AND 1, C is chunk 2 --,
        ``a2vec`` / ``post`` cover all 3F.  Front, dual-source projections and (hook permitting) attention on 2F; the tail, where
        attn2's row bias separates chunk 0 from chunk 1, as two launches: rows A with chunk 0's bias -> chunk 0, rows [A ; C]
        with chunk 1's and 2's -> chunks 1, 2.  ``L = 2``: the batch is [uncond ; cond] alone, ``x`` holds A only (``_share_ok``
        admits the hook modes that leave chunk 1 equal to chunk 0 here)."""
        c, n = p["c"], x.hw
        d, heads = c, attn1.heads
        Fn = F_ * n
        M2, M3 = (L - 1) * Fn, 3 * Fn
        cfg = getattr(attn1, "_vface_cfg", None)
        fw = attn1.__dict__.get("forward")
        if fw is not None and not getattr(fw, "_vface", False):
            raise hip.VFaceHipError("attn1.forward was replaced by a closure this engine does not know; use "
                                    "vface_amd.ldm.models.pnp_utils.register_spa_attn_injection")
        pl = plan_fusion(cfg, L * F_, n, self.halo_hw if self.halo_exchange is not None else None, self.live_chunks)
        fusion, flow, alpha, hw = pl["fusion"], pl["flow"], pl["alpha"], pl["warp_hw"]
        warp = hw is not None
        assert L == 3 or not warp
        sharded = warp and self.halo_exchange is not None
        ab = hip.groupnorm_coeffs_from_cols(x.cs, p["gn"][0], p["gn"][1], nimg=(L - 1) * F_, hw=n, C_=c, eps=1e-6)
        t0 = self._new(M2, c, torch.float32)
        # with a warp, chunk 1's q|k differ from chunk 0's: the attention then runs over three sample slots -- slot 0 = the warped
        # q|k of chunk 1, slot 1 = A (q|k of chunk 0, v of chunks 0 and 1), slot 2 = C -- addressed through the sample maps
        qkv3 = self._new(M3, 3 * d) if warp else None
        qkv = qkv3[Fn:] if warp else self._new(M2, 3 * d)
        ln = self._new(M2, c) if (fusion == hip.FUSION_LINEAR and L == 3) else None
        hip.st_front(x.t32, ab, p["front_w"], p["proj_in"]["b"], p["ln1"][0], p["ln1"][1], t0, qkv, M=M2, C_=c, hw=n, NQ=3 * d,
                     rows_full=M2 if fusion == hip.FUSION_NONE else Fn, nq_lo=0 if fusion == hip.FUSION_NONE else 2 * d, ln=ln)
        if fusion == hip.FUSION_LINEAR and L == 3:
            wlin = self._wlin(p, *pl["wlin"])
            ldl = ln.stride(0)

            def fused(src, dst):      # own rows `src`, structure rows = A (chunk 0's LayerNorm output)
                hip.gemm(src, wlin, dst, M=Fn, N=2 * d, K=2 * d, lda=ldl, ldc=dst.stride(0), ldw=2 * d, a2=ln, lda2=ldl, k1=d,
                         split_k=False)
            if warp:
                T = self._new(Fn, 2 * d)
                fused(ln, T)                                  # chunk 1 (its own rows are A's)
                halo = None
                if sharded:
                    handle = self._halo_start(T[(F_ - 1) * n:])
                fused(ln[Fn:], qkv[Fn:, :2 * d])              # chunk 2
                if sharded:
                    ev = self.exchange_events if not isinstance(self.halo_exchange, _GraphSegments) else None
                    if ev is not None:
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                    halo = self.halo_exchange.finish_exchange(handle)
                    if ev is not None:
                        e1.record()
                        ev.append((e0, e1))
                hip.flow_warp(T, qkv3[:Fn, :2 * d], flow, F=F_, h=hw[0], w=hw[1], C_=2 * d, ld_src=2 * d, fs_src=n * 2 * d,
                              ld_dst=3 * d, fs_dst=n * 3 * d, alpha=alpha, prev=halo, ld_prev=2 * d,
                              flow_prev=self.halo_flow if halo is not None else None)
            else:
                fused(ln[Fn:], qkv[Fn:, :2 * d])              # chunk 2; chunk 1's FSAI(q0, q0) is q0
        kw = dict(heads=heads, n=n, nk=n, dh=d // heads, ldq=3 * d, ldk=3 * d, ldv=3 * d, bsq=n * 3 * d, bsk=n * 3 * d,
                  bsv=n * 3 * d, ldo=d, bso=n * d,
                  scale=float(np.float32(1.0) / np.sqrt(np.float32(d // heads))))   # fp32 arithmetic, as capi.cpp computes it
        if warp:
            att = self._new(M3, d)
            hip.attention(qkv3, qkv3[:, d:], qkv3[:, 2 * d:], att, B=3 * F_, qk_map=self._map("share_qk", 3 * F_, F_),
                          v_map=self._map("share_v", 3 * F_, F_), **kw)
            att_0, att_12 = att, att[Fn:]
        else:
            att = self._new(M2, d)
            if fusion == hip.FUSION_REPLACE and hip.load().vface_attention_shared_scores_supported(d // heads, 3):
                hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=F_, v_sets=3, v_sets_live=2, set_stride=F_, **kw)
            else:
                hip.attention(qkv, qkv[:, d:], qkv[:, 2 * d:], att, B=(L - 1) * F_,
                              qk_map=self._map("qk_replace", 2 * F_, F_) if fusion == hip.FUSION_REPLACE else None, **kw)
            att_0, att_12 = att, att
        o16, cs, o32 = post
        for r0, rows, a_, s0 in ((0, Fn, att_0, 0), (Fn, (L - 1) * Fn, att_12, F_)):
            hip.attn_out_ffn_proj_fused(a_, t0, a2vec[s0:], p["tail_w"], p["wo"]["b"], p["ln3"][0], p["ln3"][1], p["ff1"]["b"], p["ff2p"],
                                        p["ff2"]["b"], p["proj_out"]["b"], x.t32, o16[r0:] if o16 is not None else None,
                                        o32[r0:] if o32 is not None else None, cs[r0 // 64:] if cs is not None else None,
                                        M=rows, C_=c, rows_per_sample=n)

    def _halo_start(self, tail: torch.Tensor):
        """``halo_exchange.start_exchange`` with the exchange's ordinal inside this forward stated first (FrameShard.set_index)."""
        ex = self.halo_exchange
        k = getattr(self, "_halo_k", 0)
        self._halo_k = k + 1
        if hasattr(ex, "set_index"):
            ex.set_index(k)
        return ex.start_exchange(tail)

    def _ffn_ok(self, M: int, c: int) -> bool:
        """``vface_ffn_fused_supported`` per (rows, width), asked once (a ctypes call per block per forward otherwise)."""
        key = (M, c)
        ok = self._ffn_supported.get(key)
        if ok is None:
            ok = self._ffn_supported[key] = bool(hip.ffn_fused_supported(M, c))
        return ok

    def _st(self, x: Act, p: dict, mod, a2_all: torch.Tensor, tgt) -> Act:
        """SpatialTransformer.forward + BasicTransformerBlock._forward (attention.py:278-289, 239-243).
        With the fp32 residual stream the block's running sum (``x`` after proj_in, after attn1 + attn2) exists in fp32
        only -- LayerNorm and the next residual add read that; its last value feeds proj_out as a 16-bit operand."""
        N, n, c = x.N, x.hw, p["c"]
        s32 = self.stream32 and c % 8 == 0
        a, b = p["a2_slice"]
        t2 = None
        if s32:
            tgt = self._new_target(x.M, c, x.hw) if tgt is None else tgt
            t2 = self._st_front(x, p, mod.transformer_blocks[0].attn1, a2_all[:, a:b], post=tgt)
        if t2 is not None:
            out, cs, o32 = tgt
            if t2 is not True:
                self._gemm(t2, p["proj_out"], out, colstats=cs, hw=x.H * x.W, out32=o32, **self._resid(x))
            return Act(out, x.N, x.H, x.W, cs, o32)
        g = self._gn(x, p["gn"], 1e-6, False)
        # interior16: the block's INTERIOR running sums (t0 after proj_in, t1 after attention) of the 640- / 1280-channel blocks in 16
        # bits -- 12 B per element less through HBM per block (proj_in, two LayerNorms, to_out's residual in and out, ff.net[2]'s
        # residual); the main residual stream (x_in + proj_out) stays fp32.  Emulated cost on the whole UNet: 1.2400e-3 vs 1.2236e-3
        # (tests/precision_budget.py `si_min_c`; the level-0 blocks, where it would cost 4x that, keep t1 in registers anyway)
        wide = s32 and not (self.interior16 and c >= 640)
        t0 = self._new(x.M, c, torch.float32 if wide else None)
        if wide:
            self._gemm(g.t, p["proj_in"], None, hw=x.H * x.W, out32=t0)
        else:
            self._gemm(g.t, p["proj_in"], t0, hw=x.H * x.W)
        a, b = p["a2_slice"]
        t2 = self._block(t0, p, mod.transformer_blocks[0].attn1, a2_all[:, a:b], N, n, (x.H, x.W))
        out, cs, o32 = self._new_target(x.M, c, x.hw) if tgt is None else tgt
        self._gemm(t2, p["proj_out"], out, colstats=cs, hw=x.H * x.W, out32=o32, **self._resid(x))
        return Act(out, x.N, x.H, x.W, cs, o32)

    # ------------------------------------------------------------------ the forward
    def embeddings(self, timesteps: torch.Tensor, context: torch.Tensor):
        """time_embed -> every ResBlock's emb_layers (openaimodel.py:874-875,264-271), and every attn2's
        ``to_out(to_v(ctx))``, as fp32 row-bias matrices."""
        P, N = self._packed, timesteps.shape[0]
        mc = self.unet.model_channels
        if self.fuse_temb and hip.linear_small_supported(N, 4 * mc, mc) and hip.linear_small_supported(N, P["emb_all"]["n"], 4 * mc):
            # three launches of the few-row kernel (csrc/linear_small.hip) instead of three GEMMs + two SiLUs; each SiLU acts on its
            # layer's fp32 sum
            temb = self._new(N, mc)
            hip.timestep_embedding(timesteps.to(device=self.device, dtype=torch.int64).contiguous(), temb, mc)
            e0 = self._new(N, 4 * mc)
            hip.linear_small(temb, P["time_embed.0"]["w"], P["time_embed.0"]["b"], e0, M=N, N=4 * mc, K=mc, silu=True)
            emb = self._new(N, 4 * mc)
            hip.linear_small(e0, P["time_embed.2"]["w"], P["time_embed.2"]["b"], emb, M=N, N=4 * mc, K=4 * mc, silu=True)
            emb_all = self._new(N, P["emb_all"]["n"], torch.float32)
            hip.linear_small(emb, P["emb_all"]["w"], P["emb_all"].get("b"), emb_all, M=N, N=P["emb_all"]["n"], K=4 * mc)
            return emb_all, self.context_projections(context, N)
        temb = self._new(N, mc)
        hip.timestep_embedding(timesteps.to(device=self.device, dtype=torch.int64).contiguous(), temb, mc)
        e0 = self._new(N, 4 * mc)
        self._gemm(temb, P["time_embed.0"], e0)
        hip.silu(e0, e0)
        emb = self._new(N, 4 * mc)
        self._gemm(e0, P["time_embed.2"], emb)
        hip.silu(emb, emb)
        emb_all = self._new(N, P["emb_all"]["n"], torch.float32)
        self._gemm(emb, P["emb_all"], emb_all, flags=hip.EPI_OUT_F32)
        return emb_all, self.context_projections(context, N)

    def context_projections(self, context: torch.Tensor, N: int) -> torch.Tensor:
        """Every attn2's ``to_out(to_v(ctx))`` as one fp32 [N, sum c] matrix.  They depend on the context only: the DDIM loop
        hands the same tensor object every step, so they are computed once per clip (the cache holds the tensor itself --
        its storage cannot be recycled under us -- and its version counter, so an in-place edit invalidates it)."""
        P = self._packed
        # (one entry per context OBJECT, a few of them: two loops interleaved through one engine -- this batch's sampling and the
        #  next batch's inversion, DDIMSampler.sample_while_inverting -- alternate two contexts and would evict a single entry at
        #  every step of the eager path)
        cache = self.__dict__.setdefault("_a2_lru", {})
        cached = self.__dict__.get("_a2_cache") or cache.get(id(context))
        if cached is not None and cached[0] is context and cached[1] == context._version and cached[2] is P:
            return cached[3]
        ctx = context.reshape(N, -1)
        if ctx.shape[1] != self.unet.context_dim:
            raise hip.VFaceHipError(f"context must be [N, 1, {self.unet.context_dim}] (single token, SURVEY F11); "
                                    f"got {tuple(context.shape)}")
        ctx16 = self._new(N, ctx.shape[1])
        hip.cast_f32(ctx.to(device=self.device, dtype=torch.float32).contiguous(), ctx16)
        v_all = self._new(N, P["a2_v_all"]["n"])
        self._gemm(ctx16, {"w": P["a2_v_all"]["w"]}, v_all)
        a2_all = self._new(N, P["a2_v_all"]["n"], torch.float32)
        for kind, prefix, _ in self.unet.layer_table():
            if kind == "st":
                a, b = P[prefix]["a2_slice"]
                self._gemm(v_all[:, a:b], P[prefix]["a2_out"], a2_all[:, a:b], flags=hip.EPI_OUT_F32)
        cache.pop(id(context), None)
        cache[id(context)] = (context, context._version, P, a2_all)      # most recently used last; the entry keeps its tensor alive
        while len(cache) > 4:
            cache.pop(next(iter(cache)))
        return a2_all

    def forward_nhwc(self, x: Act, timesteps: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
        """UNetModel.forward (openaimodel.py:860-907) on an NHWC 16-bit input (channels padded to 8k).
        Returns eps as fp32 NHWC ``[N*H*W, out_channels]``."""
        self._ensure_packed()
        self._halo_k = 0             # ordinal of the next boundary exchange of this forward (_halo_start)
        P, u = self._packed, self.unet
        emb_all, a2_all = self.embeddings(timesteps, context)
        blocks_in, mid, blocks_out = u.block_table()

        def run(block, h: Act, out) -> Act:
            for i, (kind, prefix, mod) in enumerate(block):
                last = i == len(block) - 1
                tgt = out if last else None
                if tgt is None and kind in ("res", "st"):
                    # inside a block: the 16-bit copy exists only if the next layer reads it as a matrix-core operand
                    # (a down / up convolution); a SpatialTransformer reads the fp32 carrier only
                    need16 = last or block[i + 1][0] != "st"
                    co = P[prefix]["conv2"]["cout"] if kind == "res" else P[prefix]["c"]
                    tgt = self._new_target(h.M, co, h.hw, need16=need16)
                if kind == "conv":
                    h = self._conv(h, P[prefix], tgt)
                elif kind == "res":
                    h = self._res(h, P[prefix], emb_all, tgt)
                elif kind == "st":
                    h = self._st(h, P[prefix], mod, a2_all, tgt)
                elif kind == "down":
                    h = self._conv(h, P[prefix], tgt, stride=2)
                elif kind == "up":
                    h = self._conv(h, P[prefix], tgt, upsample=True)
            return h

        # geometry pass: output shape of every input block, to size the concat buffers
        shapes = []
        H, W = x.H, x.W
        for block in blocks_in:
            for kind, prefix, _ in block:
                if kind == "down":
                    H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1
            shapes.append((H, W, u.block_out_channels(block)))
        nb = len(blocks_in)
        h_ch = [u.block_out_channels(mid)] + [u.block_out_channels(b) for b in blocks_out[:-1]]
        # The concat buffers.  An output block reads cat([h, skip]) twice: its ResBlock's in_layers GroupNorm and the 1x1 shortcut
        # fused into its second convolution -- BOTH from the 16-bit copy (the statistics are the producers' column sums; rounding the
        # GroupNorm's input costs nothing measurable: tests/precision_budget.py `gn_in16`, 1.2236e-3 vs 1.2273e-3 whole-UNet).  So the
        # fp32 carrier exists for the SKIP columns only -- there it is the input path's residual stream, read by the next input
        # block -- and the `h` columns (the previous output block's result, consumed by nothing else) are written once, 16-bit:
        # 4 B per element less from their producers and 2 B less into every concat GroupNorm (6.5 GB per 96-sample step).
        cats, cats_cs, cats32 = [], [], []
        for j in range(nb):  # output block j consumes cat([h_{j}, skip_{nb-1-j}])
            sh, sw, sc = shapes[nb - 1 - j]
            rows = x.N * sh * sw
            cats.append(self._new(rows, h_ch[j] + sc))
            cats_cs.append(self._new_cs(rows, h_ch[j] + sc, sh * sw))
            wide = self.concat32                 # (VFACE_CONCAT32=1, A/B: the carrier over ALL columns, read by the concat GroupNorm)
            cats32.append(self._new(rows, (h_ch[j] + sc) if wide else sc, torch.float32) if (self.stream32 and sc % 8 == 0) else None)

        def part(j, a, b):  # columns [a, b) of concat buffer j, of its statistics and (skip columns only) of its fp32 carrier
            cs, b32, hc = cats_cs[j], cats32[j], (0 if self.concat32 else h_ch[j])
            return (cats[j][:, a:b], (cs[:, a:b] if cs is not None else None),
                    (b32[:, a - hc:b - hc] if (b32 is not None and a >= hc) else None))

        h = x
        for i, block in enumerate(blocks_in):
            j = nb - 1 - i
            if i == 1 and self.share_prefix and self._share_ok(block, h):
                h = self._shared_block(block, h, part(j, h_ch[j], cats[j].shape[1]), emb_all, a2_all)
                continue
            h = run(block, h, part(j, h_ch[j], cats[j].shape[1]))
        h = run(mid, h, part(0, 0, h_ch[0]))
        for j, block in enumerate(blocks_out):
            sh, sw, _ = shapes[nb - 1 - j]
            inp = Act(cats[j], x.N, sh, sw, cats_cs[j], cats32[j] if self.concat32 else None)
            tgt = part(j + 1, 0, h_ch[j + 1]) if j + 1 < nb else None
            h = run(block, inp, tgt)
        oc = P["out.conv"]
        if self.fuse_out and h.cs is not None and oc["cinp"] % 64 == 0 and oc["cinp"] <= 640 and oc["cout"] in (3, 4) and h.C == oc["cinp"]:
            # out = normalization -> SiLU -> conv3x3 (openaimodel.py:712-716) in ONE launch (csrc/outconv.hip)
            ab = hip.groupnorm_coeffs_from_cols(h.cs, P["out.gn"][0], P["out.gn"][1], nimg=h.N, hw=h.hw, C_=h.C, eps=1e-5)
            eps = self._new(h.M, oc["cout"], torch.float32)
            hip.gn_silu_conv3x3_small(h.src, ab, oc["w"], oc["b"], eps, nimg=h.N, H=h.H, W=h.W, cin=oc["cinp"], cout=oc["cout"])
            return eps
        h = self._gn(h, P["out.gn"], 1e-5, True)
        return self._conv(h, P["out.conv"], None, out_f32=True).t

    # ------------------------------------------------------------------ hipGraph replay of a step's forward
    def _hook_signature(self):
        """(signature, flows): everything the hooked attn1 layers contribute to the launch sequence -- the HookCfg fields and
        the flow tensors' shapes -- and the distinct flow tensors themselves, in order of first use."""
        sig, flows = [], []
        for kind, _, mod in self.unet.layer_table():
            if kind != "st":
                continue
            cfg = getattr(mod.transformer_blocks[0].attn1, "_vface_cfg", None)
            if cfg is None:
                sig.append(None)
                continue
            fl, fi = cfg.flow, None
            if fl is not None:
                fi = next((i for i, f in enumerate(flows) if f is fl), None)
                if fi is None:
                    fi = len(flows)
                    flows.append(fl)
            sig.append((cfg.switch_on, cfg.chunks, cfg.fusion, cfg.split_ratio_fft, cfg.alpha, cfg.flow_gate,
                        (fi, tuple(fl.shape)) if fl is not None else None))
        return tuple(sig), flows

    _SPLIT_SAFE = ("replace", "fft", "fft_vfixed", "mix")      # hook modes that never read another FRAME (pnp_utils.py:133-262)
    # ... and the one that reads exactly ONE neighbour (temporal_flow.py:222-237): its halves run as two in-process frame shards,
    # half 0 handing its last frame's fused q|k to half 1 at every hooked flow layer (parallel.StreamShard)
    _SPLIT_COUPLED = ("flow_fix",)

    def _split_plan(self, N: int):
        """Index tensors of the two frame halves of an N-sample batch (frames [0, F/2) and [F/2, F) of every chunk), or None when
        this forward has to stay whole: one stream asked for, frames sharded over ranks, a hook mode that couples frames
        (flow_fix's warp, temporal, adaIn), an odd frame count, or a batch too small to be worth two launch sequences."""
        self._split_coupled = None
        if self.split_streams < 2 or self.halo_exchange is not None or N < 12:      # (8 samples: 11.75 vs 11.46 ms whole, profiles/r04_n)
            return None
        chunks, coupled = 1, []
        for kind, _, mod in self.unet.layer_table():
            if kind != "st":
                continue
            cfg = getattr(mod.transformer_blocks[0].attn1, "_vface_cfg", None)
            if cfg is None or not cfg.switch_on or cfg.chunks not in (2, 3):
                continue
            if cfg.chunks == 3 and cfg.fusion in self._SPLIT_COUPLED and os.environ.get("VFACE_SPLIT_COUPLED", "1") != "0":
                if cfg.flow is not None:
                    if coupled and coupled[0].flow is not cfg.flow:
                        return None          # (two different flow tensors in one forward: one halo field cannot serve both)
                    if not any(c is cfg for c in coupled):
                        coupled.append(cfg)
            elif cfg.chunks == 3 and cfg.fusion not in self._SPLIT_SAFE:
                return None
            there = cfg.chunks if self.live_chunks is None else self.live_chunks
            if chunks not in (1, there):
                return None
            chunks = there
        if self.share_prefix:
            # the batch is the sampler's [x ; x ; inv_t]: each half must again be three chunks of the same frames
            # (_shared_block), also when no hook says so
            want = 3 if self.live_chunks is None else self.live_chunks
            if chunks == 1:
                chunks = want
            elif chunks != want:
                return None
        if N % chunks or (N // chunks) % 2:
            return None
        if coupled and coupled[0].flow.shape[0] != N // chunks - 1:
            return None                      # (plan_fusion will raise on it: leave the whole batch to say so)
        self._split_coupled = coupled or None
        key = (N, chunks)
        plan = self._split_state.get(("plan",) + key)
        if plan is None:
            F_ = N // chunks
            plan = [torch.tensor([c * F_ + f for c in range(chunks) for f in range(lo, hi)], dtype=torch.int64, device=self.device)
                    for lo, hi in ((0, F_ // 2), (F_ // 2, F_))]
            self._split_state[("plan",) + key] = plan
        return plan

    def _concurrent_stream_pair(self):
        """Two side streams whose launches really overlap: a spin kernel on each, timed together and alone -- a pair that shares a
        hardware queue takes twice as long together and is replaced (three tries; then the pair is kept, and the halves simply run
        one after the other on it)."""
        pair = None
        for _ in range(3):
            pair = [torch.cuda.Stream(), torch.cuda.Stream()]
            cur = torch.cuda.current_stream()
            spin = 300_000          # 0.15 ms at core clock (3 ms if the counter is the 100 MHz timer): long against a launch, short once per engine

            def timed(streams):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                for s in streams:
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        torch.cuda._sleep(spin)
                for s in streams:
                    cur.wait_stream(s)
                e1.record(cur)
                e1.synchronize()
                return e0.elapsed_time(e1)
            try:
                timed(pair)
                one, both = timed(pair[:1]), timed(pair)
            except Exception:      # (no spin kernel in this torch build: keep the pair untested)
                break
            if both < 1.5 * one:
                break
        return pair

    def _step_forward_split(self, x: Act, timesteps: torch.Tensor, context: torch.Tensor, plan) -> torch.Tensor:
        """The two halves of ``plan`` through ``_step_forward_one`` on two side streams at once (each with its own hipGraph and its
        own split-K scratch), joined on the calling stream; returns the full batch's eps."""
        N, hw, C = x.N, x.H * x.W, x.t.shape[1]
        cur = torch.cuda.current_stream()
        skey = (N, x.H, x.W, C, x.t.dtype, tuple(context.shape[1:]), cur.cuda_stream, len(plan[0]))
        st = self._split_state.get(skey)
        if st is None:
            # ONE pair of side streams per engine, created back to back: HIP maps streams onto a few hardware queues round-robin
            # in creation order, and two streams that land on the same queue run their launches one after the other (measured: a
            # pair created later, for another configuration, shared a queue -- 24.2 instead of 16.3 ms per inversion step)
            if self._split_pair is None:
                self._split_pair = self._concurrent_stream_pair()
            st = self._split_state[skey] = {
                "streams": self._split_pair, "ctx_id": None, "ctx": None, "ctx_keep": None,
                "x": [torch.empty(len(i) * hw, C, dtype=x.t.dtype, device=self.device) for i in plan],
                "t": [torch.empty(len(i), dtype=torch.int64, device=self.device) for i in plan], "eps": None}
        cid = (id(context), context._version)
        if st["ctx_id"] != cid:
            # (stable tensor objects per half: the graphs' and the eager path's context caches key on identity)
            st["ctx"] = [context.index_select(0, i).contiguous() for i in plan]
            st["ctx_id"], st["ctx_keep"] = cid, context
        ts = timesteps.to(device=self.device, dtype=torch.int64)
        xv = x.t.reshape(N, hw * C)
        for h, idx in enumerate(plan):
            torch.index_select(xv, 0, idx, out=st["x"][h].view(len(idx), hw * C))
            torch.index_select(ts, 0, idx, out=st["t"][h])
        # Once per engine, on the second split step (graphs captured by the first), the overlap is MEASURED on the real work: events
        # around each half and around the fork / join.  Halves that run at once each take about as long as the whole step
        # (total / (t_A + t_B) ~ 0.5); halves that were put on one hardware queue run back to back (~ 1.0, and a half-batch
        # sequence alone is 30 % less efficient than the full batch): then the engine goes back to one launch sequence for good.
        st["calls"] = st.get("calls", 0) + 1
        probe = (not self._split_verified) and st["calls"] == 2
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)] if probe else None
        if probe:
            ev[0].record(cur)
        outs = []
        coupled = self._split_coupled
        halves = None
        if coupled:
            # flow_fix: the halves as two in-process frame shards (parallel.StreamShard).  Per half: its own exchange object, the flow
            # fields between ITS frames, and -- half 1 -- the field from half 0's last frame into its first.  The slices are kept
            # (one object per flow tensor and half): the graphs' private flow copies are refreshed by tensor identity.
            from .parallel import StreamShard
            fl = coupled[0].flow
            F_all = fl.shape[0] + 1
            fkey = (id(fl), fl._version)
            if st.get("flow_key") != fkey:
                st["flow_key"], st["flow_keep"] = fkey, fl
                st["flow_slices"] = [fl[:F_all // 2 - 1], fl[F_all // 2:]]
                st["flow_halo"] = fl[F_all // 2 - 1]
            if "shard_objs" not in st:
                shared = {}
                st["shard_objs"] = [StreamShard(h, 2, F_all, shared) for h in range(2)]
            halves = [(st["shard_objs"][h], st["flow_slices"][h] if st["flow_slices"][h].shape[0] else None,
                       st["flow_halo"] if h else None) for h in range(2)]
        saved = (self.halo_exchange, self.halo_flow, self.halo_hw, [(c, c.flow) for c in (coupled or [])])
        try:
            for h, idx in enumerate(plan):
                s = st["streams"][h]
                s.wait_stream(cur)
                if halves is not None:
                    shard, lflow, hflow = halves[h]
                    self.halo_exchange, self.halo_flow = shard, hflow
                    self.halo_hw = (int(saved[3][0][1].shape[-2]), int(saved[3][0][1].shape[-1]))
                    for c, _ in saved[3]:
                        c.flow = lflow
                with torch.cuda.stream(s), hip.workspace_domain(h + 1):
                    if probe:
                        ev[1 + 2 * h].record(s)
                    outs.append(self._step_forward_one(Act(st["x"][h], len(idx), x.H, x.W), st["t"][h], st["ctx"][h]))
                    if probe:
                        ev[2 + 2 * h].record(s)
        finally:
            self.halo_exchange, self.halo_flow, self.halo_hw = saved[0], saved[1], saved[2]
            for c, f in saved[3]:
                c.flow = f
        for s in st["streams"]:
            cur.wait_stream(s)
        if probe:
            ev[5].record(cur)
            ev[5].synchronize()
            total, ta, tb = ev[0].elapsed_time(ev[5]), ev[1].elapsed_time(ev[2]), ev[3].elapsed_time(ev[4])
            self._split_verified = True
            self.split_overlap = total / max(ta + tb, 1e-6)
            if self.split_overlap > 0.8:
                import warnings
                warnings.warn(f"vface_amd: the two launch streams do not overlap on this device (step {total:.2f} ms, halves {ta:.2f} + "
                              f"{tb:.2f} ms): back to one launch sequence")
                self.split_streams = 1
        if st["eps"] is None or st["eps"].shape[1] != outs[0].shape[1]:
            st["eps"] = torch.empty(N * hw, outs[0].shape[1], dtype=outs[0].dtype, device=self.device)
        ev = st["eps"].view(N, -1)
        for idx, o in zip(plan, outs):
            o.record_stream(cur)
            ev.index_copy_(0, idx, o.reshape(len(idx), -1))
        kind = "coupled" if coupled else "free"
        if self.split_selfcheck and st["calls"] == 2 and kind not in self.split_checked:
            whole = self._step_forward_one(x, timesteps, context)
            same = bool(torch.equal(whole, st["eps"]))
            self.split_checked[kind] = same
            if not same:
                import warnings
                warnings.warn(f"vface_amd: the two launch sequences ({kind} halves) did NOT reproduce the single sequence's bits on this "
                              "device: back to one launch sequence")
                self.split_streams = 1
                return whole
        # Whether two launch sequences beat one depends on the batch, the hook mode and the box: coupled (flow_fix) halves wait for each other
        # at every hooked flow layer (round 6: 16 frames 38.0 vs 38.5 ms, 32 frames 78.0 vs 74.2), and since the launch rules follow the
        # 48-sample half batch and the shared block issues its tail twice, free halves no longer win everywhere either (32 frames fft, same
        # box: 75.3 vs 74.4-75.0 as one sequence; round 4-5: two sequences won by 1-4 %).  So on the second split step of every (kind,
        # batch, resolution) BOTH forms are timed -- two replays each, the whole batch's graph captured for it -- and the configuration
        # keeps the faster one (the single sequence only if it wins by more than 0.5 %).  Skipped where the extra graph would crowd the
        # graph cache (the whole batch's pool ~ the two halves' together).
        tkey = (kind, N, x.H, x.W)
        if self.split_selfcheck and st["calls"] == 2 and tkey not in self.split_timing and \
                4 * sum(v["bytes"] for v in self._graphs.values()) <= self.graph_budget_bytes:
            def timed(fn):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                out = fn()
                e1.record(cur)
                e1.synchronize()
                return e0.elapsed_time(e1), out
            self._step_forward_one(x, timesteps, context)                    # (captures the whole batch's graph if the bit check above did not)
            self.split_timing[tkey] = (None, None)                           # (set first: the nested split calls below must not recurse here)
            t_whole = min(timed(lambda: self._step_forward_one(x, timesteps, context))[0] for _ in range(2))
            t_split, eps = timed(lambda: self._step_forward_split(x, timesteps, context, plan))
            t_split = min(t_split, timed(lambda: self._step_forward_split(x, timesteps, context, plan))[0])
            self.split_timing[tkey] = (t_split, t_whole)
            if t_whole < 0.995 * t_split:
                self._split_off.add(tkey)
            return eps
        return st["eps"]

    def step_forward_nhwc(self, x: Act, timesteps: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
        """``forward_nhwc`` for the DDIM loop; with two launch streams (``split_streams``) and a batch whose frames are not
        coupled, the two frame halves through ``_step_forward_one`` at once."""
        if self.use_graph and x.t32 is None and x.t.is_contiguous():
            plan = self._split_plan(x.N)
            if plan is not None and ("coupled" if self._split_coupled else "free", x.N, x.H, x.W) not in self._split_off:
                return self._step_forward_split(x, timesteps, context, plan)
        return self._step_forward_one(x, timesteps, context)

    def _step_forward_one(self, x: Act, timesteps: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
        """``forward_nhwc`` for the DDIM loop: with ``use_graph`` the launch sequence of one forward is captured into a
        hipGraph the first time a (batch, resolution, hook configuration) combination is seen and replayed afterwards -- the
        same kernels on the same buffers, one host call per step instead of ~1200 (ddim_w_inv.py:299-305 calls the UNet once
        per step with nothing but x and t changing).  What changes between steps or clips is copied into the graph's own
        input buffers: x, t every step; the context projections and the flow fields when a new clip brings new tensors.
        The returned eps is the graph's output buffer: consume it before the next call.
        A frame-sharded engine (``halo_exchange`` installed: RCCL point-to-point exchange inside the forward) is captured as
        graph SEGMENTS cut at the exchange calls -- [.. fused q|k of chunk 1] send/recv [projections it overlaps] wait
        [warp, attention, .. next hooked layer ..] -- with the two ``isend/irecv`` + ``wait`` pairs issued from the host between
        segment replays (5 segments and 4 host calls per step instead of ~1200 launches).  A capture failure (or a pool
        over the byte budget) runs the eager path for that configuration -- decided ONCE for all ranks of a sharded clip
        (``FrameShard.agree``), the failing rank first completing the exchanges its aborted pass owed its neighbours
        (``FrameShard.drain``), and the capturing call returns the warm-up forward's eps instead of running a further forward."""
        if not self.use_graph or x.t32 is not None:
            return self.forward_nhwc(x, timesteps, context)
        self._ensure_packed()
        sig, flows = self._hook_signature()
        ex = self.halo_exchange
        shard_sig = None if ex is None else (id(ex), ex.rank, ex.world, ex.first, ex.count, self.halo_hw,
                                             None if self.halo_flow is None else tuple(self.halo_flow.shape))
        # (every switch that changes the captured launch sequence is part of the key: toggling one on a live engine must not
        # replay a stale graph)
        key = (x.N, x.H, x.W, tuple(x.t.shape), x.t.dtype, self._version, self.stream32, self.fuse_gn, self.fuse_ffn, self.fuse_front, self.fuse_tail, self.fuse_post, self.fuse_temb, self.fuse_out, self.concat32, self.interior16, self.live_chunks, self.share_prefix,
               hip._ws_domain, self.decompose_attn1, self.exchange_events is not None, sig,
               tuple(context.shape), torch.cuda.current_stream().cuda_stream, shard_sig)
        g = self._graphs.get(key)
        if g is None:
            if key in self._graph_failed:
                return self.forward_nhwc(x, timesteps, context)
            g = self._capture(key, x, timesteps, context, flows)
            if "eps_only" in g:
                # no graph for this configuration (capture failed or over budget, on this rank or -- sharded -- on any rank of
                # the clip: one decision for all).  The warm-up forward already computed this call's eps from the same inputs;
                # a second forward here would issue exchanges the other ranks do not make.
                return g["eps_only"]
        else:
            self._graphs[key] = self._graphs.pop(key)      # most recently used last
        cid = (id(context), context._version)
        if cid != g["ctx_id"]:
            g["a2"].copy_(self.context_projections(context, x.N))
            g["ctx_id"], g["ctx_keep"] = cid, context
        for dst, src in zip(g["flows"], flows):
            if (id(src), src._version) != g["flow_ids"].get(id(dst)):
                dst.copy_(src)
                g["flow_ids"][id(dst)] = (id(src), src._version)
                g["flow_keep"][id(dst)] = src
        if g["halo_flow"] is not None and (id(self.halo_flow), self.halo_flow._version) != g["halo_flow_id"]:
            g["halo_flow"].copy_(self.halo_flow)      # (the graph reads its own copy: FrameShard.install may hand a new tensor)
            g["halo_flow_id"], g["halo_flow_keep"] = (id(self.halo_flow), self.halo_flow._version), self.halo_flow
        g["x"].copy_(x.t)
        g["t"].copy_(timesteps)
        for graph, host_op in g["segments"]:
            if graph is not None:
                graph.replay()
            if host_op is not None:
                host_op()
        return g["eps"]

    def _capture(self, key, x: Act, timesteps: torch.Tensor, context: torch.Tensor, flows):
        xs = torch.empty_like(x.t)
        ts = timesteps.to(device=self.device, dtype=torch.int64).clone()
        xs.copy_(x.t)
        # The graph reads two kinds of buffers that are not produced inside it: the context projections and the flow fields.
        # It gets PRIVATE copies of both (refreshed in place when a later clip brings other tensors), so neither the caller's
        # flow tensors nor the eager path's context cache are ever written to.
        cfgs = []
        for kind, _, mod in self.unet.layer_table():
            cfg = getattr(mod.transformer_blocks[0].attn1, "_vface_cfg", None) if kind == "st" else None
            if cfg is not None and cfg.flow is not None and not any(c is cfg for c in cfgs):
                cfgs.append(cfg)
        own_flows = [f.clone() for f in flows]
        saved_flows = [c.flow for c in cfgs]
        saved_cache = getattr(self, "_a2_cache", None)
        real_exchange, real_halo_flow = self.halo_exchange, self.halo_flow
        own_halo_flow = real_halo_flow.clone() if (real_exchange is not None and real_halo_flow is not None) else None
        seg, eps_warm, counting, ok, pool_bytes = None, None, None, True, 0
        try:
            if own_halo_flow is not None:
                self.halo_flow = own_halo_flow
            a2 = self.context_projections(context, x.N).clone()
            self._a2_cache = (context, context._version, self._packed, a2)
            for c in cfgs:
                c.flow = own_flows[next(i for i, f in enumerate(flows) if f is c.flow)]
            # warm-up on a side stream (the documented capture recipe): fills the folded-weight caches, sets every kernel's
            # shared-memory attribute, and brings the allocator to its steady state.  (Sharded: a real forward with real
            # exchanges -- every rank of the clip captures at the same step, so the calls pair up.)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            if real_exchange is not None:
                counting = self.halo_exchange = _CountingExchange(real_exchange)
            with torch.cuda.stream(side):
                eps_warm = self.forward_nhwc(Act(xs, x.N, x.H, x.W), ts, context)
            torch.cuda.current_stream().wait_stream(side)
            if os.environ.get("VFACE_TEST_FAIL_CAPTURE") == "warmup_done":      # (test hook: a capture that dies before its pass)
                raise RuntimeError("injected capture failure (VFACE_TEST_FAIL_CAPTURE)")
            # the captured launches write to the split-K workspace / read the zero page that exist NOW: the warm-up above must
            # have grown them to their final size (hip.py grows by REPLACING the tensor)
            ws_before = {k: v.data_ptr() for k, v in hip._splitk_ws.items()}
            z_before = {k: v.data_ptr() for k, v in hip._zeros.items()}
            torch.cuda.synchronize()
            mem0 = torch.cuda.memory_allocated(self.device)
            seg = _GraphSegments(self, real_exchange)
            if real_exchange is not None:
                self.halo_exchange = seg             # start_exchange / finish_exchange cut the capture (see _GraphSegments)
            with torch.cuda.stream(seg.stream):
                seg.begin()
                eps = self.forward_nhwc(Act(xs, x.N, x.H, x.W), ts, context)
                seg.end(None)
            torch.cuda.current_stream().wait_stream(seg.stream)
            pool_bytes = max(0, torch.cuda.memory_allocated(self.device) - mem0)
            if {k: v.data_ptr() for k, v in hip._splitk_ws.items()} != ws_before or \
                    {k: v.data_ptr() for k, v in hip._zeros.items()} != z_before:
                raise RuntimeError("the split-K workspace / zero page was re-allocated during capture (warm-up did not reach the steady state)")
        except Exception as e:  # capture is an optimisation of the same launch sequence: the eager path is the same code
            import warnings
            if seg is not None:
                seg.abort()
            if eps_warm is None:
                raise            # the warm-up forward itself failed: the eager path is the same code and would fail the same way
            warnings.warn(f"vface_amd: hipGraph capture of the UNet forward failed ({type(e).__name__}: {e}); "
                          "this configuration runs kernel by kernel")
            ok = False
            if real_exchange is not None:
                # the other ranks of the clip are in (or past) their capture pass, which exchanges for real: finish the
                # exchanges this rank's aborted pass still owes them, so every rank has made the same number of calls
                st = seg.pending if seg is not None else None
                done = seg.n_started if seg is not None else 0
                real_exchange.drain(st.pop("h", None) if st else None, counting.tails[done:])
        finally:
            self.halo_exchange, self.halo_flow = real_exchange, real_halo_flow
            for c, f in zip(cfgs, saved_flows):
                c.flow = f
            self._a2_cache = saved_cache
        over = ok and pool_bytes > self.graph_budget_bytes      # one forward larger than the whole budget: do not pin it
        if real_exchange is not None and hasattr(real_exchange, "agree"):
            # ONE decision for the ranks of a clip (pool_bytes follows each rank's own frame count): a rank that replays
            # segments and a rank that launches eagerly would still pair, but a rank that RE-RUNS the forward would not
            all_ok, any_over = real_exchange.agree(ok, over)
            ok, over = all_ok, any_over
        if not ok or over:
            self._graph_failed.add(key)
            seg = None
            return {"eps_only": eps_warm}
        while self._graphs and (len(self._graphs) >= self.graph_capacity or
                                sum(v["bytes"] for v in self._graphs.values()) + pool_bytes > self.graph_budget_bytes):
            self._graphs.pop(next(iter(self._graphs)))      # least recently used first
        g = {"segments": seg.segments, "keep": seg.keep, "bytes": pool_bytes, "x": xs, "t": ts, "eps": eps, "a2": a2,
             "ctx_id": (id(context), context._version),
             "halo_flow": own_halo_flow, "halo_flow_keep": real_halo_flow,
             "halo_flow_id": None if own_halo_flow is None else (id(real_halo_flow), real_halo_flow._version),
             "ctx_keep": context, "flows": own_flows, "flow_ids": {id(d): (id(s_), s_._version) for d, s_ in zip(own_flows, flows)},
             "flow_keep": {id(d): s_ for d, s_ in zip(own_flows, flows)}, "packed": self._packed,
             # the split-K workspace the captured launches write to (hip.py grows it by REPLACING the tensor: keep this one alive)
             "splitk_ws": dict(hip._splitk_ws), "zeros": dict(hip._zeros)}
        self._graphs[key] = g
        return g

    def forward(self, x: torch.Tensor, timesteps: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
        """NCHW fp32 in, NCHW fp32 out -- the signature of the reference's ``UNetModel.forward``."""
        if not x.is_cuda:
            raise hip.VFaceHipError("UNetModel.forward needs CUDA tensors: the VFace path has no CPU fallback")
        N, C, H, W = x.shape
        cpad = (C + 7) // 8 * 8
        xin = self._new(N * H * W, cpad)
        hip.nchw_to_nhwc(x.float().contiguous(), xin, N=N, C_=C, hw=H * W, cpad=cpad)
        eps = self.forward_nhwc(Act(xin, N, H, W), timesteps, context)
        out = torch.empty(N, eps.shape[1], H, W, dtype=torch.float32, device=x.device)
        hip.nhwc_to_nchw_f32(eps, out, N=N, C_=eps.shape[1], hw=H * W, ldx=eps.stride(0))
        return out
