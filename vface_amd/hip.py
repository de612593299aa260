"""ctypes binding of ``libvface_hip.so`` (C ABI: ``include/vface_hip.h``).

PyTorch is plumbing here: it owns device memory and the HIP stream; every call below hands raw device
pointers and ``torch.cuda.current_stream().cuda_stream`` to a hand-written gfx950 kernel.  There is no
fallback: if the library is missing, or a tensor is not on the GPU, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VFACE_HIP_LIB: another build of the same library, for A/B timing of two builds in one process tree; still the C ABI,
#  still no fallback: a path that does not load raises)
LIB_PATH = os.environ.get("VFACE_HIP_LIB") or os.path.join(_HERE, "lib", "libvface_hip.so")

F16, BF16 = 0, 1
EPI_GEGLU, EPI_OUT_F32 = 1, 2
CONV_PAD_TRAILING = 0x40000  # conv3x3: zero padding (0,1,0,1) (the VAE encoder's Downsample)
TUNE_NO_PATCH, TUNE_PATCH = 0x80000, 0x100000  # conv3x3*: never / always (where the shape allows) the patch-staged kernel
TUNE_GN8 = 0x8000000  # A/B: fixed 8-wide column groups in the plain GEMM's tile order
TUNE_NO_Q8 = 0x4000000  # A/B: the 8x8 level stays on the im2col kernel
TUNE_PATCH_BN160 = 0x2000000  # A/B: patch kernel's 160-wide tile wherever it divides Cout
TUNE_F32_TRANSPOSE = 0x1000000  # A/B: epilogue transposes through LDS in fp32 even where 16 bits would do
TUNE_NO_PERSISTENT, TUNE_PERSISTENT = 0x10000, 0x20000  # flags of gemm / conv3x3: force one workgroup per tile / the persistent form
TUNE_BIG_W256, TUNE_BIG_W320 = 0x200000, 0x400000  # gemm, big tile: force the 256- / the 320-channel width (default: by grid rounds; same bits)
TUNE_BIG_TILE, TUNE_NO_BIG_TILE = 0x10000000, 0x20000000  # gemm: always (where the launch qualifies) / never the 256 x 320 tile (csrc/gemm_big.hip)
FUSION_NONE, FUSION_REPLACE, FUSION_LINEAR = 0, 1, 2

_i64, _i32, _f32, _vp, _sz = C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_size_t


class Stream32(C.Structure):
    """``vface_stream32`` of the header: the optional fp32 residual-stream operands of a GEMM-family call."""
    _fields_ = [("residual32", _vp), ("ldr32", _i64), ("out32", _vp), ("ldo32", _i64),
                ("in_scale_shift", _vp), ("ld_scale_shift", _i64), ("in_silu", _i32)]


_s32p = C.POINTER(Stream32)

# name -> (restype, argtypes); mirrors include/vface_hip.h one to one
SIGNATURES = {
    "vface_abi_version": (C.c_int, []),
    "vface_error_string": (C.c_char_p, [_i32]),
    "vface_gemm_variants_built": (C.c_int, []),
    "vface_gemm": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp,
                             _i64, _vp, _i64, _vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _s32p]),
    "vface_conv3x3": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _i32, _vp,
                                _i64, _vp, _i64, _vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _s32p]),
    "vface_splitk_workspace_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "vface_conv_uses_patch_kernel": (C.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32]),
    "vface_conv3x3_plus_1x1": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i32,
                                         _vp, _i64, _vp, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _s32p]),
    "vface_upsample2x_conv3x3_phase": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _i32,
                                                 _vp, _i64, _vp, _i32, _i32, _vp, _i64, _vp, _s32p]),
    "vface_groupnorm_finalize_cols": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _f32, _vp, _vp]),
    "vface_attention": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i32,
                                  _i32, _i32, _i32, _i32, _f32, _i32, _i32, _i32, _vp]),
    "vface_attention_shared_scores_supported": (C.c_int, [_i32, _i32]),
    "vface_layernorm": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _f32, _i32, _i32, _vp]),
    "vface_groupnorm_partial_floats": (C.c_int, [_i32, _i32, _i32, _i32]),
    "vface_groupnorm_stats": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _i32, _i32, _vp]),
    "vface_groupnorm_coeffs_from_cols": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp]),
    "vface_groupnorm_apply": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vface_flow_warp": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _f32,
                                  _f32, _i32, _vp, _vp, _i32, _vp]),
    "vface_flow_to_latent": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vface_im2col": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i64, _i32, _vp]),
    "vface_channel_stats_partial_floats": (_i64, [_i32, _i32, _i32]),
    "vface_channel_stats": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _f32, _vp, _vp, _i32, _vp]),
    "vface_channel_norm_act": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp]),
    "vface_gru_gate": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp]),
    "vface_gru_update": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp]),
    "vface_avgpool2_f32": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "vface_corr_lookup": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _f32, _vp, _i64, _i64, _i32, _vp]),
    "vface_flow_update": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "vface_convex_upsample": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "vface_frame_to_u8": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "vface_resample_u8": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp]),
    "vface_perspective_paste": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "vface_frame_normalise_resize": (C.c_int, [_vp, _i32, _i32, _vp, _i32, _i32, _i32, _vp]),
    "vface_attn1_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "vface_attn1_forward": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _i32, _i32, _i32,
                                      _i32, _i32, _i32, _i32, _vp, _i32, _i32, _f32, _f32, _i32, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _sz, _vp, _i32, _vp, _s32p]),
    "vface_ffn_fused_supported": (C.c_int, [_i64, _i32]),
    "vface_ffn_fused": (C.c_int, [_vp, _i64, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp]),
    "vface_attn_out_ffn_fused": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _i64, _vp,
                                           _i64, _i32, _i32, _i32, _vp]),
    "vface_attn_out_ffn_proj_fused": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp,
                                                _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _vp]),
    "vface_linear_small_supported": (C.c_int, [_i32, _i32, _i32]),
    "vface_linear_small": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vface_gn_silu_conv3x3_small": (C.c_int, [_vp, _i64, _i32, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vface_st_front_supported": (C.c_int, [_i64, _i32, _i32]),
    "vface_st_front": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32,
                                 _i32, _i32, _i32, _vp]),
    "vface_temporal_gauss": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp]),
    "vface_adain_workspace_bytes": (_sz, [_i64, _i32]),
    "vface_adain_fusion": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i32, _vp, _sz, _i32, _vp]),
    "vface_timestep_embedding": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "vface_silu": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "vface_softmax_rows": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _f32, _i32, _vp]),
    "vface_vae_sample": (C.c_int, [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "vface_cast_f32": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "vface_pack_unet_input": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vface_nchw_to_nhwc": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vface_nhwc_to_nchw_f32": (C.c_int, [_vp, _i64, _vp, _i32, _i32, _i32, _vp]),
    "vface_ddim_step": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _f32,
                                  _vp, _i32, _vp]),
    "vface_copy2d": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _vp]),
}

_lib = None


class VFaceHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the library and bind every symbol of the header; raises if it is absent (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VFaceHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The VFace path has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.vface_abi_version() != 7:
        raise VFaceHipError("libvface_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = load().vface_error_string(rc).decode()
        raise VFaceHipError(f"{what}: {msg} (code {rc})")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float16:
        return F16
    if dt == torch.bfloat16:
        return BF16
    raise VFaceHipError(f"unsupported dtype {dt}: the HIP path computes in fp16 or bf16")


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise VFaceHipError("tensor is not on the GPU: the VFace hot path has no CPU fallback")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _s32(residual32: Optional[torch.Tensor], out32: Optional[torch.Tensor], gn_ab: Optional[torch.Tensor] = None,
         gn_silu: bool = False):
    """The ``vface_stream32`` argument: fp32 residual operand and / or fp32 copy of the output (2-D views, unit column
    stride), and for convolutions the fused input normalisation ``gn_ab`` ``[nimg, C, 2]`` fp32; None when nothing is given."""
    if residual32 is None and out32 is None and gn_ab is None:
        return None
    for t in (residual32, out32):
        if t is not None and (t.dtype != torch.float32 or t.dim() != 2 or t.stride(1) != 1):
            raise VFaceHipError("fp32 residual-stream tensors must be 2-D fp32 views with unit column stride")
    if gn_ab is not None and (gn_ab.dtype != torch.float32 or gn_ab.dim() != 3 or gn_ab.shape[2] != 2 or not gn_ab.is_contiguous()):
        raise VFaceHipError("gn_ab must be a contiguous fp32 [nimg, C, 2] tensor (groupnorm_coeffs_from_cols)")
    return C.byref(Stream32(_p(residual32), residual32.stride(0) if residual32 is not None else 0,
                            _p(out32), out32.stride(0) if out32 is not None else 0,
                            _p(gn_ab), gn_ab.shape[1] if gn_ab is not None else 0, int(gn_silu)))


_zeros = {}


def zeros_page(device) -> torch.Tensor:
    """A 256-byte zero page the implicit-GEMM kernels read for padding taps / tile tails."""
    key = str(device)
    if key not in _zeros:
        _zeros[key] = torch.zeros(256, dtype=torch.uint8, device=device)
    return _zeros[key]


_splitk_ws = {}
_ws_domain = 0


class workspace_domain:
    """``with hip.workspace_domain(k):`` -- launches inside take split-K scratch number k.  Launches on ONE stream use a scratch in
    order; a caller that runs two launch sequences on two streams at once (the engine's two half-batches) gives each its own."""

    def __init__(self, k: int):
        self.k = int(k)

    def __enter__(self):
        global _ws_domain
        self.prev, _ws_domain = _ws_domain, self.k
        return self

    def __exit__(self, *exc):
        global _ws_domain
        _ws_domain = self.prev
        return False


def splitk_workspace(device, M: int, N: int, K: int, flags: int = 0, rows_per_sample: int = 1):
    """Device scratch for a split-K launch of this shape (grown on demand, one per device and workspace domain; launches on
    one stream use it in order).  Returns (tensor | None, bytes)."""
    need = load().vface_splitk_workspace_bytes(M, N, K, flags, rows_per_sample)
    if need <= 0:
        return None, 0
    key = str(device) if _ws_domain == 0 else f"{device}#{_ws_domain}"
    ws = _splitk_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _splitk_ws[key] = ws
    return ws, ws.numel()


# ---------------------------------------------------------------------------------------------- wrappers
def gemm(a: torch.Tensor, wt: torch.Tensor, out: torch.Tensor, *, M: int, N: int, K: int, lda: int, ldc: int,
         ldw: Optional[int] = None, bias=None, rowbias=None, rows_per_sample: int = 1, residual=None, ldr: int = 0,
         a2=None, lda2: int = 0, k1: int = 0, a2_row_mod: int = 0, flags: int = 0, colstats=None, split_k: bool = True,
         residual32=None, out32=None):
    """out[M, :N] = a[M, :K] @ wt[:N, :K]^T (+ epilogue).  Tensors are device buffers; M/N/K/ld* describe the view.
    ``residual32`` / ``out32``: the fp32 residual stream (``vface_stream32``); ``out`` may be None with ``out32``."""
    lib = load()
    ws, ws_bytes = splitk_workspace(a.device, M, N, K, flags, rows_per_sample) if split_k else (None, 0)
    rc = lib.vface_gemm(_p(a), lda, _p(a2), lda2, k1, a2_row_mod, _p(wt), ldw if ldw is not None else K, M, N, K,
                        _p(bias), _p(rowbias), rows_per_sample, rowbias.stride(0) if rowbias is not None else 0,
                        _p(residual), ldr, _p(out), ldc, _p(zeros_page(a.device)), flags, dtype_code(a.dtype),
                        _p(colstats), colstats.stride(0) // 2 if colstats is not None else 0, _p(ws), ws_bytes, _stream(),
                        _s32(residual32, out32))
    _check(rc, "vface_gemm")


def conv3x3(x: torch.Tensor, wt: torch.Tensor, out: torch.Tensor, *, nimg: int, H: int, W: int, cin: int, cout: int,
            ldx: int, ldy: int, stride: int = 1, upsample: bool = False, bias=None, rowbias=None, residual=None,
            ldr: int = 0, flags: int = 0, colstats=None, split_k: bool = True, residual32=None, out32=None, gn_ab=None,
            gn_silu: bool = False):
    """``gn_ab`` (+ ``gn_silu``): GroupNorm-apply (+ SiLU) of the INPUT fused into the patch-staged kernel's operand path."""
    lib = load()
    vh, vw = (2 * H, 2 * W) if upsample else (H, W)
    M = nimg * ((vh - 1) // stride + 1) * ((vw - 1) // stride + 1)
    ws, ws_bytes = splitk_workspace(x.device, M, cout, 9 * cin, flags, M // nimg) if split_k else (None, 0)
    rc = lib.vface_conv3x3(_p(x), ldx, nimg, H, W, cin, _p(wt), 9 * cin, cout, stride, int(upsample), _p(bias),
                           _p(rowbias), rowbias.stride(0) if rowbias is not None else 0, _p(residual), ldr, _p(out),
                           ldy, _p(zeros_page(x.device)), flags, dtype_code(x.dtype), _p(colstats),
                           colstats.stride(0) // 2 if colstats is not None else 0, _p(ws), ws_bytes, _stream(),
                           _s32(residual32, out32, gn_ab, gn_silu))
    _check(rc, "vface_conv3x3")


def conv_uses_patch_kernel(H: int, W: int, cin: int, cout: int, window: int = 3, stride: int = 1, upsample: bool = False,
                           flags: int = 0) -> int:
    """Which kernel a convolution launch of this geometry runs: 1 = conv.hip's patch-staged kernel, 2 = its 8x8 form (four
    images per workgroup + split-K reduce), 0 = gemm.hip's implicit GEMM."""
    return int(load().vface_conv_uses_patch_kernel(H, W, cin, cout, window, stride, int(upsample), flags))


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, *, B: int, heads: int, n: int,
              nk: int, dh: int, ldq: int, ldk: int, ldv: int, bsq: int, bsk: int, bsv: int, ldo: int, bso: int,
              scale: float, qk_map=None, v_map=None, variant: int = 0, v_sets: int = 1, set_stride: int = 0,
              v_sets_live: int = 0):
    """``v_sets > 1``: B q/k samples, output sample ``b + g*set_stride`` uses v of that sample (through ``v_map``)
    with the probabilities of q/k sample b, computed once (the "replace" injection).  ``v_sets_live`` (2 of 3): only the first
    sets exist in the batch; their outputs are those of the full call bit for bit."""
    lib = load()
    rc = lib.vface_attention(_p(q), _p(k), _p(v), ldq, ldk, ldv, bsq, bsk, bsv, _p(qk_map), _p(v_map), _p(out), ldo,
                             bso, B, heads, n, nk, dh, scale, dtype_code(out.dtype) | (variant << 8),
                             v_sets | ((v_sets_live if v_sets_live != v_sets else 0) << 8), set_stride, _stream())
    _check(rc, "vface_attention")


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: torch.Tensor, *, M: int, C_: int,
              ldx: int, ldy: int, eps: float = 1e-5):
    rc = load().vface_layernorm(_p(x), ldx, _p(gamma), _p(beta), _p(out), ldy, M, C_, eps,
                                int(x.dtype == torch.float32), dtype_code(out.dtype), _stream())
    _check(rc, "vface_layernorm")


def groupnorm_stats(x: torch.Tensor, *, nimg: int, hw: int, C_: int, ldx: int, groups: int = 32, eps: float = 1e-5):
    lib = load()
    nf = lib.vface_groupnorm_partial_floats(nimg, hw, C_, groups)
    partial = torch.empty(nf, dtype=torch.float32, device=x.device)
    stats = torch.empty(nimg, groups, 2, dtype=torch.float32, device=x.device)
    in32 = x.dtype == torch.float32
    rc = lib.vface_groupnorm_stats(_p(x), ldx, nimg, hw, C_, groups, eps, _p(partial), _p(stats), int(in32),
                                   F16 if in32 else dtype_code(x.dtype), _stream())
    _check(rc, "vface_groupnorm_stats")
    return stats


def groupnorm_stats_from_cols(colstats: torch.Tensor, *, nimg: int, hw: int, C_: int, groups: int = 32, eps: float = 1e-5):
    """colstats: [nimg*hw/64, C(view), 2] fp32 view (row stride = 2 * ld)."""
    stats = torch.empty(nimg, groups, 2, dtype=torch.float32, device=colstats.device)
    rc = load().vface_groupnorm_finalize_cols(_p(colstats), colstats.stride(0) // 2, nimg, hw, C_, groups, eps, _p(stats),
                                              _stream())
    _check(rc, "vface_groupnorm_finalize_cols")
    return stats


def groupnorm_coeffs_from_cols(colstats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, *, nimg: int, hw: int, C_: int,
                               groups: int = 32, eps: float = 1e-5) -> torch.Tensor:
    """Per-(image, channel) scale / shift (a, b) of GroupNorm from producer-side column statistics: fp32 [nimg, C, 2]."""
    ab = torch.empty(nimg, C_, 2, dtype=torch.float32, device=colstats.device)
    rc = load().vface_groupnorm_coeffs_from_cols(_p(colstats), colstats.stride(0) // 2, nimg, hw, C_, groups, eps, _p(gamma),
                                                 _p(beta), _p(ab), _stream())
    _check(rc, "vface_groupnorm_coeffs_from_cols")
    return ab


def groupnorm_apply(x: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: torch.Tensor,
                    *, nimg: int, hw: int, C_: int, ldx: int, ldy: int, groups: int = 32, silu: bool = False):
    rc = load().vface_groupnorm_apply(_p(x), ldx, _p(stats), _p(gamma), _p(beta), _p(out), ldy, nimg, hw, C_, groups,
                                      int(silu), int(x.dtype == torch.float32), dtype_code(out.dtype), _stream())
    _check(rc, "vface_groupnorm_apply")


def flow_warp(src: torch.Tensor, dst: torch.Tensor, flow: Optional[torch.Tensor], *, F: int, h: int, w: int, C_: int,
              ld_src: int, fs_src: int, ld_dst: int, fs_dst: int, alpha: float, prev=None, ld_prev: int = 0,
              flow_prev=None, cuda_recip_div: bool = False, dbg_x0=None, dbg_y0=None):
    # python-double (1 - alpha) rounded to fp32, as the reference's scalar * fp32-tensor product sees it
    rc = load().vface_flow_warp(_p(src), ld_src, fs_src, _p(prev), ld_prev, _p(flow), _p(flow_prev), _p(dst), ld_dst,
                                fs_dst, F, h, w, C_, float(alpha), float(1.0 - alpha), int(cuda_recip_div),
                                _p(dbg_x0), _p(dbg_y0), dtype_code(src.dtype), _stream())
    _check(rc, "vface_flow_warp")


def flow_to_latent(flow_px: torch.Tensor, factor: int = 8) -> torch.Tensor:
    """[P, 2, H, W] fp32 pixel-resolution flow -> [P, 2, H/factor, W/factor] latent-resolution flow (area mean / factor)."""
    if flow_px.dim() != 4 or flow_px.shape[1] != 2:
        raise VFaceHipError(f"flow must be [pairs, 2, H, W]; got {tuple(flow_px.shape)}")
    f = flow_px.to(dtype=torch.float32).contiguous()
    P, _, H, W = f.shape
    out = torch.empty(P, 2, H // factor, W // factor, dtype=torch.float32, device=f.device)
    rc = load().vface_flow_to_latent(_p(f), _p(out), P, H, W, factor, _stream())
    _check(rc, "vface_flow_to_latent")
    return out


# ---- glue of the RAFT-shaped flow producer (csrc/raft.hip; include/vface_hip.h "optical-flow producer") ----
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3


def im2col(x: torch.Tensor, out: torch.Tensor, *, nimg: int, H: int, W: int, C_: int, kh: int, kw: int, stride: int = 1,
           pad_y: int = 0, pad_x: int = 0, ldx: Optional[int] = None):
    rc = load().vface_im2col(_p(x), ldx if ldx is not None else x.stride(0), nimg, H, W, C_, kh, kw, stride, pad_y, pad_x, _p(out),
                             out.stride(0), dtype_code(x.dtype), _stream())
    _check(rc, "vface_im2col")


def channel_stats(x: torch.Tensor, *, nimg: int, hw: int, C_: int, ldx: Optional[int] = None, eps: float = 1e-5) -> torch.Tensor:
    lib = load()
    partial = torch.empty(int(lib.vface_channel_stats_partial_floats(nimg, hw, C_)), dtype=torch.float32, device=x.device)
    stats = torch.empty(nimg, C_, 2, dtype=torch.float32, device=x.device)
    rc = lib.vface_channel_stats(_p(x), ldx if ldx is not None else x.stride(0), nimg, hw, C_, eps, _p(partial), _p(stats),
                                 dtype_code(x.dtype), _stream())
    _check(rc, "vface_channel_stats")
    return stats


def channel_norm_act(x: torch.Tensor, y: Optional[torch.Tensor], *, M: int, hw: int, C_: int, act: int, stats=None, residual=None,
                     y32=None, ldx: Optional[int] = None, ldy: Optional[int] = None, ldr: Optional[int] = None):
    rc = load().vface_channel_norm_act(_p(x), ldx if ldx is not None else x.stride(0), _p(stats), _p(residual),
                                       (ldr if ldr is not None else residual.stride(0)) if residual is not None else 0, _p(y),
                                       (ldy if ldy is not None else y.stride(0)) if y is not None else 0, _p(y32),
                                       y32.stride(0) if y32 is not None else 0, M, hw, C_, act, dtype_code(x.dtype), _stream())
    _check(rc, "vface_channel_norm_act")


def gru_gate(zr: torch.Tensor, h32: torch.Tensor, z: torch.Tensor, rh: torch.Tensor, *, M: int, hidden: int, ldrh: int):
    rc = load().vface_gru_gate(_p(zr), zr.stride(0), _p(h32), _p(z), z.stride(0), _p(rh), ldrh, M, hidden, dtype_code(zr.dtype), _stream())
    _check(rc, "vface_gru_gate")


def gru_update(q: torch.Tensor, z: torch.Tensor, h32: torch.Tensor, h16a, lda: int, h16b, ldb: int, *, M: int, hidden: int):
    rc = load().vface_gru_update(_p(q), q.stride(0), _p(z), z.stride(0), _p(h32), _p(h16a), lda, _p(h16b), ldb, M, hidden,
                                 dtype_code(q.dtype), _stream())
    _check(rc, "vface_gru_update")


def avgpool2_f32(x: torch.Tensor, y: torch.Tensor, *, R: int, h: int, w: int):
    _check(load().vface_avgpool2_f32(_p(x), _p(y), R, h, w, _stream()), "vface_avgpool2_f32")


def corr_lookup(vols, flow32: torch.Tensor, out: torch.Tensor, *, h: int, w: int, scale: float):
    n = len(vols)
    ptrs = (C.c_void_p * n)(*[_p(v) for v in vols])
    hs = (C.c_int * n)(*[int(v.shape[-2]) for v in vols])
    ws = (C.c_int * n)(*[int(v.shape[-1]) for v in vols])
    rc = load().vface_corr_lookup(C.cast(ptrs, C.c_void_p), C.cast(hs, C.c_void_p), C.cast(ws, C.c_void_p), n, _p(flow32), h, w,
                                  scale, _p(out), out.stride(0), flow32.shape[0], dtype_code(out.dtype), _stream())
    _check(rc, "vface_corr_lookup")


def flow_update(flow32: torch.Tensor, delta32, copies, *, dtype: torch.dtype):
    """``copies``: up to three (tensor view whose first two columns receive the 16-bit flow, row stride) pairs."""
    c = list(copies) + [(None, 0)] * (3 - len(copies))
    rc = load().vface_flow_update(_p(flow32), _p(delta32), delta32.stride(0) if delta32 is not None else 0, _p(c[0][0]), c[0][1],
                                  _p(c[1][0]), c[1][1], _p(c[2][0]), c[2][1], flow32.shape[0], dtype_code(dtype), _stream())
    _check(rc, "vface_flow_update")


def convex_upsample(mask32: torch.Tensor, flow32: torch.Tensor, *, B: int, h: int, w: int, mult: float = 0.25) -> torch.Tensor:
    out = torch.empty(B, 2, 8 * h, 8 * w, dtype=torch.float32, device=mask32.device)
    rc = load().vface_convex_upsample(_p(mask32), mask32.stride(0), _p(flow32), _p(out), B, h, w, mult, _stream())
    _check(rc, "vface_convex_upsample")
    return out


def frame_to_u8(x: torch.Tensor, half_arithmetic: bool = False) -> torch.Tensor:
    """Decoded frames [F, 3, H, W] in [-1, 1] (fp16 / bf16 / fp32) -> uint8 [F, H, W, 3]: clamp((x + 1) / 2, 0, 1) * 255, truncated.
    The arithmetic is **fp32** whatever the input type -- the reference's ``--precision full`` run.  ``half_arithmetic=True`` (fp16
    input only) rounds every operation to float16 instead, which is what the reference's default ``--precision autocast`` run
    computes on the float16 tensor ``decode_first_stage`` returns (:597-608); the two can differ by one in a pixel."""
    if x.dim() != 4 or x.shape[1] != 3:
        raise VFaceHipError(f"frame_to_u8: frames must be [F, 3, H, W]; got {tuple(x.shape)}")
    x = x.contiguous()
    if half_arithmetic and x.dtype != torch.float16:
        raise VFaceHipError("frame_to_u8(half_arithmetic=True) takes float16 frames (the autocast path's decode output)")
    kind = 3 if half_arithmetic else (2 if x.dtype == torch.float32 else dtype_code(x.dtype))
    F_, _, H, W = x.shape
    out = torch.empty(F_, H, W, 3, dtype=torch.uint8, device=x.device)
    _check(load().vface_frame_to_u8(_p(x), _p(out), F_, H, W, kind, _stream()), "vface_frame_to_u8")
    return out


def resample_u8(src: torch.Tensor, out_n: int, axis: int, bounds: torch.Tensor, kk: torch.Tensor) -> torch.Tensor:
    """One pass of Pillow's 8-bit bilinear resampling over uint8 [F, H, W, 3] frames: axis 0 along x (W -> out_n), axis 1 along y."""
    if src.dtype != torch.uint8 or src.dim() != 4 or src.shape[3] != 3 or not src.is_contiguous():
        raise VFaceHipError("resample_u8: frames must be contiguous uint8 [F, H, W, 3]")
    if bounds.dtype != torch.int32 or kk.dtype != torch.int32 or tuple(bounds.shape) != (out_n, 2) or kk.shape[0] != out_n:
        raise VFaceHipError("resample_u8: bounds [out_n, 2] / kk [out_n, ksize] must be int32 tables for this output size")
    F_, H, W, _ = src.shape
    if axis == 0:
        dst = torch.empty(F_, H, out_n, 3, dtype=torch.uint8, device=src.device)
        in_n, lines = W, H
    else:
        dst = torch.empty(F_, out_n, W, 3, dtype=torch.uint8, device=src.device)
        in_n, lines = H, W
    rc = load().vface_resample_u8(_p(src), _p(dst), F_, in_n, out_n, lines, axis, _p(bounds), _p(kk), kk.shape[1], _stream())
    _check(rc, "vface_resample_u8")
    return dst


def perspective_paste(crop: torch.Tensor, frame: torch.Tensor, coeffs) -> torch.Tensor:
    """Paste uint8 crops [F, h, w, 3] into uint8 frames [F, H, W, 3] IN PLACE through the eight PIL perspective coefficients per
    frame (``coeffs``: a float64 device tensor [F, 8], or eight host floats when F == 1)."""
    for t in (crop, frame):
        if t.dtype != torch.uint8 or t.dim() != 4 or t.shape[3] != 3 or not t.is_contiguous():
            raise VFaceHipError("perspective_paste: crops and frames must be contiguous uint8 [F, H, W, 3]")
    F_, sh, sw, _ = crop.shape
    if frame.shape[0] != F_:
        raise VFaceHipError("perspective_paste: one crop per frame")
    dev_c, host_c = None, None
    if isinstance(coeffs, torch.Tensor) and coeffs.is_cuda:
        if coeffs.dtype != torch.float64 or tuple(coeffs.shape) != (F_, 8) or not coeffs.is_contiguous():
            raise VFaceHipError("perspective_paste: device coefficients must be contiguous float64 [F, 8]")
        dev_c = coeffs
    else:
        vals = [float(v) for v in (coeffs.reshape(-1).tolist() if hasattr(coeffs, "reshape") else coeffs)]
        if len(vals) != 8 or F_ != 1:
            raise VFaceHipError("perspective_paste: host coefficients are eight numbers for a single frame")
        host_c = (C.c_double * 8)(*vals)
    rc = load().vface_perspective_paste(_p(crop), sw, sh, _p(frame), frame.shape[2], frame.shape[1], F_, _p(dev_c),
                                        C.cast(host_c, C.c_void_p) if host_c is not None else None, _stream())
    _check(rc, "vface_perspective_paste")
    return frame


def frame_normalise_resize(frame: torch.Tensor, OH: int, OW: int) -> torch.Tensor:
    """uint8 frames [F, H, W, 3] -> fp32 [F, 3, OH, OW] in [-1, 1]: ToTensor + Normalize(0.5, 0.5) + bilinear Resize (no antialias)."""
    if frame.dtype != torch.uint8 or frame.dim() != 4 or frame.shape[3] != 3 or not frame.is_contiguous():
        raise VFaceHipError("frame_normalise_resize: frames must be contiguous uint8 [F, H, W, 3]")
    F_, H, W, _ = frame.shape
    out = torch.empty(F_, 3, OH, OW, dtype=torch.float32, device=frame.device)
    _check(load().vface_frame_normalise_resize(_p(frame), W, H, _p(out), OW, OH, F_, _stream()), "vface_frame_normalise_resize")
    return out


def attn1_workspace_bytes(B: int, n: int, d: int, chunks: int) -> int:
    return int(load().vface_attn1_workspace_bytes(B, n, d, chunks))


def attn1_forward(x, wqkv, wlin, wo, bo, out, *, B, n, d, heads, chunks, fusion, ldx, ldo, workspace, rowbias=None,
                  residual=None, ldr=0, v_fixed=False, flow=None, h=0, w=0, alpha=0.8, cuda_recip_div=False,
                  halo_qk=None, halo_flow=None, tail_qk=None, qk_map=None, v_map=None, residual32=None, out32=None):
    rc = load().vface_attn1_forward(_p(x), ldx, _p(wqkv), _p(wlin), _p(wo), _p(bo), _p(rowbias),
                                    rowbias.stride(0) if rowbias is not None else 0, _p(residual), ldr, _p(out), ldo,
                                    B, n, d, heads, chunks, fusion, int(v_fixed), _p(flow), h, w, float(alpha),
                                    float(1.0 - alpha), int(cuda_recip_div), _p(halo_qk), _p(halo_flow), _p(tail_qk),
                                    _p(qk_map), _p(v_map), _p(workspace), workspace.numel() * workspace.element_size(),
                                    _p(zeros_page(x.device)), dtype_code(x.dtype), _stream(), _s32(residual32, out32))
    _check(rc, "vface_attn1_forward")


def ffn_fused_supported(M: int, C_: int) -> bool:
    return bool(load().vface_ffn_fused_supported(M, C_))


def attn_out_ffn_fused(att: torch.Tensor, resid32: torch.Tensor, rowbias: Optional[torch.Tensor], wo_w1: torch.Tensor, bo: torch.Tensor,
                       gamma: torch.Tensor, beta: torch.Tensor, b1: torch.Tensor, w2p: torch.Tensor, b2: torch.Tensor,
                       out16: Optional[torch.Tensor], *, M: int, C_: int, rows_per_sample: int, out32: Optional[torch.Tensor] = None,
                       eps: float = 1e-5):
    """attn1's out-projection + attn2's row bias + residual, norm3 and the FeedForward in ONE launch (``vface_attn_out_ffn_fused``):
    ``out = ff(LayerNorm(t1)) + t1`` with ``t1 = att @ Wo^T + bo + rowbias[sample] + resid32`` never stored."""
    o = out16 if out16 is not None else out32
    rc = load().vface_attn_out_ffn_fused(_p(att), att.stride(0), _p(resid32), resid32.stride(0), _p(rowbias),
                                         rowbias.stride(0) if rowbias is not None else 0, rows_per_sample, _p(wo_w1), _p(bo), _p(gamma),
                                         _p(beta), eps, _p(b1), _p(w2p), _p(b2), _p(out16), out16.stride(0) if out16 is not None else 0,
                                         _p(out32), out32.stride(0) if out32 is not None else 0, M, C_, dtype_code(att.dtype), _stream())
    _check(rc, "vface_attn_out_ffn_fused")


def attn_out_ffn_proj_fused(att: torch.Tensor, resid32: torch.Tensor, rowbias: Optional[torch.Tensor], w_stream: torch.Tensor,
                            bo: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, b1: torch.Tensor, w2p: torch.Tensor, b2: torch.Tensor,
                            b_po: torch.Tensor, x_in: torch.Tensor, out16: Optional[torch.Tensor], out32: Optional[torch.Tensor],
                            colstats: Optional[torch.Tensor], *, M: int, C_: int, rows_per_sample: int, eps: float = 1e-5):
    """``attn_out_ffn_fused`` + the SpatialTransformer's ``proj_out`` + ``x_in`` (+ column statistics) in ONE launch
    (``vface_attn_out_ffn_proj_fused``); ``w_stream`` = ``packing.pack_attn_out_ffn(w_o, w1_geglu, w_proj_out)``."""
    rc = load().vface_attn_out_ffn_proj_fused(
        _p(att), att.stride(0), _p(resid32), resid32.stride(0), _p(rowbias), rowbias.stride(0) if rowbias is not None else 0,
        rows_per_sample, _p(w_stream), _p(bo), _p(gamma), _p(beta), eps, _p(b1), _p(w2p), _p(b2), _p(b_po), _p(x_in), x_in.stride(0),
        _p(out16), out16.stride(0) if out16 is not None else 0, _p(out32), out32.stride(0) if out32 is not None else 0,
        _p(colstats), colstats.stride(0) // 2 if colstats is not None else 0, M, C_, dtype_code(att.dtype), _stream())
    _check(rc, "vface_attn_out_ffn_proj_fused")


def linear_small_supported(M: int, N: int, K: int) -> bool:
    return bool(load().vface_linear_small_supported(M, N, K))


def linear_small(a: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *, M: int, N: int, K: int,
                 silu: bool = False):
    """``out[:M, :N] = act(a[:M, :K] @ wt[:N, :K]^T + bias)`` on a handful of rows (``vface_linear_small``: the time-embedding chain);
    ``out``: 16-bit or fp32."""
    rc = load().vface_linear_small(_p(a), a.stride(0), _p(wt), wt.stride(0), _p(bias), _p(out),
                                   out.stride(0), int(out.dtype == torch.float32), int(silu), M, N, K, dtype_code(wt.dtype), _stream())
    _check(rc, "vface_linear_small")


def gn_silu_conv3x3_small(x: torch.Tensor, gn_ab: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *,
                          nimg: int, H: int, W: int, cin: int, cout: int):
    """GroupNorm-apply + SiLU + conv3x3 to 3 / 4 channels in one launch (the UNet's ``out`` layer, ``vface_gn_silu_conv3x3_small``);
    ``x``: the fp32 carrier or the 16-bit copy ``[nimg*H*W, cin]``; ``out``: fp32 ``[nimg*H*W, >= cout]``."""
    rc = load().vface_gn_silu_conv3x3_small(_p(x), x.stride(0), int(x.dtype == torch.float32), _p(gn_ab), gn_ab.stride(0) // 2, _p(wt),
                                            _p(bias), _p(out), out.stride(0), nimg, H, W, cin, cout, dtype_code(wt.dtype), _stream())
    _check(rc, "vface_gn_silu_conv3x3_small")


def st_front_supported(M: int, C_: int, hw: int) -> bool:
    return bool(load().vface_st_front_supported(M, C_, hw))


def st_front(x32: torch.Tensor, gn_ab: torch.Tensor, wcat: torch.Tensor, b_in: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
             t0: torch.Tensor, qkv: torch.Tensor, *, M: int, C_: int, hw: int, NQ: int, rows_full: int, nq_lo: int = 0,
             ln: Optional[torch.Tensor] = None, eps: float = 1e-5):
    """GroupNorm-apply -> proj_in (-> ``t0`` fp32) -> LayerNorm -> attn1 projection (-> ``qkv`` 16-bit) in one launch
    (csrc/stfront.hip).  ``gn_ab``: ``groupnorm_coeffs_from_cols`` of the producer of ``x32``; ``wcat``: ``packing.pack_st_front``."""
    if x32.dtype != torch.float32 or t0.dtype != torch.float32 or gn_ab.dtype != torch.float32:
        raise VFaceHipError("st_front: x32, t0 and gn_ab are fp32")
    rc = load().vface_st_front(_p(x32), x32.stride(0), _p(gn_ab), gn_ab.stride(0) // 2, hw, _p(wcat), _p(b_in), _p(gamma), _p(beta), eps,
                               _p(t0), t0.stride(0), _p(qkv), qkv.stride(0), _p(ln) if ln is not None else None,
                               ln.stride(0) if ln is not None else 0, M, C_, NQ, rows_full, nq_lo, dtype_code(qkv.dtype), _stream())
    _check(rc, "vface_st_front")


def ffn_fused_width_supported(C_: int) -> bool:
    """Does the fused FeedForward take blocks of this width at all (any row count)?  128 rows = one workgroup's tokens."""
    return bool(load().vface_ffn_fused_supported(128, C_))


def ffn_fused(x32: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, w2p: torch.Tensor,
              b2: torch.Tensor, out16: Optional[torch.Tensor], *, M: int, C_: int, out32: Optional[torch.Tensor] = None,
              eps: float = 1e-5):
    """``out = ff.net[2](GEGLU(ff.net[0](LayerNorm(x32)))) + x32`` in one launch (``vface_ffn_fused``); ``w1`` / ``b1`` in the GEGLU
    packing of ``packing.pack_geglu``, ``w2p`` from ``packing.pack_ffn_w2``."""
    if x32.dtype != torch.float32 or x32.dim() != 2 or x32.stride(1) != 1:
        raise VFaceHipError("ffn_fused: x32 must be a 2-D fp32 view with unit column stride")
    rc = load().vface_ffn_fused(_p(x32), x32.stride(0), _p(gamma), _p(beta), eps, _p(w1), _p(b1), _p(w2p), _p(b2), _p(out16),
                                out16.stride(0) if out16 is not None else 0, _p(out32),
                                out32.stride(0) if out32 is not None else 0, M, C_, dtype_code(w1.dtype), _stream())
    _check(rc, "vface_ffn_fused")


def timestep_embedding(t: torch.Tensor, out: torch.Tensor, dim: int):
    assert t.dtype == torch.int64
    rc = load().vface_timestep_embedding(_p(t), _p(out), t.numel(), dim, dtype_code(out.dtype), _stream())
    _check(rc, "vface_timestep_embedding")


def silu(x: torch.Tensor, out: torch.Tensor):
    rc = load().vface_silu(_p(x), _p(out), x.numel(), int(x.dtype == torch.float32), dtype_code(out.dtype), _stream())
    _check(rc, "vface_silu")


def cast_f32(x: torch.Tensor, out: torch.Tensor):
    assert x.dtype == torch.float32
    rc = load().vface_cast_f32(_p(x), _p(out), x.numel(), dtype_code(out.dtype), _stream())
    _check(rc, "vface_cast_f32")


def pack_unet_input(x, inv, inpaint, mask, out, *, F, h, w, cpad):
    rc = load().vface_pack_unet_input(_p(x), _p(inv), _p(inpaint), _p(mask), _p(out), F, h, w, cpad,
                                      dtype_code(out.dtype), _stream())
    _check(rc, "vface_pack_unet_input")


def nchw_to_nhwc(x: torch.Tensor, out: torch.Tensor, *, N: int, C_: int, hw: int, cpad: int):
    rc = load().vface_nchw_to_nhwc(_p(x), _p(out), N, C_, hw, cpad, dtype_code(out.dtype), _stream())
    _check(rc, "vface_nchw_to_nhwc")


def nhwc_to_nchw_f32(x: torch.Tensor, out: torch.Tensor, *, N: int, C_: int, hw: int, ldx: int):
    rc = load().vface_nhwc_to_nchw_f32(_p(x), ldx, _p(out), N, C_, hw, _stream())
    _check(rc, "vface_nhwc_to_nchw_f32")


def ddim_step(eps, x, inv, x_prev, *, F, C_, hw, lde, scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, pred_x0=None,
              x_prev_recon=None, noise=None, single_branch=False):
    """``single_branch``: False / 0 = eps is [uncond ; cond ; recon]; True / 1 = one unguided branch (inversion); 2 = [uncond ;
    cond] only (the sampler's recon third left out)."""
    rc = load().vface_ddim_step(_p(eps), lde, _p(x), _p(inv), _p(x_prev), _p(pred_x0), _p(x_prev_recon), F, C_, hw,
                                scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, _p(noise), int(single_branch),
                                _stream())
    _check(rc, "vface_ddim_step")


def copy2d(src, dst, *, rows, cols, ld_src, ld_dst):
    rc = load().vface_copy2d(_p(src), ld_src, _p(dst), ld_dst, rows, cols, dtype_code(src.dtype), _stream())
    _check(rc, "vface_copy2d")


def temporal_gauss(src, dst1, dst2, *, F, n, C_, ld_src, fs_src, ld_dst, fs_dst):
    rc = load().vface_temporal_gauss(_p(src), ld_src, fs_src, _p(dst1), _p(dst2), ld_dst, fs_dst, F, n, C_,
                                     dtype_code(src.dtype), _stream())
    _check(rc, "vface_temporal_gauss")


def adain_fusion(a, b, dst, *, rows, C_, lda, ldb, ldd):
    lib = load()
    ws = torch.empty(int(lib.vface_adain_workspace_bytes(rows, C_)), dtype=torch.uint8, device=a.device)
    rc = lib.vface_adain_fusion(_p(a), lda, _p(b), ldb, _p(dst), ldd, rows, C_, _p(ws), ws.numel(), dtype_code(a.dtype),
                                _stream())
    _check(rc, "vface_adain_fusion")


def softmax_rows(scores: torch.Tensor, out: torch.Tensor, *, M: int, N: int, scale: float, ld_s: Optional[int] = None,
                 ld_p: Optional[int] = None):
    """out[m, :N] = softmax(scores[m, :N] * scale); fp32 in, 16-bit out."""
    rc = load().vface_softmax_rows(_p(scores), ld_s if ld_s is not None else scores.stride(0), _p(out),
                                   ld_p if ld_p is not None else out.stride(0), M, N, scale, dtype_code(out.dtype), _stream())
    _check(rc, "vface_softmax_rows")


def vae_sample(moments: torch.Tensor, noise: Optional[torch.Tensor], z: torch.Tensor, *, F: int, hw: int, zc: int,
               scale: float):
    rc = load().vface_vae_sample(_p(moments), moments.stride(0), _p(noise), _p(z), F, hw, zc, scale, _stream())
    _check(rc, "vface_vae_sample")


def upsample2x_conv3x3(x: torch.Tensor, wt_phases: torch.Tensor, out: torch.Tensor, *, nimg: int, H: int, W: int, cin: int,
                       cout: int, ldx: int, ldy: int, bias=None, rowbias=None, flags: int = 0, colstats=None, out32=None):
    """conv3x3(nearest_upsample2x(x)) as four parity-phase 2x2 convolutions (4/9 of the multiply-adds).
    ``wt_phases``: [4, cout, 4*cin] from ``packing.pack_upsample_phases``; ``out``: [nimg*2H*2W, >= cout]."""
    lib = load()
    for py in (0, 1):
        for px in (0, 1):
            rc = lib.vface_upsample2x_conv3x3_phase(_p(x), ldx, nimg, H, W, cin, _p(wt_phases[2 * py + px]), 4 * cin, cout, py, px,
                                                    _p(bias), _p(rowbias), rowbias.stride(0) if rowbias is not None else 0,
                                                    _p(out), ldy, _p(zeros_page(x.device)), flags, dtype_code(x.dtype),
                                                    _p(colstats), colstats.stride(0) // 2 if colstats is not None else 0,
                                                    _stream(), _s32(None, out32))
            _check(rc, "vface_upsample2x_conv3x3_phase")


def conv3x3_plus_1x1(x: torch.Tensor, x2: torch.Tensor, wt: torch.Tensor, out: torch.Tensor, *, nimg: int, H: int, W: int,
                     cin: int, c2: int, cout: int, ldx: int, ldx2: int, ldy: int, bias=None, rowbias=None, flags: int = 0,
                     colstats=None, split_k: bool = True, out32=None, gn_ab=None, gn_silu: bool = False):
    """out = conv3x3(x) + x2 @ W2^T + bias (a ResBlock's second conv plus its 1x1 shortcut); ``wt``: [cout, 9*cin + c2]."""
    lib = load()
    M, K = nimg * H * W, 9 * cin + c2
    ws, ws_bytes = splitk_workspace(x.device, M, cout, K, flags, H * W) if split_k else (None, 0)
    rc = lib.vface_conv3x3_plus_1x1(_p(x), ldx, nimg, H, W, cin, _p(x2), ldx2, c2, _p(wt), K, cout, _p(bias), _p(rowbias),
                                    rowbias.stride(0) if rowbias is not None else 0, _p(out), ldy, _p(zeros_page(x.device)),
                                    flags, dtype_code(x.dtype), _p(colstats),
                                    colstats.stride(0) // 2 if colstats is not None else 0, _p(ws), ws_bytes, _stream(),
                                    _s32(None, out32, gn_ab, gn_silu))
    _check(rc, "vface_conv3x3_plus_1x1")
