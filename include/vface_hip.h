/* vface_hip.h -- C ABI of libvface_hip.so: the MI355X (gfx950) kernels behind VFace's per-frame DDIM
 * denoising hot path.
 *
 * The reference (Sanoojan/VFace, REFace/) has no FFI: its extension point is Python -- nn.Module
 * forwards and the closure `register_spa_attn_injection` installs on every `attn1`
 * (REFace/ldm/models/pnp_utils.py:57,289-339).  Each entry point below names the reference
 * call site(s) whose device work it replaces; the Python side that binds them (ctypes, raw device
 * pointers + the current HIP stream) is vface_amd/hip.py, and INTEGRATION.md shows the binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless said otherwise; no allocation, no synchronisation, no
 *    global state inside: calls only enqueue kernels on `stream` (a hipStream_t passed as void*), so
 *    they can be captured into a hipGraph; re-entrant per stream.
 *  - `dtype`: 0 = fp16 (the reference's autocast type), 1 = bf16.  Accumulation, softmax and
 *    normalisation statistics are always fp32.
 *  - activations are token-major / NHWC: element (sample b, pixel p, channel c) at
 *    base[(b*HW + p)*ld + c]; `ld` lets a tensor be a channel slice of a wider (concatenated) buffer.
 *  - 16-bit rows must be 16-byte aligned and channel counts multiples of 8.
 *  - return value: 0 = ok; negative = VFACE_ERR_* (nothing was launched).  No exceptions cross the ABI.
 */
#ifndef VFACE_HIP_H
#define VFACE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VFACE_ABI_VERSION 7   /* 7: + the VFACE_TUNE_BIG_W256 / VFACE_TUNE_BIG_W320 flag bits of vface_gemm (the big tile's width: 256 x 256 beside 256 x 320, chosen by the library per launch; same results), nothing else of 6 changed; 6: + the VFACE_TUNE_BIG_TILE / VFACE_TUNE_NO_BIG_TILE flag bits of vface_gemm (csrc/gemm_big.hip: the 256 x 320 tile, chosen by the library from 192 tiles on; same results), nothing else of 5 changed; 5: + vface_st_front, vface_attn_out_ffn_fused, vface_attn_out_ffn_proj_fused, vface_gn_silu_conv3x3_small, vface_linear_small; vface_attention's v_sets carries the live-set count in bits 8..15, vface_pack_unet_input / vface_ddim_step take the two-branch batch; nothing else of 4 changed (4: + vface_ffn_fused, the flow-producer glue, the paste-back entry points) */

#define VFACE_OK 0
#define VFACE_ERR_ARG (-1)
#define VFACE_ERR_ALIGN (-2)
#define VFACE_ERR_SHAPE (-3)
#define VFACE_ERR_DTYPE (-4)
#define VFACE_ERR_LAUNCH (-5)
#define VFACE_ERR_WORKSPACE (-6)

#define VFACE_F16 0
#define VFACE_BF16 1

/* epilogue flags of vface_gemm / vface_conv3x3 */
#define VFACE_EPI_GEGLU 1   /* out[m][c] = (acc_val + b) * gelu(acc_gate + b); Wt rows packed by vface layout */
#define VFACE_EPI_OUT_F32 2 /* store fp32 instead of the 16-bit type */
#define VFACE_TUNE_VARIANT(v) ((v) << 8) /* bits 8..11: force GEMM schedule variant v (1..8); 0 = automatic */
#define VFACE_CONV_PAD_TRAILING 0x40000 /* vface_conv3x3: zero padding (left 0, right 1, top 0, bottom 1) instead of 1 all round */
#define VFACE_TUNE_NO_PERSISTENT 0x10000 /* one workgroup per output tile even where the persistent form would run */
#define VFACE_TUNE_PERSISTENT 0x20000    /* persistent form for a plain GEMM too (default: implicit convolutions only) */
#define VFACE_TUNE_NO_PATCH 0x80000      /* convolutions: never the patch-staged kernel (im2col-style staging, one load per tap) */
#define VFACE_TUNE_PATCH 0x100000        /* convolutions: the patch-staged kernel wherever the shape allows, however small the grid */
#define VFACE_TUNE_GN8 0x8000000          /* A/B: fixed column groups of 8 n-tiles in the plain GEMM's XCD-aware tile order (default: per-launch width) */
#define VFACE_TUNE_NO_Q8 0x4000000        /* A/B: 3x3 convolutions on 8x8 images stay on the im2col kernel (default: four images per workgroup through the patch-staged kernel + split-K reduce) */
#define VFACE_TUNE_PATCH_BN160 0x2000000  /* A/B: the patch-staged kernel's 160-channel tile wherever it divides Cout (default: the width whose one-per-CU grid has the cheaper last round) */
#define VFACE_TUNE_F32_TRANSPOSE 0x1000000 /* A/B: the epilogue's LDS transpose in fp32 even where nothing reads the fp32 sum (default there: 16 bits) */
#define VFACE_TUNE_BIG_TILE 0x10000000     /* vface_gemm: the 256 x 320 tile (csrc/gemm_big.hip) whenever the launch qualifies (N % 320 == 0, K % 64 == 0, 16-bit output, bias / GEGLU / fp32 residual only), however few tiles; default: from 192 tiles on.  Same bits as the 128-row kernel */
#define VFACE_TUNE_NO_BIG_TILE 0x20000000  /* A/B: never the 256 x 320 tile */
#define VFACE_TUNE_BIG_W256 0x200000       /* vface_gemm, big tile: 256 channels per tile wherever N % 256 == 0 (default: the width whose one-workgroup-per-CU grid takes fewer tile-times).  Same bits */
#define VFACE_TUNE_BIG_W320 0x400000       /* A/B: always 320 channels per tile */
/* (VFACE_TUNE_PATCH on vface_gemm: run a plain GEMM with M % 256 == 0, K % 64 == 0, N % 128|160 == 0, no GEGLU / fp32-only output
 *  through the patch-staged kernel's 256-row tile -- same bits as the default kernel, measured 3-18 % SLOWER on the UNet's shapes
 *  (tools/bench_kernels.py "patch256"): an A/B switch, never chosen automatically) */

/* fusion modes of the attn1 hook (pnp_utils.py:133-262) understood by vface_attn1_forward */
#define VFACE_FUSION_NONE 0       /* switch_on == False, or unpatched CrossAttention.forward */
#define VFACE_FUSION_REPLACE 1    /* :133-143 and chunks == 2 (:259-262): q,k of every chunk <- chunk 0 */
#define VFACE_FUSION_LINEAR 2     /* "fft"/"flow_fix"/"fft_vfixed"/"mix": q,k <- own*W_a + chunk0*W_b (folded weights) */

int vface_abi_version(void);
const char* vface_error_string(int code);
/* Always 0 since round 4: the experimental GEMM schedules VFACE_TUNE_VARIANT(1..4, 9, 10) of rounds 1-3 (all measured slower,
 * HISTORY.md) were removed; those variant codes are refused with VFACE_ERR_SHAPE.  Kept so that version-4 callers still link. */
int vface_gemm_variants_built(void);

/* The fp32 residual stream (optional last argument of the GEMM-family calls; NULL = everything 16-bit).
 * The reference adds every residual in the autocast type (openaimodel.py:274 `skip_connection(x) + h`,
 * attention.py:240-242 `attn(...) + x`, :289 `x + x_in`); carrying those sums in fp32 between kernels removes the
 * largest single share of the 16-bit path's whole-network error.  A HOST struct, read during the call:
 *   residual32 / ldr32   the residual operand as fp32 [M][ldr32] -- used INSTEAD of the 16-bit `residual` argument
 *   out32 / ldo32        fp32 [M][ldo32] receives acc + bias + rowbias + residual before the rounding to 16 bits; the
 *                        16-bit output pointer of the call may then be NULL (no 16-bit copy wanted).  With `colstats`
 *                        the statistics are those of the fp32 values (what a GroupNorm reading out32 normalises).
 * 16-byte aligned, ld % 4 == 0; needs N % 8 == 0 and no GEGLU / OUT_F32 epilogue. */
typedef struct vface_stream32 {
    const float* residual32;
    int64_t ldr32;
    float* out32;
    int64_t ldo32;
    /* Convolutions only -- GroupNorm + SiLU fused into the operand path (openaimodel.py:201-205,225-232 `GroupNorm32 -> SiLU ->
     * conv`): in_scale_shift = fp32 [nimg][ld_scale_shift][2] pairs (a, b) per (image, input channel), from
     * vface_groupnorm_coeffs_from_cols; the matrix cores then see act(x * a + b) (act = SiLU if in_silu, else identity),
     * rounded once to the 16-bit type, zero padding applied AFTER the normalisation as in the reference.  Needs a launch
     * that runs the patch-staged kernel (vface_conv_uses_patch_kernel); otherwise VFACE_ERR_SHAPE.  NULL = off. */
    const float* in_scale_shift;
    int64_t ld_scale_shift;
    int in_silu;
} vface_stream32;

/* C[M][N] (+)= A[M][K] * Wt[N][K]^T with fused epilogue.
 * Replaces every nn.Linear / 1x1 conv on the path: to_q/to_k/to_v/to_out (attention.py:161-177),
 * GEGLU + FeedForward (attention.py:37-64), proj_in/proj_out (attention.py:261-276), ResBlock
 * emb_layers / skip_connection (openaimodel.py:218-241), time_embed (openaimodel.py:631-636).
 * A2 (optional): columns [K1, K) of the A operand come from A2[m % a2_row_mod][k - K1] (K1 % 64 == 0).
 * rowbias: fp32 [M/rows_per_sample][ld_rowbias] added per sample (time-embedding / cross-attention vector).
 * colstats (optional): fp32 [ceil(M/64)][ld_colstats][2] receives, per 64-row slice and output column, the (sum, sum of
 * squares) of the values as stored -- the statistics a following GroupNorm needs (vface_groupnorm_finalize_cols). */
int vface_gemm(const void* A, int64_t lda, const void* A2, int64_t lda2, int K1, int a2_row_mod, const void* Wt,
               int64_t ldw, int M, int N, int K, const float* bias, const float* rowbias, int rows_per_sample,
               int ld_rowbias, const void* residual, int64_t ldr, void* C, int64_t ldc, const void* zeros, int flags,
               int dtype, float* colstats, int64_t ld_colstats, void* workspace, int64_t workspace_bytes, void* stream,
               const vface_stream32* s32);

/* Bytes of device scratch a vface_gemm / vface_conv3x3 launch of this shape (M rows = nimg*OH*OW for a convolution,
 * K = 9*Cin) can use to split its K loop over more workgroups when M x N alone would leave most of the 256 CUs idle
 * (fp32 partial tiles summed in a fixed order: results stay reproducible).  0 = this shape is never split.
 * `workspace` may be NULL or smaller: the launch then runs unsplit.  rows_per_sample (OH*OW of a convolution, h*w of a
 * token matrix; <= 1 = unknown) makes the decision a function of the per-sample geometry only, so the bits of a sample
 * do not depend on the batch it is launched in; vface_gemm uses its own rows_per_sample argument the same way. */
int64_t vface_splitk_workspace_bytes(int M, int N, int K, int flags, int rows_per_sample);

/* Y = conv3x3(X) over NHWC, padding 1, stride 1|2, optional nearest x2 upsampling of X first, as an
 * implicit GEMM (nothing materialised).  Wt is packed [Cout][K = 9*Cin]: K order (64-channel chunk, tap, channel)
 * when Cin % 64 == 0, else (tap, channel) -- vface_amd/packing.py::pack_conv3x3.
 * Replaces nn.Conv2d in ResBlock.in_layers/out_layers (openaimodel.py:201-232), Downsample.op (:151-153),
 * Upsample.conv after F.interpolate (:108-118), input_blocks.0 (:668-674) and `out` (:821-825). */
int vface_conv3x3(const void* X, int64_t ldx, int nimg, int H, int W, int Cin, const void* Wt, int64_t ldw, int Cout,
                  int stride, int upsample, const float* bias, const float* rowbias, int ld_rowbias,
                  const void* residual, int64_t ldr, void* Y, int64_t ldy, const void* zeros, int flags, int dtype,
                  float* colstats, int64_t ld_colstats, void* workspace, int64_t workspace_bytes, void* stream,
                  const vface_stream32* s32);

/* Which kernel a vface_conv3x3 / vface_conv3x3_plus_1x1 (window = 3) / vface_upsample2x_conv3x3_phase (window = 2) launch of this
 * geometry runs: 1 = the patch-staged kernel (conv.hip: stride 1, H and W multiples of 16, Cin % 64 == 0, Cout % 160 == 0 or
 * % 128 == 0, 16-bit output, and a grid that is deep enough at the nominal 24-sample batch, or VFACE_TUNE_PATCH), 2 = its
 * 8x8 form (3x3 window on 8 x 8 images, Cout % 128 == 0: four images per workgroup, K split over channel chunks, then the
 * split-K reduce; taken when the caller passes the workspace vface_splitk_workspace_bytes asks for), 0 = the im2col-style
 * implicit GEMM (gemm.hip).  For measurement harnesses (bench.py prices the two kernels separately). */
int vface_conv_uses_patch_kernel(int H, int W, int Cin, int Cout, int window, int stride, int upsample, int flags);

/* Y = conv3x3(X) + X2 W2^T + bias: the second convolution of a ResBlock together with the block's 1x1 shortcut
 * (openaimodel.py:228-232, 274 `skip_connection(x) + h`; diffusionmodules/model.py:137-141 `nin_shortcut`), accumulated in
 * one K loop -- the shortcut's own launch, its 16-bit intermediate and the residual read disappear.  Wt: [Cout][9*Cin + C2],
 * the packed 3x3 weights followed by the 1x1 weights; X2: [nimg*H*W][ldx2 >= C2].  Cin % 64 == 0, C2 % 64 == 0, stride 1. */
int vface_conv3x3_plus_1x1(const void* X, int64_t ldx, int nimg, int H, int W, int Cin, const void* X2, int64_t ldx2, int C2,
                           const void* Wt, int64_t ldw, int Cout, const float* bias, const float* rowbias, int ld_rowbias,
                           void* Y, int64_t ldy, const void* zeros, int flags, int dtype, float* colstats,
                           int64_t ld_colstats, void* workspace, int64_t workspace_bytes, void* stream,
                           const vface_stream32* s32);

/* One output-parity phase of conv3x3(nearest_upsample2x(X)) (Upsample: openaimodel.py:108-118, diffusionmodules/model.py:
 * 55-58).  Every output pixel (2i+py, 2j+px) of the upsampled convolution sees only a 2x2 block of source pixels, so the
 * nine taps collapse -- exactly -- into four with pre-summed weights (vface_amd/packing.py::pack_upsample_phases): 4/9 of
 * the multiply-adds of running the 3x3 window over the upsampled image.  X: [nimg*H*W][ldx >= Cin]; Y: the FULL output
 * [nimg*2H*2W][ldy]; this call writes its rows (2i+py, 2j+px) only; call it for the four (py, px).  Wt: that phase's
 * [Cout][4*Cin] matrix.  No residual, 16-bit output, Cout % 8 == 0.  colstats (optional, H*W % 64 == 0): the
 * [nimg*4*H*W/64][ld_colstats][2] buffer of the FULL output; each phase fills its own quarter of every sample's slices
 * (sums over a sample are what vface_groupnorm_finalize_cols takes, so the slice order inside a sample is free). */
int vface_upsample2x_conv3x3_phase(const void* X, int64_t ldx, int nimg, int H, int W, int Cin, const void* Wt, int64_t ldw,
                                   int Cout, int py, int px, const float* bias, const float* rowbias, int ld_rowbias, void* Y,
                                   int64_t ldy, const void* zeros, int flags, int dtype, float* colstats,
                                   int64_t ld_colstats, void* stream, const vface_stream32* s32);

/* O = softmax(Q K^T * scale) V per (sample, head), streaming softmax, no [n x n] matrix.
 * Replaces attention.py:206-220 / pnp_utils.py:270-285.  Output sample b uses q,k of sample qk_map[b] and
 * v of sample v_map[b] (NULL = identity): the zero-copy form of the hook's q/k/v row assignments. */
int vface_attention(const void* Q, const void* K, const void* V, int64_t ldq, int64_t ldk, int64_t ldv, int64_t bsq,
                    int64_t bsk, int64_t bsv, const int32_t* qk_map, const int32_t* v_map, void* O, int64_t ldo,
                    int64_t bso, int B, int heads, int n, int nk, int dh, float scale, int dtype, int v_sets,
                    int set_stride, void* stream);
/* v_sets > 1 ("shared scores", the "replace" injection pnp_utils.py:133-143 / :259-262 where every chunk takes q,k of
 * chunk 0): B counts the q/k samples; for g < v_sets output sample b + g*set_stride = softmax(q_b k_b^T * scale) v_s,
 * s = v_map[b + g*set_stride] (or b + g*set_stride), the probabilities computed once per (b, head).  Supported for
 * v_sets 2|3 and dh 8|16|32|40 (vface_attention_shared_scores_supported); other shapes: use qk_map.  * Bits 8..15 of v_sets: the number of LIVE sets (0 = all; supported: 2 of 3): sets g >= live are neither read nor written --
 * a batch that came without its last chunk (the sampler's dead-branch elimination) -- while every instruction that touches a live
 * set is that of the full call, so the live outputs are bit-identical to it. */
int vface_attention_shared_scores_supported(int dh, int v_sets);

/* y = LayerNorm(x) * gamma + beta, fp32 statistics (attention.py:231-233).  in_f32: x is the fp32 residual-stream
 * copy [M][ldx] (ldx in floats); y is always 16-bit (its consumer is a GEMM). */
int vface_layernorm(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy, int M,
                    int C, float eps, int in_f32, int dtype, void* stream);

/* GroupNorm statistics (mean, rstd) per (image, group) -> stats[nimg][groups][2] fp32;
 * `partial` is scratch of vface_groupnorm_partial_floats() floats.  util.py:214-216, attention.py:76-77. */
int vface_groupnorm_partial_floats(int nimg, int hw, int C, int groups);
int vface_groupnorm_stats(const void* x, int64_t ldx, int nimg, int hw, int C, int groups, float eps, float* partial,
                          float* stats, int in_f32, int dtype, void* stream);
/* The same statistics from producer-side column sums (colstats of vface_gemm / vface_conv3x3); needs hw % 64 == 0. */
int vface_groupnorm_finalize_cols(const float* colstats, int64_t ld_colstats, int nimg, int hw, int C, int groups, float eps,
                                  float* stats, void* stream);
/* The same statistics folded with the affine parameters into per-(image, channel) pairs ab[img][c] = (a, b), a = rstd * gamma[c],
 * b = beta[c] - mean * a -- what vface_stream32.in_scale_shift takes (fp32 [nimg][C][2]). */
int vface_groupnorm_coeffs_from_cols(const float* colstats, int64_t ld_colstats, int nimg, int hw, int C, int groups, float eps,
                                     const float* gamma, const float* beta, float* ab, void* stream);
/* y = (x - mean) * rstd * gamma + beta, then SiLU if `silu` (openaimodel.py:201-205,225-232).  in_f32 (here and in
 * vface_groupnorm_stats): x is the fp32 residual-stream copy (ldx in floats); y is always 16-bit. */
int vface_groupnorm_apply(const void* x, int64_t ldx, const float* stats, const float* gamma, const float* beta,
                          void* y, int64_t ldy, int nimg, int hw, int C, int groups, int silu, int in_f32, int dtype,
                          void* stream);

/* Flow-guided temporal smoothing of a token-major map [F][h*w][C] (temporal_flow.py:40-53,222-237):
 *   dst[f] = alpha * src[f] + one_minus_alpha * bilinear(src[f-1], (x + dx, y + dy)); dst[0] = src[0]
 * (or warped from `prev` with `flow_prev` when given: the previous rank's boundary frame).
 * flow: fp32 [F-1][2][h][w] in pixels of the map; channel 0 = dx, 1 = dy.  Sampling coordinates follow
 * the reference's fp32 operation order exactly (bit-exact gather indices); flags bit 0 = use CUDA-ATen's
 * multiply-by-reciprocal for the scalar division.  dbg_x0/dbg_y0 (optional, int32 [F-1][h][w]) receive
 * the integer north-west corner of every bilinear footprint. */
int vface_flow_warp(const void* src, int64_t ld_src, int64_t fs_src, const void* prev, int64_t ld_prev,
                    const float* flow, const float* flow_prev, void* dst, int64_t ld_dst, int64_t fs_dst, int F,
                    int h, int w, int C, float alpha, float one_minus_alpha, int flags, int32_t* dbg_x0,
                    int32_t* dbg_y0, int dtype, void* stream);

/* Pixel-resolution optical flow -> the latent-resolution field vface_flow_warp takes.  The reference computes
 * RAFT flow between full-resolution frames (temporal_flow.py:163-188 `return_flow`; call site VFace_inference_batch.py:
 * 550-553) and hands it UNRESIZED to the warp of the 64 x 64 maps, where `grid + flow` fails (temporal_flow.py:43); the
 * resample is defined here: out[p][c][y][x] = mean(flow_px[p][c][y*f .. y*f+f-1][x*f .. x*f+f-1]) / f   (fp32 in and out;
 * H, W multiples of `factor`).  RAFT's own weights are third-party: the flow VALUES are outside this library. */
int vface_flow_to_latent(const float* flow_px, float* out, int pairs, int H, int W, int factor, void* stream);

/* ---- optical-flow producer (temporal_flow.py:27-38, 163-188: torchvision raft_large, 20 updates) -----------------
 * The convolutions and the all-pairs correlation of the RAFT-shaped network run through vface_gemm / vface_conv3x3; these are
 * the HBM-bound steps between them, all on token-major [M = nimg*H*W][ld] buffers of the 16-bit compute type.
 * PARITY UNPINNED: torchvision (pinned 0.14.1 by the reference's environment) is third-party, not under the reference tree and
 * not installed; the network is restated from the published architecture in oracle/raft.py and tested against that.
 *
 * vface_im2col: the KH x KW window (7x7, 1x5, 5x1: more than the implicit GEMM's 9 taps, or a non-square shape) as an explicit
 *   matrix out[m][tap*C + c], zero outside the image; C % 8 == 0, ldo >= KH*KW*C.  OH = (H + 2 pad_y - KH) / stride + 1. */
int vface_im2col(const void* X, int64_t ldx, int nimg, int H, int W, int C, int KH, int KW, int stride, int pad_y, int pad_x, void* out,
                 int64_t ldo, int dtype, void* stream);

/* nn.InstanceNorm2d statistics (per image and channel over hw pixels, biased variance): stats [nimg][C][2] = (mean, rstd);
 * `partial` = vface_channel_stats_partial_floats(nimg, hw, C) floats of scratch (fixed summation order: reproducible). */
int64_t vface_channel_stats_partial_floats(int nimg, int hw, int C);
int vface_channel_stats(const void* x, int64_t ldx, int nimg, int hw, int C, float eps, float* partial, float* stats, int dtype,
                        void* stream);

/* y = act((x - mean) * rstd + residual): stats NULL = no normalisation (the BatchNorm of the context encoder is folded into its
 * convolutions on the host), residual NULL = none; act 0 none | 1 ReLU | 2 tanh | 3 sigmoid; y (16-bit) and / or y32 (fp32). */
int vface_channel_norm_act(const void* x, int64_t ldx, const float* stats, const void* residual, int64_t ldr, void* y, int64_t ldy,
                           float* y32, int64_t ldy32, int64_t M, int hw, int C, int act, int dtype, void* stream);

/* ConvGRU gates: zr [M][2 hidden] = pre-activations of convz | convr; z = sigmoid -> z [M][ldz]; sigmoid(r) * h32 -> rh. */
int vface_gru_gate(const void* zr, int64_t ldzr, const float* h32, void* z, int64_t ldz, void* rh, int64_t ldrh, int64_t M, int hidden,
                   int dtype, void* stream);
/* h32 = (1 - z) h32 + z tanh(q) in place (fp32 master state), 16-bit copies into h16a / h16b (each may be NULL). */
int vface_gru_update(const void* q, int64_t ldq, const void* z, int64_t ldz, float* h32, void* h16a, int64_t lda, void* h16b, int64_t ldb,
                     int64_t M, int hidden, int dtype, void* stream);

/* F.avg_pool2d(x, 2, 2) over the last two dims of fp32 [R][h][w] (the correlation pyramid). */
int vface_avgpool2_f32(const float* x, float* y, int64_t R, int h, int w, void* stream);

/* CorrBlock.index_pyramid: out[m][l*81 + i*9 + j] = scale * bilinear(vols[l][m], (x + flow.x) / 2^l + i - 4, (y + flow.y) / 2^l + j - 4),
 * zero outside (grid_sample, align_corners = True); vols / hs / ws are HOST arrays of `levels` (<= 4) device pointers and sizes,
 * vols[l] = fp32 [M][hs[l]][ws[l]]; flow32 [M][2]; m = (pair*h + y)*w + x. */
int vface_corr_lookup(const float* const* vols, const int* hs, const int* ws, int levels, const float* flow32, int h, int w, float scale,
                      void* out, int64_t ldo, int64_t M, int dtype, void* stream);

/* flow32 [M][2] += delta32[m][0..1] (delta32 NULL: no change); 16-bit copies of the new flow into columns 0..1 of a / b / c. */
int vface_flow_update(float* flow32, const float* delta32, int64_t ldd, void* a, int64_t lda, void* b, int64_t ldb, void* c, int64_t ldc,
                      int64_t M, int dtype, void* stream);

/* upsample_flow: out [B][2][8h][8w] = sum_k softmax_k(mult * mask32[m][k*64 + fy*8 + fx]) * 8 * flow(neighbour k), zero padded. */
int vface_convex_upsample(const float* mask32, int64_t ldm, const float* flow32, float* out, int B, int h, int w, float mult,
                          void* stream);

/* ---- paste-back of a swapped crop into its original frame (VFace_inference_batch.py:597-636) ----------------
 * The reference runs these steps per frame on the host with numpy, Pillow and torchvision; each entry point replaces one of
 * them on device buffers, with Pillow's 8-bit arithmetic restated exactly (oracle/paste.py is pinned against Pillow itself).
 *
 * vface_frame_to_u8: `torch.clamp((x + 1) / 2, 0, 1)` (:597), `255. * x` and `.astype(np.uint8)` (:606-608).
 *   x planar [frames][3][H][W] in [-1, 1], in_kind 0 fp16 | 1 bf16 | 2 fp32: the arithmetic is fp32 (what `--precision full` computes;
 *   a 16-bit input is widened first); in_kind 3 = fp16 input AND fp16 arithmetic, every operation rounded to float16 as torch and
 *   numpy do on the float16 tensor `--precision autocast` (the reference's default) hands over -- pixels can differ by one between
 *   the two.  out interleaved [frames][H][W][3]. */
int vface_frame_to_u8(const void* x, uint8_t* out, int frames, int H, int W, int in_kind, void* stream);

/* One pass of `Image.resize(size, Image.BILINEAR)` on 8-bit RGB (:608, :623; Pillow Resample.c).  axis 0 resamples along x:
 * src [frames][lines][in_n][3] -> dst [frames][lines][out_n][3]; axis 1 along y: src [frames][in_n][lines][3] -> dst
 * [frames][out_n][lines][3] (lines = the row length).  bounds [out_n][2] = (first input sample, taps), kk [out_n][ksize] =
 * the 22-bit fixed-point taps of Pillow's precompute_coeffs + normalize_coeffs_8bpc -- host work, built once per size pair by
 * vface_amd/scripts/paste_back.py `resample_coeffs` and uploaded.  A resize is the x pass then the y pass. */
int vface_resample_u8(const uint8_t* src, uint8_t* dst, int frames, int in_n, int out_n, int lines, int axis, const int32_t* bounds,
                      const int32_t* kk, int ksize, void* stream);

/* `crop.convert('RGBA') + putalpha(255)`, `.transform(frame.size, Image.PERSPECTIVE, coeffs, Image.BILINEAR)` and
 * `frame.alpha_composite(projected)` (:627-633; Pillow Geometry.c / AlphaComposite.c) in one launch, IN PLACE on `frame`
 * [frames][H][W][3] (the background on entry, the pasted frame on return); crop [frames][crop_h][crop_w][3].
 * The eight coefficients (output pixel centre -> crop coordinates, the rows of `inv_transforms_all`, :625) come either from
 * device memory (coeffs_dev [frames][8] doubles) or, for frames == 1, from the host (coeffs_host[8], passed by value);
 * exactly one of the two must be non-NULL. */
int vface_perspective_paste(const uint8_t* crop, int crop_w, int crop_h, uint8_t* frame, int W, int H, int frames,
                            const double* coeffs_dev, const double* coeffs_host, void* stream);

/* `get_tensor()(orig_image)` (ToTensor + Normalize(0.5, 0.5), :48-56, :611) and `transforms.Resize([H, W])` on the tensor
 * (:612 = bilinear, align_corners false, no antialias): frame [frames][H][W][3] uint8 -> out planar [frames][3][OH][OW] fp32
 * in [-1, 1], the VAE encoder's input for the background round trip (:615-618).  fp32 arithmetic in ATen's order; ATen's CPU
 * kernel differs from it by <= 2 ulp (tests bound it at 2e-6). */
int vface_frame_normalise_resize(const uint8_t* frame, int W, int H, float* out, int OW, int OH, int frames, void* stream);

/* The hooked self-attention as one call (pnp_utils.py:94-287, the closure installed on attn1):
 *   x [B][n][d] (already LayerNorm'd), B = chunks * F laid out [uncond ; cond ; recon]
 *   Wqkv [3d][d]  = rows of to_q | to_k | to_v
 *   Wlin [2d][2d] = folded weights of a LINEAR fusion (rows q|k, columns [own x | chunk-0 x]) or NULL
 *   out = to_out(attention) + bo + rowbias[sample] + residual
 * flow != NULL (and h*w == n): chunk 1's fused q,k are smoothed by vface_flow_warp before attention
 * (pnp_utils.py:201-218).  v_fixed: v of chunks 1,2 <- their first frame (fft_vfixed :255-256).
 * halo_qk / halo_flow: previous rank's last-frame fused q|k [n][2d] and the flow into this rank's frame 0.
 * tail_qk (optional out): this rank's last-frame fused q|k, to hand to the next rank.
 * workspace: vface_attn1_workspace_bytes() bytes.  s32: fp32 residual / fp32 output of the out-projection. */
size_t vface_attn1_workspace_bytes(int B, int n, int d, int chunks);
int vface_attn1_forward(const void* x, int64_t ldx, const void* Wqkv, const void* Wlin, const void* Wo,
                        const float* bo, const float* rowbias, int ld_rowbias, const void* residual, int64_t ldr,
                        void* out, int64_t ldo, int B, int n, int d, int heads, int chunks, int fusion,
                        int v_fixed, const float* flow, int h, int w, float alpha, float one_minus_alpha,
                        int warp_flags, const void* halo_qk, const float* halo_flow, void* tail_qk,
                        const int32_t* qk_map, const int32_t* v_map, void* workspace, size_t workspace_bytes,
                        const void* zeros, int dtype, void* stream, const vface_stream32* s32);

/* The FeedForward third of BasicTransformerBlock._forward in ONE launch (REFace/ldm/modules/attention.py:243 `x = ff(norm3(x)) + x`,
 * FeedForward / GEGLU :37-64, LayerNorm :233):  out = W2 (a * gelu(g)) + b2 + x,  [a ; g] = W1 LN(x) + b1.
 * x32: the block's running sum, fp32 [M][ldx] (LayerNorm input and residual).  W1: ff.net[0].proj.weight [8C][C] 16-bit with rows
 * interleaved in 16-row value / gate blocks (the vface_gemm GEGLU packing), b1 in the same order; W2p: ff.net[2].weight [C][4C] with
 * the columns of every aligned 32-block stored in the order 8g + j <- 16 (j >> 2) + 4g + (j & 3) (the k order in which two 16 x 16
 * accumulator tiles of the hidden activations form one k32 MFMA operand: vface_amd/packing.py::pack_ffn_w2).  out16 (16-bit) and /
 * or out32 (fp32) receive the result.  The [M x 4C] hidden matrix never exists; LayerNorm statistics and the GEGLU are fp32, the
 * normalised activations and the hidden activations are rounded to 16 bits once, as in the three-kernel path.
 * Supported: C in {64, 128, 320} (one wave keeps C / 16 x 2 output tiles in registers), M % 128 == 0: vface_ffn_fused_supported();
 * other shapes: VFACE_ERR_SHAPE (the caller runs vface_layernorm + vface_gemm(GEGLU) + vface_gemm). */
int vface_ffn_fused_supported(int64_t M, int C);
int vface_ffn_fused(const float* x32, int64_t ldx, const float* gamma, const float* beta, float eps, const void* W1, const float* b1,
                    const void* W2p, const float* b2, void* out16, int64_t ldo, float* out32, int64_t ldo32, int M, int C,
                    int dtype, void* stream);

/* vface_ffn_fused with the attn1 OUT-PROJECTION in front, one launch (attention.py:239-243; the single-token attn2 is the per-sample
 * row bias):
 *   t1  = att_16 @ Wo^T + bo + rowbias[row / rows_per_sample] + resid          (fp32, never stored)
 *   out = ff.net[2]( GEGLU( ff.net[0]( LayerNorm(t1; gamma, beta, eps) ) ) ) + t1
 * att: [M][C] 16-bit attention output; resid: [M][C] fp32 (the running sum before attn1); rowbias optional fp32
 * [M / rows_per_sample][ld_rowbias] (rows_per_sample % 128 == 0).  WoW1 = [C + 8C][C] 16-bit: to_out's C rows, then ff.net[0]'s
 * 8C rows in the GEGLU-interleaved order of vface_ffn_fused with their k columns permuted like vface_st_front's projection rows
 * (packing.pack_attn_out_ffn).  b1, W2p, b2, out16 / out32, shapes: as vface_ffn_fused. */
int vface_attn_out_ffn_fused(const void* att, int64_t ldatt, const float* resid, int64_t ldr, const float* rowbias, int64_t ld_rowbias,
                             int rows_per_sample, const void* WoW1, const float* bo, const float* gamma, const float* beta, float eps,
                             const float* b1, const void* W2p, const float* b2, void* out16, int64_t ldo, float* out32,
                             int64_t ldo32, int M, int C, int dtype, void* stream);

/* ... and the SpatialTransformer's proj_out + residual behind it (attention.py:286-289 `x = proj_out(x); return x + x_in`), same launch:
 *   y = (out of vface_attn_out_ffn_fused)_16 @ Wpo^T + b_po + x_in
 * WoW1Wp = vface_attn_out_ffn_fused's stream + the C rows of proj_out with their k columns permuted the same way
 * (packing.pack_attn_out_ffn(.., w_proj_out)); x_in: the transformer's fp32 input [M][ld_xin]; y goes to out16 (optional) / out32
 * (optional, at least one); colstats (optional): [M / 64][ld_colstats][2] fp32 (sum, sum of squares) of y's columns per 64-row
 * slice -- what vface_gemm's epilogue hands the next GroupNorm (vface_groupnorm_finalize_cols / _coeffs_from_cols). */
int vface_attn_out_ffn_proj_fused(const void* att, int64_t ldatt, const float* resid, int64_t ldr, const float* rowbias, int64_t ld_rowbias,
                                  int rows_per_sample, const void* WoW1Wp, const float* bo, const float* gamma, const float* beta, float eps,
                                  const float* b1, const void* W2p, const float* b2, const float* b_po, const float* x_in, int64_t ld_xin,
                                  void* out16, int64_t ldo, float* out32, int64_t ldo32, float* colstats, int64_t ld_colstats, int M, int C,
                                  int dtype, void* stream);

/* The UNet's `out` layer in one launch (openaimodel.py:712-716, :905): out = conv3x3( SiLU( x * a[img] + b[img] )_16 ) + bias with Cout
 * in {3, 4}, Cin % 64 == 0; x: [nimg*H*W][ldx] fp32 residual-stream carrier (in_f32) or 16-bit; (a, b): vface_groupnorm_coeffs_from_cols;
 * Wt: [Cout][9*Cin] 16-bit in vface_conv3x3's k order; out: fp32 [nimg*H*W][ldo].  Replaces finalize + apply + an implicit-GEMM whose
 * 128-wide tile is 97 % padding at four output channels. */
int vface_gn_silu_conv3x3_small(const void* x, int64_t ldx, int in_f32, const float* gn_ab, int64_t ld_ab, const void* Wt, const float* bias,
                                float* out, int64_t ldo, int nimg, int H, int W, int Cin, int Cout, int dtype, void* stream);

/* Linear layers on a handful of rows -- the time-embedding chain (openaimodel.py:874-875 `emb = self.time_embed(timestep_embedding(t))`,
 * :264-271 `emb_layers(emb)` = Linear(SiLU(emb)) of every ResBlock as one matrix; util.py:151-171):
 *   out[m][n] = act( sum_k a[m][k] W[n][k] + bias[n] ),  M <= 96, N % 32 == 0, K in {320, 640, 1280}
 * a: [M][lda] 16-bit; act = SiLU when `silu`; out 16-bit, or fp32 when `out_f32`.  Replaces three vface_gemm + two vface_silu
 * launches of a forward (a 128-row GEMM tile is 81 % padding at M = 24) by three; the SiLU now acts on the fp32 sum (one rounding
 * instead of two). */
int vface_linear_small_supported(int M, int N, int K);
int vface_linear_small(const void* a, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out,
                       int64_t ldo, int out_f32, int silu, int M, int N, int K, int dtype, void* stream);

/* Fused FRONT of a SpatialTransformer (attention.py:278-284 norm -> proj_in -> tokens; :239 norm1; :179-183 to_q / to_k / to_v of
 * attn1), one launch for GroupNorm-apply + proj_in + LayerNorm + the attn1 projection on token matrices with C in {64, 128, 320}:
 *   t0  = (x32 * a[img] + b[img])_16 @ W_in^T + b_in            fp32 [M][C]   (the carrier the attn1 out-projection adds to)
 *   qkv = LayerNorm(t0; gamma, beta, eps)_16 @ W_p^T            16-bit [M][ldq], projection column j at qkv[:, j]
 * (a, b) = vface_groupnorm_coeffs_from_cols of x's producer (fp32 pairs [nimg][ld_ab][2]); hw = rows per image, M % 128 == 0,
 * hw % 128 == 0.  Wcat = [C + NQ][C] 16-bit: the C rows of proj_in, then the NQ projection rows with their k columns permuted
 * inside every aligned block of 16: position 8 h + j holds original column 8 (j >> 2) + 4 h + (j & 3) (packing.ffn_w2_perm).
 * Rows [0, rows_full) get projection columns [0, NQ), the others only [nq_lo, NQ) (the hook's "replace": chunks >= 1 project V
 * only, pnp_utils.py:133-142); rows_full % 128 == 0, NQ % 32 == 0, nq_lo % 32 == 0.  ln (optional): LayerNorm(t0) as a 16-bit
 * matrix too (what the dual-source projections of the hook's linear fusions read).  No workspace; capturable. */
int vface_st_front_supported(int64_t M, int C, int hw);
int vface_st_front(const float* x32, int64_t ldx, const float* gn_ab, int64_t ld_ab, int hw, const void* Wcat, const float* b_in,
                   const float* gamma, const float* beta, float eps, float* t0, int64_t ldt0, void* qkv, int64_t ldq, void* ln,
                   int64_t ldln, int M, int C, int NQ, int rows_full, int nq_lo, int dtype, void* stream);

/* fusion="temporal" (pnp_utils.py:59-90,145-154): 5-tap Gaussian (sigma 1, renormalised at the clip ends) over the
 * FRAME axis of src [F][n][C] (chunk 0's q|k), written to dst1 and dst2 (chunk 1 and chunk 2). */
int vface_temporal_gauss(const void* src, int64_t ld_src, int64_t fs_src, void* dst1, void* dst2, int64_t ld_dst,
                         int64_t fs_dst, int F, int n, int C, int dtype, void* stream);
/* fusion="adaIn" (face_swap_utils.py:372-389, normalized=True): per-row AdaIN of a (structure) to b (own) over
 * the channel axis, then division by the GLOBAL unbiased std of the fused tensor: dst [rows][C]. */
size_t vface_adain_workspace_bytes(int64_t rows, int C);
int vface_adain_fusion(const void* a, int64_t lda, const void* b, int64_t ldb, void* dst, int64_t ldd, int64_t rows, int C,
                       void* workspace, size_t workspace_bytes, int dtype, void* stream);

/* Small ops */
/* util.py:151-171 timestep_embedding: out[N][dim] = [cos(t f_i) | sin(t f_i)] */
int vface_timestep_embedding(const int64_t* t, void* out, int N, int dim, int dtype, void* stream);
/* nn.SiLU on a flat buffer (time_embed / emb_layers, openaimodel.py:218-224,631-636); in_f32: x is fp32 */
int vface_silu(const void* x, void* y, int64_t count, int in_f32, int dtype, void* stream);

/* ---- first-stage KL-VAE (ldm/models/autoencoder.py:285-335, diffusionmodules/model.py:368-570) ----
 * The encoder / decoder are sequences of vface_conv3x3 (VFACE_CONV_PAD_TRAILING for Downsample :72-77, `upsample` for
 * Upsample :55-58), vface_groupnorm_* (eps 1e-6, swish) and vface_gemm; two more entry points cover what is new. */
/* P[m][:] = softmax(scores[m][:] * scale), fp32 in, 16-bit out, N <= 16384 (AttnBlock :183-186: one head of 512 channels;
 * its scores come from vface_gemm with VFACE_EPI_OUT_F32). */
int vface_softmax_rows(const float* scores, int64_t ld_s, void* P, int64_t ld_p, int M, int N, float scale, int dtype,
                       void* stream);
/* z[F][zc][hw] (fp32 NCHW) = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scale from moments [F*hw][ld >= 2*zc] (fp32
 * NHWC: mean | logvar); noise NULL = mode.  DiagonalGaussianDistribution.sample (distributions.py:24-37) followed by
 * get_first_stage_encoding's scale_factor. */
int vface_vae_sample(const float* moments, int64_t ld_moments, const float* noise, float* z, int F, int hw, int zc,
                     float scale, void* stream);
int vface_cast_f32(const float* src, void* dst, int64_t count, int dtype, void* stream);
/* ddim_w_inv.py:633,654-655: x_in = cat[cat[x,inp,mask], cat[x,inp,mask], cat[inv_t,inp,mask]] -> NHWC [3F][hw][cpad].
 * inv == NULL: only [uncond ; cond] = [2F][hw][cpad] (the recon third of the batch is a pure sink of the sampler -- its x_prev is
 * dropped, ddim_w_inv.py:703-707,738, and no hook mode reads chunk 2 -- so a caller may leave it out: exact). */
int vface_pack_unet_input(const float* x, const float* inv, const float* inpaint, const float* mask, void* out, int F,
                          int h, int w, int cpad, int dtype, void* stream);
int vface_nchw_to_nhwc(const float* x, void* out, int N, int C, int hw, int cpad, int dtype, void* stream);
int vface_nhwc_to_nchw_f32(const float* x, int64_t ldx, float* out, int N, int C, int hw, void* stream);
/* ddim_w_inv.py:666-667,686-700: guidance + x0 prediction + x_{t-1}; eps = NHWC fp32 UNet output of
 * [uncond ; cond ; recon]; x, inv, outputs NCHW fp32 [F][C][hw].  pred_x0 / x_prev_recon / noise optional.
 * single_branch == 1: eps holds ONE branch [F] and is used unguided -- with a_t = a(t - T/S), a_prev = a(t),
 * sigma 0 this is the inversion update of ddim_invert (ddim_w_inv.py:436-449).  single_branch == 2: eps holds [uncond ; cond]
 * only ([2F]: see vface_pack_unet_input with inv == NULL); x_prev_recon is then not written. */
int vface_ddim_step(const float* eps, int64_t lde, const float* x, const float* inv, float* x_prev, float* pred_x0,
                    float* x_prev_recon, int F, int C, int hw, float scale, float a_t, float a_prev, float sigma_t,
                    float sqrt_one_minus_at, const float* noise, int single_branch, void* stream);
/* strided 2-D copy of 16-bit rows (th.cat([h, hs.pop()], 1), openaimodel.py:898, when not written in place) */
int vface_copy2d(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols, int dtype,
                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VFACE_HIP_H */
