// C ABI of libvface_hip.so (see include/vface_hip.h): argument checking + kernel sequencing only.
#include "../../include/vface_hip.h"

#include <hip/hip_runtime.h>
#include <math.h>

#include "vface_kernels.hpp"

namespace {
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline size_t align256(size_t b) { return (b + 255) & ~size_t(255); }

GemmParams plain_gemm(const void* A, long lda, const void* Wt, long ldw, int M, int N, int K, const float* bias,
                      void* C, long ldc, const void* zeros) {
    GemmParams p{};
    p.mode = 0; p.A = A; p.lda = lda; p.Wt = Wt; p.ldw = ldw; p.Kw = K; p.M = M; p.N = N; p.K = K;
    p.bias = bias; p.C = C; p.ldc = ldc; p.zeros = zeros;
    return p;
}
// the optional fp32 residual stream of a GEMM-family call
inline void apply_s32(GemmParams& p, const vface_stream32* s) {
    if (!s) return;
    if (s->residual32) { p.residual = s->residual32; p.ldr = s->ldr32; p.res_f32 = 1; }
    if (s->out32) { p.C32 = s->out32; p.ldc32 = s->ldo32; }
    if (s->in_scale_shift) { p.gn_ab = s->in_scale_shift; p.ld_gn_ab = s->ld_scale_shift; p.gn_silu = s->in_silu ? 1 : 0; }
}
}  // namespace

extern "C" {

int vface_abi_version(void) { return VFACE_ABI_VERSION; }
int vface_gemm_variants_built(void) { return vf_gemm_variants_built() ? 1 : 0; }

const char* vface_error_string(int code) {
    switch (code) {
        case VFACE_OK: return "ok";
        case VFACE_ERR_ARG: return "invalid argument (null pointer or non-positive size)";
        case VFACE_ERR_ALIGN: return "pointer or leading dimension not aligned (16-bit rows need 16 B, channels % 8)";
        case VFACE_ERR_SHAPE: return "unsupported shape";
        case VFACE_ERR_DTYPE: return "unsupported dtype (0 = fp16, 1 = bf16)";
        case VFACE_ERR_LAUNCH: return "kernel launch failed";
        case VFACE_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}

int vface_gemm(const void* A, int64_t lda, const void* A2, int64_t lda2, int K1, int a2_row_mod, const void* Wt,
               int64_t ldw, int M, int N, int K, const float* bias, const float* rowbias, int rows_per_sample,
               int ld_rowbias, const void* residual, int64_t ldr, void* C, int64_t ldc, const void* zeros, int flags,
               int dtype, float* colstats, int64_t ld_colstats, void* workspace, int64_t workspace_bytes, void* stream,
               const vface_stream32* s32) {
    GemmParams p = plain_gemm(A, lda, Wt, ldw, M, N, K, bias, C, ldc, zeros);
    p.colstats = colstats; p.ld_colstats = ld_colstats;
    p.workspace = static_cast<float*>(workspace); p.workspace_bytes = workspace ? workspace_bytes : 0;
    p.A2 = A2; p.lda2 = lda2; p.K1 = K1; p.a2_row_mod = a2_row_mod;
    p.rowbias = rowbias; p.rows_per_sample = rows_per_sample; p.ld_rowbias = ld_rowbias;
    p.residual = residual; p.ldr = ldr; p.flags = flags;
    apply_s32(p, s32);
    return vf_launch_gemm(p, dtype, S(stream));
}

int vface_conv3x3(const void* X, int64_t ldx, int nimg, int H, int W, int Cin, const void* Wt, int64_t ldw, int Cout,
                  int stride, int upsample, const float* bias, const float* rowbias, int ld_rowbias,
                  const void* residual, int64_t ldr, void* Y, int64_t ldy, const void* zeros, int flags, int dtype,
                  float* colstats, int64_t ld_colstats, void* workspace, int64_t workspace_bytes, void* stream,
                  const vface_stream32* s32) {
    if (nimg <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return VFACE_ERR_ARG;
    if (stride != 1 && stride != 2) return VFACE_ERR_SHAPE;
    if (flags & VFACE_EPI_GEGLU) return VFACE_ERR_SHAPE;
    GemmParams p{};
    p.mode = 1; p.A = X; p.lda = ldx; p.Wt = Wt; p.ldw = ldw; p.Kw = 9 * Cin;
    p.H = H; p.W = W; p.Cin = Cin; p.stride = stride; p.upsample = upsample ? 1 : 0;
    const int VH = upsample ? 2 * H : H, VW = upsample ? 2 * W : W;
    // VFACE_CONV_PAD_TRAILING: zero padding (0,1,0,1) -- F.pad then a padding-0 convolution (diffusionmodules/model.py:72-77)
    p.pad = (flags & VFACE_CONV_PAD_TRAILING) ? 0 : 1;
    p.OH = (VH + p.pad + 1 - 3) / stride + 1; p.OW = (VW + p.pad + 1 - 3) / stride + 1;
    p.M = nimg * p.OH * p.OW; p.N = Cout; p.K = 9 * Cin;
    p.bias = bias; p.rowbias = rowbias; p.rows_per_sample = p.OH * p.OW; p.ld_rowbias = ld_rowbias;
    p.residual = residual; p.ldr = ldr; p.C = Y; p.ldc = ldy; p.zeros = zeros; p.flags = flags;
    p.colstats = colstats; p.ld_colstats = ld_colstats;
    p.workspace = static_cast<float*>(workspace); p.workspace_bytes = workspace ? workspace_bytes : 0;
    apply_s32(p, s32);
    return vf_launch_gemm(p, dtype, S(stream));
}

int vface_conv_uses_patch_kernel(int H, int W, int Cin, int Cout, int window, int stride, int upsample, int flags) {
    if (H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (window != 2 && window != 3)) return 0;
    GemmParams p{};
    p.mode = 1; p.H = H; p.W = W; p.OH = H; p.OW = W; p.Cin = Cin; p.N = Cout; p.stride = stride; p.upsample = upsample ? 1 : 0;
    p.KH = p.KW = window; p.ntaps = window * window; p.pad = 1; p.pad_x = 1; p.flags = flags;
    p.K = p.K1 = p.ntaps * Cin; p.rows_per_sample = H * W; p.M = 24 * H * W;
    return vf_conv_kernel_choice(p, nullptr);     // the launcher's own rule (conv.hip), not a restatement of it
}

int vface_conv3x3_plus_1x1(const void* X, int64_t ldx, int nimg, int H, int W, int Cin, const void* X2, int64_t ldx2, int C2,
                           const void* Wt, int64_t ldw, int Cout, const float* bias, const float* rowbias, int ld_rowbias,
                           void* Y, int64_t ldy, const void* zeros, int flags, int dtype, float* colstats,
                           int64_t ld_colstats, void* workspace, int64_t workspace_bytes, void* stream,
                           const vface_stream32* s32) {
    if (nimg <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || C2 <= 0 || !X2) return VFACE_ERR_ARG;
    if (flags & (VFACE_EPI_GEGLU | VFACE_CONV_PAD_TRAILING)) return VFACE_ERR_SHAPE;
    GemmParams p{};
    p.mode = 1; p.A = X; p.lda = ldx; p.Wt = Wt; p.ldw = ldw; p.Kw = 9 * Cin + C2;
    p.A2 = X2; p.lda2 = ldx2; p.K1 = 9 * Cin;
    p.H = H; p.W = W; p.Cin = Cin; p.stride = 1; p.upsample = 0; p.pad = 1;
    p.OH = H; p.OW = W;
    p.M = nimg * H * W; p.N = Cout; p.K = 9 * Cin + C2;
    p.bias = bias; p.rowbias = rowbias; p.rows_per_sample = H * W; p.ld_rowbias = ld_rowbias;
    p.C = Y; p.ldc = ldy; p.zeros = zeros; p.flags = flags;
    p.colstats = colstats; p.ld_colstats = ld_colstats;
    p.workspace = static_cast<float*>(workspace); p.workspace_bytes = workspace ? workspace_bytes : 0;
    apply_s32(p, s32);
    return vf_launch_gemm(p, dtype, S(stream));
}

int vface_upsample2x_conv3x3_phase(const void* X, int64_t ldx, int nimg, int H, int W, int Cin, const void* Wt, int64_t ldw,
                                   int Cout, int py, int px, const float* bias, const float* rowbias, int ld_rowbias, void* Y,
                                   int64_t ldy, const void* zeros, int flags, int dtype, float* colstats,
                                   int64_t ld_colstats, void* stream, const vface_stream32* s32) {
    if (nimg <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (py & ~1) || (px & ~1)) return VFACE_ERR_ARG;
    if (s32 && s32->residual32) return VFACE_ERR_SHAPE;
    if (flags & (VFACE_EPI_GEGLU | VFACE_EPI_OUT_F32 | VFACE_CONV_PAD_TRAILING)) return VFACE_ERR_SHAPE;
    GemmParams p{};
    p.mode = 1; p.A = X; p.lda = ldx; p.Wt = Wt; p.ldw = ldw; p.Kw = 4 * Cin;
    p.H = H; p.W = W; p.Cin = Cin; p.stride = 1; p.upsample = 0;
    // output parity py sees source rows {i-1, i} (py = 0) or {i, i+1} (py = 1): a 2-tap window with leading pad 1 - py
    p.KH = p.KW = 2; p.ntaps = 4; p.pad = 1 - py; p.pad_x = 1 - px;
    p.OH = H; p.OW = W; p.out_phase = 4 | (py << 1) | px;
    p.M = nimg * H * W; p.N = Cout; p.K = 4 * Cin;
    p.bias = bias; p.rowbias = rowbias; p.rows_per_sample = H * W; p.ld_rowbias = ld_rowbias;
    p.C = Y; p.ldc = ldy; p.zeros = zeros; p.flags = flags;
    p.colstats = colstats; p.ld_colstats = ld_colstats;
    apply_s32(p, s32);
    return vf_launch_gemm(p, dtype, S(stream));
}

int64_t vface_splitk_workspace_bytes(int M, int N, int K, int flags, int rows_per_sample) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return vf_splitk_workspace_bytes(M, N, K, flags, rows_per_sample);
}

int vface_attention(const void* Q, const void* K, const void* V, int64_t ldq, int64_t ldk, int64_t ldv, int64_t bsq,
                    int64_t bsk, int64_t bsv, const int32_t* qk_map, const int32_t* v_map, void* O, int64_t ldo,
                    int64_t bso, int B, int heads, int n, int nk, int dh, float scale, int dtype, int v_sets,
                    int set_stride, void* stream) {
    AttnParams p{};
    p.v_sets = v_sets & 0xFF; p.v_sets_live = (v_sets >> 8) & 0xFF; p.set_stride = set_stride;      // (bits 8..15: live sets)
    p.Q = Q; p.K = K; p.V = V; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.bsq = bsq; p.bsk = bsk; p.bsv = bsv;
    p.qk_map = qk_map; p.v_map = v_map; p.O = O; p.ldo = ldo; p.bso = bso;
    p.B = B; p.heads = heads; p.n = n; p.nk = nk; p.dh = dh; p.scale = scale;
    p.variant = dtype >> 8;  // bits 8+ of dtype: schedule variant (benchmarking); 0 = automatic
    return vf_launch_attention(p, dtype & 0xFF, S(stream));
}

int vface_attention_shared_scores_supported(int dh, int v_sets) {
    return vf_attention_shared_scores_supported(dh, v_sets) ? 1 : 0;
}

int vface_layernorm(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy, int M,
                    int C, float eps, int in_f32, int dtype, void* stream) {
    return vf_launch_layernorm(x, ldx, gamma, beta, y, ldy, M, C, eps, in_f32, dtype, S(stream));
}

int vface_groupnorm_partial_floats(int nimg, int hw, int C, int groups) {
    if (nimg <= 0 || hw <= 0 || groups <= 0) return VFACE_ERR_ARG;
    return vf_gn_partial_floats(nimg, hw, C, groups);
}

int vface_groupnorm_stats(const void* x, int64_t ldx, int nimg, int hw, int C, int groups, float eps, float* partial,
                          float* stats, int in_f32, int dtype, void* stream) {
    return vf_launch_gn_stats(x, ldx, nimg, hw, C, groups, eps, partial, stats, in_f32, dtype, S(stream));
}

int vface_groupnorm_finalize_cols(const float* colstats, int64_t ld_colstats, int nimg, int hw, int C, int groups, float eps,
                                  float* stats, void* stream) {
    return vf_launch_gn_finalize_cols(colstats, ld_colstats, nimg, hw, C, groups, eps, stats, S(stream));
}

int vface_groupnorm_coeffs_from_cols(const float* colstats, int64_t ld_colstats, int nimg, int hw, int C, int groups, float eps,
                                     const float* gamma, const float* beta, float* ab, void* stream) {
    return vf_launch_gn_coeffs_cols(colstats, ld_colstats, nimg, hw, C, groups, eps, gamma, beta, ab, S(stream));
}

int vface_groupnorm_apply(const void* x, int64_t ldx, const float* stats, const float* gamma, const float* beta,
                          void* y, int64_t ldy, int nimg, int hw, int C, int groups, int silu, int in_f32, int dtype,
                          void* stream) {
    return vf_launch_gn_apply(x, ldx, stats, gamma, beta, y, ldy, nimg, hw, C, groups, silu, in_f32, dtype, S(stream));
}

int vface_flow_warp(const void* src, int64_t ld_src, int64_t fs_src, const void* prev, int64_t ld_prev,
                    const float* flow, const float* flow_prev, void* dst, int64_t ld_dst, int64_t fs_dst, int F,
                    int h, int w, int C, float alpha, float one_minus_alpha, int flags, int32_t* dbg_x0,
                    int32_t* dbg_y0, int dtype, void* stream) {
    return vf_launch_flow_warp(src, ld_src, fs_src, prev, ld_prev, flow, flow_prev, dst, ld_dst, fs_dst, F, h, w, C,
                               alpha, one_minus_alpha, flags, dbg_x0, dbg_y0, dtype, S(stream));
}

int vface_flow_to_latent(const float* flow_px, float* out, int pairs, int H, int W, int factor, void* stream) {
    return vf_launch_flow_to_latent(flow_px, out, pairs, H, W, factor, S(stream));
}

int vface_im2col(const void* X, int64_t ldx, int nimg, int H, int W, int C, int KH, int KW, int stride, int pad_y, int pad_x, void* out,
                 int64_t ldo, int dtype, void* stream) {
    return vf_launch_im2col(X, ldx, nimg, H, W, C, KH, KW, stride, pad_y, pad_x, out, ldo, dtype, S(stream));
}

int64_t vface_channel_stats_partial_floats(int nimg, int hw, int C) {
    if (nimg <= 0 || hw <= 0 || C <= 0) return 0;
    return (int64_t)nimg * vf_chan_stats_slices(hw) * C * 2;
}

int vface_channel_stats(const void* x, int64_t ldx, int nimg, int hw, int C, float eps, float* partial, float* stats, int dtype,
                        void* stream) {
    return vf_launch_chan_stats(x, ldx, nimg, hw, C, eps, partial, stats, dtype, S(stream));
}

int vface_channel_norm_act(const void* x, int64_t ldx, const float* stats, const void* residual, int64_t ldr, void* y, int64_t ldy,
                           float* y32, int64_t ldy32, int64_t M, int hw, int C, int act, int dtype, void* stream) {
    return vf_launch_chan_norm_act(x, ldx, stats, residual, ldr, y, ldy, y32, ldy32, M, hw, C, act, dtype, S(stream));
}

int vface_gru_gate(const void* zr, int64_t ldzr, const float* h32, void* z, int64_t ldz, void* rh, int64_t ldrh, int64_t M, int hidden,
                   int dtype, void* stream) {
    return vf_launch_gru_gate(zr, ldzr, h32, z, ldz, rh, ldrh, M, hidden, dtype, S(stream));
}

int vface_gru_update(const void* q, int64_t ldq, const void* z, int64_t ldz, float* h32, void* h16a, int64_t lda, void* h16b, int64_t ldb,
                     int64_t M, int hidden, int dtype, void* stream) {
    return vf_launch_gru_update(q, ldq, z, ldz, h32, h16a, lda, h16b, ldb, M, hidden, dtype, S(stream));
}

int vface_avgpool2_f32(const float* x, float* y, int64_t R, int h, int w, void* stream) {
    return vf_launch_avgpool2_f32(x, y, R, h, w, S(stream));
}

int vface_corr_lookup(const float* const* vols, const int* hs, const int* ws, int levels, const float* flow32, int h, int w, float scale,
                      void* out, int64_t ldo, int64_t M, int dtype, void* stream) {
    return vf_launch_corr_lookup(vols, hs, ws, levels, flow32, h, w, scale, out, ldo, M, dtype, S(stream));
}

int vface_flow_update(float* flow32, const float* delta32, int64_t ldd, void* a, int64_t lda, void* b, int64_t ldb, void* c, int64_t ldc,
                      int64_t M, int dtype, void* stream) {
    return vf_launch_flow_update(flow32, delta32, ldd, a, lda, b, ldb, c, ldc, M, dtype, S(stream));
}

int vface_convex_upsample(const float* mask32, int64_t ldm, const float* flow32, float* out, int B, int h, int w, float mult,
                          void* stream) {
    return vf_launch_convex_upsample(mask32, ldm, flow32, out, B, h, w, mult, S(stream));
}

int vface_frame_to_u8(const void* x, uint8_t* out, int frames, int H, int W, int in_kind, void* stream) {
    return vf_launch_frame_to_u8(x, out, frames, H, W, in_kind, S(stream));
}

int vface_resample_u8(const uint8_t* src, uint8_t* dst, int frames, int in_n, int out_n, int lines, int axis, const int32_t* bounds,
                      const int32_t* kk, int ksize, void* stream) {
    return vf_launch_resample_u8(src, dst, frames, in_n, out_n, lines, axis, bounds, kk, ksize, S(stream));
}

int vface_perspective_paste(const uint8_t* crop, int crop_w, int crop_h, uint8_t* frame, int W, int H, int frames,
                            const double* coeffs_dev, const double* coeffs_host, void* stream) {
    return vf_launch_perspective_paste(crop, crop_w, crop_h, frame, W, H, frames, coeffs_dev, coeffs_host, S(stream));
}

int vface_frame_normalise_resize(const uint8_t* frame, int W, int H, float* out, int OW, int OH, int frames, void* stream) {
    return vf_launch_frame_normalise_resize(frame, W, H, out, OW, OH, frames, S(stream));
}

size_t vface_attn1_workspace_bytes(int B, int n, int d, int chunks) {
    if (B <= 0 || n <= 0 || d <= 0 || chunks <= 0) return 0;
    const size_t F = (size_t)B / chunks;
    return align256((size_t)B * n * 3 * d * 2) + align256(F * n * 2 * d * 2) + align256((size_t)B * n * d * 2);
}

int vface_attn1_forward(const void* x, int64_t ldx, const void* Wqkv, const void* Wlin, const void* Wo,
                        const float* bo, const float* rowbias, int ld_rowbias, const void* residual, int64_t ldr,
                        void* out, int64_t ldo, int B, int n, int d, int heads, int chunks, int fusion,
                        int v_fixed, const float* flow, int h, int w, float alpha, float one_minus_alpha,
                        int warp_flags, const void* halo_qk, const float* halo_flow, void* tail_qk,
                        const int32_t* qk_map, const int32_t* v_map, void* workspace, size_t workspace_bytes,
                        const void* zeros, int dtype, void* stream, const vface_stream32* s32) {
    if (!x || !Wqkv || !Wo || (!out && !(s32 && s32->out32)) || !workspace || !zeros) return VFACE_ERR_ARG;
    if (B <= 0 || n <= 0 || d <= 0 || heads <= 0 || chunks <= 0 || d % heads) return VFACE_ERR_ARG;
    if (dtype != VFACE_F16 && dtype != VFACE_BF16) return VFACE_ERR_DTYPE;
    if (fusion != VFACE_FUSION_NONE && B % chunks) return VFACE_ERR_SHAPE;
    if (workspace_bytes < vface_attn1_workspace_bytes(B, n, d, chunks)) return VFACE_ERR_WORKSPACE;
    if (fusion == VFACE_FUSION_REPLACE && !qk_map) return VFACE_ERR_ARG;
    if (fusion == VFACE_FUSION_LINEAR && (!Wlin || (d % 64))) return VFACE_ERR_SHAPE;
    if (v_fixed && !v_map) return VFACE_ERR_ARG;
    const bool warp = fusion == VFACE_FUSION_LINEAR && flow != nullptr;
    if (warp && (h * w != n || chunks < 2 || chunks > 3)) return VFACE_ERR_SHAPE;      // (2: a batch without the recon third)
    hipStream_t st = S(stream);
    const size_t esz = 2;
    const long F = B / chunks, Fn = F * n;
    char* ws = static_cast<char*>(workspace);
    char* qkv = ws;
    char* T = qkv + align256((size_t)B * n * 3 * d * esz);
    char* att = T + align256((size_t)Fn * 2 * d * esz);
    const char* xb = static_cast<const char*>(x);
    const char* wq = static_cast<const char*>(Wqkv);
    int rc;
    auto at = [&](char* base, long row, long col, long ld) { return base + ((size_t)row * ld + col) * esz; };

    if (fusion == VFACE_FUSION_NONE) {
        GemmParams p = plain_gemm(x, ldx, Wqkv, d, B * n, 3 * d, d, nullptr, qkv, 3 * d, zeros);
        if ((rc = vf_launch_gemm(p, dtype, st))) return rc;
    } else {
        // chunk 0: q, k, v as projected (pnp_utils.py:106,127-128)
        GemmParams p0 = plain_gemm(x, ldx, Wqkv, d, (int)Fn, 3 * d, d, nullptr, qkv, 3 * d, zeros);
        if ((rc = vf_launch_gemm(p0, dtype, st))) return rc;
        // other chunks: v only -- their q, k are overwritten by the fusion
        GemmParams pv = plain_gemm(xb + (size_t)Fn * ldx * esz, ldx, wq + (size_t)2 * d * d * esz, d, (int)(B * n - Fn),
                                   d, d, nullptr, at(qkv, Fn, 2 * d, 3 * d), 3 * d, zeros);
        if ((rc = vf_launch_gemm(pv, dtype, st))) return rc;
        if (fusion == VFACE_FUSION_LINEAR) {
            // fused q|k of chunk c >= 1: [x_c | x_0] (K = 2d) times the folded weights
            for (int c = 1; c < chunks; ++c) {
                const bool to_tmp = warp && c == 1;
                GemmParams pf = plain_gemm(xb + (size_t)c * Fn * ldx * esz, ldx, Wlin, 2 * d, (int)Fn, 2 * d, 2 * d,
                                           nullptr, to_tmp ? T : at(qkv, c * Fn, 0, 3 * d), to_tmp ? 2 * d : 3 * d,
                                           zeros);
                pf.A2 = x; pf.lda2 = ldx; pf.K1 = d; pf.a2_row_mod = 0;
                if ((rc = vf_launch_gemm(pf, dtype, st))) return rc;
            }
            if (warp) {
                if (tail_qk) {
                    rc = vf_launch_copy2d(T + (size_t)(F - 1) * n * 2 * d * esz, 2 * d, tail_qk, 2 * d, n, 2 * d, dtype, st);
                    if (rc) return rc;
                }
                rc = vf_launch_flow_warp(T, 2 * d, (long)n * 2 * d, halo_qk, 2 * d, flow, halo_flow,
                                         at(qkv, Fn, 0, 3 * d), 3 * d, (long)n * 3 * d, (int)F, h, w, 2 * d, alpha,
                                         one_minus_alpha, warp_flags, nullptr, nullptr, dtype, st);
                if (rc) return rc;
            }
        }
    }
    AttnParams a{};
    a.Q = qkv; a.K = qkv + (size_t)d * esz; a.V = qkv + (size_t)2 * d * esz;
    a.ldq = a.ldk = a.ldv = 3 * d; a.bsq = a.bsk = a.bsv = (long)n * 3 * d;
    a.qk_map = (fusion == VFACE_FUSION_REPLACE) ? qk_map : nullptr;
    a.v_map = v_fixed ? v_map : nullptr;
    a.O = att; a.ldo = d; a.bso = (long)n * d;
    a.B = B; a.heads = heads; a.n = n; a.nk = n; a.dh = d / heads;
    if (fusion == VFACE_FUSION_REPLACE && vf_attention_shared_scores_supported(d / heads, chunks)) {
        // every chunk attends with q,k of chunk 0 (pnp_utils.py:136-142): softmax once per frame, `chunks` value sets
        a.qk_map = nullptr; a.B = (int)F; a.v_sets = chunks; a.set_stride = (int)F;
    }
    a.scale = 1.0f / sqrtf((float)(d / heads));
    if ((rc = vf_launch_attention(a, dtype, st))) return rc;
    GemmParams po = plain_gemm(att, d, Wo, d, B * n, d, d, bo, out, ldo, zeros);
    po.rowbias = rowbias; po.rows_per_sample = n; po.ld_rowbias = ld_rowbias;
    po.residual = residual; po.ldr = ldr;
    apply_s32(po, s32);
    return vf_launch_gemm(po, dtype, st);
}

int vface_ffn_fused_supported(int64_t M, int C) { return vf_ffn_fused_supported((long)M, C) ? 1 : 0; }

int vface_ffn_fused(const float* x32, int64_t ldx, const float* gamma, const float* beta, float eps, const void* W1, const float* b1,
                    const void* W2p, const float* b2, void* out16, int64_t ldo, float* out32, int64_t ldo32, int M, int C,
                    int dtype, void* stream) {
    FfnParams p{};
    p.x32 = x32; p.ldx = ldx; p.gamma = gamma; p.beta = beta; p.eps = eps; p.W1 = W1; p.b1 = b1; p.W2p = W2p; p.b2 = b2;
    p.out16 = out16; p.ldo = ldo; p.out32 = out32; p.ldo32 = ldo32; p.M = M; p.C = C;
    return vf_launch_ffn_fused(p, dtype, S(stream));
}

int vface_attn_out_ffn_fused(const void* att, int64_t ldatt, const float* resid, int64_t ldr, const float* rowbias, int64_t ld_rowbias,
                             int rows_per_sample, const void* WoW1, const float* bo, const float* gamma, const float* beta, float eps,
                             const float* b1, const void* W2p, const float* b2, void* out16, int64_t ldo, float* out32,
                             int64_t ldo32, int M, int C, int dtype, void* stream) {
    if (!att) return VFACE_ERR_ARG;
    FfnParams p{};
    p.att = att; p.ldatt = ldatt; p.resid = resid; p.ldr = ldr; p.rowbias = rowbias; p.ld_rowbias = ld_rowbias;
    p.rows_per_sample = rows_per_sample; p.bo = bo; p.gamma = gamma; p.beta = beta; p.eps = eps; p.W1 = WoW1; p.b1 = b1; p.W2p = W2p;
    p.b2 = b2; p.out16 = out16; p.ldo = ldo; p.out32 = out32; p.ldo32 = ldo32; p.M = M; p.C = C;
    return vf_launch_ffn_fused(p, dtype, S(stream));
}

int vface_attn_out_ffn_proj_fused(const void* att, int64_t ldatt, const float* resid, int64_t ldr, const float* rowbias, int64_t ld_rowbias,
                                  int rows_per_sample, const void* WoW1Wp, const float* bo, const float* gamma, const float* beta, float eps,
                                  const float* b1, const void* W2p, const float* b2, const float* b_po, const float* x_in, int64_t ld_xin,
                                  void* out16, int64_t ldo, float* out32, int64_t ldo32, float* colstats, int64_t ld_colstats, int M, int C,
                                  int dtype, void* stream) {
    if (!att || !x_in || !b_po) return VFACE_ERR_ARG;
    FfnParams p{};
    p.att = att; p.ldatt = ldatt; p.resid = resid; p.ldr = ldr; p.rowbias = rowbias; p.ld_rowbias = ld_rowbias;
    p.rows_per_sample = rows_per_sample; p.bo = bo; p.gamma = gamma; p.beta = beta; p.eps = eps; p.W1 = WoW1Wp; p.b1 = b1; p.W2p = W2p;
    p.b2 = b2; p.out16 = out16; p.ldo = ldo; p.out32 = out32; p.ldo32 = ldo32; p.M = M; p.C = C;
    p.Wpo_in_stream = 1; p.b_po = b_po; p.x_in = x_in; p.ld_xin = ld_xin; p.colstats = colstats; p.ld_colstats = ld_colstats;
    return vf_launch_ffn_fused(p, dtype, S(stream));
}

int vface_gn_silu_conv3x3_small(const void* x, int64_t ldx, int in_f32, const float* gn_ab, int64_t ld_ab, const void* Wt, const float* bias,
                                float* out, int64_t ldo, int nimg, int H, int W, int Cin, int Cout, int dtype, void* stream) {
    OutConvParams p{};
    p.x = x; p.ldx = ldx; p.in_f32 = in_f32; p.ab = gn_ab; p.ld_ab = ld_ab; p.Wt = Wt; p.bias = bias; p.out = out; p.ldo = ldo;
    p.nimg = nimg; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    return vf_launch_out_conv(p, dtype, S(stream));
}

int vface_linear_small_supported(int M, int N, int K) { return vf_linear_small_supported(M, N, K) ? 1 : 0; }

int vface_linear_small(const void* a, int64_t lda, const void* W, int64_t ldw, const float* bias, void* out,
                       int64_t ldo, int out_f32, int silu, int M, int N, int K, int dtype, void* stream) {
    LinearSmallParams p{};
    p.a = a; p.lda = lda; p.W = W; p.ldw = ldw; p.bias = bias; p.out = out;
    p.ldo = ldo; p.out_f32 = out_f32; p.silu = silu; p.M = M; p.N = N; p.K = K;
    return vf_launch_linear_small(p, dtype, S(stream));
}

int vface_st_front_supported(int64_t M, int C, int hw) { return vf_st_front_supported((long)M, C, hw) ? 1 : 0; }

int vface_st_front(const float* x32, int64_t ldx, const float* gn_ab, int64_t ld_ab, int hw, const void* Wcat, const float* b_in,
                   const float* gamma, const float* beta, float eps, float* t0, int64_t ldt0, void* qkv, int64_t ldq, void* ln,
                   int64_t ldln, int M, int C, int NQ, int rows_full, int nq_lo, int dtype, void* stream) {
    StFrontParams p{};
    p.x32 = x32; p.ldx = ldx; p.ab = gn_ab; p.ld_ab = ld_ab; p.hw = hw; p.Wcat = Wcat; p.b_in = b_in; p.gamma = gamma; p.beta = beta;
    p.eps = eps; p.t0 = t0; p.ldt0 = ldt0; p.qkv = qkv; p.ldq = ldq; p.ln = ln; p.ldln = ldln; p.M = M; p.C = C; p.NQ = NQ;
    p.rows_full = rows_full; p.nq_lo = nq_lo;
    return vf_launch_st_front(p, dtype, S(stream));
}

int vface_temporal_gauss(const void* src, int64_t ld_src, int64_t fs_src, void* dst1, void* dst2, int64_t ld_dst,
                         int64_t fs_dst, int F, int n, int C, int dtype, void* stream) {
    return vf_launch_temporal_gauss(src, ld_src, fs_src, dst1, dst2, ld_dst, fs_dst, F, n, C, dtype, S(stream));
}
size_t vface_adain_workspace_bytes(int64_t rows, int C) { return rows > 0 && C > 0 ? vf_adain_workspace_bytes(rows, C) : 0; }
int vface_adain_fusion(const void* a, int64_t lda, const void* b, int64_t ldb, void* dst, int64_t ldd, int64_t rows, int C,
                       void* workspace, size_t workspace_bytes, int dtype, void* stream) {
    if (rows > 0 && C > 0 && workspace_bytes < vf_adain_workspace_bytes(rows, C)) return VFACE_ERR_WORKSPACE;
    return vf_launch_adain(a, lda, b, ldb, dst, ldd, rows, C, workspace, dtype, S(stream));
}

int vface_timestep_embedding(const int64_t* t, void* out, int N, int dim, int dtype, void* stream) {
    return vf_launch_timestep_embedding(reinterpret_cast<const long long*>(t), out, N, dim, dtype, S(stream));
}
int vface_softmax_rows(const float* scores, int64_t ld_s, void* P, int64_t ld_p, int M, int N, float scale, int dtype, void* stream) {
    return vf_launch_softmax_rows(scores, ld_s, P, ld_p, M, N, scale, dtype, S(stream));
}

int vface_vae_sample(const float* moments, int64_t ld_moments, const float* noise, float* z, int F, int hw, int zc,
                     float scale, void* stream) {
    return vf_launch_vae_sample(moments, ld_moments, noise, z, F, hw, zc, scale, S(stream));
}

int vface_silu(const void* x, void* y, int64_t count, int in_f32, int dtype, void* stream) {
    return vf_launch_silu(x, y, count, in_f32, dtype, S(stream));
}
int vface_cast_f32(const float* src, void* dst, int64_t count, int dtype, void* stream) {
    return vf_launch_cast(src, dst, count, dtype, S(stream));
}
int vface_pack_unet_input(const float* x, const float* inv, const float* inpaint, const float* mask, void* out, int F,
                          int h, int w, int cpad, int dtype, void* stream) {
    return vf_launch_pack_input(x, inv, inpaint, mask, out, F, h, w, cpad, dtype, S(stream));
}
int vface_nchw_to_nhwc(const float* x, void* out, int N, int C, int hw, int cpad, int dtype, void* stream) {
    return vf_launch_nchw_to_nhwc(x, out, N, C, hw, cpad, dtype, S(stream));
}
int vface_nhwc_to_nchw_f32(const float* x, int64_t ldx, float* out, int N, int C, int hw, void* stream) {
    return vf_launch_nhwc_to_nchw_f32(x, ldx, out, N, C, hw, S(stream));
}
int vface_ddim_step(const float* eps, int64_t lde, const float* x, const float* inv, float* x_prev, float* pred_x0,
                    float* x_prev_recon, int F, int C, int hw, float scale, float a_t, float a_prev, float sigma_t,
                    float sqrt_one_minus_at, const float* noise, int single_branch, void* stream) {
    return vf_launch_ddim_step(eps, lde, x, inv, x_prev, pred_x0, x_prev_recon, F, C, hw, scale, a_t, a_prev, sigma_t,
                               sqrt_one_minus_at, noise, single_branch, S(stream));
}
int vface_copy2d(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int cols, int dtype,
                 void* stream) {
    return vf_launch_copy2d(src, ld_src, dst, ld_dst, rows, cols, dtype, S(stream));
}

}  // extern "C"
