// Internal launch interface between the C ABI (capi.cpp / include/vface_hip.h) and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting of a kernel: one bit per device ordinal, set with an
// atomic OR (two host threads launching the same kernel for the first time both set the attribute -- idempotent -- and neither
// skips it; a process that drives a second GPU sets it there too)
struct VfOncePerDevice {
    std::atomic<unsigned long long> done{0};
    bool set_lds(const void* kern, int bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        const unsigned long long bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return true;
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
        done.fetch_or(bit, std::memory_order_release);
        return true;
    }
};

enum { GEMM_GEGLU = 1, GEMM_OUT_F32 = 2, GEMM_NO_XCD_REMAP = 0x1000, GEMM_NO_SETPRIO = 0x2000, GEMM_NARROW_EPILOGUE = 0x8000, GEMM_NO_PERSIST = 0x10000, GEMM_PERSIST = 0x20000,
       GEMM_NO_PATCH = 0x80000, GEMM_PATCH = 0x100000, GEMM_GN8 = 0x8000000 /* A/B: fixed column groups of 8 n-tiles in the plain GEMM's tile order */, GEMM_NO_Q8 = 0x4000000 /* A/B: the 8x8 level stays on the im2col kernel */, GEMM_PATCH_BN160 = 0x2000000 /* A/B: the patch kernel's 160-wide tile wherever it divides N */, GEMM_F32_TRANSPOSE = 0x1000000 /* A/B: epilogue transposes in fp32 even where 16 bits would do */,
       GEMM_BIG = 0x10000000 /* take gemm_big.hip's 256 x 320 tile whenever the launch qualifies */, GEMM_NO_BIG = 0x20000000 /* A/B: never */,
       GEMM_BIG_W256 = 0x200000 /* gemm_big.hip: the 256-channel tile width wherever N allows it */, GEMM_BIG_W320 = 0x400000 /* ... never */ };  // (0x40000 = VFACE_CONV_PAD_TRAILING)  // bits 8..11 of flags: forced schedule variant (0 = automatic)

struct GemmParams {
    int mode;  // 0: plain A[M][K]; 1: implicit 3x3 conv over NHWC
    const void* A;
    long lda;  // plain: row stride; conv: pixel stride (>= Cin) in elements
    const void* A2;  // optional second source for k >= K1 (plain mode), row = m % a2_row_mod
    long lda2;
    int K1, a2_row_mod;
    const void* Wt;  // [N][ldw]
    long ldw;
    int Kw;  // valid k columns of Wt (multiple of 8)
    int M, N, K;
    // conv geometry (stored input H x W, output OH x OW)
    int H, W, Cin, OH, OW, stride, upsample;
    int pad;  // leading (top) zero padding: 1, or 0 for the VAE encoder's (0,1,0,1) stride-2 convolution
    int pad_x;           // leading (left) zero padding; KH x KW window (ntaps = KH*KW <= 9; 0 = the launcher fills 3x3)
    int KH, KW, ntaps;
    int out_phase;       // 0, or 4 | (py << 1) | px: output rows are the (py, px) parity phase of a 2OH x 2OW image
    const float* bias;     // [N] or null
    const float* rowbias;  // [M / rows_per_sample][ld_rowbias] or null
    int rows_per_sample;
    int ld_rowbias;
    const void* residual;  // [M][ldr] or null
    long ldr;
    void* C;  // [M][ldc] 16-bit (or fp32 with GEMM_OUT_F32); may be null when C32 is given
    long ldc;
    // fp32 residual stream: `residual` is fp32 [M][ldr] when res_f32; C32 (optional) receives the fp32 sum
    // bias + rowbias + residual + acc BEFORE the rounding to 16 bits -- the carrier the next residual add / norm reads
    int res_f32;
    float* C32;
    long ldc32;
    // fused input normalisation (patch-staged convolution only): the operand the matrix cores see is
    // act(x * a[img][ci] + b[img][ci]) with (a, b) fp32 pairs at gn_ab[(img * ld_gn_ab + ci) * 2]; act = SiLU if gn_silu
    const float* gn_ab;
    long ld_gn_ab;
    int gn_silu;
    unsigned gn_ab_bytes;   // filled by the launcher
    const void* zeros;  // >= 16 zero bytes, 16-B aligned
    int flags;
    unsigned a_bytes, a2_bytes, w_bytes;  // filled by the launcher: extents of the operand views
    float* colstats;     // optional [ceil(M/64)][ld_colstats][2] per-64-row-slice column (sum, sumsq) of the stored values
    long ld_colstats;
    // split-K (launcher-chosen when a workspace is supplied and the tile grid underfills the chip): fp32 partial tiles
    // [split_k][M][N] in `workspace`, summed in split order by a second kernel that also runs the epilogue
    float* workspace;
    long workspace_bytes;
    int split_k, kt_per_split;  // filled by the launcher
    int tile_group;             // patch-staged kernel: n-tiles per column group of its XCD-aware tile order (launcher; 0 = 8)
    unsigned res_bytes, c_bytes, c32_bytes;  // gemm_big.hip (filled by its launcher): extents of the fp32 residual view, of the 16-bit output view and of the fp32 carrier view
};
long vf_splitk_workspace_bytes(int M, int N, int K, int flags, int rows_per_sample);
bool vf_gemm_variants_built();
int vf_launch_gemm(const GemmParams& p, int dtype, hipStream_t stream);
// conv.hip: the patch-staged stride-1 convolution; vf_conv_patch_tile = 0 (not a patch shape) | 160 | 128
int vf_conv_patch_tile(const GemmParams& p);
int vf_conv_kernel_choice(const GemmParams& p, int* arg);                        // 0 im2col | 1 patch (*arg = tile width) | 2 8x8 form (*arg = K split)
int vf_launch_conv_patch(const GemmParams& p, int dtype, hipStream_t stream);
int vf_gemm_patch_tile(const GemmParams& p);                                   // plain GEMM through the patch kernel's 256-row tile: 0 | 160 | 128
int vf_launch_gemm_patch(const GemmParams& p, int dtype, hipStream_t stream);
bool vf_gemm_big_ok(const GemmParams& p);
bool vf_gemm_big_choice(const GemmParams& p);                                  // ... and the rule says it should                                      // gemm_big.hip: the 256 x 320 tile takes this plain-GEMM launch
int vf_launch_gemm_big(const GemmParams& p, int dtype, hipStream_t stream);
bool vf_conv_in16_ok(const GemmParams& p);                                       // inconv.hip: the 16-stored-channel input convolution (K = 144 in one MFMA pass)
int vf_launch_conv_in16(const GemmParams& p, int dtype, hipStream_t stream);
int vf_conv_q8_split(const GemmParams& p);                                      // the 8x8 level through the patch kernel: 0 | K split
int vf_launch_conv_q8(const GemmParams& p, int dtype, hipStream_t stream);     // (main pass only: the caller runs the split-K reduce)
bool vf_attention_shared_scores_supported(int dh, int v_sets);

// ffn.hip: fused LayerNorm -> ff.net[0] (GEGLU) -> ff.net[2] -> + x over token matrices with C in {64, 128, 320}, M % 128 == 0
struct FfnParams {
    const float* x32; long ldx;      // [M][C] fp32: the block's running sum (LayerNorm input AND residual)
    const float* gamma; const float* beta; float eps;
    const void* W1;                  // [8C][C] 16-bit, GEGLU rows interleaved in 16-row value / gate blocks (packing.pack_geglu)
    const float* b1;                 // [8C] in the same row order
    const void* W2p;                 // [C][4C] 16-bit, columns permuted per 32 (packing.pack_ffn_w2)
    const float* b2;                 // [C]
    void* out16; long ldo;           // [M][C] 16-bit result (or null)
    float* out32; long ldo32;        // optional fp32 result
    int M, C;
    // att != null: the attn1 out-projection runs in front (ffn.hip "PRE"): x32 is unused, the running sum is formed in-kernel as
    // t1 = att @ Wo^T + bo + rowbias[row / rows_per_sample] + resid; W1 is then [C + 8C][C]: to_out's rows, then ff.net[0]'s
    // (GEGLU-interleaved rows, k columns in ffn_w2_perm order)
    const void* att; long ldatt;     // [M][C] 16-bit attention output
    const float* resid; long ldr;    // [M][C] fp32: the running sum before attn1 (t0)
    const float* bo;                 // [C] to_out bias
    const float* rowbias; long ld_rowbias; int rows_per_sample;   // optional fp32 [M / rows_per_sample][ld]: attn2's contribution
    // Wpo_in_stream (with att): the SpatialTransformer's proj_out runs behind (ffn.hip "POST"): W1 is [C + 8C + C][C] -- proj_out's rows
    // last, k columns in ffn_w2_perm order --, the outputs are y = proj_out(t3) + b_po + x_in and its per-64-row column statistics
    int Wpo_in_stream;
    const float* b_po;               // [C]
    const float* x_in; long ld_xin;  // [M][C] fp32: the SpatialTransformer's input
    float* colstats; long ld_colstats;   // optional [M / 64][ld][2] (sum, sum of squares) of y
};
bool vf_ffn_fused_supported(long M, int C);
int vf_launch_ffn_fused(const FfnParams& p, int dtype, hipStream_t stream);

// linear_small.hip: out[M][N] = act(a[M][K] W[N][K]^T + bias) on M <= 96 rows (the time-embedding chain)
struct LinearSmallParams {
    const void* a; long lda;         // [M][K] 16-bit
    const void* W; long ldw;         // [N][K] 16-bit
    const float* bias;               // [N] or null
    void* out; long ldo; int out_f32;
    int silu;
    int M, N, K;
};
bool vf_linear_small_supported(int M, int N, int K);
int vf_launch_linear_small(const LinearSmallParams& p, int dtype, hipStream_t stream);

// stfront.hip: GroupNorm-apply -> proj_in -> (t0 out) -> LayerNorm -> attn1 projection, one launch; C in {64, 128, 320}
struct StFrontParams {
    const float* x32; long ldx;      // [M][C] fp32: the SpatialTransformer's input (residual-stream carrier)
    const float* ab; long ld_ab;     // GroupNorm (scale, shift) pairs [nimg][ld_ab][2] (vface_groupnorm_coeffs_from_cols)
    int hw;                          // rows per image (a multiple of 128)
    const void* Wcat;                // [C + NQ][C] 16-bit: proj_in rows, then the projection rows with k columns in ffn_w2_perm order
    const float* b_in;               // [C] proj_in bias
    const float* gamma; const float* beta; float eps;   // norm1
    float* t0; long ldt0;            // [M][C] fp32 out: proj_in(GroupNorm(x)) + bias
    void* qkv; long ldq;             // [M][ldq] 16-bit out: projection column j at qkv[:, j]
    void* ln; long ldln;             // optional [M][C] 16-bit out: LayerNorm(t0)
    int M, C, NQ;
    int rows_full, nq_lo;            // rows [0, rows_full): projection columns [0, NQ); the other rows: columns [nq_lo, NQ)
    int gridA;                       // filled by the launcher
};
bool vf_st_front_supported(long M, int C, int hw);
int vf_launch_st_front(const StFrontParams& p, int dtype, hipStream_t stream);

// outconv.hip: GroupNorm-apply + SiLU + 3x3 convolution to 3 / 4 output channels in one launch (the UNet's `out` layer)
struct OutConvParams {
    const void* x; long ldx; int in_f32;   // [nimg*H*W][ldx] fp32 carrier (in_f32) or 16-bit
    const float* ab; long ld_ab;           // GroupNorm (scale, shift) pairs [nimg][ld_ab][2]
    const void* Wt;                        // [Cout][9*Cin] 16-bit, k order (64-channel chunk, tap, channel) (packing.pack_conv3x3)
    const float* bias;                     // [Cout] or null
    float* out; long ldo;                  // [nimg*H*W][ldo] fp32
    int nimg, H, W, Cin, Cout;
};
bool vf_out_conv_supported(int Cin, int Cout);
int vf_launch_out_conv(const OutConvParams& p, int dtype, hipStream_t stream);

struct AttnParams {
    const void* Q; const void* K; const void* V;  // [B][n][ld*], head h at column h*dh
    long ldq, ldk, ldv;           // row (token) strides in elements
    long bsq, bsk, bsv;           // batch (sample) strides in elements
    const int* qk_map;            // [B] source sample of q,k for output sample b, or null (identity)
    const int* v_map;             // [B] source sample of v, or null
    void* O; long ldo, bso;       // [B][n][ldo]
    int B, heads, n, nk, dh;      // nk = number of keys (== n for self-attention)
    float scale;
    int variant;  // 0 = automatic; queries-per-wave variants for A/B benchmarking
    unsigned k_bytes, v_bytes;    // filled by the launcher: extents of the K / V views (buffer descriptors, < 4 GiB)
    int v_sets, set_stride;  // > 1: B q/k samples; output (and value) sample of set g is b + g*set_stride, scores shared
    int v_sets_live;         // 0 = all; 2 of v_sets == 3: only sets 0, 1 are read / written, with the arithmetic of the full call
};
int vf_launch_attention(const AttnParams& p, int dtype, hipStream_t stream);

// in_f32: x is the fp32 residual-stream copy (ldx in floats)
int vf_launch_layernorm(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy, int M,
                        int C, float eps, int in_f32, int dtype, hipStream_t stream);
int vf_launch_gn_finalize_cols(const float* colstats, long ld, int nimg, int hw, int C, int groups, float eps, float* stats,
                               hipStream_t stream);
int vf_launch_gn_coeffs_cols(const float* colstats, long ld, int nimg, int hw, int C, int groups, float eps, const float* gamma,
                             const float* beta, float* ab, hipStream_t stream);
int vf_launch_gn_stats(const void* x, long ldx, int nimg, int hw, int C, int groups, float eps, float* partial,
                       float* stats, int in_f32, int dtype, hipStream_t stream);
int vf_gn_partial_floats(int nimg, int hw, int C, int groups);
int vf_launch_gn_apply(const void* x, long ldx, const float* stats, const float* gamma, const float* beta, void* y,
                       long ldy, int nimg, int hw, int C, int groups, int silu, int in_f32, int dtype, hipStream_t stream);
int vf_launch_flow_warp(const void* src, long ld_src, long fs_src, const void* prev, long ld_prev,
                        const float* flow, const float* flow_prev, void* dst, long ld_dst, long fs_dst, int F, int h,
                        int w, int C, float alpha, float one_minus_alpha, int flags, int* dbg_x0, int* dbg_y0, int dtype,
                        hipStream_t stream);
int vf_launch_flow_to_latent(const float* flow_px, float* out, int pairs, int H, int W, int factor, hipStream_t stream);
// raft.hip: glue kernels of the RAFT-shaped flow producer (temporal_flow.py:27-38, 163-188)
int vf_launch_im2col(const void* x, long ldx, int nimg, int H, int W, int C, int KH, int KW, int stride, int pad_y, int pad_x, void* out,
                     long ldo, int dtype, hipStream_t stream);
int vf_chan_stats_slices(int hw);
int vf_launch_chan_stats(const void* x, long ldx, int nimg, int hw, int C, float eps, float* partial, float* stats, int dtype,
                         hipStream_t stream);
int vf_launch_chan_norm_act(const void* x, long ldx, const float* stats, const void* res, long ldr, void* y, long ldy, float* y32,
                            long ldy32, long M, int hw, int C, int act, int dtype, hipStream_t stream);
int vf_launch_gru_gate(const void* zr, long ldzr, const float* h32, void* z, long ldz, void* rh, long ldrh, long M, int Hd, int dtype,
                       hipStream_t stream);
int vf_launch_gru_update(const void* q, long ldq, const void* z, long ldz, float* h32, void* h16a, long lda, void* h16b, long ldb, long M,
                         int Hd, int dtype, hipStream_t stream);
int vf_launch_avgpool2_f32(const float* x, float* y, long R, int h, int w, hipStream_t stream);
int vf_launch_corr_lookup(const float* const* vols, const int* hs, const int* ws, int levels, const float* flow32, int h, int w, float scale,
                          void* out, long ldo, long M, int dtype, hipStream_t stream);
int vf_launch_flow_update(float* flow32, const float* delta32, long ldd, void* a, long lda, void* b, long ldb, void* c, long ldc, long M,
                          int dtype, hipStream_t stream);
int vf_launch_convex_upsample(const float* mask32, long ldm, const float* flow32, float* out, int B, int h, int w, float mult,
                              hipStream_t stream);
// paste.hip: per-frame paste-back (VFace_inference_batch.py:597-636); in_kind 0 fp16 | 1 bf16 | 2 fp32
int vf_launch_frame_to_u8(const void* x, unsigned char* out, int frames, int H, int W, int in_kind, hipStream_t stream);
int vf_launch_resample_u8(const unsigned char* src, unsigned char* dst, int frames, int in_n, int out_n, int lines, int axis,
                          const int* bounds, const int* kk, int ksize, hipStream_t stream);
int vf_launch_perspective_paste(const unsigned char* crop, int sw, int sh, unsigned char* frame, int W, int H, int frames,
                                const double* coeffs_dev, const double* coeffs_host, hipStream_t stream);
int vf_launch_frame_normalise_resize(const unsigned char* frame, int W, int H, float* out, int OW, int OH, int frames,
                                     hipStream_t stream);
int vf_launch_timestep_embedding(const long long* t, void* out, int N, int dim, int dtype, hipStream_t stream);
int vf_launch_silu(const void* x, void* y, long count, int in_f32, int dtype, hipStream_t stream);
int vf_launch_softmax_rows(const float* S, long lds_, void* P, long ldp, int M, int N, float scale, int dtype, hipStream_t stream);
int vf_launch_vae_sample(const float* moments, long ldm, const float* noise, float* z, int F, int hw, int zc, float scale,
                         hipStream_t stream);
int vf_launch_pack_input(const float* x, const float* inv, const float* inpaint, const float* mask, void* out,
                         int F, int h, int w, int cpad, int dtype, hipStream_t stream);
int vf_launch_nchw_to_nhwc(const float* x, void* out, int N, int C, int hw, int cpad, int dtype, hipStream_t stream);
int vf_launch_nhwc_to_nchw_f32(const float* x, long ldx, float* out, int N, int C, int hw, hipStream_t stream);
int vf_launch_ddim_step(const float* eps, long lde, const float* x, const float* inv, float* x_prev, float* pred_x0,
                        float* x_prev_recon, int F, int C, int hw, float scale, float a_t, float a_prev, float sigma_t,
                        float sqrt_1m_at, const float* noise, int single, hipStream_t stream);
int vf_launch_copy2d(const void* src, long lds, void* dst, long ldd, long rows, int cols, int dtype,
                     hipStream_t stream);
int vf_launch_cast(const float* src, void* dst, long count, int dtype, hipStream_t stream);
int vf_launch_temporal_gauss(const void* src, long ld_src, long fs_src, void* dst1, void* dst2, long ld_dst, long fs_dst,
                             int F, int n, int C, int dtype, hipStream_t stream);
size_t vf_adain_workspace_bytes(long rows, int C);
int vf_launch_adain(const void* a, long lda, const void* b, long ldb, void* dst, long ldd, long rows, int C, void* ws,
                    int dtype, hipStream_t stream);
