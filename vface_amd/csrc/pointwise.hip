// Bandwidth-bound kernels of the VFace UNet path (gfx950): normalisations, the flow-guided warp of Q/K
// maps, layout packing and the DDIM update.  All loads/stores of 16-bit data are 16 B per lane.
#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

// 8 consecutive channels as fp32: from a 16-bit tensor, or (IN32) from the fp32 residual-stream copy of it
// (`off` counts elements of the tensor's own type)
template <class TT, bool IN32>
__device__ __forceinline__ void load8(const void* base, long off, float (&f)[8]) {
    if constexpr (IN32) {
        const float4 a = *reinterpret_cast<const float4*>(static_cast<const float*>(base) + off);
        const float4 b = *reinterpret_cast<const float4*>(static_cast<const float*>(base) + off + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    } else {
        const typename TT::v8 t = *reinterpret_cast<const typename TT::v8*>(static_cast<const typename TT::elem*>(base) + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = to_f32(t[j]);
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm
// attention.py:231-233 (nn.LayerNorm, eps 1e-5).  fp32 statistics (autocast keeps layer_norm in fp32);
// the result is written in the 16-bit type because its only consumer is a 16-bit GEMM.
// One wave per row: the row (<= 2048 channels) lives in registers between the two passes.
template <class TT, int CH8, bool IN32>  // CH8: 8-channel chunks per lane (row length <= 512*CH8); IN32: x is fp32
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ x, long ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        typename TT::elem* __restrict__ y, long ldy, int M, int C,
                                                        float eps) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[CH8][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH8; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
            load8<TT, IN32>(x, (long)row * ldx + c, v[i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CH8; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    E* yr = y + (long)row * ldy;
#pragma unroll
    for (int i = 0; i < CH8; ++i) {
        const int c = (i * 64 + lane) * 8;
        if (c < C) {
            V8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = from_f32<E>((v[i][j] - mean) * rstd * gamma[c + j] + beta[c + j]);
            *reinterpret_cast<V8*>(yr + c) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------ GroupNorm
// util.py:214-216 (GroupNorm32, fp32 math) and attention.py:76-77 (eps 1e-6).  NHWC activations: a
// group is a slab of C/groups adjacent channels of every pixel.  Pass 1: each workgroup reduces a
// range of pixels over ALL channels (coalesced rows) into per-group (sum, sumsq) partials; pass 2
// folds the partials in double precision into (mean, rstd).
constexpr int GN_PIX = 128;  // pixels per workgroup

// Thread t owns one 16-B channel chunk (t % CPB) and one pixel lane (t / CPB): its 8 channels' (sum, sumsq) stay in
// registers across the pixel loop; the pixel lanes are then folded through LDS in a FIXED order (no atomics), so the
// statistics -- and everything downstream -- are bitwise reproducible.
template <class TT, bool IN32>
__global__ __launch_bounds__(256) void gn_partial_kernel(const void* __restrict__ x, long ldx, int hw,
                                                         int C, int groups, float* __restrict__ partial) {
    extern __shared__ float gn_lds[];  // [PP][CPB*8][2] lane partials, then [C][2] channel sums
    const int img = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
    const int cpg = C / groups, c8 = C / 8;
    const int p0 = chunk * GN_PIX, p1 = min(hw, p0 + GN_PIX);
    const long base = (long)img * hw * ldx;
    const int CPB = min(c8, 256), PP = 256 / CPB;
    const int lanep = t / CPB, cl = t - lanep * CPB;
    float* lane_part = gn_lds;                      // PP * CPB * 16 floats
    float* chan = gn_lds + PP * CPB * 16;           // C * 2 floats
    for (int cb = 0; cb < c8; cb += CPB) {
        const int ch = cb + cl;
        float s[8], q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
        if (lanep < PP && ch < c8) {
            for (int p = p0 + lanep; p < p1; p += PP) {
                float f[8];
                load8<TT, IN32>(x, base + (long)p * ldx + ch * 8, f);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
            }
        }
        if (lanep < PP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lane_part[((lanep * CPB + cl) * 8 + j) * 2] = s[j];
                lane_part[((lanep * CPB + cl) * 8 + j) * 2 + 1] = q[j];
            }
        }
        __syncthreads();
        // channel totals of this pass: one thread per channel, pixel lanes in ascending order
        for (int i = t; i < CPB * 8; i += 256) {
            const int c = cb * 8 + i;
            if (c < C) {
                float ss = 0.f, qq = 0.f;
                for (int l = 0; l < PP; ++l) { ss += lane_part[(l * CPB * 8 + i) * 2]; qq += lane_part[(l * CPB * 8 + i) * 2 + 1]; }
                chan[2 * c] = ss; chan[2 * c + 1] = qq;
            }
        }
        __syncthreads();
    }
    float* out = partial + ((long)img * gridDim.x + chunk) * 2 * groups;
    for (int g = t; g < groups; g += 256) {
        float s = 0.f, q = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s += chan[2 * c]; q += chan[2 * c + 1]; }
        out[2 * g] = s; out[2 * g + 1] = q;
    }
}

__global__ void gn_finalize_kernel(const float* __restrict__ partial, int nchunks, int groups, double count,
                                   float eps, float* __restrict__ stats) {
    const int img = blockIdx.x;
    for (int g = threadIdx.x; g < groups; g += blockDim.x) {
        double s = 0.0, q = 0.0;
        for (int c = 0; c < nchunks; ++c) {
            const float* pp = partial + ((long)img * nchunks + c) * 2 * groups;
            s += pp[2 * g]; q += pp[2 * g + 1];
        }
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[((long)img * groups + g) * 2] = (float)mean;
        stats[((long)img * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// y = silu?( x * a[c] + b[c] ) with a = rstd * gamma, b = beta - mean * rstd * gamma held in registers per thread
// (mean, rstd) from producer-side column statistics: colstats[slice][ld][2] with 64-row slices, hw % 64 == 0
__global__ __launch_bounds__(64) void gn_finalize_cols_kernel(const float* __restrict__ colstats, long ld, int hw, int C,
                                                              int groups, float eps, float* __restrict__ stats) {
    const int img = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
    const int cpg = C / groups, spi = hw / 64;
    const int total = cpg * spi;
    double s = 0.0, q = 0.0;
    for (int i = lane; i < total; i += 64) {
        const int sl = i / cpg, c = g * cpg + (i - sl * cpg);
        const float2 v = *reinterpret_cast<const float2*>(colstats + (((long)img * spi + sl) * ld + c) * 2);
        s += v.x; q += v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (lane == 0) {
        const double count = (double)hw * cpg;
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[((long)img * groups + g) * 2] = (float)mean;
        stats[((long)img * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// gn_finalize_cols + the per-channel scale / shift of gn_apply in one launch: ab[img][c] = (rstd * gamma[c], beta[c] - mean * a) --
// bit for bit the a[j], b[j] gn_apply_kernel forms, so a convolution that applies them to its operand in LDS (conv.hip) sees
// exactly what the separate normalisation pass would have stored.
__global__ __launch_bounds__(64) void gn_coeffs_cols_kernel(const float* __restrict__ colstats, long ld, int hw, int C, int groups,
                                                            float eps, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ ab) {
    const int img = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
    const int cpg = C / groups, spi = hw / 64;
    const int total = cpg * spi;
    double s = 0.0, q = 0.0;
    for (int i = lane; i < total; i += 64) {
        const int sl = i / cpg, c = g * cpg + (i - sl * cpg);
        const float2 v = *reinterpret_cast<const float2*>(colstats + (((long)img * spi + sl) * ld + c) * 2);
        s += v.x; q += v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    const double count = (double)hw * cpg;
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float meanf = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int c = g * cpg + lane; c < (g + 1) * cpg; c += 64) {
        const float a = rstd * gamma[c];
        *reinterpret_cast<float2*>(ab + ((long)img * C + c) * 2) = make_float2(a, beta[c] - meanf * a);
    }
}

// DEEP: four pixels per trip with their loads issued together (the small maps); !DEEP: one pixel per trip (the large maps: enough
// workgroups in flight to hide a single load per trip) -- its own instantiation since round 6: 40 instead of 96 registers per lane, so
// that a wave of it fits beside the matrix kernels of another launch stream (gemm256 leaves 48 free, eight-wave attention 80)
template <class TT, bool IN32, bool DEEP>
__global__ __launch_bounds__(256) void gn_apply_kernel(const void* __restrict__ x, long ldx,
                                                       const float* __restrict__ stats,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       typename TT::elem* __restrict__ y, long ldy, int hw, int C,
                                                       int groups, int silu, int pix_per_block) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    const int img = blockIdx.y, t = threadIdx.x;
    const int c8 = C / 8, cpg = C / groups;
    const int p0 = blockIdx.x * pix_per_block, p1 = min(hw, p0 + pix_per_block);
    const float* st = stats + (long)img * groups * 2;
    const long xb = (long)img * hw * ldx;
    E* yb = y + (long)img * hw * ldy;
    const int CPB = min(c8, 256), PP = 256 / CPB;
    const int lanep = t / CPB, cl = t - lanep * CPB;
    for (int cb = 0; cb < c8; cb += CPB) {
        const int ch = cb + cl;
        if (lanep >= PP || ch >= c8) continue;
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ch * 8 + j, g = c / cpg;
            a[j] = st[2 * g + 1] * gamma[c];
            b[j] = beta[c] - st[2 * g] * a[j];
        }
        // four pixels per trip, their loads issued together: on the small maps (8x8, 16x16 images) a thread's pixel loop was a
        // chain of dependent load -> store round trips (17-22 us per launch for ~10 MB)
        if constexpr (!DEEP) {
            for (int p = p0 + lanep; p < p1; p += PP) {
                float v[8];
                load8<TT, IN32>(x, xb + (long)p * ldx + ch * 8, v);
                V8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float f = v[j] * a[j] + b[j];
                    if (silu) f = silu_f(f);
                    o[j] = from_f32<E>(f);
                }
                *reinterpret_cast<V8*>(yb + (long)p * ldy + ch * 8) = o;
            }
            continue;
        }
        for (int p = p0 + lanep; p < p1; p += 4 * PP) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (p + u * PP < p1) load8<TT, IN32>(x, xb + (long)(p + u * PP) * ldx + ch * 8, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p + u * PP >= p1) break;
                V8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float f = v[u][j] * a[j] + b[j];
                    if (silu) f = silu_f(f);
                    o[j] = from_f32<E>(f);
                }
                *reinterpret_cast<V8*>(yb + (long)(p + u * PP) * ldy + ch * 8) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ flow warp
// temporal_flow.py:40-53 (warp_image) + :222-237 (align_by_flow) on token-major maps [F][h*w][C]:
//   out[f] = alpha * src[f] + (1 - alpha) * bilinear(src[f-1], (x + dx, y + dy)),  out[0] = src[0]
// reading the UNMODIFIED source (not recurrent).  The sampling coordinate follows the reference's fp32
// operation order exactly -- x + dx; 2.0 * v / max(W-1, 1) - 1.0; ATen's ((g + 1) / 2) * (W - 1);
// clamp to the border; floor -- with no FMA contraction, because the normalise / un-normalise round
// trip moves integer coordinates by an ulp and decides floor().  flags bit 0 selects CUDA-ATen's
// "multiply by the reciprocal" form of the scalar division (what the reference computes on its
// native CUDA device); the default true division is what it computes on CPU, where the oracle and
// the golden vectors were produced.
// `prev` (optional) is the frame preceding src[0] -- the last frame of the previous rank's shard when
// frames are sharded across GPUs -- with its own flow field `flow_prev`.
__device__ __forceinline__ float unnorm_coord(float pos, float d, int size, bool recip) {
    const float v = __fadd_rn(pos, d);
    const float den = (float)max(size - 1, 1);
    const float two_v = __fmul_rn(2.0f, v);
    const float qn = recip ? __fmul_rn(two_v, __fdiv_rn(1.0f, den)) : __fdiv_rn(two_v, den);
    const float g = __fsub_rn(qn, 1.0f);
    float c = __fmul_rn(__fmul_rn(__fadd_rn(g, 1.0f), 0.5f), (float)(size - 1));
    c = fminf((float)(size - 1), fmaxf(c, 0.0f));
    return c;
}

template <class TT>
__global__ __launch_bounds__(256) void flow_warp_kernel(const typename TT::elem* __restrict__ src, long ld_src,
                                                        long fs_src, const typename TT::elem* __restrict__ prev,
                                                        long ld_prev, const float* __restrict__ flow,
                                                        const float* __restrict__ flow_prev,
                                                        typename TT::elem* __restrict__ dst, long ld_dst,
                                                        long fs_dst, int F, int h, int w, int C, float alpha,
                                                        float oma, int flags, int* __restrict__ dbg_x0,
                                                        int* __restrict__ dbg_y0) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    const int c8 = C / 8;
    const int f = blockIdx.y;
    const long total = (long)h * w * c8;
    const bool recip = flags & 1;
    const E* cur = src + (long)f * fs_src;
    E* out = dst + (long)f * fs_dst;
    const E* from = nullptr;
    long ld_from = ld_src;
    const float* fl = nullptr;
    if (f > 0) { from = src + (long)(f - 1) * fs_src; fl = flow + (long)(f - 1) * 2 * h * w; }
    else if (prev) { from = prev; ld_from = ld_prev; fl = flow_prev; }
    // The four bilinear taps are read through a buffer descriptor over the source frame (32-bit offsets, one wave-uniform
    // base; the launcher checks the frame view is < 4 GiB): a tap's address is one 32-bit multiply-add instead of a 64-bit
    // chain, and the load's destination registers never alias an address pair.
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    const unsigned from_bytes = from ? (unsigned)((((long)h * w - 1) * ld_from + C) * sizeof(E)) : 0u;
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(const_cast<E*>(from ? from : cur), 0, (int)from_bytes, 0x00020000);
    const unsigned row_b = (unsigned)(ld_from * sizeof(E));
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int pix = (int)(i / c8), cc = (int)(i - (long)pix * c8) * 8;
        const V8 xv = *reinterpret_cast<const V8*>(cur + (long)pix * ld_src + cc);
        if (!from) { *reinterpret_cast<V8*>(out + (long)pix * ld_dst + cc) = xv; continue; }
        const int py = pix / w, px = pix - py * w;
        const float ix = unnorm_coord((float)px, fl[pix], w, recip);
        const float iy = unnorm_coord((float)py, fl[h * w + pix], h, recip);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        if (cc == 0 && f > 0 && dbg_x0) {
            dbg_x0[(long)(f - 1) * h * w + pix] = x0;
            dbg_y0[(long)(f - 1) * h * w + pix] = y0;
        }
        const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        // The coordinate was clamped to [0, size - 1] (border padding), so a neighbour past the border (x0 + 1 > w - 1) only ever
        // occurs with ix == w - 1 exactly, i.e. wx1 == +0: its weight is zero by arithmetic and needs no select -- the neighbour's
        // index is clamped with a min instead.  (The selects this replaces -- `vx ? wx1 * wy0 : 0` on a VCC lane mask, and x0 + vx
        // through an add-with-carry on the same mask -- are what returned the wrong branch for lanes 48-63 while attention waves of
        // another queue shared the SIMD: tools/warp_coresidency_probe.py, HISTORY R5.)  Same bits: 0 * w = +0 either way.
        const int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
        const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
        const unsigned cb = (unsigned)cc * (unsigned)sizeof(E);
        const V8 a = __builtin_bit_cast(V8, (u4_t)__builtin_amdgcn_raw_buffer_load_b128(rF, (unsigned)(y0 * w + x0) * row_b + cb, 0, 0));
        const V8 b = __builtin_bit_cast(V8, (u4_t)__builtin_amdgcn_raw_buffer_load_b128(rF, (unsigned)(y0 * w + x1) * row_b + cb, 0, 0));
        const V8 c = __builtin_bit_cast(V8, (u4_t)__builtin_amdgcn_raw_buffer_load_b128(rF, (unsigned)(y1 * w + x0) * row_b + cb, 0, 0));
        const V8 d = __builtin_bit_cast(V8, (u4_t)__builtin_amdgcn_raw_buffer_load_b128(rF, (unsigned)(y1 * w + x1) * row_b + cb, 0, 0));
        V8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float wv = to_f32(a[j]) * w00;
            wv += to_f32(b[j]) * w01;
            wv += to_f32(c[j]) * w10;
            wv += to_f32(d[j]) * w11;
            // alpha * x keeps the 16-bit type in the reference (python scalar * half tensor), then the
            // fp32 sum is rounded on store (temporal_flow.py:234-235)
            const float ax = to_f32(from_f32<E>(alpha * to_f32(xv[j])));
            o[j] = from_f32<E>(ax + oma * wv);
        }
        *reinterpret_cast<V8*>(out + (long)pix * ld_dst + cc) = o;
    }
}

// Pixel-resolution optical flow -> latent-resolution flow (the resample temporal_flow.py:163-188's output needs
// before it can meet a 64 x 64 attention map): area mean over each f x f block, divided by f.
// flow_px [P][2][H][W] fp32 -> out [P][2][H/f][W/f] fp32.  One thread per output element, rows of the block read with
// consecutive threads on consecutive blocks (f floats apart: every byte of a row is still used by the wave).
__global__ __launch_bounds__(256) void flow_to_latent_kernel(const float* __restrict__ src, float* __restrict__ dst, long planes,
                                                             int H, int W, int f) {
    const int h = H / f, w = W / f;
    const long total = planes * h * w;
    const float inv = 1.0f / ((float)f * (float)f * (float)f);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const long pl = i / ((long)w * h);
        const float* b = src + (pl * H + (long)y * f) * W + (long)x * f;
        float acc = 0.f;
        for (int r = 0; r < f; ++r)
            for (int c = 0; c < f; ++c) acc += b[(long)r * W + c];
        dst[i] = acc * inv;
    }
}

// ------------------------------------------------------------------------------------------ inactive hook modes
// fusion="temporal" (pnp_utils.py:59-90,145-154): Gaussian-weighted mean over the FRAME axis of chunk 0's q|k
// (window 5, sigma 1, renormalised at the clip ends), written to chunk 1 and chunk 2.  fp32 math, one rounding.
template <class TT>
__global__ __launch_bounds__(256) void temporal_gauss_kernel(const typename TT::elem* __restrict__ src, long ld_src,
                                                             long fs_src, typename TT::elem* __restrict__ dst1,
                                                             typename TT::elem* __restrict__ dst2, long ld_dst,
                                                             long fs_dst, int F, int n, int C) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    const float g1 = 0.60653065971263342f, g2 = 0.13533528323661270f;  // exp(-0.5), exp(-2)
    const float gs = 1.0f + 2.0f * g1 + 2.0f * g2;
    const float w[5] = {g2 / gs, g1 / gs, 1.0f / gs, g1 / gs, g2 / gs};
    const int c8 = C / 8, f = blockIdx.y;
    const long total = (long)n * c8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long tok = i / c8;
        const int cc = (int)(i - tok * c8) * 8;
        float acc[8], wt = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int o = -2; o <= 2; ++o) {
            const int ff = f + o;
            if (ff < 0 || ff >= F) continue;
            const V8 v = *reinterpret_cast<const V8*>(src + (long)ff * fs_src + tok * ld_src + cc);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += w[o + 2] * to_f32(v[j]);
            wt += w[o + 2];
        }
        V8 o8;
#pragma unroll
        for (int j = 0; j < 8; ++j) o8[j] = from_f32<E>(acc[j] / wt);
        *reinterpret_cast<V8*>(dst1 + (long)f * fs_dst + tok * ld_dst + cc) = o8;
        if (dst2) *reinterpret_cast<V8*>(dst2 + (long)f * fs_dst + tok * ld_dst + cc) = o8;
    }
}

// fusion="adaIn" (face_swap_utils.py:372-389 with normalized=True): per token, AdaIN of the structure row `a`
// to the own row `b` over the channel axis (unbiased std), then the whole tensor is divided by its GLOBAL
// unbiased std.  Pass 1: one wave per token row writes the fused row (fp32) and block partial (sum, sumsq);
// pass 2 (adain_scale) folds the partials in fp64 and writes fused / (std + 1e-5) in the 16-bit type.
template <class TT, int CH8>
__global__ __launch_bounds__(256) void adain_rows_kernel(const typename TT::elem* __restrict__ a, long lda,
                                                         const typename TT::elem* __restrict__ b, long ldb,
                                                         float* __restrict__ fused, long ldf, int rows, int C,
                                                         double* __restrict__ partial) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    __shared__ double red[8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wv;
    double bs = 0.0, bq = 0.0;
    if (row < rows) {
        float va[CH8][8], vb[CH8][8];
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int i = 0; i < CH8; ++i) {
            const int c = (i * 64 + lane) * 8;
            if (c < C) {
                const V8 ta = *reinterpret_cast<const V8*>(a + (long)row * lda + c);
                const V8 tb = *reinterpret_cast<const V8*>(b + (long)row * ldb + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) { va[i][j] = to_f32(ta[j]); vb[i][j] = to_f32(tb[j]); sa += va[i][j]; sb += vb[i][j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) va[i][j] = vb[i][j] = 0.f;
            }
        }
        const float ma = wave_sum(sa) / (float)C, mb = wave_sum(sb) / (float)C;
        float qa = 0.f, qb = 0.f;
#pragma unroll
        for (int i = 0; i < CH8; ++i) {
            const int c = (i * 64 + lane) * 8;
            if (c < C) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float da = va[i][j] - ma, db = vb[i][j] - mb;
                    qa += da * da; qb += db * db;
                }
            }
        }
        const float sda = sqrtf(wave_sum(qa) / (float)(C - 1)), sdb = sqrtf(wave_sum(qb) / (float)(C - 1));
        const float k = sdb / (sda + 1e-5f);
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < CH8; ++i) {
            const int c = (i * 64 + lane) * 8;
            if (c < C) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { o[j] = (va[i][j] - ma) * k + mb; s += o[j]; q += o[j] * o[j]; }
                *reinterpret_cast<float4*>(fused + (long)row * ldf + c) = make_float4(o[0], o[1], o[2], o[3]);
                *reinterpret_cast<float4*>(fused + (long)row * ldf + c + 4) = make_float4(o[4], o[5], o[6], o[7]);
            }
        }
        bs = (double)wave_sum(s); bq = (double)wave_sum(q);
    }
    if (lane == 0) { red[2 * wv] = bs; red[2 * wv + 1] = bq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = red[0] + red[2] + red[4] + red[6];
        partial[2 * blockIdx.x + 1] = red[1] + red[3] + red[5] + red[7];
    }
}

__global__ void adain_reduce_kernel(const double* __restrict__ partial, int nblocks, double count, float* __restrict__ inv) {
    __shared__ double ss[256], qq[256];
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) { s += partial[2 * i]; q += partial[2 * i + 1]; }
    ss[threadIdx.x] = s; qq[threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { ss[threadIdx.x] += ss[threadIdx.x + o]; qq[threadIdx.x] += qq[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = ss[0] / count;
        double var = (qq[0] - count * mean * mean) / (count - 1.0);  // unbiased, as torch.std
        if (var < 0.0) var = 0.0;
        inv[0] = (float)(1.0 / (sqrt(var) + 1e-5));
    }
}

template <class TT>
__global__ void adain_scale_kernel(const float* __restrict__ fused, long ldf, const float* __restrict__ inv,
                                   typename TT::elem* __restrict__ dst, long ldd, long rows, int C) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    const int c8 = C / 8;
    const float k = inv[0];
    const long total = rows * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c8;
        const int c = (int)(i - r * c8) * 8;
        const float4 x0 = *reinterpret_cast<const float4*>(fused + r * ldf + c);
        const float4 x1 = *reinterpret_cast<const float4*>(fused + r * ldf + c + 4);
        V8 o;
        o[0] = from_f32<E>(x0.x * k); o[1] = from_f32<E>(x0.y * k); o[2] = from_f32<E>(x0.z * k); o[3] = from_f32<E>(x0.w * k);
        o[4] = from_f32<E>(x1.x * k); o[5] = from_f32<E>(x1.y * k); o[6] = from_f32<E>(x1.z * k); o[7] = from_f32<E>(x1.w * k);
        *reinterpret_cast<V8*>(dst + r * ldd + c) = o;
    }
}

// ------------------------------------------------------------------------------------------ small ops
// util.py:151-171: [cos(t * f_i) | sin(t * f_i)], f_i = exp(-ln(10000) * i / half)
template <class TT>
__global__ void timestep_embedding_kernel(const long long* __restrict__ t, typename TT::elem* __restrict__ out,
                                          int N, int dim) {
    using E = typename TT::elem;
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * half) return;
    const int n = i / half, k = i - n * half;
    const float freq = expf(-9.210340371976184f * (float)k / (float)half);
    const float arg = (float)t[n] * freq;
    out[(long)n * dim + k] = from_f32<E>(cosf(arg));
    out[(long)n * dim + half + k] = from_f32<E>(sinf(arg));
    if ((dim & 1) && k == 0) out[(long)n * dim + dim - 1] = from_f32<E>(0.f);
}

template <class TT, bool IN_F32>
__global__ void silu_kernel(const void* __restrict__ x, typename TT::elem* __restrict__ y, long count) {
    using E = typename TT::elem;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float f = IN_F32 ? reinterpret_cast<const float*>(x)[i] : to_f32(reinterpret_cast<const E*>(x)[i]);
        y[i] = from_f32<E>(silu_f(f));
    }
}

template <class TT>
__global__ void cast_kernel(const float* __restrict__ x, typename TT::elem* __restrict__ y, long count) {
    using E = typename TT::elem;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        y[i] = from_f32<E>(x[i]);
}

// ddim_w_inv.py:633,654-655: x9 = cat[x, inpaint, mask]; x_in = cat[x9, x9, cat[inv_t, inpaint, mask]]
// written straight into the NHWC 16-bit layout the first convolution reads, channels padded to cpad.
template <class TT>
__global__ void pack_input_kernel(const float* __restrict__ x, const float* __restrict__ inv,
                                  const float* __restrict__ inpaint, const float* __restrict__ mask,
                                  typename TT::elem* __restrict__ out, int F, int hw, int cpad) {
    using E = typename TT::elem;
    const long total = (long)(inv ? 3 : 2) * F * hw * cpad;      // inv == nullptr: [uncond ; cond] only (no recon chunk)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpad);
        const long pi = i / cpad;
        const int pix = (int)(pi % hw);
        const int s = (int)(pi / hw);
        const int chunk = s / F, f = s - chunk * F;
        float v = 0.f;
        if (c < 4) v = (chunk == 2 ? inv : x)[((long)f * 4 + c) * hw + pix];
        else if (c < 8) v = inpaint[((long)f * 4 + (c - 4)) * hw + pix];
        else if (c == 8) v = mask[(long)f * hw + pix];
        out[i] = from_f32<E>(v);
    }
}

template <class TT>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, typename TT::elem* __restrict__ out, int N, int C,
                                    int hw, int cpad) {
    using E = typename TT::elem;
    const long total = (long)N * hw * cpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpad);
        const long pi = i / cpad;
        const int pix = (int)(pi % hw);
        const int n = (int)(pi / hw);
        out[i] = from_f32<E>(c < C ? x[((long)n * C + c) * hw + pix] : 0.f);
    }
}

__global__ void nhwc_to_nchw_f32_kernel(const float* __restrict__ x, long ldx, float* __restrict__ out, int N, int C,
                                        int hw) {
    const long total = (long)N * C * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % hw);
        const long nc = i / hw;
        const int c = (int)(nc % C);
        const int n = (int)(nc / C);
        out[i] = x[((long)n * hw + pix) * ldx + c];
    }
}

// ddim_w_inv.py:666-667 (guidance; the recon branch formula as written) and :686-700 (x0 prediction,
// direction, x_{t-1}); eps is the UNet output for [uncond ; cond ; recon], NHWC fp32 with row stride lde.
__global__ void ddim_step_kernel(const float* __restrict__ eps, long lde, const float* __restrict__ x,
                                 const float* __restrict__ inv, float* __restrict__ x_prev,
                                 float* __restrict__ pred_x0, float* __restrict__ x_prev_recon, int F, int C, int hw,
                                 float scale, float a_t, float a_prev, float sigma_t, float sqrt_1m_at,
                                 const float* __restrict__ noise, int single) {
    const long total = (long)F * C * hw;
    const float sqrt_at = sqrtf(a_t), sqrt_ap = sqrtf(a_prev);
    const float dir = sqrtf(1.0f - a_prev - sigma_t * sigma_t);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % hw);
        const long fc = i / hw;
        const int c = (int)(fc % C);
        const int f = (int)(fc / C);
        const float eu = eps[((long)f * hw + pix) * lde + c];
        // single == 1: eps holds one branch only (DDIM inversion, no guidance, ddim_w_inv.py:426-427); single == 2: eps holds
        // [uncond ; cond] only -- the recon branch, whose x_prev the sampler drops (ddim_w_inv.py:703-707,738), was not computed
        const float ec = single == 1 ? eu : eps[((long)(F + f) * hw + pix) * lde + c];
        const float er = single ? eu : eps[((long)(2 * F + f) * hw + pix) * lde + c];
        const float e_t = single == 1 ? eu : eu + scale * (ec - eu);
        const float p0 = (x[i] - sqrt_1m_at * e_t) / sqrt_at;
        const float nz = noise ? sigma_t * noise[i] : 0.f;
        x_prev[i] = sqrt_ap * p0 + dir * e_t + nz;
        if (pred_x0) pred_x0[i] = p0;
        if (x_prev_recon && inv && !single) {
            const float e_r = er + scale * (er - eu);
            const float p0r = (inv[i] - sqrt_1m_at * e_r) / sqrt_at;
            x_prev_recon[i] = sqrt_ap * p0r + dir * e_r;
        }
    }
}

template <class TT>
__global__ void copy2d_kernel(const typename TT::elem* __restrict__ src, long lds_, typename TT::elem* __restrict__ dst,
                              long ldd, long rows, int cols) {
    using V8 = typename TT::v8;
    const int c8 = cols / 8;
    const long total = rows * c8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / c8;
        const int c = (int)(i - r * c8) * 8;
        *reinterpret_cast<V8*>(dst + r * ldd + c) = *reinterpret_cast<const V8*>(src + r * lds_ + c);
    }
}

inline int grid_for(long total, int block = 256, int cap = 8192) {
    long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
inline int ok() { return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH; }

}  // namespace


// P[m][:] = softmax(S[m][:] * scale) over N columns, fp32 in, 16-bit out: the VAE's single-head 512-channel AttnBlock
// (diffusionmodules/model.py:183-186), whose head dim is far beyond what the streaming attention kernel keeps in LDS, so
// its scores go through the GEMM kernel (fp32 out) and this pass.  One workgroup per row, row held in registers.
template <class TT>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, long lds_, typename TT::elem* __restrict__ P,
                                                           long ldp, int N, float scale) {
    using E = typename TT::elem;
    __shared__ float red[8];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* row = S + (long)blockIdx.x * lds_;
    E* out = P + (long)blockIdx.x * ldp;
    constexpr int MAXV = 16;                      // up to 256 * 16 * 4 = 16384 columns
    float4 v[MAXV];
    const int nv = (N / 4 + 255) / 256;
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (i < nv) {
            const int c4 = i * 256 + t;
            v[i] = c4 * 4 < N ? *reinterpret_cast<const float4*>(row + c4 * 4) : make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
            mx = fmaxf(mx, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float c = scale * 1.44269504088896340736f, mc = mx * c;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (i < nv) {
            v[i].x = __builtin_amdgcn_exp2f(fmaf(v[i].x, c, -mc)); v[i].y = __builtin_amdgcn_exp2f(fmaf(v[i].y, c, -mc));
            v[i].z = __builtin_amdgcn_exp2f(fmaf(v[i].z, c, -mc)); v[i].w = __builtin_amdgcn_exp2f(fmaf(v[i].w, c, -mc));
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        if (i < nv) {
            const int c4 = i * 256 + t;
            if (c4 * 4 < N) {
                typename TT::v4 o;
                o[0] = from_f32<E>(v[i].x * inv); o[1] = from_f32<E>(v[i].y * inv);
                o[2] = from_f32<E>(v[i].z * inv); o[3] = from_f32<E>(v[i].w * inv);
                *reinterpret_cast<typename TT::v4*>(out + c4 * 4) = o;
            }
        }
    }
}

// z = (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) * scale from the quant_conv moments [rows][2*zc] (NHWC fp32):
// DiagonalGaussianDistribution.sample (distributions.py:24-37) and get_first_stage_encoding's scale_factor; noise == null
// gives the mode.  Output NCHW fp32 [F][zc][hw], the layout the sampler keeps latents in.
__global__ void vae_sample_kernel(const float* __restrict__ moments, long ldm, const float* __restrict__ noise,
                                  float* __restrict__ z, int F, int hw, int zc, float scale) {
    const long total = (long)F * zc * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % hw);
        const int c = (int)((i / hw) % zc);
        const long f = i / ((long)hw * zc);
        const float* m = moments + (f * hw + p) * ldm;
        const float mean = m[c];
        float v = mean;
        if (noise) {
            const float lv = fminf(fmaxf(m[zc + c], -30.0f), 20.0f);
            v = mean + expf(0.5f * lv) * noise[i];
        }
        z[i] = v * scale;
    }
}

#define DISPATCH_DTYPE(dtype, CALL)            \
    if ((dtype) == VF_DTYPE_F16) { using TT = F16; CALL; } \
    else if ((dtype) == VF_DTYPE_BF16) { using TT = BF16; CALL; } \
    else return VF_ERR_DTYPE;

int vf_launch_layernorm(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy, int M,
                        int C, float eps, int in_f32, int dtype, hipStream_t stream) {
    if (!x || !gamma || !beta || !y || M <= 0 || C <= 0) return VF_ERR_ARG;
    if ((C & 7) || (ldx & (in_f32 ? 3 : 7)) || (ldy & 7) || (((uintptr_t)x | (uintptr_t)y) & 15)) return VF_ERR_ALIGN;
    if (C > 2048) return VF_ERR_SHAPE;
    const int ch8 = (C + 511) / 512;
    dim3 grid((M + 3) / 4);
#define LN_LAUNCH(CH, IN) hipLaunchKernelGGL((layernorm_kernel<TT, CH, IN>), grid, dim3(256), 0, stream, x, ldx, gamma, beta, yo, ldy, M, C, eps)
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        E* yo = (E*)y;
        switch (ch8) {
            case 1: if (in_f32) LN_LAUNCH(1, true); else LN_LAUNCH(1, false); break;
            case 2: if (in_f32) LN_LAUNCH(2, true); else LN_LAUNCH(2, false); break;
            case 3: if (in_f32) LN_LAUNCH(3, true); else LN_LAUNCH(3, false); break;
            default: if (in_f32) LN_LAUNCH(4, true); else LN_LAUNCH(4, false); break;
        }
    });
#undef LN_LAUNCH
    return ok();
}

int vf_gn_partial_floats(int nimg, int hw, int C, int groups) {
    (void)C;
    return nimg * ((hw + GN_PIX - 1) / GN_PIX) * 2 * groups;
}

int vf_launch_gn_stats(const void* x, long ldx, int nimg, int hw, int C, int groups, float eps, float* partial,
                       float* stats, int in_f32, int dtype, hipStream_t stream) {
    if (!x || !partial || !stats || nimg <= 0 || hw <= 0 || C <= 0 || groups <= 0) return VF_ERR_ARG;
    if ((C & 7) || (ldx & (in_f32 ? 3 : 7)) || ((uintptr_t)x & 15)) return VF_ERR_ALIGN;
    if (groups > 64 || C % groups) return VF_ERR_SHAPE;
    const int nchunks = (hw + GN_PIX - 1) / GN_PIX;
    dim3 grid(nchunks, nimg);
    DISPATCH_DTYPE(dtype, {
        if (in_f32) hipLaunchKernelGGL((gn_partial_kernel<TT, true>), grid, dim3(256), (size_t)(2 * C + 256 * 16) * sizeof(float), stream, x, ldx, hw, C, groups, partial);
        else hipLaunchKernelGGL((gn_partial_kernel<TT, false>), grid, dim3(256), (size_t)(2 * C + 256 * 16) * sizeof(float), stream, x, ldx, hw, C, groups, partial);
    });
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(nimg), dim3(64), 0, stream, (const float*)partial, nchunks, groups,
                       (double)hw * (C / groups), eps, stats);
    return ok();
}

int vf_launch_gn_coeffs_cols(const float* colstats, long ld, int nimg, int hw, int C, int groups, float eps, const float* gamma,
                             const float* beta, float* ab, hipStream_t stream) {
    if (!colstats || !gamma || !beta || !ab || nimg <= 0 || hw <= 0 || C <= 0 || groups <= 0) return VF_ERR_ARG;
    if ((hw & 63) || groups > 64 || C % groups || (ld & 1)) return VF_ERR_SHAPE;
    hipLaunchKernelGGL(gn_coeffs_cols_kernel, dim3(groups, nimg), dim3(64), 0, stream, colstats, ld, hw, C, groups, eps, gamma, beta, ab);
    return ok();
}

int vf_launch_gn_apply(const void* x, long ldx, const float* stats, const float* gamma, const float* beta, void* y,
                       long ldy, int nimg, int hw, int C, int groups, int silu, int in_f32, int dtype, hipStream_t stream) {
    if (!x || !stats || !gamma || !beta || !y || nimg <= 0 || hw <= 0) return VF_ERR_ARG;
    if ((C & 7) || (ldx & (in_f32 ? 3 : 7)) || (ldy & 7) || (((uintptr_t)x | (uintptr_t)y) & 15)) return VF_ERR_ALIGN;
    if (groups > 64 || C % groups) return VF_ERR_SHAPE;
    // enough workgroups to fill 256 CUs several times over, but long enough pixel loops to amortise the
    // per-thread scale/shift set-up
    int ppb = 128;
    while (ppb > 4 && (long)nimg * ((hw + ppb - 1) / ppb) < 1024) ppb >>= 1;
    dim3 grid((hw + ppb - 1) / ppb, nimg);
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        if (ppb > 32) {
            if (in_f32) hipLaunchKernelGGL((gn_apply_kernel<TT, true, false>), grid, dim3(256), 0, stream, x, ldx, stats, gamma, beta, (E*)y, ldy, hw, C, groups, silu, ppb);
            else hipLaunchKernelGGL((gn_apply_kernel<TT, false, false>), grid, dim3(256), 0, stream, x, ldx, stats, gamma, beta, (E*)y, ldy, hw, C, groups, silu, ppb);
        } else {
            if (in_f32) hipLaunchKernelGGL((gn_apply_kernel<TT, true, true>), grid, dim3(256), 0, stream, x, ldx, stats, gamma, beta, (E*)y, ldy, hw, C, groups, silu, ppb);
            else hipLaunchKernelGGL((gn_apply_kernel<TT, false, true>), grid, dim3(256), 0, stream, x, ldx, stats, gamma, beta, (E*)y, ldy, hw, C, groups, silu, ppb);
        }
    });
    return ok();
}

int vf_launch_flow_warp(const void* src, long ld_src, long fs_src, const void* prev, long ld_prev,
                        const float* flow, const float* flow_prev, void* dst, long ld_dst, long fs_dst, int F, int h,
                        int w, int C, float alpha, float one_minus_alpha, int flags, int* dbg_x0, int* dbg_y0,
                        int dtype, hipStream_t stream) {
    if (!src || !dst || F <= 0 || h <= 0 || w <= 0 || C <= 0) return VF_ERR_ARG;
    if (F > 1 && !flow) return VF_ERR_ARG;
    if (prev && !flow_prev) return VF_ERR_ARG;
    if ((C & 7) || (ld_src & 7) || (ld_dst & 7) || (fs_src & 7) || (fs_dst & 7) || (prev && (ld_prev & 7))) return VF_ERR_ALIGN;
    if (((uintptr_t)src | (uintptr_t)dst | (uintptr_t)prev) & 15) return VF_ERR_ALIGN;
    if ((dbg_x0 == nullptr) != (dbg_y0 == nullptr)) return VF_ERR_ARG;
    // one source frame is addressed through a buffer descriptor with 32-bit byte offsets
    if ((((long)h * w - 1) * (ld_src > ld_prev ? ld_src : ld_prev) + C) * 2 >= 0xFFFFFFF0l) return VF_ERR_SHAPE;
    const long total = (long)h * w * (C / 8);
    dim3 grid(grid_for(total, 256, 2048), F);
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((flow_warp_kernel<TT>), grid, dim3(256), 0, stream, (const E*)src, ld_src, fs_src,
                           (const E*)prev, ld_prev, flow, flow_prev, (E*)dst, ld_dst, fs_dst, F, h, w, C, alpha, one_minus_alpha, flags,
                           dbg_x0, dbg_y0);
    });
    return ok();
}

int vf_launch_flow_to_latent(const float* flow_px, float* out, int pairs, int H, int W, int factor, hipStream_t stream) {
    if (!flow_px || !out || pairs <= 0 || H <= 0 || W <= 0 || factor <= 0) return VF_ERR_ARG;
    if ((H % factor) || (W % factor)) return VF_ERR_SHAPE;
    const long total = (long)pairs * 2 * (H / factor) * (W / factor);
    hipLaunchKernelGGL(flow_to_latent_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, stream, flow_px, out, (long)pairs * 2,
                       H, W, factor);
    return ok();
}

int vf_launch_timestep_embedding(const long long* t, void* out, int N, int dim, int dtype, hipStream_t stream) {
    if (!t || !out || N <= 0 || dim <= 1) return VF_ERR_ARG;
    const int total = N * (dim / 2);
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((timestep_embedding_kernel<TT>), dim3((total + 255) / 256), dim3(256), 0, stream, t, (E*)out, N, dim);
    });
    return ok();
}

int vf_launch_silu(const void* x, void* y, long count, int in_f32, int dtype, hipStream_t stream) {
    if (!x || !y || count <= 0) return VF_ERR_ARG;
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        if (in_f32) hipLaunchKernelGGL((silu_kernel<TT, true>), dim3(grid_for(count)), dim3(256), 0, stream, x, (E*)y, count);
        else hipLaunchKernelGGL((silu_kernel<TT, false>), dim3(grid_for(count)), dim3(256), 0, stream, x, (E*)y, count);
    });
    return ok();
}

int vf_launch_softmax_rows(const float* S, long lds_, void* P, long ldp, int M, int N, float scale, int dtype, hipStream_t stream) {
    if (!S || !P || M <= 0 || N <= 0) return VF_ERR_ARG;
    if ((N & 3) || (lds_ & 3) || (ldp & 3) || ((uintptr_t)S & 15) || ((uintptr_t)P & 7)) return VF_ERR_ALIGN;
    if (N > 16384) return VF_ERR_SHAPE;
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((softmax_rows_kernel<TT>), dim3(M), dim3(256), 0, stream, S, lds_, (E*)P, ldp, N, scale);
    });
    return ok();
}

int vf_launch_vae_sample(const float* moments, long ldm, const float* noise, float* z, int F, int hw, int zc, float scale,
                         hipStream_t stream) {
    if (!moments || !z || F <= 0 || hw <= 0 || zc <= 0 || ldm < 2 * zc) return VF_ERR_ARG;
    hipLaunchKernelGGL(vae_sample_kernel, dim3(grid_for((long)F * hw * zc)), dim3(256), 0, stream, moments, ldm, noise, z, F, hw, zc, scale);
    return ok();
}

int vf_launch_cast(const float* src, void* dst, long count, int dtype, hipStream_t stream) {
    if (!src || !dst || count <= 0) return VF_ERR_ARG;
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((cast_kernel<TT>), dim3(grid_for(count)), dim3(256), 0, stream, src, (E*)dst, count);
    });
    return ok();
}

int vf_launch_pack_input(const float* x, const float* inv, const float* inpaint, const float* mask, void* out,
                         int F, int h, int w, int cpad, int dtype, hipStream_t stream) {
    if (!x || !inpaint || !mask || !out || F <= 0 || h <= 0 || w <= 0) return VF_ERR_ARG;
    if (cpad < 9 || (cpad & 7)) return VF_ERR_SHAPE;
    const long total = (long)(inv ? 3 : 2) * F * h * w * cpad;
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((pack_input_kernel<TT>), dim3(grid_for(total)), dim3(256), 0, stream, x, inv, inpaint, mask, (E*)out, F, h * w, cpad);
    });
    return ok();
}

int vf_launch_nchw_to_nhwc(const float* x, void* out, int N, int C, int hw, int cpad, int dtype, hipStream_t stream) {
    if (!x || !out || N <= 0 || C <= 0 || hw <= 0 || cpad < C) return VF_ERR_ARG;
    const long total = (long)N * hw * cpad;
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<TT>), dim3(grid_for(total)), dim3(256), 0, stream, x, (E*)out, N, C, hw, cpad);
    });
    return ok();
}

int vf_launch_nhwc_to_nchw_f32(const float* x, long ldx, float* out, int N, int C, int hw, hipStream_t stream) {
    if (!x || !out || N <= 0 || C <= 0 || hw <= 0 || ldx < C) return VF_ERR_ARG;
    const long total = (long)N * C * hw;
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, ldx, out, N, C, hw);
    return ok();
}

int vf_launch_ddim_step(const float* eps, long lde, const float* x, const float* inv, float* x_prev, float* pred_x0,
                        float* x_prev_recon, int F, int C, int hw, float scale, float a_t, float a_prev, float sigma_t,
                        float sqrt_1m_at, const float* noise, int single, hipStream_t stream) {
    if (!eps || !x || !x_prev || F <= 0 || C <= 0 || hw <= 0 || lde < C || single < 0 || single > 2) return VF_ERR_ARG;
    const long total = (long)F * C * hw;
    hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(total)), dim3(256), 0, stream, eps, lde, x, inv, x_prev, pred_x0,
                       x_prev_recon, F, C, hw, scale, a_t, a_prev, sigma_t, sqrt_1m_at, noise, single);
    return ok();
}

int vf_launch_copy2d(const void* src, long lds_, void* dst, long ldd, long rows, int cols, int dtype,
                     hipStream_t stream) {
    if (!src || !dst || rows <= 0 || cols <= 0) return VF_ERR_ARG;
    if ((cols & 7) || (lds_ & 7) || (ldd & 7) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return VF_ERR_ALIGN;
    const long total = rows * (cols / 8);
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((copy2d_kernel<TT>), dim3(grid_for(total)), dim3(256), 0, stream, (const E*)src, lds_, (E*)dst, ldd, rows, cols);
    });
    return ok();
}

int vf_launch_temporal_gauss(const void* src, long ld_src, long fs_src, void* dst1, void* dst2, long ld_dst, long fs_dst,
                             int F, int n, int C, int dtype, hipStream_t stream) {
    if (!src || !dst1 || F <= 0 || n <= 0 || C <= 0) return VF_ERR_ARG;      // (dst2 optional: a batch without its last chunk)
    if ((C & 7) || (ld_src & 7) || (ld_dst & 7) || (fs_src & 7) || (fs_dst & 7)) return VF_ERR_ALIGN;
    if (((uintptr_t)src | (uintptr_t)dst1 | (uintptr_t)dst2) & 15) return VF_ERR_ALIGN;
    const long total = (long)n * (C / 8);
    dim3 grid(grid_for(total, 256, 2048), F);
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((temporal_gauss_kernel<TT>), grid, dim3(256), 0, stream, (const E*)src, ld_src, fs_src, (E*)dst1,
                           (E*)dst2, ld_dst, fs_dst, F, n, C);
    });
    return ok();
}

size_t vf_adain_workspace_bytes(long rows, int C) {
    const long nblocks = (rows + 3) / 4;
    return (size_t)rows * C * sizeof(float) + (size_t)nblocks * 2 * sizeof(double) + 256;
}

int vf_launch_adain(const void* a, long lda, const void* b, long ldb, void* dst, long ldd, long rows, int C, void* ws,
                    int dtype, hipStream_t stream) {
    if (!a || !b || !dst || !ws || rows <= 0 || C <= 1) return VF_ERR_ARG;
    if ((C & 7) || (lda & 7) || (ldb & 7) || (ldd & 7)) return VF_ERR_ALIGN;
    if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)dst | (uintptr_t)ws) & 15) return VF_ERR_ALIGN;
    if (C > 2048) return VF_ERR_SHAPE;
    float* fused = (float*)ws;
    const long nblocks = (rows + 3) / 4;
    double* partial = (double*)((char*)ws + (((size_t)rows * C * sizeof(float) + 15) & ~(size_t)15));
    float* inv = (float*)(partial + 2 * nblocks);
    const int ch8 = (C + 511) / 512;
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        dim3 g((unsigned)nblocks);
        switch (ch8) {
            case 1: hipLaunchKernelGGL((adain_rows_kernel<TT, 1>), g, dim3(256), 0, stream, (const E*)a, lda, (const E*)b, ldb, fused, (long)C, (int)rows, C, partial); break;
            case 2: hipLaunchKernelGGL((adain_rows_kernel<TT, 2>), g, dim3(256), 0, stream, (const E*)a, lda, (const E*)b, ldb, fused, (long)C, (int)rows, C, partial); break;
            case 3: hipLaunchKernelGGL((adain_rows_kernel<TT, 3>), g, dim3(256), 0, stream, (const E*)a, lda, (const E*)b, ldb, fused, (long)C, (int)rows, C, partial); break;
            default: hipLaunchKernelGGL((adain_rows_kernel<TT, 4>), g, dim3(256), 0, stream, (const E*)a, lda, (const E*)b, ldb, fused, (long)C, (int)rows, C, partial); break;
        }
        hipLaunchKernelGGL(adain_reduce_kernel, dim3(1), dim3(256), 0, stream, (const double*)partial, (int)nblocks, (double)rows * C, inv);
        hipLaunchKernelGGL((adain_scale_kernel<TT>), dim3(grid_for(rows * (C / 8))), dim3(256), 0, stream, (const float*)fused, (long)C,
                           (const float*)inv, (E*)dst, ldd, rows, C);
    });
    return ok();
}

int vf_launch_gn_finalize_cols(const float* colstats, long ld, int nimg, int hw, int C, int groups, float eps, float* stats,
                               hipStream_t stream) {
    if (!colstats || !stats || nimg <= 0 || hw <= 0 || C <= 0 || groups <= 0) return VF_ERR_ARG;
    if ((hw & 63) || (C % groups) || ld < C) return VF_ERR_SHAPE;
    hipLaunchKernelGGL(gn_finalize_cols_kernel, dim3(groups, nimg), dim3(64), 0, stream, colstats, ld, hw, C, groups, eps, stats);
    return ok();
}
