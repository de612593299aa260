// Glue kernels of the RAFT-shaped optical-flow producer behind `return_flow` (REFace/scripts/temporal_flow.py:27-38,
// 163-188 call torchvision's raft_large with 20 updates).  The convolutions and the all-pairs correlation run on the GEMM /
// implicit-GEMM kernels (gemm.hip, conv.hip); what is here is the HBM-bound work between them, on token-major (NHWC) buffers:
//
//   im2col_kernel             the 7x7 / 1x5 / 5x1 windows as an explicit [M][KH KW C] matrix (the implicit GEMM covers <= 9 taps)
//   chan_stats_* / chan_norm_act_kernel   InstanceNorm2d (per image and channel, biased variance, eps 1e-5) + ReLU / tanh /
//                             sigmoid / residual add -- also the plain activations of the BatchNorm-folded context encoder
//   gru_gate / gru_update     ConvGRU: z, r = sigmoid(conv(hx)), r * h  and  h = (1 - z) h + z tanh(conv([r h, x])), fp32 master state
//   avgpool2_f32_kernel       the 4-level correlation pyramid
//   corr_lookup_kernel        9 x 9 bilinear window per level around (x + flow) / 2^level, zero outside (grid_sample, align_corners)
//   flow_update_kernel        coords1 += delta_flow; 16-bit copies of the flow where the next convolutions read it
//   convex_upsample_kernel    softmax over the 9 neighbours of the 0.25-scaled mask, x8 convex combination of 8 * flow
//
// PARITY UNPINNED: torchvision is a third-party dependency that is neither under /root/reference nor installed here, and its
// weights are not available offline; these kernels are tested against oracle/raft.py, a restatement of the published model.
#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

inline int ok() { return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH; }
inline unsigned grid1(long total, int per = 256) { return (unsigned)std::min<long>((total + per - 1) / per, 16384); }

#define DISPATCH_DTYPE(dtype, CALL)                                \
    if ((dtype) == VF_DTYPE_F16) { using TT = F16; CALL; }         \
    else if ((dtype) == VF_DTYPE_BF16) { using TT = BF16; CALL; }  \
    else return VF_ERR_DTYPE;

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3 };
__device__ __forceinline__ float act_f(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    if (act == ACT_TANH) return tanhf(v);
    if (act == ACT_SIGMOID) return 1.0f / (1.0f + __expf(-v));
    return v;
}

// out[m][tap * C + c] = x[img][oy * stride - pad_y + ky][ox * stride - pad_x + kx][c] (zero outside), 8 channels per thread
template <class TT>
__global__ __launch_bounds__(256) void im2col_kernel(const typename TT::elem* __restrict__ x, long ldx, int H, int W, int C, int KH,
                                                     int KW, int stride, int pad_y, int pad_x, int OH, int OW,
                                                     typename TT::elem* __restrict__ out, long ldo, long M) {
    using V8 = typename TT::v8;
    const int c8 = C / 8, per_row = KH * KW * c8;
    const long total = M * per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / per_row;
        const int r = (int)(i - m * per_row), tap = r / c8, ch = r - tap * c8;
        const int ky = tap / KW, kx = tap - ky * KW;
        const int ox = (int)(m % OW), oy = (int)((m / OW) % OH);
        const long img = m / ((long)OW * OH);
        const int iy = oy * stride - pad_y + ky, ix = ox * stride - pad_x + kx;
        V8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const V8*>(x + ((img * H + iy) * W + ix) * ldx + ch * 8);
        *reinterpret_cast<V8*>(out + m * ldo + (long)tap * C + ch * 8) = v;
    }
}

// partial[(img * S + s)][C][2] = (sum, sum of squares) over pixel slice s of image img; a block covers 64 channels
template <class TT>
__global__ __launch_bounds__(256) void chan_stats_partial_kernel(const typename TT::elem* __restrict__ x, long ldx, int hw, int C, int S,
                                                                 float* __restrict__ partial) {
    using V8 = typename TT::v8;
    __shared__ float red[32][64][2];
    const int img = blockIdx.y, s = blockIdx.z, t = threadIdx.x;
    const int chunk = t & 7, lane = t >> 3, c0 = blockIdx.x * 64 + chunk * 8;
    const int per = (hw + S - 1) / S, p0 = s * per, p1 = min(hw, p0 + per);
    float sum[8], sq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[j] = sq[j] = 0.0f;
    if (c0 < C)
        for (int p = p0 + lane; p < p1; p += 32) {
            const V8 v = *reinterpret_cast<const V8*>(x + ((long)img * hw + p) * ldx + c0);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = to_f32(v[j]); sum[j] += f; sq[j] += f * f; }
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[lane][chunk * 8 + j][0] = sum[j]; red[lane][chunk * 8 + j][1] = sq[j]; }
    __syncthreads();
    if (t < 128) {
        const int c = t >> 1, k = t & 1;
        float a = 0.0f;
        for (int l = 0; l < 32; ++l) a += red[l][c][k];          // fixed order: reproducible
        if (blockIdx.x * 64 + c < C) partial[(((long)img * S + s) * C + blockIdx.x * 64 + c) * 2 + k] = a;
    }
}

__global__ __launch_bounds__(256) void chan_stats_finalize_kernel(const float* __restrict__ partial, int nimg, int hw, int C, int S, float eps,
                                                                  float* __restrict__ stats) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)nimg * C) return;
    const long img = i / C;
    const int c = (int)(i - img * C);
    double s = 0.0, q = 0.0;
    for (int k = 0; k < S; ++k) {
        s += partial[((img * S + k) * C + c) * 2];
        q += partial[((img * S + k) * C + c) * 2 + 1];
    }
    const double mean = s / hw;
    double var = q / hw - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[i * 2] = (float)mean;
    stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// y = act((x - mean) * rstd [stats != null] + res [res != null]); rows m = img * hw + p; 8 channels per thread
template <class TT>
__global__ __launch_bounds__(256) void chan_norm_act_kernel(const typename TT::elem* __restrict__ x, long ldx, const float* __restrict__ stats,
                                                            const typename TT::elem* __restrict__ res, long ldr,
                                                            typename TT::elem* __restrict__ y, long ldy, float* __restrict__ y32, long ldy32,
                                                            long M, int hw, int C, int act) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    const int c8 = C / 8;
    const long total = M * c8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / c8;
        const int c0 = (int)(i - m * c8) * 8;
        const V8 v = *reinterpret_cast<const V8*>(x + m * ldx + c0);
        V8 r;
        if (res) r = *reinterpret_cast<const V8*>(res + m * ldr + c0);
        const float* st = stats ? stats + ((m / hw) * C + c0) * 2 : nullptr;
        V8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = to_f32(v[j]);
            if (st) f = (f - st[2 * j]) * st[2 * j + 1];
            if (res) f += to_f32(r[j]);
            f = act_f(f, act);
            o[j] = from_f32<E>(f);
            if (y32) y32[m * ldy32 + c0 + j] = f;
        }
        if (y) *reinterpret_cast<V8*>(y + m * ldy + c0) = o;
    }
}

// zr [M][2 Hd] pre-activations (z | r); h32 [M][Hd] fp32 state: z -> zout (16-bit), sigmoid(r) * h -> rh (16-bit)
template <class TT>
__global__ __launch_bounds__(256) void gru_gate_kernel(const typename TT::elem* __restrict__ zr, long ldzr, const float* __restrict__ h32,
                                                       typename TT::elem* __restrict__ zout, long ldz, typename TT::elem* __restrict__ rh,
                                                       long ldrh, long M, int Hd) {
    using E = typename TT::elem;
    const long total = M * Hd;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / Hd;
        const int c = (int)(i - m * Hd);
        const float z = act_f(to_f32(zr[m * ldzr + c]), ACT_SIGMOID), r = act_f(to_f32(zr[m * ldzr + Hd + c]), ACT_SIGMOID);
        zout[m * ldz + c] = from_f32<E>(z);
        rh[m * ldrh + c] = from_f32<E>(r * h32[i]);
    }
}

// h = (1 - z) h + z tanh(q): fp32 master in place, 16-bit copies into up to two buffers
template <class TT>
__global__ __launch_bounds__(256) void gru_update_kernel(const typename TT::elem* __restrict__ q, long ldq, const typename TT::elem* __restrict__ z,
                                                         long ldz, float* __restrict__ h32, typename TT::elem* __restrict__ h16a, long lda,
                                                         typename TT::elem* __restrict__ h16b, long ldb, long M, int Hd) {
    using E = typename TT::elem;
    const long total = M * Hd;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / Hd;
        const int c = (int)(i - m * Hd);
        const float zz = to_f32(z[m * ldz + c]);
        const float h = (1.0f - zz) * h32[i] + zz * tanhf(to_f32(q[m * ldq + c]));
        h32[i] = h;
        if (h16a) h16a[m * lda + c] = from_f32<E>(h);
        if (h16b) h16b[m * ldb + c] = from_f32<E>(h);
    }
}

__global__ __launch_bounds__(256) void avgpool2_f32_kernel(const float* __restrict__ x, float* __restrict__ y, long R, int h, int w) {
    const int oh = h / 2, ow = w / 2;
    const long total = R * oh * ow;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
        const long r = i / ((long)ow * oh);
        const float* p = x + (r * h + 2 * oy) * w + 2 * ox;
        y[i] = (p[0] + p[1] + p[w] + p[w + 1]) * 0.25f;
    }
}

struct CorrLevels { const float* vol[4]; int h[4], w[4]; };

// out[m][l * 81 + i * 9 + j] = scale * bilinear(vol_l[m], (x + fx) / 2^l + (i - 4), (y + fy) / 2^l + (j - 4)), zero outside
template <class TT>
__global__ __launch_bounds__(256) void corr_lookup_kernel(CorrLevels lv, int levels, const float* __restrict__ flow32, int h, int w, float scale,
                                                          typename TT::elem* __restrict__ out, long ldo, long M) {
    using E = typename TT::elem;
    const int per = levels * 81;
    const long total = M * per;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / per;
        const int k = (int)(idx - m * per), l = k / 81, ij = k - l * 81, i = ij / 9, j = ij - i * 9;
        const int px = (int)(m % w), py = (int)((m / w) % h);
        const float inv = 1.0f / (float)(1 << l);
        const float sx = ((float)px + flow32[m * 2]) * inv + (float)(i - 4), sy = ((float)py + flow32[m * 2 + 1]) * inv + (float)(j - 4);
        const int hl = lv.h[l], wl = lv.w[l];
        const float* v = lv.vol[l] + m * (long)hl * wl;
        const float fx = floorf(sx), fy = floorf(sy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float ax = sx - fx, ay = sy - fy;
        auto at = [&](int yy, int xx) { return (yy >= 0 && yy < hl && xx >= 0 && xx < wl) ? v[(long)yy * wl + xx] : 0.0f; };
        float val = 0.0f;
        if (sx > -1.0f && sx < (float)wl && sy > -1.0f && sy < (float)hl)
            val = (1.0f - ay) * ((1.0f - ax) * at(y0, x0) + ax * at(y0, x0 + 1)) + ay * ((1.0f - ax) * at(y0 + 1, x0) + ax * at(y0 + 1, x0 + 1));
        out[m * ldo + k] = from_f32<E>(val * scale);
    }
}

// flow32 += delta32[:, 0:2]; 16-bit copies of the new flow into up to three [M][ld] buffers at a column offset
template <class TT>
__global__ __launch_bounds__(256) void flow_update_kernel(float* __restrict__ flow32, const float* __restrict__ delta32, long ldd,
                                                          typename TT::elem* __restrict__ a, long lda, typename TT::elem* __restrict__ b, long ldb,
                                                          typename TT::elem* __restrict__ c, long ldc, long M) {
    using E = typename TT::elem;
    for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float f = flow32[m * 2 + k] + (delta32 ? delta32[m * ldd + k] : 0.0f);
            flow32[m * 2 + k] = f;
            const E e = from_f32<E>(f);
            if (a) a[m * lda + k] = e;
            if (b) b[m * ldb + k] = e;
            if (c) c[m * ldc + k] = e;
        }
    }
}

// out [B][2][8 h][8 w] fp32; mask32 [M][576] (index k * 64 + fy * 8 + fx), multiplied by `mult` (0.25) before the softmax over k
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ mask32, long ldm, const float* __restrict__ flow32,
                                                              float* __restrict__ out, int B, int h, int w, float mult) {
    const long total = (long)B * h * w * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int f = (int)(i & 63), fy = f >> 3, fx = f & 7;
        const long m = i >> 6;
        const int x = (int)(m % w), y = (int)((m / w) % h);
        const long b = m / ((long)w * h);
        float lg[9], mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { lg[k] = mult * mask32[m * ldm + k * 64 + f]; mx = fmaxf(mx, lg[k]); }
        float den = 0.0f, ux = 0.0f, uy = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float e = __expf(lg[k] - mx);
            den += e;
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
                const long mm = (b * h + yy) * w + xx;
                ux += e * 8.0f * flow32[mm * 2];
                uy += e * 8.0f * flow32[mm * 2 + 1];
            }
        }
        const long o = ((b * 2) * (8L * h) + (8 * y + fy)) * (8L * w) + 8 * x + fx;
        out[o] = ux / den;
        out[o + 64L * h * w] = uy / den;
    }
}

}  // namespace

int vf_launch_im2col(const void* x, long ldx, int nimg, int H, int W, int C, int KH, int KW, int stride, int pad_y, int pad_x, void* out,
                     long ldo, int dtype, hipStream_t stream) {
    if (!x || !out || nimg <= 0 || H <= 0 || W <= 0 || C <= 0 || KH <= 0 || KW <= 0 || stride <= 0) return VF_ERR_ARG;
    if ((C & 7) || (ldx & 7) || (ldo & 7) || (((uintptr_t)x | (uintptr_t)out) & 15)) return VF_ERR_ALIGN;
    const int OH = (H + 2 * pad_y - KH) / stride + 1, OW = (W + 2 * pad_x - KW) / stride + 1;
    if (OH <= 0 || OW <= 0 || ldo < (long)KH * KW * C || ldx < C) return VF_ERR_SHAPE;
    const long M = (long)nimg * OH * OW;
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((im2col_kernel<TT>), dim3(grid1(M * KH * KW * (C / 8))), dim3(256), 0, stream, (const E*)x, ldx, H, W, C, KH, KW, stride,
                           pad_y, pad_x, OH, OW, (E*)out, ldo, M);
    });
    return ok();
}

int vf_chan_stats_slices(int hw) { return hw >= 16384 ? 16 : (hw >= 2048 ? 4 : 1); }

int vf_launch_chan_stats(const void* x, long ldx, int nimg, int hw, int C, float eps, float* partial, float* stats, int dtype,
                         hipStream_t stream) {
    if (!x || !partial || !stats || nimg <= 0 || hw <= 0 || C <= 0) return VF_ERR_ARG;
    if ((C & 7) || (ldx & 7) || ((uintptr_t)x & 15)) return VF_ERR_ALIGN;
    const int S = vf_chan_stats_slices(hw);
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((chan_stats_partial_kernel<TT>), dim3((C + 63) / 64, nimg, S), dim3(256), 0, stream, (const E*)x, ldx, hw, C, S, partial);
    });
    hipLaunchKernelGGL(chan_stats_finalize_kernel, dim3(((long)nimg * C + 255) / 256), dim3(256), 0, stream, partial, nimg, hw, C, S, eps, stats);
    return ok();
}

int vf_launch_chan_norm_act(const void* x, long ldx, const float* stats, const void* res, long ldr, void* y, long ldy, float* y32,
                            long ldy32, long M, int hw, int C, int act, int dtype, hipStream_t stream) {
    if (!x || (!y && !y32) || M <= 0 || hw <= 0 || C <= 0 || act < 0 || act > 3) return VF_ERR_ARG;
    if ((C & 7) || (ldx & 7) || (ldy & 7) || (ldr & 7) || (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) & 15)) return VF_ERR_ALIGN;
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((chan_norm_act_kernel<TT>), dim3(grid1(M * (C / 8))), dim3(256), 0, stream, (const E*)x, ldx, stats, (const E*)res, ldr,
                           (E*)y, ldy, y32, ldy32, M, hw, C, act);
    });
    return ok();
}

int vf_launch_gru_gate(const void* zr, long ldzr, const float* h32, void* z, long ldz, void* rh, long ldrh, long M, int Hd, int dtype,
                       hipStream_t stream) {
    if (!zr || !h32 || !z || !rh || M <= 0 || Hd <= 0) return VF_ERR_ARG;
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((gru_gate_kernel<TT>), dim3(grid1(M * Hd)), dim3(256), 0, stream, (const E*)zr, ldzr, h32, (E*)z, ldz, (E*)rh, ldrh, M, Hd);
    });
    return ok();
}

int vf_launch_gru_update(const void* q, long ldq, const void* z, long ldz, float* h32, void* h16a, long lda, void* h16b, long ldb, long M,
                         int Hd, int dtype, hipStream_t stream) {
    if (!q || !z || !h32 || M <= 0 || Hd <= 0) return VF_ERR_ARG;
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((gru_update_kernel<TT>), dim3(grid1(M * Hd)), dim3(256), 0, stream, (const E*)q, ldq, (const E*)z, ldz, h32, (E*)h16a, lda,
                           (E*)h16b, ldb, M, Hd);
    });
    return ok();
}

int vf_launch_avgpool2_f32(const float* x, float* y, long R, int h, int w, hipStream_t stream) {
    if (!x || !y || R <= 0 || h < 2 || w < 2) return VF_ERR_ARG;
    hipLaunchKernelGGL(avgpool2_f32_kernel, dim3(grid1(R * (h / 2) * (w / 2))), dim3(256), 0, stream, x, y, R, h, w);
    return ok();
}

int vf_launch_corr_lookup(const float* const* vols, const int* hs, const int* ws, int levels, const float* flow32, int h, int w, float scale,
                          void* out, long ldo, long M, int dtype, hipStream_t stream) {
    if (!vols || !hs || !ws || !flow32 || !out || levels <= 0 || levels > 4 || M <= 0 || h <= 0 || w <= 0) return VF_ERR_ARG;
    if (ldo < levels * 81) return VF_ERR_SHAPE;
    CorrLevels lv{};
    for (int l = 0; l < levels; ++l) {
        if (!vols[l] || hs[l] <= 0 || ws[l] <= 0) return VF_ERR_ARG;
        lv.vol[l] = vols[l]; lv.h[l] = hs[l]; lv.w[l] = ws[l];
    }
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((corr_lookup_kernel<TT>), dim3(grid1(M * levels * 81)), dim3(256), 0, stream, lv, levels, flow32, h, w, scale, (E*)out, ldo, M);
    });
    return ok();
}

int vf_launch_flow_update(float* flow32, const float* delta32, long ldd, void* a, long lda, void* b, long ldb, void* c, long ldc, long M,
                          int dtype, hipStream_t stream) {
    if (!flow32 || M <= 0) return VF_ERR_ARG;
    DISPATCH_DTYPE(dtype, {
        using E = typename TT::elem;
        hipLaunchKernelGGL((flow_update_kernel<TT>), dim3(grid1(M)), dim3(256), 0, stream, flow32, delta32, ldd, (E*)a, lda, (E*)b, ldb, (E*)c, ldc, M);
    });
    return ok();
}

int vf_launch_convex_upsample(const float* mask32, long ldm, const float* flow32, float* out, int B, int h, int w, float mult,
                              hipStream_t stream) {
    if (!mask32 || !flow32 || !out || B <= 0 || h <= 0 || w <= 0) return VF_ERR_ARG;
    if (ldm < 576) return VF_ERR_SHAPE;
    hipLaunchKernelGGL(convex_upsample_kernel, dim3(grid1((long)B * h * w * 64)), dim3(256), 0, stream, mask32, ldm, flow32, out, B, h, w, mult);
    return ok();
}
