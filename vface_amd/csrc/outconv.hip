// The UNet's OUT layer in one launch (gfx950):  eps = conv3x3( SiLU( GroupNorm32(h) ) ),  320 -> 4 channels
// (REFace/ldm/modules/diffusionmodules/openaimodel.py:712-716 `self.out = normalization, SiLU, zero_module(conv_nd(.., model_channels,
// out_channels, 3, padding=1))`, applied at :905).
//
// Through the general kernels this was gn_finalize + gn_apply (read 126 MB, write 63 MB) + an implicit-GEMM convolution whose 128-wide
// output tile is 97 % padding for four output channels (72 GFLOP executed for 2.3 useful: 109 us) = 153 us per forward.  Here a
// workgroup owns a 16 x 8-pixel tile: per 64-channel chunk it reads the 18 x 10 halo patch of the fp32 carrier once, applies the
// GroupNorm scale / shift of its image (vface_groupnorm_coeffs_from_cols: gn_apply's arithmetic) and SiLU, rounds to 16 bits -- the
// same single rounding the separate pass made -- into LDS, and every thread accumulates 9 taps x 16 channels x Cout products of its pixel with
// v_dot2_f32_f16 (fp16 products are exact in fp32, fp32 accumulation: what the matrix cores compute, in another order) against weights
// that sit in scalar registers.  HBM traffic: the carrier once (126 MB) + 1.5 MB out.
#include <type_traits>

#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

constexpr int OC_TX = 16, OC_TY = 8;     // output tile: 128 pixels (a 64 x 64 map x 24 images = 768 workgroups = 3 per CU)
constexpr int OC_PX = OC_TX + 2, OC_PY = OC_TY + 2;      // patch with its halo
constexpr int OC_THREADS = 512;          // 128 pixels x 4 channel slices: eight waves per workgroup, 24 per CU to hide each other's latencies
constexpr int OC_ITEMS = OC_PX * OC_PY * 8;              // (patch pixel, 8-channel group) per 64-channel chunk
constexpr int OC_NIT = (OC_ITEMS + OC_THREADS - 1) / OC_THREADS;
constexpr int OC_MAXC = 640;             // input channels the scale / shift table in LDS holds
constexpr int OC_PITCH = 64 + 8;         // halfs per patch pixel in LDS (16 B of padding: tap-shifted rows spread over the banks)

template <class TT, bool IN32, int COUT>
__global__ __launch_bounds__(OC_THREADS) void gn_silu_conv3x3_small_kernel(OutConvParams p) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    __shared__ __attribute__((aligned(16))) E patch[OC_PX * OC_PY * OC_PITCH];
    __shared__ float sAB[2 * OC_MAXC];
    static_assert(sizeof(patch) >= 3 * OC_TX * OC_TY * 4 * sizeof(float), "the partial sums are reduced through the patch");
    const int t = threadIdx.x;
    const int tiles_x = (p.W + OC_TX - 1) / OC_TX, tiles_y = (p.H + OC_TY - 1) / OC_TY;
    const int img = blockIdx.x / (tiles_x * tiles_y), tl = blockIdx.x - img * (tiles_x * tiles_y);
    const int ty0 = (tl / tiles_x) * OC_TY, tx0 = (tl % tiles_x) * OC_TX;
    const int pix = t & 127, py = pix >> 4, px = pix & 15;   // this thread's output pixel inside the tile ...
    const int slice = __builtin_amdgcn_readfirstlane(t >> 7);            // ... and its 16 channels of every 64-channel chunk
    const int oy = ty0 + py, ox = tx0 + px;
    const long img_row0 = (long)img * p.H * p.W;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    const unsigned* wq = reinterpret_cast<const unsigned*>(p.Wt);      // channel pairs
    const int nchunks = p.Cin / 64;
    for (int ch = t; ch < p.Cin; ch += OC_THREADS) {
        const float2 ab = *reinterpret_cast<const float2*>(p.ab + ((long)img * p.ld_ab + ch) * 2);
        sAB[ch] = ab.x; sAB[OC_MAXC + ch] = ab.y;
    }
    // The patch of chunk c + 1 is in flight (registers) while chunk c's products are formed.
    using RAW = typename std::conditional<IN32, float4, V8>::type;
    RAW raw[OC_NIT][IN32 ? 2 : 1];
    auto item_row = [&](int i, long& row) -> bool {          // is item i of this thread inside the image?  (and its row)
        const int it = t + i * OC_THREADS, pp = it >> 3;
        const int iy = ty0 - 1 + pp / OC_PX, ix = tx0 - 1 + pp % OC_PX;
        row = img_row0 + (long)iy * p.W + ix;
        return it < OC_ITEMS && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    };
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < OC_NIT; ++i) {
            long row;
            if (!item_row(i, row)) continue;
            const int g8 = (t + i * OC_THREADS) & 7;
            if constexpr (IN32) {
                const float* xr = reinterpret_cast<const float*>(p.x) + row * p.ldx + c * 64 + g8 * 8;
                raw[i][0] = *reinterpret_cast<const float4*>(xr);
                raw[i][1] = *reinterpret_cast<const float4*>(xr + 4);
            } else {
                raw[i][0] = *reinterpret_cast<const V8*>(reinterpret_cast<const E*>(p.x) + row * p.ldx + c * 64 + g8 * 8);
            }
        }
    };
    auto store_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < OC_NIT; ++i) {
            const int it = t + i * OC_THREADS, pp = it >> 3, g8 = it & 7;
            if (it >= OC_ITEMS) continue;
            long row;
            V8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (E)0.0f;      // zero padding applies to the ACTIVATED tensor
            if (item_row(i, row)) {
                float v[8];
                if constexpr (IN32) {
                    const float4 a = raw[i][0], b = raw[i][1];
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = to_f32(raw[i][0][j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = v[j] * sAB[c * 64 + g8 * 8 + j] + sAB[OC_MAXC + c * 64 + g8 * 8 + j];
                    o[j] = from_f32<E>(silu_f(f));
                }
            }
            *reinterpret_cast<V8*>(patch + pp * OC_PITCH + g8 * 8) = o;
        }
    };
    load_chunk(0);
    __syncthreads();                                         // sAB
    for (int c = 0; c < nchunks; ++c) {
        store_chunk(c);
        __syncthreads();
        if (c + 1 < nchunks) load_chunk(c + 1);
        // ---- this pixel's 9 taps x 16 channels x COUT products; weights: [o][(chunk, tap, ch)] 16-bit, wave-uniform channel pairs
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const E* pr = patch + ((py + tap / 3) * OC_PX + (px + tap % 3)) * OC_PITCH + slice * 16;
#pragma unroll
            for (int g8 = 0; g8 < 2; ++g8) {
                const V8 xv = *reinterpret_cast<const V8*>(pr + g8 * 8);
#pragma unroll
                for (int o = 0; o < COUT; ++o) {
                    const unsigned* wo = wq + ((long)o * 9 * p.Cin + (long)(c * 9 + tap) * 64 + slice * 16 + g8 * 8) / 2;
#pragma unroll
                    for (int j2 = 0; j2 < 4; ++j2) {
                        const unsigned wpair = wo[j2];
                        if constexpr (std::is_same<TT, F16>::value) {
                            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                            const h2 xa = {xv[2 * j2], xv[2 * j2 + 1]};
                            acc[o] = __builtin_amdgcn_fdot2(xa, __builtin_bit_cast(h2, wpair), acc[o], false);
                        } else {
                            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                            const b2 wb = __builtin_bit_cast(b2, wpair);
                            acc[o] = fmaf((float)xv[2 * j2], (float)wb[0], acc[o]);
                            acc[o] = fmaf((float)xv[2 * j2 + 1], (float)wb[1], acc[o]);
                        }
                    }
                }
            }
        }
        __syncthreads();                                     // every thread is done with this chunk's patch
    }
    // ---- the four channel slices of a pixel, added in a fixed order through LDS (the patch is free now)
    float* red = reinterpret_cast<float*>(patch);
    if (slice > 0) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) red[((slice - 1) * 128 + pix) * 4 + o] = acc[o];
    }
    __syncthreads();
    if (slice == 0 && oy < p.H && ox < p.W) {
        float* d = p.out + (img_row0 + (long)oy * p.W + ox) * p.ldo;
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            const float r = ((acc[o] + red[pix * 4 + o]) + red[(128 + pix) * 4 + o]) + red[(256 + pix) * 4 + o];
            d[o] = r + (p.bias ? p.bias[o] : 0.f);
        }
    }
}

template <class TT>
int launch_t(const OutConvParams& p, hipStream_t stream) {
    const int tiles = ((p.W + OC_TX - 1) / OC_TX) * ((p.H + OC_TY - 1) / OC_TY);
    dim3 grid((unsigned)(p.nimg * tiles));
#define OC_LAUNCH(IN32_, CO_) hipLaunchKernelGGL((gn_silu_conv3x3_small_kernel<TT, IN32_, CO_>), grid, dim3(OC_THREADS), 0, stream, p)
    if (p.Cout == 4) { if (p.in_f32) OC_LAUNCH(true, 4); else OC_LAUNCH(false, 4); }
    else if (p.Cout == 3) { if (p.in_f32) OC_LAUNCH(true, 3); else OC_LAUNCH(false, 3); }
    else return VF_ERR_SHAPE;
#undef OC_LAUNCH
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

}  // namespace

bool vf_out_conv_supported(int Cin, int Cout) { return Cin > 0 && Cin <= 640 && (Cin % 64) == 0 && (Cout == 3 || Cout == 4); }

int vf_launch_out_conv(const OutConvParams& p, int dtype, hipStream_t stream) {
    if (!p.x || !p.ab || !p.Wt || !p.out || p.nimg <= 0 || p.H <= 0 || p.W <= 0) return VF_ERR_ARG;
    if (!vf_out_conv_supported(p.Cin, p.Cout)) return VF_ERR_SHAPE;
    if ((p.ldx & (p.in_f32 ? 3 : 7)) || p.ld_ab < p.Cin || p.ldo < p.Cout) return VF_ERR_ALIGN;
    if (((uintptr_t)p.x | (uintptr_t)p.Wt) & 15 || ((uintptr_t)p.ab & 7)) return VF_ERR_ALIGN;
    if (dtype == VF_DTYPE_F16) return launch_t<F16>(p, stream);
    if (dtype == VF_DTYPE_BF16) return launch_t<BF16>(p, stream);
    return VF_ERR_DTYPE;
}
