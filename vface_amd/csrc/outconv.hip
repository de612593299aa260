// The UNet's OUT layer in one launch (gfx950):  eps = conv3x3( SiLU( GroupNorm32(h) ) ),  320 -> 4 channels
// (REFace/ldm/modules/diffusionmodules/openaimodel.py:712-716 `self.out = normalization, SiLU, zero_module(conv_nd(.., model_channels,
// out_channels, 3, padding=1))`, applied at :905).
//
// Through the general kernels this was gn_finalize + gn_apply (read 126 MB, write 63 MB) + an implicit-GEMM convolution whose 128-wide
// output tile is 97 % padding for four output channels (72 GFLOP executed for 2.3 useful: 109 us) = 153 us per forward.  Here a
// workgroup owns a 16 x 16-pixel tile: per 64-channel chunk it reads the 18 x 18 halo patch of the fp32 carrier ONCE, applies the
// GroupNorm scale / shift of its image (vface_groupnorm_coeffs_from_cols: gn_apply's arithmetic) and SiLU, rounds to 16 bits -- the
// same single rounding the separate pass made -- into LDS, and every thread accumulates its pixel's 9 x 64 x Cout products with
// v_dot2_f32_f16 (fp16 products are exact in fp32, fp32 accumulation: what the matrix cores compute, in another order) against weights
// that sit in scalar registers.  HBM traffic: the carrier once (126 MB) + 1.5 MB out.
#include <type_traits>

#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

constexpr int OC_T = 16;                 // output tile side
constexpr int OC_P = OC_T + 2;           // patch side
constexpr int OC_PITCH = 64 + 8;         // halfs per patch pixel in LDS (16 B of padding: tap-shifted rows spread over the banks)

template <class TT, bool IN32, int COUT>
__global__ __launch_bounds__(256) void gn_silu_conv3x3_small_kernel(OutConvParams p) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    __shared__ __attribute__((aligned(16))) E patch[OC_P * OC_P * OC_PITCH];
    __shared__ float sAB[2 * 64];
    const int t = threadIdx.x;
    const int tiles_x = (p.W + OC_T - 1) / OC_T, tiles_y = (p.H + OC_T - 1) / OC_T;
    const int img = blockIdx.x / (tiles_x * tiles_y), tl = blockIdx.x - img * (tiles_x * tiles_y);
    const int ty0 = (tl / tiles_x) * OC_T, tx0 = (tl % tiles_x) * OC_T;
    const int py = t >> 4, px = t & 15;                      // this thread's output pixel inside the tile
    const int oy = ty0 + py, ox = tx0 + px;
    const long img_row0 = (long)img * p.H * p.W;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    const unsigned* wq = reinterpret_cast<const unsigned*>(p.Wt);      // channel pairs
    const int nchunks = p.Cin / 64;
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();                                     // every thread is done with the previous chunk's patch
        if (t < 64) {
            const float2 ab = *reinterpret_cast<const float2*>(p.ab + ((long)img * p.ld_ab + c * 64 + t) * 2);
            sAB[t] = ab.x; sAB[64 + t] = ab.y;
        }
        __syncthreads();
        // ---- stage the activated patch: items = (patch pixel, 8-channel group)
        for (int it = t; it < OC_P * OC_P * 8; it += 256) {
            const int pp = it >> 3, g8 = it & 7;
            const int iy = ty0 - 1 + pp / OC_P, ix = tx0 - 1 + pp % OC_P;
            V8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (E)0.0f;      // zero padding applies to the ACTIVATED tensor
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                const long row = img_row0 + (long)iy * p.W + ix;
                float v[8];
                if (IN32) {
                    const float* xr = reinterpret_cast<const float*>(p.x) + row * p.ldx + c * 64 + g8 * 8;
                    const float4 a = *reinterpret_cast<const float4*>(xr), b = *reinterpret_cast<const float4*>(xr + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                } else {
                    const V8 xv = *reinterpret_cast<const V8*>(reinterpret_cast<const E*>(p.x) + row * p.ldx + c * 64 + g8 * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = to_f32(xv[j]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = v[j] * sAB[g8 * 8 + j] + sAB[64 + g8 * 8 + j];
                    o[j] = from_f32<E>(silu_f(f));
                }
            }
            *reinterpret_cast<V8*>(patch + pp * OC_PITCH + g8 * 8) = o;
        }
        __syncthreads();
        // ---- this pixel's 9 x 64 x COUT products; weights: [o][(chunk, tap, ch)] 16-bit, read as wave-uniform channel pairs
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const E* pr = patch + ((py + tap / 3) * OC_P + (px + tap % 3)) * OC_PITCH;
#pragma unroll
            for (int g8 = 0; g8 < 8; ++g8) {
                const V8 xv = *reinterpret_cast<const V8*>(pr + g8 * 8);
#pragma unroll
                for (int o = 0; o < COUT; ++o) {
                    const unsigned* wo = wq + ((long)o * 9 * p.Cin + (long)(c * 9 + tap) * 64 + g8 * 8) / 2;
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
    }
    if (oy < p.H && ox < p.W) {
        float* d = p.out + (img_row0 + (long)oy * p.W + ox) * p.ldo;
#pragma unroll
        for (int o = 0; o < COUT; ++o) d[o] = acc[o] + (p.bias ? p.bias[o] : 0.f);
    }
}

template <class TT>
int launch_t(const OutConvParams& p, hipStream_t stream) {
    const int tiles = ((p.W + OC_T - 1) / OC_T) * ((p.H + OC_T - 1) / OC_T);
    dim3 grid((unsigned)(p.nimg * tiles));
#define OC_LAUNCH(IN32_, CO_) hipLaunchKernelGGL((gn_silu_conv3x3_small_kernel<TT, IN32_, CO_>), grid, dim3(256), 0, stream, p)
    if (p.Cout == 4) { if (p.in_f32) OC_LAUNCH(true, 4); else OC_LAUNCH(false, 4); }
    else if (p.Cout == 3) { if (p.in_f32) OC_LAUNCH(true, 3); else OC_LAUNCH(false, 3); }
    else return VF_ERR_SHAPE;
#undef OC_LAUNCH
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

}  // namespace

bool vf_out_conv_supported(int Cin, int Cout) { return Cin > 0 && (Cin % 64) == 0 && (Cout == 3 || Cout == 4); }

int vf_launch_out_conv(const OutConvParams& p, int dtype, hipStream_t stream) {
    if (!p.x || !p.ab || !p.Wt || !p.out || p.nimg <= 0 || p.H <= 0 || p.W <= 0) return VF_ERR_ARG;
    if (!vf_out_conv_supported(p.Cin, p.Cout)) return VF_ERR_SHAPE;
    if ((p.ldx & (p.in_f32 ? 3 : 7)) || p.ld_ab < p.Cin || p.ldo < p.Cout) return VF_ERR_ALIGN;
    if (((uintptr_t)p.x | (uintptr_t)p.Wt) & 15 || ((uintptr_t)p.ab & 7)) return VF_ERR_ALIGN;
    if (dtype == VF_DTYPE_F16) return launch_t<F16>(p, stream);
    if (dtype == VF_DTYPE_BF16) return launch_t<BF16>(p, stream);
    return VF_ERR_DTYPE;
}
