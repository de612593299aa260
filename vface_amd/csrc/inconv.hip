// The UNet's INPUT convolution (gfx950): conv3x3 over a map of 16 stored channels (the 9 = 4 + 4 + 1 input channels of the inpainting
// UNet, zero-padded), 16 -> Cout, stride 1, padding 1  (REFace/ldm/modules/diffusionmodules/openaimodel.py:639-645
// `TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))`, applied first at :882-885) -- the mirror of
// outconv.hip at the other end of the network.
//
// K = 9 taps x 16 channels = 144: ONE pass of five k32 MFMA steps (the last one half zeros), no K loop to pipeline.  Through the
// generic-window implicit GEMM (gemm.hip, Cin % 64 != 0) this launch staged 64-deep K tiles of which 2.25 were real, and ran at twice
// its HBM bound (285 us for 96 images at 64 x 64: 755 MB of output, profiles/r05_j_step_trace).  It is bound by its OUTPUT -- 6 bytes
// per element (fp32 carrier + 16-bit copy) against 32 bytes of input per pixel -- so this kernel is built around the stores:
//
// * a WAVE owns 80 output channels and walks 64-pixel slices (= the column-statistics slices); its 5 x 5 weight fragments (100
//   registers) are loaded once and stay in registers: no operand goes through LDS, no barrier anywhere in the kernel;
// * per slice: four 16-pixel tiles; the activation fragment of (tile, k32 step) is ONE 16-byte load per lane straight from the
//   map -- lane (pixel fr, group fq) reads channels 8 (fq & 1) .. +7 of the pixel shifted by tap 2 s + (fq >> 1); a tap outside
//   the image (the zero padding) or tap 9 is an out-of-range buffer offset (zeros).  The 32 bytes of a pixel are re-read by its
//   nine taps and by the four channel slices of the workgroup out of L1 / L2;
// * epilogue: tile by tile the 16 x 80 fp32 sums cross the wave's own 5 KB of LDS and leave as whole rows (16 bytes per lane, 320
//   contiguous bytes of the fp32 carrier and 160 of the 16-bit copy per pixel); the per-slice column sums (of the fp32 values
//   stored, as every producer's) are read back from that scratch, one channel per lane, row by row: a fixed order.
#include <type_traits>

#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

constexpr int IC_CH = 80;           // output channels per wave (five 16-channel tiles)
constexpr int IC_K = 144;           // 9 taps x 16 stored channels
constexpr int IC_SP = 84;           // scratch row pitch (floats): 80 + 4, the accumulator tiles' float4 writes spread over the banks

template <class TT>
__global__ __launch_bounds__(256, 2) void conv3x3_c16_kernel(GemmParams p, int slices_per_wg, int nslices) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    __shared__ __attribute__((aligned(16))) float smem[4 * 16 * IC_SP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* const scr = smem + wave * (16 * IC_SP);
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = (blockIdx.y * 4 + wave) * IC_CH;
    if (n0 >= p.N) return;                                    // (no barrier in this kernel: a wave may leave alone)
    constexpr unsigned OOB = 0xFFFF0000u;                     // (the launcher keeps every view below this; + an immediate does not wrap)
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C ? p.C : const_cast<void*>(p.Wt), 0, (int)(p.C ? p.c_bytes : 0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rC32 = __builtin_amdgcn_make_buffer_rsrc(p.C32 ? (void*)p.C32 : const_cast<void*>(p.Wt), 0, (int)(p.C32 ? p.c32_bytes : 0u), 0x00020000);

    // ---- weights: A operand of step s, channel tile j = row n0 + 16 j + fr, k = 32 s + 8 fq .. + 7 (k >= 144: zeros)
    V8 wf[5][5];
    const E* Wt = reinterpret_cast<const E*>(p.Wt);
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            V8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (E)0.0f;
            const int k = s * 32 + fq * 8;
            wf[j][s] = k < IC_K ? *reinterpret_cast<const V8*>(Wt + (long)(n0 + j * 16 + fr) * p.ldw + k) : z;
        }
    f4_t bj[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
        bj[j] = p.bias ? *reinterpret_cast<const f4_t*>(p.bias + n0 + j * 16 + fq * 4) : f4_t{0.f, 0.f, 0.f, 0.f};
    // ---- this lane's tap per step: tap = 2 s + (fq >> 1) -> (dy, dx) in {-1, 0, 1}; step 4's upper half is tap 9 = nothing
    int dy[5], dx[5];
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int tap = 2 * s + (fq >> 1);
        dy[s] = tap < 9 ? tap / 3 - 1 : 1 << 20;              // (tap 9: a row far outside every image)
        dx[s] = tap % 3 - 1;
    }
    const int ch8 = (fq & 1) * 8;
    const int hw = p.H * p.W, spi = hw >> 6;

    const int sl_end = min(nslices, (int)(blockIdx.x + 1) * slices_per_wg);
    for (int sl = blockIdx.x * slices_per_wg; sl < sl_end; ++sl) {
        const int img = sl / spi, rem0 = (sl - img * spi) << 6;      // the slice lies inside one image (hw % 64 == 0)
        V8 bf[2][5];
        auto load_tile = [&](int i) {                          // the five activation fragments of pixel tile i (W % 16 == 0: one image row)
            const int r = rem0 + i * 16;
            const int y = r / p.W, x = r - y * p.W + fr;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const int yy = y + dy[s], xx = x + dx[s];
                const bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
                const unsigned off = ok ? (unsigned)(((((long)img * p.H + yy) * p.W + xx) * p.lda + ch8) * 2) : OOB;
                bf[i & 1][s] = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)off, 0, 0));
            }
        };
        // column sums of the slice: lane l sums channel l (lanes 0..15 also channel 64 + l) down the 16 rows of every tile in the
        // scratch -- row by row, tile by tile: a fixed order
        float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
        load_tile(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // one 16-pixel tile at a time, start to finish (there is no K loop to keep accumulators for: 20 registers, not 80)
            if (i + 1 < 4) load_tile(i + 1);
            f4_t acc[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[j] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 5; ++s)
#pragma unroll
                for (int j = 0; j < 5; ++j) acc[j] = TT::mfma32(wf[j][s], bf[i & 1][s], acc[j]);
            // ---- epilogue: + bias; lane (fr, fq) holds channels n0 + 16 j + 4 fq .. + 3 of pixel 64 sl + 16 i + fr.  The 16 x 80 fp32
            // values cross the wave's own LDS scratch (no barrier: LDS operations of one wave execute in order) and leave as whole
            // rows: 320 contiguous bytes of the carrier per pixel (20 lanes x 16 B), 160 of the 16-bit copy (10 lanes x 16 B) -- stores
            // straight from the accumulator layout (64- / 32-byte pieces of a row per instruction) ran at 3.0 TB/s (profiles/r05_k)
#pragma unroll
            for (int j = 0; j < 5; ++j) *reinterpret_cast<f4_t*>(scr + fr * IC_SP + j * 16 + fq * 4) = acc[j] + bj[j];
            if (p.colstats) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = scr[r * IC_SP + lane];
                    s0 += v; q0 = fmaf(v, v, q0);
                }
                if (lane < 16) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = scr[r * IC_SP + 64 + lane];
                        s1 += v; q1 = fmaf(v, v, q1);
                    }
                }
            }
            const long mt = (long)sl * 64 + i * 16;
            if (p.C32) {
#pragma unroll
                for (int it = 0; it < 5; ++it) {
                    const int idx = lane + 64 * it, r = idx / 20, c = idx - r * 20;
                    const f4_t x = *reinterpret_cast<const f4_t*>(scr + r * IC_SP + c * 4);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, x), rC32, (int)(unsigned)(((mt + r) * p.ldc32 + n0 + c * 4) * 4), 0, 0);
                }
            }
            if (p.C) {
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    const int idx = lane + 64 * it, r = idx / 10, c = idx - r * 10;
                    if (idx < 160) {
                        const f4_t x0 = *reinterpret_cast<const f4_t*>(scr + r * IC_SP + c * 8), x1 = *reinterpret_cast<const f4_t*>(scr + r * IC_SP + c * 8 + 4);
                        const V8 o = V8{from_f32<E>(x0[0]), from_f32<E>(x0[1]), from_f32<E>(x0[2]), from_f32<E>(x0[3]),
                                        from_f32<E>(x1[0]), from_f32<E>(x1[1]), from_f32<E>(x1[2]), from_f32<E>(x1[3])};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, o), rC, (int)(unsigned)(((mt + r) * p.ldc + n0 + c * 8) * 2), 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // (tile by tile: interleaving two tiles' epilogues costs registers the weights occupy)
        }
        if (p.colstats) {
            float* d = p.colstats + ((long)sl * p.ld_colstats + n0) * 2;
            *reinterpret_cast<float2*>(d + lane * 2) = make_float2(s0, q0);
            if (lane < 16) *reinterpret_cast<float2*>(d + (64 + lane) * 2) = make_float2(s1, q1);
        }
    }
}

}  // namespace

// Is this convolution launch the 16-stored-channel input form this kernel takes?  (vf_launch_gemm has validated alignments and filled
// the operand extents.)  Per-sample geometry and epilogue form only: a sample's bits do not depend on its batch.
bool vf_conv_in16_ok(const GemmParams& p) {
    if (p.mode != 1 || p.Cin != 16 || p.ntaps != 9 || p.KH != 3 || p.KW != 3 || p.stride != 1 || p.upsample || p.pad != 1 || p.pad_x != 1) return false;
    if (p.A2 || p.residual || p.rowbias || p.gn_ab || p.out_phase || p.res_f32) return false;
    if (p.flags & (GEMM_GEGLU | GEMM_OUT_F32 | GEMM_NO_PATCH | 0x4000 | (0xF << 8))) return false;      // (GEMM_NO_PATCH: A/B, the generic kernel)
    if ((p.N % IC_CH) || p.K != IC_K || p.Kw < IC_K || (p.ldw & 7)) return false;
    if (p.OH != p.H || p.OW != p.W || ((p.H * p.W) & 63) || (p.W & 15) || p.M != (p.M / (p.H * p.W)) * p.H * p.W) return false;
    if (!p.C && !p.C32) return false;
    if (p.C && (((uintptr_t)p.C & 15) || (p.ldc & 7))) return false;
    if (p.C32 && (((uintptr_t)p.C32 & 15) || (p.ldc32 & 3))) return false;
    if (p.colstats && (((uintptr_t)p.colstats & 15) || (p.ld_colstats & 1))) return false;
    if (p.bias && ((uintptr_t)p.bias & 15)) return false;
    const unsigned long cb = p.C ? ((unsigned long)(p.M - 1) * p.ldc + p.N) * 2ul : 0ul, c32b = p.C32 ? ((unsigned long)(p.M - 1) * p.ldc32 + p.N) * 4ul : 0ul;
    return cb < 0xFFFF0000ul && c32b < 0xFFFF0000ul && p.a_bytes < 0xFFFF0000u;
}

int vf_launch_conv_in16(const GemmParams& p_in, int dtype, hipStream_t stream) {
    if (!vf_conv_in16_ok(p_in)) return VF_ERR_SHAPE;
    GemmParams p = p_in;
    p.c_bytes = p.C ? (unsigned)(((unsigned long)(p.M - 1) * p.ldc + p.N) * 2ul) : 0u;
    p.c32_bytes = p.C32 ? (unsigned)(((unsigned long)(p.M - 1) * p.ldc32 + p.N) * 4ul) : 0u;
    const int nslices = p.M / 64;
    // a workgroup = the (up to) four 80-channel slices of a run of pixel slices; about two workgroups per CU in flight, each wave
    // keeping its weight fragments over its whole run (at least 4 slices where the launch has them: the fragments are 26 KB)
    int spw = (nslices + 511) / 512;
    if (spw < 4) spw = nslices < 4 ? nslices : 4;
    const dim3 grid((nslices + spw - 1) / spw, (p.N / IC_CH + 3) / 4);
    if (dtype == VF_DTYPE_F16) hipLaunchKernelGGL(conv3x3_c16_kernel<F16>, grid, dim3(256), 0, stream, p, spw, nslices);
    else if (dtype == VF_DTYPE_BF16) hipLaunchKernelGGL(conv3x3_c16_kernel<BF16>, grid, dim3(256), 0, stream, p, spw, nslices);
    else return VF_ERR_DTYPE;
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}
