// Pipelined MFMA GEMM / implicit-GEMM 3x3 convolution (gfx950): the same math, operand forms and epilogue as
// gemm.hip, with the K loop restructured around an LDS RING:
//
//   * STAGES (3..4) LDS stages; the global_load_lds of tile kt+STAGES-1 is issued while tile kt is consumed, so
//     (STAGES-2) whole tiles stay in flight ACROSS the per-tile barrier;
//   * the wait in front of the barrier is a COUNTED `s_waitcnt vmcnt(N)`, N = (STAGES-2) x (loads this wave
//     issues per tile) -- never 0 in the steady state -- and the barrier is a raw s_barrier (a __syncthreads()
//     would drain the LDS-DMA queue: guide, "Pipelining across barriers");
//   * one barrier per K tile: a stage is re-filled one full iteration after its last ds_read (every wave passes
//     the barrier of iteration kt only after finishing the MFMAs of kt-1, whose operands were that stage).
//
// Tile shapes are template parameters: WM x WN waves, each MT x NT mfma_f32_16x16x32 tiles, BK = 32 | 64.
//   big    : 8 waves (4x2), 256 x 128 x 64, 3 stages (144 KiB LDS, one workgroup per CU)   -- large-K convs / GEMMs
//   big160 : 8 waves (4x2), 256 x 160 x 64, 3 stages (156 KiB)                              -- N = 320 / 960
//   small  : 4 waves (2x2), 128 x 128 x 32, 4 stages (64 KiB, two workgroups per CU)        -- small-K GEMMs
#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

enum { MODE_PLAIN = 0, MODE_CONV_FAST = 1, MODE_CONV_GENERIC = 2 };

template <int N> __device__ __forceinline__ void wait_vmcnt_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vmcnt(int n) {  // n is wave-uniform
    switch (n) {
#define WC(i) case i: wait_vmcnt_imm<i>(); break;
        WC(0) WC(1) WC(2) WC(3) WC(4) WC(5) WC(6) WC(7) WC(8) WC(9) WC(10) WC(11) WC(12) WC(13) WC(14) WC(15) WC(16)
        WC(17) WC(18) WC(19) WC(20) WC(21) WC(22) WC(23) WC(24) WC(25) WC(26) WC(27) WC(28) WC(29) WC(30) WC(31) WC(32)
#undef WC
        default: wait_vmcnt_imm<0>();
    }
}

template <class TT, int MODE, int WM, int WN, int MT, int NT, int BKK, int STAGES>
__global__ __launch_bounds__(WM* WN * 64) void gemm_pipe_kernel(GemmParams p) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    constexpr int NTHR = WM * WN * 64, BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr int CPR = BKK / 8;                 // 16-B chunks per LDS row
    constexpr int RPR = NTHR / CPR;              // rows staged per round
    constexpr int AR = (BM + RPR - 1) / RPR, BR = (BN + RPR - 1) / RPR;
    constexpr int A_ELEMS = BM * BKK, B_ELEMS = BN * BKK, STAGE = A_ELEMS + B_ELEMS;
    constexpr int SWS = BKK == 64 ? 1 : 2, SWM = CPR - 1;  // slot ^= (row >> SWS) & SWM
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E* smem = reinterpret_cast<E*>(smem_raw);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const E* __restrict__ A = reinterpret_cast<const E*>(p.A);
    const E* __restrict__ A2 = reinterpret_cast<const E*>(p.A2);
    const E* __restrict__ Wt = reinterpret_cast<const E*>(p.Wt);
    const E* zeros = reinterpret_cast<const E*>(p.zeros);

    const int srow = t / CPR;
    const int schunk = (t % CPR) ^ ((srow >> SWS) & SWM);
    const int wave_row0 = (wave * 64) / CPR;     // first row this wave stages in round 0 (wave-uniform)

    const E* a_ptr[AR];
    const E* a2_ptr[AR];
    unsigned a_mask[AR];
    int g_oy[AR], g_ox[AR];
    long g_img[AR];
#pragma unroll
    for (int rr = 0; rr < AR; ++rr) {
        const int m = m0 + rr * RPR + srow;
        const bool ok = m < p.M && (rr * RPR + srow) < BM;
        a2_ptr[rr] = zeros;
        a_ptr[rr] = zeros;
        g_oy[rr] = g_ox[rr] = 0; g_img[rr] = 0;
        if (MODE == MODE_PLAIN) {
            a_ptr[rr] = A + (long)m * p.lda + schunk * 8;
            a_mask[rr] = ok ? 1u : 0u;
            if (p.A2) a2_ptr[rr] = A2 + (long)(p.a2_row_mod > 0 ? m % p.a2_row_mod : m) * p.lda2 + schunk * 8;
        } else {
            const int hw = p.OH * p.OW;
            const int img = m / hw;
            const int rem = m - img * hw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            const int y0 = oy * p.stride - p.pad, x0 = ox * p.stride - p.pad;
            if (MODE == MODE_CONV_FAST) {
                const int VH = p.upsample ? 2 * p.H : p.H, VW = p.upsample ? 2 * p.W : p.W;
                unsigned mk = 0;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int vy = y0 + tp / 3, vx = x0 + tp % 3;
                    if (ok && (unsigned)vy < (unsigned)VH && (unsigned)vx < (unsigned)VW) mk |= 1u << tp;
                }
                a_mask[rr] = mk;
                a_ptr[rr] = A + (((long)img * p.H + y0) * p.W + x0) * p.lda + schunk * 8;
                g_oy[rr] = y0; g_ox[rr] = x0; g_img[rr] = (long)img * p.H * p.W;
            } else {
                a_mask[rr] = ok ? 1u : 0u;
                g_oy[rr] = y0; g_ox[rr] = x0; g_img[rr] = (long)img * p.H * p.W;
            }
        }
    }
    const E* b_ptr[BR];
    bool b_ok[BR];
#pragma unroll
    for (int rr = 0; rr < BR; ++rr) {
        const int n = n0 + rr * RPR + srow;
        b_ok[rr] = n < p.N && (rr * RPR + srow) < BN;
        b_ptr[rr] = Wt + (long)n * p.ldw + schunk * 8;
    }
    // loads this wave issues per tile (rounds whose rows lie entirely beyond the tile are skipped wave-uniformly)
    int loads_per_tile = 0;
#pragma unroll
    for (int rr = 0; rr < AR; ++rr) loads_per_tile += (rr * RPR + wave_row0 < BM) ? 1 : 0;
#pragma unroll
    for (int rr = 0; rr < BR; ++rr) loads_per_tile += (rr * RPR + wave_row0 < BN) ? 1 : 0;

    auto stage = [&](int kt, int buf) {
        E* sA = smem + buf * STAGE;
        E* sB = sA + A_ELEMS;
        const int kbase = kt * BKK;
        const int k = kbase + schunk * 8;
        const bool kin = k < p.K;
        if (MODE == MODE_PLAIN) {
            const bool second = A2 && kbase >= p.K1;
            const long koff = second ? kbase - p.K1 : kbase;
#pragma unroll
            for (int rr = 0; rr < AR; ++rr) {
                if (rr * RPR + wave_row0 >= BM) continue;
                const E* src = (second ? a2_ptr[rr] : a_ptr[rr]) + koff;
                src = (a_mask[rr] && kin) ? src : zeros;
                __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sA + (rr * NTHR + wave * 64) * 8), 16, 0, 0);
            }
        } else if (MODE == MODE_CONV_FAST) {
            // (64-channel chunk, tap, channel) K order, see gemm.hip
            constexpr int TPC = 9 * (64 / BKK);           // K tiles per 64-channel chunk
            const int cc = kt / TPC, rem = kt - cc * TPC;
            const int tap = rem / (64 / BKK), sub = rem - tap * (64 / BKK);
            const int ky = tap / 3, kx = tap - ky * 3;
            const int ci0 = cc * 64 + sub * BKK;
            if (!p.upsample) {
                const long toff = ((long)ky * p.W + kx) * p.lda + ci0;
#pragma unroll
                for (int rr = 0; rr < AR; ++rr) {
                    if (rr * RPR + wave_row0 >= BM) continue;
                    const E* src = ((a_mask[rr] >> tap) & 1u) ? a_ptr[rr] + toff : zeros;
                    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sA + (rr * NTHR + wave * 64) * 8), 16, 0, 0);
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < AR; ++rr) {
                    if (rr * RPR + wave_row0 >= BM) continue;
                    const int sy = (g_oy[rr] + ky) >> 1, sx = (g_ox[rr] + kx) >> 1;
                    const E* src = ((a_mask[rr] >> tap) & 1u)
                                       ? A + (g_img[rr] + (long)sy * p.W + sx) * p.lda + ci0 + schunk * 8 : zeros;
                    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sA + (rr * NTHR + wave * 64) * 8), 16, 0, 0);
                }
            }
        } else {
            const int tap = k / p.Cin;
            const int ci = k - tap * p.Cin;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int VH = p.upsample ? 2 * p.H : p.H, VW = p.upsample ? 2 * p.W : p.W;
#pragma unroll
            for (int rr = 0; rr < AR; ++rr) {
                if (rr * RPR + wave_row0 >= BM) continue;
                const int vy = g_oy[rr] + ky, vx = g_ox[rr] + kx;
                const bool ok = a_mask[rr] && kin && (unsigned)vy < (unsigned)VH && (unsigned)vx < (unsigned)VW;
                const int sy = p.upsample ? (vy >> 1) : vy, sx = p.upsample ? (vx >> 1) : vx;
                const E* src = ok ? A + (g_img[rr] + (long)sy * p.W + sx) * p.lda + ci : zeros;
                __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sA + (rr * NTHR + wave * 64) * 8), 16, 0, 0);
            }
        }
        const bool kwin = k < p.Kw;
#pragma unroll
        for (int rr = 0; rr < BR; ++rr) {
            if (rr * RPR + wave_row0 >= BN) continue;
            const E* src = (b_ok[rr] && kwin) ? b_ptr[rr] + kbase : zeros;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sB + (rr * NTHR + wave * 64) * 8), 16, 0, 0);
        }
    };

    f4_t acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = f4_t{0.f, 0.f, 0.f, 0.f};

    const int nt = (p.K + BKK - 1) / BKK;
    const int fr = lane & 15, fq = lane >> 4;

    // all fragment reads of the tile are issued ahead of its MFMAs (register sets per k32 step)
    auto compute = [&](int buf) {
        const E* sA = smem + buf * STAGE;
        const E* sB = sA + A_ELEMS;
        constexpr int KS = BKK / 32;
        V8 af[KS][MT], bf[KS][NT];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = wm * (MT * 16) + i * 16 + fr;
                const int slot = (kk * 4 + fq) ^ ((row >> SWS) & SWM);
                af[kk][i] = *reinterpret_cast<const V8*>(sA + row * BKK + slot * 8);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int row = wn * (NT * 16) + j * 16 + fr;
                const int slot = (kk * 4 + fq) ^ ((row >> SWS) & SWM);
                bf[kk][j] = *reinterpret_cast<const V8*>(sB + row * BKK + slot * 8);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i) acc[j][i] = TT::mfma32(bf[kk][j], af[kk][i], acc[j][i]);
    };

    // ---- LDS ring: prologue fills STAGES-1 stages
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < nt) stage(s, s);
    const int steady = (STAGES - 2) * loads_per_tile;
    for (int kt = 0; kt < nt; ++kt) {
        // tile kt has landed when at most the (STAGES-2) younger tiles of THIS wave are still in flight
        if (kt + STAGES - 2 < nt) wait_vmcnt(steady); else wait_vmcnt_imm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + STAGES - 1 < nt) stage(kt + STAGES - 1, (kt + STAGES - 1) % STAGES);
        compute(kt % STAGES);
    }

    // ---- epilogue (identical to gemm.hip): lane holds 4 consecutive output channels nb.. of row m
    const float* bias = p.bias;
    const float* rowbias = p.rowbias;
    const E* res = reinterpret_cast<const E*>(p.residual);
    const bool geglu = p.flags & GEMM_GEGLU;
    const bool out32 = p.flags & GEMM_OUT_F32;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * (MT * 16) + i * 16 + fr;
        if (m >= p.M) continue;
        const float* rb = rowbias ? rowbias + (long)(m / p.rows_per_sample) * p.ld_rowbias : nullptr;
        if (!geglu) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int nb = n0 + wn * (NT * 16) + j * 16 + fq * 4;
                if (nb >= p.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[j][i][r];
                if (bias) {
                    const float4 b = *reinterpret_cast<const float4*>(bias + nb);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                if (rb) {
                    const float4 b = *reinterpret_cast<const float4*>(rb + nb);
                    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                }
                if (res) {
                    const typename TT::v4 r4 = *reinterpret_cast<const typename TT::v4*>(res + (long)m * p.ldr + nb);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += to_f32(r4[r]);
                }
                if (out32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + nb) =
                        make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    typename TT::v4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = from_f32<E>(v[r]);
                    *reinterpret_cast<typename TT::v4*>(reinterpret_cast<E*>(p.C) + (long)m * p.ldc + nb) = o;
                }
            }
        } else if constexpr ((NT & 1) == 0) {
            // value / gate 16-row tiles alternate (packing.pack_geglu); NT even => each wave starts on a value tile
#pragma unroll
            for (int jj = 0; jj < NT / 2; ++jj) {
                const int nb = n0 + wn * (NT * 16) + jj * 32 + fq * 4;
                if (nb >= p.N) continue;
                const int oc = (nb >> 5) * 16 + (nb & 15);
                float a[4], g[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { a[r] = acc[2 * jj][i][r]; g[r] = acc[2 * jj + 1][i][r]; }
                if (bias) {
                    const float4 ba = *reinterpret_cast<const float4*>(bias + nb);
                    const float4 bg = *reinterpret_cast<const float4*>(bias + nb + 16);
                    a[0] += ba.x; a[1] += ba.y; a[2] += ba.z; a[3] += ba.w;
                    g[0] += bg.x; g[1] += bg.y; g[2] += bg.z; g[3] += bg.w;
                }
                typename TT::v4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = from_f32<E>(a[r] * gelu_erf_f(g[r]));
                *reinterpret_cast<typename TT::v4*>(reinterpret_cast<E*>(p.C) + (long)m * p.ldc + oc) = o;
            }
        }
    }
}

template <class TT, int MODE, int WM, int WN, int MT, int NT, int BKK, int STAGES>
int launch_cfg(const GemmParams& p, hipStream_t stream) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    constexpr size_t lds = (size_t)STAGES * (BM + BN) * BKK * sizeof(typename TT::elem);
    static_assert(lds <= 160 * 1024, "LDS ring exceeds 160 KiB");
    auto kern = gemm_pipe_kernel<TT, MODE, WM, WN, MT, NT, BKK, STAGES>;
    static bool attr_set = false;
    if (lds > 64 * 1024 && !attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return VF_ERR_LAUNCH;
        attr_set = true;
    }
    dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, stream, p);
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

template <class TT, int MODE>
int launch_mode(const GemmParams& p, int variant, hipStream_t stream) {
    switch (variant) {
        case 1: return launch_cfg<TT, MODE, 4, 2, 4, 4, 64, 3>(p, stream);   // big: 256x128x64, 3 stages
        case 2: return launch_cfg<TT, MODE, 4, 2, 4, 5, 64, 3>(p, stream);   // big160: 256x160x64, 3 stages
        case 3: return launch_cfg<TT, MODE, 2, 2, 4, 4, 32, 2>(p, stream);   // 128x128x32, 2 stages (3 wg/CU)
        case 4: return launch_cfg<TT, MODE, 2, 2, 4, 5, 32, 2>(p, stream);   // 128x160x32, 2 stages (3 wg/CU)
        default: return VF_ERR_ARG;
    }
}

}  // namespace

int vf_launch_gemm_pipe(const GemmParams& p, int dtype, int variant, hipStream_t stream) {
    if ((p.flags & GEMM_GEGLU) && (variant == 2 || variant == 4)) return VF_ERR_SHAPE;
    const int mode = p.mode == 0 ? MODE_PLAIN : ((p.Cin % 64 == 0) ? MODE_CONV_FAST : MODE_CONV_GENERIC);
#define GO(TT)                                                                   \
    switch (mode) {                                                              \
        case MODE_PLAIN: return launch_mode<TT, MODE_PLAIN>(p, variant, stream); \
        case MODE_CONV_FAST: return launch_mode<TT, MODE_CONV_FAST>(p, variant, stream); \
        default: return launch_mode<TT, MODE_CONV_GENERIC>(p, variant, stream);  \
    }
    if (dtype == VF_DTYPE_F16) { GO(F16) }
    if (dtype == VF_DTYPE_BF16) { GO(BF16) }
#undef GO
    return VF_ERR_DTYPE;
}
