// "Ping-pong" MFMA GEMM / implicit-GEMM 3x3 convolution (gfx950).  Same math, operand forms and epilogue as gemm.hip.
//
// One workgroup = 8 waves = two GROUPS of four.  Each group owns a 128 x BN output tile (two vertically adjacent
// m-tiles of the same n-tile, so the weight tile is loaded once for both).  The groups run the K loop half a tile
// out of phase, separated by workgroup barriers:
//
//     phase A(kt):  group 0: 8 x NT x 2 MFMAs on tile kt (fragments already in registers)
//                   group 1: ds_read its fragments of tile kt, then issue its half of the loads of tile kt+2
//     phase B(kt):  group 1: MFMAs on tile kt
//                   group 0: ds_read its fragments of tile kt+1, issue its half of the loads of tile kt+2
//
// A SIMD hosts one wave of each group, so at any time it has one wave streaming MFMAs back to back and one wave doing
// LDS reads / address arithmetic / buffer loads: the matrix pipe never waits for an LDS round trip or a barrier that
// its own wave caused (the overlap two independent workgroups per CU only get by luck).  Loads run two tiles ahead
// through a 3-stage LDS ring (144 KiB at BN = 128, 156 KiB at BN = 160): a stage is refilled two phases after its
// last reader, and read one barrier after every loader's `s_waitcnt vmcnt(0)` (which sits at the END of that wave's
// MFMA phase, i.e. ~500 cycles after issue).
#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

constexpr int BMG = 128, BM2 = 256, BK = 64;
enum { MODE_PLAIN = 0, MODE_CONV_FAST = 1, MODE_CONV_GENERIC = 2 };

template <class TT, int MODE, int NT>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(GemmParams p) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    constexpr int BN = 32 * NT;                 // 128 | 160
    constexpr int BH = BN / 2;                  // weight rows each group stages (64 | 80)
    constexpr int BR = (BH + 31) / 32;          // rounds of 32 rows
    constexpr int A_ELEMS = BM2 * BK, B_ELEMS = BN * BK, STAGE = A_ELEMS + B_ELEMS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E* smem = reinterpret_cast<E*>(smem_raw);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wm = w4 >> 1, wn = w4 & 1;
    const int tg = t & 255;

    int m0, n0;
    {
        const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM2 - 1) / BM2;
        const int nwg = gridDim.x, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, loc = id >> 3;
        int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        if ((p.flags & GEMM_NO_XCD_REMAP) || MODE != MODE_PLAIN) L = id;
        constexpr int GN = 8;
        const int g = L / (GN * ntm);
        const int rem = L - g * (GN * ntm);
        const int gw = min(GN, ntn - g * GN);
        const int tm = rem / gw;
        m0 = tm * BM2;
        n0 = (g * GN + (rem - tm * gw)) * BN;
    }

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A2 ? p.A2 : p.A), 0, (int)(p.A2 ? p.a2_bytes : 0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wt), 0, (int)p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    constexpr unsigned ES = sizeof(E);

    // staging map inside a group: slot = rr*256 + tg -> row rr*32 + (tg>>3), 16-B slot tg&7 holds k-chunk schunk
    const int srow = tg >> 3;
    const int schunk = (tg & 7) ^ ((tg >> 4) & 7);

    unsigned a_off[4], a2_off[4], a_mask[4], g_img[4];
    int g_oy[4], g_ox[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int m = m0 + grp * BMG + rr * 32 + srow;   // a group stages the A rows of its own m-tile
        const bool ok = m < p.M;
        a2_off[rr] = OOB; a_off[rr] = OOB;
        g_oy[rr] = g_ox[rr] = 0; g_img[rr] = 0;
        if (MODE == MODE_PLAIN) {
            a_off[rr] = ok ? (unsigned)(((long)m * p.lda + schunk * 8) * ES) : OOB;
            a_mask[rr] = ok ? 1u : 0u;
            if (p.A2 && ok) a2_off[rr] = (unsigned)(((long)(p.a2_row_mod > 0 ? m % p.a2_row_mod : m) * p.lda2 + schunk * 8) * ES);
        } else {
            const int hw = p.OH * p.OW;
            const int img = m / hw;
            const int rem = m - img * hw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
            const int y0 = oy * p.stride - p.pad, x0 = ox * p.stride - p.pad;
            g_oy[rr] = y0; g_ox[rr] = x0; g_img[rr] = (unsigned)(img * p.H * p.W);
            if (MODE == MODE_CONV_FAST) {
                const int VH = p.upsample ? 2 * p.H : p.H, VW = p.upsample ? 2 * p.W : p.W;
                unsigned mk = 0;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int vy = y0 + tp / 3, vx = x0 + tp % 3;
                    if (ok && (unsigned)vy < (unsigned)VH && (unsigned)vx < (unsigned)VW) mk |= 1u << tp;
                }
                a_mask[rr] = mk;
                a_off[rr] = (unsigned)(((((long)img * p.H + y0) * p.W + x0) * p.lda + schunk * 8) * ES);
            } else {
                a_mask[rr] = ok ? 1u : 0u;
                a_off[rr] = 0;
            }
        }
    }
    unsigned b_off[BR];
    bool b_act[BR];  // wave-uniform: this wave stages rows in round rr
#pragma unroll
    for (int rr = 0; rr < BR; ++rr) {
        const int rloc = rr * 32 + srow;                 // row inside this group's half of the weight tile
        b_act[rr] = (rr * 32 + w4 * 8) < BH;
        const int n = n0 + grp * BH + rloc;
        b_off[rr] = (rloc < BH && n < p.N) ? (unsigned)(((long)n * p.ldw + schunk * 8) * ES) : OOB;
    }

    // this group's share of the loads of K tile kt into ring stage `buf`
    auto stage = [&](int kt, int buf) {
        E* sA = smem + buf * STAGE + grp * (BMG * BK);
        E* sB = smem + buf * STAGE + A_ELEMS + grp * (BH * BK);
        const int kbase = kt * BK;
        const int k = kbase + schunk * 8;
        if (MODE == MODE_PLAIN) {
            const bool second = p.A2 && kbase >= p.K1;
            const unsigned koff = (unsigned)(second ? kbase - p.K1 : kbase) * ES;
            const bool kin = k < p.K;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const unsigned base = second ? a2_off[rr] : a_off[rr];
                const unsigned off = (kin && base != OOB) ? base + koff : OOB;
                if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA2, LDS_PTR(sA + (rr * 256 + w4 * 64) * 8), 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sA + (rr * 256 + w4 * 64) * 8), 16, off, 0, 0, 0);
            }
        } else if (MODE == MODE_CONV_FAST) {
            const int cc = kt / 9, tap = kt - cc * 9;    // (64-channel chunk, tap, channel) K order, see gemm.hip
            const int ky = tap / 3, kx = tap - ky * 3;
            const int ci0 = cc * BK;
            if (!p.upsample) {
                const unsigned toff = (unsigned)((((long)ky * p.W + kx) * p.lda + ci0) * ES);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const unsigned off = ((a_mask[rr] >> tap) & 1u) ? a_off[rr] + toff : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sA + (rr * 256 + w4 * 64) * 8), 16, off, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int sy = (g_oy[rr] + ky) >> 1, sx = (g_ox[rr] + kx) >> 1;
                    const unsigned off = ((a_mask[rr] >> tap) & 1u)
                                             ? (unsigned)((((long)g_img[rr] + (long)sy * p.W + sx) * p.lda + ci0 + schunk * 8) * ES) : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sA + (rr * 256 + w4 * 64) * 8), 16, off, 0, 0, 0);
                }
            }
        } else {
            const int tap = k / p.Cin;
            const int ci = k - tap * p.Cin;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int VH = p.upsample ? 2 * p.H : p.H, VW = p.upsample ? 2 * p.W : p.W;
            const bool kin = k < p.K;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int vy = g_oy[rr] + ky, vx = g_ox[rr] + kx;
                const bool ok = a_mask[rr] && kin && (unsigned)vy < (unsigned)VH && (unsigned)vx < (unsigned)VW;
                const int sy = p.upsample ? (vy >> 1) : vy, sx = p.upsample ? (vx >> 1) : vx;
                const unsigned off = ok ? (unsigned)((((long)g_img[rr] + (long)sy * p.W + sx) * p.lda + ci) * ES) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sA + (rr * 256 + w4 * 64) * 8), 16, off, 0, 0, 0);
            }
        }
        const bool kwin = k < p.Kw;
        const unsigned kboff = (unsigned)kbase * ES;
#pragma unroll
        for (int rr = 0; rr < BR; ++rr) {
            if (!b_act[rr]) continue;
            const unsigned off = (kwin && b_off[rr] != OOB) ? b_off[rr] + kboff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(sB + (rr * 256 + w4 * 64) * 8), 16, off, 0, 0, 0);
        }
    };

    f4_t acc[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f4_t{0.f, 0.f, 0.f, 0.f};
    V8 af[2][4], bf[2][NT];

    const int nt = (p.K + BK - 1) / BK;
    const int fr = lane & 15, fq = lane >> 4;

    auto read_frags = [&](int buf) {
        const E* sA = smem + buf * STAGE + grp * (BMG * BK);
        const E* sB = smem + buf * STAGE + A_ELEMS;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + fr;
                const int slot = (kk * 4 + fq) ^ ((row >> 1) & 7);
                af[kk][i] = *reinterpret_cast<const V8*>(sA + row * BK + slot * 8);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int row = wn * (BN / 2) + j * 16 + fr;
                const int slot = (kk * 4 + fq) ^ ((row >> 1) & 7);
                bf[kk][j] = *reinterpret_cast<const V8*>(sB + row * BK + slot * 8);
            }
        }
    };
    auto mfmas = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = TT::mfma32(bf[kk][j], af[kk][i], acc[j][i]);
        __builtin_amdgcn_s_setprio(0);
    };
    // `drain`: the group that just finished its MFMA phase waits for the LDS-DMA it issued one phase earlier
    // (~one MFMA phase ago, so normally already landed); the group that just ISSUED loads must not wait for them.
    auto phase_end = [&](bool drain) {
        if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- prologue: tiles 0 and 1 in flight, group 0 pre-reads tile 0
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    phase_end(true);
    if (grp == 0) read_frags(0);

    for (int kt = 0; kt < nt; ++kt) {
        // phase A: group 0 computes tile kt; group 1 fetches its fragments of tile kt and prefetches tile kt+2
        if (grp == 0) {
            mfmas();
        } else {
            read_frags(kt % 3);
            if (kt + 2 < nt) stage(kt + 2, (kt + 2) % 3);
        }
        phase_end(grp == 0);
        // phase B: group 1 computes tile kt; group 0 fetches its fragments of tile kt+1 and prefetches tile kt+2
        if (grp == 1) {
            mfmas();
        } else {
            if (kt + 1 < nt) read_frags((kt + 1) % 3);
            if (kt + 2 < nt) stage(kt + 2, (kt + 2) % 3);
        }
        phase_end(grp == 1);
    }

    // ---- epilogue (as gemm.hip): lane holds 4 consecutive output channels nb.. of row m
    const float* bias = p.bias;
    const float* rowbias = p.rowbias;
    const E* res = reinterpret_cast<const E*>(p.residual);
    const bool geglu = p.flags & GEMM_GEGLU;
    const bool out32 = p.flags & GEMM_OUT_F32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + grp * BMG + wm * 64 + i * 16 + fr;
        if (m >= p.M) continue;
        const float* rb = rowbias ? rowbias + (long)(m / p.rows_per_sample) * p.ld_rowbias : nullptr;
        if (!geglu) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int nb = n0 + wn * (BN / 2) + j * 16 + fq * 4;
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
#pragma unroll
            for (int jj = 0; jj < NT / 2; ++jj) {
                const int nb = n0 + wn * (BN / 2) + jj * 32 + fq * 4;
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

template <class TT, int MODE, int NT>
int launch_pp(const GemmParams& p, hipStream_t stream) {
    constexpr int BN = 32 * NT;
    constexpr size_t lds = (size_t)3 * (BM2 + BN) * BK * sizeof(typename TT::elem);
    static_assert(lds <= 160 * 1024, "LDS ring exceeds 160 KiB");
    auto kern = gemm_pp_kernel<TT, MODE, NT>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return VF_ERR_LAUNCH;
        attr_set = true;
    }
    dim3 grid(((p.M + BM2 - 1) / BM2) * ((p.N + BN - 1) / BN));
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, stream, p);
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

}  // namespace

// variant 9: BN = 128, variant 10: BN = 160
int vf_launch_gemm_pp(const GemmParams& p, int dtype, int variant, hipStream_t stream) {
    if ((p.flags & GEMM_GEGLU) && variant == 10) return VF_ERR_SHAPE;
    const int mode = p.mode == 0 ? MODE_PLAIN : ((p.Cin % 64 == 0) ? MODE_CONV_FAST : MODE_CONV_GENERIC);
#define GO(TT, NT)                                                      \
    switch (mode) {                                                     \
        case MODE_PLAIN: return launch_pp<TT, MODE_PLAIN, NT>(p, stream); \
        case MODE_CONV_FAST: return launch_pp<TT, MODE_CONV_FAST, NT>(p, stream); \
        default: return launch_pp<TT, MODE_CONV_GENERIC, NT>(p, stream); \
    }
    if (dtype == VF_DTYPE_F16) { if (variant == 10) { GO(F16, 5) } else { GO(F16, 4) } }
    if (dtype == VF_DTYPE_BF16) { if (variant == 10) { GO(BF16, 5) } else { GO(BF16, 4) } }
#undef GO
    return VF_ERR_DTYPE;
}
