// MFMA GEMM / implicit-GEMM 3x3 convolution for the VFace UNet (gfx950).
//
//   C[m, n] = epilogue( sum_k A(m, k) * Wt[n, k] )
//
// * Wt is the nn.Linear / packed conv weight: [N][Kp] row-major, K contiguous (the natural "B^T" form).
// * A is either a plain row-major [M][K] matrix with leading dimension lda (tokens x channels, NHWC
//   activations are exactly this), or -- MODE_CONV -- the im2col view of an NHWC image
//   [img][H][W][ldpix >= Cin] under a 3x3 window with padding 1, stride 1|2 and optional nearest x2
//   upsampling of the input (openaimodel.py Downsample :151-153, Upsample :116-118, ResBlock convs
//   :201-205,225-232), k = (ky*3 + kx)*Cin + ci.  Nothing is materialised: each 16-byte k-chunk of the
//   A tile is fetched straight from its source pixel by a direct global->LDS load; padding taps read a
//   zero page.
// * Tile 128(m) x 128(n) x 64(k), 4 waves (2x2), each wave 64x64 = 4x4 mfma_f32_16x16x32 tiles, fp32
//   accumulate.  Two LDS buffers; the next K tile's global_load_lds (16 B/lane) is in flight while the
//   current one feeds the MFMAs.  LDS rows are 128 B; the 16-B slot of k-chunk c of row r is
//   c ^ ((r>>1)&7), applied on the per-lane SOURCE address (the LDS image of a wave-instruction is
//   lane-linear) and again on the fragment read: ds_read_b128 of 16 rows x one chunk is conflict-free.
// * Weights are the MFMA "A" operand and activations the "B" operand, so a lane ends up with 4
//   consecutive output channels of one row: 8-byte stores, and bias / residual / GEGLU pair up in-lane.
//
// Epilogue (all fp32): + bias[n] + rowbias[m / rows_per_sample][n] (time-embedding add of ResBlock
// :264-271, or the degenerate single-token cross-attention vector), optional GEGLU
// (attention.py:37-45, weight rows pre-permuted so value/gate tiles alternate), + residual[m][n],
// store as fp16/bf16 or fp32.
#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_ELEMS = BM * BK;  // 8192 elems = 16 KiB

template <class TT, int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E* smem = reinterpret_cast<E*>(smem_raw);  // [2][A tile | B tile]

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const E* __restrict__ A = reinterpret_cast<const E*>(p.A);
    const E* __restrict__ A2 = reinterpret_cast<const E*>(p.A2);
    const E* __restrict__ Wt = reinterpret_cast<const E*>(p.Wt);
    const E* zeros = reinterpret_cast<const E*>(p.zeros);

    // ---- staging map: slot = rr*256 + t -> row rr*32 + (t>>3), 16-B slot t&7, logical k-chunk below
    const int srow = t >> 3;
    const int schunk = (t & 7) ^ ((t >> 4) & 7);

    // per-round source row state
    long a_row_off[4], a2_row_off[4];
    int a_oy[4], a_ox[4];
    bool a_ok[4];
    long b_row_off[4];
    bool b_ok[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int m = m0 + rr * 32 + srow;
        a_ok[rr] = m < p.M;
        a2_row_off[rr] = 0;
        if (MODE == 0) {
            a_row_off[rr] = (long)m * p.lda;
            a_oy[rr] = a_ox[rr] = 0;
            if (p.A2) a2_row_off[rr] = (long)(p.a2_row_mod > 0 ? m % p.a2_row_mod : m) * p.lda2;
        } else {
            const int hw = p.OH * p.OW;
            const int img = m / hw;
            const int rem = m - img * hw;
            const int oy = rem / p.OW;
            a_oy[rr] = oy * p.stride - 1;
            a_ox[rr] = (rem - oy * p.OW) * p.stride - 1;
            a_row_off[rr] = (long)img * p.H * p.W;
        }
        const int n = n0 + rr * 32 + srow;
        b_ok[rr] = n < p.N;
        b_row_off[rr] = (long)n * p.ldw;
    }

    auto stage = [&](int kt, int buf) {
        const int k = kt * BK + schunk * 8;
        E* sA = smem + buf * 2 * TILE_ELEMS;
        E* sB = sA + TILE_ELEMS;
        const bool kin = k < p.K;
        int ky = 0, kx = 0, ci = k;
        if (MODE == 1) {
            const int tap = k / p.Cin;
            ci = k - tap * p.Cin;
            ky = tap / 3;
            kx = tap - ky * 3;
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const E* src;
            if (MODE == 0) {
                // dual-source K: columns [0, K1) come from A, [K1, K) from A2 (K1 is a multiple of BK,
                // so a K tile never straddles).  This is how [own | structure] feeds the folded FSAI
                // projection without concatenating anything.
                if (A2 && kt * BK >= p.K1) src = (a_ok[rr] && kin) ? A2 + a2_row_off[rr] + (k - p.K1) : zeros;
                else src = (a_ok[rr] && kin) ? A + a_row_off[rr] + k : zeros;
            } else {
                const int vy = a_oy[rr] + ky, vx = a_ox[rr] + kx;
                const int VH = p.upsample ? 2 * p.H : p.H, VW = p.upsample ? 2 * p.W : p.W;
                const bool ok = a_ok[rr] && kin && (unsigned)vy < (unsigned)VH && (unsigned)vx < (unsigned)VW;
                const int sy = p.upsample ? (vy >> 1) : vy, sx = p.upsample ? (vx >> 1) : vx;
                src = ok ? A + (a_row_off[rr] + (long)sy * p.W + sx) * p.lda + ci : zeros;
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sA + (rr * 256 + wave * 64) * 8), 16, 0, 0);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const E* src = (b_ok[rr] && k < p.Kw) ? Wt + b_row_off[rr] + k : zeros;
            __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(sB + (rr * 256 + wave * 64) * 8), 16, 0, 0);
        }
    };

    f4_t acc[4][4];  // [n tile j][m tile i]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f4_t{0.f, 0.f, 0.f, 0.f};

    const int nt = (p.K + BK - 1) / BK;
    const int fr = lane & 15, fq = lane >> 4;

    stage(0, 0);
    for (int kt = 0; kt < nt; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nt) stage(kt + 1, cur ^ 1);
        const E* sA = smem + cur * 2 * TILE_ELEMS;
        const E* sB = sA + TILE_ELEMS;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            V8 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wm * 64 + i * 16 + fr;
                const int slot = (kk * 4 + fq) ^ ((row >> 1) & 7);
                af[i] = *reinterpret_cast<const V8*>(sA + row * BK + slot * 8);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wn * 64 + j * 16 + fr;
                const int slot = (kk * 4 + fq) ^ ((row >> 1) & 7);
                bf[j] = *reinterpret_cast<const V8*>(sB + row * BK + slot * 8);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = TT::mfma32(bf[j], af[i], acc[j][i]);
        }
    }

    // ---- epilogue: lane holds rows n = nb + 0..3 (consecutive output channels) of column m
    const float* bias = p.bias;
    const float* rowbias = p.rowbias;
    const E* res = reinterpret_cast<const E*>(p.residual);
    const bool geglu = p.flags & GEMM_GEGLU;
    const bool out32 = p.flags & GEMM_OUT_F32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        if (m >= p.M) continue;
        const float* rb = rowbias ? rowbias + (long)(m / p.rows_per_sample) * p.ld_rowbias : nullptr;
        if (!geglu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nb = n0 + wn * 64 + j * 16 + fq * 4;
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
        } else {
            // packed rows: every 32-row block of Wt is [16 value rows ; 16 gate rows] of the same 16
            // output channels -> tiles (j, j+1) pair up; output channel = (n/32)*16 + n%16.
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int nb = n0 + wn * 64 + jj * 32 + fq * 4;  // packed index of the value rows
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

template <class TT>
int launch_gemm(const GemmParams& p, hipStream_t stream) {
    dim3 grid((p.M + BM - 1) / BM, (p.N + BN - 1) / BN);
    const size_t lds = 2 * 2 * TILE_ELEMS * sizeof(typename TT::elem);
    if (p.mode == 0) {
        hipLaunchKernelGGL((gemm_kernel<TT, 0>), grid, dim3(256), lds, stream, p);
    } else {
        hipLaunchKernelGGL((gemm_kernel<TT, 1>), grid, dim3(256), lds, stream, p);
    }
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

}  // namespace

int vf_launch_gemm(const GemmParams& p, int dtype, hipStream_t stream) {
    if (!p.A || !p.Wt || !p.C || !p.zeros) return VF_ERR_ARG;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return VF_ERR_ARG;
    if ((p.K & 7) || (p.Kw & 7) || (p.lda & 7) || (p.ldw & 7) || (p.N & 3) || (p.ldc & 3)) return VF_ERR_ALIGN;
    if (((uintptr_t)p.A | (uintptr_t)p.Wt | (uintptr_t)p.zeros) & 15) return VF_ERR_ALIGN;
    if ((uintptr_t)p.C & 7) return VF_ERR_ALIGN;
    if (p.residual && (((uintptr_t)p.residual & 7) || (p.ldr & 3))) return VF_ERR_ALIGN;
    if (p.rowbias && (p.rows_per_sample <= 0 || (p.ld_rowbias & 3))) return VF_ERR_ARG;
    if ((p.flags & GEMM_GEGLU) && ((p.N & 31) || (p.flags & GEMM_OUT_F32) || p.residual || p.rowbias)) return VF_ERR_SHAPE;
    if (p.A2 && (p.mode != 0 || p.K1 <= 0 || (p.K1 % BK) || (p.lda2 & 7) || ((uintptr_t)p.A2 & 15))) return VF_ERR_ALIGN;
    if (p.mode == 1) {
        if (p.Cin <= 0 || (p.Cin & 7) || p.K != 9 * p.Cin) return VF_ERR_SHAPE;
        if (p.stride != 1 && p.stride != 2) return VF_ERR_SHAPE;
    }
    if (dtype == VF_DTYPE_F16) return launch_gemm<F16>(p, stream);
    if (dtype == VF_DTYPE_BF16) return launch_gemm<BF16>(p, stream);
    return VF_ERR_DTYPE;
}
