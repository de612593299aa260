// 256 x 320 x 64 MFMA GEMM for the long token matrices of the VFace UNet (gfx950): the Linear layers of the 640- / 1280-channel
// transformer blocks (attention.py:37-64 FeedForward / GEGLU, :171-172 to_q|to_k|to_v) once a batch holds enough rows to fill
// the chip with tiles this size.
//
//   C[m, n] = epilogue( sum_k A(m, k) * Wt[n, k] ),  16-bit C, every channel count of this UNet is a multiple of 320
//
// Why a second plain-GEMM kernel.  gemm.hip's 128 x 160 tile moves 36 KB into LDS per 64-deep K tile for 2.6 MFLOP: nine LDS-DMA
// wave-instructions (60-100 issue cycles each) and 18 fragment reads per wave against 40 MFMAs (640 matrix-pipe cycles) -- the
// K loop is co-limited by DMA issue, LDS read bandwidth (0.9 of the MFMA time) and the matrix pipe, and ends near 0.8-1.0
// PFLOP/s where the vendor library's 256-wide tiles reach 1.25-1.4 on the same shapes (tools/vs_library_f32.py).  Here a
// workgroup owns 256 rows x 320 channels: 72 KB per K tile for 10.5 MFLOP -- nine DMA pieces and 28 fragment reads per wave
// against 80 MFMAs: every other resource sits at <= 0.65 of the matrix pipe's time.
//
// * 512 threads = 8 waves as 4 (row quarters) x 2 (channel halves), two waves per SIMD, one workgroup per CU; a wave owns
//   64 rows x 160 channels = 4 x 10 mfma_f32_16x16x32 tiles (160 accumulator registers).
// * LDS: two stages of [256 activation rows | 320 weight rows] x 128 B, rows swizzled exactly as in gemm.hip (16-B slot of
//   k-chunk c of row r at c ^ ((r >> 1) & 7)); everything arrives by `buffer_load_dwordx4 ... lds` (rows past M are
//   out-of-range offsets: the hardware writes zeros).  One barrier per K tile; the nine pieces of tile t+1 go out one per
//   MFMA group at the head of tile t's compute phase.
// * K-tile compute as 20 steps (2 k32 halves x 10 channel tiles): per step one weight fragment is read two steps ahead and four
//   MFMAs (the wave's four row tiles) issue; the activation fragments of a half stay in registers for its ten steps.
// * Operand roles, fragment layout and the per-accumulator MFMA order (K tile by K tile, half 0 then half 1) are gemm.hip's:
//   the results are BIT-IDENTICAL to that kernel's, so which of the two a launch takes may depend on the batch (tests).
// * Epilogues (template EPI): 0: + bias, round, transposed through LDS as 16-bit rows, 16-byte stores; 1: GEGLU (weight rows
//   interleaved in 16-row value / gate blocks, packing.pack_geglu) the same way; 2: + bias + residual rows (the fp32 stream's, or
//   16-bit ones), one rounding, transposed in fp32; 3: the fp32 residual-stream form -- + bias + per-sample row bias + fp32 residual rows -> fp32
//   carrier (+ optional 16-bit copy, + optional per-64-row column statistics of the carrier), gemm.hip's NT = 5 epilogue on each
//   80-channel half of the wave's tile.
#include <type_traits>

#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

constexpr int BM2 = 256, BN2 = 320;
constexpr int A_BYTES = BM2 * 128, B_BYTES = BN2 * 128, STAGE_BYTES = A_BYTES + B_BYTES;   // 32 KB + 40 KB

// NJ: 16-channel tiles per wave -- 10 = the 256 x 320 tile; 8 = a 256 x 256 tile (round 6): the same kernel with eight weight pieces and
// sixteen steps per K tile, for the launches whose 320-wide tile count leaves the last round of the one-workgroup-per-CU grid mostly
// empty (N = 1280: 4 n-tiles x 96 m-tiles = 1.5 rounds; five 256-wide n-tiles make 1.875 rounds of tiles 0.8 as long).  Every output
// element's MFMA order over K is unchanged, so the bits are too and the launcher may pick the width by the batch.
template <class TT, int EPI, int NJ = 10>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(GemmParams p) {
    static_assert(NJ == 10 || (NJ == 8 && EPI != 3), "channel tiles per wave");
    constexpr int BN2 = 32 * NJ, WN = 16 * NJ;                       // tile width; channels per wave
    constexpr int B_BYTES = BN2 * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NPC = 4 + NJ / 2, NS = 2 * NJ;                      // LDS-DMA pieces per wave and K tile; steps per K tile
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    using V4 = typename TT::v4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    constexpr unsigned OOB = 0xFFFFFFF0u;

    // ---- tile origin: gemm.hip's XCD-aware order (a contiguous run of the tile sequence per XCD; column groups of GN n-tiles,
    // m-major inside a group)
    const int ntn = p.N / BN2, ntm = (p.M + BM2 - 1) / BM2;
    int m0, n0;
    {
        const int nwg = ntn * ntm, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, loc = id >> 3;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        const int GN = p.tile_group > 0 ? p.tile_group : 8;
        const int g = L / (GN * ntm);
        const int rem = L - g * (GN * ntm);
        const int gw = min(GN, ntn - g * GN);
        const int tm = rem / gw;
        m0 = tm * BM2;
        n0 = (g * GN + (rem - tm * gw)) * BN2;
    }

    const float* rowbias_row = nullptr;      // EPI 3: the per-sample row bias of this tile (every tile lies inside one sample)
    if constexpr (EPI == 3) {
        const int rps = max(p.rows_per_sample, 1);
        const int smp = __builtin_amdgcn_readfirstlane(m0 / rps);      // (uniform: keeps the pointer in scalar registers across the K loop)
        rowbias_row = p.rowbias ? p.rowbias + (long)smp * p.ld_rowbias : nullptr;
    }
    const i32x4_t rA = raw_buffer_rsrc(p.A, p.a_bytes);
    const i32x4_t rA2 = raw_buffer_rsrc(p.A2 ? p.A2 : p.A, p.A2 ? p.a2_bytes : 0u);
    const i32x4_t rW = raw_buffer_rsrc(p.Wt, p.w_bytes);

    // ---- staging map.  Piece q = rows 8q .. 8q+7 of a tile (1 KiB, lane-linear in LDS): lane l -> row 8q + (l >> 3), 16-B slot
    // l & 7 holds logical k-chunk (l & 7) ^ ((row >> 1) & 7).  Wave w issues activation pieces w, w+8, w+16, w+24 and weight
    // pieces w, w+8, .., w+32.
    unsigned a_off[4], a2_off[4], b_off[NJ / 2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave + 8 * i) * 8 + (lane >> 3);
        const unsigned ch = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
        const long m = m0 + row;
        a_off[i] = m < p.M ? (unsigned)((m * p.lda + ch * 8) * 2) : OOB;
        a2_off[i] = (p.A2 && m < p.M) ? (unsigned)((m * p.lda2 + ch * 8) * 2) : OOB;
    }
#pragma unroll
    for (int i = 0; i < NJ / 2; ++i) {
        const int row = (wave + 8 * i) * 8 + (lane >> 3);
        const unsigned ch = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
        b_off[i] = (unsigned)((((long)(n0 + row)) * p.ldw + ch * 8) * 2);
    }
    const unsigned lds0 = lds_addr_of(smem_raw);
    // piece s (0..3 activations, 4..8 weights; compile-time) of K tile kt into stage `buf`
    auto issue_piece = [&](int s, int kt, int buf) {
        const unsigned kbase = (unsigned)kt * 64u;
        if (s < 4) {
            const bool second = p.A2 && (int)kbase >= p.K1;      // wave-uniform
            const unsigned base = second ? a2_off[s] : a_off[s];
            const unsigned off = base != OOB ? base + (second ? kbase - (unsigned)p.K1 : kbase) * 2u : OOB;
            const unsigned dst = lds0 + (unsigned)buf * STAGE_BYTES + (unsigned)(wave + 8 * s) * 1024u;
            if (second) raw_lds_dma16(rA2, dst, (int)off, 0);
            else raw_lds_dma16(rA, dst, (int)off, 0);
        } else {
            const unsigned dst = lds0 + (unsigned)buf * STAGE_BYTES + A_BYTES + (unsigned)(wave + 8 * (s - 4)) * 1024u;
            raw_lds_dma16(rW, dst, (int)(b_off[s - 4] + kbase * 2u), 0);
        }
    };

    f4_t acc[NJ][4];   // [channel tile j][row tile i]
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f4_t{0.f, 0.f, 0.f, 0.f};

    // fragment addresses inside a stage (k32 half 0; half 1 = the same address with byte bit 6 flipped; row tiles and channel
    // tiles are 16 rows = 2 KiB apart and keep the swizzle term)
    const int arow = wm * 64 + fr, wrow = wn * WN + fr;
    const unsigned aadr = (unsigned)(arow * 128 + ((fq ^ ((arow >> 1) & 7)) << 4));
    const unsigned wadr = (unsigned)(A_BYTES + wrow * 128 + ((fq ^ ((wrow >> 1) & 7)) << 4));

    // ---- the K loop.  A K tile is 20 steps (k32 half kk = s / 10, channel tile j = s % 10); step s issues the four MFMAs of
    // (kk, j) on the wave's four row tiles.  Weight fragments are read TWO steps ahead into a ring of four; the activation
    // fragments of half 1 are read at step 4, those of the NEXT tile's half 0 at step 18.  The one barrier per tile sits at step 18:
    // by then every LDS read of tile kt has been issued (the read for step 19 went out at step 17), so behind the wave's own
    // `lgkmcnt(0)` + `vmcnt(0)` and the barrier (i) tile kt+1 is complete in the other stage and its first fragments are requested
    // while the MFMAs of steps 18 and 19 -- operands already in registers -- run, and (ii) tile kt's stage is free: the nine pieces
    // of tile kt+2 go out at steps 18, 19 of tile kt and 0..6 of tile kt+1, a full tile ahead of their use.
    const int nt = p.K >> 6;
#pragma unroll
    for (int s = 0; s < NPC; ++s) issue_piece(s, 0, 0);
    if (nt > 1) { issue_piece(0, 1, 1); issue_piece(1, 1, 1); }      // (what steps 18, 19 of a tile "-1" would have issued)
    V8 af[2][4], wf[4];
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");          // tile 0 has landed (the two pieces of tile 1 may still fly)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) af[0][i] = *reinterpret_cast<const V8*>(smem_raw + aadr + i * 2048);
    wf[0] = *reinterpret_cast<const V8*>(smem_raw + wadr);
    wf[1] = *reinterpret_cast<const V8*>(smem_raw + wadr + 2048);

    for (int kt = 0; kt < nt; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nt, more2 = kt + 2 < nt;
        const unsigned char* st = smem_raw + cur * STAGE_BYTES;
        const unsigned char* sn = smem_raw + (cur ^ 1) * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int kk = s / NJ, j = s - kk * NJ;
            if (s == NS - 2 && more) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            {
                const int s2 = s + 2;
                if (s2 < NS) {
                    const int kk2 = s2 / NJ, j2 = s2 - kk2 * NJ;
                    wf[s2 & 3] = *reinterpret_cast<const V8*>(st + ((wadr + j2 * 2048) ^ (kk2 ? 64u : 0u)));
                } else if (more) {
                    wf[s2 & 3] = *reinterpret_cast<const V8*>(sn + wadr + (s2 - NS) * 2048);
                }
            }
            if (s == 4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[1][i] = *reinterpret_cast<const V8*>(st + ((aadr + i * 2048) ^ 64u));
            }
            if (s == NS - 2 && more) {
#pragma unroll
                for (int i = 0; i < 4; ++i) af[0][i] = *reinterpret_cast<const V8*>(sn + aadr + i * 2048);
            }
            // pieces of the tile after next: 0, 1 at steps 18, 19 (into the stage this tile just left), 2..8 at steps 0..6 of the next
            if (s >= NS - 2) { if (more2) issue_piece(s - (NS - 2), kt + 2, cur); }
            else if (s < NPC - 2) { if (more) issue_piece(s + 2, kt + 1, cur ^ 1); }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = TT::mfma32(wf[s & 3], af[kk][i], acc[j][i]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue.  Lane (fr, fq) holds, per (j, i), channels n0 + wn*160 + j*16 + fq*4 .. +3 of row m0 + wm*64 + i*16 + fr.
    __syncthreads();                    // every wave is done with the last stage: the stages become transpose scratch
    {   // (lane coordinates re-derived from the thread id behind an opaque asm: hipcc otherwise carries the K loop's copies across
        //  the loop in spilled registers instead of re-computing two ANDs)
    int te = threadIdx.x;
    asm volatile("" : "+v"(te));
    const int lane = te & 63, fr = lane & 15, fq = lane >> 4;
    const int wrow0 = m0 + wm * 64;
    {
        // bias (and, EPI 3, the per-sample row bias: the launcher only takes tiles that lie inside one sample) summed into the
        // accumulators up front -- gemm.hip's order of fp32 additions, ((acc + bias) + rowbias) + residual -- five channel tiles at a
        // time (20 registers beside the 160 accumulators), unconditionally: an absent vector is loaded as zeros
        // (through a buffer descriptor: an absent vector is a descriptor of zero bytes, every load of it returns zeros -- no branch
        //  per load, which hipcc turns into a region the accumulators are spilled around)
        auto add_vec = [&](const float* vec) {
            const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(vec ? (const void*)vec : p.Wt), 0, vec ? p.N * 4 : 0, 0x00020000);
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
                constexpr int HJ = NJ / 2;
                f4_t bj[HJ];
#pragma unroll
                for (int j = 0; j < HJ; ++j)
                    bj[j] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rV, (n0 + wn * WN + (jh * HJ + j) * 16 + fq * 4) * 4, 0, 0));
#pragma unroll
                for (int j = 0; j < HJ; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[jh * HJ + j][i] += bj[j];
                        asm volatile("" : "+v"(acc[jh * HJ + j][i]));      // (pinned HERE: hipcc otherwise sinks the sums to their first use and spills the vector meanwhile)
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        add_vec(p.bias);
        if constexpr (EPI == 3) {
            // (unconditional: a conditional update of the 160 accumulator registers makes hipcc keep two copies of them; the row's
            //  sample was divided out before the K loop: a branch here and hipcc sinks the bias sums below it, spilling the bias)
            add_vec(rowbias_row);
        }
    }
    __builtin_amdgcn_sched_barrier(0);      // (nothing of the passes below is hoisted above the bias sums: its registers are those the bias held)
    E* Cout = reinterpret_cast<E*>(p.C);
    if constexpr (EPI == 0 || EPI == 1) {
        constexpr int OW = EPI == 1 ? WN / 2 : WN;        // output channels of this wave
        constexpr int CH = OW / 8;                       // 16-byte chunks per output row
        constexpr int SPH = OW * 2 + 16;                 // scratch row pitch (bytes)
        constexpr int NIT = (16 * CH + 63) / 64;
        unsigned char* scr = smem_raw + wave * (16 * SPH);
        const int ncol0 = EPI == 1 ? ((n0 + wn * WN) >> 1) : (n0 + wn * WN);
        int rr[NIT], cc[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = lane + 64 * it;
            rr[it] = idx < 16 * CH ? idx / CH : -1;
            cc[it] = idx - (idx / CH) * CH;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned char* hrow = scr + fr * SPH + fq * 8;
            if constexpr (EPI == 0) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    *reinterpret_cast<V4*>(hrow + j * 32) = V4{from_f32<E>(acc[j][i][0]), from_f32<E>(acc[j][i][1]), from_f32<E>(acc[j][i][2]), from_f32<E>(acc[j][i][3])};
            } else {
#pragma unroll
                for (int jj = 0; jj < NJ / 2; ++jj) {
                    // value rows in the even channel tiles, gate rows 16 further in the odd ones (packing.pack_geglu)
                    const float a0 = acc[2 * jj][i][0], a1 = acc[2 * jj][i][1], a2 = acc[2 * jj][i][2], a3 = acc[2 * jj][i][3];
                    const float g0 = acc[2 * jj + 1][i][0], g1 = acc[2 * jj + 1][i][1], g2 = acc[2 * jj + 1][i][2], g3 = acc[2 * jj + 1][i][3];
                    *reinterpret_cast<V4*>(hrow + jj * 32) = V4{from_f32<E>(a0 * gelu_erf_f(g0)), from_f32<E>(a1 * gelu_erf_f(g1)),
                                                                from_f32<E>(a2 * gelu_erf_f(g2)), from_f32<E>(a3 * gelu_erf_f(g3))};
                }
            }
            // (LDS operations of one wave execute in order: the reads below see the writes above, the next row tile's writes
            // cannot overtake these reads)
            V8 o[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) o[it] = *reinterpret_cast<const V8*>(scr + max(rr[it], 0) * SPH + cc[it] * 16);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const long m = wrow0 + i * 16 + rr[it];
                if (rr[it] >= 0 && m < p.M) *reinterpret_cast<V8*>(Cout + m * p.ldc + ncol0 + cc[it] * 8) = o[it];
            }
        }
    } else if constexpr (EPI == 3) {
        // the fp32 residual-stream form.  Per 80-channel half of the wave's tile: row tile by row tile, five accumulator tiles cross
        // LDS in fp32 (16 rows x 84 floats per wave), lanes re-read them as (row rrow + 6 it, 8-channel chunk rch) -- 10 chunks x 6
        // rows per pass, three passes per row tile -- add the residual rows, store the carrier (and the rounded copy), and sum
        // the stored values per channel; the six row-lanes of a channel are folded through the scratch in a fixed order.
        constexpr int SP = 84;
        float* scr = reinterpret_cast<float*>(smem_raw) + wave * (16 * SP);
        const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual ? p.residual : p.Wt), 0, (int)(p.residual ? p.res_bytes : 0u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C ? p.C : const_cast<void*>(p.Wt), 0, (int)(p.C ? p.c_bytes : 0u), 0x00020000);
        const __amdgpu_buffer_rsrc_t rC32 = __builtin_amdgcn_make_buffer_rsrc(p.C32, 0, (int)p.c32_bytes, 0x00020000);
        const int rrow = lane / 10, rch = lane - rrow * 10;
        const bool act = lane < 60;
        const bool has_res = p.residual != nullptr, has_c16 = p.C != nullptr;
        // byte offsets of (row wrow0 + rrow, this lane's chunk) in the three views, and their strides per 6 rows (pass) / 16 rows (row
        // tile): kept as RUNNING values behind an opaque asm -- left to itself hipcc precomputes all 72 (pass, view) offsets of the
        // unrolled passes up front and spills the accumulators to hold them
        const unsigned st_r6 = (unsigned)(6 * p.ldr * 4), st_c6 = (unsigned)(6 * p.ldc32 * 4), st_h6 = (unsigned)(6 * p.ldc * 2);
        const unsigned st_r16 = (unsigned)(16 * p.ldr * 4), st_c16 = (unsigned)(16 * p.ldc32 * 4), st_h16 = (unsigned)(16 * p.ldc * 2);
#pragma unroll
        for (int hc = 0; hc < 2; ++hc) {
            const int ncol = n0 + wn * 160 + hc * 80 + rch * 8;      // first channel of this lane's chunk
            unsigned o_r = (unsigned)(((long)(wrow0 + rrow) * p.ldr + ncol) * 4), o_c = (unsigned)(((long)(wrow0 + rrow) * p.ldc32 + ncol) * 4);
            unsigned o_h = (unsigned)(((long)(wrow0 + rrow) * p.ldc + ncol) * 2);
            unsigned o_rn = o_r;                                     // the residual loads run one row tile ahead
            int mrow = wrow0 + rrow, mrow_n = mrow;                   // absolute row of pass 0 of the current / the prefetched row tile
            float s8[8], q8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) s8[e] = q8[e] = 0.f;
            f4_t r32[3][2];
            auto load_res = [&](int it) {      // pass `it` of the row tile (o_rn, mrow_n) point at
                const int r = rrow + 6 * it;
                const unsigned off = (act && r < 16 && mrow_n + 6 * it < p.M && has_res) ? o_rn + it * st_r6 : OOB;
                r32[it][0] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rR, (int)off, 0, 0));
                r32[it][1] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rR, (int)off, 16, 0));
            };
#pragma unroll
            for (int it = 0; it < 3; ++it) load_res(it);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* srow = scr + fr * SP + fq * 4;
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    *reinterpret_cast<float4*>(srow + j * 16) = make_float4(acc[hc * 5 + j][i][0], acc[hc * 5 + j][i][1], acc[hc * 5 + j][i][2], acc[hc * 5 + j][i][3]);
                o_rn += st_r16; mrow_n += 16;
                asm volatile("" : "+v"(o_rn), "+v"(mrow_n));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    const int r = rrow + 6 * it;
                    const float* lrd = scr + min(r, 15) * SP + rch * 8;
                    const f4_t x0 = *reinterpret_cast<const f4_t*>(lrd), x1 = *reinterpret_cast<const f4_t*>(lrd + 4);
                    const f4_t a = r32[it][0], b = r32[it][1];
                    const bool ok = act && r < 16 && mrow + 6 * it < p.M;
                    float v[8] = {x0[0] + a[0], x0[1] + a[1], x0[2] + a[2], x0[3] + a[3], x1[0] + b[0], x1[1] + b[1], x1[2] + b[2], x1[3] + b[3]};
                    if (i + 1 < 4) load_res(it);
                    const unsigned o32 = ok ? o_c + it * st_c6 : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, f4_t{v[0], v[1], v[2], v[3]}), rC32, (int)o32, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, f4_t{v[4], v[5], v[6], v[7]}), rC32, (int)o32, 16, 0);
                    if (has_c16) {
                        V8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = from_f32<E>(v[e]);
                        const unsigned o16 = ok ? o_h + it * st_h6 : OOB;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, o), rC, (int)o16, 0, 0);
                    }
                    if (ok) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { s8[e] += v[e]; q8[e] = fmaf(v[e], v[e], q8[e]); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                o_c += st_c16; o_h += st_h16; mrow += 16;
                asm volatile("" : "+v"(o_c), "+v"(o_h), "+v"(mrow));
            }
            if (p.colstats) {
                // fold the six row-lanes of every channel through the scratch (fixed order: reproducible); the wave's 64 rows are
                // one statistics slice
                if (act) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        scr[(rrow * 80 + rch * 8 + e) * 2] = s8[e];
                        scr[(rrow * 80 + rch * 8 + e) * 2 + 1] = q8[e];
                    }
                }
                const long slice = (long)wrow0 >> 6;
                if (wrow0 < p.M) {
                    for (int c = lane; c < 80; c += 64) {
                        float ss = 0.f, qq = 0.f;
                        for (int l = 0; l < 6; ++l) { ss += scr[(l * 80 + c) * 2]; qq += scr[(l * 80 + c) * 2 + 1]; }
                        *reinterpret_cast<float2*>(p.colstats + (slice * p.ld_colstats + n0 + wn * 160 + hc * 80 + c) * 2) = make_float2(ss, qq);
                    }
                }
            }
        }
    } else {
        // fp32 transposes; the residual rows and the output go through buffer descriptors (32-bit offsets, rows past M are
        // out-of-range offsets: loads return zeros, stores are dropped) -- one offset register per chunk instead of a pointer pair
        constexpr int SP = WN + 4;                       // fp32 scratch row pitch (floats)
        constexpr int CPR = WN / 8, NPASS = 16 * CPR / 64;       // 8-channel chunks per row; passes of 64 lanes per row tile (5 | 4)
        float* scr = reinterpret_cast<float*>(smem_raw) + wave * (16 * SP);
        const int ncol0 = n0 + wn * WN;
        const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.residual), 0, (int)p.res_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)p.c_bytes, 0x00020000);
        unsigned lofs[NPASS];                            // LDS byte offset of this lane's chunk (row rr, chunk cc) per pass
        int rr[NPASS], cc[NPASS];
#pragma unroll
        for (int it = 0; it < NPASS; ++it) {
            const int idx = lane + 64 * it;
            rr[it] = idx / CPR;
            cc[it] = idx - rr[it] * CPR;
            lofs[it] = (unsigned)((rr[it] * SP + cc[it] * 8) * 4);
        }
        // residual rows as a RING of one row tile (40 registers): pass `it` of row tile i + 1 is requested the moment pass `it` of row
        // tile i has been summed -- five passes ahead of its use; two full row tiles in flight do not fit beside the accumulators
        f4_t r32[NPASS][2];
        const bool res16 = !p.res_f32;      // 16-bit residual rows (a block's interior running sum kept in 16 bits): one 16-byte load per chunk
        auto load_res = [&](int i, int it) {
            const long m = wrow0 + i * 16 + rr[it];
            if (res16) {
                const unsigned off = m < p.M ? (unsigned)((m * p.ldr + ncol0 + cc[it] * 8) * 2) : OOB;
                const V8 h = __builtin_bit_cast(V8, __builtin_amdgcn_raw_buffer_load_b128(rR, (int)off, 0, 0));
                r32[it][0] = f4_t{to_f32(h[0]), to_f32(h[1]), to_f32(h[2]), to_f32(h[3])};
                r32[it][1] = f4_t{to_f32(h[4]), to_f32(h[5]), to_f32(h[6]), to_f32(h[7])};
            } else {
                const unsigned off = m < p.M ? (unsigned)((m * p.ldr + ncol0 + cc[it] * 8) * 4) : OOB;
                r32[it][0] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rR, (int)off, 0, 0));
                r32[it][1] = __builtin_bit_cast(f4_t, __builtin_amdgcn_raw_buffer_load_b128(rR, (int)off, 16, 0));
            }
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* srow = scr + fr * SP + fq * 4;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                *reinterpret_cast<float4*>(srow + j * 16) = make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) {      // (the first row tile's residual rows: requested once its accumulators have left their registers)
#pragma unroll
                for (int it = 0; it < NPASS; ++it) load_res(0, it);
            }
#pragma unroll
            for (int it = 0; it < NPASS; ++it) {
                const unsigned char* lrd = reinterpret_cast<const unsigned char*>(scr) + lofs[it];
                const f4_t x0 = *reinterpret_cast<const f4_t*>(lrd), x1 = *reinterpret_cast<const f4_t*>(lrd + 16);
                const f4_t a = r32[it][0], b = r32[it][1];
                const long m = wrow0 + i * 16 + rr[it];
                V8 o;
                o[0] = from_f32<E>(x0[0] + a[0]); o[1] = from_f32<E>(x0[1] + a[1]); o[2] = from_f32<E>(x0[2] + a[2]); o[3] = from_f32<E>(x0[3] + a[3]);
                o[4] = from_f32<E>(x1[0] + b[0]); o[5] = from_f32<E>(x1[1] + b[1]); o[6] = from_f32<E>(x1[2] + b[2]); o[7] = from_f32<E>(x1[3] + b[3]);
                if (i + 1 < 4) load_res(i + 1, it);
                const unsigned off = m < p.M ? (unsigned)((m * p.ldc + ncol0 + cc[it] * 8) * 2) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, o), rC, (int)off, 0, 0);
                __builtin_amdgcn_sched_barrier(0);      // (pass by pass: hoisting all ten LDS reads of the row tile costs 40 registers the kernel does not have)
            }
        }
    }
    }
}

template <class TT, int EPI, int NJ>
int launch_big(const GemmParams& p, hipStream_t stream) {
    auto kern = gemm256_kernel<TT, EPI, NJ>;
    static VfOncePerDevice attr_set;
    constexpr int BN = 32 * NJ;
    constexpr int lds = 2 * (A_BYTES + BN * 128);
    if (!attr_set.set_lds(reinterpret_cast<const void*>(kern), lds)) return VF_ERR_LAUNCH;
    const int ntn = p.N / BN, ntm = (p.M + BM2 - 1) / BM2;
    GemmParams q = p;
    {
        // column-group width of the XCD-aware tile order: gemm.hip's rule (bytes through an XCD's L2 are least at
        // GN = sqrt(tiles_xcd * A_mt / W_nt); the GN weight panels stay within half of the 4 MiB L2)
        const double a_mt = (double)BM2 * p.K * 2.0, w_nt = (double)BN * p.K * 2.0;
        int gn = (int)(__builtin_sqrt((double)ntm * ntn / 8.0 * a_mt / w_nt) + 0.5);
        const int cap = (int)(2097152.0 / w_nt);
        if (gn > cap) gn = cap;
        q.tile_group = gn < 1 ? 1 : (gn > ntn ? ntn : gn);
    }
    hipLaunchKernelGGL(kern, dim3(ntn * ntm), dim3(512), lds, stream, q);
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

// Tile width of a launch: 320 channels, or 256 where N allows both, K is short and the one-workgroup-per-CU grid then takes fewer
// tile-times (rounds of 256 workgroups).  MEASURED (tools/bench_gemm_widths.py, profiles/r06_h): a 256-wide tile is NOT 0.8 of a 320-wide
// one when K is deep -- ff2 at K = 5120: 240 tiles in one round take as long as 192 (175 vs 180 us at 48 samples; 361 vs 349 at 96),
// the K loop waits for its activation pieces, which five n-tiles per m-tile request instead of four -- it pays where the prologue and
// epilogue weigh: the 1280 x 1280 projections (55 -> 45 us at 48 samples, 90 -> 80 at 96) and the dual-source projection under one
// round (80 -> 70).  Hence: K <= 1280 only, and only where the rounds say so at 0.85 of a tile-time.  The two widths give the same bits,
// so the choice may follow the batch.  GEMM_BIG_W256 / _W320 (tuning bits of vface_gemm: tests and A/B runs) force one.
template <class TT, int EPI>
int launch_big_width(const GemmParams& p, hipStream_t stream) {
    if constexpr (EPI != 3) {
        bool w256 = false;
        if (p.N % 256 == 0 && !(p.flags & GEMM_BIG_W320)) {
            const long ntm = (p.M + BM2 - 1) / BM2;
            const long r320 = (ntm * (p.N / 320) + 255) / 256, r256 = (ntm * (p.N / 256) + 255) / 256;
            w256 = (p.flags & GEMM_BIG_W256) || (p.K <= 1280 && (double)r256 * 0.85 < (double)r320);
        }
        if (w256) return launch_big<TT, EPI, 8>(p, stream);
    }
    return launch_big<TT, EPI, 10>(p, stream);
}

template <class TT>
int launch_big_dtype(const GemmParams& p, hipStream_t stream) {
    if (p.C32) return launch_big_width<TT, 3>(p, stream);
    if (p.flags & GEMM_GEGLU) return launch_big_width<TT, 1>(p, stream);
    if (p.residual) return launch_big_width<TT, 2>(p, stream);
    return launch_big_width<TT, 0>(p, stream);
}

}  // namespace

// Is this plain-GEMM launch one the 256 x 320 tile takes?  Shape, operands and epilogue form only (vf_launch_gemm has already
// validated alignments and filled the operand extents).  Since the two kernels' outputs are bit-identical, the caller is free to
// look at the batch (tile count) when it chooses between them.
bool vf_gemm_big_ok(const GemmParams& p) {
    if (p.mode != 0 || p.out_phase || p.gn_ab || p.split_k > 1 || p.a2_row_mod) return false;
    if ((p.N % BN2) || (p.K & 63) || p.K < 128 || (p.Kw < p.K)) return false;
    if (p.A2 && ((p.K1 & 63) || p.K1 <= 0 || p.K1 >= p.K)) return false;
    if (p.flags & (GEMM_OUT_F32 | GEMM_NARROW_EPILOGUE | GEMM_F32_TRANSPOSE | 0x4000 | (0xF << 8))) return false;
    if (p.C32) {
        // the fp32 residual-stream form (EPI 3): carrier required; the per-sample row bias needs every 256-row tile inside one sample
        if (p.flags & GEMM_GEGLU) return false;
        if (((uintptr_t)p.C32 & 15) || (p.ldc32 & 3)) return false;
        if (p.rowbias && (p.rows_per_sample <= 0 || (p.rows_per_sample % BM2) || (p.ld_rowbias & 3) || ((uintptr_t)p.rowbias & 15))) return false;
        if (p.colstats && ((p.M & 63) || (p.ld_colstats & 1) || ((uintptr_t)p.colstats & 7))) return false;
    } else if (p.colstats || p.rowbias || !p.C) {
        return false;
    }
    if (p.C && (((uintptr_t)p.C & 15) || (p.ldc & 7))) return false;
    if ((p.lda & 7) || (p.ldw & 7)) return false;
    if (p.residual && ((p.flags & GEMM_GEGLU) || ((uintptr_t)p.residual & 15) || (p.ldr & (p.res_f32 ? 3 : 7)))) return false;
    if (p.residual && !p.res_f32 && p.C32) return false;      // (16-bit residual rows: the 16-bit-output form only)
    if (p.bias && ((uintptr_t)p.bias & 15)) return false;
    return true;
}

// THE rule for which plain-GEMM kernel a launch runs (vf_launch_gemm dispatches on it).  Outputs and the fp32 carrier are
// bit-identical between the two kernels (the per-64-row column statistics of the residual-stream form are not: another fold
// order -- one more reason that form is never chosen automatically: a sample's bits must not depend on which other samples
// share its launch, what frame sharding and the two launch streams rely on); 16-bit-output launches take the 256 x 320 tile
// from 192 tiles on -- about a round of the one-workgroup-per-CU grid (measured: profiles/r05_c, r05_d).
bool vf_gemm_big_choice(const GemmParams& p) {
    if ((p.flags & GEMM_NO_BIG) || !vf_gemm_big_ok(p)) return false;
    if (p.flags & GEMM_BIG) return true;
    // the fp32 residual-stream form (EPI 3: proj_in / to_out / proj_out) is built, bit-identical and MEASURED SLOWER than the 128-row
    // kernel on 14 of 18 of the UNet's launches (profiles/r05_e: 0.78-1.29x): these launches are bound by their epilogue's HBM
    // traffic (4 + 4 B per output element around a K = 640 loop), which two 128-row workgroups per CU overlap with each other's
    // K loops and one 256-row workgroup per CU cannot.  It stays selectable (VFACE_TUNE_BIG_TILE) and is never chosen here.
    if (p.C32) return false;
    return (long)((p.M + BM2 - 1) / BM2) * (p.N / BN2) >= 192;
}

int vf_launch_gemm_big(const GemmParams& p_in, int dtype, hipStream_t stream) {
    if (!vf_gemm_big_ok(p_in)) return VF_ERR_SHAPE;
    GemmParams p = p_in;
    {
        // extents of the residual and output views for the epilogue's buffer descriptors (bytes; below 4 GiB - 16)
        const unsigned long nout = (p.flags & GEMM_GEGLU) ? (unsigned long)p.N / 2 : (unsigned long)p.N;
        const unsigned long cb = p.C ? ((unsigned long)(p.M - 1) * p.ldc + nout) * 2ul : 0ul;
        const unsigned long rb = p.residual ? ((unsigned long)(p.M - 1) * p.ldr + p.N) * (p.res_f32 ? 4ul : 2ul) : 0ul;
        const unsigned long c32b = p.C32 ? ((unsigned long)(p.M - 1) * p.ldc32 + p.N) * 4ul : 0ul;
        if (cb >= 0xFFFFFFF0ul || rb >= 0xFFFFFFF0ul || c32b >= 0xFFFFFFF0ul) return VF_ERR_SHAPE;
        p.c_bytes = (unsigned)cb; p.res_bytes = (unsigned)rb; p.c32_bytes = (unsigned)c32b;
    }
    if (dtype == VF_DTYPE_F16) return launch_big_dtype<F16>(p, stream);
    if (dtype == VF_DTYPE_BF16) return launch_big_dtype<BF16>(p, stream);
    return VF_ERR_DTYPE;
}
