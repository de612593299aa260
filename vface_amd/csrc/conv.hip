// Patch-staged implicit-GEMM convolution for the VFace UNet (gfx950): every input pixel is staged into LDS ONCE per
// 64-channel chunk and feeds all KH x KW taps from there.
//
//   Y[pixel, n] = epilogue( sum_{chunk c} sum_{tap (ky,kx)} sum_{ci < 64} X[pixel + (ky,kx) - pad, c*64 + ci] * Wt[n, (c, tap, ci)]
//                           (+ sum_k X2[pixel, k] * Wt[n, K1 + k]  -- the fused 1x1 shortcut) )
//
// (openaimodel.py ResBlock convs :201-205,225-232, Upsample.conv :108-118 as its four parity-phase 2x2 windows,
// diffusionmodules/model.py ResnetBlock / Upsample; stride 1 only -- the three stride-2 Downsample convs of a forward and
// images that are not multiples of 16 stay on gemm.hip's im2col kernel, which re-stages a pixel once per tap.)
//
// Why: gemm.hip's implicit GEMM moves 16 KB of activations AND 20 KB of weights into LDS per 64-deep K tile of a
// 128 x 160 tile -- nine LDS-DMA wave-instructions per wave and tile, which is what its main loop is bound by.
// Here a workgroup owns 16 x 16 output pixels x BN output channels: the 18 x 18 x 64-channel halo patch (41 KB) is
// staged once per chunk, double-buffered, and the nine taps read it at shifted pixel addresses; only the weight tile
// (BN x 64, 20 KB) streams per tap.  Per wave and K tile: 40 MFMAs against ~3.1 LDS-DMA instructions (was 9), and the
// activation bytes a workgroup pulls through L2 drop 6.3x.
//
// * 512 threads = 8 waves as 4 (pixel rows) x 2 (channel halves); a wave owns 4 image rows x 16 pixels x BN/2 channels
//   = 4 x NT mfma_f32_16x16x32 tiles (NT = 5 | 4), fp32 accumulate, one workgroup per CU (LDS 143 KB, <= 256 VGPRs).
// * LDS: patch[2] (324 pixels x 128 B, pixel-major, 16-B slot of chunk q of pixel p at q ^ (p & 7): the 16
//   consecutive pixels of an MFMA tile read conflict-free at every tap shift) + a 3-slot ring of weight tiles
//   (row-major 128-B rows, same swizzle as gemm.hip).  Everything arrives by `buffer_load ... lds` (zero padding and the
//   patch's out-of-image halo are out-of-range offsets: the hardware writes zeros).
// * Schedule: one barrier per K tile.  Weight tile t+2 and one piece of the NEXT chunk's patch are issued right after
//   barrier t; `s_waitcnt vmcnt(n)` with n = this wave's weight pieces of tile t+1 retires everything older (in-order).
// * Weights are the MFMA A operand, activations the B operand -- the same fragment layout, K order (chunk, tap, channel)
//   and per-accumulator MFMA order as gemm.hip, so results are BIT-IDENTICAL to that kernel's (tests rely on it).
// * Epilogue: the wide (LDS-transposed, 16 B per lane) epilogue of gemm.hip: bias + per-sample row bias + residual
//   (16-bit or the fp32 stream) summed in fp32, single rounding, optional fp32 carrier, per-64-pixel column statistics.
#include <type_traits>
#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

constexpr int TP = 16;                 // output tile: TP x TP pixels of one image

// DIAG: diagnostic build (flag 0x4000, tools/stamp_conv.py; never used by the engine): s_memtime stamps around the parts of
// the kernel and of every K-tile period; the sums go to the colstats pointer as [workgroup][wave][8] floats and feed no output.
// GNT: the build with the fused input normalisation (GroupNorm-apply + SiLU on the staged patch); a separate instantiation so
// the plain convolution keeps its register allocation.
// RM: the residual form of the epilogue (0 none, 1 16-bit, 2 the fp32 stream) as an instantiation of its own -- three copies of
// the epilogue inside one kernel cost 19 spilled registers, some reloaded inside the K loop; -1 = chosen at run time (the
// diagnostic and fused-normalisation builds, which the engine's default path does not launch).
// Q8: the 8 x 8-pixel level.  A workgroup's 16 x 16 "tile" is FOUR images (2 x 2 quadrants of 8 x 8 pixels, each staged with
// its own one-pixel halo: a 20 x 20-pixel patch), K is split over the workgroups of a tile by channel chunks, and the kernel
// ends with the raw fp32 partial tile -- bias / residual / rounding / statistics run in gemm.hip's splitk_reduce_kernel.
template <class TT, int NT, int KH, int KW, bool DIAG = false, bool GNT = false, int RM = -1, bool Q8 = false>
__global__ __launch_bounds__(512, 2) void conv_patch_kernel(GemmParams p) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    constexpr int BN = 32 * NT;                    // 160 | 128
    constexpr int TAPS = KH * KW;
    static_assert(!Q8 || (KH == 3 && KW == 3 && NT == 4), "the 8x8 form: 3x3 windows, 128-channel tiles");
    constexpr int PW = Q8 ? 20 : TP + KW - 1, PH = Q8 ? 20 : TP + KH - 1, NPIX = PW * PH;
    constexpr int NPIECES = (NPIX + 7) / 8;        // 1-KiB LDS-DMA pieces per patch (8 pixels x 128 B each)
    constexpr int PATCH_BYTES = NPIECES * 1024;
    constexpr int PPW = (NPIECES + 7) / 8;         // patch pieces per wave (wave w: pieces w, w + 8, ...)
    constexpr int BPIECES = BN / 8;                // pieces per weight tile (8 rows x 128 B each)
    constexpr int BSLOT = BN * 128;                // bytes per weight slot
    constexpr int NB_HI = (BPIECES + 7) / 8, NB_LO = BPIECES / 8;   // pieces per wave: waves < BPIECES % 8 issue NB_HI
    constexpr int NB_SPLIT = BPIECES % 8;          // (0 = every wave issues NB_LO == NB_HI pieces)
    // weight ring: the slot of K tile (chunk c, tap) is tap % NSLOT at compile time (TAPS % NSLOT == 0)
    constexpr int NSLOT = (TAPS % 3 == 0) ? 3 : 4;
    static_assert(TAPS % NSLOT == 0 && (NSLOT - 1) * BSLOT < 65536, "slot offsets must fold into the ds_read immediate");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* sB = smem_raw;                          // [NSLOT][BSLOT]   (first: slot offsets stay < 64 KiB)
    unsigned char* sP = smem_raw + NSLOT * BSLOT;          // [2][PATCH_BYTES]
    // fused input normalisation (3x3 windows only): (a, b) of the 64 channels of a chunk, [2][1 KiB] (512 B used: one 16-B
    // LDS-DMA instruction of 32 live lanes; the other 32 lanes write zeros behind it)
    constexpr bool GNF = GNT && KH == 3;
    unsigned char* sT = sP + 2 * PATCH_BYTES;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    constexpr unsigned ES = sizeof(E);
    constexpr unsigned OOB = 0xFFFFFFF0u;
    auto stamp = [&]() -> unsigned long long {
        if constexpr (!DIAG) return 0ull;
        unsigned long long tt;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return tt;
    };
    const unsigned long long d_t0 = stamp();
    const unsigned long long d_sync = 0, d_issue = 0, d_comp = 0;

    // ---- tile origin: XCD-aware order as in gemm.hip (contiguous run of the tile sequence per XCD; column groups of
    // 8 n-tiles, m-major inside a group)
    const int tiles_x = p.W / TP, tiles_y = p.H / TP, tiles_img = tiles_x * tiles_y;
    const int nimg_all = p.M / (p.OH * p.OW);
    const int ntm = Q8 ? (nimg_all + 3) / 4 : nimg_all * tiles_img, ntn = p.N / BN;
    int tm, tn, sk = 0;
    {
        const int nwg = ntm * ntn * (Q8 ? p.split_k : 1), id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, loc = id >> 3;
        int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        if constexpr (Q8) { sk = L / (ntm * ntn); L -= sk * (ntm * ntn); }
        const int GN = p.tile_group > 0 ? p.tile_group : 8;
        const int g = L / (GN * ntm);
        const int rem = L - g * (GN * ntm);
        const int gw = min(GN, ntn - g * GN);
        tm = rem / gw;
        tn = g * GN + (rem - tm * gw);
    }
    // (Q8: `img` = the first of the tile's four images; the tile has no origin inside an image)
    const int img = Q8 ? tm * 4 : tm / tiles_img, trem = Q8 ? 0 : tm - img * tiles_img;
    const int ty0 = Q8 ? 0 : (trem / tiles_x) * TP, tx0 = Q8 ? 0 : (trem - (trem / tiles_x) * tiles_x) * TP;
    const int n0 = tn * BN;

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A2 ? p.A2 : p.A), 0, (int)(p.A2 ? p.a2_bytes : 0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wt), 0, (int)p.w_bytes, 0x00020000);
    const bool gn = GNF && p.gn_ab != nullptr;
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gn ? p.gn_ab : (const float*)p.Wt), 0, (int)(gn ? p.gn_ab_bytes : 0u), 0x00020000);

    // ---- patch staging map.  Piece j = patch pixels 8j .. 8j+7; lane l -> pixel pp = 8j + (l >> 3), LDS 16-B slot l & 7 of
    // that pixel's 128-B row holds channel chunk (l & 7) ^ (pp & 7) (conflict-free fragment reads at every tap shift).
    constexpr bool TAIL = KH == 3;     // the fused 1x1 source only ever follows a 3x3 window (no registers for it elsewhere)
    unsigned poff[PPW], poff2[TAIL ? 4 : 1];   // byte offset of (pixel, swizzled chunk) in A (window patch) / A2 (1x1 source)
    const unsigned pch = (unsigned)((lane & 7) ^ ((lane >> 3) & 7));
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int pp = (wave + 8 * i) * 8 + (lane >> 3);
        const int py = pp / PW, px = pp - py * PW;
        if constexpr (Q8) {
            // quadrant (py / 10, px / 10) = image img + 2 qy + qx, its pixel (py % 10 - 1, px % 10 - 1) or the zero halo
            const int qy = py / 10, qx = px / 10, iy = py - qy * 10 - 1, ix = px - qx * 10 - 1, im = img + qy * 2 + qx;
            const bool ok = pp < NPIX && (unsigned)iy < 8u && (unsigned)ix < 8u && im < nimg_all;
            poff[i] = ok ? (unsigned)(((((long)im * 8 + iy) * 8 + ix) * p.lda + pch * 8) * ES) : OOB;
            continue;
        }
        const int iy = ty0 - p.pad + py, ix = tx0 - p.pad_x + px;
        const bool ok = pp < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        poff[i] = ok ? (unsigned)(((((long)img * p.H + iy) * p.W + ix) * p.lda + pch * 8) * ES) : OOB;
    }
#pragma unroll
    for (int i = 0; i < (TAIL ? 4 : 0); ++i) {
        // the 1x1 source is staged as a 16 x 16 "patch" without halo: pixel index pp = ly * 16 + lx, 32 pieces
        const int pp = (wave + 8 * i) * 8 + (lane >> 3);
        const int ly = pp >> 4, lx = pp & 15;
        if constexpr (Q8) {
            const int im = img + (ly >> 3) * 2 + (lx >> 3);
            poff2[i] = (p.A2 && im < nimg_all) ? (unsigned)(((((long)im * 8 + (ly & 7)) * 8 + (lx & 7)) * p.lda2 + pch * 8) * ES) : OOB;
            continue;
        }
        poff2[i] = p.A2 ? (unsigned)(((((long)img * p.H + ty0 + ly) * p.W + tx0 + lx) * p.lda2 + pch * 8) * ES) : OOB;
    }
    // ---- weight staging map: piece r = rows 8r .. 8r+7 of the tile, lane l -> row 8r + (l >> 3), chunk (l & 7) ^ swz(row)
    unsigned boff[NB_HI];
#pragma unroll
    for (int i = 0; i < NB_HI; ++i) {
        const int piece = wave + 8 * i;
        const int row = piece * 8 + (lane >> 3);
        const unsigned ch = (unsigned)((lane & 7) ^ ((row >> 1) & 7));
        boff[i] = (piece < BPIECES && n0 + row < p.N) ? (unsigned)((((long)(n0 + row)) * p.ldw + ch * 8) * ES) : OOB;
    }

    const int nchunks_all = p.Cin >> 6;
    // Q8: this workgroup's share of K = channel chunks [c_begin, nchunks); the fused 1x1 source goes with the last share, so the
    // chunk boundaries are placed on the K TILES of window + 1x1 source together (a 1280-channel window with a 2560-channel
    // shortcut: 180 + 40 tiles = 54 | 54 | 54 | 58 over four shares; equal chunk counts made it 45 | 45 | 45 | 85 and the launch as
    // long as its last share).  Every share keeps at least one chunk; geometry only: the same split for every batch.
    int c_begin = 0, nchunks = nchunks_all;
    if constexpr (Q8) {
        const long total = (long)nchunks_all * TAPS + ((TAIL && p.A2) ? ((p.K - p.K1) >> 6) : 0);
        int prev = 0;
        for (int k = 1; k <= p.split_k; ++k) {
            int b = k == p.split_k ? nchunks_all : (int)((k * total / p.split_k + TAPS / 2) / TAPS);
            b = min(max(b, prev + 1), nchunks_all - (p.split_k - k));
            if (k == sk) c_begin = b;
            if (k == sk + 1) { nchunks = b; break; }
            prev = b;
        }
    }
    const int kt0 = c_begin * TAPS;                       // global index of this workgroup's first K tile (weight columns)
    const int T1 = (nchunks - c_begin) * TAPS;            // K tiles of the window
    const int ntail = (TAIL && p.A2 && (!Q8 || sk == p.split_k - 1)) ? ((p.K - p.K1) >> 6) : 0;     // K tiles of the fused 1x1 source
    const int T = T1 + ntail;

    // weight tile kt -> ring slot `slot` (every K tile is 64 columns = 128 B of every weight row)
    auto issue_B = [&](int kt, int slot) {
        if (DIAG && (p.flags & 0x200000) && kt >= 2) return;     // ablation: no LDS-DMA inside the K loop
        unsigned char* dst = sB + slot * BSLOT;
        const unsigned koff = (unsigned)(kt0 + kt) * 64u * ES;
#pragma unroll
        for (int i = 0; i < NB_HI; ++i) {
            if (i < NB_LO || wave < NB_SPLIT) {
                const unsigned off = boff[i] != OOB ? boff[i] + koff : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(dst + (wave + 8 * i) * 1024), 16, off, 0, 0, 0);
            }
        }
    };
    // piece i (compile-time) of this wave for the window patch of channel chunk c
    auto issue_P = [&](int c, int i) {
        if (DIAG && (p.flags & 0x200000) && c >= 1) return;
        const int piece = wave + 8 * i;
        if (piece < NPIECES) {
            const unsigned off = poff[i] != OOB ? poff[i] + (unsigned)c * 64u * ES : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sP + (c & 1) * PATCH_BYTES + piece * 1024), 16, off, 0, 0, 0);
        }
    };
    // piece i of the 1x1-source "patch" of tail tile u (patch buffer (nchunks + u) & 1)
    auto issue_P2 = [&](int u, int i) {
        const unsigned off = poff2[TAIL ? i : 0] != OOB ? poff2[TAIL ? i : 0] + (unsigned)u * 64u * ES : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA2, LDS_PTR(sP + ((nchunks + u) & 1) * PATCH_BYTES + (wave + 8 * i) * 1024), 16, off, 0, 0, 0);
    };

    // (a, b) table of channel chunk c -> sT[c & 1]: 64 channels x 2 floats = 512 B, one instruction of wave 7
    auto issue_T = [&](int c) {
        if (gn && wave == 7) {
            const unsigned off = lane < 32 ? (unsigned)((((long)img * p.ld_gn_ab + c * 64) * 2) * 4 + lane * 16) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rT, LDS_PTR(sT + (c & 1) * 1024), 16, off, 0, 0, 0);
        }
    };
    // GroupNorm-apply (+ SiLU) of patch piece i of chunk c, in place in LDS, by the wave that issued it (so its own
    // counted vmcnt wait is all the ordering the landing needs): y = act(x * a[ch] + b[ch]) with exactly the a, b and
    // the arithmetic of gn_apply_kernel; lanes whose pixel lies outside the image keep the zeros the hardware wrote --
    // the convolution pads the NORMALISED activation with zeros (openaimodel.py:201-205)
    auto transform_P = [&](int c, int i) {
        const int piece = wave + 8 * i;
        if (piece < NPIECES && poff[i] != OOB) {
            unsigned char* a = sP + (c & 1) * PATCH_BYTES + piece * 1024 + lane * 16;
            const V8 v = *reinterpret_cast<const V8*>(a);
            const float4* tb = reinterpret_cast<const float4*>(sT + (c & 1) * 1024 + pch * 64);
            const float4 t0 = tb[0], t1 = tb[1], t2 = tb[2], t3 = tb[3];
            const float aa[8] = {t0.x, t0.z, t1.x, t1.z, t2.x, t2.z, t3.x, t3.z};
            const float bb[8] = {t0.y, t0.w, t1.y, t1.w, t2.y, t2.w, t3.y, t3.w};
            V8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = to_f32(v[e]) * aa[e] + bb[e];
                if (p.gn_silu) f = silu_f(f);
                o[e] = from_f32<E>(f);
            }
            *reinterpret_cast<V8*>(a) = o;
        }
    };

    f4_t acc[NT][4];  // [n tile j][pixel row i]
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f4_t{0.f, 0.f, 0.f, 0.f};

    // weight fragment addresses (constant over the K loop): row = wn * BN/2 + j*16 + fr, 16-B slot (kk*4 + fq) ^ swz(row);
    // the kk = 1 half is the same address with byte bit 6 flipped; the ring slot is an immediate offset
    // (ONE register: channel tile j is 16 rows = 2 KiB further and keeps the swizzle term -- ((row + 16 j) >> 1) & 7 = (row >> 1) & 7 --
    //  so the other NT - 1 addresses are immediates of the ds_read; as NT registers the NT = 5 kernel spilled one)
    //  -- where slot offset + tile offset still fit the 16-bit immediate: the three-slot ring of the 3x3 windows; the four-slot
    //  ring of the 2x2 windows keeps one register per tile)
    constexpr bool WADR1 = (NSLOT - 1) * BSLOT + (NT - 1) * 2048 + 64 < 65536;
    const int wrow = wn * (BN / 2) + fr;
    const unsigned wadr0 = (unsigned)(wrow * 128 + ((fq ^ ((wrow >> 1) & 7)) << 4));
    unsigned wadrs[WADR1 ? 1 : NT];
    if constexpr (!WADR1) {
#pragma unroll
        for (int j = 0; j < NT; ++j) wadrs[j] = wadr0 + j * 2048;
    }
    auto wadr = [&](int j) -> unsigned { if constexpr (WADR1) return wadr0 + j * 2048; else return wadrs[j]; };
    // patch pixel of (pixel row 0 of this wave, tap (0,0)); Q8: rows 8.. and columns 8.. lie in the next quadrant, 2 halo pixels on
    const int lane_pp = Q8 ? (wm * 4 + (wm >= 2 ? 2 : 0)) * PW + fr + (fr >= 8 ? 2 : 0) : wm * 4 * PW + fr;

    // ---- the K-tile pipeline.  A K tile is two k32 halves; per half a wave reads 4 + NT fragments (ds_read_b128) and issues
    // 4 x NT MFMAs.  Every wave software-pipelines at HALF-tile granularity across the barrier:
    //     period t:  sync(t) | read h0(t) -> X | MFMA h1(t-1) from Y | read h1(t) -> Y | MFMA h0(t) from X
    // so after a barrier a wave has 4 x NT MFMAs ready at once (operands in registers since the previous period) and the
    // LDS latency of the new tile's first reads runs under them; all LDS reads of tile t still happen inside period t (the
    // 3-slot ring stays valid).  The LDS-DMA issue of the period (~100 cycles per piece, in-order in the wave's stream) is
    // placed BEFORE the first MFMA batch (optionally, for A/B runs, BETWEEN the two batches by waves 4..7, the SIMD partners
    // of waves 0..3: MI355X_MICROARCH.md "Two waves per SIMD").  Every accumulator sees its MFMAs in the order (tile t half 0, tile t half 1, tile t+1 half 0, ...),
    // the same as gemm.hip: the output bits do not change.
    V8 afX[4], bfX[NT], afY[4], bfY[NT];
    unsigned padr[4];
    bool abl_reads = false;
    auto read_h0 = [&](unsigned pbase, int ppk, int pw, unsigned wbase) {
        if (DIAG && abl_reads) return;                           // ablation: fragments stay what the first K tile read
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pp = lane_pp + ppk + i * pw;
            padr[i] = pbase + (unsigned)(pp * 128) + (unsigned)((fq ^ (pp & 7)) << 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) afX[i] = *reinterpret_cast<const V8*>(smem_raw + padr[i]);
#pragma unroll
        for (int j = 0; j < NT; ++j) bfX[j] = *reinterpret_cast<const V8*>(sB + wbase + wadr(j));
    };
    auto read_h1 = [&](unsigned wbase) {
        if (DIAG && abl_reads) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) afY[i] = *reinterpret_cast<const V8*>(smem_raw + (padr[i] ^ 64u));
#pragma unroll
        for (int j = 0; j < NT; ++j) bfY[j] = *reinterpret_cast<const V8*>(sB + wbase + (wadr(j) ^ 64u));
    };
    auto mfma_X = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = TT::mfma32(bfX[j], afX[i], acc[j][i]);
        __builtin_amdgcn_s_setprio(0);
    };
    auto mfma_Y = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = TT::mfma32(bfY[j], afY[i], acc[j][i]);
        __builtin_amdgcn_s_setprio(0);
    };
    // measured (tools/bench_kernels.py, 7 UNet shapes): the same order in all waves is 3..12 % faster on 6 of 7 shapes than
    // the split placement; the split stays selectable for A/B runs (flag GEMM_NO_SETPRIO, unused otherwise by this kernel)
    const bool late = wave >= 4 && (p.flags & GEMM_NO_SETPRIO);
    float* const diag_out = DIAG ? p.colstats : nullptr;
    if (DIAG) p.colstats = nullptr;
    // start of a K-tile period: everything older than this wave's pieces of weight tile kt+1 has landed (LDS-DMA retires in
    // order): tile kt, and every patch piece issued before it; then the workgroup barrier that publishes them
    auto period_sync = [&](bool more) {
        if (gn) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's in-place patch transforms are in LDS
        if (DIAG && (p.flags & 0x200000)) { if (!(p.flags & 0x400000)) __builtin_amdgcn_s_barrier(); return; }   // ablation: no DMA wait
        if (more) {
            if (NB_SPLIT == 0 || wave >= NB_SPLIT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB_LO) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB_HI) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!(DIAG && (p.flags & 0x400000))) __builtin_amdgcn_s_barrier();                                        // ablation: no barrier
        asm volatile("" ::: "memory");
    };

    // ---- prologue: patch of chunk 0 (all pieces), weight tiles 0 and 1
    if (nchunks > c_begin) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) issue_P(c_begin, i);
        issue_T(c_begin);
    } else if constexpr (TAIL) {
        // plain GEMM (vf_launch_gemm_patch): no window at all, every K tile is a tile of the halo-free "1x1 source"
        if (ntail > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_P2(0, i);
        }
    }
    issue_B(0, 0);
    if (T > 1) issue_B(1, 1 % NSLOT);

    // ---- main loop: chunks x taps, the tap loop fully unrolled (tap offsets, ring slots and the patch-piece schedule are
    // compile-time; per K tile the scalar side only counts)
    const unsigned long long d_t1 = stamp();
    int kt = 0;
    for (int c = c_begin; c < nchunks; ++c) {
        const unsigned pbase = (unsigned)(NSLOT * BSLOT + (c & 1) * PATCH_BYTES);
        const bool next_window = c + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            period_sync(kt + 1 < T);
            if (GNF && gn) {
                if (tap == 0 && c == c_begin) {
                    // the first chunk's patch: every wave normalises the pieces it issued, then one extra barrier
#pragma unroll
                    for (int i = 0; i < PPW; ++i) transform_P(c_begin, i);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                // later chunks: the piece this wave issued one K tile ago has landed (the counted wait above) -- normalise it.
                // (Measured: these ~70 vector instructions per piece sit in the period's critical path and cost more than the
                // separate normalisation pass they replace -- DESIGN 4; the engine keeps the fused form opt-in.)
                if (tap >= 1 && tap <= PPW && next_window) transform_P(c + 1, tap - 1);
            }
            // read_h0 first: its LDS latency runs under the MFMAs of the previous tile's second half
            read_h0(pbase, (tap / KW) * PW + (tap % KW), PW, (unsigned)((tap % NSLOT) * BSLOT));
            __builtin_amdgcn_sched_barrier(0);
            if (late && kt > 0) { mfma_Y(); __builtin_amdgcn_sched_barrier(0); }
            // the next chunk's patch goes out over this chunk's K tiles, oldest first, BEFORE this period's weight pieces
            // (so the counted wait covers them one period later)
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                if (i % TAPS == tap) {
                    if (next_window) issue_P(c + 1, i);
                    else if (TAIL && ntail > 0 && i < 4) issue_P2(0, i);
                }
            }
            if (GNF && tap == 0 && next_window) issue_T(c + 1);
            if (kt + 2 < T) issue_B(kt + 2, (tap + 2) % NSLOT);
            __builtin_amdgcn_sched_barrier(0);
            if (!late && kt > 0) { mfma_Y(); __builtin_amdgcn_sched_barrier(0); }
            read_h1((unsigned)((tap % NSLOT) * BSLOT));
            __builtin_amdgcn_sched_barrier(0);
            mfma_X();
            __builtin_amdgcn_sched_barrier(0);
            ++kt;
            if (DIAG && (p.flags & 0x800000)) abl_reads = true;
        }
    }
    // ---- tail: the fused 1x1 shortcut -- one 16 x 16-pixel x 64-channel tile of the second source per K tile
    for (int u = 0; u < ntail; ++u) {
        period_sync(kt + 1 < T);
        const unsigned wb = (unsigned)((kt % NSLOT) * BSLOT);
        // lane_pp is in units of the window patch's pitch: rebase to the halo-free 16-pixel pitch
        read_h0((unsigned)(NSLOT * BSLOT + ((nchunks + u) & 1) * PATCH_BYTES), wm * 4 * TP + fr - lane_pp, TP, wb);
        __builtin_amdgcn_sched_barrier(0);
        if (late && kt > 0) { mfma_Y(); __builtin_amdgcn_sched_barrier(0); }
        if (u + 1 < ntail) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_P2(u + 1, i);
        }
        if (kt + 2 < T) issue_B(kt + 2, (kt + 2) % NSLOT);
        __builtin_amdgcn_sched_barrier(0);
        if (!late && kt > 0) { mfma_Y(); __builtin_amdgcn_sched_barrier(0); }   // (kt == 0: a plain GEMM's first tile)
        read_h1(wb);
        __builtin_amdgcn_sched_barrier(0);
        mfma_X();
        __builtin_amdgcn_sched_barrier(0);
        ++kt;
    }
    mfma_Y();      // the last half tile
    const unsigned long long d_t2 = stamp();

    if constexpr (Q8) {
        // raw fp32 partial tile of this K share: lane (fr, fq) holds channels fq*4 .. +3 of pixel (row wm*4 + i, column fr)
        float* part = p.workspace + (long)sk * p.M * p.N;
        const int im = img + (wm >= 2 ? 2 : 0) + (fr >> 3);
        if (im < nimg_all) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long m = (long)im * 64 + ((wm * 4 + i) & 7) * 8 + (fr & 7);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    *reinterpret_cast<float4*>(part + m * p.N + n0 + wn * (BN / 2) + j * 16 + fq * 4) =
                        make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
            }
        }
        return;
    }
    // ---- wide epilogue (see gemm.hip): per pixel row i, transpose the wave's 16 x WN accumulator rows through LDS in
    // fp32, sum bias / row bias / residual, round once, 16 B per lane
    __syncthreads();
    // RMODE: the residual form as a compile-time constant (0 none, 1 16-bit, 2 the fp32 stream), every load of the epilogue
    // requested up front and retired by one explicit wait -- see gemm.hip's wide epilogue for the measurement behind this
    auto epilogue = [&](auto rmode_tag) {
        constexpr int RMODE = decltype(rmode_tag)::value;
        constexpr int WN = NT * 16;
        constexpr int SP = WN + 4;
        const float* bias = p.bias;
        const float* rowbias = p.rowbias;
        const E* res = RMODE == 1 ? reinterpret_cast<const E*>(p.residual) : nullptr;
        const float* res32 = RMODE == 2 ? reinterpret_cast<const float*>(p.residual) : nullptr;
        E* Cout = reinterpret_cast<E*>(p.C);
        float* C32 = p.C32;
        float* colstats = p.colstats;
        float* scr = reinterpret_cast<float*>(smem_raw) + wave * (16 * SP);
        constexpr int CH = WN >> 3;               // 16-byte chunks per output row of this wave
        constexpr int LPR = 64 / CH;              // rows covered per read pass
        constexpr int RI = (16 + LPR - 1) / LPR;
        const bool act = lane < LPR * CH;
        const int rch = lane % CH, rrow = lane / CH;
        const int ncol = n0 + wn * WN + rch * 8;  // (< N: the launch takes whole channel tiles only)
        float s8[8], q8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s8[e] = q8[e] = 0.f;
        const long mrow0 = ((long)img * p.H + ty0 + wm * 4) * p.W + tx0;   // output row (pixel index) of (i = 0, r = 0)
        // (plain GEMM through this kernel: an "image" is 256 consecutive rows of a sample of rows_per_sample rows)
        const int smp = p.mode == 0 ? (int)(((long)img * (TP * TP)) / p.rows_per_sample) : img;
        const float* rb = rowbias ? rowbias + (long)smp * p.ld_rowbias : nullptr;
        // bias / row bias of this wave's NT column tiles, requested once up front (see gemm.hip's wide epilogue)
        float4 bj[NT], rbj[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int nb = n0 + wn * WN + j * 16 + fq * 4;
            bj[j] = bias ? *reinterpret_cast<const float4*>(bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
            rbj[j] = rb ? *reinterpret_cast<const float4*>(rb + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // residual rows: DEPTH pixel rows in flight -- the whole tile where the registers allow, else (NT = 5: up to 24
        // registers per pixel row) two, the row i + 2 requested as soon as row i has been summed, a full pass ahead of its use
        constexpr int DEPTH = (NT == 5 && RMODE != 0) ? 2 : 4;
        V8 r16[RMODE == 1 ? DEPTH : 1][RI];
        float4 r32[RMODE == 2 ? DEPTH : 1][RI][2];
        auto load_res = [&](int i, int slot) {
#pragma unroll
            for (int it = 0; it < RI; ++it) {
                const int r = rrow + it * LPR;
                const long off = (mrow0 + (long)i * p.W + r) * p.ldr + ncol;
                const bool in = act && r < 16;
                if constexpr (RMODE == 1) {
                    r16[slot][it] = V8{};
                    if (in) r16[slot][it] = *reinterpret_cast<const V8*>(res + off);
                } else if constexpr (RMODE == 2) {
                    r32[slot][it][0] = r32[slot][it][1] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (in) {
                        r32[slot][it][0] = *reinterpret_cast<const float4*>(res32 + off);
                        r32[slot][it][1] = *reinterpret_cast<const float4*>(res32 + off + 4);
                    }
                }
            }
        };
        if constexpr (RMODE != 0) {
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) load_res(i, i);
        }
        // bias, then row bias, summed into the accumulators -- (acc + bias) + rowbias, as in gemm.hip; absent terms are zeros
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[j][i][0] = (acc[j][i][0] + bj[j].x) + rbj[j].x; acc[j][i][1] = (acc[j][i][1] + bj[j].y) + rbj[j].y;
                acc[j][i][2] = (acc[j][i][2] + bj[j].z) + rbj[j].z; acc[j][i][3] = (acc[j][i][3] + bj[j].w) + rbj[j].w;
            }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): from here on only stores are in flight
        // Output rows: one pointer per lane (pixel rrow of the wave's first image row) + a wave-uniform offset per (i, it).  A
        // launch that is one output-parity phase of conv(nearest x2 upsample) writes pixel (oy, ox) of its grid to pixel
        // (2 oy + py, 2 ox + px) of the 2H x 2W output: image rows are 4W pixels apart there, pixels 2.
        const bool phased = p.out_phase != 0;
        const long brow = phased ? ((long)img * 2 * p.H + 2 * (ty0 + wm * 4) + ((p.out_phase >> 1) & 1)) * (2 * p.W) + 2 * (tx0 + rrow) + (p.out_phase & 1)
                                 : mrow0 + rrow;
        const long step_i = phased ? 4L * p.W : (long)p.W;
        const int step_r = phased ? 2 : 1;
        E* pC = Cout + brow * p.ldc + ncol;
        float* pC32 = C32 + brow * p.ldc32 + ncol;
        const float* lrd = scr + rch * 8;
        // HALF (no residual, no fp32 carrier): the accumulators are rounded before the transpose and cross LDS as 16-bit values --
        // see gemm.hip's wide epilogue
        constexpr int SPH = WN * 2 + 16;
        using V4 = typename TT::v4;
        auto passes = [&](auto half_tag) {
            constexpr bool HALF = decltype(half_tag)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                {
                    float* srow = scr + fr * SP + fq * 4;
                    unsigned char* hrow = reinterpret_cast<unsigned char*>(scr) + fr * SPH + fq * 8;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        if constexpr (HALF) *reinterpret_cast<V4*>(hrow + j * 32) = V4{from_f32<E>(acc[j][i][0]), from_f32<E>(acc[j][i][1]), from_f32<E>(acc[j][i][2]), from_f32<E>(acc[j][i][3])};
                        else *reinterpret_cast<float4*>(srow + j * 16) = make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
                    }
                }
                // LDS operations of one wave execute in order: the reads below see the writes above.  Every lane reads (rows clamped
                // into the scratch), only the stores are predicated: the RI x 2 reads of the pass go out back to back.
                float4 x[HALF ? 1 : RI][2];
                V8 xh[HALF ? RI : 1];
#pragma unroll
                for (int it = 0; it < RI; ++it) {
                    const int rc = min(rrow + it * LPR, 15);
                    if constexpr (HALF) {
                        xh[it] = *reinterpret_cast<const V8*>(reinterpret_cast<const unsigned char*>(scr) + rc * SPH + rch * 16);
                    } else {
                        x[it][0] = *reinterpret_cast<const float4*>(lrd + rc * SP);
                        x[it][1] = *reinterpret_cast<const float4*>(lrd + rc * SP + 4);
                    }
                }
#pragma unroll
                for (int it = 0; it < RI; ++it) {
                    if (act && rrow + it * LPR < 16) {
                        float v[8];
                        V8 o;
                        if constexpr (HALF) {
                            o = xh[it];
                        } else {
                            const float4 x0 = x[it][0], x1 = x[it][1];
                            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
                        }
                        if constexpr (RMODE == 1) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += to_f32(r16[i % DEPTH][it][e]);
                        }
                        if constexpr (RMODE == 2) {
                            const float4 a = r32[i % DEPTH][it][0], b = r32[i % DEPTH][it][1];
                            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
                        }
                        if constexpr (!HALF) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = from_f32<E>(v[e]);
                        }
                        const long roff = i * step_i + (long)(it * LPR) * step_r;   // wave-uniform
                        if (Cout) *reinterpret_cast<V8*>(pC + roff * p.ldc) = o;
                        if (!HALF && C32) {
                            float* d32 = pC32 + roff * p.ldc32;
                            *reinterpret_cast<float4*>(d32) = make_float4(v[0], v[1], v[2], v[3]);
                            *reinterpret_cast<float4*>(d32 + 4) = make_float4(v[4], v[5], v[6], v[7]);
                        }
                        if (colstats) {
                            if (!HALF && C32) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) { s8[e] += v[e]; q8[e] = fmaf(v[e], v[e], q8[e]); }
                            } else {
#pragma unroll
                                for (int e = 0; e < 8; ++e) { const float f = to_f32(o[e]); s8[e] += f; q8[e] = fmaf(f, f, q8[e]); }
                            }
                        }
                    }
                }
                if constexpr (RMODE != 0 && DEPTH < 4) {
                    if (i + DEPTH < 4) load_res(i + DEPTH, i % DEPTH);
                }
            }
        };
        if (RMODE == 0 && !C32 && !(p.flags & GEMM_F32_TRANSPOSE)) passes(std::integral_constant<bool, RMODE == 0>{});
        else passes(std::false_type{});
        if (colstats) {
            // fold the LPR row-lanes of every channel through the scratch (fixed order: reproducible).  The wave's 64 pixels
            // (4 image rows x 16) are one statistics slice; slices only have to lie inside one sample and be numbered
            // uniquely within it: slice = (tile index inside the sample) * 4 + wm, phase launches interleave by phase
            if (act) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    scr[(rrow * WN + rch * 8 + e) * 2] = s8[e];
                    scr[(rrow * WN + rch * 8 + e) * 2 + 1] = q8[e];
                }
            }
            const int spp = (p.H * p.W) >> 6;      // slices per sample of this launch's (phase) grid
            long slice = (long)trem * 4 + wm;
            slice = p.out_phase ? ((long)img * 4 + (p.out_phase & 3)) * spp + slice : (long)img * spp + slice;
            for (int c = lane; c < WN; c += 64) {
                float ss = 0.f, qq = 0.f;
                for (int l = 0; l < LPR; ++l) { ss += scr[(l * WN + c) * 2]; qq += scr[(l * WN + c) * 2 + 1]; }
                const int n = n0 + wn * WN + c;
                if (n < p.N) *reinterpret_cast<float2*>(colstats + (slice * p.ld_colstats + n) * 2) = make_float2(ss, qq);
            }
        }
    };
    if constexpr (RM >= 0) epilogue(std::integral_constant<int, RM>{});
    else if (p.res_f32) epilogue(std::integral_constant<int, 2>{});
    else if (p.residual) epilogue(std::integral_constant<int, 1>{});
    else epilogue(std::integral_constant<int, 0>{});
    if constexpr (DIAG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long d_t3 = stamp();
        if (lane == 0 && diag_out) {
            float* d = diag_out + ((long)blockIdx.x * 8 + wave) * 8;
            d[0] = (float)(d_t1 - d_t0); d[1] = (float)(d_t2 - d_t1); d[2] = (float)(d_t3 - d_t2);
            d[3] = (float)d_sync; d[4] = (float)d_issue; d[5] = (float)d_comp; d[6] = (float)T; d[7] = 0.f;
        }
    }
}

template <class TT, int NT, int KH, int KW>
int launch_patch(const GemmParams& p, hipStream_t stream) {
    constexpr int BN = 32 * NT, TAPS = KH * KW, NSLOT = (TAPS % 3 == 0) ? 3 : 4;
    constexpr int NPIECES = ((TP + KW - 1) * (TP + KH - 1) + 7) / 8;
    const size_t lds = 2 * (size_t)NPIECES * 1024 + NSLOT * (size_t)BN * 128 + (KH == 3 ? 2048 : 0);
    const int rm = p.res_f32 ? 2 : (p.residual ? 1 : 0);
    // Every instantiation built here keeps its values in registers (0 B scratch: tools/codeobj_audit.py, tests/test_host_cpu.py).  Three
    // forms did not at 160 channels per tile and are NOT built: the 2 x 2 window with a residual operand (12 / 16 B) and the fused
    // GroupNorm (36 B) -- vf_conv_patch_tile hands such launches the 128-wide tile or none (the im2col kernel / an error for gn_ab).
    auto kern = conv_patch_kernel<TT, NT, KH, KW, false, false, 0>;
    if constexpr (!(KH == 2 && NT == 5)) {
        if (rm == 2) kern = conv_patch_kernel<TT, NT, KH, KW, false, false, 2>;
        else if (rm == 1) kern = conv_patch_kernel<TT, NT, KH, KW, false, false, 1>;
    } else if (rm) return VF_ERR_SHAPE;
    int which = rm;
    if constexpr (NT == 5 && KH == 3 && sizeof(typename TT::elem) == 2) {
        if (p.flags & 0x4000) { kern = conv_patch_kernel<TT, NT, KH, KW, true>; which = 3; }   // diagnostic stamps (tools/stamp_conv.py)
    }
    if constexpr (KH == 3 && NT == 4) {
        if (p.gn_ab) { kern = conv_patch_kernel<TT, NT, KH, KW, false, true>; which = 4; }       // fused GroupNorm-apply + SiLU
    } else if (p.gn_ab) return VF_ERR_SHAPE;
    static VfOncePerDevice attr_set[5];
    if (!attr_set[which].set_lds(reinterpret_cast<const void*>(kern), (int)lds)) return VF_ERR_LAUNCH;
    const int ntm = (p.M / (p.OH * p.OW)) * (p.H / TP) * (p.W / TP), ntn = p.N / BN;
    // Column-group width of the tile order.  An XCD (own L2) works through tiles_xcd = ntm * ntn / 8 consecutive tiles = a block
    // of (tiles_xcd / GN) m-tiles x GN n-tiles, and fetches that block's patches and weight panels once: bytes per XCD =
    // tiles_xcd / GN * A_mt + GN * W_nt, least at GN = sqrt(tiles_xcd * A_mt / W_nt).  On the 16x16 level (weight panels of 3 MB
    // against 0.65 MB patches) the old fixed GN = 8 made every XCD fetch 80 % of the weights: 256 MB per launch measured.
    GemmParams q = p;
    {
        const double a_mt = (double)(TP + KW - 1) * (TP + KH - 1) * (p.Cin + (p.A2 ? (p.K - p.K1) : 0)) * 2.0;
        const double w_nt = (double)BN * p.K * 2.0;
        const double tiles_xcd = (double)ntm * ntn / 8.0;
        int gn = (int)(__builtin_sqrt(tiles_xcd * a_mt / w_nt) + 0.5);
        q.tile_group = gn < 1 ? 1 : (gn > ntn ? ntn : gn);
    }
    hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(512), lds, stream, q);
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

template <class TT>
int launch_patch_dtype(const GemmParams& p, int bn, hipStream_t stream) {
    if (p.KH == 3 && p.KW == 3) return bn == 160 ? launch_patch<TT, 5, 3, 3>(p, stream) : launch_patch<TT, 4, 3, 3>(p, stream);
    if (p.KH == 2 && p.KW == 2) return bn == 160 ? launch_patch<TT, 5, 2, 2>(p, stream) : launch_patch<TT, 4, 2, 2>(p, stream);
    return VF_ERR_SHAPE;
}

template <class TT>
int launch_q8(const GemmParams& p, hipStream_t stream) {
    constexpr int NT = 4, BN = 128, NPIECES = 50;
    const size_t lds = 2 * (size_t)NPIECES * 1024 + 3 * (size_t)BN * 128 + 2048;
    auto kern = conv_patch_kernel<TT, NT, 3, 3, false, false, 0, true>;
    static VfOncePerDevice attr_set;
    if (!attr_set.set_lds(reinterpret_cast<const void*>(kern), (int)lds)) return VF_ERR_LAUNCH;
    const int nimg = p.M / 64, ntm = (nimg + 3) / 4, ntn = p.N / BN;
    hipLaunchKernelGGL(kern, dim3(ntm * ntn * p.split_k), dim3(512), lds, stream, p);
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

}  // namespace

// The 8 x 8 level through the patch-staged kernel (Q8 form): 0 = not such a launch, else the K split (number of fp32 partial
// tiles per output tile).  Four images per workgroup and N / 128 channel tiles give (nimg / 4) * (N / 128) workgroups -- 120 at
// the nominal 48-sample batch of the UNet's 1280-channel level -- so K is split over channel chunks until the one-per-CU grid
// is about one round; the split follows the NOMINAL batch (the fp32 summation order must not depend on the launch's batch).
// Nominal = 48 samples since round 6, like the tile-width rule below (one launch stream's half of the 32-frame headline, a rank's
// 16-frame share at N > 1): K in two shares instead of four -- half the prologues, epilogues and fp32 partials.  Same-box A/B
// (profiles/r06_i): 32 frames 76.22 -> 75.83 ms/step (conv family 31.65 -> 30.96); the 8-frame clip pays for it, 20.0 -> 20.6.
int vf_conv_q8_split(const GemmParams& p) {
    if (p.mode != 1 || p.stride != 1 || p.upsample || p.out_phase || (p.Cin & 63) || p.gn_ab) return 0;
    if (p.H != 8 || p.W != 8 || p.OH != 8 || p.OW != 8 || p.KH != 3 || p.KW != 3 || p.ntaps != 9 || p.pad != 1 || p.pad_x != 1) return 0;
    if ((p.flags & (GEMM_GEGLU | GEMM_OUT_F32 | GEMM_NARROW_EPILOGUE | GEMM_NO_PATCH | 0x4000 | (0xF << 8))) || (p.N % 128)) return 0;
    if (p.A2 && ((p.K - p.K1) & 63)) return 0;
    if (p.rows_per_sample != 64 && p.rowbias) return 0;
    const int nchunks = p.Cin >> 6;
    const long tiles24 = 12L * (p.N / 128);      // (the name is history: the nominal batch is 48 samples = 12 image quads)
    int s = (int)(256 / tiles24);
    if (s > 8) s = 8;
    if (s > nchunks / 2) s = nchunks / 2;       // at least two chunks (18 K tiles) per share
    return s < 1 ? 0 : s;
}

int vf_launch_conv_q8(const GemmParams& p, int dtype, hipStream_t stream) {
    if (dtype == VF_DTYPE_F16) return launch_q8<F16>(p, stream);
    if (dtype == VF_DTYPE_BF16) return launch_q8<BF16>(p, stream);
    return VF_ERR_DTYPE;
}

namespace {
}  // namespace

// THE rule for which kernel a convolution launch runs -- 0: gemm.hip's implicit GEMM, 1: the patch-staged kernel (*arg = channel-tile
// width), 2: its 8x8 form (*arg = K split; the launcher additionally needs the caller's split-K workspace, which every wrapper
// passes).  vf_launch_gemm dispatches on it and vface_conv_uses_patch_kernel (what bench.py prices launches by and the engine
// decides GroupNorm fusion by) reports it: one copy, so the two cannot drift apart.  Geometry and flags only, grid depth at the
// nominal 24-sample batch: a sample's bits never depend on which other samples share its launch.
int vf_conv_kernel_choice(const GemmParams& p, int* arg) {
    if (arg) *arg = 0;
    if (p.mode != 1 || ((p.flags >> 8) & 0xF) || (p.flags & GEMM_NO_PATCH)) return 0;
    const int bn = vf_conv_patch_tile(p);
    if (bn) {
        const long tiles24 = 24L * (p.H / TP) * (p.W / TP) * (p.N / bn);
        if (tiles24 >= 160 || (p.flags & GEMM_PATCH)) { if (arg) *arg = bn; return 1; }
    }
    if (!(p.flags & GEMM_NO_Q8)) {
        const int s = vf_conv_q8_split(p);
        if (s) { if (arg) *arg = s; return 2; }
    }
    return 0;
}

// 0 = this launch is not a patch-kernel shape; else the channel-tile width (160 | 128) the launch would use
int vf_conv_patch_tile(const GemmParams& p) {
    if (p.mode != 1 || p.stride != 1 || p.upsample || (p.Cin & 63)) return 0;
    if (p.OH != p.H || p.OW != p.W || (p.H % TP) || (p.W % TP)) return 0;
    if (!((p.KH == 3 && p.KW == 3) || (p.KH == 2 && p.KW == 2)) || p.ntaps != p.KH * p.KW) return 0;
    if (p.pad < 0 || p.pad_x < 0 || p.pad > 1 || p.pad_x > 1) return 0;
    if ((p.flags & (GEMM_GEGLU | GEMM_OUT_F32 | GEMM_NARROW_EPILOGUE)) || (p.N & 7)) return 0;
    if ((p.flags & 0x4000) && !(p.KH == 3 && p.N % 160 == 0 && p.colstats)) return 0;   // diagnostic build: one instantiation
    if (p.A2 && (((p.K - p.K1) & 63) || p.KH != 3)) return 0;
    if (p.gn_ab && (p.KH != 3 || (p.ld_gn_ab < p.Cin) || ((uintptr_t)p.gn_ab & 15) || (p.flags & 0x4000))) return 0;
    if (p.colstats && ((p.H * p.W) & 63)) return 0;
    if (p.residual && !p.res_f32 && (((uintptr_t)p.residual & 15) || (p.ldr & 7))) return 0;
    if (((uintptr_t)p.C & 15) || (p.C && (p.ldc & 7))) return 0;
    const bool ok160 = p.N % 160 == 0, ok128 = p.N % 128 == 0;
    if (p.gn_ab) return ok128 ? 128 : 0;   // the fused-normalisation build exists 128 wide only (the 160-wide one spilled 36 B)
    if (p.KH == 2 && (p.residual || p.res_f32)) return ok128 ? 128 : 0;      // ... and so does the 2 x 2 window with a residual operand
    if (ok160 && ok128 && !(p.flags & (GEMM_PATCH_BN160 | 0x4000))) {
        // Both widths tile N (640, 1280, 1920 channels): one workgroup per CU, so a launch takes ceil(workgroups / CUs) rounds, and
        // a 128-wide workgroup takes ~0.8 of a 160-wide one (32 instead of 40 MFMAs per wave and K tile, same fixed costs).
        // Measured (tools/quantisation_probe.py, 24 samples): 32x32 640->640: 160-wide 384 workgroups = 1.5 rounds 181.6 us,
        // 128-wide 480 = 1.9 rounds 153.6 us; 16x16 1280->1280: 192 = 0.75 rounds 175.3 us vs 240 = 0.94 rounds 151.8 us.
        // The grid is taken at a NOMINAL batch on 256 CUs -- the width changes the summation order of the column statistics, so it
        // must not depend on how many samples share the launch (batch invariance).  Nominal = 48 samples since round 5:
        // one launch stream's half of the 32-frame headline and a rank's 16-frame share at N > 1 (at 24 -- the 8-frame clip --
        // the 32 x 32 640-channel launches would take the 128-wide tile: 15 % faster THERE, 6 % slower at 48 and 96 samples;
        // measured on the headline: 79.34 -> 79.10 ms/step, conv family 33.4 -> 32.8, profiles/r05_o).
        const long t24 = 48L * (p.H / TP) * (p.W / TP);
        const long r160 = (t24 * (p.N / 160) + 255) / 256, r128 = (t24 * (p.N / 128) + 255) / 256;
        return (r128 * 4 < r160 * 5) ? 128 : 160;      // r128 * 0.8 < r160
    }
    if (ok160) return 160;
    if (ok128) return 128;
    return 0;
}

// Plain GEMM C[M, N] = A[M, K] Wt[N, K]^T through the patch-staged kernel's 256 x BN tile (8 waves, one workgroup per CU): the
// kernel's "fused 1x1 source" K loop IS a GEMM K loop -- 256 consecutive rows of A per workgroup as a halo-free 16 x 16 "image",
// no window tiles at all.  Versus gemm.hip's 128-row tile: 28 % fewer L2->LDS bytes per FLOP (measured where it pays: DESIGN 4).
// 0 = not a shape this form takes; else the channel-tile width.
int vf_gemm_patch_tile(const GemmParams& p) {
    if (p.mode != 0 || p.A2 || p.out_phase || p.gn_ab || p.split_k > 1) return 0;
    if ((p.M % (TP * TP)) || (p.K & 63) || p.K < 128) return 0;
    if ((p.flags & (GEMM_GEGLU | GEMM_OUT_F32 | GEMM_NARROW_EPILOGUE | 0x4000)) || (p.N & 7)) return 0;
    if (p.rowbias && (p.rows_per_sample <= 0 || (p.rows_per_sample % (TP * TP)))) return 0;
    if (p.residual && !p.res_f32 && (((uintptr_t)p.residual & 15) || (p.ldr & 7))) return 0;
    if (((uintptr_t)p.C & 15) || (p.C && (p.ldc & 7))) return 0;
    if (p.N % 160 == 0) return 160;
    if (p.N % 128 == 0) return 128;
    return 0;
}

int vf_launch_gemm_patch(const GemmParams& p_in, int dtype, hipStream_t stream) {
    const int bn = vf_gemm_patch_tile(p_in);
    if (!bn) return VF_ERR_SHAPE;
    GemmParams p = p_in;
    // the geometry the kernel derives its tile origin and output rows from: M / 256 images of 16 x 16 pixels, no window
    p.H = p.W = p.OH = p.OW = TP; p.Cin = 0; p.K1 = 0; p.KH = p.KW = 3; p.ntaps = 9; p.pad = p.pad_x = 0; p.stride = 1; p.upsample = 0;
    p.A2 = p.A; p.lda2 = p.lda; p.a2_bytes = p.a_bytes; p.a2_row_mod = 0;
    if (dtype == VF_DTYPE_F16) return bn == 160 ? launch_patch<F16, 5, 3, 3>(p, stream) : launch_patch<F16, 4, 3, 3>(p, stream);
    if (dtype == VF_DTYPE_BF16) return bn == 160 ? launch_patch<BF16, 5, 3, 3>(p, stream) : launch_patch<BF16, 4, 3, 3>(p, stream);
    return VF_ERR_DTYPE;
}

int vf_launch_conv_patch(const GemmParams& p, int dtype, hipStream_t stream) {
    const int bn = vf_conv_patch_tile(p);
    if (!bn) return VF_ERR_SHAPE;
    if (dtype == VF_DTYPE_F16) return launch_patch_dtype<F16>(p, bn, stream);
    if (dtype == VF_DTYPE_BF16) return launch_patch_dtype<BF16>(p, bn, stream);
    return VF_ERR_DTYPE;
}
