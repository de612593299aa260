// Fused FeedForward of a BasicTransformerBlock for the level-0 (C = 320) token matrices of the VFace UNet (gfx950):
//
//     out = ff.net[2]( GEGLU( ff.net[0].proj( LayerNorm(x) ) ) ) + x          (REFace/ldm/modules/attention.py:37-64, 231-243:
//                                                                               x = ff(norm3(x)) + x)
//
// ONE launch instead of LayerNorm + GEMM(GEGLU) + GEMM(+residual), and the [M x 4C] hidden matrix (252 MB per level-0 block at
// F = 8) never exists: a workgroup keeps its 128 tokens' normalised activations IN REGISTERS as MFMA B operands for the whole
// kernel (activation-stationary) and streams the two weight matrices through LDS once.
//
//   * 256 threads = 4 waves, ONE wave per SIMD with the whole 512-register file (`__launch_bounds__(256, 1)`); a wave owns 32
//     tokens = one column tile of `mfma_f32_32x32x16`.  With one wave per SIMD nothing hides a wave's own stalls, and the
//     kernel is bound by INSTRUCTION ISSUE and by what is NOT overlapped (stamps of the earlier chunk-at-a-time versions: 9 100
//     cycles per 64 hidden units for 3 840 cycles of MFMA, 2 900 of them the GEGLU standing alone between the two GEMMs).
//     Hence the 32 x 32 shape (half the MFMA instructions, 32-cycle gaps for other work) and the software pipeline below.
//   * Prologue: the wave reads its tokens' fp32 rows in the B-operand lane layout (lane = token l & 31, half h = l >> 5 holds
//     channels 16 s + 8 h .. + 7 of every k16 step s), does the two-pass LayerNorm across its two lane halves, and keeps
//     fp16(LN(x)) as C / 16 fragments (80 registers).
//   * Unit of work = one hidden TILE: 16 hidden units = 32 rows of the GEGLU-interleaved ff.net[0] weight (16 value rows, then
//     their 16 gate rows): in the 32 x 32 accumulator layout (row = (reg & 3) + 8 (reg >> 2) + 4 h) register q < 8 is a value row
//     and register q + 8 ITS gate row, in the same lane.  Interval t of the pipeline issues, interleaved in one stream:
//        GEMM 1 of tile t      acc1[t & 1] = W1[tile t] LN(x)        C / 16 MFMAs, one accumulation chain, A fragments from LDS
//        GEGLU of tile t - 1   hb = value * gelu(gate) (erf form)    8 values per lane, cut into 24 pieces, one per MFMA gap
//        GEMM 2 of tile t - 2  out[C / 32 tiles] += W2[:, tile] hb   C / 32 MFMAs: the eight GEGLU results of a lane, rounded to
//                              16 bits, ARE the B operand of a k16 step (guide 3, "an accumulator tile as the next MFMA's
//                              operand": element j of lane half h is hidden unit 8 (j >> 2) + 4 h + (j & 3) of the tile), so
//                              ff.net[2]'s weight columns are stored in THAT order (packing.pack_ffn_w2) and no lane ever moves.
//     Both accumulator sets are pinned by inline-asm MFMAs (acc1 in vector registers -- the GEGLU reads it --, out in
//     accumulator registers): left to hipcc, loop-carried 16-register tiles were renamed at ~190 v_accvgpr copies per 64 hidden.
//   * Weights: W1 tile stages [32 rows x C] (20 KB), LDS-DMA, ring of four, landed ONE interval ahead so a rolling window of A
//     fragments runs across interval boundaries; W2 stages [C rows x 32 hidden] (20 KB, two tiles), ring of three, issued every
//     other interval, the DMA instructions spread one at a time over the interval's k loop.  Static schedule, counted
//     `s_waitcnt vmcnt(N)` + raw `s_barrier` once per interval; the issue continues past the end (wrapping to stages nobody
//     reads) so the count never changes.  No ordinary global load lives inside the loop (hipcc would drain the DMA queue for
//     it): ff.net[0]'s bias and LayerNorm's gamma / beta are staged in LDS.
//   * Measured: 328-336 us at M = 98 304, C = 320 against ~450 us for the three launches it replaces; the loop
//     runs at ~2 300 cycles per interval for 960 cycles of MFMA -- ablations remove the GEGLU, the fragment reads and GEMM 2's
//     MFMAs at ~15 / 6 / 13 % each, ADDITIVELY, in every arrangement of the stream tried (chunked, pipelined, piecewise).
//   * Epilogue: (out + b2) + x in fp32, transposed through LDS (the rings are free by then) to whole 32-byte row chunks; the
//     residual rows of a half are requested together.
//
// L2 -> LDS bytes per FLOP are half those of the 128 x 160 tile GEMM (the activations are never re-staged): 7.6 B / kFLOP.
#include <type_traits>

#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

constexpr int ffn_ring(int n, int want) { for (int r = want; r > 1; --r) if (n % r == 0) return r; return 1; }

// PRE (round 4): the attn1 OUT-PROJECTION in front of it, in the same launch (attention.py:239-243: x = attn1(norm1(x)) + x;
// x = attn2(norm2(x), context) + x; x = ff(norm3(x)) + x -- attn2 being a per-sample row bias here):
//     t1  = to_out(att) + bo + a2[sample] + t0                     (the running sum the three-kernel path kept in HBM as fp32)
//     out = ff.net[2]( GEGLU( ff.net[0].proj( LayerNorm(t1) ) ) ) + t1
// The weight stream simply starts NOT stages earlier -- p.W1 = [to_out ; ff.net[0]] rows -- and those stages accumulate
// to_out's output-channel tiles into the SAME accumulator registers GEMM 2 later adds to: they are initialised with
// t0 + bo + a2 and hold t1 when the LayerNorm reads them, so neither t1's 126 MB write nor its two 126 MB reads (LayerNorm
// input, residual) happen, and one launch goes.  The LayerNorm'd tile, register by register, is GEMM 1's B operand (as in
// stfront.hip), so ff.net[0]'s k columns are stored in ffn_w2_perm order for this form.
//
// POST (round 4, with PRE): the SpatialTransformer's proj_out BEHIND it (attention.py:286-289: x = proj_out(x); return x + x_in),
// same launch:
//     t3  = out (above), rounded to 16 bits -- the operand the separate GEMM read from HBM
//     y   = proj_out(t3) + b_po + x_in                  (+ the per-64-row column statistics the next GroupNorm reads)
// C / 32 more stages at the END of the W1 stream ([to_out ; ff.net[0] ; proj_out] rows, proj_out's k columns in ffn_w2_perm
// order: the rounded accumulator tiles are the B operand, register by register, as after the LayerNorm).  The accumulators are
// re-initialised with x_in (loaded straight into the accumulator registers) once t3 has left them, so t3's 63 + 126 MB of
// writes, proj_out's 63 MB operand read and a 91-us launch go.
template <class TT, int C, bool PRE = false, bool POST = false>
__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(FfnParams p) {
    static_assert(!POST || PRE, "the proj_out stage exists in the block-tail form only");
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    constexpr int KS = C / 16;          // k16 steps of GEMM 1 = MFMAs per tile
    constexpr int NOT = C / 32;         // output row tiles of GEMM 2 = its MFMAs per tile
    constexpr int NT = 4 * C / 16;      // hidden tiles
    constexpr int NPRE = PRE ? C / 32 : 0;   // to_out stages in front of the W1 stream
    constexpr int NPOST = POST ? C / 32 : 0; // proj_out stages behind it
    constexpr int NSTREAM = NT + NPRE + NPOST;
    constexpr int NW1 = 4, NW2 = 3;     // stage rings
    constexpr int W1E = 32 * C;         // elements per W1 stage: [32 rows][C]
    constexpr int W2E = C * 32;         // elements per W2 stage: [C rows][32 hidden]
    constexpr int OPS = C / 64;         // LDS-DMA instructions per wave per stage (either kind): 1 KB each
    constexpr int RA = ffn_ring(KS, 5); // A-fragment window of GEMM 1 (reads run RA steps ahead, across interval boundaries)
    constexpr int RB = ffn_ring(NOT, 5);// the same for GEMM 2
    static_assert(C % 64 == 0 && C <= 320, "C: a multiple of 64, at most 320 (accumulators of C / 32 output tiles per wave)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E* sW1 = reinterpret_cast<E*>(smem_raw);
    E* sW2 = sW1 + NW1 * W1E;
    float* sB1 = reinterpret_cast<float*>(sW2 + NW2 * W2E);   // ff.net[0] bias, [8 C] fp32 in packed row order
    float* sGB = sB1 + 8 * C;                                 // gamma [C], beta [C]
    float* sBS = sGB + 2 * C;                                 // PRE: to_out bias + attn2 row bias of this workgroup's sample [C]
    float* sBP = sBS + C;                                     // POST: ff.net[2]'s bias + sBS [C]

    const int t_ = threadIdx.x, lane = t_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t_ >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const long tok0 = (long)blockIdx.x * 128 + wave * 32;

    // ---- bias of GEMM 1, gamma, beta into LDS (ordinary loads: before any LDS-DMA is in flight)
    for (int i = t_; i < 8 * C / 4; i += 256) reinterpret_cast<float4*>(sB1)[i] = reinterpret_cast<const float4*>(p.b1)[i];
    for (int i = t_; i < C / 4; i += 256) {
        reinterpret_cast<float4*>(sGB)[i] = reinterpret_cast<const float4*>(p.gamma)[i];
        reinterpret_cast<float4*>(sGB + C)[i] = reinterpret_cast<const float4*>(p.beta)[i];
    }

    // ---- LayerNorm (attention.py:233 norm3, eps 1e-5, fp32, two passes like layernorm_kernel) straight into B fragments
    V8 xf[KS];
    f16_t out[NOT];
    if constexpr (PRE) {
        // the attention output's rows ARE B fragments (16-bit, k = channel); the accumulators start as t0 + bo + a2[sample]
        const E* ar = reinterpret_cast<const E*>(p.att) + (tok0 + fr) * p.ldatt + fh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const V8*>(ar + ks * 16);
        // the accumulators start as t0 itself, loaded STRAIGHT into the accumulator registers (every load of the tile in flight at
        // once, no vector-register staging); bo + a2[sample] -- one vector per workgroup: rows_per_sample % 128 == 0 -- is staged in
        // LDS (sBS) and added where the values are read: by the LayerNorm below and by the epilogue
        const float* rr = p.resid + (tok0 + fr) * p.ldr + fh * 4;
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 v = *reinterpret_cast<const float4*>(rr + ot * 32 + q4 * 8);
                out[ot][4 * q4] = v.x; out[ot][4 * q4 + 1] = v.y; out[ot][4 * q4 + 2] = v.z; out[ot][4 * q4 + 3] = v.w;
            }
        {
            const float* rb = p.rowbias ? p.rowbias + (long)(tok0 / p.rows_per_sample) * p.ld_rowbias : nullptr;
            for (int i = t_; i < C; i += 256) {
                const float e = p.bo[i] + (rb ? rb[i] : 0.f);
                sBS[i] = e;
                if constexpr (POST) sBP[i] = p.b2[i] + e;
            }
        }
        __syncthreads();                 // sGB (and sB1) written
    } else {
        // every load of the token rows is issued before anything waits for one (hipcc serialises load / use pairs otherwise)
        float v[KS][8];
        const float* xr = p.x32 + (tok0 + fr) * p.ldx + fh * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float4 a = *reinterpret_cast<const float4*>(xr + ks * 16);
            const float4 b = *reinterpret_cast<const float4*>(xr + ks * 16 + 4);
            v[ks][0] = a.x; v[ks][1] = a.y; v[ks][2] = a.z; v[ks][3] = a.w; v[ks][4] = b.x; v[ks][5] = b.y; v[ks][6] = b.z; v[ks][7] = b.w;
        }
        __builtin_amdgcn_sched_barrier(0);
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[ks][j];
        s += __shfl_xor(s, 32, 64);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[ks][j] - mean; q += d * d; }
        q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q / (float)C + p.eps);
        __syncthreads();                 // sGB (and sB1) written
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float* gp = sGB + ks * 16 + fh * 8;
            const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
            const float4 b0 = *reinterpret_cast<const float4*>(gp + C), b1 = *reinterpret_cast<const float4*>(gp + C + 4);
            const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            V8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = from_f32<E>((v[ks][j] - mean) * rstd * gm[j] + bt[j]);
            xf[ks] = o;
        }
    }

    // ---- weight streams.  The LDS image of one DMA instruction is lane-linear (64 sixteen-byte slots); a stage is written in
    // its reading order and the XOR swizzles that make the fragment reads conflict-free are applied on the SOURCE chunk.
    //   W1 stage = [32 rows][C / 8 slots]: slot of k chunk c of row r = (c & ~7) | ((c & 7) ^ ((r >> 1) & 7))
    //   W2 stage = [C rows][4 slots]:      slot of chunk c of row r   = c ^ ((r >> 2) & 3)
    const i32x4_t rW1 = raw_buffer_rsrc(p.W1, (8u * C + (PRE ? C : 0) + (POST ? C : 0)) * C * 2u), rW2 = raw_buffer_rsrc(p.W2p, 4u * C * C * 2u);
    const unsigned ldsW1 = lds_addr_of(sW1), ldsW2 = lds_addr_of(sW2);
    constexpr int SPR = C / 8;          // 16-byte slots per W1 row
    int w1_off[OPS], w2_off[OPS];
#pragma unroll
    for (int i = 0; i < OPS; ++i) {
        const int id = (wave * OPS + i) * 64 + lane;          // slot index inside the stage
        const int r1 = id / SPR, s1 = id - r1 * SPR;
        const int c1 = (s1 & ~7) | ((s1 & 7) ^ ((r1 >> 1) & 7));
        w1_off[i] = (r1 * C + c1 * 8) * 2;
        const int r2 = id >> 2, s2 = id & 3;
        const int c2 = s2 ^ ((r2 >> 2) & 3);
        w2_off[i] = (r2 * 4 * C + c2 * 8) * 2;
    }
    auto issue_w1_op = [&](int tile, int i) {     // piece i of the W1 stage of hidden tile `tile` (wrapped) -> ring slot tile % NW1
        const int base = ((tile % NSTREAM) * 32 * C) * 2;          // (stage index of the stream: PRE puts to_out's tiles first)
        const unsigned dst = ldsW1 + (unsigned)(((tile % NW1) * W1E + wave * OPS * 512) * 2);
        raw_lds_dma16(rW1, dst + i * 1024, w1_off[i], base);
    };
    auto issue_w2_op = [&](int st, int i) {       // piece i of W2 stage st = hidden tiles 2 st, 2 st + 1 (wrapped) -> slot st % NW2
        const int base = ((st % (NT / 2)) * 32) * 2;
        const unsigned dst = ldsW2 + (unsigned)(((st % NW2) * W2E + wave * OPS * 512) * 2);
        raw_lds_dma16(rW2, dst + i * 1024, w2_off[i], base);
    };
    auto issue_w1 = [&](int tile) {
#pragma unroll
        for (int i = 0; i < OPS; ++i) issue_w1_op(tile, i);
    };
    auto issue_w2 = [&](int st) {
#pragma unroll
        for (int i = 0; i < OPS; ++i) issue_w2_op(st, i);
    };
    // Loop-carried scalar state of the two weight streams, advanced by add + wrap at the end of an interval -- written as `tile %
    // NT` / `st % NW2` of the runtime interval index, hipcc rebuilt every offset per DMA piece by magic-number division (40-60
    // scalar instructions per interval, each an issue slot of the one wave this SIMD has).
    //   issue side: the stage an interval issues (W1 stage t + 3; W2 stage t / 2 + 1 in even intervals)
    int w1i_glob = 3 * 32 * C * 2, w1i_slot = 3 % NW1;                 // byte offset into W1 / ring slot of stage t + 3 at t = 0
    int w2i_glob = 1 * 32 * 2, w2i_slot = 1 % NW2;                     // the same for W2 stage 1
    const unsigned ldsW1w = ldsW1 + (unsigned)(wave * OPS * 1024), ldsW2w = ldsW2 + (unsigned)(wave * OPS * 1024);
    auto issue_w1_piece = [&](int i) {
#if !(defined(FFN_ABL) && FFN_ABL == 4)      // (ablation build: no DMA issue inside the loop -- timing only, results garbage)
        raw_lds_dma16(rW1, ldsW1w + (unsigned)(w1i_slot * (W1E * 2) + i * 1024), w1_off[i], w1i_glob);
#endif
    };
    auto issue_w2_piece = [&](int i) {
#if !(defined(FFN_ABL) && FFN_ABL == 4)
        raw_lds_dma16(rW2, ldsW2w + (unsigned)(w2i_slot * (W2E * 2) + i * 1024), w2_off[i], w2i_glob);
#endif
    };
    auto advance_w1_issue = [&]() {
        w1i_glob += 32 * C * 2;
        if (w1i_glob == NSTREAM * 32 * C * 2) w1i_glob = 0;            // past the end: wrap to stages nobody reads
        w1i_slot = (w1i_slot + 1) & (NW1 - 1);
    };
    auto advance_w2_issue = [&]() {
        w2i_glob += 32 * 2;
        if (w2i_glob == (NT / 2) * 32 * 2) w2i_glob = 0;
        w2i_slot = (w2i_slot + 1 == NW2) ? 0 : w2i_slot + 1;
    };
    static_assert((NW1 & (NW1 - 1)) == 0, "W1 ring: a power of two");
    // fragment reads (A operands of the 32x32x16 MFMA): 32 rows x one 16-byte chunk per lane half
    // (the slot of k chunk c = 2 ks + fh is 8 (ks >> 2) + ((2 (ks & 3) + fh) ^ swz): four lane offsets, one per ks & 3, and a
    // compile-time 128-byte step per ks >> 2 -- one address register per (tile, ks & 3) instead of one vector add per read)
    int w1_lane[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w1_lane[j] = fr * C + (((2 * j + fh) ^ ((fr >> 1) & 7)) << 3);
    auto w1_frag_at = [&](const E* const (&base)[4], int ks) -> V8 {
        return *reinterpret_cast<const V8*>(base[ks & 3] + (ks >> 2) * 64);
    };
    // W2: row ot * 32 + fr, chunk (2 (tile & 1) + fh) ^ ((fr >> 2) & 3) -- a lane offset per tile parity and 2 KiB per ot
    int w2_lane[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) w2_lane[par] = fr * 32 + (((2 * par + fh) ^ ((fr >> 2) & 3)) << 3);
    auto w2_frag_at = [&](const E* base, int ot) -> V8 { return *reinterpret_cast<const V8*>(base + ot * 1024); };
    //   read side: ring slot of W1 tile t + 1 (the window reads RA - 1 steps ahead) and of the W2 stage of tile t - 2
    int w1r_slot = 1 % NW1, w2r_slot = 0;
    const E* wa[4];          // W1 fragment bases of tile t ...
    const E* wb[4];          // ... and of tile t + 1
#pragma unroll
    for (int j = 0; j < 4; ++j) { wa[j] = sW1 + w1_lane[j]; wb[j] = sW1 + (1 % NW1) * W1E + w1_lane[j]; }

    if constexpr (!PRE) {
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[ot][r] = 0.f;
    }

    // ---- prime the pipeline: W1 stages 0..2, W2 stage 0
    __syncthreads();                     // nothing else touches LDS before the DMA lands
#pragma unroll
    for (int g = 0; g < 3; ++g) issue_w1(g);
    issue_w2(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();

    V8 af[RA], wf[RB], hbE, hbO;         // hbE / hbO: GEGLU results (GEMM 2 B operands) of even / odd tiles
    f16_t accA, accB;                    // GEMM 1 accumulators of even / odd tiles
#pragma unroll
    for (int i = 0; i < RA; ++i) af[i] = w1_frag_at(wa, i);

    // One interval of the three-deep pipeline: GEMM 1 of tile t (H1) || GEGLU of tile t - 1 (HG) || GEMM 2 of tile t - 2 (H2),
    // one instruction stream: per k16 step one MFMA of GEMM 1, every other step one of GEMM 2, every other step one GEGLU value.
    // `cur` receives tile t's sums, `prev` holds tile t - 1's; `hw` receives tile t - 1's GEGLU, `hr` holds tile t - 2's.
    auto interval = [&](int t, f16_t& cur, f16_t& prev, V8& hw, V8& hr, auto h1_tag, auto hg_tag, auto h2_tag, auto even_tag,
                        auto first_tag, auto last_tag) {
        // (POST, the last interval: the stage it would issue -- proj_out's stage 4 -- belongs in the ring slot of proj_out's stage 0,
        // which nobody has read yet: it issues nothing, and the proj_out loop below counts its own instructions)
        constexpr bool NO_W1_ISSUE = POST && decltype(last_tag)::value;
        constexpr bool H1 = decltype(h1_tag)::value, HG = decltype(hg_tag)::value, H2 = decltype(h2_tag)::value;
        constexpr bool EVEN = decltype(even_tag)::value;
        // (PRE, interval 0: the stage it needs was issued by a to_out interval, which issues no W2 piece -- one instruction less
        // may be in flight than in the steady state)
        constexpr bool FIRST_AFTER_PRE = decltype(first_tag)::value;
        // top of the interval: this wave's share of W1 stage t + 1 has landed (everything but its 2 OPS (= 10) youngest DMA
        // instructions: intervals t - 1 and t - 2 issued that many after it); behind the barrier every wave's has, every wave is
        // done with W1 stage t - 1 (ring slot of stage t + 3) and, in even intervals, with the W2 stage of tiles t - 4, t - 3
        // (issue order inside an even interval is W1 piece, W2 piece, W1 piece, ...: behind the last W1 piece of stage t + 1 come
        // one W2 piece and interval t - 1's OPS pieces when t is even, interval t - 1's 2 OPS pieces when t is odd)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FIRST_AFTER_PRE ? OPS : (EVEN ? OPS + 1 : 2 * OPS)) : "memory");
        raw_barrier();
        // (the stages of this interval -- W1 stage t + 3, in even intervals W2 stage t / 2 + 1 -- are issued one DMA instruction
        // at a time inside the k loop below: a burst of them at the top cost each wave the whole CU's address-unit time)
        float hv[8];
        // GEMM 2's fragment bases: tile t - 2 (parity of t) in W2 ring slot w2r_slot; the window's look-ahead into tile t - 1 is
        // the other parity -- of the same stage in even intervals, of the next stage in odd ones
        const E* w2a = sW2 + w2r_slot * W2E + w2_lane[EVEN ? 0 : 1];
        const E* w2b = sW2 + (EVEN ? w2r_slot : (w2r_slot + 1 == NW2 ? 0 : w2r_slot + 1)) * W2E + w2_lane[EVEN ? 1 : 0];
        if constexpr (HG) gap_mfma_result_to_valu(prev);     // tile t - 1's last MFMA -> the vector reads below
        if constexpr (H1) {
            // ff.net[0]'s bias IS the accumulator's initial value (rows of the 32 x 32 layout: register q -> value row
            // (q & 3) + 8 (q >> 2) + 4 h, register q + 8 its gate row): four broadcast LDS reads straight into the accumulator
            // registers instead of 16 vector adds per tile beside the MFMAs (hipcc packed them into v_pk_add_f32 -- an
            // anti-lever beside MFMAs, guide "price of one filler" -- with a v_mov per operand pair)
            const float* bp = sB1 + t * 32 + fh * 4;
            const float4 v0 = *reinterpret_cast<const float4*>(bp), v1 = *reinterpret_cast<const float4*>(bp + 8);
            const float4 g0 = *reinterpret_cast<const float4*>(bp + 16), g1 = *reinterpret_cast<const float4*>(bp + 24);
            cur[0] = v0.x; cur[1] = v0.y; cur[2] = v0.z; cur[3] = v0.w; cur[4] = v1.x; cur[5] = v1.y; cur[6] = v1.z; cur[7] = v1.w;
            cur[8] = g0.x; cur[9] = g0.y; cur[10] = g0.z; cur[11] = g0.w; cur[12] = g1.x; cur[13] = g1.y; cur[14] = g1.z; cur[15] = g1.w;
        }
        // GEGLU (attention.py:37-45: x, gate = proj(x).chunk(2); x * gelu(gate), erf form) of tile t - 1's eight hidden units per
        // lane, cut into 24 pieces of 5-7 vector instructions -- piece (value q, stage s) goes into the gap behind the (3 q + s)-th
        // MFMA of the interval.  A wave issues in order: an MFMA behind an MFMA waits for the matrix pipe (32 cycles), and only the
        // instructions BETWEEN two MFMAs run beside the first; the earlier layout (two MFMAs, then a whole value's 22 dependent
        // vector instructions) measured as the plain SUM of matrix and vector time.  The arithmetic and its order are those of
        // gelu_erf_f (common.hpp); the bias enters as the accumulator's initial value (below), so a sum differs from the
        // three-kernel path's (acc + bias) in its last fp32 bit at most: the 16-bit results agree to ~1e-5 rel-L2.  Empty asm statements pin each
        // piece in its gap (volatile asm keeps its order; hipcc otherwise packs pieces pairwise into v_pk_*_f32 and hoists them).
        float gG[8], gA[8], gT[8], gE[8], gP[8];
        auto gelu_piece = [&](int idx) {
            if (idx >= 24) return;
            const int q = idx / 3, st = idx % 3;
            if (st == 0) {               // t = 1 / (1 + p |g| / sqrt 2), e = exp(-g^2 / 2)
                float a = prev[q], gt = prev[q + 8];
                asm volatile("" : "+v"(a), "+v"(gt));
                float tt = __builtin_amdgcn_rcpf(fmaf(GELU_P, fabsf(gt), 1.0f));
                float ee = __builtin_amdgcn_exp2f(GELU_K * (gt * gt));
                asm volatile("" : "+v"(tt), "+v"(ee));
                gG[q] = gt; gA[q] = a; gT[q] = tt; gE[q] = ee;
            } else if (st == 1) {        // the half-scaled polynomial
                float tt = gT[q];
                asm volatile("" : "+v"(tt));
                float poly = fmaf(GELU_A5, tt, GELU_A4);
                poly = fmaf(poly, tt, GELU_A3);
                poly = fmaf(poly, tt, GELU_A2);
                poly = fmaf(poly, tt, GELU_A1);
                poly *= tt;
                asm volatile("" : "+v"(poly));
                gP[q] = poly;
            } else {                     // value * (max(g, 0) - |g| u)
                float poly = gP[q];
                asm volatile("" : "+v"(poly));
                float hq = gA[q] * fmaf(-fabsf(gG[q]), poly * gE[q], fmaxf(gG[q], 0.0f));
                asm volatile("" : "+v"(hq));
                hv[q] = hq;
            }
        };
        int gap = 0;                                 // (compile-time after unrolling)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (H1) {
                TT::mfma32x32_vacc(cur, af[ks % RA], xf[ks]);
                // refill the window slot of the PREVIOUS MFMA (a load into the registers the MFMA just issued is still reading
                // waits for it, and everything behind the load with it: measured as matrix + vector time in series): step
                // ks - 1 + RA of this tile, or of the next (its stage has landed)
                af[(ks + RA - 1) % RA] = (ks - 1 + RA < KS) ? w1_frag_at(wa, ks - 1 + RA) : w1_frag_at(wb, ks - 1 + RA - KS);
            }
            if constexpr (!NO_W1_ISSUE) { if (ks % 4 == 1) issue_w1_piece(ks / 4); }      // (KS = 4 OPS)
            if constexpr (EVEN) { if (ks % 4 == 3) issue_w2_piece(ks / 4); }
            if constexpr (HG) {
                if constexpr (KS >= 16) { gelu_piece(gap); ++gap; }
                else { for (int u = 0; u < (24 + KS + KS / 2 - 1) / (KS + KS / 2); ++u) { gelu_piece(gap); ++gap; } }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (H2) {
                if (ks % 2 == 0) {
                    const int ot = ks / 2;           // (KS = 2 NOT)
                    TT::mfma32x32_acc(out[ot], wf[ot % RB], hr);
                    // refill the previous MFMA's slot: tile t - 2's fragment ot - 1 + RB, or the next tile's (tile t - 1: its W2
                    // stage landed intervals ago)
                    wf[(ot + RB - 1) % RB] = (ot - 1 + RB < NOT) ? w2_frag_at(w2a, ot - 1 + RB) : w2_frag_at(w2b, ot - 1 + RB - NOT);
                }
            } else if (ks >= KS - RB && t == 1) {
                wf[ks - (KS - RB)] = w2_frag_at(sW2 + w2_lane[0], ks - (KS - RB));      // interval 1: GEMM 2's window for tile 0 (stage 0, slot 0)
            }
            if constexpr (HG) {
                if (ks % 2 == 0) {
                    if constexpr (KS >= 16) { gelu_piece(gap); ++gap; }
                    else { for (int u = 0; u < (24 + KS + KS / 2 - 1) / (KS + KS / 2); ++u) { gelu_piece(gap); ++gap; } }
                }
            }
            // pin the gap (hipcc otherwise regroups the interleave, which is the point of the pipeline)
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (HG) {
#pragma unroll
            for (int qi = 0; qi < 8; ++qi) hw[qi] = from_f32<E>(hv[qi]);
            gap_valu_result_to_mfma(hw);
        }
        // the streams' state for interval t + 1
        if constexpr (!NO_W1_ISSUE) advance_w1_issue();
        if constexpr (EVEN) advance_w2_issue();
        w1r_slot = (w1r_slot + 1) & (NW1 - 1);                          // ring slot of tile t + 2
#pragma unroll
        for (int j = 0; j < 4; ++j) { wa[j] = wb[j]; wb[j] = sW1 + w1r_slot * W1E + w1_lane[j]; }
        if constexpr (H2 && !EVEN) w2r_slot = (w2r_slot + 1 == NW2) ? 0 : w2r_slot + 1;      // tile t - 1 (even) opens the next W2 stage
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    static_assert(KS == 2 * NOT && NT % 2 == 0 && NT >= 4);
    //        t      cur   prev  hw   hr    H1   HG   H2   EVEN     (tile t -> accA / hbE when t is even)
    if constexpr (PRE) {
        // ---- the out-projection: stage s of the stream = rows 32 s .. of to_out -> accumulator tile s (which already holds
        // t0 + bo + a2); KS MFMAs per stage, the A window rolling on into the next stage, the stage three ahead issued in the loop
#pragma unroll
        for (int ot = 0; ot < NPRE; ++ot) gap_valu_result_to_acc_mfma(out[ot]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xf[ks]));
#pragma unroll
        for (int s_ = 0; s_ < NPRE; ++s_) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");      // stage s_ + 1 landed (behind it: stage s_ + 2's pieces)
            raw_barrier();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                TT::mfma32x32_acc(out[s_], af[ks % RA], xf[ks]);
                af[(ks + RA - 1) % RA] = (ks - 1 + RA < KS) ? w1_frag_at(wa, ks - 1 + RA) : w1_frag_at(wb, ks - 1 + RA - KS);
                if (ks % 4 == 1) issue_w1_piece(ks / 4);
                __builtin_amdgcn_sched_barrier(0);
            }
            advance_w1_issue();
            w1r_slot = (w1r_slot + 1) & (NW1 - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { wa[j] = wb[j]; wb[j] = sW1 + w1r_slot * W1E + w1_lane[j]; }
        }
        // ---- LayerNorm (norm3) of t1 = the accumulator tiles (a token's C channels sit in lanes l, l ^ 32), two passes, fp32;
        // the rounded result, register by register, is GEMM 1's B operand: element e of lane half h of k16 step (tile, j) is
        // channel 32 tile + 16 j + 8 (e >> 2) + 4 h + (e & 3) -- ff.net[0]'s k columns are stored in that order (ffn_w2_perm)
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot) gap_acc_result_to_valu(out[ot]);
        // t1 = accumulator + (bo + a2): register q of tile ot is channel 32 ot + 8 (q >> 2) + 4 fh + (q & 3)
        auto bsum4 = [&](int ot, int q4) -> float4 { return *reinterpret_cast<const float4*>(sBS + ot * 32 + q4 * 8 + fh * 4); };
        float s = 0.f;
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 bb = bsum4(ot, q4);
                s += (out[ot][4 * q4] + bb.x) + (out[ot][4 * q4 + 1] + bb.y) + (out[ot][4 * q4 + 2] + bb.z) + (out[ot][4 * q4 + 3] + bb.w);
            }
        s += __shfl_xor(s, 32, 64);
        const float mean = s / (float)C;
        // (every pass reads the accumulator registers again: kept in vector registers across the passes, the 160 values spill)
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot) asm volatile("" : "+a"(out[ot]));
        float qq = 0.f;
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 bb = bsum4(ot, q4);
                const float d0 = out[ot][4 * q4] + bb.x - mean, d1 = out[ot][4 * q4 + 1] + bb.y - mean;
                const float d2 = out[ot][4 * q4 + 2] + bb.z - mean, d3 = out[ot][4 * q4 + 3] + bb.w - mean;
                qq += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
        qq += __shfl_xor(qq, 32, 64);
        const float rstd = rsqrtf(qq / (float)C + p.eps);
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot) asm volatile("" : "+a"(out[ot]));
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot) {
            const float* gp = sGB + ot * 32 + fh * 4;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                V8 o;
#pragma unroll
                for (int e4 = 0; e4 < 2; ++e4) {
                    const int q4 = 2 * j + e4;
                    const float4 g = *reinterpret_cast<const float4*>(gp + q4 * 8), b = *reinterpret_cast<const float4*>(gp + C + q4 * 8);
                    const float4 bb = bsum4(ot, q4);
                    o[4 * e4 + 0] = from_f32<E>((out[ot][4 * q4 + 0] + bb.x - mean) * rstd * g.x + b.x);
                    o[4 * e4 + 1] = from_f32<E>((out[ot][4 * q4 + 1] + bb.y - mean) * rstd * g.y + b.y);
                    o[4 * e4 + 2] = from_f32<E>((out[ot][4 * q4 + 2] + bb.z - mean) * rstd * g.z + b.z);
                    o[4 * e4 + 3] = from_f32<E>((out[ot][4 * q4 + 3] + bb.w - mean) * rstd * g.w + b.w);
                }
                xf[2 * ot + j] = o;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xf[ks]));
        gap_valu_result_to_mfma(xf[KS - 1]);
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot) gap_valu_result_to_acc_mfma(out[ot]);      // (out goes back to the matrix pipe untouched)
    }
    using P_ = std::integral_constant<bool, PRE>;
    interval(0,      accA, accB, hbO, hbE,  T_{}, F_{}, F_{}, T_{}, P_{}, F_{});
    interval(1,      accB, accA, hbE, hbO,  T_{}, T_{}, F_{}, F_{}, F_{}, F_{});
    for (int t = 2; t < NT; t += 2) {
        interval(t,     accA, accB, hbO, hbE, T_{}, T_{}, T_{}, T_{}, F_{}, F_{});
        interval(t + 1, accB, accA, hbE, hbO, T_{}, T_{}, T_{}, F_{}, F_{}, F_{});
    }
    interval(NT,     accA, accB, hbO, hbE,  F_{}, T_{}, T_{}, T_{}, F_{}, F_{});
    interval(NT + 1, accB, accA, hbE, hbO,  F_{}, F_{}, T_{}, F_{}, F_{}, T_{});

    if constexpr (POST) {
        // ---- proj_out.  Its stages 0..3 (stream stages NPRE + NT ..) were issued by the intervals above; behind this wait and
        // barrier they have landed and every wave is done with the FeedForward's stages.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raw_barrier();
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot) gap_acc_result_to_valu(out[ot]);
        // t3 = accumulator + (b2 + bo + a2), rounded: register by register the B operand (the k order of the LayerNorm'd tile above)
        const float* xin = p.x_in + (tok0 + fr) * p.ld_xin + fh * 4;
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                V8 o;
#pragma unroll
                for (int e4 = 0; e4 < 2; ++e4) {
                    const int q4 = 2 * j + e4;
                    const float4 bb = *reinterpret_cast<const float4*>(sBP + ot * 32 + q4 * 8 + fh * 4);
                    o[4 * e4 + 0] = from_f32<E>(out[ot][4 * q4 + 0] + bb.x);
                    o[4 * e4 + 1] = from_f32<E>(out[ot][4 * q4 + 1] + bb.y);
                    o[4 * e4 + 2] = from_f32<E>(out[ot][4 * q4 + 2] + bb.z);
                    o[4 * e4 + 3] = from_f32<E>(out[ot][4 * q4 + 3] + bb.w);
                }
                xf[2 * ot + j] = o;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xf[ks]));
        // the accumulators start again as x_in (the SpatialTransformer's input, fp32 carrier), straight into the accumulator registers
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 v = *reinterpret_cast<const float4*>(xin + ot * 32 + q4 * 8);
                out[ot][4 * q4] = v.x; out[ot][4 * q4 + 1] = v.y; out[ot][4 * q4 + 2] = v.z; out[ot][4 * q4 + 3] = v.w;
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // x_in is in (one exposed HBM latency per workgroup: ~0.5 % of the kernel)
        gap_valu_result_to_mfma(xf[KS - 1]);
#pragma unroll
        for (int ot = 0; ot < NOT; ++ot) gap_valu_result_to_acc_mfma(out[ot]);
        constexpr int S0 = (NPRE + NT) % NW1;        // ring slot of proj_out's stage 0
        w1r_slot = (S0 + 1) % NW1;
#pragma unroll
        for (int j = 0; j < 4; ++j) { wa[j] = sW1 + S0 * W1E + w1_lane[j]; wb[j] = sW1 + ((S0 + 1) % NW1) * W1E + w1_lane[j]; }
#pragma unroll
        for (int i = 0; i < RA; ++i) af[i] = w1_frag_at(wa, i);
#pragma unroll
        for (int s_ = 0; s_ < NPOST; ++s_) {
            if (s_ > 0) {
                // stage s_ + 1 has landed (issued two stages ago; behind it only stage s_ - 1's OPS pieces) and, behind the barrier,
                // every wave is done with stage s_ - 1 -- the ring slot this stage's pieces (stage s_ + 3) go to
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(OPS) : "memory");
                raw_barrier();
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                TT::mfma32x32_acc(out[s_], af[ks % RA], xf[ks]);
                af[(ks + RA - 1) % RA] = (ks - 1 + RA < KS) ? w1_frag_at(wa, ks - 1 + RA) : w1_frag_at(wb, ks - 1 + RA - KS);
                if (s_ > 0 && ks % 4 == 1) issue_w1_piece(ks / 4);       // (past proj_out's last stage: wrapped, nobody reads it)
                __builtin_amdgcn_sched_barrier(0);
            }
            if (s_ > 0) advance_w1_issue();
            w1r_slot = (w1r_slot + 1) & (NW1 - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { wa[j] = wb[j]; wb[j] = sW1 + w1r_slot * W1E + w1_lane[j]; }
        }
    }

    // ---- epilogue: (out + b2) + x, half the channels of a wave's 32 tokens at a time through LDS (row pitch C / 2 + 4 floats:
    // consecutive tokens one 16-byte slot apart), read back as whole 32-byte row chunks
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();                       // every wave is done with the weight rings: they become scratch
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot) gap_acc_result_to_valu(out[ot]);     // out's last MFMA -> the accumulator reads below
    constexpr int NH = (NOT + 1) / 2;    // output tiles per half
    constexpr int SP = NH * 32 + 4;
    float* scr = reinterpret_cast<float*>(smem_raw) + wave * (32 * SP);
    float* sCS = reinterpret_cast<float*>(smem_raw) + 4 * (32 * SP);      // POST: column (sum, sum of squares) per wave, [4][C][2]
    const float* bias_out = POST ? p.b_po : p.b2;
    E* out16 = reinterpret_cast<E*>(p.out16);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int ot0 = hf * NH, nt = (hf == 0) ? NH : NOT - NH;      // tiles [ot0, ot0 + nt)
        if (nt <= 0) continue;
        constexpr int ITM = (32 * NH * 4 + 63) / 64;                  // items (token, 8-channel chunk) per lane, upper bound
        const int CH = nt * 4;           // 8-channel chunks per row of this half
        // the residual rows of the half are requested before the transposes: one HBM latency per half, not one per item
        float4 r0[ITM], r1[ITM];
#pragma unroll
        for (int k = 0; k < ITM; ++k) {
            const int it = lane + 64 * k;
            if constexpr (PRE) {
                r0[k] = make_float4(0.f, 0.f, 0.f, 0.f); r1[k] = r0[k];      // (the residual t1 is inside the accumulators already)
            } else if (it < 32 * CH) {
                const int tok = it / CH, ch = it - tok * CH;
                const float* xr = p.x32 + (tok0 + tok) * p.ldx + ot0 * 32 + ch * 8;
                r0[k] = *reinterpret_cast<const float4*>(xr);
                r1[k] = *reinterpret_cast<const float4*>(xr + 4);
            }
        }
#pragma unroll
        for (int o = 0; o < NH; ++o) {
            if (o >= nt) continue;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                *reinterpret_cast<float4*>(scr + fr * SP + o * 32 + q4 * 8 + fh * 4) =
                    make_float4(out[ot0 + o][4 * q4], out[ot0 + o][4 * q4 + 1], out[ot0 + o][4 * q4 + 2], out[ot0 + o][4 * q4 + 3]);
        }
        // (LDS operations of one wave execute in order: the reads below see the writes above)
#pragma unroll
        for (int k = 0; k < ITM; ++k) {
            const int it = lane + 64 * k;
            if (it < 32 * CH) {
                const int tok = it / CH, ch = it - tok * CH;
                const float* sp = scr + tok * SP + ch * 8;
                const float4 a0 = *reinterpret_cast<const float4*>(sp), a1 = *reinterpret_cast<const float4*>(sp + 4);
                const int col = ot0 * 32 + ch * 8;
                float4 c0 = *reinterpret_cast<const float4*>(bias_out + col), c1 = *reinterpret_cast<const float4*>(bias_out + col + 4);
                if constexpr (PRE && !POST) {      // (the out-projection's bias + attn2's row bias, held back from the accumulators' initial value)
                    const float4 e0 = *reinterpret_cast<const float4*>(sBS + col), e1 = *reinterpret_cast<const float4*>(sBS + col + 4);
                    c0 = make_float4(c0.x + e0.x, c0.y + e0.y, c0.z + e0.z, c0.w + e0.w);
                    c1 = make_float4(c1.x + e1.x, c1.y + e1.y, c1.z + e1.z, c1.w + e1.w);
                }
                const long row = tok0 + tok;
                const float v[8] = {(a0.x + c0.x) + r0[k].x, (a0.y + c0.y) + r0[k].y, (a0.z + c0.z) + r0[k].z, (a0.w + c0.w) + r0[k].w,
                                    (a1.x + c1.x) + r1[k].x, (a1.y + c1.y) + r1[k].y, (a1.z + c1.z) + r1[k].z, (a1.w + c1.w) + r1[k].w};
                if (out16) {
                    V8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = from_f32<E>(v[e]);
                    *reinterpret_cast<V8*>(out16 + row * p.ldo + col) = o;
                }
                if (p.out32) {
                    float* d = p.out32 + row * p.ldo32 + col;
                    *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
                if constexpr (POST) {
                    if (p.colstats) {      // the finished values go back to the scratch for the column sums below
                        float* sw = scr + tok * SP + ch * 8;
                        *reinterpret_cast<float4*>(sw) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(sw + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                }
            }
        }
        if constexpr (POST) {
            // per-wave column sums of the half (32 tokens, fixed order): lane l owns channels 4 l .. 4 l + 3 of it
            if (p.colstats && lane < nt * 8) {
                float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
                for (int tok = 0; tok < 32; ++tok) {
                    const float4 a = *reinterpret_cast<const float4*>(scr + tok * SP + lane * 4);
                    s4[0] += a.x; s4[1] += a.y; s4[2] += a.z; s4[3] += a.w;
                    q4s[0] = fmaf(a.x, a.x, q4s[0]); q4s[1] = fmaf(a.y, a.y, q4s[1]);
                    q4s[2] = fmaf(a.z, a.z, q4s[2]); q4s[3] = fmaf(a.w, a.w, q4s[3]);
                }
                float* cw = sCS + ((long)wave * C + ot0 * 32 + lane * 4) * 2;
                *reinterpret_cast<float4*>(cw) = make_float4(s4[0], q4s[0], s4[1], q4s[1]);
                *reinterpret_cast<float4*>(cw + 4) = make_float4(s4[2], q4s[2], s4[3], q4s[3]);
            }
        }
    }
    if constexpr (POST) {
        if (p.colstats) {
            // a 64-row slice = two waves' tokens: first wave + second wave, written once
            __syncthreads();
            for (int i = t_; i < 2 * C; i += 256) {
                const int sl = i / C, ch = i - sl * C;
                const float2 a = *reinterpret_cast<const float2*>(sCS + ((2 * sl) * C + ch) * 2);
                const float2 b = *reinterpret_cast<const float2*>(sCS + ((2 * sl + 1) * C + ch) * 2);
                *reinterpret_cast<float2*>(p.colstats + (((long)blockIdx.x * 2 + sl) * p.ld_colstats + ch) * 2) = make_float2(a.x + b.x, a.y + b.y);
            }
        }
    }
}

template <class TT, int C, bool PRE = false, bool POST = false>
int launch_c(const FfnParams& p, hipStream_t stream) {
    constexpr size_t lds = (size_t)(4 * 32 * C + 3 * C * 32) * 2 + (size_t)12 * C * 4;
    static_assert((size_t)4 * 32 * (((C / 32 + 1) / 2) * 32 + 4) * 4 + (size_t)4 * C * 8 <= (size_t)(4 * 32 * C + 3 * C * 32) * 2,
                  "epilogue scratch (+ the column sums) fits the weight rings");
    auto kern = ffn_fused_kernel<TT, C, PRE, POST>;
    static VfOncePerDevice attr_set;
    if (lds > 64 * 1024 && !attr_set.set_lds(reinterpret_cast<const void*>(kern), (int)lds)) return VF_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(p.M / 128), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

template <class TT>
int launch_t(const FfnParams& p, hipStream_t stream) {
    if (p.att && p.Wpo_in_stream) {
        switch (p.C) {
            case 64: return launch_c<TT, 64, true, true>(p, stream);
            case 128: return launch_c<TT, 128, true, true>(p, stream);
            case 320: return launch_c<TT, 320, true, true>(p, stream);
            default: return VF_ERR_SHAPE;
        }
    }
    if (p.att) {
        switch (p.C) {
            case 64: return launch_c<TT, 64, true>(p, stream);
            case 128: return launch_c<TT, 128, true>(p, stream);
            case 320: return launch_c<TT, 320, true>(p, stream);
            default: return VF_ERR_SHAPE;
        }
    }
    switch (p.C) {
        case 64: return launch_c<TT, 64>(p, stream);
        case 128: return launch_c<TT, 128>(p, stream);
        case 320: return launch_c<TT, 320>(p, stream);
        default: return VF_ERR_SHAPE;
    }
}

}  // namespace

bool vf_ffn_fused_supported(long M, int C) { return M > 0 && (M % 128) == 0 && (C == 64 || C == 128 || C == 320); }

int vf_launch_ffn_fused(const FfnParams& p, int dtype, hipStream_t stream) {
    if ((!p.x32 && !p.att) || !p.gamma || !p.beta || !p.W1 || !p.b1 || !p.W2p || !p.b2 || (!p.out16 && !p.out32)) return VF_ERR_ARG;
    if (!vf_ffn_fused_supported(p.M, p.C)) return VF_ERR_SHAPE;
    if (p.att) {      // the out-projection in front: att [M][C] 16-bit, resid [M][C] fp32, bo [C], rowbias [M / rows_per_sample][ld]
        if (!p.resid || !p.bo || (p.rowbias && (p.rows_per_sample <= 0 || (p.rows_per_sample % 128) || (p.ld_rowbias & 3)))) return VF_ERR_ARG;
        if ((p.ldatt & 7) || (p.ldr & 3)) return VF_ERR_ALIGN;
        if (((uintptr_t)p.att | (uintptr_t)p.resid | (uintptr_t)p.bo | (uintptr_t)p.rowbias) & 15) return VF_ERR_ALIGN;
    }
    if (p.Wpo_in_stream) {      // proj_out behind: x_in [M][C] fp32, b_po [C], optional column statistics [M / 64][ld][2]
        if (!p.att || !p.x_in || !p.b_po) return VF_ERR_ARG;
        if ((p.ld_xin & 3) || (p.colstats && (p.ld_colstats & 1))) return VF_ERR_ALIGN;
        if (((uintptr_t)p.x_in | (uintptr_t)p.b_po | (uintptr_t)p.colstats) & 15) return VF_ERR_ALIGN;
    }
    if ((p.x32 && (p.ldx & 3)) || (p.out16 && (p.ldo & 7)) || (p.out32 && (p.ldo32 & 3))) return VF_ERR_ALIGN;
    if (((uintptr_t)p.x32 | (uintptr_t)p.gamma | (uintptr_t)p.beta | (uintptr_t)p.W1 | (uintptr_t)p.b1 | (uintptr_t)p.W2p |
         (uintptr_t)p.b2 | (uintptr_t)p.out16 | (uintptr_t)p.out32) & 15) return VF_ERR_ALIGN;
    if (dtype == VF_DTYPE_F16) return launch_t<F16>(p, stream);
    if (dtype == VF_DTYPE_BF16) return launch_t<BF16>(p, stream);
    return VF_ERR_DTYPE;
}
