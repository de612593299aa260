// Streaming-softmax multi-head attention for the VFace UNet (gfx950): O = softmax(Q K^T * scale) V
// without materialising the [n x n] score matrix the reference builds (attention.py:206-220,
// pnp_utils.py:270-287).  8 heads, head dim 40 / 80 / 160 (8 / 16 / 32 for the test-size UNet).
//
// * One workgroup = 4 waves = 64*QT queries of one (sample, head); each wave owns 16*QT queries.
// * Keys are walked in blocks of 64.  K and V blocks go global -> LDS by LDS-DMA (two LDS buffers, the next
//   block's loads are issued before the current block's math and waited for after it).
// * Scores are computed TRANSPOSED, S^T = K Q^T, with mfma_f32_16x16x32: a lane then holds 16 keys of
//   ONE query, so the row max / row sum are in-lane plus two cross-lane steps, and the fp32 accumulator
//   registers, converted to 16 bit, are directly the B operand of O^T += V^T P^T (no LDS round trip).
//   V^T fragments come from the row-major V block with ds_read_b64_tr_b16.
// * Softmax is fp32 (as autocast keeps it, SURVEY precision map).  Default form: scale*log2e is folded into q and the
//   running reference enters the score MFMA as its C operand, p = exp2(s - m_ref) (see LAZY below); the exact-scale
//   form p = exp2((s - m) * scale*log2e) is kept behind variant bit 1.
// * Sample remapping: output sample b reads q,k of sample qk_map[b] and v of sample v_map[b].  This is
//   how the hook's "replace" / chunks==2 injection (pnp_utils.py:136-142,259-262) and fft_vfixed's
//   V broadcast (:255-256) run without copying anything.
#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

constexpr int KVB = 64;

constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }
constexpr int k_row_elems(int dkp) { return dkp <= 64 ? 64 : (dkp <= 128 ? 128 : dkp + 8); }
// 16-B slot of logical chunk c in row r
template <int KROW> __device__ __forceinline__ int k_slot(int r, int c) {
    if (KROW == 64) return c ^ ((r >> 1) & 7);
    if (KROW == 128) return c ^ (r & 15);
    return c;
}
// V row pitch in bytes: an odd multiple of 32 B so the 8 rows a 32-lane half touches in one
// ds_read_b64_tr_b16 fall in 8 distinct 32-B bank ranges of the 256-B bank row.
constexpr int v_pitch_bytes(int dvp) {
    int b = round_up(dvp * 2, 32);
    return ((b / 32) & 1) ? b : b + 32;
}

// W32 form: a 32-lane half reads 4 rows x 64 B per ds_read_b64_tr_b16 -- a pitch of 64 (mod 256) bytes tiles the bank row.
constexpr int v_pitch_bytes32(int dvp) {
    int b = round_up(dvp * 2, 64);
    while (b % 256 != 64 && b % 256 != 192) b += 64;
    return b;
}

// G > 1: "shared score" form for the hook's "replace" injection (pnp_utils.py:133-143, :259-262).  There every chunk
// uses q,k of chunk 0, so softmax(q k^T) is the SAME matrix for the G chunks of a frame and only V differs: one
// workgroup computes the probabilities once and multiplies them with the G value blocks side by side (a G*DH-wide
// V tile), writing G output samples.  1/G of the QK^T MFMAs and of the exponentials.
// LAZY: the softmax scale is folded into Q once (q <- fp16(q * scale * log2 e)) and the running reference m_ref enters
// the score MFMA as its C operand, so a score leaves the matrix pipe as (s - m_ref) in base-2 units and needs only the
// exponential: one VALU op per score less than the exact form.  m_ref follows the running max loosely -- it is raised
// (with the usual rescale of O) only when a block's max exceeds it by more than 8, so P stays below 2^8 (exact in fp32
// accumulation; fp16 keeps its relative precision).  Costs one extra fp16 rounding of q.
// W32: the same kernel on mfma_f32_32x32x16 (dh = 40, LAZY only).  A wave's 16 QT queries are QT / 2 tiles of 32; a score tile is
// 32 keys x 32 queries (lane = query l & 31, half h = l >> 5 holds keys (r & 3) + 8 (r >> 2) + 4 h), so registers 8 j .. 8 j + 7 of
// a tile are the B operand of the j-th 16-key step of O^T += V^T P^T as they stand.  The kernel is bound by VECTOR ISSUE (64
// exponentials per lane and block), and every MFMA holds the SIMD's vector issue for 8 cycles whatever its shape: 28 MFMAs of 32
// cycles per wave and block instead of 56 of 16 / 8 give half the held slots back at 17 % more matrix-pipe time (the 48 value
// columns become two tiles of 32).  MEASURED SLOWER (690 vs 660 us plain, 457 vs 388 us shared-score): the pipe time it adds
// costs more than the issue slots it returns -- the two co-resident workgroups' matrix and vector phases do not overlap as
// freely as that budget assumed.  Kept behind variant bit 2 as a tested A/B form, never chosen by the dispatcher.
// NWV: waves per workgroup, 4 or 8: eight waves share one staged K / V block, so a wave issues half the LDS-DMA pieces per block
// and the K / V stream through L2 halves, for a barrier across eight waves instead of four.  Measured: -5 % for the shared-score
// form (round 3), -3 ... -12 % for the plain dh = 40 kernel since its speculative reference (round 5; equal before it), slower for
// dh = 80 (353 vs 329 us); both dh = 40 forms take it by default (variant bit 3: four waves).  dh = 160 at 256 keys takes it with
// two query tiles per wave: ONE workgroup per (sample, head) -- that level is bound by the latency of its four key blocks.
// GL: LIVE value sets of the G (GL < G: the batch came without its last chunk(s) -- the sampler's dead-branch elimination): the sets
// g >= GL are neither read nor written and their column tiles are skipped; the tile layout, the ones column and every instruction
// that touches a live set are those of GL == G, so the live outputs are the full call's bit for bit.
template <class TT, int DH, int QT, int G, bool LAZY, bool W32 = false, int NWV = 4, int GL = G>
__global__ __launch_bounds__(64 * NWV, (NWV == 8 && DH > 128) ? 1 : 2) void attn_kernel(AttnParams p) {
    constexpr int NTH = 64 * NWV;
    static_assert(GL >= 1 && GL <= G && (GL == G || !W32), "live sets");
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    using V4 = typename TT::v4;
    static_assert(!W32 || (LAZY && QT % 2 == 0 && DH % 8 == 0 && DH <= 64), "W32: lazy softmax, pairs of query tiles, one 128-byte K row");
    // QK^T contraction: NKS steps of 32 (mfma 16x16x32) + one step of 16 (mfma 16x16x16) when DH % 32 is 8 or 16,
    // so head dim 40 costs 48 instead of 64, and 80 costs exactly 80.
    constexpr int NKS = DH / 32, TAIL = (DH % 32) ? 1 : 0;
    static_assert(DH % 32 == 0 || DH % 32 == 8 || DH % 32 == 16, "head dim");
    constexpr int DKP = NKS * 32 + TAIL * 16;
    constexpr int DV = G * DH;                       // value columns side by side (G sets)
    constexpr int DVP = W32 ? round_up(DV + 1, 32) : round_up(DV, 16), NC = DVP / 16;
    // spare V column (DVP > DV): filled with ones, so the MFMA that builds O also builds the softmax denominator
    constexpr bool ONES = DVP > DV;
    // K block rows: 128 B (DKP <= 64) or 256 B (DKP <= 128), 16-B slots XOR-swizzled by the row so a ds_read_b128 of
    // 16 keys x one k-chunk is bank-conflict free (same rule as the GEMM tiles); larger head dims keep padded rows.
    constexpr int KROW = k_row_elems(DKP);           // elements
    constexpr int VROW = (W32 ? v_pitch_bytes32(DVP) : v_pitch_bytes(DVP)) / 2;     // elements
    constexpr int CPR = DH / 8;                      // 16-B chunks per K row (and per V row of one set)


    constexpr int CPRV = G * CPR;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E* sK = reinterpret_cast<E*>(smem_raw);          // [2][KVB][KROW]
    E* sV = sK + 2 * KVB * KROW;                     // [2][KVB][VROW]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // provably wave-uniform: the staging's per-wave tests become scalar branches
    const int fr = lane & 15, fg = lane >> 4;
    // XCD-aware order (guide T1): workgroups are dealt round-robin to the 8 XCDs; give each XCD a CONTIGUOUS run of the
    // (sample, head, query tile) sequence, so the query tiles of one (sample, head) -- which all walk the same K / V --
    // share one XCD's 4 MiB L2 instead of spreading every head's K / V over all eight (measured: 1.04 GB of fabric
    // reads per launch at n = 4096, dh = 40 against 0.19 GB of q, k, v).
    int b, h, qtile;
    {
        const int gx = (p.n + 16 * NWV * QT - 1) / (16 * NWV * QT);
        const int nwg = gridDim.x, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, loc = id >> 3;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        qtile = L % gx;
        const int bh = L / gx;
        h = bh % p.heads;
        b = bh / p.heads;
    }
    const int q0 = qtile * (16 * NWV * QT) + wave * (16 * QT);
    const int bqk = p.qk_map ? p.qk_map[b] : b;
    const int gs = p.set_stride;                     // output / value sample of set g: b + g*gs
    const E* Qg = reinterpret_cast<const E*>(p.Q) + (long)bqk * p.bsq + h * DH;
    const int nk = p.nk;

    // zero LDS once: pad columns (DH..DKP of K, DH..DVP of V) are never written again
    {
        uint4* z = reinterpret_cast<uint4*>(smem_raw);
        constexpr int total16 = (2 * KVB * KROW + 2 * KVB * VROW) * 2 / 16;
        for (int i = t; i < total16; i += NTH) z[i] = make_uint4(0, 0, 0, 0);
    }
    if (ONES) {
        __syncthreads();
        if (t < 2 * KVB) sV[t * VROW + DV] = (E)1.0f;
    }

    // Q fragments (B operand of S^T = K Q^T): lane (query fr, group fg) holds dh 32*ks + 8*fg .. +7
    V8 qf[QT][NKS > 0 ? NKS : 1];
    V4 qt4[QT];  // tail step: dh 32*NKS + 4*fg .. +3
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int q = q0 + qt * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int d = ks * 32 + fg * 8;
            V8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (E)0.0f;
            if (q < p.n) v = *reinterpret_cast<const V8*>(Qg + (long)q * p.ldq + d);
            qf[qt][ks] = v;
        }
        V4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (E)0.0f;
        if (TAIL && q < p.n && NKS * 32 + fg * 4 < DH) w = *reinterpret_cast<const V4*>(Qg + (long)q * p.ldq + NKS * 32 + fg * 4);
        qt4[qt] = w;
    }
    if constexpr (LAZY) {
        const float cq = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[qt][ks][j] = from_f32<E>(to_f32(qf[qt][ks][j]) * cq);
#pragma unroll
            for (int j = 0; j < 4; ++j) qt4[qt][j] = from_f32<E>(to_f32(qt4[qt][j]) * cq);
        }
    }

    // K / V blocks go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 16 B per lane, lane-linear destination): no
    // staging registers, no ds_write pass, and a padding slot / a key row past nk is an out-of-range offset (zeros).
    // One DMA instruction fills 64 consecutive 16-B slots of a block; slot id -> (row, slot in row); the K rows' XOR
    // swizzle is applied on the SOURCE chunk index.  V slots past the value columns are masked out, so the ones column
    // and the zero padding written once at kernel start survive.
    constexpr int SK = KROW / 8, SV = VROW / 8;            // 16-B slots per K / V row
    constexpr int RK = (KVB * SK + NTH - 1) / NTH, RV = (KVB * SV + NTH - 1) / NTH;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.K), 0, (int)p.k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.V), 0, (int)p.v_bytes, 0x00020000);
    unsigned koff[RK], voff[RV];       // byte offset of this lane's chunk in key block 0 (OOB: never loads)
    int krow[RK], vrow[RV];
    bool vact[RV];
#pragma unroll
    for (int r = 0; r < RK; ++r) {
        const int id = r * NTH + t;
        const int row = id / SK, sl = id - row * SK;
        int c = sl;
        if (KROW == 64) c = sl ^ ((row >> 1) & 7);
        if (KROW == 128) c = sl ^ (row & 15);
        krow[r] = row;
        koff[r] = (row < KVB && c < CPR) ? (unsigned)((((long)bqk * p.bsk + h * DH + (long)row * p.ldk + c * 8)) * 2) : OOB;
    }
#pragma unroll
    for (int r = 0; r < RV; ++r) {
        const int id = r * NTH + t;
        const int row = id / SV, sl = id - row * SV;
        const int g = sl / CPR, c = sl - g * CPR;
        vrow[r] = row;
        vact[r] = row < KVB && sl < CPRV;
        const int bo = b + (g < GL ? g : 0) * gs;
        const int bv = p.v_map ? p.v_map[bo] : bo;
        // (a dead set's slots load from an out-of-range offset = zeros: the same instructions under the same lane masks as the
        // full call -- masking those lanes out instead gave wrong values in the eight-wave form)
        voff[r] = (vact[r] && g < GL) ? (unsigned)((((long)bv * p.bsv + h * DH + (long)row * p.ldv + c * 8)) * 2) : OOB;
    }
    const unsigned kstep = (unsigned)(KVB * p.ldk * 2), vstep = (unsigned)(KVB * p.ldv * 2);
    auto stage_block = [&](int kb, int buf) {
        E* dK = sK + buf * KVB * KROW;
        E* dV = sV + buf * KVB * VROW;
        const int r0 = kb * KVB;
#pragma unroll
        for (int r = 0; r < RK; ++r) {
            if (r * NTH + wave * 64 < KVB * SK) {          // wave-uniform: this instruction has slots to fill
                const unsigned off = (koff[r] != OOB && r0 + krow[r] < nk) ? koff[r] + (unsigned)kb * kstep : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, LDS_PTR(dK + (r * NTH + wave * 64) * 8), 16, off, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < RV; ++r) {
            if (r * NTH + wave * 64 < KVB * SV) {
                const unsigned off = ((GL == G || voff[r] != OOB) && r0 + vrow[r] < nk) ? voff[r] + (unsigned)kb * vstep : OOB;
                if (vact[r]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, LDS_PTR(dV + (r * NTH + wave * 64) * 8), 16, off, 0, 0, 0);
            }
        }
    };

    if constexpr (W32) {
        // ================= the 32 x 32 x 16 form (see the template comment) =================
        constexpr int QT2 = QT / 2;                  // 32-query tiles of this wave
        constexpr int KS16 = DKP / 16;               // k16 steps of the score MFMAs (dh 40: 3, the last one half zeros)
        constexpr int NC32 = DVP / 32;               // 32-column value tiles (the ones column included)
        const int r32 = lane & 31, hh = lane >> 5, cb = (lane >> 4) & 1;
        // Q fragments: lane (query r32, half hh) holds dh 16 ks + 8 hh .. + 7, scaled by scale * log2(e), zero past DH
        V8 qf2[QT2][KS16];
        const float cq = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int qt = 0; qt < QT2; ++qt) {
            const int q = q0 + qt * 32 + r32;
#pragma unroll
            for (int ks = 0; ks < KS16; ++ks) {
                const int d = ks * 16 + hh * 8;
                V8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (E)0.0f;
                if (q < p.n && d < DH) v = *reinterpret_cast<const V8*>(Qg + (long)q * p.ldq + d);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = from_f32<E>(to_f32(v[j]) * cq);
                qf2[qt][ks] = v;
            }
        }
        f16_t o2[NC32][QT2], cneg2[QT2];
#pragma unroll
        for (int qt = 0; qt < QT2; ++qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cneg2[qt][r] = 0.f;
#pragma unroll
            for (int c = 0; c < NC32; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) o2[c][qt][r] = 0.f;
        }
        const int nblocks2 = (nk + KVB - 1) / KVB;
        __syncthreads();  // zero fill (and the ones column) done before the first DMA lands
        stage_block(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kb = 0; kb < nblocks2; ++kb) {
            const int cur = kb & 1;
            if (kb + 1 < nblocks2) stage_block(kb + 1, cur ^ 1);
            const E* cK = sK + cur * KVB * KROW;
            const E* cV = sV + cur * KVB * VROW;
            // ---- S^T = K Q^T - m_ref: s2[kt][qt], lane holds keys 32 kt + (r & 3) + 8 (r >> 2) + 4 hh of query r32
            f16_t s2[2][QT2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int qt = 0; qt < QT2; ++qt) s2[kt][qt] = cneg2[qt];
#pragma unroll
                for (int ks = 0; ks < KS16; ++ks) {
                    const int row = kt * 32 + r32;
                    const V8 kf = *reinterpret_cast<const V8*>(cK + row * KROW + k_slot<KROW>(row, ks * 2 + hh) * 8);
#pragma unroll
                    for (int qt = 0; qt < QT2; ++qt) s2[kt][qt] = TT::mfma32x32(kf, qf2[qt][ks], s2[kt][qt]);
                }
            }
            // ---- online softmax (fp32, lazy reference)
            const bool tail = (kb + 1) * KVB > nk;
            V8 pf2[QT2][2][2];
#pragma unroll
            for (int qt = 0; qt < QT2; ++qt) {
                if (tail) {
                    int kbase = kb * KVB + hh * 4;
                    asm volatile("" : "+v"(kbase));
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (kbase + kt * 32 + (r & 3) + 8 * (r >> 2) >= nk) s2[kt][qt][r] = -1e30f;
                }
                float mx = -1e30f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s2[kt][qt][r]);
                if (__any((kb == 0) || (mx > 8.0f))) {
                    {   // the query's maximum: its other half lives in lane l ^ 32
                        unsigned u = __builtin_bit_cast(unsigned, mx);
                        auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                        mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
                    }
                    const bool shift = (kb == 0) || (mx > 8.0f);
                    const float delta = shift ? mx : 0.f;
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int r = 0; r < 16; ++r) cneg2[qt][r] -= delta;
#pragma unroll
                    for (int c = 0; c < NC32; ++c)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o2[c][qt][r] *= alpha;
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s2[kt][qt][r] -= delta;
                }
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        V8 v;
#pragma unroll
                        for (int i = 0; i < 8; ++i) v[i] = from_f32<E>(__builtin_amdgcn_exp2f(s2[kt][qt][8 * j + i]));
                        pf2[qt][kt][j] = v;
                    }
            }
            // ---- O^T += V^T P^T: the j-th 16-key step of key tile kt pairs B k index 8 hh + i with key 16 j + 4 hh + (i & 3) + 8 (i >> 2)
#pragma unroll
            for (int c = 0; c < NC32; ++c) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const E* base = cV + (kt * 32 + j * 16 + hh * 4 + (fr >> 2)) * VROW + c * 32 + cb * 16 + (fr & 3) * 4;
                        const V4 lo = TT::tr_read(base);
                        const V4 hi = TT::tr_read(base + 8 * VROW);
                        V8 vf;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { vf[r] = lo[r]; vf[4 + r] = hi[r]; }
#pragma unroll
                        for (int qt = 0; qt < QT2; ++qt) o2[c][qt] = TT::mfma32x32(vf, pf2[qt][kt][j], o2[c][qt]);
                    }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // ---- normalise and store: lane (query r32, half hh) holds value columns 32 c + 8 rq + 4 hh .. + 3 in registers 4 rq .. 4 rq + 3
        E* Og2 = reinterpret_cast<E*>(p.O) + (long)b * p.bso + h * DH;
#pragma unroll
        for (int qt = 0; qt < QT2; ++qt) {
            constexpr int drow = DV % 32;            // the ones column: row drow of value tile DV / 32
            const float l = __shfl(o2[DV / 32][qt][(drow & 3) + 4 * (drow >> 3)], r32 + 32 * ((drow >> 2) & 1), 64);
            const float inv = 1.0f / l;
            const int q = q0 + qt * 32 + r32;
            if (q >= p.n) continue;
#pragma unroll
            for (int c = 0; c < NC32; ++c)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int d0 = c * 32 + rq * 8 + hh * 4;
                    if (d0 >= DV) continue;
                    const int g = d0 / DH, d = d0 - g * DH;
                    V4 ov;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ov[r] = from_f32<E>(o2[c][qt][4 * rq + r] * inv);
                    *reinterpret_cast<V4*>(Og2 + (long)g * gs * p.bso + (long)q * p.ldo + d) = ov;
                }
        }
        return;
    }
    f4_t o[NC][QT];
    float m_run[QT], l_run[QT];
    f4_t cneg[QT];   // LAZY: -m_ref of this lane's query, the C operand of the score MFMAs
    float lsum[QT];  // softmax denominators of this lane's queries (after the key loop)

    const float cexp = p.scale * 1.44269504088896340736f;
    const int nblocks = (nk + KVB - 1) / KVB;

    // SPECULATIVE reference (round 4; LAZY form only).  The lazy softmax raises its reference only when a block's maximum
    // exceeds it by more than 8 -- which, after the first key block has set the reference to that block's maximum, almost never
    // happens; but DETECTING it costs a lane-local maximum per query tile and key block (8 v_max3 + compare + branch: ~140 of the
    // ~1 830 issue cycles of a block, profiles/r04_g_attn_issue_budget.txt) in a kernel that is bound by instruction issue.
    // Pass 0 therefore takes the maximum in the FIRST block only and never looks again: P = exp2(s - m_ref) stays finite in 16
    // bits as long as no later score exceeds the first block's maximum by 2^16 -- and if one does, P overflows to inf, the
    // denominator (sum of P) is inf, and the workgroup (one vote, the K / V staging is shared) runs the tile again with the
    // checked loop.  Exact either way: every P / sum(P) is formed from one reference per query.
    // Taken where it measured faster: the plain dh = 40 kernel (661 -> 610 us at F = 8, same box, bit-identical outputs on inputs
    // that never trip it; the shared-score form and dh = 80 / 160 lost 3-4 % to the second loop's control flow and keep one pass).
    constexpr bool SPEC = LAZY && !W32 && G == 1 && DH == 40;
    constexpr int NPASS = SPEC ? 2 : 1;
#ifdef VFACE_ATTN_STAMPS
    const int first_pass = 0;
#else
    const int first_pass = (SPEC && (p.variant & 16)) ? 1 : 0;      // variant bit 4: checked loop only (A/B)
#endif
  for (int pass = first_pass; pass < NPASS; ++pass) {
    const bool spec = SPEC && pass == 0;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) o[c][qt] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) { m_run[qt] = -1e30f; l_run[qt] = 0.f; }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) cneg[qt] = f4_t{0.f, 0.f, 0.f, 0.f};

    __syncthreads();  // zero fill (and the ones column) done before the first DMA lands; a second pass: every wave left the loop
    stage_block(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // DIAGNOSTIC (compiled only with -DVFACE_ATTN_STAMPS; variant bit 8, never set by the engine): s_memtime stamps around the phases of a key block; the sums
    // replace the first bytes of O as [workgroup][wave][8] floats.
#ifdef VFACE_ATTN_STAMPS
    const bool dbg = p.variant & 0x100;
#else
    constexpr bool dbg = false;
#endif
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
    auto stamp = [&](int ph) {
        if (!dbg) return;
        unsigned long long tn;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        if (ph >= 0) tph[ph] += tn - tlast;
        tlast = tn;
    };
    stamp(-1);
    for (int kb = 0; kb < nblocks; ++kb) {
        const int cur = kb & 1;
        if (kb + 1 < nblocks) stage_block(kb + 1, cur ^ 1);   // lands under this block's math; the other buffer is free
        stamp(0);
        const E* cK = sK + cur * KVB * KROW;
        const E* cV = sV + cur * KVB * VROW;

        // ---- S^T = K Q^T : s[tile][qt], lane holds keys 16*tile + 4*fg + r of query fr
        f4_t s[4][QT];
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[tl][qt] = LAZY ? cneg[qt] : f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const V8 kf = *reinterpret_cast<const V8*>(cK + (tl * 16 + fr) * KROW + k_slot<KROW>(tl * 16 + fr, ks * 4 + fg) * 8);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) s[tl][qt] = TT::mfma32(kf, qf[qt][ks], s[tl][qt]);
            }
        }
        if (TAIL) {
            // The k16 tail step of a tile reads, as its C operand, what that tile's k32 steps wrote -- a DIFFERENT MFMA opcode.
            // Issued straight behind its producer (where hipcc's scheduler put some of them: it inserts no wait states for this
            // pair on gfx950) the tail returned stale sums: dh = 40 / 80 kernels that were wrong or right depending on unrelated
            // edits elsewhere in the kernel (profiles/r04_f_attention_mfma_hazard.txt).  All k32 steps of the four key tiles
            // therefore go first, the tails after a scheduling fence: >= 3 QT other MFMAs between a tile's two steps.
            if (NKS > 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) {
                const V4 kf = *reinterpret_cast<const V4*>(cK + (tl * 16 + fr) * KROW + k_slot<KROW>(tl * 16 + fr, NKS * 4 + (fg >> 1)) * 8 + (fg & 1) * 4);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) s[tl][qt] = TT::mfma16(kf, qt4[qt], s[tl][qt]);
            }
            if (NKS > 0) __builtin_amdgcn_sched_barrier(0);
        }
        stamp(1);
        // ---- online softmax (fp32)
        const bool tail = (kb + 1) * KVB > nk;
        V8 pf[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (tail) {
                // (the key index goes through an opaque register INSIDE the branch: hipcc otherwise hoists the 16 index
                //  computations and 16 compares of the mask above the branch, into every block of the loop -- 32 vector
                //  instructions per block of a VALU-bound kernel for a branch only the last block can take)
                int kbase = kb * KVB + fg * 4;
                asm volatile("" : "+v"(kbase));
#pragma unroll
                for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kbase + tl * 16 + r >= nk) s[tl][qt][r] = -1e30f;
            }
            float mx = -1e30f;
            // (speculative pass: the maximum is taken in the first key block only -- a wave-uniform branch around 8 v_max3)
            const bool look = !spec || kb == 0;
            if (look) {
#pragma unroll
                for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[tl][qt][r]);
            }
            if constexpr (!LAZY) mx = quad_row_max(mx);
            float ls = 0.f;
            if constexpr (LAZY) {
                // scores are relative to m_ref already.  Raise m_ref (first block: set it) only where needed.  The decision
                // "does any query of this wave exceed its reference by more than 8" needs only the LANE-local maxima (a
                // query's four lane groups hold 16 of its 64 keys each: the query's max exceeds 8 iff one of theirs does), so
                // the cross-lane reduction (two lane swaps + the canonicalising maxima hipcc wraps around them: 10 vector
                // instructions per query tile) runs only inside the rare branch -- the kernel is VALU-issue-bound.
                if (look && __any((kb == 0) || (mx > 8.0f))) {
                    mx = quad_row_max(mx);
                    const bool shift = (kb == 0) || (mx > 8.0f);
                    const float delta = shift ? mx : 0.f;
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                    for (int r = 0; r < 4; ++r) cneg[qt][r] -= delta;
                    if (!ONES) l_run[qt] *= alpha;
#pragma unroll
                    for (int c = 0; c < NC; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[c][qt][r] *= alpha;
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[tl][qt][r] -= delta;
                }
#pragma unroll
                for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(s[tl][qt][r]);
                        s[tl][qt][r] = pv;
                        if (!ONES) ls += pv;
                    }
            } else {
            const float m_new = fmaxf(m_run[qt], mx);
            // the running max rarely moves after the first key blocks: rescale only when some query's did
            // (alpha == 1 exactly otherwise, so skipping is exact, not an approximation)
            if (__any(m_new > m_run[qt])) {
                const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * cexp);
                m_run[qt] = m_new;
                if (!ONES) l_run[qt] *= alpha;
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[c][qt][r] *= alpha;
            }
            const float mc = m_run[qt] * cexp;
#pragma unroll
            for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(s[tl][qt][r], cexp, -mc));
                    s[tl][qt][r] = pv;
                    if (!ONES) ls += pv;
                }
            }
            if (!ONES) l_run[qt] += ls;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                V8 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = from_f32<E>(s[2 * st][qt][r]);
                    v[4 + r] = from_f32<E>(s[2 * st + 1][qt][r]);
                }
                pf[qt][st] = v;
            }
        }
        stamp(2);
        // ---- O^T += V^T P^T
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c * 16 >= GL * DH && (c + 1) * 16 <= DV) continue;      // a tile of dead sets only (compile-time after unrolling)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const E* base = cV + (st * 32 + fg * 4 + (fr >> 2)) * VROW + c * 16 + (fr & 3) * 4;
                const V4 lo = TT::tr_read(base);
                const V4 hi = TT::tr_read(base + 16 * VROW);
                V8 vf;
#pragma unroll
                for (int r = 0; r < 4; ++r) { vf[r] = lo[r]; vf[4 + r] = hi[r]; }
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) o[c][qt] = TT::mfma32(vf, pf[qt][st], o[c][qt]);
            }
        }
        stamp(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(4);
        __syncthreads();
        stamp(5);
    }
    if (dbg) {
        if (lane == 0) {
            float* d = reinterpret_cast<float*>(p.O) + ((long)blockIdx.x * 4 + wave) * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i] = (float)tph[i];
        }
        return;
    }

    // ---- denominators; the speculative pass votes: any of them not finite -> the workgroup runs the checked loop
    bool bad = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        if (ONES) {
            // the denominator is row DV of O^T: held by lane group (DV % 16) / 4 in register 0 of tile DV / 16
            lsum[qt] = __shfl(o[DV / 16][qt][0], fr + 16 * ((DV % 16) / 4), 64);
        } else {
            lsum[qt] = quad_row_sum(l_run[qt]);
        }
        bad = bad || !(lsum[qt] < 3.0e38f);      // inf or NaN
        if (!ONES) {
            // (without the ones column the denominator is an fp32 sum of the fp32 exponentials: it stays finite when a P overflowed
            // only in its 16-bit rounding -- look at the sums the matrix pipe formed from the rounded P instead)
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) bad = bad || !(fabsf(o[c][qt][r]) < 3.0e38f);
        }
    }
    if (!spec) break;
    if constexpr (SPEC) { if (!__syncthreads_or(bad ? 1 : 0)) break; }
  }

    // ---- normalise and store: lane holds value columns 16c + 4fg + r of query fr (set = column / DH)
    E* Og = reinterpret_cast<E*>(p.O) + (long)b * p.bso + h * DH;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float l = lsum[qt];
        const float inv = 1.0f / l;
        const int q = q0 + qt * 16 + fr;
        if (q >= p.n) continue;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int d0 = c * 16 + fg * 4;
            if (d0 >= DV) continue;
            const int g = d0 / DH, d = d0 - g * DH;   // DH % 4 == 0: a lane's 4 columns stay inside one set
            if (g >= GL) continue;
            V4 ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) ov[r] = from_f32<E>(o[c][qt][r] * inv);
            *reinterpret_cast<V4*>(Og + (long)g * gs * p.bso + (long)q * p.ldo + d) = ov;
        }
    }
}

template <class TT, int DH, int QT, int G = 1, bool LAZY = false, bool W32 = false, int NWV = 4, int GL = G>
int launch(const AttnParams& p, hipStream_t stream) {
    constexpr int DKP = (DH / 32) * 32 + ((DH % 32) ? 16 : 0), DVP = W32 ? round_up(G * DH + 1, 32) : round_up(G * DH, 16);
    constexpr int KROW = k_row_elems(DKP), VROW = (W32 ? v_pitch_bytes32(DVP) : v_pitch_bytes(DVP)) / 2;
    constexpr size_t lds = (size_t)(2 * KVB * KROW + 2 * KVB * VROW) * 2;
    auto kern = attn_kernel<TT, DH, QT, G, LAZY, W32, NWV, GL>;
    static VfOncePerDevice attr_set;
    if (lds > 64 * 1024 && !attr_set.set_lds(reinterpret_cast<const void*>(kern), (int)lds)) return VF_ERR_LAUNCH;
    dim3 grid(((p.n + 16 * NWV * QT - 1) / (16 * NWV * QT)) * p.heads * p.B);
    hipLaunchKernelGGL(kern, grid, dim3(64 * NWV), lds, stream, p);
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

template <class TT, bool LAZY>
int dispatch_l(const AttnParams& p, hipStream_t stream) {
    if (p.v_sets == 2) {
        switch (p.dh) {
            case 8: return launch<TT, 8, 2, 2, LAZY>(p, stream);
            case 16: return launch<TT, 16, 2, 2, LAZY>(p, stream);
            case 32: return launch<TT, 32, 2, 2, LAZY>(p, stream);
            case 40: return launch<TT, 40, 2, 2, LAZY>(p, stream);
            default: return VF_ERR_SHAPE;
        }
    }
    if (p.v_sets == 3 && p.v_sets_live == 2) {      // [chunk 0 ; chunk 1] of a three-chunk hook: the G = 3 arithmetic, two sets live
        switch (p.dh) {
            case 8: return launch<TT, 8, 2, 3, LAZY, false, 4, 2>(p, stream);
            case 16: return launch<TT, 16, 2, 3, LAZY, false, 4, 2>(p, stream);
            case 32: return launch<TT, 32, 2, 3, LAZY, false, 4, 2>(p, stream);
            // eight waves per workgroup like the full call (variant bit 3: four, bit-identical).  Round 4's note that the eight-wave
            // instantiation "did not reproduce the full call's bits" dates from the form that MASKED the dead set's LDS-DMA lanes out;
            // with the dead slots loading from an out-of-range offset under the full call's lane masks (stage_block) both wave counts
            // give the full call's bits (round 6: tools/attn_live_sets_probe.py, tests/test_kernels_gpu.py)
            case 40:
                return (p.variant & 8) ? launch<TT, 40, 2, 3, LAZY, false, 4, 2>(p, stream) : launch<TT, 40, 2, 3, LAZY, false, 8, 2>(p, stream);
            default: return VF_ERR_SHAPE;
        }
    }
    if (p.v_sets == 3) {
        switch (p.dh) {
            case 8: return launch<TT, 8, 2, 3, LAZY>(p, stream);
            case 16: return launch<TT, 16, 2, 3, LAZY>(p, stream);
            case 32: return launch<TT, 32, 2, 3, LAZY>(p, stream);
            case 40:
                // eight waves per workgroup by default for the shared-score form (one staged K / V block serves 512 queries:
                // 403 -> 381 us at F = 8, bit-identical -- profiles/r03_g_attention_8wave_ab.txt); variant bit 3: four (A/B)
                if (LAZY && (p.variant & 4)) return launch<TT, 40, 2, 3, LAZY, LAZY>(p, stream);
                return (p.variant & 8) ? launch<TT, 40, 2, 3, LAZY>(p, stream) : launch<TT, 40, 2, 3, LAZY, false, 8>(p, stream);
            default: return VF_ERR_SHAPE;
        }
    }
    switch (p.dh) {
        case 8: return launch<TT, 8, 2, 1, LAZY>(p, stream);
        case 16: return launch<TT, 16, 2, 1, LAZY>(p, stream);
        case 32: return launch<TT, 32, 2, 1, LAZY>(p, stream);
        case 40:
            if (LAZY && (p.variant & 5) == 4) return launch<TT, 40, 4, 1, LAZY, LAZY>(p, stream);      // A/B: the 32 x 32 x 16 form
            if (p.variant & 1) return (p.variant & 8) ? launch<TT, 40, 2, 1, LAZY>(p, stream) : launch<TT, 40, 2, 1, LAZY, false, 8>(p, stream);   // A/B: two query tiles per wave
            // eight waves per workgroup by default since round 5 (one staged K / V block serves 512 queries; bit-identical to the
            // four-wave form): 2 472 -> 2 388 us at 96 samples, 1 268 -> 1 201 at 48, 679 -> 600 at 24 (profiles/r05_m; in round 3,
            // before the speculative reference, the two forms measured equal); variant bit 3: four waves (A/B)
            return (p.variant & 8) ? launch<TT, 40, 4, 1, LAZY>(p, stream) : launch<TT, 40, 4, 1, LAZY, false, 8>(p, stream);
        case 80:
            // (the four-query-tiles-per-wave A/B form of rounds 3-5 spilled 28 B at 256 registers and lost its A/B: not built)
            if (p.variant & 8) return launch<TT, 80, 2, 1, LAZY, false, 8>(p, stream);                 // A/B: eight waves per workgroup
            return launch<TT, 80, 2, 1, LAZY>(p, stream);
        // dh = 160 (the 16 x 16 and 8 x 8 levels: 256 / 64 keys): a workgroup is bound by the latency of its few key blocks, one
        // workgroup per CU (88 KB of LDS) -- two query tiles per wave halve the workgroups and the K / V re-reads where the map has the
        // queries for it (208 -> see profiles/r05_m at 96 samples); per-query arithmetic is the same either way
        case 160: return p.n >= 256 ? launch<TT, 160, 2, 1, LAZY, false, 8>(p, stream) : p.n >= 128 ? launch<TT, 160, 2, 1, LAZY>(p, stream) : launch<TT, 160, 1, 1, LAZY>(p, stream);
        default: return VF_ERR_SHAPE;
    }
}

// variant bit 2 (value 4): dh = 40 on the 32 x 32 x 16 form (W32; A/B only: measured 4 % / 18 % SLOWER than the 16 x 16 forms,
// plain / shared-score -- profiles/r03_d_attention_w32_ab.txt).
// variant bit 0: 2 query tiles per wave at dh = 40 (A/B); bit 1: the exact-scale softmax (scale applied to the fp32 scores
// instead of folded into q: one more VALU op per score, one fp16 rounding of q less)
template <class TT>
int dispatch(const AttnParams& p, hipStream_t stream) {
    return (p.variant & 2) ? dispatch_l<TT, false>(p, stream) : dispatch_l<TT, true>(p, stream);
}

}  // namespace

bool vf_attention_shared_scores_supported(int dh, int v_sets) {
    return (v_sets == 2 || v_sets == 3) && (dh == 8 || dh == 16 || dh == 32 || dh == 40);
}

int vf_launch_attention(const AttnParams& p_in, int dtype, hipStream_t stream) {
    AttnParams p = p_in;
    if (p.v_sets <= 1) { p.v_sets = 1; p.set_stride = 0; p.v_sets_live = 0; }
    else if (p.set_stride <= 0 || !vf_attention_shared_scores_supported(p.dh, p.v_sets)) return VF_ERR_SHAPE;
    if (p.v_sets_live == p.v_sets) p.v_sets_live = 0;
    if (p.v_sets_live != 0 && !(p.v_sets == 3 && p.v_sets_live == 2)) return VF_ERR_SHAPE;
    if (!p.Q || !p.K || !p.V || !p.O) return VF_ERR_ARG;
    if (p.B <= 0 || p.heads <= 0 || p.n <= 0 || p.nk <= 0) return VF_ERR_ARG;
    if (((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.V) & 15) return VF_ERR_ALIGN;
    if ((uintptr_t)p.O & 7) return VF_ERR_ALIGN;
    if ((p.ldq | p.ldk | p.ldv | p.bsq | p.bsk | p.bsv) & 7) return VF_ERR_ALIGN;
    if ((p.ldo | p.bso) & 3) return VF_ERR_ALIGN;
    // extents of the K / V views for the buffer descriptors (sources are bounds-checked: a map entry past the batch reads zeros)
    {
        const unsigned long nsamp = p.v_sets > 1 ? (unsigned long)((p.v_sets_live ? p.v_sets_live : p.v_sets) - 1) * p.set_stride + p.B : (unsigned long)p.B;
        const unsigned long kb = ((nsamp - 1) * p.bsk + (unsigned long)(p.nk - 1) * p.ldk + (unsigned long)p.heads * p.dh) * 2;
        const unsigned long vb = ((nsamp - 1) * p.bsv + (unsigned long)(p.nk - 1) * p.ldv + (unsigned long)p.heads * p.dh) * 2;
        if (kb >= 0xFFFFFFF0ul || vb >= 0xFFFFFFF0ul) return VF_ERR_SHAPE;
        p.k_bytes = (unsigned)kb; p.v_bytes = (unsigned)vb;
    }
    if (dtype == VF_DTYPE_F16) return dispatch<F16>(p, stream);
    if (dtype == VF_DTYPE_BF16) return dispatch<BF16>(p, stream);
    return VF_ERR_DTYPE;
}
