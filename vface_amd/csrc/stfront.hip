// Fused FRONT of a SpatialTransformer for the level-0 (C = 320) token matrices of the VFace UNet (gfx950):
//
//     t0  = proj_in( GroupNorm(x) )                       REFace/ldm/modules/attention.py:278-284 (norm, proj_in, rearrange)
//     qkv = [to_q | to_k | to_v]( LayerNorm(t0) )         attention.py:239 (norm1), :179-183 (to_q, to_k, to_v of attn1)
//
// ONE launch instead of gn_apply + GEMM(proj_in) + layernorm + GEMM(qkv): the normalised activations never exist in HBM.  The
// chain it replaces moved 819 MB per level-0 block at F = 8 (x fp32 in, GN out, t0 fp32 out, t0 in, LN out, LN in, qkv out);
// this kernel reads x once (126 MB), writes t0 (126 MB, the residual carrier the attn1 out-projection adds to) and qkv (189 MB).
//
// Activation-stationary, like ffn.hip (whose weight-stream machinery this is): a workgroup = 4 waves, ONE per SIMD, a wave owns 32
// tokens = one column tile of mfma_f32_32x32x16.
//   * P1   the wave reads its tokens' fp32 rows in the B-operand lane layout, applies the GroupNorm scale / shift of its image
//          (a, b per channel: vface_groupnorm_coeffs_from_cols -- the arithmetic of gn_apply_kernel, same bits) and keeps the
//          16-bit result as C / 16 fragments (80 registers);
//   * IN   proj_in: C / 32 weight stages [32 rows x C] stream through LDS (LDS-DMA, ring of four, three stages ahead); stage s
//          accumulates output-channel tile s into 16 accumulator registers (bias = initial value): [C x 32 tokens] fp32 per wave;
//   * MID  t0 goes out (fp32, through a per-wave LDS transpose: whole 128-byte row segments); LayerNorm (two passes, fp32) over
//          the accumulator tile -- a token's C channels sit in the two lanes l, l ^ 32 -- and the rounded result, taken
//          register by register, IS the B operand of the next GEMM (guide 3, "an accumulator tile as the next MFMA's operand":
//          element e of lane half h of k16 step (tile, j) is channel 32 tile + 16 j + 8 (e >> 2) + 4 h + (e & 3)), the
//          projection weights' k columns being stored in that order (packing.ffn_w2_perm);
//   * Q    the projection: one stage = 32 output columns, 20 MFMAs, the tile leaves through the LDS transpose as 16-byte row
//          pieces.  Token tiles below `rows_full` get every projection column, the others only columns >= nq_lo (under the
//          hook's "replace", chunks >= 1 project V only: pnp_utils.py:133-142).
// Persistent: a workgroup walks token tiles (its own stage sequence never stops between them); tiles that need all columns and
// tiles that need the tail only are served by disjoint sets of workgroups, sized by their work.
#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

constexpr int sf_ring(int n, int want) { for (int r = want; r > 1; --r) if (n % r == 0) return r; return 1; }

template <class TT, int C>
__global__ __launch_bounds__(256, 1) void st_front_kernel(StFrontParams p) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    using V4 = typename TT::v4;
    constexpr int KS = C / 16;          // k16 steps per stage = MFMAs per stage
    constexpr int TIN = C / 32;         // proj_in stages = output-channel tiles of t0
    constexpr int NW = 4;               // weight stage ring
    constexpr int WE = 32 * C;          // elements per stage: [32 rows][C]
    constexpr int OPS = C / 64;         // LDS-DMA instructions per wave per stage (1 KiB each)
    constexpr int RA = sf_ring(KS, 5);  // A-fragment window (reads run RA - 1 steps ahead, across stage boundaries)
    constexpr int SPF = 36;             // fp32 scratch row pitch (floats): 32 + 4
    constexpr int SPH = 40;             // 16-bit scratch row pitch (elements): 32 + 8
    static_assert(C % 64 == 0 && C <= 320, "C: a multiple of 64, at most 320");
    // Output tiles leave UNDER the next stage's MFMAs: a finished tile's values go through the per-wave LDS transpose at k step
    // PA of the following stage and to global memory at k step PB.  The counted waits at a stage top need to know how many of
    // this wave's memory instructions are YOUNGER than the last DMA piece of the stage that must have landed: the stores of a
    // stage are younger than that stage's own last piece (issued at k step LASTP) iff PB >= LASTP.
    constexpr int LASTP = 4 * (OPS - 1) + 1;
    constexpr int PA = 2 < KS ? 2 : KS - 1, PB = 6 < KS ? 6 : KS - 1;
    constexpr bool ST_AFTER = PB >= LASTP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E* sW = reinterpret_cast<E*>(smem_raw);
    float* sBin = reinterpret_cast<float*>(sW + NW * WE);   // proj_in bias [C]
    float* sGam = sBin + C;                                  // LayerNorm gamma [C], beta [C]
    float* sBet = sGam + C;
    float* sA = sBet + C;                                    // GroupNorm scale [C], shift [C] of the current token tile's image
    float* sB = sA + C;
    float* scrAll = sB + C;                                  // per-wave transpose scratch: 32 x SPF floats each

    const int t_ = threadIdx.x, lane = t_ & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t_ >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    float* scr = scrAll + wave * (32 * SPF);
    E* scrh = reinterpret_cast<E*>(scr);

    // ---- which token tiles this workgroup walks: set A = tiles [0, tilesA) (every projection column), set B = the rest
    const int tilesA = p.rows_full / 128, tilesT = p.M / 128;
    const bool inA = (int)blockIdx.x < p.gridA;
    const int t_first = inA ? (int)blockIdx.x : tilesA + ((int)blockIdx.x - p.gridA);
    const int t_stride = inA ? p.gridA : ((int)gridDim.x - p.gridA);
    const int t_end = inA ? tilesA : tilesT;
    const int qf = inA ? 0 : p.nq_lo / 32;           // first projection tile of this set
    const int TQ = p.NQ / 32 - qf;                   // projection stages per token tile
    const int S = TIN + TQ;                          // stages per token tile
    if (t_first >= t_end) return;

    for (int i = t_; i < C / 4; i += 256) {
        reinterpret_cast<float4*>(sBin)[i] = reinterpret_cast<const float4*>(p.b_in)[i];
        reinterpret_cast<float4*>(sGam)[i] = reinterpret_cast<const float4*>(p.gamma)[i];
        reinterpret_cast<float4*>(sBet)[i] = reinterpret_cast<const float4*>(p.beta)[i];
    }

    // ---- the weight stream (ffn.hip's W1 stream): a stage is written in its reading order, the XOR swizzle that makes the
    // fragment reads conflict-free is applied on the SOURCE chunk: slot of k chunk c of row r = (c & ~7) | ((c & 7) ^ ((r >> 1) & 7))
    const i32x4_t rW = raw_buffer_rsrc(p.Wcat, (unsigned)((C + p.NQ) * C) * 2u);
    const unsigned ldsW = lds_addr_of(sW);
    constexpr int SPR = C / 8;          // 16-byte slots per row
    int w_off[OPS];
#pragma unroll
    for (int i = 0; i < OPS; ++i) {
        const int id = (wave * OPS + i) * 64 + lane;
        const int r1 = id / SPR, s1 = id - r1 * SPR;
        const int c1 = (s1 & ~7) | ((s1 & 7) ^ ((r1 >> 1) & 7));
        w_off[i] = (r1 * C + c1 * 8) * 2;
    }
    const unsigned ldsWw = ldsW + (unsigned)(wave * OPS * 1024);
    // issue-side state: stage index inside the token tile's sequence, its byte offset into Wcat, its ring slot
    int wi_s = 0, wi_glob = 0, wi_slot = 0;
    auto issue_piece = [&](int i) { raw_lds_dma16(rW, ldsWw + (unsigned)(wi_slot * (WE * 2) + i * 1024), w_off[i], wi_glob); };
    auto advance_issue = [&]() {
        ++wi_s;
        wi_glob += 32 * C * 2;
        if (wi_s == TIN) wi_glob += qf * (32 * C * 2);      // the projection starts at its tile qf
        if (wi_s == S) { wi_s = 0; wi_glob = 0; }           // next token tile (or, past the end, stages nobody reads)
        wi_slot = (wi_slot + 1) & (NW - 1);
    };
    // fragment reads (A operands of the 32x32x16 MFMA): row fr, k chunk 2 ks + fh at slot 8 (ks >> 2) + ((2 (ks & 3) + fh) ^ swz)
    int w_lane[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w_lane[j] = fr * C + (((2 * j + fh) ^ ((fr >> 1) & 7)) << 3);
    auto frag_at = [&](const E* const (&base)[4], int ks) -> V8 { return *reinterpret_cast<const V8*>(base[ks & 3] + (ks >> 2) * 64); };
    int wr_slot = 1;                    // ring slot of the stage AFTER the one being computed
    const E* wa[4];
    const E* wb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { wa[j] = sW + w_lane[j]; wb[j] = sW + WE + w_lane[j]; }

    // ---- prime: stages 0, 1, 2
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int i = 0; i < OPS; ++i) issue_piece(i);
        advance_issue();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();
    V8 af[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) af[i] = frag_at(wa, i);

    f16_t out[TIN];                     // t0 accumulators: [C x 32 tokens] per wave

    // one stage: KS MFMAs into `acc` with B operands `bf`, the A window rolling into the next stage, the DMA pieces of the stage
    // three ahead spread over the k loop.  NWAIT: the DMA / store instructions of this wave that may still be in flight at the
    // top (everything older has landed: the stage after this one in particular).
    auto stage_tail = [&]() {
        advance_issue();
        wr_slot = (wr_slot + 1) & (NW - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) { wa[j] = wb[j]; wb[j] = sW + wr_slot * WE + w_lane[j]; }
    };
#define SF_STAGE_TOP(NWAIT)                                                   \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");             \
    raw_barrier();
#define SF_KLOOP(MFMA_CALL)                                                                                              \
    _Pragma("unroll") for (int ks = 0; ks < KS; ++ks) {                                                                  \
        MFMA_CALL;                                                                                                       \
        af[(ks + RA - 1) % RA] = (ks - 1 + RA < KS) ? frag_at(wa, ks - 1 + RA) : frag_at(wb, ks - 1 + RA - KS);          \
        if (ks % 4 == 1) issue_piece(ks / 4);                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
    }

    V8 xf[KS];                          // B fragments: GroupNorm'd x during IN, LayerNorm'd t0 during Q
    for (int tile = t_first; tile < t_end; tile += t_stride) {
        const long tok0 = (long)tile * 128 + wave * 32;
        // ---- P1: GroupNorm scale / shift of this tile's image into LDS, the token rows into B fragments
        {
            const int img = (int)(((long)tile * 128) / p.hw);
            const float* abp = p.ab + (long)img * p.ld_ab * 2;
            __syncthreads();             // (every wave is done with the previous tile's sA / sB)
            for (int c = t_; c < C; c += 256) {
                const float2 v = *reinterpret_cast<const float2*>(abp + 2 * c);
                sA[c] = v.x; sB[c] = v.y;
            }
            float v[KS][8];
            const float* xr = p.x32 + (tok0 + fr) * p.ldx + fh * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float4 a = *reinterpret_cast<const float4*>(xr + ks * 16);
                const float4 b = *reinterpret_cast<const float4*>(xr + ks * 16 + 4);
                v[ks][0] = a.x; v[ks][1] = a.y; v[ks][2] = a.z; v[ks][3] = a.w; v[ks][4] = b.x; v[ks][5] = b.y; v[ks][6] = b.z; v[ks][7] = b.w;
            }
            __syncthreads();             // sA / sB written
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float* ap = sA + ks * 16 + fh * 8;
                const float* bp = sB + ks * 16 + fh * 8;
                const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
                const float am[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                const float bm[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                V8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = from_f32<E>(v[ks][j] * am[j] + bm[j]);
                xf[ks] = o;
            }
            // EVERY fragment's conversion stays above this point: hipcc otherwise sinks them between the inline-asm MFMAs of the
            // first stage, whose operand reads it cannot see (a v_cvt_pk one instruction ahead of the MFMA that reads its result)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xf[ks]));
            gap_valu_result_to_mfma(xf[KS - 1]);
        }
        // ---- IN: proj_in, one output-channel tile per stage; the bias is the accumulator's initial value.  Tile ot - 1 (final
        // since the previous stage) goes out under this stage's MFMAs: fp32, through the per-wave transpose, rows of 128 bytes
        auto t0_to_scratch = [&](int ot) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                *reinterpret_cast<float4*>(scr + fr * SPF + q4 * 8 + fh * 4) =
                    make_float4(out[ot][4 * q4], out[ot][4 * q4 + 1], out[ot][4 * q4 + 2], out[ot][4 * q4 + 3]);
        };
        auto t0_scratch_to_global = [&](int ot) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int it = lane + 64 * k, tok = it >> 3, ch = it & 7;
                const float4 vv = *reinterpret_cast<const float4*>(scr + tok * SPF + ch * 4);
                *reinterpret_cast<float4*>(p.t0 + (tok0 + tok) * p.ldt0 + ot * 32 + ch * 4) = vv;
            }
        };
#pragma unroll
        for (int ot = 0; ot < TIN; ++ot) {
            // younger than the stage that must have landed: the previous stage's OPS pieces and its 4 stores (of tile ot - 2), and
            // -- if a stage's stores come after its last piece -- the 4 stores of the stage before (tile ot - 3)
            if (ot == 0) { SF_STAGE_TOP(OPS) }
            else if (ot == 1) { SF_STAGE_TOP(OPS) }
            else if (ot == 2) { SF_STAGE_TOP(OPS + 4) }
            else { SF_STAGE_TOP(OPS + 4 + (ST_AFTER ? 4 : 0)) }
            {
                const float* bp = sBin + ot * 32 + fh * 4;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 b = *reinterpret_cast<const float4*>(bp + q4 * 8);
                    out[ot][4 * q4] = b.x; out[ot][4 * q4 + 1] = b.y; out[ot][4 * q4 + 2] = b.z; out[ot][4 * q4 + 3] = b.w;
                }
                gap_valu_result_to_acc_mfma(out[ot]);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                TT::mfma32x32_acc(out[ot], af[ks % RA], xf[ks]);
                af[(ks + RA - 1) % RA] = (ks - 1 + RA < KS) ? frag_at(wa, ks - 1 + RA) : frag_at(wb, ks - 1 + RA - KS);
                if (ks % 4 == 1) issue_piece(ks / 4);
                if (ot > 0 && ks == PA) t0_to_scratch(ot - 1);
                if (ot > 0 && ks == PB) t0_scratch_to_global(ot - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            stage_tail();
        }
        // ---- MID: t0 out (fp32, 128-byte row segments through the per-wave transpose), LayerNorm -> B fragments of the projection
        {
#pragma unroll
            for (int ot = 0; ot < TIN; ++ot) gap_acc_result_to_valu(out[ot]);
            float s = 0.f;
#pragma unroll
            for (int ot = 0; ot < TIN; ++ot)
#pragma unroll
                for (int q = 0; q < 16; ++q) s += out[ot][q];
            s += __shfl_xor(s, 32, 64);
            const float mean = s / (float)C;
            float qq = 0.f;
#pragma unroll
            for (int ot = 0; ot < TIN; ++ot)
#pragma unroll
                for (int q = 0; q < 16; ++q) { const float d = out[ot][q] - mean; qq += d * d; }
            qq += __shfl_xor(qq, 32, 64);
            const float rstd = rsqrtf(qq / (float)C + p.eps);
            t0_to_scratch(TIN - 1);              // (the last tile of t0: the others left under the IN stages)
            t0_scratch_to_global(TIN - 1);
            asm volatile("" ::: "memory");       // (the scratch changes its element type below: no reordering across this point)
#pragma unroll
            for (int ot = 0; ot < TIN; ++ot) {
                // LayerNorm of this tile's 16 values -> two k16 B fragments
                const float* gp = sGam + ot * 32 + fh * 4;
                const float* bp = sBet + ot * 32 + fh * 4;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    V8 o;
#pragma unroll
                    for (int e4 = 0; e4 < 2; ++e4) {
                        const int q4 = 2 * j + e4;
                        const float4 g = *reinterpret_cast<const float4*>(gp + q4 * 8), b = *reinterpret_cast<const float4*>(bp + q4 * 8);
                        o[4 * e4 + 0] = from_f32<E>((out[ot][4 * q4 + 0] - mean) * rstd * g.x + b.x);
                        o[4 * e4 + 1] = from_f32<E>((out[ot][4 * q4 + 1] - mean) * rstd * g.y + b.y);
                        o[4 * e4 + 2] = from_f32<E>((out[ot][4 * q4 + 2] - mean) * rstd * g.z + b.z);
                        o[4 * e4 + 3] = from_f32<E>((out[ot][4 * q4 + 3] - mean) * rstd * g.w + b.w);
                    }
                    xf[2 * ot + j] = o;
                }
                if (p.ln) {
                    // the LayerNorm output itself (the dual-source projections of the hook's linear fusions read it): 16-bit tile
                    // through the scratch -> 16-byte row pieces
                    E* lnp = reinterpret_cast<E*>(p.ln);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        V4 h4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) h4[e] = xf[2 * ot + (q4 >> 1)][4 * (q4 & 1) + e];
                        *reinterpret_cast<V4*>(scrh + fr * SPH + q4 * 8 + fh * 4) = h4;
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int it = lane + 64 * k, tok = it >> 2, ch = it & 3;
                        const V8 vv = *reinterpret_cast<const V8*>(scrh + tok * SPH + ch * 8);
                        *reinterpret_cast<V8*>(lnp + (tok0 + tok) * p.ldln + ot * 32 + ch * 8) = vv;
                    }
                }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xf[ks]));
            gap_valu_result_to_mfma(xf[KS - 1]);
        }
        // ---- Q: the projection, one 32-column tile per stage.  A finished tile is rounded to 16 bits at once (8 registers) and
        // leaves under the NEXT stage's MFMAs: through the per-wave transpose at k step PA, as 16-byte row pieces at k step PB
        asm volatile("" ::: "memory");
        E* qp = reinterpret_cast<E*>(p.qkv);
        V4 hprev[4];
        auto q_to_scratch = [&]() {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) *reinterpret_cast<V4*>(scrh + fr * SPH + q4 * 8 + fh * 4) = hprev[q4];
        };
        auto q_scratch_to_global = [&](int qt_) {
            const int col0 = (qf + qt_) * 32;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int it = lane + 64 * k, tok = it >> 2, ch = it & 3;
                const V8 vv = *reinterpret_cast<const V8*>(scrh + tok * SPH + ch * 8);
                *reinterpret_cast<V8*>(qp + (tok0 + tok) * p.ldq + col0 + ch * 8) = vv;
            }
        };
        for (int qt = 0; qt < TQ; ++qt) {
            // memory instructions younger than the stage that must have landed (see ST_AFTER): the MID phase's stores force a
            // drain at the first stage, a stricter count than needed at the second and third
            if (qt == 0) { SF_STAGE_TOP(0) } else if (qt == 1) { SF_STAGE_TOP(OPS) } else if (qt == 2) { SF_STAGE_TOP(OPS + 2) }
            else { SF_STAGE_TOP(OPS + 2 + (ST_AFTER ? 2 : 0)) }
            f16_t acc;
            TT::mfma32x32_vzero(acc, af[0], xf[0]);
            af[(RA - 1) % RA] = (RA - 1 < KS) ? frag_at(wa, RA - 1) : frag_at(wb, RA - 1 - KS);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 1; ks < KS; ++ks) {
                TT::mfma32x32_vacc(acc, af[ks % RA], xf[ks]);
                af[(ks + RA - 1) % RA] = (ks - 1 + RA < KS) ? frag_at(wa, ks - 1 + RA) : frag_at(wb, ks - 1 + RA - KS);
                if (ks % 4 == 1) issue_piece(ks / 4);
                if (qt > 0 && ks == PA) q_to_scratch();
                if (qt > 0 && ks == PB) q_scratch_to_global(qt - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            stage_tail();
            gap_mfma_result_to_valu(acc);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                for (int e = 0; e < 4; ++e) hprev[q4][e] = from_f32<E>(acc[4 * q4 + e]);
        }
        q_to_scratch();
        q_scratch_to_global(TQ - 1);
        asm volatile("" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA of this workgroup may land after it has left the CU
#undef SF_STAGE_TOP
#undef SF_KLOOP
}

template <class TT, int C>
int launch_c(const StFrontParams& p0, hipStream_t stream) {
    StFrontParams p = p0;
    constexpr size_t lds = (size_t)4 * 32 * C * 2 + (size_t)5 * C * 4 + (size_t)4 * 32 * 36 * 4;
    auto kern = st_front_kernel<TT, C>;
    static VfOncePerDevice attr_set;
    if (lds > 64 * 1024 && !attr_set.set_lds(reinterpret_cast<const void*>(kern), (int)lds)) return VF_ERR_LAUNCH;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    // persistent grid: one workgroup per CU, split between the tiles that project every column and those that project the tail
    const int tilesA = p.rows_full / 128, tilesB = p.M / 128 - tilesA;
    const long workA = (long)tilesA * (C / 32 + p.NQ / 32), workB = (long)tilesB * (C / 32 + (p.NQ - p.nq_lo) / 32);
    int grid = (int)((tilesA + tilesB) < ncu ? (tilesA + tilesB) : ncu);
    int gridA = 0;
    if (tilesA > 0 && tilesB > 0) {
        gridA = (int)((grid * workA + (workA + workB) / 2) / (workA + workB));
        if (gridA < 1) gridA = 1;
        if (gridA > tilesA) gridA = tilesA;
        if (grid - gridA > tilesB) grid = gridA + tilesB;
        if (grid - gridA < 1) gridA = grid - 1;
    } else if (tilesA > 0) {
        gridA = grid;
    }
    p.gridA = gridA;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

template <class TT>
int launch_t(const StFrontParams& p, hipStream_t stream) {
    switch (p.C) {
        case 64: return launch_c<TT, 64>(p, stream);
        case 128: return launch_c<TT, 128>(p, stream);
        case 320: return launch_c<TT, 320>(p, stream);
        default: return VF_ERR_SHAPE;
    }
}

}  // namespace

bool vf_st_front_supported(long M, int C, int hw) {
    return M > 0 && (M % 128) == 0 && hw > 0 && (hw % 128) == 0 && (M % hw) == 0 && (C == 64 || C == 128 || C == 320);
}

int vf_launch_st_front(const StFrontParams& p, int dtype, hipStream_t stream) {
    if (!p.x32 || !p.ab || !p.Wcat || !p.b_in || !p.gamma || !p.beta || !p.t0 || !p.qkv) return VF_ERR_ARG;
    if (!vf_st_front_supported(p.M, p.C, p.hw)) return VF_ERR_SHAPE;
    if (p.NQ <= 0 || (p.NQ % 32) || p.nq_lo < 0 || (p.nq_lo % 32) || p.nq_lo >= p.NQ || p.rows_full < 0 || p.rows_full > p.M ||
        (p.rows_full % 128) || (p.rows_full == 0 && p.nq_lo == 0)) return VF_ERR_SHAPE;
    if ((p.ldx & 3) || (p.ldt0 & 3) || (p.ldq & 7) || (p.ln && (p.ldln & 7)) || p.ld_ab < p.C) return VF_ERR_ALIGN;
    if (((uintptr_t)p.x32 | (uintptr_t)p.ab | (uintptr_t)p.Wcat | (uintptr_t)p.b_in | (uintptr_t)p.gamma | (uintptr_t)p.beta |
         (uintptr_t)p.t0 | (uintptr_t)p.qkv | (uintptr_t)p.ln) & 15) return VF_ERR_ALIGN;
    if (dtype == VF_DTYPE_F16) return launch_t<F16>(p, stream);
    if (dtype == VF_DTYPE_BF16) return launch_t<BF16>(p, stream);
    return VF_ERR_DTYPE;
}
