// Fused FeedForward of a BasicTransformerBlock for the level-0 (C = 320) token matrices of the VFace UNet (gfx950):
//
//     out = ff.net[2]( GEGLU( ff.net[0].proj( LayerNorm(x) ) ) ) + x          (REFace/ldm/modules/attention.py:37-64, 231-243:
//                                                                               x = ff(norm3(x)) + x)
//
// ONE launch instead of LayerNorm + GEMM(GEGLU) + GEMM(+residual), and the [M x 4C] hidden matrix (252 MB per level-0 block at
// F = 8) never exists: a workgroup keeps its 128 tokens' normalised activations IN REGISTERS as MFMA B operands for the whole
// kernel (activation-stationary) and streams the two weight matrices through LDS once, hidden chunk by hidden chunk:
//
//   * 256 threads = 4 waves, ONE wave per SIMD with the whole 512-register file (`__launch_bounds__(256, 1)`); a wave owns 32
//     tokens = one column tile of `mfma_f32_32x32x16`.  With one wave per SIMD the kernel is bound by INSTRUCTION ISSUE, not by
//     the matrix pipe (a first version on 16x16x32 tiles issued 1 400 instructions per hidden chunk for 240 MFMAs and ran at
//     their issue time): the 32 x 32 shape does the same work in half the MFMA instructions and leaves 32-cycle gaps for the
//     LDS reads and the GEGLU's vector work.
//   * Prologue: the wave reads its tokens' fp32 rows in the B-operand lane layout (lane = token l & 31, half h = l >> 5 holds
//     channels 16 s + 8 h .. + 7 of every k16 step s), does the two-pass LayerNorm across its two lane halves, and keeps
//     fp16(LN(x)) as C / 16 fragments (80 registers).
//   * Hidden chunk = 64 hidden units = 128 rows of the GEGLU-interleaved ff.net[0] weight (16 value rows, 16 gate rows, ...) =
//     four 32-row A tiles, each holding 16 hidden units' value AND gate rows: in the 32 x 32 accumulator layout
//     (row = (reg & 3) + 8 (reg >> 2) + 4 h) register q < 8 is a value row and register q + 8 ITS gate row, in the same lane.
//     GEMM 1: acc1[4 tiles] over K = C in 64-deep stages (16 KB, LDS-DMA, ring of four, landed ONE interval ahead so the first
//     fragments of a stage are read before its interval starts); GEGLU in registers (value * gelu(gate), erf form); the eight
//     results of a tile, rounded to 16 bits, ARE the B operand of one k16 step of GEMM 2 (guide 3, "an accumulator tile as
//     the next MFMA's operand": element j of lane half h is hidden unit 8 (j >> 2) + 4 h + (j & 3) of the tile), so
//     ff.net[2]'s weight columns are stored in THAT order (packing.pack_ffn_w2) and no lane ever moves.  GEMM 2:
//     out[C / 32 tiles] += W2[:, chunk] h over the chunk's four k16 steps (the whole [C x 64] slice is one 40 KB stage,
//     double-buffered, fetched a chunk ahead in pieces beside GEMM 1's stages).
//   * Epilogue: (out + b2) + x in fp32, transposed through LDS (the ring is free by then) to whole 32-byte row chunks.
//
// L2 -> LDS bytes per FLOP are half those of the 128 x 160 tile GEMM (the activations are never re-staged): 7.6 B / kFLOP.
// Static DMA schedule (counted `s_waitcnt vmcnt`, raw `s_barrier`): every GEMM-1 interval issues 4 + 2 LDS-DMA instructions per
// wave (next-but-two W1 stage, a fifth of the next chunk's W2 slice) -- also past the end, wrapping to stages nobody reads, so
// the counts never change -- and waits `vmcnt(8)` = "the stage after the current one has landed".  No ordinary global load
// lives inside the loop (hipcc would drain the DMA queue for it): ff.net[0]'s bias and LayerNorm's gamma / beta are staged in LDS.
#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

template <class TT, int C>
__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(FfnParams p) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    constexpr int KS = C / 16;          // k16 steps of GEMM 1
    constexpr int KT = C / 64;          // 64-deep W1 stages per hidden chunk (4 k16 steps each)
    constexpr int NOT = C / 32;         // output row tiles of GEMM 2
    constexpr int NCH = 4 * C / 64;     // hidden chunks
    constexpr int NW1 = 4;              // W1 stage ring
    constexpr int W1E = 128 * 64;       // elements per W1 stage
    constexpr int W2E = C * 64;         // elements per W2 stage (one chunk's [C x 64] slice)
    constexpr int W2OPS = C / 32;       // LDS-DMA instructions per wave per W2 stage: 2 per GEMM-1 interval
    constexpr int TOT = NCH * KT;       // W1 stages in all
    static_assert(C % 64 == 0 && W2OPS == 2 * KT, "C must be a multiple of 64");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E* sW1 = reinterpret_cast<E*>(smem_raw);
    E* sW2 = sW1 + NW1 * W1E;
    float* sB1 = reinterpret_cast<float*>(sW2 + 2 * W2E);   // ff.net[0] bias, [8 C] fp32 in packed row order
    float* sGB = sB1 + 8 * C;                               // gamma [C], beta [C]

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const long tok0 = (long)blockIdx.x * 128 + wave * 32;

    // ---- bias of GEMM 1, gamma, beta into LDS (ordinary loads: before any LDS-DMA is in flight)
    for (int i = t; i < 8 * C / 4; i += 256) reinterpret_cast<float4*>(sB1)[i] = reinterpret_cast<const float4*>(p.b1)[i];
    for (int i = t; i < C / 4; i += 256) {
        reinterpret_cast<float4*>(sGB)[i] = reinterpret_cast<const float4*>(p.gamma)[i];
        reinterpret_cast<float4*>(sGB + C)[i] = reinterpret_cast<const float4*>(p.beta)[i];
    }

    // ---- LayerNorm (attention.py:233 norm3, eps 1e-5, fp32, two passes like layernorm_kernel) straight into B fragments
    V8 xf[KS];
    {
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

    // ---- weight streams: buffer descriptors, per-lane source offsets (the LDS image of one DMA instruction is lane-linear:
    // 8 rows x 8 sixteen-byte slots; the rows' XOR swizzle is applied on the SOURCE chunk, as in gemm.hip); the stage's base goes
    // in the instruction's scalar offset
    const __amdgpu_buffer_rsrc_t rW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W1), 0, (int)(8u * C * C * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W2p), 0, (int)(4u * C * C * 2u), 0x00020000);
    int w1_off[4];          // (row * C + 8 * chunk) * 2 for this lane's slot of the wave's i-th instruction of a W1 stage
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        w1_off[i] = (row * C + ch * 8) * 2;
    }
    int w2_off[W2OPS];      // (row * 4C + 8 * chunk) * 2
#pragma unroll
    for (int i = 0; i < W2OPS; ++i) {
        const int row = wave * (C / 4) + i * 8 + (lane >> 3);
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        w2_off[i] = (row * 4 * C + ch * 8) * 2;
    }
    auto issue_w1 = [&](int g) {        // stage g (wrapped): chunk g / KT rows, k tile g % KT -> ring slot g % NW1
        const int gw = g % TOT;
        const int c = gw / KT, kt = gw - c * KT;
        const int base = (c * 128 * C + kt * 64) * 2;
        E* dst = sW1 + (g % NW1) * W1E + (wave * 32) * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = w1_off[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW1, LDS_PTR(dst + i * 8 * 64), 16, off, base, 0, 0);
        }
    };
    auto issue_w2 = [&](int c, int i) { // the wave's i-th instruction of chunk c's (wrapped) [C x 64] slice -> buffer c & 1
        const int base = ((c % NCH) * 64) * 2;
        E* dst = sW2 + (c & 1) * W2E + (wave * (C / 4) + i * 8) * 64;
        const int off = w2_off[i];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW2, LDS_PTR(dst), 16, off, base, 0, 0);
    };

    // fragment reads (A operands of the 32x32x16 MFMA): 32 rows x one 16-byte k chunk; rows 2i and 2i + 1 share a slot of the
    // (row >> 1) & 7 swizzle but sit in different halves of the 64-bank row: conflict-free
    auto w1_frag = [&](int buf, int ks, int rt) -> V8 {       // ks: k16 step inside the stage (0..3)
        const int row = rt * 32 + fr;
        const int slot = (ks * 2 + fh) ^ ((row >> 1) & 7);
        return *reinterpret_cast<const V8*>(sW1 + buf * W1E + row * 64 + slot * 8);
    };
    auto w2_frag = [&](int buf, int ks, int ot) -> V8 {       // ks: k16 step inside the chunk (0..3) = hidden tile
        const int row = ot * 32 + fr;
        const int slot = (ks * 2 + fh) ^ ((row >> 1) & 7);
        return *reinterpret_cast<const V8*>(sW2 + buf * W2E + row * 64 + slot * 8);
    };

    f16_t out[NOT];
#pragma unroll
    for (int ot = 0; ot < NOT; ++ot)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ot][r] = 0.f;

    // ---- prime the pipeline: W1 stages 0..2, all of chunk 0's W2 slice
    __syncthreads();                     // nothing else touches LDS before the DMA lands
#pragma unroll
    for (int g = 0; g < 3; ++g) issue_w1(g);
#pragma unroll
    for (int i = 0; i < W2OPS; ++i) issue_w2(0, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();

    V8 af[2][4];                         // W1 fragments of two consecutive k16 steps
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) af[0][rt] = w1_frag(0, 0, rt);

    for (int c = 0; c < NCH; ++c) {
        f16_t acc1[4];                   // in VECTOR registers (the GEGLU reads them), `out` in accumulator registers: both pinned
                                         // through inline-asm MFMAs -- left to hipcc, loop-carried tiles were renamed across the
                                         // chunk loop at ~190 v_accvgpr copies per chunk of a kernel that is bound by issue slots
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int g = c * KT + kt;
            // top of the interval: this wave's share of stage g + 1 has landed (everything but its 8 youngest DMA instructions);
            // behind the barrier every wave's has, and every wave is done with stage g - 1, whose ring slot stage g + 3 takes
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            raw_barrier();
            issue_w1(g + 3);
            issue_w2(c + 1, 2 * kt);
            issue_w2(c + 1, 2 * kt + 1);
            const int buf = g % NW1, nbuf = (g + 1) % NW1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                // fragments of the NEXT k16 step go out before this step's MFMAs (the first step of the next stage too: it has
                // landed, see the wait above; after the chunk's last stage they are read behind GEMM 2 instead)
                if (ks < 3) {
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) af[(ks + 1) & 1][rt] = w1_frag(buf, ks + 1, rt);
                } else if (kt + 1 < KT) {
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) af[0][rt] = w1_frag(nbuf, 0, rt);
                }
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    if (kt == 0 && ks == 0) TT::mfma32x32_vzero(acc1[rt], af[ks & 1][rt], xf[4 * kt + ks]);
                    else TT::mfma32x32_vacc(acc1[rt], af[ks & 1][rt], xf[4 * kt + ks]);
                }
            }
        }
        // ---- GEMM 2's first weight fragments go out before the GEGLU: their LDS latency runs under its vector work
        // (this chunk's slice landed and became visible at the barrier of interval (c, 2) at the latest; KT < 3: see below)
        constexpr int G2 = 5 <= NOT ? (NOT % 5 == 0 ? 5 : 2) : NOT;      // fragments per read group
        constexpr int NG2 = 4 * NOT / G2;
        static_assert((4 * NOT) % G2 == 0);
        const int b2 = c & 1;
        V8 wfr[2][G2];
        auto rd2 = [&](int grp, int slot) {
#pragma unroll
            for (int u = 0; u < G2; ++u) {
                const int idx = grp * G2 + u;
                wfr[slot][u] = w2_frag(b2, idx / NOT, idx % NOT);
            }
        };
        if constexpr (KT < 3) {
            // short K: the last pieces of this chunk's W2 slice may still be among the 8 youngest DMA instructions
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            raw_barrier();
        }
        rd2(0, 0);
        raw_mfma_to_valu_gap();          // acc1's last MFMA -> the vector reads below
        // ---- GEGLU (attention.py:37-45: x, gate = proj(x).chunk(2); x * gelu(gate)) on the accumulators; bias from LDS.
        // Tile rt = packed rows 32 rt .. : hidden units 16 rt .. 16 rt + 15 of the chunk, value rows first, then their gates.
        // Register q < 8 of a lane is value row (q & 3) + 8 (q >> 2) + 4 h, register q + 8 the same unit's gate.
        V8 hb[4];                        // B operand of GEMM 2's k16 step rt
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const float* bp = sB1 + c * 128 + rt * 32 + fh * 4;
            const float4 v0 = *reinterpret_cast<const float4*>(bp), v1 = *reinterpret_cast<const float4*>(bp + 8);
            const float4 g0 = *reinterpret_cast<const float4*>(bp + 16), g1 = *reinterpret_cast<const float4*>(bp + 24);
            const float ba[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const float bg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int qi = 0; qi < 8; ++qi) {
                const float a = acc1[rt][qi] + ba[qi], gt = acc1[rt][qi + 8] + bg[qi];
                hb[rt][qi] = from_f32<E>(a * gelu_erf_f(gt));
            }
        }
        raw_valu_to_mfma_gap();          // hb's last conversion -> the MFMAs below
        // ---- GEMM 2: out[C x 32 tokens] += W2[:, chunk] h.  Fragment reads run one group ahead, pinned: hipcc otherwise reads
        // each fragment right before its MFMA and waits for it there (one exposed LDS round trip per MFMA).
#pragma unroll
        for (int grp = 0; grp < NG2; ++grp) {
            if (grp + 1 < NG2) rd2(grp + 1, (grp + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < G2; ++u) {
                const int idx = grp * G2 + u;
                const int ks = idx / NOT, ot = idx % NOT;
                TT::mfma32x32_acc(out[ot], wfr[grp & 1][u], hb[ks]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < NCH) {               // first fragments of the next chunk's first stage (landed since interval (c, KT - 1))
            const int nb = ((c + 1) * KT) % NW1;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) af[0][rt] = w1_frag(nb, 0, rt);
        }
    }

    // ---- epilogue: (out + b2) + x, half the channels of a wave's 32 tokens at a time through LDS (row pitch C / 2 + 4 floats:
    // consecutive tokens one 16-byte slot apart), read back as whole 32-byte row chunks
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();                       // every wave is done with the weight ring: it becomes scratch
    raw_mfma_to_valu_gap();              // out's last MFMA -> the accumulator reads below
    constexpr int NH = (NOT + 1) / 2;    // output tiles per half
    constexpr int SP = NH * 32 + 4;
    float* scr = reinterpret_cast<float*>(smem_raw) + wave * (32 * SP);
    E* out16 = reinterpret_cast<E*>(p.out16);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int ot0 = hf * NH, nt = (hf == 0) ? NH : NOT - NH;      // tiles [ot0, ot0 + nt)
        if (nt <= 0) continue;
#pragma unroll
        for (int o = 0; o < NH; ++o) {
            if (o >= nt) continue;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                *reinterpret_cast<float4*>(scr + fr * SP + o * 32 + q4 * 8 + fh * 4) =
                    make_float4(out[ot0 + o][4 * q4], out[ot0 + o][4 * q4 + 1], out[ot0 + o][4 * q4 + 2], out[ot0 + o][4 * q4 + 3]);
        }
        // (LDS operations of one wave execute in order: the reads below see the writes above)
        const int CH = nt * 4;           // 8-channel chunks per row of this half
        for (int it = lane; it < 32 * CH; it += 64) {
            const int tok = it / CH, ch = it - tok * CH;
            const float* sp = scr + tok * SP + ch * 8;
            const float4 a0 = *reinterpret_cast<const float4*>(sp), a1 = *reinterpret_cast<const float4*>(sp + 4);
            const int col = ot0 * 32 + ch * 8;
            const float4 c0 = *reinterpret_cast<const float4*>(p.b2 + col), c1 = *reinterpret_cast<const float4*>(p.b2 + col + 4);
            const long row = tok0 + tok;
            const float* xr = p.x32 + row * p.ldx + col;
            const float4 r0 = *reinterpret_cast<const float4*>(xr), r1 = *reinterpret_cast<const float4*>(xr + 4);
            const float v[8] = {(a0.x + c0.x) + r0.x, (a0.y + c0.y) + r0.y, (a0.z + c0.z) + r0.z, (a0.w + c0.w) + r0.w,
                                (a1.x + c1.x) + r1.x, (a1.y + c1.y) + r1.y, (a1.z + c1.z) + r1.z, (a1.w + c1.w) + r1.w};
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
        }
    }
}

template <class TT, int C>
int launch_c(const FfnParams& p, hipStream_t stream) {
    constexpr size_t lds = (size_t)(4 * 128 * 64 + 2 * C * 64) * 2 + (size_t)10 * C * 4;
    static_assert((size_t)4 * 32 * (((C / 32 + 1) / 2) * 32 + 4) * 4 <= (size_t)(4 * 128 * 64 + 2 * C * 64) * 2, "epilogue scratch fits the weight ring");
    auto kern = ffn_fused_kernel<TT, C>;
    static VfOncePerDevice attr_set;
    if (lds > 64 * 1024 && !attr_set.set_lds(reinterpret_cast<const void*>(kern), (int)lds)) return VF_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(p.M / 128), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

template <class TT>
int launch_t(const FfnParams& p, hipStream_t stream) {
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
    if (!p.x32 || !p.gamma || !p.beta || !p.W1 || !p.b1 || !p.W2p || !p.b2 || (!p.out16 && !p.out32)) return VF_ERR_ARG;
    if (!vf_ffn_fused_supported(p.M, p.C)) return VF_ERR_SHAPE;
    if ((p.ldx & 3) || (p.out16 && (p.ldo & 7)) || (p.out32 && (p.ldo32 & 3))) return VF_ERR_ALIGN;
    if (((uintptr_t)p.x32 | (uintptr_t)p.gamma | (uintptr_t)p.beta | (uintptr_t)p.W1 | (uintptr_t)p.b1 | (uintptr_t)p.W2p |
         (uintptr_t)p.b2 | (uintptr_t)p.out16 | (uintptr_t)p.out32) & 15) return VF_ERR_ALIGN;
    if (dtype == VF_DTYPE_F16) return launch_t<F16>(p, stream);
    if (dtype == VF_DTYPE_BF16) return launch_t<BF16>(p, stream);
    return VF_ERR_DTYPE;
}
