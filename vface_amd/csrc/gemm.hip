// MFMA GEMM / implicit-GEMM 3x3 convolution for the VFace UNet (gfx950).
//
//   C[m, n] = epilogue( sum_k A(m, k) * Wt[n, k] )
//
// * Wt is the nn.Linear / packed conv weight: [N][Kp] row-major, K contiguous (the natural "B^T" form).
// * A is either a plain row-major [M][K] matrix with leading dimension lda (tokens x channels; NHWC
//   activations are exactly this), or the im2col view of an NHWC image [img][H][W][ld >= Cin] under a 3x3
//   window with padding 1, stride 1|2 and optional nearest x2 upsampling of the input (openaimodel.py
//   Downsample :151-153, Upsample :116-118, ResBlock convs :201-205,225-232), k = (ky*3 + kx)*Cin + ci.
//   Nothing is materialised: each 16-byte k-chunk of the A tile is fetched straight from its source pixel by
//   a direct global->LDS load; padding taps and tile tails read a zero page (no divergent loads).
// * Block tile 128(m) x BN(n) x 64(k), BN = 128 or 160, 4 waves as 2(m) x 2(n); a wave owns 64 x BN/2 =
//   4 x NT mfma_f32_16x16x32 tiles (NT = 4 | 5), fp32 accumulate.  Every channel count of the real UNet is
//   a multiple of 320, so BN = 160 tiles N exactly where BN = 128 would idle 1/6 of the MFMAs at N = 320.
// * LDS rows are 128 B (64 k); the 16-B slot of k-chunk c of row r is c ^ ((r>>1)&7), applied on the
//   per-lane SOURCE address (the LDS image of one global_load_lds wave-instruction is lane-linear) and
//   again on the fragment read: a ds_read_b128 of 16 rows x one chunk is conflict-free.
// * Source pointers live in registers and advance by one K tile per step; for convolutions with
//   Cin % 64 == 0 the tap of a K tile is wave-uniform, so the per-row work per step is one validity-bit
//   test and one 64-bit add.
// * Weights are the MFMA "A" operand and activations the "B" operand, so a lane ends up with 4
//   consecutive output channels of one row: 8-byte stores, and bias / residual / GEGLU pair up in-lane.
//
// Epilogue (fp32): + bias[n] + rowbias[m / rows_per_sample][n] (time-embedding add of ResBlock :264-271, or
// the degenerate single-token cross-attention vector), optional GEGLU (attention.py:37-45,
// weight rows pre-permuted so value / gate tiles alternate), + residual[m][n], store 16-bit or fp32.
#include <type_traits>

#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

constexpr int BM = 128, BK = 64;
enum { MODE_PLAIN = 0, MODE_CONV_FAST = 1, MODE_CONV_GENERIC = 2 };

// RM: the residual form of the wide epilogue (0 none, 1 16-bit, 2 the fp32 stream) as an instantiation of its own; -1 = chosen
// at run time (three copies of the epilogue in one kernel: 16-38 spilled registers at NT = 5, kept for the rare forms only)
template <class TT, int MODE, int NT, bool DB, bool PERSIST, int RM = -1>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmParams p) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    constexpr int BN = 32 * NT;               // 128 | 160
    constexpr int BROUNDS = BN / 32;          // staging rounds of the weight tile (32 rows per round)
    constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, STAGE = A_ELEMS + B_ELEMS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    E* smem = reinterpret_cast<E*>(smem_raw);

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    unsigned long long dbg_t0 = 0, dbg_t1 = 0, dbg_t2 = 0, dbg_e0 = 0, dbg_ew = 0, dbg_er = 0, dbg_es = 0;   // (diagnostic launches only)
    if (p.flags & 0x4000) dbg_t0 = __builtin_amdgcn_s_memtime();
    // XCD-aware tile order (guide T1).  Workgroups are dealt round-robin to the 8 XCDs, each with its own 4 MiB L2.
    // Every XCD gets a contiguous run of the tile sequence L (bijective for any grid size: q = nwg/8, r = nwg%8),
    // and L walks the tile grid in column groups of GN = 8 n-tiles, m-major inside a group: the ~64 tiles an XCD
    // has in flight then cover ~8 m-tiles x 8 n-tiles, so both the activation panels (and the neighbouring
    // m-tiles whose 3x3 windows overlap them) and the weight panels are re-used out of that XCD's L2.
    // A persistent workgroup w walks the virtual block ids w, w + G, w + 2G, ... (G = grid size, a multiple of 8, so all
    // of them map to w's own XCD exactly as the blocks of a one-tile-per-workgroup launch would).
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    const int ntiles = ntn * ntm;
    int m0 = 0, n0 = 0;
    auto tile_origin = [&](int id) {
        const int nwg = ntiles;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7, loc = id >> 3;
        int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
        // implicit-conv K tiles already re-use their activation lines inside one workgroup (chunk-major K), and measured
        // faster in plain launch order; plain GEMMs keep the XCD grouping
        if ((p.flags & GEMM_NO_XCD_REMAP) || MODE != MODE_PLAIN) L = id;
        const int GN = p.tile_group > 0 ? p.tile_group : 8;
        const int g = L / (GN * ntm);
        const int rem = L - g * (GN * ntm);
        const int gw = min(GN, ntn - g * GN);
        const int tm = rem / gw;
        m0 = tm * BM;
        n0 = (g * GN + (rem - tm * gw)) * BN;
    };

    // Operands are read through buffer descriptors (guide T8): `buffer_load_dwordx4 ... offen lds` takes a 32-bit
    // per-lane byte offset and returns zeros for any offset >= num_records, so a padding tap / tile tail is one
    // select of the out-of-range offset instead of a 64-bit pointer select, and the per-tile address update is one
    // 32-bit add.  (Tensors are < 4 GiB: checked by the launcher.)
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A2 ? p.A2 : p.A), 0, (int)(p.A2 ? p.a2_bytes : 0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wt), 0, (int)p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;  // >= num_records of every descriptor (launcher keeps sizes below it)
    constexpr unsigned ES = sizeof(E);

    // staging map: slot = rr*256 + t -> row rr*32 + (t>>3), 16-B slot t&7 holds logical k-chunk schunk
    const int srow = t >> 3;
    const int schunk = (t & 7) ^ ((t >> 4) & 7);

    unsigned a_off[4];       // byte offset of (row, k = schunk*8) [plain] / of tap (0,0), channel schunk*8 [conv]
    unsigned a2_off[4];
    unsigned a_mask[4];      // conv-fast: bit tap = tap in bounds; else: row valid
    int g_oy[4], g_ox[4];    // conv: top-left tap coordinates (generic / upsample paths)
    unsigned g_img[4];       // conv: first pixel index of the row's image
    unsigned b_off[BROUNDS];
    // per-tile operand addresses (after tile_origin)
    auto tile_addresses = [&]() {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int m = m0 + rr * 32 + srow;
        const bool ok = m < p.M;
        a2_off[rr] = OOB;
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
            const int y0 = oy * p.stride - p.pad, x0 = ox * p.stride - p.pad_x;
            g_oy[rr] = y0; g_ox[rr] = x0; g_img[rr] = (unsigned)(img * p.H * p.W);
            if (MODE == MODE_CONV_FAST) {
                const int VH = p.upsample ? 2 * p.H : p.H, VW = p.upsample ? 2 * p.W : p.W;
                unsigned mk = 0;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    if (tp >= p.ntaps) break;
                    const int vy = y0 + tp / p.KW, vx = x0 + tp % p.KW;
                    if (ok && (unsigned)vy < (unsigned)VH && (unsigned)vx < (unsigned)VW) mk |= 1u << tp;
                }
                a_mask[rr] = mk;
                // may wrap below zero for the halo row above the first image: only used when the tap's bit is set
                a_off[rr] = (unsigned)(((((long)img * p.H + y0) * p.W + x0) * p.lda + schunk * 8) * ES);
                // appended 1x1 source (ResBlock shortcut): output row m of a second matrix
                if (p.A2 && ok) a2_off[rr] = (unsigned)(((long)m * p.lda2 + schunk * 8) * ES);
            } else {
                a_mask[rr] = ok ? 1u : 0u;
                a_off[rr] = 0;
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < BROUNDS; ++rr) {
        const int n = n0 + rr * 32 + srow;
        b_off[rr] = n < p.N ? (unsigned)(((long)n * p.ldw + schunk * 8) * ES) : OOB;
    }
    };

    tile_origin(PERSIST ? (int)blockIdx.x : (int)(blockIdx.x % ntiles));
    tile_addresses();

    auto stage = [&](int kt, int buf) {
        E* sA = smem + buf * STAGE;
        E* sB = sA + A_ELEMS;
        const int kbase = kt * BK;           // wave-uniform
        const int k = kbase + schunk * 8;
        if (MODE == MODE_PLAIN) {
            const bool second = p.A2 && kbase >= p.K1;
            const unsigned koff = (unsigned)(second ? kbase - p.K1 : kbase) * ES;
            const bool kin = k < p.K;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const unsigned base = second ? a2_off[rr] : a_off[rr];
                const unsigned off = (kin && base != OOB) ? base + koff : OOB;
                if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA2, LDS_PTR(sA + (rr * 256 + wave * 64) * 8), 16, off, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sA + (rr * 256 + wave * 64) * 8), 16, off, 0, 0, 0);
            }
        } else if (MODE == MODE_CONV_FAST) {
            // K order for Cin % 64 == 0 is (64-channel chunk, tap, channel): the 9 taps of one chunk are
            // consecutive K tiles, so the 9 re-reads of a pixel's 128-B chunk happen within 9 tiles and hit L2
            // (tap-major order re-reads a line only after sweeping all Cin: L2 hit rate 76 %, 10x over-fetch).
            const int cc = kt / p.ntaps, tap = kt - cc * p.ntaps;  // wave-uniform
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const int ci0 = cc * BK;
            if (kbase >= p.K1 && p.A2) {
                // K tiles past the window: the 1x1 shortcut of a ResBlock (openaimodel.py:228-232, 274: skip_connection(x)
                // + h) accumulated into the same tile -- its own launch, its 16-bit intermediate and the residual read
                // of the second convolution disappear
                const unsigned koff = (unsigned)(kbase - p.K1) * ES;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const unsigned off = a2_off[rr] != OOB ? a2_off[rr] + koff : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA2, LDS_PTR(sA + (rr * 256 + wave * 64) * 8), 16, off, 0, 0, 0);
                }
            } else if (!p.upsample) {
                const unsigned toff = (unsigned)((((long)ky * p.W + kx) * p.lda + ci0) * ES);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                        const unsigned off = ((a_mask[rr] >> tap) & 1u) ? a_off[rr] + toff : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sA + (rr * 256 + wave * 64) * 8), 16, off, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                        const int sy = (g_oy[rr] + ky) >> 1, sx = (g_ox[rr] + kx) >> 1;
                    const unsigned off = ((a_mask[rr] >> tap) & 1u)
                                             ? (unsigned)((((long)g_img[rr] + (long)sy * p.W + sx) * p.lda + ci0 + schunk * 8) * ES) : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sA + (rr * 256 + wave * 64) * 8), 16, off, 0, 0, 0);
                }
            }
        } else {
            const int tap = k / p.Cin;
            const int ci = k - tap * p.Cin;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const int VH = p.upsample ? 2 * p.H : p.H, VW = p.upsample ? 2 * p.W : p.W;
            const bool kin = k < p.K;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int vy = g_oy[rr] + ky, vx = g_ox[rr] + kx;
                const bool ok = a_mask[rr] && kin && (unsigned)vy < (unsigned)VH && (unsigned)vx < (unsigned)VW;
                const int sy = p.upsample ? (vy >> 1) : vy, sx = p.upsample ? (vx >> 1) : vx;
                const unsigned off = ok ? (unsigned)((((long)g_img[rr] + (long)sy * p.W + sx) * p.lda + ci) * ES) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sA + (rr * 256 + wave * 64) * 8), 16, off, 0, 0, 0);
            }
        }
        const bool kwin = k < p.Kw;
        const unsigned kboff = (unsigned)kbase * ES;
#pragma unroll
        for (int rr = 0; rr < BROUNDS; ++rr) {
            const unsigned off = (kwin && b_off[rr] != OOB) ? b_off[rr] + kboff : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(sB + (rr * 256 + wave * 64) * 8), 16, off, 0, 0, 0);
        }
    };

    f4_t acc[NT][4];  // [n tile j][m tile i]
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f4_t{0.f, 0.f, 0.f, 0.f};

    int kt_begin = 0, nt = (p.K + BK - 1) / BK;   // this workgroup's K tiles [kt_begin, nt)
    if (p.split_k > 1) {
        const int sk = blockIdx.x / (gridDim.x / p.split_k);
        kt_begin = sk * p.kt_per_split;
        nt = min(nt, kt_begin + p.kt_per_split);
    }
    const int fr = lane & 15, fq = lane >> 4;

    // Fragment reads are software-pipelined by hand (hipcc otherwise sinks every ds_read next to its MFMA and waits
    // lgkmcnt(0) four times per tile): the k32-half-0 reads, the first half of its MFMAs, then the half-1 reads are
    // issued UNDER the remaining half-0 MFMAs, so only the first read burst of a tile is exposed.
    auto compute = [&](int buf) {
        const E* sA = smem + buf * STAGE;
        const E* sB = sA + A_ELEMS;
        V8 af[2][4], bf[2][NT];
        auto read_half = [&](int kk) {
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
        };
        constexpr int JH = NT / 2;  // n-tiles whose half-0 MFMAs run before the half-1 reads are issued
        read_half(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.flags & GEMM_NO_SETPRIO)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < JH; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = TT::mfma32(bf[0][j], af[0][i], acc[j][i]);
        __builtin_amdgcn_sched_barrier(0);
        read_half(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = JH; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = TT::mfma32(bf[0][j], af[0][i], acc[j][i]);
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = TT::mfma32(bf[1][j], af[1][i], acc[j][i]);
        if (!(p.flags & GEMM_NO_SETPRIO)) __builtin_amdgcn_s_setprio(0);
    };

    // ---- wide epilogue (16-bit outputs, N % 8 == 0).  Measured: storing the accumulator layout directly -- 8 bytes per
    // lane, 32-byte row fragments -- costs ~800 cycles per store instruction (16 k cycles per workgroup, more than three
    // K tiles).  Instead each wave transposes its 64 x WN tile through LDS, 16 rows (one MFMA tile row) at a time, in
    // fp32 -- so bias / row bias / residual are still summed before the single rounding -- and writes whole 16-byte
    // chunks of consecutive channels per lane: 8-10x fewer, fully coalesced stores; the residual is read the same way.
    // Scratch: 4 waves x 16 rows x (WN + 4) floats (17-21 KB) at `scr_base` (a free stage buffer).
    // RMODE (compile-time residual form: 0 none, 1 16-bit, 2 the fp32 stream) -- three copies of the epilogue instead of run-time
    // tests inside it: with the tests, hipcc lost count of the loads in flight at every branch join and waited `vmcnt(0)` in
    // EVERY 16-row pass -- i.e. for the previous pass's STORES to reach memory (stamps: 11-13 k cycles of epilogue body, more
    // than the whole K loop of the K = 320 GEMMs).  Now every load of the epilogue (bias, row bias, all residual rows of the
    // tile) is requested up front and retired by ONE explicit `s_waitcnt vmcnt(0)` the compiler's wait-count pass can see;
    // after it only stores are in flight and nothing waits for them.
    auto wide_epilogue = [&](auto rmode_tag, int em0, int en0, float* scr_base) {
        constexpr int RMODE = decltype(rmode_tag)::value;
        constexpr int WN = NT * 16;
        constexpr int SP = WN + 4;   // fp32 scratch row pitch (floats); +16 B keeps rows off the same banks
        const bool gg = p.flags & GEMM_GEGLU;
        const float* bias = p.bias;
        const float* rowbias = p.rowbias;
        // residual: 16-bit, or the fp32 residual-stream carrier (res_f32); output: 16-bit C and / or the fp32 carrier C32
        const E* res = RMODE == 1 ? reinterpret_cast<const E*>(p.residual) : nullptr;
        const float* res32 = RMODE == 2 ? reinterpret_cast<const float*>(p.residual) : nullptr;
        E* Cout = reinterpret_cast<E*>(p.C);
        float* C32 = p.C32;
        float* colstats = p.colstats;
        const bool want_stats = colstats && !(p.flags & 0x4000);
        const bool diag = p.flags & 0x4000;          // diagnostic launches (tools/stamp_gemm.py): s_memtime stamps around the parts
        auto estamp = [&]() -> unsigned long long {
            if (!diag) return 0;
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            return t;
        };
        const unsigned long long e_in = estamp();
        float* scr = scr_base + wave * (16 * SP);
        const int OW = gg ? WN / 2 : WN;              // output columns of this wave
        const int CH = OW >> 3;                       // 16-byte chunks per output row
        const int LPR = 64 / CH;                      // rows covered per read pass
        const bool act = lane < LPR * CH;
        const int rch = lane % CH, rrow = lane / CH;
        const int ncol = (gg ? ((en0 + wn * WN) >> 1) : (en0 + wn * WN)) + rch * 8;   // first output channel of this lane
        const int nout = gg ? (p.N >> 1) : p.N;
        // The body below is written for INSTRUCTION COUNT: stamps showed the epilogue at 10-11 k cycles per wave with every
        // global access ablated -- ~800 executed instructions (64-bit row-address products per store, per-row bounds tests,
        // one LDS round trip at a time), two waves per SIMD.  Now: one pointer per lane and a wave-uniform row offset per
        // store; the bounds of the wave's 64-row slab as one uniform limit; the LDS reads of a pass issued together.
        const int wrow0 = em0 + wm * 64;              // first row of this wave's 64-row slab
        const int lim = min(p.M - wrow0, 64);         // rows of the slab that exist (uniform; <= 0: none)
        const bool colok = act && ncol < nout;
        float s8[8], q8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s8[e] = q8[e] = 0.f;
        // bias and per-sample row bias of this wave's NT column tiles: requested ONCE, up front (stamps: loading them inside
        // the row loop -- a dependent L2 round trip per 16-row pass -- was 35 % of the epilogue's 16.6 k cycles).  The row
        // bias is preloaded when the whole 128-row tile belongs to one sample (wave-uniform test), else read per row as before;
        // the two are kept apart so the fp32 sum order stays (acc + bias) + rowbias.
        float4 bj[NT], rbj[NT];
        const bool one_sample = rowbias && (em0 / p.rows_per_sample) == (min(em0 + BM - 1, p.M - 1) / p.rows_per_sample);
        {
            const float* rb0 = one_sample ? rowbias + (long)(em0 / p.rows_per_sample) * p.ld_rowbias : nullptr;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int nb = en0 + wn * WN + j * 16 + fq * 4;
                const bool in = nb < p.N;
                bj[j] = (bias && in) ? *reinterpret_cast<const float4*>(bias + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
                rbj[j] = (rb0 && in) ? *reinterpret_cast<const float4*>(rb0 + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // the residual rows of the whole tile are requested up front (never with GEGLU, so the chunk geometry is static):
        // their HBM latency then runs under the LDS transposes instead of once per 16-row pass
        constexpr int LPRC = 64 / (NT * 2), RI = (16 + LPRC - 1) / LPRC;
        // DEPTH tile rows of residual in flight: the whole tile where the registers allow; two (row i + 2 requested once row i
        // has been summed) for the fp32 stream at NT = 5 (24 registers per tile row); one for the persistent form, which has
        // none to spare -- the UNet never pairs it with a residual, tools/bench_kernels.py can
        constexpr int DEPTH = PERSIST ? 1 : ((NT == 5 && RMODE == 2) ? 2 : 4);
        V8 rres[RMODE == 1 ? DEPTH : 1][RI];
        float4 r32[RMODE == 2 ? DEPTH : 1][RI][2];
        auto load_res = [&](int i, int slot) {
#pragma unroll
            for (int it = 0; it < RI; ++it) {
                const int r = rrow + it * LPRC;
                const bool in = colok && r < 16 && i * 16 + r < lim;
                const long off = (long)(wrow0 + i * 16 + r) * p.ldr + ncol;
                if constexpr (RMODE == 1) {
                    rres[slot][it] = V8{};
                    if (in) rres[slot][it] = *reinterpret_cast<const V8*>(res + off);
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
        // bias, then row bias, summed into the accumulators here -- (acc + bias) + rowbias, the order of every epilogue of this
        // file -- so their 2 x NT x 4 registers are free again before the passes start
        // (unconditional: absent terms were loaded as zeros -- a conditional update of 80 accumulator registers made the
        //  compiler keep two copies of them)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[j][i][0] = (acc[j][i][0] + bj[j].x) + rbj[j].x; acc[j][i][1] = (acc[j][i][1] + bj[j].y) + rbj[j].y;
                acc[j][i][2] = (acc[j][i][2] + bj[j].z) + rbj[j].z; acc[j][i][3] = (acc[j][i][3] + bj[j].w) + rbj[j].w;
            }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): every load of this epilogue has landed; from here on only stores fly
        unsigned long long e_t = estamp();
        dbg_e0 = e_t - e_in;
        // per-lane output pointers at slab row `rrow`; the row of (tile row i, pass it) is a wave-uniform offset from them
        const bool phased = MODE != MODE_PLAIN && p.out_phase;
        E* pC = Cout + (long)(wrow0 + rrow) * p.ldc + ncol;
        float* pC32 = C32 + (long)(wrow0 + rrow) * p.ldc32 + ncol;
        const float* lrd = scr + rch * 8;             // this lane's chunk column in the scratch
        // HALF: no residual and no fp32 carrier -- nothing downstream reads the fp32 sum, so the accumulators are rounded BEFORE
        // the transpose and go through LDS as 16-bit values: half the bytes through the LDS store path (the largest single
        // item of the epilogue: stamps), one 16-byte read per output chunk instead of two.  Same values, same single
        // rounding, same statistics (taken from the rounded values, as before).
        constexpr int SPH = WN * 2 + 16;              // 16-bit scratch row pitch (bytes; rows stay 16-byte aligned)
        using V4 = typename TT::v4;
        auto passes = [&](auto half_tag) {
            constexpr bool HALF = decltype(half_tag)::value;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                {
                    const int m = wrow0 + i * 16 + fr;
                    const float* rb = (rowbias && !one_sample && m < p.M) ? rowbias + (long)(m / p.rows_per_sample) * p.ld_rowbias : nullptr;
                    float* srow = scr + fr * SP + fq * 4;
                    unsigned char* hrow = reinterpret_cast<unsigned char*>(scr) + fr * SPH + fq * 8;
                    if (!gg) {
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            float4 v = make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
                            if (rowbias && !one_sample) {
                                const int nb = en0 + wn * WN + j * 16 + fq * 4;
                                if (rb && nb < p.N) { const float4 b = *reinterpret_cast<const float4*>(rb + nb); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
                            }
                            if constexpr (HALF) *reinterpret_cast<V4*>(hrow + j * 32) = V4{from_f32<E>(v.x), from_f32<E>(v.y), from_f32<E>(v.z), from_f32<E>(v.w)};
                            else *reinterpret_cast<float4*>(srow + j * 16) = v;
                        }
                    } else if constexpr ((NT & 1) == 0) {
#pragma unroll
                        for (int jj = 0; jj < NT / 2; ++jj) {
                            float a[4], g[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) { a[r] = acc[2 * jj][i][r]; g[r] = acc[2 * jj + 1][i][r]; }
                            // (bias already summed: value rows in the even column tiles, gate rows 16 further in the odd ones)
                            const float4 v = make_float4(a[0] * gelu_erf_f(g[0]), a[1] * gelu_erf_f(g[1]), a[2] * gelu_erf_f(g[2]), a[3] * gelu_erf_f(g[3]));
                            if constexpr (HALF) *reinterpret_cast<V4*>(hrow + jj * 32) = V4{from_f32<E>(v.x), from_f32<E>(v.y), from_f32<E>(v.z), from_f32<E>(v.w)};
                            else *reinterpret_cast<float4*>(srow + jj * 16) = v;
                        }
                    }
                }
                // LDS operations of one wave execute in order: the reads below see the writes above, and the next
                // tile row's writes cannot overtake these reads.  Every lane reads (rows clamped into the scratch), only
                // the stores are predicated: the RI x 2 reads of the pass go out back to back.
                { const unsigned long long t = estamp(); dbg_ew += t - e_t; e_t = t; }
                float4 x[HALF ? 1 : RI][2];
                V8 xh[HALF ? RI : 1];
#pragma unroll
                for (int it = 0; it < RI; ++it) {       // (RI passes without GEGLU; its narrower rows need fewer: r < 16 cuts them)
                    const int rc = min(rrow + it * LPR, 15);
                    if constexpr (HALF) {
                        xh[it] = *reinterpret_cast<const V8*>(reinterpret_cast<const unsigned char*>(scr) + rc * SPH + rch * 16);
                    } else {
                        x[it][0] = *reinterpret_cast<const float4*>(lrd + rc * SP);
                        x[it][1] = *reinterpret_cast<const float4*>(lrd + rc * SP + 4);
                    }
                }
                if (diag) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t = estamp(); dbg_er += t - e_t; e_t = t; }
#pragma unroll
                for (int it = 0; it < RI; ++it) {
                    const int r = rrow + it * LPR;
                    const int srow_u = i * 16 + it * LPR;            // wave-uniform part of the slab row
                    if (colok && r < 16 && srow_u + rrow < lim) {
                        float v[8];
                        V8 o;
                        if constexpr (HALF) {
                            o = xh[it];      // already rounded: the only values anything downstream reads
                        } else {
                            const float4 x0 = x[it][0], x1 = x[it][1];
                            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
                        }
                        if constexpr (RMODE == 1) {
                            const V8 r8 = rres[i % DEPTH][it];
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += to_f32(r8[e]);
                        }
                        if constexpr (RMODE == 2) {
                            const float4 a = r32[i % DEPTH][it][0], b = r32[i % DEPTH][it][1];
                            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
                        }
                        if constexpr (!HALF) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = from_f32<E>(v[e]);
                        }
                        if (!phased) {
                            if (Cout) *reinterpret_cast<V8*>(pC + (long)srow_u * p.ldc) = o;
                            if (!HALF && C32) {
                                float* d32 = pC32 + (long)srow_u * p.ldc32;
                                *reinterpret_cast<float4*>(d32) = make_float4(v[0], v[1], v[2], v[3]);
                                *reinterpret_cast<float4*>(d32 + 4) = make_float4(v[4], v[5], v[6], v[7]);
                            }
                        } else {
                            // one output-parity phase of conv(nearest x2 upsample): pixel (oy, ox) of the phase grid is
                            // pixel (2 oy + py, 2 ox + px) of the 2OH x 2OW output
                            const int m = wrow0 + srow_u + rrow;
                            const int hw = p.OH * p.OW, img = m / hw, rem = m - img * hw, oy = rem / p.OW, ox = rem - oy * p.OW;
                            const long orow = ((long)img * 2 * p.OH + 2 * oy + ((p.out_phase >> 1) & 1)) * (2 * p.OW) + 2 * ox + (p.out_phase & 1);
                            if (Cout) *reinterpret_cast<V8*>(Cout + orow * p.ldc + ncol) = o;
                            if (!HALF && C32) {
                                float* d32 = C32 + orow * p.ldc32 + ncol;
                                *reinterpret_cast<float4*>(d32) = make_float4(v[0], v[1], v[2], v[3]);
                                *reinterpret_cast<float4*>(d32 + 4) = make_float4(v[4], v[5], v[6], v[7]);
                            }
                        }
                        if (want_stats) {
                            // statistics of the values the following GroupNorm will read: the fp32 carrier if there is one
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
                { const unsigned long long t = estamp(); dbg_es += t - e_t; e_t = t; }
            }
        };
        if (RMODE == 0 && !C32 && !(p.flags & GEMM_F32_TRANSPOSE)) passes(std::integral_constant<bool, RMODE == 0>{});
        else passes(std::false_type{});
        if (want_stats) {
            // fold the LPR row-lanes of every channel through the scratch (fixed order: reproducible)
            if (act) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    scr[(rrow * OW + rch * 8 + e) * 2] = s8[e];
                    scr[(rrow * OW + rch * 8 + e) * 2 + 1] = q8[e];
                }
            }
            long slice = (em0 + wm * 64) >> 6;
            if (MODE != MODE_PLAIN && p.out_phase) {
                // the statistics only need every slice to lie inside one sample: phase ph of sample s files its
                // hw/64 slices at [s*4 + ph] * (hw/64) of the 4hw/64 slices the full-resolution sample owns
                const int spp = (p.OH * p.OW) >> 6;
                const long smp = slice / spp;
                slice = (smp * 4 + (p.out_phase & 3)) * spp + (slice - smp * spp);
            }
            if (em0 + wm * 64 < p.M) {
                for (int c = lane; c < OW; c += 64) {
                    float ss = 0.f, qq = 0.f;
                    for (int l = 0; l < LPR; ++l) { ss += scr[(l * OW + c) * 2]; qq += scr[(l * OW + c) * 2 + 1]; }
                    const int n = en0 + wn * WN + c;
                    if (n < p.N) *reinterpret_cast<float2*>(colstats + (slice * p.ld_colstats + n) * 2) = make_float2(ss, qq);
                }
            }
        }
    };
    auto run_wide = [&](int em0, int en0, float* scr_base) {
        if constexpr (RM >= 0) wide_epilogue(std::integral_constant<int, RM>{}, em0, en0, scr_base);
        else if (p.res_f32) wide_epilogue(std::integral_constant<int, 2>{}, em0, en0, scr_base);
        else if (p.residual) wide_epilogue(std::integral_constant<int, 1>{}, em0, en0, scr_base);
        else wide_epilogue(std::integral_constant<int, 0>{}, em0, en0, scr_base);
    };

    if constexpr (PERSIST) {
        // Persistent form (grid = 2 workgroups per CU): the two-stage K pipeline keeps running ACROSS output tiles.  While
        // the last K tile of tile T feeds the MFMAs, the first K tile of tile T+1 is already in flight into the other
        // stage, and the epilogue of T (scratch = the stage it just finished with) overlaps that load's latency; a
        // workgroup pays the launch, descriptor and first-load latencies once instead of once per tile.
        const int ntk = (p.K + BK - 1) / BK;
        int tile = blockIdx.x, g = 0;
        stage(0, 0);
        while (true) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = f4_t{0.f, 0.f, 0.f, 0.f};
            int em0 = m0, en0 = n0;
            for (int kt = 0; kt < ntk; ++kt) {
                const int cur = g & 1;
                ++g;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (kt + 1 < ntk) {
                    stage(kt + 1, cur ^ 1);
                } else {
                    em0 = m0; en0 = n0;
                    tile += gridDim.x;
                    if (tile < ntiles) {   // wave-uniform
                        tile_origin(tile);
                        tile_addresses();
                        stage(0, cur ^ 1);
                    }
                }
                compute(cur);
            }
            __syncthreads();   // every wave is done reading the last stage: it becomes the epilogue scratch
            {
                float* scr0 = reinterpret_cast<float*>(smem + ((g - 1) & 1) * STAGE);
                run_wide(em0, en0, scr0);
            }
            if (tile >= ntiles) break;
        }
        return;
    }

    if (DB) {
        // two LDS stages: tile kt+1 is in flight while tile kt feeds the MFMAs; one barrier per K tile
        stage(kt_begin, 0);
        if (p.flags & 0x4000) {
            // DIAGNOSTIC build path (never used by the engine): s_memtime stamps around the three parts of a K tile;
            // the sums go to the colstats pointer as [workgroup][wave][4] floats and feed no output value.
            auto stamp = [&]() {
                unsigned long long t;
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
                __builtin_amdgcn_sched_barrier(0);
                return t;
            };
            unsigned long long tw = 0, ts = 0, tc = 0;
            const unsigned long long t_begin = stamp();
            for (int kt = kt_begin; kt < nt; ++kt) {
                const int cur = (kt - kt_begin) & 1;
                const unsigned long long t0 = stamp();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const unsigned long long t1 = stamp();
                if (kt + 1 < nt) stage(kt + 1, cur ^ 1);
                const unsigned long long t2 = stamp();
                compute(cur);
                const unsigned long long t3 = stamp();
                tw += t1 - t0; ts += t2 - t1; tc += t3 - t2;
            }
            const unsigned long long t_end = stamp();
            dbg_t1 = t_begin; dbg_t2 = t_end;
            (void)tw; (void)ts; (void)tc;
        } else
        for (int kt = kt_begin; kt < nt; ++kt) {
            const int cur = (kt - kt_begin) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nt) stage(kt + 1, cur ^ 1);
            compute(cur);
        }
    } else {
        // one LDS stage, two barriers per K tile; latency is hidden by the other workgroups on the CU
        for (int kt = kt_begin; kt < nt; ++kt) {
            stage(kt, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();
        }
    }

    // ---- split-K: raw fp32 partial tile; bias / row bias / residual / rounding / statistics run in splitk_reduce_kernel
    if (p.split_k > 1) {
        const int sk = blockIdx.x / (gridDim.x / p.split_k);
        float* part = p.workspace + (long)sk * p.M * p.N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int nb = n0 + wn * (BN / 2) + j * 16 + fq * 4;
                if (nb < p.N)
                    *reinterpret_cast<float4*>(part + (long)m * p.N + nb) = make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
            }
        }
        return;
    }

    if (DB && !(p.flags & GEMM_OUT_F32) && !(p.N & 7) && !(p.flags & GEMM_NARROW_EPILOGUE)) {
        __syncthreads();             // all waves are done with the last K tile: the stage buffers become scratch
        unsigned long long dbg_tb = 0;
        if (p.flags & 0x4000) { __builtin_amdgcn_sched_barrier(0); dbg_tb = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        run_wide(m0, n0, reinterpret_cast<float*>(smem_raw));
        if (p.flags & 0x4000) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t3a = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t3 = __builtin_amdgcn_s_memtime();
            if (lane == 0 && p.colstats) {
                float* d = p.colstats + ((long)blockIdx.x * 4 + wave) * 4;
                // [prologue, K loop, epilogue incl. the barrier before it and the drain of its stores, barrier wait alone]
                d[0] = (float)(dbg_t1 - dbg_t0); d[1] = (float)(dbg_t2 - dbg_t1); d[2] = (float)(t3 - dbg_t2);
                d[3] = (float)(dbg_tb - dbg_t2) + 65536.f * (float)((t3 - t3a) >> 4);   // (packed: barrier wait | store drain / 16)
                float* e = p.colstats + ((long)gridDim.x * 4 + (long)blockIdx.x * 4 + wave) * 4;
                e[0] = (float)dbg_e0; e[1] = (float)dbg_ew; e[2] = (float)dbg_er; e[3] = (float)dbg_es;   // epilogue: setup | LDS writes | LDS read wait | convert + store
            }
        }
        return;
    }

    // ---- epilogue: lane holds rows n = nb + 0..3 (consecutive output channels) of column m
    const float* bias = p.bias;
    const float* rowbias = p.rowbias;
    const E* res = reinterpret_cast<const E*>(p.residual);
    const bool geglu = p.flags & GEMM_GEGLU;
    const bool out32 = p.flags & GEMM_OUT_F32;
    // optional per-channel (sum, sum of squares) of the STORED values over this wave's 64 rows: the GroupNorm that
    // consumes this tensor gets its statistics from the producer instead of re-reading the tensor (util.py:214-216)
    float* colstats = p.colstats;
    const bool want_stats = colstats && !(p.flags & 0x4000);
    float cs[NT][4], cq[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[j][r] = cq[j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
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
                    if (want_stats) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float f = to_f32(o[r]); cs[j][r] += f; cq[j][r] += f * f; }
                    }
                }
            }
        } else {
            // packed rows: every 32-row block of Wt is [16 value rows ; 16 gate rows] of the same 16 output
            // channels -> global 16-row tile index even = value, odd = gate; output channel = (n/32)*16 + n%16.
            // A wave's first tile index wn*(BN/32)... is even for BN = 128; for BN = 160 (5 tiles per wave) the
            // second wave starts on an odd tile, so pair by GLOBAL tile parity through the block's 10 tiles.
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int gt = wn * NT + j;              // tile index inside the block (BN/16 tiles)
                if (gt & 1) continue;                    // value tiles only; the gate is tile gt+1
                const int nb = n0 + gt * 16 + fq * 4;    // packed index of the value rows
                if (nb >= p.N) continue;
                const int oc = (nb >> 5) * 16 + (nb & 15);
                float a[4], g[4];
                // the gate tile lives in this wave iff j+1 < NT
                if (j + 1 < NT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { a[r] = acc[j][i][r]; g[r] = acc[j + 1 < NT ? j + 1 : j][i][r]; }
                } else {
                    continue;  // handled below through LDS exchange (BN = 160 only)
                }
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
    if (p.flags & 0x4000) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t3 = __builtin_amdgcn_s_memtime();
        if (lane == 0 && colstats) {
            float* d = colstats + ((long)blockIdx.x * 4 + wave) * 4;
            d[0] = (float)(dbg_t1 - dbg_t0); d[1] = (float)(dbg_t2 - dbg_t1); d[2] = (float)(t3 - dbg_t2); d[3] = 0.f;
        }
        return;
    }
    if (colstats && !geglu && !out32) {
        // fold the 16 rows-lanes of each column with DPP adds (quad_perm xor 1, xor 2, then row_ror 4, 8): no LDS
        auto row16 = [](float v) {
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
            v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
            return v;
        };
        const long slice = (m0 + wm * 64) >> 6;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int nb = n0 + wn * (BN / 2) + j * 16 + fq * 4;
            float s4[4], q4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { s4[r] = row16(cs[j][r]); q4[r] = row16(cq[j][r]); }
            if (fr == 0 && nb < p.N && m0 + wm * 64 < p.M) {
                float* dst = colstats + (slice * p.ld_colstats + nb) * 2;
                *reinterpret_cast<float4*>(dst) = make_float4(s4[0], q4[0], s4[1], q4[1]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(s4[2], q4[2], s4[3], q4[3]);
            }
        }
    }
}


// Second pass of a split-K launch: C = round(sum_s partial[s] + bias + rowbias + residual), 8 channels per lane, the
// same order of fp32 additions as the one-pass epilogue after the (split-ordered) K sum; per-64-row-slice column
// statistics of the stored values in a fixed order (reproducible).  Grid (ceil(M/64), ceil(N/256)), 256 threads:
// thread = (row group t>>5, 8-channel chunk t&31), rows rg + 8*it of the slice.
template <class TT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    __shared__ float red[8][256][2];
    const int t = threadIdx.x, c8 = t & 31, rg = t >> 5;
    const int n = blockIdx.y * 256 + c8 * 8;
    const int mbase = blockIdx.x * 64;
    const bool ncol_ok = n < p.N;
    const E* res = p.res_f32 ? nullptr : reinterpret_cast<const E*>(p.residual);
    const float* res32 = p.res_f32 ? reinterpret_cast<const float*>(p.residual) : nullptr;
    E* Cout = reinterpret_cast<E*>(p.C);
    float s8[8], q8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s8[e] = q8[e] = 0.f;
    float b8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) b8[e] = (p.bias && ncol_ok) ? p.bias[n + e] : 0.f;
    for (int it = 0; it < 8; ++it) {
        const int m = mbase + rg + it * 8;
        if (m >= p.M || !ncol_ok) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        for (int s = 0; s < p.split_k; ++s) {
            const float* src = p.workspace + ((long)s * p.M + m) * p.N + n;
            const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
            v[0] += x0.x; v[1] += x0.y; v[2] += x0.z; v[3] += x0.w; v[4] += x1.x; v[5] += x1.y; v[6] += x1.z; v[7] += x1.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += b8[e];
        if (p.rowbias) {
            const float* rb = p.rowbias + (long)(m / p.rows_per_sample) * p.ld_rowbias + n;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += rb[e];
        }
        if (res) {
            const V8 r8 = *reinterpret_cast<const V8*>(res + (long)m * p.ldr + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += to_f32(r8[e]);
        }
        if (res32) {
            const float* rp = res32 + (long)m * p.ldr + n;
            const float4 a = *reinterpret_cast<const float4*>(rp), b = *reinterpret_cast<const float4*>(rp + 4);
            v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
        }
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = from_f32<E>(v[e]);
        if (Cout) *reinterpret_cast<V8*>(Cout + (long)m * p.ldc + n) = o;
        if (p.C32) {
            float* d32 = p.C32 + (long)m * p.ldc32 + n;
            *reinterpret_cast<float4*>(d32) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(d32 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = p.C32 ? v[e] : to_f32(o[e]); s8[e] += f; q8[e] += f * f; }
    }
    if (p.colstats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[rg][c8 * 8 + e][0] = s8[e]; red[rg][c8 * 8 + e][1] = q8[e]; }
        __syncthreads();
        const int nn = blockIdx.y * 256 + t;
        if (nn < p.N) {
            float ss = 0.f, qq = 0.f;
#pragma unroll
            for (int l = 0; l < 8; ++l) { ss += red[l][t][0]; qq += red[l][t][1]; }
            *reinterpret_cast<float2*>(p.colstats + ((long)blockIdx.x * p.ld_colstats + nn) * 2) = make_float2(ss, qq);
        }
    }
}

// workgroups a persistent launch keeps resident: 2 per CU (LDS 64-74 KB and <= 256 VGPRs each)
int persistent_grid() {
    static std::atomic<int> g{0};
    int v = g.load(std::memory_order_relaxed);
    if (!v) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        v = 2 * cus;
        v -= v % 8;   // whole rounds of the 8 XCDs
        g.store(v, std::memory_order_relaxed);   // (every MI355X of a node has the same CU count)
    }
    return v;
}

template <class TT, int MODE, int NT, bool DB>
int launch_one(const GemmParams& p, hipStream_t stream) {
    constexpr int BN = 32 * NT;
    const int ntiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    const size_t lds = (size_t)(DB ? 2 : 1) * (BM + BN) * BK * sizeof(typename TT::elem);
    // persistent form: only where a workgroup would get more than one tile and the wide epilogue applies
    // (measured on the UNet's shapes: +1..3 % for the implicit convolutions, +-2 % noise for plain GEMMs, which keep the
    // one-tile-per-workgroup launch unless GEMM_PERSIST asks otherwise)
    // (never for the generic-window form -- Cin % 64 != 0: the UNet's 9 -> 320 input convolution --: its persistent instantiation
    //  carries all three residual forms of the epilogue and spilled 35 registers; the one-tile-per-workgroup form does not)
    const bool persist_shape = DB && MODE != MODE_CONV_GENERIC && p.split_k == 1 &&
                         !(p.flags & (GEMM_OUT_F32 | GEMM_NARROW_EPILOGUE | GEMM_NO_PERSIST | 0x4000)) &&
                         !(p.N & 7) && ntiles > persistent_grid() && (MODE != MODE_PLAIN || (p.flags & GEMM_PERSIST));
    // residual form of the wide epilogue as a kernel of its own for the forms the UNet launches (see gemm_kernel's RM)
    const int rm = p.res_f32 ? 2 : (p.residual ? 1 : 0);
    const bool persist = persist_shape && rm == 0;
    void (*kern)(GemmParams) = nullptr;
    int which = 0;
    if constexpr (DB && MODE != MODE_CONV_GENERIC) {
        // (the persistent form exists without a residual only -- what the UNet launches it with; its run-time-residual instantiation
        //  spilled 22-41 registers and is gone: a persistent launch with a residual runs one tile per workgroup)
        if (persist) { which = 1; kern = gemm_kernel<TT, MODE, NT, DB, DB, 0>; }
        else { which = 3 + rm; kern = rm == 2 ? gemm_kernel<TT, MODE, NT, DB, false, 2> : rm == 1 ? gemm_kernel<TT, MODE, NT, DB, false, 1> : gemm_kernel<TT, MODE, NT, DB, false, 0>; }
    } else {
        which = 0;
        kern = gemm_kernel<TT, MODE, NT, DB, false>;
    }
    static VfOncePerDevice attr_set[6];
    if (lds > 64 * 1024 && !attr_set[which].set_lds(reinterpret_cast<const void*>(kern), (int)lds)) return VF_ERR_LAUNCH;
    dim3 grid(persist ? persistent_grid() : ntiles * p.split_k);
    GemmParams q = p;
    if (MODE == MODE_PLAIN && p.split_k == 1 && !p.tile_group) {
        // column-group width of the XCD-aware tile order (see conv.hip launch_patch): an XCD's share of the grid is a block of
        // (tiles_xcd / GN) m-tiles x GN n-tiles; bytes through its L2 are least at GN = sqrt(tiles_xcd * A_mt / W_nt), as long as
        // the GN weight panels (re-used over many rounds here) stay resident: at most half of the 4 MiB L2
        const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
        const double a_mt = (double)BM * p.K * 2.0, w_nt = (double)BN * p.K * 2.0;
        int gn = (int)(__builtin_sqrt((double)ntm * ntn / 8.0 * a_mt / w_nt) + 0.5);
        const int cap = (int)(2097152.0 / w_nt);
        if (gn > cap) gn = cap;
        q.tile_group = gn < 1 ? 1 : (gn > ntn ? ntn : gn);
        if (p.flags & GEMM_GN8) q.tile_group = 8;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, q);
    if (p.split_k > 1)
        hipLaunchKernelGGL(splitk_reduce_kernel<TT>, dim3((p.M + 63) / 64, (p.N + 255) / 256), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

template <class TT, int MODE>
int launch_mode(const GemmParams& p, int variant, hipStream_t stream) {
    switch (variant) {
        case 5: return launch_one<TT, MODE, 4, true>(p, stream);   // 128x128x64, two stages
        case 6: return launch_one<TT, MODE, 5, true>(p, stream);   // 128x160x64, two stages
        case 7: return launch_one<TT, MODE, 4, false>(p, stream);  // 128x128x64, one stage
        default: return launch_one<TT, MODE, 5, false>(p, stream); // 128x160x64, one stage
    }
}

template <class TT>
int launch_gemm(const GemmParams& p, int variant, hipStream_t stream) {
    if (p.mode == 0) return launch_mode<TT, MODE_PLAIN>(p, variant, stream);
    if (p.Cin % 64 == 0) return launch_mode<TT, MODE_CONV_FAST>(p, variant, stream);
    return launch_mode<TT, MODE_CONV_GENERIC>(p, variant, stream);
}

// Schedule choice.  `flags` bits 8..11 force a variant (A/B benchmarking); 0 = automatic.
int pick_variant(const GemmParams& p) {
    const int forced = (p.flags >> 8) & 0xF;
    if (forced) return forced;
    const bool geglu = p.flags & GEMM_GEGLU;
    // BN = 160 moves 10 % fewer L2->LDS bytes per FLOP.  Where N is a multiple of 128 as well (640, 1280, 1920) it only
    // pays once the grid is several rounds deep (measured: +5..10 % at >= 1536 tiles, -3..5 % below: 160-wide tiles
    // quantise a 1-2 round grid worse)
    // (grid depth taken at the nominal 24-sample batch when the per-sample geometry is known, like split_for: the column
    // statistics are folded in a tile-shape-dependent order, so the choice must not depend on the batch)
    const long mref = p.rows_per_sample > 1 ? 24L * p.rows_per_sample : p.M;
    const long tiles160 = ((mref + BM - 1) / BM) * (p.N / 160);
    const bool n160 = !geglu && (p.N % 160 == 0) && ((p.N % 128 != 0) || tiles160 >= 1536);
    return n160 ? 6 : 5;
}

// Split-K factor for a launch whose tile grid would leave most of the 256 CUs (2 workgroups each) idle while every
// workgroup walks a long K loop (the 8x8-level convolutions: 120 tiles x 180..360 K tiles).  Depends on the shape only.
// The factor is a function of the PER-SAMPLE geometry when the caller states it (rows_per_sample = OH*OW of a conv, h*w
// of a token matrix): the tile count is taken at a nominal batch of 24 samples, so a frame's bits do not depend on how
// many other frames share its launch (the fp32 summation order changes with the split) -- what frame sharding relies on.
int split_for(int M, int N, int K, int flags, int rows_per_sample) {
    if (flags & (GEMM_GEGLU | GEMM_OUT_F32)) return 1;
    if ((N & 7) || ((flags >> 8) & 0xF) == 0xF) return 1;
    const bool n160 = (N % 160 == 0) && (N % 128 != 0);
    const int bn = n160 ? 160 : 128;
    const long mref = rows_per_sample > 1 ? 24L * rows_per_sample : M;
    const int tiles = (int)((mref + BM - 1) / BM) * ((N + bn - 1) / bn);
    const int nt = (K + BK - 1) / BK;
    if (tiles > 192 || nt < 16) return 1;
    int s = 512 / tiles;
    if (s > 8) s = 8;
    if (s > nt / 8) s = nt / 8;
    return s < 2 ? 1 : s;
}

}  // namespace

// The file is compiled as two translation units so that the two element types build in parallel (half the wall time of
// `make -j`): VF_GEMM_TU == 1 holds the bf16 instantiations behind vf_launch_gemm_bf16, the default unit everything else.
int vf_launch_gemm_bf16(const GemmParams& p, int variant, hipStream_t stream);
#if defined(VF_GEMM_TU) && VF_GEMM_TU == 1
int vf_launch_gemm_bf16(const GemmParams& p, int variant, hipStream_t stream) { return launch_gemm<BF16>(p, variant, stream); }
#else
// (rounds 1-3 kept four experimental schedules -- 3-stage / single-stage / BK = 32 pipelines, a 256-row ping-pong kernel -- behind
// `make VARIANTS=1`; all measured slower than this kernel on every shape of the UNet (HISTORY.md) and were removed in round 4.
// The entry point stays in the ABI and answers 0.)
bool vf_gemm_variants_built() { return false; }

long vf_splitk_workspace_bytes(int M, int N, int K, int flags, int rows_per_sample) {
    int s = split_for(M, N, K, flags, rows_per_sample);
    if (s < 2) s = 0;
    if (rows_per_sample == 64 && (N % 128) == 0 && !(flags & (GEMM_GEGLU | GEMM_OUT_F32))) {
        // a 3x3 convolution on 8x8 images may take the Q8 form (conv.hip: vf_conv_q8_split), which always ends in fp32 partials:
        // room for the largest split that rule can choose for this N
        int s8 = (int)(256 / (12L * (N / 128)));      // (vf_conv_q8_split's nominal 48-sample grid)
        if (s8 > 8) s8 = 8;
        if (s8 < 1) s8 = 1;
        if (s8 > s) s = s8;
    }
    return (long)s * M * N * 4;
}

int vf_launch_gemm(const GemmParams& p_in, int dtype, hipStream_t stream) {
    GemmParams p = p_in;
    p.split_k = 1; p.kt_per_split = 0;
    if (!p.A || !p.Wt || (!p.C && !p.C32) || !p.zeros) return VF_ERR_ARG;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return VF_ERR_ARG;
    if (p.res_f32 && !p.residual) p.res_f32 = 0;
    if (p.res_f32 || p.C32) {
        // the fp32 residual stream lives in the wide epilogue (and the split-K reduce) only
        if ((p.flags & (GEMM_GEGLU | GEMM_OUT_F32 | GEMM_NARROW_EPILOGUE | 0x4000)) || (p.N & 7)) return VF_ERR_SHAPE;
        if (p.C32 && (((uintptr_t)p.C32 & 15) || (p.ldc32 & 3))) return VF_ERR_ALIGN;
        if (p.res_f32 && (((uintptr_t)p.residual & 15) || (p.ldr & 3))) return VF_ERR_ALIGN;
        if (p.out_phase && p.res_f32) return VF_ERR_SHAPE;
    }
    if ((p.K & 7) || (p.Kw & 7) || (p.lda & 7) || (p.ldw & 7) || (p.N & 3) || (p.ldc & 3)) return VF_ERR_ALIGN;
    if (((uintptr_t)p.A | (uintptr_t)p.Wt | (uintptr_t)p.zeros) & 15) return VF_ERR_ALIGN;
    if ((uintptr_t)p.C & 7) return VF_ERR_ALIGN;
    if (p.C32 && (((uintptr_t)p.C & 15) || (p.ldc & 7))) return VF_ERR_ALIGN;
    if (p.residual && (((uintptr_t)p.residual & 7) || (p.ldr & 3))) return VF_ERR_ALIGN;
    if (p.rowbias && (p.rows_per_sample <= 0 || (p.ld_rowbias & 3))) return VF_ERR_ARG;
    if ((p.flags & GEMM_GEGLU) && ((p.N & 31) || (p.flags & GEMM_OUT_F32) || p.residual || p.rowbias)) return VF_ERR_SHAPE;
    if (p.colstats && ((p.flags & (GEMM_GEGLU | GEMM_OUT_F32)) || (p.ld_colstats & 1) || ((uintptr_t)p.colstats & 15))) return VF_ERR_ARG;
    if (p.A2 && (p.K1 <= 0 || (p.K1 % BK) || (p.lda2 & 7) || ((uintptr_t)p.A2 & 15))) return VF_ERR_ALIGN;
    if (p.A2 && p.mode == 1 && ((p.Cin % 64) || ((p.K - p.K1) % BK) || p.upsample || p.a2_row_mod)) return VF_ERR_SHAPE;
    if (p.mode == 1) {
        if (p.ntaps == 0) { p.KH = p.KW = 3; p.ntaps = 9; p.pad_x = p.pad; }   // the plain 3x3 window
        if (p.Cin <= 0 || (p.Cin & 7) || p.ntaps != p.KH * p.KW || p.ntaps > 9) return VF_ERR_SHAPE;
        if (p.A2 ? (p.K1 != p.ntaps * p.Cin || p.K <= p.K1) : (p.K != p.ntaps * p.Cin)) return VF_ERR_SHAPE;
        if (p.out_phase && (p.residual || (p.flags & GEMM_OUT_F32) || (p.N & 7))) return VF_ERR_SHAPE;
        if (p.out_phase && p.colstats && ((p.OH * p.OW) & 63)) return VF_ERR_SHAPE;
        if (p.stride != 1 && p.stride != 2) return VF_ERR_SHAPE;
    }
    // extents of the operand views for the buffer descriptors (bytes; must stay below 4 GiB - 16)
    {
        const unsigned long es = 2, lim = 0xFFFFFFF0ul;
        const unsigned long rows = p.mode == 1 ? (unsigned long)(p.M / (p.OH * p.OW)) * p.H * p.W : (unsigned long)p.M;
        const unsigned long ab = ((rows - 1) * (unsigned long)p.lda + (p.mode == 1 ? p.Cin : (p.A2 ? p.K1 : p.K))) * es;
        const unsigned long wb = ((unsigned long)(p.N - 1) * p.ldw + p.Kw) * es;
        unsigned long a2b = 0;
        if (p.A2) a2b = ((unsigned long)((p.a2_row_mod > 0 ? p.a2_row_mod : p.M) - 1) * p.lda2 + (p.K - p.K1)) * es;
        if (ab >= lim || wb >= lim || a2b >= lim) return VF_ERR_SHAPE;
        p.a_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb; p.a2_bytes = (unsigned)a2b;
    }
    // stride-1 convolutions over images that tile into 16 x 16 pixel patches run the patch-staged kernel (conv.hip): every
    // input pixel goes through LDS once per channel chunk instead of once per tap.  The choice looks at the PER-SAMPLE
    // geometry only (grid depth taken at the nominal 24-sample batch): the two kernels produce the same output bits but
    // slice the column statistics differently, so a sample's bits must not depend on which other samples share its launch.
    if (p.gn_ab) {
        if (p.mode != 1 || p.M <= 0) return VF_ERR_SHAPE;
        const unsigned long nimg = (unsigned long)(p.M / (p.OH * p.OW));
        const unsigned long gb = ((nimg - 1) * (unsigned long)p.ld_gn_ab + p.Cin) * 8ul;
        if (gb >= 0xFFFFFFF0ul) return VF_ERR_SHAPE;
        p.gn_ab_bytes = (unsigned)gb;
    }
    // the UNet's 9 (16 stored) -> 320 input convolution: K = 144 in one MFMA pass, bound by its stores (inconv.hip)
    if (vf_conv_in16_ok(p)) return vf_launch_conv_in16(p, dtype, stream);
    int karg = 0;
    const int kchoice = vf_conv_kernel_choice(p, &karg);      // (conv.hip: the one statement of the rule)
    if (kchoice == 1) return vf_launch_conv_patch(p, dtype, stream);
    if (kchoice == 2 && p.workspace) {
        // the 8x8 level: four images per workgroup through the patch-staged kernel, K split over channel chunks, then the
        // ordinary split-K reduce (which runs the whole epilogue)
        const int s = karg;
        if (s && p.workspace_bytes >= (long)s * p.M * p.N * 4 && !((uintptr_t)p.workspace & 15) && (p.M % 64) == 0 &&
            !(p.residual && (((uintptr_t)p.residual & 15) || (p.ldr & 7))) && !(p.ldc & 7) && !((uintptr_t)p.C & 15)) {
            p.split_k = s;
            const int rc = vf_launch_conv_q8(p, dtype, stream);
            if (rc != VF_OK) return rc;
            if (dtype == VF_DTYPE_F16) hipLaunchKernelGGL(splitk_reduce_kernel<F16>, dim3((p.M + 63) / 64, (p.N + 255) / 256), dim3(256), 0, stream, p);
            else hipLaunchKernelGGL(splitk_reduce_kernel<BF16>, dim3((p.M + 63) / 64, (p.N + 255) / 256), dim3(256), 0, stream, p);
            return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
        }
    }
    if (p.mode == 0 && (p.flags & GEMM_PATCH) && !((p.flags >> 8) & 0xF) && vf_gemm_patch_tile(p)) return vf_launch_gemm_patch(p, dtype, stream);
    // long token matrices: gemm_big.hip's 256 x 320 tile (the one statement of the rule: vf_gemm_big_choice)
    if (p.mode == 0 && vf_gemm_big_choice(p)) return vf_launch_gemm_big(p, dtype, stream);
    if (p.gn_ab) return VF_ERR_SHAPE;   // the fused input normalisation exists in the patch-staged kernel only
    const int variant = pick_variant(p);
    if ((p.res_f32 || p.C32) && variant != 5 && variant != 6) return VF_ERR_SHAPE;
    if (p.mode == 1 && (p.ntaps != 9 || p.out_phase || p.A2) && variant != 5 && variant != 6) return VF_ERR_SHAPE;
    if (p.workspace && !p.out_phase && (variant == 5 || variant == 6) && !((p.flags >> 8) & 0xF) && !(p.flags & 0x4000)) {
        const int s = split_for(p.M, p.N, p.K, p.flags, p.rows_per_sample);
        if (s > 1 && p.workspace_bytes >= (long)s * p.M * p.N * 4 && !((uintptr_t)p.workspace & 15) &&
            !(p.residual && (((uintptr_t)p.residual & 15) || (p.ldr & 7))) && !(p.ldc & 7) && !((uintptr_t)p.C & 15)) {
            const int nt = (p.K + BK - 1) / BK;
            p.split_k = s;
            p.kt_per_split = (nt + s - 1) / s;
            if ((long)(s - 1) * p.kt_per_split >= nt) p.split_k = (nt + p.kt_per_split - 1) / p.kt_per_split;  // no empty split
        }
    }
    if ((p.flags & GEMM_GEGLU) && (variant == 2 || variant == 4 || variant == 6 || variant == 8 || variant == 10)) return VF_ERR_SHAPE;
    if (p.colstats && (variant < 5 || variant > 8)) return VF_ERR_SHAPE;
    if ((variant >= 1 && variant <= 4) || variant == 9 || variant == 10) return VF_ERR_SHAPE;      // (removed experimental schedules)
    if (dtype == VF_DTYPE_F16) return launch_gemm<F16>(p, variant, stream);
    if (dtype == VF_DTYPE_BF16) return vf_launch_gemm_bf16(p, variant, stream);
    return VF_ERR_DTYPE;
}
#endif  // VF_GEMM_TU
