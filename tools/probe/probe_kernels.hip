// Probe aggressor (tools/warp_coresidency_probe.py): a kernel that does nothing but issue LDS-DMA loads (`buffer_load_dwordx4 ... lds`)
// under a chosen EXEC mask -- all 64 lanes, or a partial mask as attention.hip's value-slot staging uses to keep padding slots intact.
#include <hip/hip_runtime.h>
#include <stdint.h>
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__global__ __launch_bounds__(256) void dma_partial_kernel(const void* src, unsigned bytes, int iters, unsigned long long mask, unsigned long long oob, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, (int)bytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool act = (mask >> lane) & 1ull, out = (oob >> lane) & 1ull;      // out: this lane's offset is out of range (the hardware writes zeros)
    unsigned off = (unsigned)(((blockIdx.x * 256u + threadIdx.x) * 16u) % (bytes - 16u)) & ~15u;
    for (int i = 0; i < iters; ++i) {
        if (act) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(lds + wave * 1024), 16, out ? 0xFFFFFFF0u : off, 0, 0, 0);
        off += 4096u * 16u;
        if (off >= bytes - 16u) off -= (bytes - 16u) & ~15u;
        if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && sink) sink[blockIdx.x] = (float)lds[lane];
}

extern "C" int launch_dma_partial(const void* src, unsigned bytes, int iters, unsigned long long mask, unsigned long long oob, int blocks, float* sink, void* stream) {
    hipLaunchKernelGGL(dma_partial_kernel, dim3(blocks), dim3(256), 4096, (hipStream_t)stream, src, bytes, iters, mask, oob, sink);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Probe VICTIM: every lane issues NL back-to-back 16-byte loads (MUBUF `buffer_load_dwordx4 ... offen`, or `global_load_dwordx4`) from a
// table whose 16-byte element i holds {i, i, i, i}, checks what came back and counts, per (load ordinal, quarter-wave), the loads
// that returned something else -- and how many of those returned zeros.  err[(l * 4 + quarter) * 2 + {0: wrong, 1: wrong and zero}].
template <int NL, bool MUBUF>
__global__ __launch_bounds__(256) void victim_kernel(const uint4* table, unsigned nelem, int iters, unsigned* err) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(table), 0, (int)(nelem * 16u), 0x00020000);
    const int lane = threadIdx.x & 63;
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 7u % nelem;
    for (int i = 0; i < iters; ++i) {
        unsigned want[NL];
        uint4 got[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) want[l] = (idx + (unsigned)l * 4099u) % nelem;
        if constexpr (MUBUF) {
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                typedef unsigned u4_t __attribute__((ext_vector_type(4)));
                const u4_t v = (u4_t)__builtin_amdgcn_raw_buffer_load_b128(r, (int)(want[l] * 16u), 0, 0);
                got[l] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        } else {
#pragma unroll
            for (int l = 0; l < NL; ++l) got[l] = table[want[l]];
        }
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const bool bad = got[l].x != want[l] || got[l].y != want[l] || got[l].z != want[l] || got[l].w != want[l];
            if (bad) {
                atomicAdd(&err[(l * 4 + (lane >> 4)) * 2], 1u);
                if (got[l].x == 0 && got[l].y == 0 && got[l].z == 0 && got[l].w == 0) atomicAdd(&err[(l * 4 + (lane >> 4)) * 2 + 1], 1u);
            }
        }
        idx = (idx * 5u + 12345u) % nelem;
    }
}

extern "C" int launch_victim(const void* table, unsigned nelem, int iters, int nloads, int mubuf, int blocks, unsigned* err, void* stream) {
    auto s = (hipStream_t)stream;
    const uint4* t = (const uint4*)table;
#define V(NL, MB) hipLaunchKernelGGL((victim_kernel<NL, MB>), dim3(blocks), dim3(256), 0, s, t, nelem, iters, err)
    if (mubuf) { if (nloads == 1) V(1, true); else if (nloads == 2) V(2, true); else if (nloads == 4) V(4, true); else V(8, true); }
    else { if (nloads == 1) V(1, false); else if (nloads == 2) V(2, false); else if (nloads == 4) V(4, false); else V(8, false); }
#undef V
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// The SELECT form of the flow warp's inner step, as vface_amd/csrc/pointwise.hip had it until round 5 (fp16, no previous-rank frame): the
// neighbour taps' validity as a boolean, `x1 = vx ? x0 + 1 : x0` (hipcc: an add-with-carry on the VCC lane mask) and
// `w01 = vx ? wx1 * wy0 : 0` (a v_cndmask on the same mask).  Kept HERE, outside the product, as the reproducer of what
// tools/warp_coresidency_probe.py measures: beside the dh = 32 / 40 / 80 attention kernels of another stream this form writes, in
// lanes 48..63 of sporadic waves, the value with the (y0, x1) tap missing -- the select returned its zero branch although the mask bit
// was set (the (y1, x1) weight, selected on an SGPR-pair copy of the same mask, is right).
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float unnorm_coord_probe(float pos, float d, int size) {
    const float v = __fadd_rn(pos, d);
    const float den = (float)max(size - 1, 1);
    const float qn = __fdiv_rn(__fmul_rn(2.0f, v), den);
    const float g = __fsub_rn(qn, 1.0f);
    float c = __fmul_rn(__fmul_rn(__fadd_rn(g, 1.0f), 0.5f), (float)(size - 1));
    return fminf((float)(size - 1), fmaxf(c, 0.0f));
}

__global__ __launch_bounds__(256) void warp_select_form_kernel(const _Float16* __restrict__ src, long ld_src, long fs_src, const float* __restrict__ flow,
                                                               _Float16* __restrict__ dst, long ld_dst, long fs_dst, int F, int h, int w, int C,
                                                               float alpha, float oma) {
    const int c8 = C / 8, f = blockIdx.y;
    const long total = (long)h * w * c8;
    const _Float16* cur = src + (long)f * fs_src;
    _Float16* out = dst + (long)f * fs_dst;
    const _Float16* from = f > 0 ? src + (long)(f - 1) * fs_src : nullptr;
    const float* fl = f > 0 ? flow + (long)(f - 1) * 2 * h * w : nullptr;
    const unsigned from_bytes = from ? (unsigned)((((long)h * w - 1) * ld_src + C) * 2) : 0u;
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(from ? from : cur), 0, (int)from_bytes, 0x00020000);
    const unsigned row_b = (unsigned)(ld_src * 2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int pix = (int)(i / c8), cc = (int)(i - (long)pix * c8) * 8;
        const h8v xv = *reinterpret_cast<const h8v*>(cur + (long)pix * ld_src + cc);
        if (!from) { *reinterpret_cast<h8v*>(out + (long)pix * ld_dst + cc) = xv; continue; }
        const int py = pix / w, px = pix - py * w;
        const float ix = unnorm_coord_probe((float)px, fl[pix], w), iy = unnorm_coord_probe((float)py, fl[h * w + pix], h);
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const int x0 = (int)fx0, y0 = (int)fy0;
        const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        const bool vx = x0 + 1 <= w - 1, vy = y0 + 1 <= h - 1;
        const int x1 = vx ? x0 + 1 : x0, y1 = vy ? y0 + 1 : y0;
        float p01 = wx1 * wy0;
#ifdef PROBE_SELECT_NOP
        // round 6 (tools/select_hazard_probe.py): eight idle issue slots between the instruction that forms the product and the select that reads it
        asm volatile("s_nop 7" : "+v"(p01));
#endif
        const float w00 = wx0 * wy0, w01 = vx ? p01 : 0.f, w10 = vy ? wx0 * wy1 : 0.f, w11 = (vx && vy) ? wx1 * wy1 : 0.f;
        const unsigned cb = (unsigned)cc * 2u;
        const h8v a = __builtin_bit_cast(h8v, (u4v)__builtin_amdgcn_raw_buffer_load_b128(rF, (unsigned)(y0 * w + x0) * row_b + cb, 0, 0));
        const h8v b = __builtin_bit_cast(h8v, (u4v)__builtin_amdgcn_raw_buffer_load_b128(rF, (unsigned)(y0 * w + x1) * row_b + cb, 0, 0));
        const h8v c = __builtin_bit_cast(h8v, (u4v)__builtin_amdgcn_raw_buffer_load_b128(rF, (unsigned)(y1 * w + x0) * row_b + cb, 0, 0));
        const h8v d = __builtin_bit_cast(h8v, (u4v)__builtin_amdgcn_raw_buffer_load_b128(rF, (unsigned)(y1 * w + x1) * row_b + cb, 0, 0));
        h8v o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float wv = (float)a[j] * w00;
            wv += (float)b[j] * w01;
            wv += (float)c[j] * w10;
            wv += (float)d[j] * w11;
            const float ax = (float)(_Float16)(alpha * (float)xv[j]);
            o[j] = (_Float16)(ax + oma * wv);
        }
        *reinterpret_cast<h8v*>(out + (long)pix * ld_dst + cc) = o;
#ifdef PROBE_SELECT_DIAG
        // round 6 (tools/select_hazard_probe.py --diag): what the wrong lanes actually held -- the selected weight of the (y0, x1) tap and
        // the first two dwords that tap's load returned -- into the unused columns [C, C + C/2) of the destination row (8 B per chunk)
        {
            const u4v braw = __builtin_bit_cast(u4v, b);
            u4v dg; dg[0] = __float_as_uint(w01); dg[1] = braw[0];
            *reinterpret_cast<uint2*>(out + (long)pix * ld_dst + C + (cc / 8) * 4) = make_uint2(dg[0], dg[1]);
        }
#endif
    }
}

extern "C" int launch_warp_select_form(const void* src, long ld_src, long fs_src, const float* flow, void* dst, long ld_dst, long fs_dst, int F, int h,
                                       int w, int C, float alpha, float oma, void* stream) {
    const long total = (long)h * w * (C / 8);
    long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(warp_select_form_kernel, dim3((unsigned)blocks, F), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, ld_src, fs_src, flow,
                       (_Float16*)dst, ld_dst, fs_dst, F, h, w, C, alpha, oma);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- round 6: micro-victims for tools/select_hazard_probe.py --micro.  One pinned instruction sequence per MODE (fixed physical registers
// inside one asm block, so hipcc cannot reorder or re-encode it): a PRODUCER of v54 (the low half of a packed-fp32 product v[54:55], or a
// plain v_mul_f32), FILL independent vector instructions, then a CONSUMER of v54; the result is compared in the kernel with the same value
// computed far from any packed instruction.  err[0] = wrong results, err[1 + quarter-wave] = by the owning lane's quarter.
//   MODE 0  v_pk_mul_f32 -> 1 filler -> v_cndmask_b32_e32 .., 0, v54, vcc       (the pair of the old flow warp)
//   MODE 1  v_pk_mul_f32 -> 1 filler -> v_cndmask_b32_e64 .., 0, v54, s[10:11]
//   MODE 2  v_pk_mul_f32 -> 1 filler -> v_add_f32_e32 .., v54, v56
//   MODE 3  v_mul_f32    -> 1 filler -> v_cndmask_b32_e32 .., 0, v54, vcc
//   MODE 4  v_pk_mul_f32 -> 0 filler -> v_cndmask_b32_e32
//   MODE 5  v_pk_mul_f32 -> s_nop 7  -> v_cndmask_b32_e32
//   MODE 6  v_pk_mul_f32 -> 1 filler -> v_mul_f32_e32 .., v54, v56
//   MODE 7  v_pk_mul_f32 -> 1 filler -> v_mov_b32_e32 .., v54
//   MODE 8  v_pk_mul_f32 -> 1 filler -> v_add_f32_e64 (VOP3 encoding of MODE 2)
template <int MODE>
__global__ __launch_bounds__(256) void pk_victim_kernel(const float* __restrict__ in, unsigned nelem, int iters, unsigned* err) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    float a = in[tid % nelem], c = in[(tid * 7 + 3) % nelem], e = in[(tid * 13 + 5) % nelem];
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        a = a * 1.0001f + 0.25f; c = c * 0.9999f + 0.5f;          // (positive inputs: the mask `c > 0` is set in every lane)
        float r;
        asm volatile(
            "v_mov_b32 v50, %[a]\n v_mov_b32 v51, %[e]\n v_mov_b32 v52, %[c]\n v_mov_b32 v53, %[e]\n v_mov_b32 v56, %[e]\n"
            "v_cmp_lt_f32 vcc, 0, %[c]\n s_mov_b64 s[10:11], vcc\n s_nop 7\n"
            : : [a] "v"(a), [c] "v"(c), [e] "v"(e) : "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "vcc", "s10", "s11");
        // producer, filler and consumer in ONE asm statement: nothing can be scheduled between them
#define PK_PROD "v_pk_mul_f32 v[54:55], v[50:51], v[52:53]\n"
#define PK_FILL "v_cvt_pk_f16_f32 v58, v50, v51\n"
#define PK_SEQ(txt) asm volatile(txt ::: "v54", "v55", "v57", "v58")
        if (MODE == 0) PK_SEQ(PK_PROD PK_FILL "v_cndmask_b32_e32 v57, 0, v54, vcc\n");
        else if (MODE == 1) PK_SEQ(PK_PROD PK_FILL "v_cndmask_b32_e64 v57, 0, v54, s[10:11]\n");
        else if (MODE == 2) PK_SEQ(PK_PROD PK_FILL "v_add_f32_e32 v57, v54, v56\n");
        else if (MODE == 3) PK_SEQ("v_mul_f32 v54, v50, v52\n" PK_FILL "v_cndmask_b32_e32 v57, 0, v54, vcc\n");
        else if (MODE == 4) PK_SEQ(PK_PROD "v_cndmask_b32_e32 v57, 0, v54, vcc\n");
        else if (MODE == 5) PK_SEQ(PK_PROD "s_nop 7\n" "v_cndmask_b32_e32 v57, 0, v54, vcc\n");
        else if (MODE == 6) PK_SEQ(PK_PROD PK_FILL "v_mul_f32_e32 v57, v54, v56\n");
        else if (MODE == 7) PK_SEQ(PK_PROD PK_FILL "v_mov_b32_e32 v57, v54\n");
        else PK_SEQ(PK_PROD PK_FILL "v_add_f32_e64 v57, v54, v56\n");
#undef PK_SEQ
#undef PK_FILL
#undef PK_PROD
        asm volatile("s_nop 7\n v_mov_b32 %[r], v57\n" : [r] "=v"(r) : : "v57");
        const float prod = __fmul_rn(a, c);
        const float want = (MODE == 2 || MODE == 8) ? __fadd_rn(prod, e) : (MODE == 6 ? __fmul_rn(prod, e) : prod);
        bad += (__float_as_uint(r) != __float_as_uint(want)) ? 1u : 0u;
    }
    if (bad) { atomicAdd(err, bad); atomicAdd(err + 1 + ((threadIdx.x & 63) >> 4), bad); }
}

extern "C" int launch_pk_victim(int mode, const float* in, unsigned nelem, int iters, int blocks, unsigned* err, void* stream) {
#define PKV(M) case M: hipLaunchKernelGGL(pk_victim_kernel<M>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, nelem, iters, err); break;
    switch (mode) { PKV(0) PKV(1) PKV(2) PKV(3) PKV(4) PKV(5) PKV(6) PKV(7) PKV(8) default: return -2; }
#undef PKV
    return hipGetLastError() == hipSuccess ? 0 : -1;
}


// ---- round 6, second hypothesis: in the failing build the select sits one slot behind `v_cvt_pk_f16_f32 v7, ..`, which OVERWRITES the address
// register of the `buffer_load_dwordx4 v[32:35], v7, .. offen` issued two slots earlier (a write-after-read on a VMEM address operand; hipcc
// pads nothing there).  MODE 0: load, one packed filler, then a vector write of the address register, as in that listing; MODE 1: the same with
// eight idle slots between the load and the write; MODE 2: four loads back to back first (as the warp has in flight), then MODE 0's pair.
// The loaded dword is compared with the table entry the ORIGINAL address names.  err[] as above.
template <int MODE>
__global__ __launch_bounds__(256) void vmem_war_victim_kernel(const unsigned* __restrict__ table, unsigned nelem, int iters, unsigned* err) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(table), 0, (int)(nelem * 4u), 0x00020000);
    unsigned bad = 0, idx = (tid * 2654435761u) % nelem;
    for (int it = 0; it < iters; ++it) {
        idx = (idx * 1664525u + 1013904223u) % nelem;
        const unsigned off = idx & ~3u;                 // 16-byte aligned dword index
        unsigned r;
        if (MODE == 2)
            asm volatile("v_lshlrev_b32 v59, 2, %[o]\n v_mov_b32 v50, 1.0\n v_mov_b32 v51, 2.0\n v_mov_b32 v52, 3.0\n v_mov_b32 v53, 4.0\n s_nop 7\n"
                         "buffer_load_dwordx4 v[36:39], v59, %[rs], 0 offen\n buffer_load_dwordx4 v[40:43], v59, %[rs], 0 offen\n"
                         "buffer_load_dwordx4 v[44:47], v59, %[rs], 0 offen\n"
                         "buffer_load_dwordx4 v[60:63], v59, %[rs], 0 offen\n"
                         "v_pk_mul_f32 v[54:55], v[50:51], v[52:53]\n"
                         "v_cvt_pk_f16_f32 v59, v52, v53\n"
                         "v_cndmask_b32_e32 v54, 0, v54, vcc\n"
                         "s_waitcnt vmcnt(0)\n v_mov_b32 %[r], v60\n"
                         : [r] "=v"(r) : [o] "v"(off), [rs] "s"(rs)
                         : "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v50", "v51", "v52", "v53", "v54", "v55",
                           "v59", "v60", "v61", "v62", "v63", "memory");
        else if (MODE == 1)
            asm volatile("v_lshlrev_b32 v59, 2, %[o]\n v_mov_b32 v50, 1.0\n v_mov_b32 v51, 2.0\n v_mov_b32 v52, 3.0\n v_mov_b32 v53, 4.0\n s_nop 7\n"
                         "buffer_load_dwordx4 v[60:63], v59, %[rs], 0 offen\n"
                         "s_nop 7\n"
                         "v_cvt_pk_f16_f32 v59, v52, v53\n"
                         "s_waitcnt vmcnt(0)\n v_mov_b32 %[r], v60\n"
                         : [r] "=v"(r) : [o] "v"(off), [rs] "s"(rs)
                         : "v50", "v51", "v52", "v53", "v54", "v55", "v59", "v60", "v61", "v62", "v63", "memory");
        else
            asm volatile("v_lshlrev_b32 v59, 2, %[o]\n v_mov_b32 v50, 1.0\n v_mov_b32 v51, 2.0\n v_mov_b32 v52, 3.0\n v_mov_b32 v53, 4.0\n s_nop 7\n"
                         "buffer_load_dwordx4 v[60:63], v59, %[rs], 0 offen\n"
                         "v_pk_mul_f32 v[54:55], v[50:51], v[52:53]\n"
                         "v_cvt_pk_f16_f32 v59, v52, v53\n"
                         "s_waitcnt vmcnt(0)\n v_mov_b32 %[r], v60\n"
                         : [r] "=v"(r) : [o] "v"(off), [rs] "s"(rs)
                         : "v50", "v51", "v52", "v53", "v54", "v55", "v59", "v60", "v61", "v62", "v63", "memory");
        bad += (r != table[off]) ? 1u : 0u;
    }
    if (bad) { atomicAdd(err, bad); atomicAdd(err + 1 + ((threadIdx.x & 63) >> 4), bad); }
}

extern "C" int launch_vmem_war_victim(int mode, const unsigned* table, unsigned nelem, int iters, int blocks, unsigned* err, void* stream) {
    if (mode == 0) hipLaunchKernelGGL(vmem_war_victim_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, nelem, iters, err);
    else if (mode == 1) hipLaunchKernelGGL(vmem_war_victim_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, nelem, iters, err);
    else if (mode == 2) hipLaunchKernelGGL(vmem_war_victim_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, table, nelem, iters, err);
    else return -2;
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
