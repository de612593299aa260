// Shared device helpers for the VFace gfx950 kernels (CDNA4, wave64).  Written for MI355X only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));
typedef __bf16 b4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef float f16_t __attribute__((ext_vector_type(16)));
typedef short s4_t __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// 16-bit storage types the path computes in.  fp16 is the reference's autocast type; bf16 is the
// north-star's MFMA type.  Same MFMA rate and fragment layout on gfx950 (guide §3).
struct F16 {
    using elem = _Float16;
    using v8 = h8_t;
    using v4 = h4_t;
    static __device__ __forceinline__ f4_t mfma32(v8 a, v8 b, f4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f4_t mfma16(v4 a, v4 b, f4_t c) {  // 16x16x16: lane holds k = 4*(lane>>4) + j
        return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
    }
    // 32x32x16: lane l holds A[row l & 31][k = 8 (l >> 5) + j], B[k = 8 (l >> 5) + j][col l & 31]; C/D: col = l & 31,
    // row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5)
    static __device__ __forceinline__ f16_t mfma32x32(v8 a, v8 b, f16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    // in-place accumulation pinned to ONE accumulator-register tuple (the register allocator otherwise renames loop-carried
    // 16-register accumulators and pays for it in v_accvgpr copies every iteration).  Operands must not have been written by
    // a vector instruction in the two preceding issue slots (callers: fragments come from LDS reads, B operands are old)
    static __device__ __forceinline__ void mfma32x32_acc(f16_t& c, v8 a, v8 b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
    // the same pinned to VECTOR registers (an accumulator that vector instructions read afterwards: no v_accvgpr_read per value),
    // and its first step with C = 0.  NOTE for callers: no wait states are inserted around inline asm -- a vector instruction
    // that reads the result needs >= 18 idle issue slots after the last MFMA that wrote it (gap_mfma_result_to_valu()).
    static __device__ __forceinline__ void mfma32x32_vacc(f16_t& c, v8 a, v8 b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void mfma32x32_vzero(f16_t& c, v8 a, v8 b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ v4 tr_read(const elem* lds) {
        return __builtin_bit_cast(v4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                          (__attribute__((address_space(3))) s4_t*)(lds)));
    }
};
struct BF16 {
    using elem = __bf16;
    using v8 = b8_t;
    using v4 = b4_t;
    static __device__ __forceinline__ f4_t mfma32(v8 a, v8 b, f4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f4_t mfma16(v4 a, v4 b, f4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s4_t, a), __builtin_bit_cast(s4_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f16_t mfma32x32(v8 a, v8 b, f16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void mfma32x32_acc(f16_t& c, v8 a, v8 b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
    // the same pinned to VECTOR registers (an accumulator that vector instructions read afterwards: no v_accvgpr_read per value),
    // and its first step with C = 0.  NOTE for callers: no wait states are inserted around inline asm -- a vector instruction
    // that reads the result needs >= 18 idle issue slots after the last MFMA that wrote it (gap_mfma_result_to_valu()).
    static __device__ __forceinline__ void mfma32x32_vacc(f16_t& c, v8 a, v8 b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ void mfma32x32_vzero(f16_t& c, v8 a, v8 b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
    }
    static __device__ __forceinline__ v4 tr_read(const elem* lds) {
        return __builtin_bit_cast(v4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                          (__attribute__((address_space(3))) s4_t*)(lds)));
    }
};

template <class E> __device__ __forceinline__ float to_f32(E x) { return (float)x; }
template <class E> __device__ __forceinline__ E from_f32(float x) { return (E)x; }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// F.gelu (erf form, attention.py:44) = x Phi(x) with Phi from Abramowitz & Stegun 7.1.26 (|error| <= 7.5e-8 absolute in Phi):
//   erfc(z) = (a1 t + ... + a5 t^5) exp(-z^2),  t = 1 / (1 + p z),  z = |x| / sqrt(2)
//   x Phi(x) = max(x, 0) - |x| u,   u = erfc(z) / 2
// (x >= 0: x (1 - u); x < 0: x u = -|x| u.)  Two transcendentals (v_rcp, v_exp) and ten multiply-adds with the 1/2, 1/sqrt(2)
// and log2(e) folded into the constants -- four fewer vector instructions per value than 0.5 x (1 + copysign(1 - 2u, x)), and no
// 1 - (1 - 2u) cancellation for x << 0.  The fused FeedForward (ffn.hip) issues exactly this sequence piecewise.
constexpr float GELU_P = 0.3275911f * 0.70710678118654752440f;                   // p / sqrt(2)
constexpr float GELU_A1 = 0.5f * 0.254829592f, GELU_A2 = 0.5f * -0.284496736f, GELU_A3 = 0.5f * 1.421413741f,
                GELU_A4 = 0.5f * -1.453152027f, GELU_A5 = 0.5f * 1.061405429f;
constexpr float GELU_K = -0.5f * 1.44269504088896340736f;                         // exp(-x^2 / 2) = exp2(K x^2)
__device__ __forceinline__ float gelu_erf_f(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(GELU_P, ax, 1.0f));
    float poly = fmaf(GELU_A5, t, GELU_A4);
    poly = fmaf(poly, t, GELU_A3);
    poly = fmaf(poly, t, GELU_A2);
    poly = fmaf(poly, t, GELU_A1);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(GELU_K * (x * x));
    return fmaf(-ax, poly * e, fmaxf(x, 0.0f));
}

// max over the 4 lanes {l, l^16, l^32, l^48} with VALU lane swaps (no LDS round trip)
__device__ __forceinline__ float quad_row_max(float v) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    u = __builtin_bit_cast(unsigned, v);
    auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}
__device__ __forceinline__ float quad_row_sum(float v) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    u = __builtin_bit_cast(unsigned, v);
    auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}

// raw workgroup barrier (no fence, no vmcnt drain: LDS-DMA stays in flight across it; callers place their own counted waits).
// Behind a __device__ function: called straight from a __global__ template, the builtin makes hipcc's HOST pass drop the
// kernel's stub without a diagnostic (undefined kernel symbol at dlopen).
__device__ __forceinline__ void raw_barrier() { __builtin_amdgcn_s_barrier(); }

// Wait states hipcc would insert itself if the MFMAs around them were not inline asm (invisible to its hazard recogniser).  Each
// takes the value it protects as a "+v" operand: a plain asm nop orders nothing -- hipcc moves register-only instructions across
// it (guide 5.7 item 3) -- whereas this pins every producer of `x` above the nop and every consumer below it.
template <class T> __device__ __forceinline__ void gap_mfma_result_to_valu(T& x) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(x)); }
template <class T> __device__ __forceinline__ void gap_valu_result_to_mfma(T& x) { asm volatile("s_nop 3" : "+v"(x)); }
template <class T> __device__ __forceinline__ void gap_valu_result_to_acc_mfma(T& x) { asm volatile("s_nop 3" : "+a"(x)); }   // v_accvgpr_write -> MFMA SrcC
template <class T> __device__ __forceinline__ void gap_acc_result_to_valu(T& x) { asm volatile("s_nop 15\n\ts_nop 3" : "+a"(x)); }

// One LDS-DMA instruction (16 bytes per lane, lane-linear LDS destination) as inline asm: through the builtin, hipcc treats the
// LDS write as a store it must order against every later LDS read it cannot disambiguate and drains the whole DMA queue
// (`s_waitcnt vmcnt(0)`) before the first ds_read after an issue -- which serialises a multi-stage ring.  Here the caller
// owns the counting (`s_waitcnt vmcnt(N)`) and the visibility (barrier).  m0 = LDS byte address of lane 0's slot.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_t raw_buffer_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    return i32x4_t{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xFFFFu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void raw_lds_dma16(i32x4_t rsrc, unsigned lds_addr, int voffset, int soffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// error codes of the C ABI
enum {
    VF_OK = 0,
    VF_ERR_ARG = -1,      // null pointer / non-positive size
    VF_ERR_ALIGN = -2,    // pointer or leading dimension not aligned as the kernel requires
    VF_ERR_SHAPE = -3,    // unsupported shape (head dim, channel multiple, ...)
    VF_ERR_DTYPE = -4,
    VF_ERR_LAUNCH = -5,   // hipGetLastError() after launch
};

#define VF_DTYPE_F16 0
#define VF_DTYPE_BF16 1
