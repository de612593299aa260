// Linear layers on a HANDFUL of rows (gfx950): the time-embedding chain of the UNet,
//     emb = time_embed( timestep_embedding(t) )          (openaimodel.py:874-875; util.py:151-171; time_embed = Linear, SiLU, Linear)
//     emb_out[block] = emb_layers(emb) = Linear(SiLU(emb))   for every ResBlock (openaimodel.py:264-271), all blocks as one matrix
// on M = 3F rows (24 for an 8-frame clip).  Through the tiled GEMM these were three launches of 25-39 us -- a 128-row tile is 81 %
// padding at M = 24 and a 20 160-column output gives 126 workgroups -- plus two SiLU launches: 117 us per forward for 52 MB
// of weights (now 41).  Here:
//   out[m][n] = act( sum_k a[m][k] W[n][k] + bias[n] ),   M <= 96
// * a workgroup owns 32 output channels (630 workgroups at N = 20 160); its four waves split K and each streams its quarter of
//   the 32 weight rows straight from HBM into mfma_f32_32x32x16 A fragments, EVERY load of the wave in flight before the first
//   MFMA (one memory latency per launch instead of one per K tile);
// * the activations (L2-resident: M x K x 2 bytes) are the B fragments, 32 rows per tile, up to three tiles;
// * the four partial sums meet in LDS in a fixed order, then bias, optional SiLU, ONE rounding.
// (Computing the sinusoidal embedding inside the first layer's launch was tried: 40 sinf / cosf per lane, repeated by all 160 waves,
// took 34 us against 7.9 + 6 for the embedding kernel and a plain launch -- the embedding stays its own launch.)
#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

constexpr int LS_MAXT = 3;               // token tiles of 32 rows
constexpr int LS_PITCH = 36;             // floats per token row of the reduction buffer (32 channels + 4: float4 rows, spread banks)

template <class TT, int NS>
__global__ __launch_bounds__(256) void linear_small_kernel(LinearSmallParams p) {
    using E = typename TT::elem;
    using V8 = typename TT::v8;
    __shared__ __attribute__((aligned(16))) float red[4][LS_MAXT * 32][LS_PITCH];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int kq = wave * (p.K >> 2);                       // this wave's quarter of K: NS k16 steps
    const int mt = (p.M + 31) >> 5;
    // ---- weights: row n0 + fr, 8 consecutive k per lane half and step -- all NS loads in flight
    V8 wf[NS];
    const E* wr = reinterpret_cast<const E*>(p.W) + (long)(n0 + fr) * p.ldw + kq + fh * 8;
#pragma unroll
    for (int s = 0; s < NS; ++s) wf[s] = *reinterpret_cast<const V8*>(wr + s * 16);
    for (int tile = 0; tile < mt; ++tile) {
        const int row = tile * 32 + fr;
        V8 af[NS];
        {
            const E* ar = reinterpret_cast<const E*>(p.a) + (long)row * p.lda + kq + fh * 8;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                V8 z;
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = (E)0.0f;
                af[s] = row < p.M ? *reinterpret_cast<const V8*>(ar + s * 16) : z;
            }
        }
        f16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) acc = TT::mfma32x32(wf[s], af[s], acc);
        // accumulator: column = token fr, row (channel) = (q & 3) + 8 (q >> 2) + 4 fh
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            *reinterpret_cast<float4*>(&red[wave][tile * 32 + fr][q4 * 8 + fh * 4]) =
                make_float4(acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]);
    }
    __syncthreads();
    // ---- (w0 + w1) + w2 + w3, bias, activation, one rounding: thread = (token, 4 channels)
    for (int it = t; it < mt * 32 * 8; it += 256) {
        const int tok = it >> 3, c4 = (it & 7) * 4;
        if (tok >= p.M) continue;
        const float4 a0 = *reinterpret_cast<const float4*>(&red[0][tok][c4]), a1 = *reinterpret_cast<const float4*>(&red[1][tok][c4]);
        const float4 a2 = *reinterpret_cast<const float4*>(&red[2][tok][c4]), a3 = *reinterpret_cast<const float4*>(&red[3][tok][c4]);
        const float4 b = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float v[4] = {(((a0.x + a1.x) + a2.x) + a3.x) + b.x, (((a0.y + a1.y) + a2.y) + a3.y) + b.y,
                      (((a0.z + a1.z) + a2.z) + a3.z) + b.z, (((a0.w + a1.w) + a2.w) + a3.w) + b.w};
        if (p.silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
        }
        if (p.out_f32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (long)tok * p.ldo + n0 + c4) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            typename TT::v4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = from_f32<E>(v[e]);
            *reinterpret_cast<typename TT::v4*>(reinterpret_cast<E*>(p.out) + (long)tok * p.ldo + n0 + c4) = o;
        }
    }
}

template <class TT>
int launch_t(const LinearSmallParams& p, hipStream_t stream) {
    const dim3 grid((unsigned)(p.N / 32));
    switch (p.K / 64) {
        case 5: hipLaunchKernelGGL((linear_small_kernel<TT, 5>), grid, dim3(256), 0, stream, p); break;
        case 10: hipLaunchKernelGGL((linear_small_kernel<TT, 10>), grid, dim3(256), 0, stream, p); break;
        case 20: hipLaunchKernelGGL((linear_small_kernel<TT, 20>), grid, dim3(256), 0, stream, p); break;
        default: return VF_ERR_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH;
}

}  // namespace

bool vf_linear_small_supported(int M, int N, int K) {
    return M > 0 && M <= LS_MAXT * 32 && N > 0 && (N % 32) == 0 && (K == 320 || K == 640 || K == 1280);
}

int vf_launch_linear_small(const LinearSmallParams& p, int dtype, hipStream_t stream) {
    if (!p.a || !p.W || !p.out) return VF_ERR_ARG;
    if (!vf_linear_small_supported(p.M, p.N, p.K)) return VF_ERR_SHAPE;
    if ((p.ldw & 7) || (p.lda & 7) || (p.ldo & 3) || p.ldw < p.K || p.lda < p.K || p.ldo < p.N) return VF_ERR_ALIGN;
    if (((uintptr_t)p.W | (uintptr_t)p.a | (uintptr_t)p.bias | (uintptr_t)p.out) & 15) return VF_ERR_ALIGN;
    if (dtype == VF_DTYPE_F16) return launch_t<F16>(p, stream);
    if (dtype == VF_DTYPE_BF16) return launch_t<BF16>(p, stream);
    return VF_ERR_DTYPE;
}
