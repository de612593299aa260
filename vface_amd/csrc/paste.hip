// Per-frame paste-back of a swapped crop into its original frame (REFace/scripts/VFace_inference_batch.py:597-636).
//
// The reference does this frame by frame on the HOST: the decoded crop goes device -> host as fp32, is quantised by numpy and
// resized / projected / composited by Pillow, and the background frame makes a host -> device -> host round trip through the
// VAE with a torchvision resize before and a Pillow resize after.  Here every step is a kernel on the frames where they already
// are (HBM), with the 8-bit arithmetic of Pillow restated exactly:
//
//   frame_to_u8_kernel          clamp((x + 1) / 2, 0, 1) * 255 -> uint8 (truncation), planar fp32 or 16-bit -> interleaved RGB (:597-608)
//   resample_u8_kernel<AXIS>    one pass of Pillow's 8-bit separable resampling (Resample.c): 22-bit fixed-point taps, 2^21
//                               rounding term, clip to 0..255.  The tap tables (precompute_coeffs + normalize_coeffs_8bpc of
//                               the bilinear filter) are host work: vface_amd/scripts/paste_back.py builds and uploads them.
//   perspective_paste_kernel    Image.transform(size, PERSPECTIVE, coeffs, BILINEAR) of the alpha-255 crop + alpha_composite
//                               over the background (Geometry.c perspective_transform / bilinear_filter32RGB, AlphaComposite.c):
//                               pixel CENTRES mapped in double precision, inside test on [0, w) x [0, h), border taps clamped,
//                               double blend truncated to 8 bits; with a hard 0 / 255 alpha the composite is a select.
//   frame_normalise_resize_kernel   ToTensor + Normalize(0.5, 0.5) + transforms.Resize on a tensor (= F.interpolate bilinear,
//                               align_corners = False, no antialias): uint8 RGB frame -> planar fp32 in [-1, 1] at the VAE's size.
//
// All HBM-bound byte work (a 1080p frame is 6 MB); no LDS, one thread per output pixel or sample, coalesced along x.
// Contraction is OFF for the double / float blends: Pillow (C, no FMA on generic x86-64) and ATen round every product.
#include "common.hpp"
#include "vface_kernels.hpp"

namespace {

inline int ok() { return hipGetLastError() == hipSuccess ? VF_OK : VF_ERR_LAUNCH; }

template <class T>
__device__ inline float ld_f32(const T* p, long i) { return to_f32(p[i]); }
template <>
__device__ inline float ld_f32<float>(const float* p, long i) { return p[i]; }

// x: planar [3][H][W] (one frame per blockIdx.y, frame stride 3 H W) in [-1, 1]; out: [H][W][3] uint8
// H16: the arithmetic of the reference's DEFAULT `--precision autocast` run -- decode_first_stage hands back float16, so `x + 1.0`,
// `/ 2.0` (torch, one rounding to float16 each) and numpy's `255. * x` on the float16 array all round to float16 before the
// truncation; near 255 the float16 spacing is 0.125, so a pixel can differ by one from the fp32 arithmetic of `--precision full`.
template <class T, bool H16 = false>
__global__ __launch_bounds__(256) void frame_to_u8_kernel(const T* __restrict__ x, unsigned char* __restrict__ out, int hw) {
#pragma clang fp contract(off)
    const long fb = (long)blockIdx.y * 3 * hw;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < hw; p += gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v;
            if constexpr (H16) {
                auto r16 = [](float f) { return (float)(_Float16)f; };
                v = r16(r16(ld_f32(x, fb + (long)c * hw + p) + 1.0f) / 2.0f);
                v = fminf(fmaxf(v, 0.0f), 1.0f);
                v = r16(255.0f * v);
                out[fb + (long)p * 3 + c] = (unsigned char)(int)v;
                continue;
            }
            v = (ld_f32(x, fb + (long)c * hw + p) + 1.0f) / 2.0f;
            v = fminf(fmaxf(v, 0.0f), 1.0f);                    // (torch.clamp: NaN stays NaN there; a NaN frame is already lost)
            out[fb + (long)p * 3 + c] = (unsigned char)(int)(255.0f * v);
        }
    }
}

// One separable pass.  AXIS 0: along x (src [lines][in_n][3] -> dst [lines][out_n][3]); AXIS 1: along y (src [in_n][line_len][3]
// -> dst [out_n][line_len][3]).  bounds[o] = (first input index, taps); kk[o][ksize] = 22-bit weights.
template <int AXIS>
__global__ __launch_bounds__(256) void resample_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                          int in_n, int out_n, int lines, const int* __restrict__ bounds,
                                                          const int* __restrict__ kk, int ksize, long frame_in, long frame_out) {
    src += (long)blockIdx.y * frame_in;
    dst += (long)blockIdx.y * frame_out;
    const long total = (long)out_n * lines * 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int o, line, c;
        if (AXIS == 0) { c = (int)(i % 3); o = (int)((i / 3) % out_n); line = (int)(i / (3L * out_n)); }
        else { c = (int)(i % 3); line = (int)((i / 3) % lines); o = (int)(i / (3L * lines)); }
        const int x0 = bounds[2 * o], n = bounds[2 * o + 1];
        const int* k = kk + (long)o * ksize;
        int ss = 1 << 21;
        for (int t = 0; t < n; ++t) {
            const long si = AXIS == 0 ? ((long)line * in_n + x0 + t) * 3 + c : ((long)(x0 + t) * lines + line) * 3 + c;
            ss += (int)src[si] * k[t];
        }
        ss >>= 22;
        dst[i] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
    }
}

struct Persp { double a[8]; };

// frame [H][W][3] is the background on entry and the pasted frame on return (in place: a pixel reads only itself from `frame`)
__global__ __launch_bounds__(256) void perspective_paste_kernel(const unsigned char* __restrict__ crop, int sw, int sh,
                                                                unsigned char* __restrict__ frame, int W, int H, Persp cf,
                                                                long crop_stride, long frame_stride, int coeff_stride,
                                                                const double* __restrict__ coeffs) {
#pragma clang fp contract(off)
    crop += (long)blockIdx.y * crop_stride;
    frame += (long)blockIdx.y * frame_stride;
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = coeffs ? coeffs[(long)blockIdx.y * coeff_stride + j] : cf.a[j];
    const long total = (long)W * H;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int py = (int)(i / W), px = (int)(i - (long)py * W);
        const double xc = px + 0.5, yc = py + 0.5;
        const double den = a[6] * xc + a[7] * yc + 1;
        double xin = (a[0] * xc + a[1] * yc + a[2]) / den;
        double yin = (a[3] * xc + a[4] * yc + a[5]) / den;
        if (!(xin >= 0.0 && xin < (double)sw && yin >= 0.0 && yin < (double)sh)) continue;      // transparent: background stays
        xin -= 0.5;
        yin -= 0.5;
        const int x = xin < 0.0 ? (int)floor(xin) : (int)xin;
        const int y = yin < 0.0 ? (int)floor(yin) : (int)yin;
        const double dx = xin - x, dy = yin - y;
        const int x0 = min(max(x, 0), sw - 1), x1 = min(max(x + 1, 0), sw - 1);
        const int y0 = min(max(y, 0), sh - 1);
        const bool has2 = (y + 1 >= 0) && (y + 1 < sh);
        const unsigned char* r0 = crop + (long)y0 * sw * 3;
        const unsigned char* r1 = crop + (long)(has2 ? y + 1 : y0) * sw * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double p00 = r0[x0 * 3 + c], p01 = r0[x1 * 3 + c];
            double v1 = p00 + (p01 - p00) * dx;
            double v2 = v1;
            if (has2) {
                const double p10 = r1[x0 * 3 + c], p11 = r1[x1 * 3 + c];
                v2 = p10 + (p11 - p10) * dx;
            }
            v1 = v1 + (v2 - v1) * dy;
            frame[i * 3 + c] = (unsigned char)(int)v1;
        }
    }
}

// frame [H][W][3] uint8 -> out planar [3][OH][OW] fp32 = resize_bilinear((frame / 255 - 0.5) / 0.5)
__global__ __launch_bounds__(256) void frame_normalise_resize_kernel(const unsigned char* __restrict__ frame, int W, int H,
                                                                     float* __restrict__ out, int OW, int OH, long frame_stride,
                                                                     long out_stride) {
#pragma clang fp contract(off)
    frame += (long)blockIdx.y * frame_stride;
    out += (long)blockIdx.y * out_stride;
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    const long total = (long)OW * OH;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int oy = (int)(i / OW), ox = (int)(i - (long)oy * OW);
        const float fy = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.0f), fx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.0f);
        const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
        const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            auto px = [&](int yy, int xx) { return ((float)frame[((long)yy * W + xx) * 3 + c] / 255.0f - 0.5f) / 0.5f; };
            const float top = lx0 * px(y0, x0) + lx1 * px(y0, x1);
            const float bot = lx0 * px(y1, x0) + lx1 * px(y1, x1);
            out[(long)c * total + i] = ly0 * top + ly1 * bot;
        }
    }
}

inline unsigned grid1(long total) { return (unsigned)std::min<long>((total + 255) / 256, 8192); }

}  // namespace

int vf_launch_frame_to_u8(const void* x, unsigned char* out, int frames, int H, int W, int in_kind, hipStream_t stream) {
    if (!x || !out || frames <= 0 || H <= 0 || W <= 0) return VF_ERR_ARG;
    if ((long)H * W > 0x7fffffffL / 3) return VF_ERR_SHAPE;
    const dim3 grid(grid1((long)H * W), frames);
    if (in_kind == 2) hipLaunchKernelGGL(frame_to_u8_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, out, H * W);
    else if (in_kind == 0) hipLaunchKernelGGL((frame_to_u8_kernel<F16::elem>), grid, dim3(256), 0, stream, (const F16::elem*)x, out, H * W);
    else if (in_kind == 1) hipLaunchKernelGGL((frame_to_u8_kernel<BF16::elem>), grid, dim3(256), 0, stream, (const BF16::elem*)x, out, H * W);
    else if (in_kind == 3) hipLaunchKernelGGL((frame_to_u8_kernel<F16::elem, true>), grid, dim3(256), 0, stream, (const F16::elem*)x, out, H * W);
    else return VF_ERR_ARG;
    return ok();
}

int vf_launch_resample_u8(const unsigned char* src, unsigned char* dst, int frames, int in_n, int out_n, int lines, int axis,
                          const int* bounds, const int* kk, int ksize, hipStream_t stream) {
    if (!src || !dst || !bounds || !kk || frames <= 0 || in_n <= 0 || out_n <= 0 || lines <= 0 || ksize <= 0) return VF_ERR_ARG;
    if (axis != 0 && axis != 1) return VF_ERR_ARG;
    const long fin = (long)in_n * lines * 3, fout = (long)out_n * lines * 3;
    const dim3 grid(grid1(fout), frames);
    if (axis == 0) hipLaunchKernelGGL(resample_u8_kernel<0>, grid, dim3(256), 0, stream, src, dst, in_n, out_n, lines, bounds, kk, ksize, fin, fout);
    else hipLaunchKernelGGL(resample_u8_kernel<1>, grid, dim3(256), 0, stream, src, dst, in_n, out_n, lines, bounds, kk, ksize, fin, fout);
    return ok();
}

int vf_launch_perspective_paste(const unsigned char* crop, int sw, int sh, unsigned char* frame, int W, int H, int frames,
                                const double* coeffs_dev, const double* coeffs_host, hipStream_t stream) {
    if (!crop || !frame || frames <= 0 || sw <= 0 || sh <= 0 || W <= 0 || H <= 0) return VF_ERR_ARG;
    if (!coeffs_dev == !coeffs_host) return VF_ERR_ARG;          // exactly one source of the eight coefficients
    if (coeffs_host && frames != 1) return VF_ERR_ARG;           // host coefficients ride in the kernel arguments: one frame
    Persp cf{};
    if (coeffs_host) for (int j = 0; j < 8; ++j) cf.a[j] = coeffs_host[j];
    hipLaunchKernelGGL(perspective_paste_kernel, dim3(grid1((long)W * H), frames), dim3(256), 0, stream, crop, sw, sh, frame, W, H, cf,
                       (long)sw * sh * 3, (long)W * H * 3, 8, coeffs_dev);
    return ok();
}

int vf_launch_frame_normalise_resize(const unsigned char* frame, int W, int H, float* out, int OW, int OH, int frames,
                                     hipStream_t stream) {
    if (!frame || !out || frames <= 0 || W <= 0 || H <= 0 || OW <= 0 || OH <= 0) return VF_ERR_ARG;
    hipLaunchKernelGGL(frame_normalise_resize_kernel, dim3(grid1((long)OW * OH), frames), dim3(256), 0, stream, frame, W, H, out, OW, OH,
                       (long)W * H * 3, (long)OW * OH * 3);
    return ok();
}
