// upsample.hip -- nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) of the bilinear Up block
// (unet_parts.py:48-50), forward and input gradient, NHWC f32.  HBM-bound streaming kernels: 4 channels per thread.
//
// Source coordinate of output index o: src = o * (in - 1) / (out - 1) evaluated in f32 (scale first, as ATen's
// area_pixel_compute_scale does), i0 = (int)src, i1 = i0 + (i0 < in - 1), l1 = src - i0, l0 = 1 - l1.
// The gradient is the exact adjoint with the SAME weights, gathered per input pixel in a fixed order (no atomics):
// every output row / column whose i0 or i1 is this input index contributes.
#include "common.h"
#include "loader.h"

namespace ustrun {
namespace {

struct Lerp { int i0, i1; float l0, l1; };

__device__ __forceinline__ Lerp lerp_of(int o, int in, float scale) {
    Lerp L;
    const float src = scale * (float)o;
    L.i0 = (int)src;
    L.i1 = L.i0 + (L.i0 < in - 1 ? 1 : 0);
    L.l1 = src - (float)L.i0;
    L.l0 = 1.f - L.l1;
    return L;
}

__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C,
                                                            float* __restrict__ y, float sy, float sx) {
    const int OH = 2 * H, OW = 2 * W, C4 = C / 4;
    const long total = (long)N * OH * OW * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        long p = i / C4;
        const int ox = (int)(p % OW); p /= OW;
        const int oy = (int)(p % OH);
        const int n = (int)(p / OH);
        const Lerp ly = lerp_of(oy, H, sy), lx = lerp_of(ox, W, sx);
        const float* b = x + (long)n * H * W * C + c;
        const f32x4 v00 = *(const f32x4*)(b + ((long)ly.i0 * W + lx.i0) * C), v01 = *(const f32x4*)(b + ((long)ly.i0 * W + lx.i1) * C);
        const f32x4 v10 = *(const f32x4*)(b + ((long)ly.i1 * W + lx.i0) * C), v11 = *(const f32x4*)(b + ((long)ly.i1 * W + lx.i1) * C);
        *(f32x4*)(y + i * 4) = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
    }
}

// weight with which output index o reads input index `at` (0 when it does not)
__device__ __forceinline__ float weight_at(int o, int in, float scale, int at) {
    const Lerp L = lerp_of(o, in, scale);
    return (L.i0 == at ? L.l0 : 0.f) + (L.i1 == at ? L.l1 : 0.f);
}

__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const float* __restrict__ dy, int N, int H, int W, int C,
                                                            float* __restrict__ dx, float sy, float sx) {
    const int OH = 2 * H, OW = 2 * W, C4 = C / 4;
    const long total = (long)N * H * W * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        long p = i / C4;
        const int ix = (int)(p % W); p /= W;
        const int iy = (int)(p % H);
        const int n = (int)(p / H);
        // outputs that can touch this input: src in (i - 1, i + 1)  ->  o in ((i-1)/s, (i+1)/s); two indices of slack
        const int oy_lo = sy > 0.f ? max(0, (int)((float)(iy - 1) / sy) - 2) : 0;
        const int oy_hi = sy > 0.f ? min(OH - 1, (int)((float)(iy + 1) / sy) + 2) : OH - 1;
        const int ox_lo = sx > 0.f ? max(0, (int)((float)(ix - 1) / sx) - 2) : 0;
        const int ox_hi = sx > 0.f ? min(OW - 1, (int)((float)(ix + 1) / sx) + 2) : OW - 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* b = dy + (long)n * OH * OW * C + c;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const float wy = weight_at(oy, H, sy, iy);
            if (wy == 0.f) continue;
            f32x4 row = {0.f, 0.f, 0.f, 0.f};
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const float wx = weight_at(ox, W, sx, ix);
                if (wx != 0.f) row += wx * *(const f32x4*)(b + ((long)oy * OW + ox) * C);
            }
            acc += wy * row;
        }
        *(f32x4*)(dx + i * 4) = acc;
    }
}

// The same two passes inside the fused U-Net plan (round 6: bilinear=True in every storage dtype): the forward reads the RAW output of
// the previous block's second convolution through its BatchNorm constants + ReLU (per pass: gN images share a table, gstride floats
// apart) and interpolates the activated values in f32; both directions store in the plan's element type.
template <int ESZ>
__global__ __launch_bounds__(256) void upsample2x_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, int relu, int gN, long gstride, int N, int H,
                                                            int W, int C, float* __restrict__ y, float sy, float sx) {
    const int OH = 2 * H, OW = 2 * W, C4 = C / 4;
    const long total = (long)N * OH * OW * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        long p = i / C4;
        const int ox = (int)(p % OW); p /= OW;
        const int oy = (int)(p % OH);
        const int n = (int)(p / OH);
        const Lerp ly = lerp_of(oy, H, sy), lx = lerp_of(ox, W, sx);
        const long b = (long)n * H * W * C + c;
        f32x4 v00 = ld4t<ESZ>(x, b + ((long)ly.i0 * W + lx.i0) * C), v01 = ld4t<ESZ>(x, b + ((long)ly.i0 * W + lx.i1) * C);
        f32x4 v10 = ld4t<ESZ>(x, b + ((long)ly.i1 * W + lx.i0) * C), v11 = ld4t<ESZ>(x, b + ((long)ly.i1 * W + lx.i1) * C);
        if (scale) {
            const long go = gN > 0 ? (long)(n / gN) * gstride : 0;
            const f32x4 sc = *(const f32x4*)(scale + go + c), sh = *(const f32x4*)(shift + go + c);
            v00 = v00 * sc + sh; v01 = v01 * sc + sh; v10 = v10 * sc + sh; v11 = v11 * sc + sh;
            if (relu) { v00 = relu4(v00); v01 = relu4(v01); v10 = relu4(v10); v11 = relu4(v11); }
        }
        st4t<ESZ>(y, i * 4, ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11));
    }
}

template <int ESZ>
__global__ __launch_bounds__(256) void upsample2x_bwd_t_kernel(const float* __restrict__ dy, int N, int H, int W, int C,
                                                              float* __restrict__ dx, float sy, float sx) {
    const int OH = 2 * H, OW = 2 * W, C4 = C / 4;
    const long total = (long)N * H * W * C4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        long p = i / C4;
        const int ix = (int)(p % W); p /= W;
        const int iy = (int)(p % H);
        const int n = (int)(p / H);
        const int oy_lo = sy > 0.f ? max(0, (int)((float)(iy - 1) / sy) - 2) : 0;
        const int oy_hi = sy > 0.f ? min(OH - 1, (int)((float)(iy + 1) / sy) + 2) : OH - 1;
        const int ox_lo = sx > 0.f ? max(0, (int)((float)(ix - 1) / sx) - 2) : 0;
        const int ox_hi = sx > 0.f ? min(OW - 1, (int)((float)(ix + 1) / sx) + 2) : OW - 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const long b = (long)n * OH * OW * C + c;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const float wy = weight_at(oy, H, sy, iy);
            if (wy == 0.f) continue;
            f32x4 row = {0.f, 0.f, 0.f, 0.f};
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const float wx = weight_at(ox, W, sx, ix);
                if (wx != 0.f) row += wx * ld4t<ESZ>(dy, b + ((long)oy * OW + ox) * C);
            }
            acc += wy * row;
        }
        st4t<ESZ>(dx, i * 4, acc);
    }
}

int stream_blocks(long work_items) {
    long b = (work_items + 256 * 4 - 1) / (256 * 4);
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

inline float scale_of(int in) { return in > 1 ? (float)(in - 1) / (float)(2 * in - 1) : 0.f; }

}  // namespace
}  // namespace ustrun

using namespace ustrun;

extern "C" int ustrun_upsample2x_fwd(const float* x, int N, int H, int W, int C, float* y, ustrun_stream_t s) {
    USTRUN_CHECK(x && y, "upsample2x_fwd: null pointer");
    USTRUN_CHECK(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample2x_fwd: bad shape (C=%d must be a multiple of 4)", C);
    hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(stream_blocks((long)N * 4 * H * W * (C / 4))), dim3(256), 0, (hipStream_t)s, x,
                       N, H, W, C, y, scale_of(H), scale_of(W));
    USTRUN_LAUNCH_CHECK("upsample2x_fwd");
    return 0;
}

extern "C" int ustrun_upsample2x_bwd(const float* dy, int N, int H, int W, int C, float* dx, ustrun_stream_t s) {
    USTRUN_CHECK(dy && dx, "upsample2x_bwd: null pointer");
    USTRUN_CHECK(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample2x_bwd: bad shape (C=%d must be a multiple of 4)", C);
    hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(stream_blocks((long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)s, dy, N,
                       H, W, C, dx, scale_of(H), scale_of(W));
    USTRUN_LAUNCH_CHECK("upsample2x_bwd");
    return 0;
}

extern "C" int ustrun_upsample2x_act(const ustrun_src_t* src, int N, void* out, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(dtype_ok(dtype), "upsample2x_act: dtype %d not built", dtype);
    USTRUN_CHECK(src && src->ptr && out && N > 0, "upsample2x_act: bad args");
    const int C = src->C, H = src->H, W = src->W;
    USTRUN_CHECK(src->sC == 1 && src->sW == C && src->sH == (int64_t)W * C && src->sN == (int64_t)H * W * C && !src->pool && !src->off_y &&
                 !src->off_x && !src->f32, "upsample2x_act: source must be a plain contiguous NHWC tensor");
    USTRUN_CHECK((src->scale == nullptr) == (src->shift == nullptr) && C % 4 == 0 && H > 0 && W > 0, "upsample2x_act: C=%d / constants", C);
    const int blocks = stream_blocks((long)N * 4 * H * W * (C / 4));
    if (dtype == USTRUN_D16)
        hipLaunchKernelGGL(upsample2x_act_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)src->ptr, src->scale, src->shift,
                           src->relu, src->gN, (long)src->gstride, N, H, W, C, (float*)out, scale_of(H), scale_of(W));
    else
        hipLaunchKernelGGL(upsample2x_act_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)src->ptr, src->scale, src->shift,
                           src->relu, src->gN, (long)src->gstride, N, H, W, C, (float*)out, scale_of(H), scale_of(W));
    USTRUN_LAUNCH_CHECK("upsample2x_act");
    return 0;
}

extern "C" int ustrun_upsample2x_bwd_t(const void* dy, int N, int H, int W, int C, void* dx, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(dtype_ok(dtype), "upsample2x_bwd_t: dtype %d not built", dtype);
    USTRUN_CHECK(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample2x_bwd_t: bad args (C=%d must be a multiple of 4)", C);
    const int blocks = stream_blocks((long)N * H * W * (C / 4));
    if (dtype == USTRUN_D16)
        hipLaunchKernelGGL(upsample2x_bwd_t_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)dy, N, H, W, C, (float*)dx,
                           scale_of(H), scale_of(W));
    else
        hipLaunchKernelGGL(upsample2x_bwd_t_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)dy, N, H, W, C, (float*)dx,
                           scale_of(H), scale_of(W));
    USTRUN_LAUNCH_CHECK("upsample2x_bwd_t");
    return 0;
}
