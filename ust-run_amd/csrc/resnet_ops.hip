// resnet_ops.hip -- the operators DeepLabV2-ResNet needs beyond the U-Net's (reference
// networks/deeplabv2.py:10-33, networks/backbone/resnet.py:55-176): a general convolution entry point on the generic
// implicit-GEMM kernels (k x k taps on a regular grid, stride 1 / 2, any dilation: the 7x7 stem, the 1x1 and dilated 3x3
// convolutions of the bottlenecks, the four dilated classifier convolutions), the stem's 3x3 / stride-2 max-pool of the
// activated tensor, the bottleneck's residual join relu(bn3(y) + identity), and the bilinear resize (align_corners) of the
// summed classifier maps to the input extent.  Forward only this round.
#include "common.h"
#include "loader.h"

namespace ustrun {
namespace {

// torch conv weight [Cout][Cin][taps] -> [taps][Cin][Cout] (the f32 kernels' layout)
__global__ void pack_conv_f32_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, float* __restrict__ wf) {
    const long total = (long)Cout * Cin * taps;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int co = (int)(e % Cout);
        const long t = e / Cout;
        const int ci = (int)(t % Cin), tap = (int)(t / Cin);
        wf[e] = w[((long)co * Cin + ci) * taps + tap];
    }
}

// MaxPool2d(3, stride 2, padding 1) of relu(y * scale + shift): thread = (output pixel, 4 channels)
template <int ESZ>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int N, int H, int W, int C, int Ho, int Wo,
                                                          float* __restrict__ out) {
    const int C4 = C / 4;
    const long total = (long)N * Ho * Wo * C4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c4 = (int)(e % C4);
        long t = e / C4;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const f32x4 sc = *(const f32x4*)(scale + 4 * c4), sh = *(const f32x4*)(shift + 4 * c4);
        f32x4 m = {0.f, 0.f, 0.f, 0.f};            // the activation is >= 0 and every window holds an in-image pixel
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int iy = 2 * oy + dy, ix = 2 * ox + dx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                    m = max4(m, relu4(ld4t<ESZ>(y, (((long)n * H + iy) * W + ix) * C + 4 * c4) * sc + sh));
            }
        st4t<ESZ>(out, (((long)n * Ho + oy) * Wo + ox) * C + 4 * c4, m);
    }
}

// out = relu(y * scale + shift + (idn * iscale + ishift)); iscale == null: the identity is an activation already
template <int ESZ>
__global__ __launch_bounds__(256) void bn_add_relu_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ idn,
                                                         const float* __restrict__ iscale, const float* __restrict__ ishift, long npix,
                                                         int C, float* __restrict__ out) {
    const int C4 = C / 4;
    const long total = npix * C4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c4 = (int)(e % C4);
        f32x4 v = ld4t<ESZ>(y, e * 4) * *(const f32x4*)(scale + 4 * c4) + *(const f32x4*)(shift + 4 * c4);
        f32x4 r = ld4t<ESZ>(idn, e * 4);
        if (iscale) r = r * *(const f32x4*)(iscale + 4 * c4) + *(const f32x4*)(ishift + 4 * c4);
        st4t<ESZ>(out, e * 4, relu4(v + r));
    }
}

// sum of up to four NHWC f32 maps [N, h, w, K], resized to [N, K, H, W] (NCHW f32) bilinearly with align_corners=True
__global__ __launch_bounds__(256) void sum_resize_kernel(const float* __restrict__ a0, const float* __restrict__ a1,
                                                        const float* __restrict__ a2, const float* __restrict__ a3, int N, int h, int w,
                                                        int K, int H, int W, float* __restrict__ out) {
    const long total = (long)N * K * H * W;
    const float ry = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, rx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int x = (int)(e % W);
        long t = e / W;
        const int yy = (int)(t % H); t /= H;
        const int k = (int)(t % K);
        const int n = (int)(t / K);
        const float fy = ry * yy, fx = rx * x;                  // torch: src = scale * dst (area_pixel_compute_source_index)
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < h - 1), x1 = x0 + (x0 < w - 1);
        const float ly = fy - y0, lx = fx - x0;
        auto at = [&](int py, int px) {
            const long i = (((long)n * h + py) * w + px) * K + k;
            float v = a0[i];
            if (a1) v += a1[i];
            if (a2) v += a2[i];
            if (a3) v += a3[i];
            return v;
        };
        const float top = at(y0, x0) * (1.f - lx) + at(y0, x1) * lx, bot = at(y1, x0) * (1.f - lx) + at(y1, x1) * lx;
        out[e] = top * (1.f - ly) + bot * ly;
    }
}

// The classifier's dilated 3x3 branches as ONE 1x1 GEMM + shifted adds (a convolution is linear in its taps):
//   z[p][(r * 9 + tap) * K + k] = sum_c x[p][c] * w_r[k][c][tap]              (the GEMM, on the matrix cores)
//   out[p][k] = sum_r bias_r[k] + sum_{r, tap} z[p + rate_r * (tap / 3 - 1, tap % 3 - 1)][(r * 9 + tap) * K + k]
// this kernel is the second line: thread = (pixel, class), taps outside the image contribute nothing (zero padding).
__global__ __launch_bounds__(256) void aspp_gather_kernel(const float* __restrict__ z, int N, int h, int w, int K, int nr, int r0, int r1,
                                                         int r2, int r3, const float* __restrict__ bias_sum, float* __restrict__ out) {
    const long total = (long)N * h * w * K;
    const int ZC = nr * 9 * K;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int k = (int)(e % K);
        long t = e / K;
        const int x = (int)(t % w); t /= w;
        const int y = (int)(t % h);
        const int n = (int)(t / h);
        float acc = bias_sum[k];
#pragma unroll 1
        for (int r = 0; r < nr; ++r) {
            const int rate = r == 0 ? r0 : (r == 1 ? r1 : (r == 2 ? r2 : r3));
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = y + rate * (tap / 3 - 1), xx = x + rate * (tap % 3 - 1);
                if (yy >= 0 && yy < h && xx >= 0 && xx < w) acc += z[(((long)n * h + yy) * w + xx) * ZC + (r * 9 + tap) * K + k];
            }
        }
        out[e] = acc;
    }
}

int stream_blocks(long total) { long b = (total + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace
}  // namespace ustrun

using namespace ustrun;

extern "C" int ustrun_pack_conv(const float* w, int Cout, int Cin, int taps, void* w_fwd, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(dtype_ok(dtype), "pack_conv: dtype %d not built", dtype);
    USTRUN_CHECK(w && w_fwd && Cout > 0 && Cin > 0 && taps >= 1 && taps <= 49, "pack_conv: bad args");
    if (dtype == USTRUN_BF16) return pack_bf16(w, Cout, Cin, taps, 0, w_fwd, nullptr, (hipStream_t)s);
    hipLaunchKernelGGL(pack_conv_f32_kernel, dim3(stream_blocks((long)Cout * Cin * taps)), dim3(256), 0, (hipStream_t)s, w, Cout, Cin,
                       taps, (float*)w_fwd);
    USTRUN_LAUNCH_CHECK("pack_conv");
    return 0;
}

extern "C" int64_t ustrun_pack_conv_elems(int Cout, int Cin, int taps) {
    return (int64_t)taps * ((Cin + 7) / 8 * 8) * ((Cout + 7) / 8 * 8);
}

extern "C" int ustrun_conv2d_fwd(const ustrun_src_t* srcs, int nsrc, const void* w_fwd, const float* bias, int N, int Ho, int Wo,
                                 int Cout, int k, int stride, int dilation, void* y, int y_f32, float* stat, int* stat_rows, int dtype,
                                 ustrun_stream_t s) {
    USTRUN_CHECK(srcs && (nsrc == 1 || nsrc == 2) && w_fwd && y && N > 0 && Ho > 0 && Wo > 0 && Cout > 0, "conv2d_fwd: bad args");
    USTRUN_CHECK((k == 1 || k == 3 || k == 5 || k == 7) && (stride == 1 || stride == 2) && dilation >= 1, "conv2d_fwd: k=%d stride=%d dilation=%d", k,
                 stride, dilation);
    IgemmArgs a = {};
    a.nsrc = nsrc; a.Cin = 0;
    for (int i = 0; i < nsrc; ++i) {
        USTRUN_CHECK(srcs[i].ptr && srcs[i].C > 0 && !srcs[i].pool && srcs[i].gN == 0, "conv2d_fwd: bad source %d", i);
        a.src[i] = make_src(srcs[i], dtype); a.Cin += srcs[i].C;
    }
    a.W = (const float*)w_fwd; a.Cout = Cout;
    a.N = N; a.Hb = Ho; a.Wb = Wo; a.M = N * Ho * Wo;
    a.s_in = stride; a.nseg = k * k; a.segw = k; a.d0 = -dilation * (k / 2); a.dstep = dilation;     // padding = dilation * (k / 2)
    a.nz = 1; a.s_out = 1;
    a.out0 = (float*)y; a.out1 = nullptr; a.C0 = Cout; a.Ho = Ho; a.Wo = Wo;
    a.bias = bias; a.stat = stat; a.out_esz = y_f32 ? 4 : act_esz(dtype);
    if (stat_rows) *stat_rows = igemm_stat_rows_used(a, dtype);
    return igemm_launch(a, dtype, (hipStream_t)s);
}

// A k x k convolution over few input channels as `nrows` row segments: with an NHWC source of C channels, the k horizontally
// adjacent pixels of one kernel row are k * C CONTIGUOUS elements, so the caller describes the source with "channels" = that
// window (rounded up to a multiple of 8: the extra elements meet zero weights), pixel stride = C elements, and a border it has
// padded itself; segment s reads the window at row stride * y + s, column stride * x.  The 7x7 / stride-2 stem (resnet.py:124):
// 7 K-chunks instead of 49 (5.4 -> ~0.3 ms at N = 8, 512^2).
extern "C" int ustrun_conv_rowwin_fwd(const ustrun_src_t* src, const void* w_fwd, int N, int Ho, int Wo, int Cout, int nrows, int stride,
                                      void* y, float* stat, int* stat_rows, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(src && src->ptr && w_fwd && y && N > 0 && Ho > 0 && Wo > 0 && Cout > 0 && nrows >= 1 && nrows <= 9 && stride >= 1,
                 "conv_rowwin_fwd: bad args");
    USTRUN_CHECK(src->sC == 1 && !src->pool && src->gN == 0 && (stride * (Ho - 1) + nrows) <= src->H && stride * (Wo - 1) < src->W,
                 "conv_rowwin_fwd: the padded source [%d x %d] does not cover the windows", src->H, src->W);
    IgemmArgs a = {};
    a.nsrc = 1; a.src[0] = make_src(*src, dtype); a.Cin = src->C;
    a.W = (const float*)w_fwd; a.Cout = Cout;
    a.N = N; a.Hb = Ho; a.Wb = Wo; a.M = N * Ho * Wo;
    a.s_in = stride; a.nseg = nrows; a.segw = 1; a.d0 = 0; a.dstep = 1;      // segment s: (dy, dx) = (s, 0)
    a.nz = 1; a.s_out = 1;
    a.out0 = (float*)y; a.out1 = nullptr; a.C0 = Cout; a.Ho = Ho; a.Wo = Wo;
    a.bias = nullptr; a.stat = stat; a.out_esz = act_esz(dtype);
    if (stat_rows) *stat_rows = igemm_stat_rows_used(a, dtype);
    return igemm_launch(a, dtype, (hipStream_t)s);
}

extern "C" int ustrun_maxpool3x3s2(const void* y, const float* scale, const float* shift, int N, int H, int W, int C, void* out,
                                   int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(y && scale && shift && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && dtype_ok(dtype), "maxpool3x3s2: bad args");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;             // floor((H + 2 - 3) / 2) + 1
    const long total = (long)N * Ho * Wo * (C / 4);
    if (dtype == USTRUN_BF16)
        hipLaunchKernelGGL(maxpool3x3s2_kernel<2>, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)s, (const float*)y, scale, shift, N,
                           H, W, C, Ho, Wo, (float*)out);
    else
        hipLaunchKernelGGL(maxpool3x3s2_kernel<4>, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)s, (const float*)y, scale, shift, N,
                           H, W, C, Ho, Wo, (float*)out);
    USTRUN_LAUNCH_CHECK("maxpool3x3s2");
    return 0;
}

extern "C" int ustrun_bn_add_relu(const void* y, const float* scale, const float* shift, const void* idn, const float* iscale,
                                  const float* ishift, int64_t npix, int C, void* out, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(y && scale && shift && idn && out && npix > 0 && C > 0 && C % 4 == 0 && dtype_ok(dtype), "bn_add_relu: bad args");
    USTRUN_CHECK((iscale == nullptr) == (ishift == nullptr), "bn_add_relu: iscale/ishift must come together");
    const long total = npix * (C / 4);
    if (dtype == USTRUN_BF16)
        hipLaunchKernelGGL(bn_add_relu_kernel<2>, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)s, (const float*)y, scale, shift,
                           (const float*)idn, iscale, ishift, (long)npix, C, (float*)out);
    else
        hipLaunchKernelGGL(bn_add_relu_kernel<4>, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)s, (const float*)y, scale, shift,
                           (const float*)idn, iscale, ishift, (long)npix, C, (float*)out);
    USTRUN_LAUNCH_CHECK("bn_add_relu");
    return 0;
}

extern "C" int ustrun_sum_resize_bilinear(const float* const* maps, int nmaps, int N, int h, int w, int K, int H, int W, float* out,
                                          ustrun_stream_t s) {
    USTRUN_CHECK(maps && nmaps >= 1 && nmaps <= 4 && maps[0] && out && N > 0 && h > 0 && w > 0 && K > 0 && H > 0 && W > 0,
                 "sum_resize_bilinear: bad args");
    const float* p[4] = {maps[0], nmaps > 1 ? maps[1] : nullptr, nmaps > 2 ? maps[2] : nullptr, nmaps > 3 ? maps[3] : nullptr};
    hipLaunchKernelGGL(sum_resize_kernel, dim3(stream_blocks((long)N * K * H * W)), dim3(256), 0, (hipStream_t)s, p[0], p[1], p[2], p[3], N,
                       h, w, K, H, W, out);
    USTRUN_LAUNCH_CHECK("sum_resize_bilinear");
    return 0;
}

extern "C" int ustrun_aspp_gather(const float* z, int N, int h, int w, int K, int nrates, const int* rates, const float* bias_sum,
                                  float* out, ustrun_stream_t s) {
    USTRUN_CHECK(z && rates && bias_sum && out && N > 0 && h > 0 && w > 0 && K > 0 && nrates >= 1 && nrates <= 4, "aspp_gather: bad args");
    int r[4] = {1, 1, 1, 1};
    for (int i = 0; i < nrates; ++i) r[i] = rates[i];
    hipLaunchKernelGGL(aspp_gather_kernel, dim3(stream_blocks((long)N * h * w * K)), dim3(256), 0, (hipStream_t)s, z, N, h, w, K, nrates,
                       r[0], r[1], r[2], r[3], bias_sum, out);
    USTRUN_LAUNCH_CHECK("aspp_gather");
    return 0;
}
