// resnet_ops.hip -- the operators DeepLabV2-ResNet needs beyond the U-Net's (reference
// networks/deeplabv2.py:10-33, networks/backbone/resnet.py:55-176): a general convolution entry point on the generic
// implicit-GEMM kernels (k x k taps on a regular grid, stride 1 / 2, any dilation: the 7x7 stem, the 1x1 and dilated 3x3
// convolutions of the bottlenecks, the four dilated classifier convolutions), the stem's 3x3 / stride-2 max-pool of the
// activated tensor, the bottleneck's residual join relu(bn3(y) + identity), and the bilinear resize (align_corners) of the
// summed classifier maps to the input extent -- and their backward counterparts (weight gradients on the generic TN GEMMs,
// input gradients as convolutions with flipped weights through the same forward entry point, the join / max-pool / resize /
// shifted-add adjoints).
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

// ---- backward -------------------------------------------------------------------------------------------------------------
// g = (a + b) * (ref > 0): the gradient of the bottleneck's join relu(bn3(y3) + identity) with respect to its pre-ReLU sum, formed
// from the two gradients that reach the block's output (the next block's conv1 input gradient and its identity branch); b = null:
// one contribution; ref = null: no ReLU in between (the stem's max-pool output)
template <int ESZ>
__global__ __launch_bounds__(256) void relu_bwd_add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const float* __restrict__ ref, long total4, float* __restrict__ g) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long)gridDim.x * 256) {
        f32x4 v = ld4t<ESZ>(a, e * 4);
        if (b) v += ld4t<ESZ>(b, e * 4);
        if (ref) {
            const f32x4 r = ld4t<ESZ>(ref, e * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = r[j] > 0.f ? v[j] : 0.f;
        }
        st4t<ESZ>(g, e * 4, v);
    }
}

// MaxPool2d(3, 2, 1) backward as a gather (fixed summation order, no atomics): input pixel (iy, ix) belongs to up to 2 x 2
// windows; each window's arg-max of relu(y * scale + shift) is recomputed in torch's scan order (rows, then columns; a strict >
// keeps the FIRST maximum) and the window's gradient is taken when that arg-max is this pixel.  thread = (input pixel, 4 channels)
template <int ESZ>
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ y,
                                                              const float* __restrict__ scale, const float* __restrict__ shift, int N,
                                                              int H, int W, int C, int Ho, int Wo, float* __restrict__ da) {
    const int C4 = C / 4;
    const long total = (long)N * H * W * C4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c4 = (int)(e % C4);
        long t = e / C4;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const int n = (int)(t / H);
        const f32x4 sc = *(const f32x4*)(scale + 4 * c4), sh = *(const f32x4*)(shift + 4 * c4);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int oy0 = iy >> 1, ox0 = ix >> 1;                       // windows oy with 2*oy-1 <= iy <= 2*oy+1: iy/2 and (iy+1)/2
        const int oy1 = (iy + 1) >> 1, ox1 = (ix + 1) >> 1;
#pragma unroll 1
        for (int oy = oy0; oy <= oy1; ++oy) {
            if (oy >= Ho) continue;
#pragma unroll 1
            for (int ox = ox0; ox <= ox1; ++ox) {
                if (ox >= Wo) continue;
                f32x4 best = {-1.f, -1.f, -1.f, -1.f};                // below every activation (>= 0): the first in-image pixel is taken
                int bi[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 9; ++q) {
                    const int py = 2 * oy - 1 + q / 3, px = 2 * ox - 1 + q % 3;
                    if (py < 0 || py >= H || px < 0 || px >= W) continue;
                    const f32x4 v = relu4(ld4t<ESZ>(y, (((long)n * H + py) * W + px) * C + 4 * c4) * sc + sh);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (v[j] > best[j]) { best[j] = v[j]; bi[j] = q; }
                }
                const int mine = (iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1));
                const f32x4 g = ld4t<ESZ>(dp, (((long)n * Ho + oy) * Wo + ox) * C + 4 * c4);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += bi[j] == mine ? g[j] : 0.f;
            }
        }
        st4t<ESZ>(da, e * 4, acc);
    }
}

// adjoint of sum_resize_kernel for one map: dlow[n][ly][lx][k] = sum over the output pixels whose bilinear footprint holds
// (ly, lx) of their weight * dout; the candidate range is generous and every candidate is tested with the forward's own
// arithmetic, the sum runs in a fixed order.  thread = (low pixel, class)
__global__ __launch_bounds__(256) void sum_resize_bwd_kernel(const float* __restrict__ dout, int N, int h, int w, int K, int H, int W,
                                                            float* __restrict__ dlow) {
    const long total = (long)N * h * w * K;
    const float ry = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, rx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int k = (int)(e % K);
        long t = e / K;
        const int lx = (int)(t % w); t /= w;
        const int ly = (int)(t % h);
        const int n = (int)(t / h);
        int ya = 0, yb = H - 1, xa = 0, xb = W - 1;
        if (ry > 0.f) { ya = max(0, (int)floorf((ly - 1) / ry) - 1); yb = min(H - 1, (int)ceilf((ly + 1) / ry) + 1); }
        if (rx > 0.f) { xa = max(0, (int)floorf((lx - 1) / rx) - 1); xb = min(W - 1, (int)ceilf((lx + 1) / rx) + 1); }
        float acc = 0.f;
        for (int oy = ya; oy <= yb; ++oy) {
            const float fy = ry * oy;
            const int y0 = (int)fy, y1 = y0 + (y0 < h - 1);
            const float fr = fy - y0;
            const float wy = (y0 == ly ? 1.f - fr : 0.f) + (y1 == ly ? fr : 0.f);
            if (y0 != ly && y1 != ly) continue;
            float row = 0.f;
            for (int ox = xa; ox <= xb; ++ox) {
                const float fx = rx * ox;
                const int x0 = (int)fx, x1 = x0 + (x0 < w - 1);
                const float fc = fx - x0;
                if (x0 != lx && x1 != lx) continue;
                const float wx = (x0 == lx ? 1.f - fc : 0.f) + (x1 == lx ? fc : 0.f);
                row += wx * dout[(((long)n * K + k) * H + oy) * W + ox];
            }
            acc += wy * row;
        }
        dlow[e] = acc;
    }
}

// adjoint of aspp_gather_kernel: dz[p][(r*9+tap)*K+k] = dlow[p - rate_r * (tap/3-1, tap%3-1)][k] (zero outside), written in the
// compute dtype with the column count padded to ZCp (zero columns) so that the matrix-core kernels take it
template <int ESZ>
__global__ __launch_bounds__(256) void aspp_scatter_kernel(const float* __restrict__ dlow, int N, int h, int w, int K, int nr, int r0,
                                                          int r1, int r2, int r3, int ZCp, float* __restrict__ dz) {
    const long total = (long)N * h * w * ZCp;
    const int ZC = nr * 9 * K;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int col = (int)(e % ZCp);
        long t = e / ZCp;
        const int x = (int)(t % w); t /= w;
        const int y = (int)(t % h);
        const int n = (int)(t / h);
        float v = 0.f;
        if (col < ZC) {
            const int k = col % K, rt = col / K, tap = rt % 9, r = rt / 9;
            const int rate = r == 0 ? r0 : (r == 1 ? r1 : (r == 2 ? r2 : r3));
            const int yy = y - rate * (tap / 3 - 1), xx = x - rate * (tap % 3 - 1);
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) v = dlow[(((long)n * h + yy) * w + xx) * K + k];
        }
        st1t<ESZ>(dz, e, v);
    }
}

// out[c] (+)= sum over rows of x[row][c] (f32): one block per column, fixed order (thread-strided sums, then a tree in LDS)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long rows, int C, float* __restrict__ out, int accumulate) {
    __shared__ double red[256];
    const int c = blockIdx.x;
    double s = 0.0;
    for (long r = threadIdx.x; r < rows; r += 256) s += (double)x[r * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = (accumulate ? out[c] : 0.f) + (float)red[0];
}

// Row-window patches of a zero-padded NHWC source as GEMM rows: out[m][s * win + j] = xp[n][stride * y + s][stride * x * Cin + j]
// (s < nrows kernel rows, j < win = the kernel row's k * Cin contiguous elements rounded up to 8; columns >= nrows * win are
// zero).  The 7x7 / stride-2 stem becomes a plain 1x1 GEMM over 192 columns on the fast kernels, forward and weight gradient,
// instead of seven 24-wide segments on the generic one.  thread = (output pixel, 8-element group); the source groups are only
// 2-byte aligned (pixel stride 3 elements), the stores are 16-byte aligned.
template <int ESZ>
__global__ __launch_bounds__(256) void rowwin_patches_kernel(const float* __restrict__ xp, long sN, long sH, int Cin, int N, int Ho, int Wo,
                                                            int nrows, int win, int stride, int Kp, float* __restrict__ out) {
    const int G = Kp / 8, gw = win / 8;
    const long total = (long)N * Ho * Wo * G;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int g = (int)(e % G);
        long m = e / G;
        const int x = (int)(m % Wo); m /= Wo;
        const int y = (int)(m % Ho);
        const int n = (int)(m / Ho);
        const int srow = g / gw, j = (g - srow * gw) * 8;
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
        if (srow < nrows) {
            const long src = n * sN + (long)(stride * y + srow) * sH + (long)stride * x * Cin + j;
            lo = ld4t<ESZ>(xp, src); hi = ld4t<ESZ>(xp, src + 4);      // (unaligned vector loads: fine for global memory on gfx950)
        }
        st4t<ESZ>(out, e * 8, lo);
        st4t<ESZ>(out, e * 8 + 4, hi);
    }
}

int stream_blocks(long total) { long b = (total + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace
}  // namespace ustrun

using namespace ustrun;

extern "C" int ustrun_pack_conv(const float* w, int Cout, int Cin, int taps, void* w_fwd, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(dtype_ok(dtype), "pack_conv: dtype %d not built", dtype);
    USTRUN_CHECK(w && w_fwd && Cout > 0 && Cin > 0 && taps >= 1 && taps <= 49, "pack_conv: bad args");
    if (dtype == USTRUN_D16) return pack_bf16(w, Cout, Cin, taps, 0, w_fwd, nullptr, (hipStream_t)s);
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
    if (stat) USTRUN_TRY(stat_rows_within_bound(igemm_stat_rows_used(a, dtype), N, Ho, Wo, Cout, "conv2d_fwd"));
    if (stat_rows) *stat_rows = igemm_stat_rows_used(a, dtype);
    return igemm_launch(a, dtype, (hipStream_t)s);
}

// The input gradient of a 1x1, stride-1 convolution with what follows it in a bottleneck's backward, in ONE launch:
//   g = (dy (*) w_dgrad + add) * (ref > 0)         [N,H,W,Cin]; add / ref may be NULL (one contribution / no ReLU in between)
// -- the residual join behind conv1's input gradient -- and, with y, the BatchNorm-backward sums of the layer whose OUTPUT gradient g
// is, as rows of [2][Cin] = {sum(g mask), sum(g mask y)} per 128 pixels for ustrun_bn_bwd_finalize_stat: mask = y scale + shift > 0
// for a BatchNorm + ReLU layer (conv3's input gradient feeding bn2), all ones with scale = shift = NULL (the previous block's bn3
// behind the join).  *fused = 0 and NO launch when the shape is not covered: the caller then runs ustrun_conv2d_fwd,
// ustrun_relu_bwd_add and ustrun_bn_bwd_reduce.
extern "C" int ustrun_conv1x1_dgrad_join(const void* dy, const void* w_dgrad, int N, int H, int W, int Cout, int Cin, const void* add,
                                         const void* ref, void* g, const void* y, const float* scale, const float* shift, float* stat,
                                         int* stat_rows, int* fused, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(dy && w_dgrad && g && fused && N > 0 && H > 0 && W > 0 && Cout > 0 && Cin > 0, "conv1x1_dgrad_join: bad args");
    USTRUN_CHECK(!y || (stat && stat_rows), "conv1x1_dgrad_join: the sums need their rows");
    USTRUN_CHECK((scale == nullptr) == (shift == nullptr) && (!scale || y), "conv1x1_dgrad_join: scale / shift come together, with y");
    *fused = 0;
    if (stat_rows) *stat_rows = 0;
    if (dtype != USTRUN_D16 || (!add && !ref && !y) || (g_debug_flags2 & 8)) return 0;         // (ustrun_debug_flags2 bit 3: never fused, A/B runs)
    IgemmArgs a = {};
    ustrun_src_t sd = {};
    sd.ptr = dy; sd.C = Cout; sd.H = H; sd.W = W;
    sd.sC = 1; sd.sW = Cout; sd.sH = (int64_t)W * Cout; sd.sN = (int64_t)H * W * Cout;
    a.nsrc = 1; a.src[0] = make_src(sd, dtype); a.Cin = Cout;
    a.W = (const float*)w_dgrad; a.Cout = Cin;
    a.N = N; a.Hb = H; a.Wb = W; a.M = N * H * W;
    a.s_in = 1; a.nseg = 1; a.segw = 1; a.d0 = 0; a.dstep = 1;
    a.nz = 1; a.s_out = 1;
    a.out0 = (float*)g; a.C0 = Cin; a.Ho = H; a.Wo = W; a.out_esz = 2;
    if (!conv1x1_join_supported(a)) return 0;
    a.join_add = add; a.join_ref = ref;
    if (y) {
        a.bny = y; a.bnsc = scale; a.bnsh = shift; a.stat = stat;
        const int used = cdiv(a.M, 128);
        USTRUN_TRY(stat_rows_within_bound(used, N, H, W, Cin, "conv1x1_dgrad_join"));
        *stat_rows = used;
    }
    USTRUN_TRY(igemm_launch(a, dtype, (hipStream_t)s));
    *fused = 1;
    return 0;
}

// The input gradient of a (dilated) 3x3, stride-1 convolution as ustrun_conv2d_fwd forms it -- a 3x3 convolution of dy with the pack
// of w.flip(2, 3).transpose(0, 1) -- that also forms the BatchNorm-backward sums of the BatchNorm + ReLU layer whose output gradient
// da is (conv2's input gradient feeding bn1): rows of [2][Cin] = {sum(da mask), sum(da mask y)}, mask = y scale + shift > 0, as
// ustrun_conv3x3_dgrad_bnsum.  *stat_rows = 0 and NO launch when the fused epilogue does not cover the shape.
extern "C" int ustrun_conv2d_dgrad_bnsum(const void* dy, const void* w_flipped, int N, int H, int W, int Cout, int Cin, int dilation, void* da,
                                         const void* y, const float* scale, const float* shift, float* stat, int* stat_rows, int dtype,
                                         ustrun_stream_t s) {
    USTRUN_CHECK(dy && w_flipped && da && y && scale && shift && stat && stat_rows && N > 0 && H > 0 && W > 0 && Cout > 0 && Cin > 0 && dilation >= 1,
                 "conv2d_dgrad_bnsum: bad args");
    *stat_rows = 0;
    if (dtype != USTRUN_D16 || (g_debug_flags2 & 8)) return 0;
    IgemmArgs a = {};
    ustrun_src_t sd = {};
    sd.ptr = dy; sd.C = Cout; sd.H = H; sd.W = W;
    sd.sC = 1; sd.sW = Cout; sd.sH = (int64_t)W * Cout; sd.sN = (int64_t)H * W * Cout;
    a.nsrc = 1; a.src[0] = make_src(sd, dtype); a.Cin = Cout;
    a.W = (const float*)w_flipped; a.Cout = Cin;
    a.N = N; a.Hb = H; a.Wb = W; a.M = N * H * W;
    a.s_in = 1; a.nseg = 9; a.segw = 3; a.d0 = -dilation; a.dstep = dilation;
    a.nz = 1; a.s_out = 1;
    a.out0 = (float*)da; a.C0 = Cin; a.Ho = H; a.Wo = W; a.out_esz = 2;
    if (!halo_supported(a) || !halo_bnsum_supported(a)) return 0;
    a.bny = y; a.bnsc = scale; a.bnsh = shift; a.stat = stat;
    const int used = halo_stat_rows_used(a);
    USTRUN_TRY(stat_rows_within_bound(used, N, H, W, Cin, "conv2d_dgrad_bnsum"));
    USTRUN_TRY(igemm_launch(a, dtype, (hipStream_t)s));
    *stat_rows = used;
    return 0;
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
    if (stat) USTRUN_TRY(stat_rows_within_bound(igemm_stat_rows_used(a, dtype), N, Ho, Wo, Cout, "conv_rowwin_fwd"));
    if (stat_rows) *stat_rows = igemm_stat_rows_used(a, dtype);
    return igemm_launch(a, dtype, (hipStream_t)s);
}

extern "C" int ustrun_maxpool3x3s2(const void* y, const float* scale, const float* shift, int N, int H, int W, int C, void* out,
                                   int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(y && scale && shift && out && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && dtype_ok(dtype), "maxpool3x3s2: bad args");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;             // floor((H + 2 - 3) / 2) + 1
    const long total = (long)N * Ho * Wo * (C / 4);
    if (dtype == USTRUN_D16)
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
    if (dtype == USTRUN_D16)
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

// Space-to-batch for the weight gradient of a dilated 3x3 convolution (round 6).  With dilation r and padding r, output pixel
// (y, x) only meets input pixels of its own residue class (y mod r, x mod r): on the r x r sub-grids the convolution is an ORDINARY
// 3x3 (padding 1), and its weight gradient is the sum of the sub-grids' -- which the all-taps kernel (wgrad_halo_bf16.hip: the dY
// tile read once for nine taps) forms at 2-3 x the rate of nine one-tap GEMMs that each stream both operands again.  This pass
// writes out[(n r + a) r + b][i][j][c] = act(src[n][i r + a][j r + b][c]) for every sub-grid (a, b), all padded to ceil(H / r) x
// ceil(W / r) with ZEROS (the positions past a shorter sub-grid's edge, and what the 3x3's own padding reads there): zeros in both
// operands add nothing to the sums.  act = relu(x scale + shift) with constants (it then replaces ustrun_act16), else a copy.
// thread = (output pixel, 8 channels): 16-byte loads and stores.
__global__ __launch_bounds__(256) void space_to_batch_kernel(const elt_t* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int relu, int N, int H, int W, int C, int r,
                                                             int Hs, int Ws, elt_t* __restrict__ out) {
    typedef __attribute__((ext_vector_type(8))) elt_t v8;
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    const int C8 = C / 8;
    const long total = (long)N * r * r * Hs * Ws * C8;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int c8 = (int)(e % C8);
        long t = e / C8;
        const int j = (int)(t % Ws); t /= Ws;
        const int i = (int)(t % Hs); t /= Hs;
        const int b = (int)(t % r); t /= r;
        const int a = (int)(t % r);
        const int n = (int)(t / r);
        const int yy = i * r + a, xx = j * r + b;
        u4 o = {0u, 0u, 0u, 0u};
        if (yy < H && xx < W) {
            v8 v = *(const v8*)(x + (((long)n * H + yy) * W + xx) * C + c8 * 8);
            if (scale) {
                const f32x4 s0 = *(const f32x4*)(scale + c8 * 8), s1 = *(const f32x4*)(scale + c8 * 8 + 4);
                const f32x4 b0 = *(const f32x4*)(shift + c8 * 8), b1 = *(const f32x4*)(shift + c8 * 8 + 4);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float f = (float)v[q] * (q < 4 ? s0[q & 3] : s1[q & 3]) + (q < 4 ? b0[q & 3] : b1[q & 3]);
                    f = relu ? fmaxf(f, 0.f) : f;
                    v[q] = (elt_t)f;
                }
            }
            o = __builtin_bit_cast(u4, v);
        }
        *(u4*)(out + e * 8) = o;
    }
}

// ---- backward entry points --------------------------------------------------------------------------------------------------
#define USTRUN_BY_DTYPE(KERNEL, BLOCKS, ...)                                                                                    \
    do {                                                                                                                        \
        if (dtype == USTRUN_D16) hipLaunchKernelGGL(KERNEL<2>, dim3(BLOCKS), dim3(256), 0, (hipStream_t)s, __VA_ARGS__);       \
        else hipLaunchKernelGGL(KERNEL<4>, dim3(BLOCKS), dim3(256), 0, (hipStream_t)s, __VA_ARGS__);                            \
    } while (0)

extern "C" int ustrun_relu_bwd_add(const void* a, const void* b, const void* ref, int64_t n, void* g, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(a && g && n > 0 && n % 4 == 0 && dtype_ok(dtype), "relu_bwd_add: bad args");
    USTRUN_BY_DTYPE(relu_bwd_add_kernel, stream_blocks(n / 4), (const float*)a, (const float*)b, (const float*)ref, (long)(n / 4), (float*)g);
    USTRUN_LAUNCH_CHECK("relu_bwd_add");
    return 0;
}

extern "C" int ustrun_space_to_batch(const ustrun_src_t* src, int N, int r, void* out, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(dtype == USTRUN_D16, "space_to_batch: 16-bit storage only (dtype %d)", dtype);
    USTRUN_CHECK(src && src->ptr && out && N > 0 && r >= 2 && r <= 8, "space_to_batch: bad args");
    const int C = src->C, H = src->H, W = src->W;
    USTRUN_CHECK(src->sC == 1 && src->sW == C && src->sH == (int64_t)W * C && src->sN == (int64_t)H * W * C && !src->pool && !src->off_y &&
                 !src->off_x && !src->f32 && src->gN == 0, "space_to_batch: source must be a plain contiguous NHWC activation");
    USTRUN_CHECK((src->scale == nullptr) == (src->shift == nullptr) && (!src->relu || src->scale) && C % 8 == 0, "space_to_batch: C=%d / constants", C);
    const int Hs = (H + r - 1) / r, Ws = (W + r - 1) / r;
    hipLaunchKernelGGL(space_to_batch_kernel, dim3(stream_blocks((long)N * r * r * Hs * Ws * (C / 8))), dim3(256), 0, (hipStream_t)s,
                       (const elt_t*)src->ptr, src->scale, src->shift, src->relu, N, H, W, C, r, Hs, Ws, (elt_t*)out);
    USTRUN_LAUNCH_CHECK("space_to_batch");
    return 0;
}

extern "C" int ustrun_maxpool3x3s2_bwd(const void* dp, const void* y, const float* scale, const float* shift, int N, int H, int W, int C,
                                       void* da, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(dp && y && scale && shift && da && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && dtype_ok(dtype), "maxpool3x3s2_bwd: bad args");
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    USTRUN_BY_DTYPE(maxpool3x3s2_bwd_kernel, stream_blocks((long)N * H * W * (C / 4)), (const float*)dp, (const float*)y, scale, shift, N, H,
                    W, C, Ho, Wo, (float*)da);
    USTRUN_LAUNCH_CHECK("maxpool3x3s2_bwd");
    return 0;
}

extern "C" int ustrun_sum_resize_bilinear_bwd(const float* dout, int N, int h, int w, int K, int H, int W, float* dlow, ustrun_stream_t s) {
    USTRUN_CHECK(dout && dlow && N > 0 && h > 0 && w > 0 && K > 0 && H > 0 && W > 0, "sum_resize_bilinear_bwd: bad args");
    hipLaunchKernelGGL(sum_resize_bwd_kernel, dim3(stream_blocks((long)N * h * w * K)), dim3(256), 0, (hipStream_t)s, dout, N, h, w, K, H, W,
                       dlow);
    USTRUN_LAUNCH_CHECK("sum_resize_bilinear_bwd");
    return 0;
}

extern "C" int ustrun_aspp_scatter(const float* dlow, int N, int h, int w, int K, int nrates, const int* rates, int zc_padded, void* dz,
                                   int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(dlow && rates && dz && N > 0 && h > 0 && w > 0 && K > 0 && nrates >= 1 && nrates <= 4 && dtype_ok(dtype) &&
                 zc_padded >= nrates * 9 * K, "aspp_scatter: bad args");
    int r[4] = {1, 1, 1, 1};
    for (int i = 0; i < nrates; ++i) r[i] = rates[i];
    USTRUN_BY_DTYPE(aspp_scatter_kernel, stream_blocks((long)N * h * w * zc_padded), dlow, N, h, w, K, nrates, r[0], r[1], r[2], r[3],
                    zc_padded, (float*)dz);
    USTRUN_LAUNCH_CHECK("aspp_scatter");
    return 0;
}

extern "C" int ustrun_colsum(const float* x, int64_t rows, int C, float* out, int accumulate, ustrun_stream_t s) {
    USTRUN_CHECK(x && out && rows > 0 && C > 0 && C <= 65535, "colsum: bad args");
    hipLaunchKernelGGL(colsum_kernel, dim3(C), dim3(256), 0, (hipStream_t)s, x, (long)rows, C, out, accumulate);
    USTRUN_LAUNCH_CHECK("colsum");
    return 0;
}

// weight gradient of ustrun_conv2d_fwd's convolution: dw[Cout][Cin][k][k] (torch layout, f32) = sum over output pixels of
// loader(srcs)(stride * p + dilation * (tap - k/2)) x dy(p)
extern "C" int ustrun_conv2d_wgrad(const ustrun_src_t* srcs, int nsrc, const void* dy, int N, int Ho, int Wo, int Cout, int k, int stride,
                                   int dilation, float* dw, int accumulate, float* partials, int64_t partials_bytes, int dtype,
                                   ustrun_stream_t s) {
    USTRUN_CHECK(srcs && (nsrc == 1 || nsrc == 2) && dy && dw && partials && N > 0 && Ho > 0 && Wo > 0 && Cout > 0, "conv2d_wgrad: bad args");
    USTRUN_CHECK((k == 1 || k == 3) && (stride == 1 || stride == 2) && dilation >= 1, "conv2d_wgrad: k=%d stride=%d dilation=%d", k, stride,
                 dilation);
    WgradArgs a = {};
    a.nsrc = nsrc; a.Cin = 0;
    for (int i = 0; i < nsrc; ++i) {
        USTRUN_CHECK(srcs[i].ptr && srcs[i].C > 0 && !srcs[i].pool && srcs[i].gN == 0, "conv2d_wgrad: bad source %d", i);
        a.src[i] = make_src(srcs[i], dtype); a.Cin += srcs[i].C;
    }
    a.dy = (const float*)dy; a.Cout = Cout; a.dy_esz = act_esz(dtype);
    a.N = N; a.Hb = Ho; a.Wb = Wo; a.M = (long)N * Ho * Wo;
    a.nseg = k * k; a.segw = k; a.d0 = -dilation * (k / 2); a.astep = dilation; a.dy_s = 1; a.dyH = Ho; a.dyW = Wo;
    a.ashift = stride == 2 ? 1 : 0;
    a.partials = partials;
    int slabs;
    if (dtype == USTRUN_D16 && wgrad_halo_supported(a)) {
        int per;
        wgrad_halo_plan(a, &slabs, &per);
        USTRUN_CHECK(partials_bytes >= (int64_t)slabs * 9 * a.Cin * Cout * 4, "conv2d_wgrad: partials too small");
        prof_begin(1, 2.0 * a.M * 9 * a.Cin * Cout, 2.0 * ((double)a.M * a.Cin + (double)a.M * Cout) + 36.0 * a.Cin * Cout, (hipStream_t)s);
        const int rc = wgrad_halo_launch_bf16(a, slabs, per, (hipStream_t)s);
        prof_end((hipStream_t)s);
        USTRUN_TRY(rc);
        return reduce_partials(partials, slabs, 9, a.Cin, Cout, dw, 2, accumulate, (hipStream_t)s);
    }
    if (dtype == USTRUN_D16 && wgrad_tap_supported(a)) {
        wgrad_tap_plan(a, &a.ksplit, &a.kchunk);
        USTRUN_CHECK(partials_bytes >= (int64_t)a.ksplit * a.nseg * a.Cin * Cout * 4, "conv2d_wgrad: partials too small");
        double in_elems = (double)N * srcs[0].H * srcs[0].W * srcs[0].C;
        prof_begin(1, 2.0 * a.M * a.nseg * a.Cin * Cout, 2.0 * (in_elems + (double)a.M * Cout) + 4.0 * a.nseg * a.Cin * Cout, (hipStream_t)s);
        const int rc = wgrad_tap_launch_bf16(a, (hipStream_t)s);
        prof_end((hipStream_t)s);
        USTRUN_TRY(rc);
        // 1x1: the slabs are [Cout][Cin] = the torch layout already (plain streaming sum); k x k: [tap][Cin][Cout] -> transposed
        return reduce_partials(partials, a.ksplit, a.nseg, a.Cin, Cout, dw, a.nseg == 1 ? 2 : 0, accumulate, (hipStream_t)s);
    }
    wgrad_plan(a.nseg, a.Cin, Cout, a.M, &a.ksplit, &a.kchunk, &slabs);
    USTRUN_CHECK(partials_bytes >= (int64_t)slabs * a.nseg * a.Cin * Cout * 4, "conv2d_wgrad: partials too small");
    USTRUN_TRY(wgrad_launch(a, dtype, (hipStream_t)s));
    return reduce_partials(partials, slabs, a.nseg, a.Cin, Cout, dw, 0, accumulate, (hipStream_t)s);
}

// weight gradient of ustrun_conv_rowwin_fwd: dw[Cout][src->C][nrows] (the layout its weights were packed from)
extern "C" int ustrun_conv_rowwin_wgrad(const ustrun_src_t* src, const void* dy, int N, int Ho, int Wo, int Cout, int nrows, int stride,
                                        float* dw, int accumulate, float* partials, int64_t partials_bytes, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(src && src->ptr && dy && dw && partials && N > 0 && Ho > 0 && Wo > 0 && Cout > 0 && nrows >= 1 && nrows <= 9 &&
                 (stride == 1 || stride == 2), "conv_rowwin_wgrad: bad args");
    USTRUN_CHECK(src->sC == 1 && !src->pool && src->gN == 0 && (stride * (Ho - 1) + nrows) <= src->H && stride * (Wo - 1) < src->W,
                 "conv_rowwin_wgrad: the padded source [%d x %d] does not cover the windows", src->H, src->W);
    WgradArgs a = {};
    a.nsrc = 1; a.src[0] = make_src(*src, dtype); a.Cin = src->C;
    a.dy = (const float*)dy; a.Cout = Cout; a.dy_esz = act_esz(dtype);
    a.N = N; a.Hb = Ho; a.Wb = Wo; a.M = (long)N * Ho * Wo;
    a.nseg = nrows; a.segw = 1; a.d0 = 0; a.astep = 1; a.dy_s = 1; a.dyH = Ho; a.dyW = Wo;
    a.ashift = stride == 2 ? 1 : 0;
    a.partials = partials;
    int slabs;
    wgrad_plan(a.nseg, a.Cin, Cout, a.M, &a.ksplit, &a.kchunk, &slabs);
    USTRUN_CHECK(partials_bytes >= (int64_t)slabs * a.nseg * a.Cin * Cout * 4, "conv_rowwin_wgrad: partials too small");
    USTRUN_TRY(wgrad_launch(a, dtype, (hipStream_t)s));
    return reduce_partials(partials, slabs, a.nseg, a.Cin, Cout, dw, 0, accumulate, (hipStream_t)s);
}

// patches of a padded NHWC source as GEMM rows [N*Ho*Wo][k_padded] (see rowwin_patches_kernel): src as for ustrun_conv_rowwin_fwd
extern "C" int ustrun_rowwin_patches(const ustrun_src_t* src, int N, int Ho, int Wo, int nrows, int stride, int k_padded, void* out, int dtype,
                                     ustrun_stream_t s) {
    USTRUN_CHECK(src && src->ptr && out && N > 0 && Ho > 0 && Wo > 0 && nrows >= 1 && nrows <= 9 && (stride == 1 || stride == 2) &&
                 dtype_ok(dtype), "rowwin_patches: bad args");
    USTRUN_CHECK(src->sC == 1 && src->C % 8 == 0 && k_padded % 8 == 0 && k_padded >= nrows * src->C && !src->pool && !src->scale && !src->f32,
                 "rowwin_patches: window %d / padded K %d", src->C, k_padded);
    USTRUN_CHECK((stride * (Ho - 1) + nrows) <= src->H && stride * (Wo - 1) < src->W,
                 "rowwin_patches: the padded source [%d x %d] does not cover the windows", src->H, src->W);
    const long total = (long)N * Ho * Wo * (k_padded / 8);
    USTRUN_BY_DTYPE(rowwin_patches_kernel, stream_blocks(total), (const float*)src->ptr, (long)src->sN, (long)src->sH, (int)src->sW, N, Ho, Wo,
                    nrows, src->C, stride, k_padded, (float*)out);
    USTRUN_LAUNCH_CHECK("rowwin_patches");
    return 0;
}
