// fftmix.hip -- low-frequency amplitude mix (reference train.py:158-207,628-636) on the device.
//
// The reference swaps a (2b+1)^2 window of the fft-shifted amplitude spectrum of `src` towards that
// of `trg` (blend ratio r), keeps src's phase, and inverse-transforms.  Only those bins change, so by
// linearity   out = src + Re IDFT( r (|F_trg| - |F_src|) e^{j arg F_src} )   restricted to the window:
// a (2b+1)^2-bin forward DFT of both images (kernel 1, direct summation with twiddle tables in LDS)
// and a (2b+1)^2-term correction per pixel (kernel 2).  b = floor(min(H,W)*L) = 2 at 256/288, 3 at 384, 5 at 512.
#include "common.h"

namespace ustrun {
namespace {

constexpr int MAXNB = 11;           // (2b+1) <= 11 (b = 2 at 256/288, 3 at 384, 5 at 512)
constexpr float TWO_PI = 6.283185307179586f;

// bins[((img*C + c)*2 + which)*RS + rs][u][v] = sum over the block's rows of I[y,x] e^{-2 pi j (fu y/H + fv x/W)},
// fu = u-b, fv = v-b.  grid = (n*C*2, RS row splits, bands of UB window rows u); every thread keeps its band's UB x (2b+1)
// bins in registers (the whole window up to 7 x 7; 9 x 9 and 11 x 11 in bands of 3 and 4 rows) and makes ONE pass over
// its pixels; the RS partial rows are summed by the consumer.
constexpr int RS = 8;

template <int NB, int UB>
__global__ __launch_bounds__(256) void dft_bins_kernel(const float* __restrict__ src, const float* __restrict__ trg,
                                                      int C, int H, int W, float2* __restrict__ bins) {
    extern __shared__ float2 tw[];     // [NB][H] then [NB][W]
    __shared__ float red[4][2 * UB * NB];
    constexpr int b = NB / 2;
    const int u0 = blockIdx.z * UB;
    float2* twy = tw; float2* twx = tw + NB * H;
    const int ic = blockIdx.x >> 1, which = blockIdx.x & 1;
    const float* img = (which ? trg : src) + (long)ic * H * W;
    for (int t = threadIdx.x; t < NB * H; t += 256) {
        const int u = t / H, y = t % H;
        float s, c; sincosf(-TWO_PI * (float)(((long)(u - b) * y) % H) / (float)H, &s, &c);
        twy[t] = make_float2(c, s);
    }
    for (int t = threadIdx.x; t < NB * W; t += 256) {
        const int v = t / W, x = t % W;
        float s, c; sincosf(-TWO_PI * (float)(((long)(v - b) * x) % W) / (float)W, &s, &c);
        twx[t] = make_float2(c, s);
    }
    __syncthreads();
    float2 acc[UB][NB];
#pragma unroll
    for (int u = 0; u < UB; ++u)
#pragma unroll
        for (int v = 0; v < NB; ++v) acc[u][v] = make_float2(0.f, 0.f);
    const int rows = (H + RS - 1) / RS, y0 = blockIdx.y * rows, y1 = min(H, y0 + rows);
    for (int p = y0 * W + threadIdx.x; p < y1 * W; p += 256) {
        const int y = p / W, x = p - y * W;
        const float val = (img[p] + 1.f) * 127.5f;
        float2 tx[NB];
#pragma unroll
        for (int v = 0; v < NB; ++v) { const float2 c2 = twx[v * W + x]; tx[v] = make_float2(val * c2.x, val * c2.y); }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const float2 a = twy[min(u0 + u, NB - 1) * H + y];      // (rows past the window in the last band: computed, not stored)
#pragma unroll
            for (int v = 0; v < NB; ++v) {
                acc[u][v].x += a.x * tx[v].x - a.y * tx[v].y;
                acc[u][v].y += a.x * tx[v].y + a.y * tx[v].x;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int u = 0; u < UB; ++u)
#pragma unroll
        for (int v = 0; v < NB; ++v) {
            float re = acc[u][v].x, im = acc[u][v].y;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { re += __shfl_xor(re, o); im += __shfl_xor(im, o); }
            if (lane == 0) { red[wave][2 * (u * NB + v)] = re; red[wave][2 * (u * NB + v) + 1] = im; }
        }
    __syncthreads();
    if (threadIdx.x < UB * NB && u0 * NB + threadIdx.x < NB * NB) {
        const int t = threadIdx.x;
        bins[((long)blockIdx.x * RS + blockIdx.y) * NB * NB + u0 * NB + t] =
            make_float2(red[0][2 * t] + red[1][2 * t] + red[2][2 * t] + red[3][2 * t],
                        red[0][2 * t + 1] + red[1][2 * t + 1] + red[2][2 * t + 1] + red[3][2 * t + 1]);
    }
}

// out = clip(S + corr, 0, 255)/127.5 - 1;  grid = (row tiles, n*C)
__global__ __launch_bounds__(256) void freq_apply_kernel(const float* __restrict__ src, const float2* __restrict__ bins,
                                                        const float* __restrict__ ratios, int C, int H, int W, int b,
                                                        float* __restrict__ out) {
    extern __shared__ float2 sm[];     // coef[nb*nb], twy[nb][H], twx[nb][W]
    const int nb = 2 * b + 1;
    float2* coef = sm; float2* twy = sm + nb * nb; float2* twx = twy + nb * H;
    const int ic = blockIdx.y, n = ic / C;
    const float r = ratios[n];
    const float inv = 1.f / ((float)H * (float)W);
    for (int t = threadIdx.x; t < nb * nb; t += 256) {
        float2 fs = make_float2(0.f, 0.f), ft = make_float2(0.f, 0.f);
        for (int r2 = 0; r2 < RS; ++r2) {        // fixed-order sum of the row-split partials
            const float2 p0 = bins[(((long)ic * 2 + 0) * RS + r2) * nb * nb + t], p1 = bins[(((long)ic * 2 + 1) * RS + r2) * nb * nb + t];
            fs.x += p0.x; fs.y += p0.y; ft.x += p1.x; ft.y += p1.y;
        }
        const float as = sqrtf(fs.x * fs.x + fs.y * fs.y), at = sqrtf(ft.x * ft.x + ft.y * ft.y);
        const float2 ph = as > 0.f ? make_float2(fs.x / as, fs.y / as) : make_float2(1.f, 0.f);   // np.angle(0) = 0
        const float g = r * (at - as) * inv;
        coef[t] = make_float2(g * ph.x, g * ph.y);
    }
    for (int t = threadIdx.x; t < nb * H; t += 256) {
        const int u = t / H, y = t % H;
        float s, c; sincosf(TWO_PI * (float)(((long)(u - b) * y) % H) / (float)H, &s, &c);
        twy[t] = make_float2(c, s);
    }
    for (int t = threadIdx.x; t < nb * W; t += 256) {
        const int v = t / W, x = t % W;
        float s, c; sincosf(TWO_PI * (float)(((long)(v - b) * x) % W) / (float)W, &s, &c);
        twx[t] = make_float2(c, s);
    }
    __syncthreads();
    const long base = (long)ic * H * W;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < H * W; p += gridDim.x * 256) {
        const int y = p / W, x = p - y * W;
        float corr = 0.f;
        for (int u = 0; u < nb; ++u) {
            const float2 a = twy[u * H + y];
            float2 acc = make_float2(0.f, 0.f);      // sum_v coef[u][v] e^{+j..x}
            for (int v = 0; v < nb; ++v) {
                const float2 k = coef[u * nb + v], c2 = twx[v * W + x];
                acc.x += k.x * c2.x - k.y * c2.y;
                acc.y += k.x * c2.y + k.y * c2.x;
            }
            corr += acc.x * a.x - acc.y * a.y;
        }
        const float s = (src[base + p] + 1.f) * 127.5f;
        out[base + p] = fminf(fmaxf(s + corr, 0.f), 255.f) / 127.5f - 1.f;
    }
}

}  // namespace
}  // namespace ustrun

using namespace ustrun;

extern "C" int64_t ustrun_freq_mix_work_bytes(int n, int C, int b) {
    const int nb = 2 * b + 1;
    return (int64_t)n * C * 2 * RS * nb * nb * sizeof(float2);
}

extern "C" int ustrun_freq_mix(const float* src, const float* trg, const float* ratios, int n, int C, int H, int W,
                               int b, float* out, void* work, int64_t work_bytes, ustrun_stream_t s) {
    USTRUN_CHECK(src && trg && ratios && out && work && n > 0 && C > 0 && H > 0 && W > 0, "freq_mix: bad args");
    const int nb = 2 * b + 1;
    USTRUN_CHECK(b >= 0 && nb <= MAXNB, "freq_mix: window half-width %d unsupported", b);
    USTRUN_CHECK(work_bytes >= ustrun_freq_mix_work_bytes(n, C, b), "freq_mix: work buffer too small");
    const size_t lds1 = (size_t)nb * (H + W) * sizeof(float2);
    const size_t lds2 = lds1 + (size_t)nb * nb * sizeof(float2);
    USTRUN_CHECK(lds2 <= 160 * 1024, "freq_mix: extent %dx%d too large for the twiddle tables", H, W);
#define USTRUN_DFT_BINS(NB_, UB_)                                                                                              \
    do {                                                                                                                       \
        if (lds1 > 64 * 1024) USTRUN_TRY(ensure_dynamic_lds((const void*)dft_bins_kernel<NB_, UB_>, (int)lds1, "dft_bins"));          \
        hipLaunchKernelGGL((dft_bins_kernel<NB_, UB_>), dim3(n * C * 2, RS, (NB_ + UB_ - 1) / UB_), dim3(256), lds1,           \
                           (hipStream_t)s, src, trg, C, H, W, (float2*)work);                                                  \
    } while (0)
    switch (nb) {
        case 1: USTRUN_DFT_BINS(1, 1); break;
        case 3: USTRUN_DFT_BINS(3, 3); break;
        case 5: USTRUN_DFT_BINS(5, 5); break;
        case 7: USTRUN_DFT_BINS(7, 7); break;
        case 9: USTRUN_DFT_BINS(9, 3); break;
        default: USTRUN_DFT_BINS(11, 4); break;
    }
#undef USTRUN_DFT_BINS
    USTRUN_LAUNCH_CHECK("dft_bins");
    int tiles = cdiv((long)H * W, 256 * 8);
    if (lds2 > 64 * 1024) USTRUN_TRY(ensure_dynamic_lds((const void*)freq_apply_kernel, (int)lds2, "freq_apply"));
    hipLaunchKernelGGL(freq_apply_kernel, dim3(tiles, n * C), dim3(256), lds2, (hipStream_t)s, src, (const float2*)work, ratios,
                       C, H, W, b, out);
    USTRUN_LAUNCH_CHECK("freq_apply");
    return 0;
}
