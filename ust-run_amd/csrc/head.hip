// head.hip -- the 1x1 classification head (OutConv) forward/backward.  AI ~ 1 flop/B: HBM-bound,
// so it is a stream over the 64-channel NHWC feature map with the producer's BatchNorm+ReLU
// applied on load; LPP lanes share one pixel (16 B of channels each, one coalesced line per pixel)
// and combine their partial dot products with wavefront shuffles.  Logits are written NCHW.
#include "common.h"
#include "loader.h"

namespace ustrun {
namespace {

constexpr int KMAX = 8;

__device__ __forceinline__ f32x4 act4(f32x4 v, const float* scale, const float* shift, int c) {
    if (scale) {
        v = v * *(const f32x4*)(scale + c) + *(const f32x4*)(shift + c);
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    return v;
}

template <int ESZ>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, long npix, int HW, int C, int K,
                                                      int LPP, const float* __restrict__ w, const float* __restrict__ bias,
                                                      float* __restrict__ logits) {
    const int C4 = C / 4, PPB = 256 / LPP;
    const int cq0 = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    for (long p0 = (long)blockIdx.x * PPB; p0 < npix; p0 += (long)gridDim.x * PPB) {
        const long p = p0 + pl;
        float acc[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
        if (p < npix)
            for (int cq = cq0; cq < C4; cq += LPP) {
                const f32x4 a = act4(ld4t<ESZ>(y, p * C + cq * 4), scale, shift, cq * 4);
#pragma unroll
                for (int k = 0; k < KMAX; ++k)
                    if (k < K) {
                        const f32x4 wk = *(const f32x4*)(w + k * C + cq * 4);
                        acc[k] += a[0] * wk[0] + a[1] * wk[1] + a[2] * wk[2] + a[3] * wk[3];
                    }
            }
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K)
                for (int o = LPP >> 1; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
        if (cq0 == 0 && p < npix) {
            const long n = (unsigned)p / (unsigned)HW, hw = p - n * HW;   // 32-bit divide: p < 2^32 (checked on the host)
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) logits[(n * K + k) * HW + hw] = acc[k] + bias[k];
        }
    }
}

// bf16 feature map, C = 8*G channels with G a power of two: G lanes share a pixel, 16 bytes (8 channels) each, the
// lane's scale/shift/weights live in registers for the whole launch, eight pixels in flight per lane; after the
// butterfly every lane of the group holds the logits and lane k stores class k.
template <int K>
__global__ __launch_bounds__(256) void head_fwd_bf16_kernel(const elt_t* __restrict__ y, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, long npix, int HW, int C, int G,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ logits) {
    typedef __attribute__((ext_vector_type(8))) elt_t bf16x8;
    const int g = threadIdx.x % G, pl = threadIdx.x / G, PPB = 256 / G;
    float sc[8], sh[8], wk[K][8];
    {   // the lane's constants as 16-byte loads, all issued together (they were 8 x (2 + K) dependent 4-byte loads: a few
        // microseconds in front of a block that streams for ten)
        f32x4 t[2 + K][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            t[0][h] = scale ? *(const f32x4*)(scale + 8 * g + 4 * h) : (f32x4){1.f, 1.f, 1.f, 1.f};
            t[1][h] = scale ? *(const f32x4*)(shift + 8 * g + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < K; ++k) t[2 + k][h] = *(const f32x4*)(w + k * C + 8 * g + 4 * h);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = t[0][j >> 2][j & 3]; sh[j] = t[1][j >> 2][j & 3];
#pragma unroll
            for (int k = 0; k < K; ++k) wk[k][j] = t[2 + k][j >> 2][j & 3];
        }
    }
    const bool relu = scale != nullptr;
    const float bk = g < K ? bias[g] : 0.f;
    constexpr int U = 8;                                    // eight 16-byte loads in flight per lane
    for (long p0 = (long)blockIdx.x * PPB * U; p0 < npix; p0 += (long)gridDim.x * PPB * U) {
        bf16x8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = p0 + u * PPB + pl;
            v[u] = *(const bf16x8*)(y + (p < npix ? p : npix - 1) * C + 8 * g);   // unconditional: all U loads in flight
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = p0 + u * PPB + pl;
            float acc[K];
#pragma unroll
            for (int k = 0; k < K; ++k) acc[k] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {                   // (out-of-range lanes compute on the clamped pixel, never store)
                float a = __builtin_fmaf((float)v[u][j], sc[j], sh[j]);      // (explicit fused forms: this kernel and the 64-channel
                a = relu ? fmaxf(a, 0.f) : a;                                // one below must round alike, whatever hipcc contracts)
#pragma unroll
                for (int k = 0; k < K; ++k) acc[k] = __builtin_fmaf(a, wk[k][j], acc[k]);
            }
#pragma unroll
            for (int k = 0; k < K; ++k)
                for (int o = G >> 1; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
            if (g < K && p < npix) {
                float out = acc[0];
#pragma unroll
                for (int k = 1; k < K; ++k) if (g == k) out = acc[k];
                const long n = (unsigned)p / (unsigned)HW, hw = p - n * HW;   // 32-bit divide: p < 2^32 (checked on the host)
                logits[(n * K + g) * HW + hw] = out + bk;
            }
        }
    }
}

// The same for C = 64 (eight lanes per pixel) and H*W a multiple of 256 -- the head of every U-Net configuration of the step --
// with what made the generic kernel VALU-bound (about 100 VALU instructions per 16-byte piece: 3.2 TB/s) taken out: a trip's 256
// pixels lie inside one image (no per-pixel division, no clamped addresses, one scalar divide per trip); the eight lanes of a pixel
// reduce the K class sums as a reduce-scatter (at every xor step a lane keeps half of the classes it still holds and sends the
// other half: K = 2 takes three shuffles instead of six) -- same pairing and the same additions as the butterfly, commuted:
// bit-identical logits; class k ends on lane 4 (k & 1) [K <= 2] or 4 (k >> 1) + 2 (k & 1) [K <= 4].
template <int K>
__global__ __launch_bounds__(256) void head_fwd_bf16_g8_kernel(const elt_t* __restrict__ y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, long npix, int HW,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              float* __restrict__ logits) {
    typedef __attribute__((ext_vector_type(8))) elt_t bf16x8;
    constexpr int C = 64, PPB = 32, U = 8, KP = K <= 1 ? 1 : (K <= 2 ? 2 : 4);
    const int g = threadIdx.x & 7, pl = threadIdx.x >> 3;
    float sc[8], sh[8], wk[KP][8];
    {
        f32x4 t[2 + KP][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            t[0][h] = scale ? *(const f32x4*)(scale + 8 * g + 4 * h) : (f32x4){1.f, 1.f, 1.f, 1.f};
            t[1][h] = scale ? *(const f32x4*)(shift + 8 * g + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < KP; ++k) t[2 + k][h] = k < K ? *(const f32x4*)(w + k * C + 8 * g + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = t[0][j >> 2][j & 3]; sh[j] = t[1][j >> 2][j & 3];
#pragma unroll
            for (int k = 0; k < KP; ++k) wk[k][j] = t[2 + k][j >> 2][j & 3];
        }
    }
    const bool relu = scale != nullptr;
    // the class this lane ends up holding, and whether it stores
    const int kcls = KP == 1 ? 0 : (KP == 2 ? (g >> 2) : ((g >> 2) * 2 + ((g >> 1) & 1)));
    const bool writer = (KP == 1 ? g == 0 : (KP == 2 ? (g & 3) == 0 : (g & 1) == 0)) && kcls < K;
    const float bk = writer ? bias[kcls] : 0.f;
    for (long p0 = (long)blockIdx.x * PPB * U; p0 < npix; p0 += (long)gridDim.x * PPB * U) {
        const int n = __builtin_amdgcn_readfirstlane((int)((unsigned)p0 / (unsigned)HW));     // the trip's image (uniform)
        const int hw0 = (int)(p0 - (long)n * HW);
        bf16x8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *(const bf16x8*)(y + (p0 + u * PPB + pl) * C + 8 * g);
        float* lrow = logits + ((long)n * K + kcls) * HW + hw0 + pl;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float acc[KP];
#pragma unroll
            for (int k = 0; k < KP; ++k) acc[k] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = __builtin_fmaf((float)v[u][j], sc[j], sh[j]);
                a = relu ? fmaxf(a, 0.f) : a;
#pragma unroll
                for (int k = 0; k < KP; ++k) acc[k] = __builtin_fmaf(a, wk[k][j], acc[k]);
            }
            float r;
            if constexpr (KP == 1) {
                r = acc[0];
                r += __shfl_xor(r, 4); r += __shfl_xor(r, 2); r += __shfl_xor(r, 1);
            } else if constexpr (KP == 2) {
                const bool hi = g & 4;
                r = (hi ? acc[1] : acc[0]) + __shfl_xor(hi ? acc[0] : acc[1], 4);
                r += __shfl_xor(r, 2); r += __shfl_xor(r, 1);
            } else {
                const bool hi = g & 4, mid = g & 2;
                const float k0 = (hi ? acc[2] : acc[0]) + __shfl_xor(hi ? acc[0] : acc[2], 4);
                const float k1 = (hi ? acc[3] : acc[1]) + __shfl_xor(hi ? acc[1] : acc[3], 4);
                r = (mid ? k1 : k0) + __shfl_xor(mid ? k0 : k1, 2);
                r += __shfl_xor(r, 1);
            }
            if (writer) lrow[u * PPB] = r + bk;
        }
    }
}

// da[p][c] = sum_k dl[k][p] w[k][c];  block partials of dW[k][c] = sum_p dl[k][p] a[p][c], db[k] = sum_p dl[k][p]
// partials[block][K*C + K]
// KT > 0: the class count is a compile-time constant (no per-class branches: every dl load of a pixel is issued before
// the first use -- with the runtime K the loads sat one behind the other, each inside its own `k < K` branch);
// KT == 0: any K <= KMAX.
template <int ESZ, int KT>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ y,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      long npix, int HW, int C, int Krt, int LPP,
                                                      const float* __restrict__ w, float* __restrict__ da,
                                                      float* __restrict__ partials, long pass_aff, int bnr) {
    constexpr int KN = KT > 0 ? KT : KMAX;           // accumulators kept
    const int K = KT > 0 ? KT : Krt;
    // blockIdx.y = forward pass of a batched call: its own slice of dl / y / da (npix pixels each), its own BatchNorm
    // constants (pass_aff floats apart) and its own partial rows
    {
        const long g = blockIdx.y;
        dl += g * npix * K;
        y = (const float*)((const char*)y + g * npix * C * ESZ);
        da = (float*)((char*)da + g * npix * C * ESZ);
        if (scale) { scale += g * pass_aff; shift += g * pass_aff; }
        partials += g * (long)gridDim.x * (K * C + K + (bnr ? 2 * C : 0));
    }
    extern __shared__ float red[];   // [PPB][K*C + K] would be large; reduce per k instead (below)
    const int C4 = C / 4, PPB = 256 / LPP;
    const int cq0 = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    // bnr: the BatchNorm backward of the layer under the head needs sum(da * mask) and sum(da * mask * y) per channel over
    // exactly the tensors this kernel already holds in registers (y on load, da before its store): formed here, as two more
    // 'classes' of the row, instead of by a bn_bwd_reduce pass that reads both tensors again (1.07 GB per step at N = 64)
    const int row = K * C + K + (bnr ? 2 * C : 0);
    float* out = partials + (long)blockIdx.x * row;
    for (int cb = 0; cb < C4; cb += LPP) {          // uniform trip count (barriers inside)
        const int cq = cb + cq0;
        const bool active = cq < C4;
        const int c = active ? cq * 4 : 0;
        f32x4 dwp[KN]; float dbp[KN];
        f32x4 bs1 = {0.f, 0.f, 0.f, 0.f}, bs2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KN; ++k) { dwp[k] = (f32x4){0.f, 0.f, 0.f, 0.f}; dbp[k] = 0.f; }
        f32x4 wk[KN];
#pragma unroll
        for (int k = 0; k < KN; ++k) wk[k] = (k < K) ? *(const f32x4*)(w + k * C + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 scv = {1.f, 1.f, 1.f, 1.f}, shv = {0.f, 0.f, 0.f, 0.f};
        if (scale && active) { scv = *(const f32x4*)(scale + c); shv = *(const f32x4*)(shift + c); }
        if (active)
            for (long p = (long)blockIdx.x * PPB + pl; p < npix; p += (long)gridDim.x * PPB) {
                const long n = (unsigned)p / (unsigned)HW, hw = p - n * HW;   // 32-bit divide: p < 2^32 (checked on the host)
                f32x4 a = ld4t<ESZ>(y, p * C + c);
                const f32x4 yraw = a;
                float d[KN];
#pragma unroll
                for (int k = 0; k < KN; ++k) d[k] = (KT > 0 || k < K) ? dl[(n * K + k) * HW + hw] : 0.f;
                if (scale) {
                    a = a * scv + shv;
                    a[0] = fmaxf(a[0], 0.f); a[1] = fmaxf(a[1], 0.f); a[2] = fmaxf(a[2], 0.f); a[3] = fmaxf(a[3], 0.f);
                }
                f32x4 g = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < KN; ++k)
                    if (KT > 0 || k < K) {
                        g += d[k] * wk[k];
                        dwp[k] += d[k] * a;
                        dbp[k] += d[k];
                    }
                st4t<ESZ>(da, p * C + c, g);
                if (bnr) {                                   // on the values as stored (what a reduce pass would read back)
                    if (ESZ == 2) { g[0] = (float)(elt_t)g[0]; g[1] = (float)(elt_t)g[1]; g[2] = (float)(elt_t)g[2]; g[3] = (float)(elt_t)g[3]; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float dz = a[q] > 0.f ? g[q] : 0.f;
                        bs1[q] += dz; bs2[q] += dz * yraw[q];
                    }
                }
            }
        // fixed-order block reduction over the PPB pixel lanes, one class at a time
        for (int k = 0; k < K; ++k) {
            f32x4 v = dwp[0]; float b = dbp[0];
#pragma unroll
            for (int kk = 1; kk < KN; ++kk) if (kk == k) { v = dwp[kk]; b = dbp[kk]; }
            ((f32x4*)red)[threadIdx.x] = v;
            red[1024 + threadIdx.x] = b;
            __syncthreads();
            if (pl == 0 && active) {
                for (int q = 1; q < PPB; ++q) v += ((f32x4*)red)[q * LPP + cq0];
                *(f32x4*)(out + k * C + c) = v;
                if (cb == 0 && cq0 == 0) {
                    for (int q = 1; q < PPB; ++q) b += red[1024 + q * LPP];
                    out[K * C + k] = b;
                }
            }
            __syncthreads();
        }
        if (bnr)
            for (int h = 0; h < 2; ++h) {
                f32x4 v = h ? bs2 : bs1;
                ((f32x4*)red)[threadIdx.x] = v;
                __syncthreads();
                if (pl == 0 && active) {
                    for (int q = 1; q < PPB; ++q) v += ((f32x4*)red)[q * LPP + cq0];
                    *(f32x4*)(out + K * C + K + h * C + c) = v;
                }
                __syncthreads();
            }
    }
}

// block = 32 outputs x 32 slab lanes (1024 threads), four rows in flight per lane, fixed-order f64 combine
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ part, int nslab, long stride, long offset,
                                                          int count, float* __restrict__ out, int accumulate) {
    __shared__ double red[32][32];
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + el;
    double v = 0.0;
    if (e < count) {
        int k = sl;
        for (; k + 96 < nslab; k += 128) {            // rows k, k+32, k+64, k+96: loads issued together, summed in order
            const float a0 = part[(long)k * stride + offset + e], a1 = part[(long)(k + 32) * stride + offset + e];
            const float a2 = part[(long)(k + 64) * stride + offset + e], a3 = part[(long)(k + 96) * stride + offset + e];
            v += (double)a0; v += (double)a1; v += (double)a2; v += (double)a3;
        }
        for (; k < nslab; k += 32) v += (double)part[(long)k * stride + offset + e];
    }
    red[sl][el] = v;
    __syncthreads();
    if (sl == 0 && e < count) {
        for (int q = 1; q < 32; ++q) v += red[q][el];
        out[e] = accumulate ? out[e] + (float)v : (float)v;
    }
}

int lanes_per_pixel(int C4) { int g = 1; while (g < C4 && g < 64) g <<= 1; return g; }

}  // namespace

int reduce_rows(const float* part, int nslab, long stride, long offset, int count, float* out, int accumulate,
                hipStream_t st) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(cdiv(count, 32)), dim3(1024), 0, st, part, nslab, stride, offset, count,
                       out, accumulate);
    USTRUN_LAUNCH_CHECK("reduce_rows");
    return 0;
}

}  // namespace ustrun

using namespace ustrun;

static int head_blocks(int64_t npix, int LPP) {
    const int PPB = 256 / LPP;
    long b = (npix + (long)PPB * 16 - 1) / ((long)PPB * 16);
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int ustrun_head_fwd(const void* y, const float* scale, const float* shift, int64_t npix, int HW, int C,
                               int K, const float* w, const float* bias, float* logits, int dtype,
                               ustrun_stream_t s) {
    USTRUN_CHECK(dtype_ok(dtype), "head_fwd: dtype %d not built", dtype);
    USTRUN_CHECK(y && w && bias && logits, "head_fwd: null pointer");
    USTRUN_CHECK(C % 4 == 0 && C > 0 && K >= 1 && K <= KMAX, "head_fwd: C=%d K=%d unsupported", C, K);
    USTRUN_CHECK(npix > 0 && HW > 0 && npix % HW == 0 && npix < (1LL << 32), "head_fwd: bad extent");
    const int G = C / 8;
    const bool al16 = !(((uintptr_t)w | (uintptr_t)scale | (uintptr_t)shift) & 15);      // the constants go in as 16-byte loads
    if (dtype == USTRUN_D16 && C % 8 == 0 && G <= 64 && (G & (G - 1)) == 0 && K >= 1 && K <= 4 && K <= G && al16) {
        long nb = (npix + (256 / G) * 32 - 1) / ((256 / G) * 32);    // >= four rounds of eight pixels per lane (the constants'
        if (nb > 2048) nb = 2048;                                    // loads and the launch are then a small part of a block)
        dim3 grid((int)nb), block(256);
        if (C == 64 && HW % 256 == 0 && !(g_debug_flags & (1 << 27))) {      // (bit 27: the generic kernel, A/B and parity runs)
#define USTRUN_HF8(KK) hipLaunchKernelGGL(head_fwd_bf16_g8_kernel<KK>, grid, block, 0, (hipStream_t)s, (const elt_t*)y, scale, shift, \
                                          (long)npix, HW, w, bias, logits)
            if (K == 1) USTRUN_HF8(1); else if (K == 2) USTRUN_HF8(2); else if (K == 3) USTRUN_HF8(3); else USTRUN_HF8(4);
#undef USTRUN_HF8
            USTRUN_LAUNCH_CHECK("head_fwd_bf16_g8");
            return 0;
        }
#define USTRUN_HF(KK) hipLaunchKernelGGL(head_fwd_bf16_kernel<KK>, grid, block, 0, (hipStream_t)s, (const elt_t*)y, scale, shift, \
                                         (long)npix, HW, C, G, w, bias, logits)
        if (K == 1) USTRUN_HF(1); else if (K == 2) USTRUN_HF(2); else if (K == 3) USTRUN_HF(3); else USTRUN_HF(4);
#undef USTRUN_HF
        USTRUN_LAUNCH_CHECK("head_fwd_bf16");
        return 0;
    }
    const int LPP = lanes_per_pixel(C / 4);
    long blocks = (npix + (256 / LPP) * 4 - 1) / ((256 / LPP) * 4);
    if (blocks > 8192) blocks = 8192;
    if (dtype == USTRUN_D16)
        hipLaunchKernelGGL(head_fwd_kernel<2>, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, (const float*)y, scale, shift,
                           (long)npix, HW, C, K, LPP, w, bias, logits);
    else
        hipLaunchKernelGGL(head_fwd_kernel<4>, dim3((int)blocks), dim3(256), 0, (hipStream_t)s, (const float*)y, scale, shift,
                           (long)npix, HW, C, K, LPP, w, bias, logits);
    USTRUN_LAUNCH_CHECK("head_fwd");
    return 0;
}

namespace ustrun {
// `passes` forward passes (npix pixels each) in one launch: four times the blocks in flight of a per-pass call; the
// passes' partial rows are summed together in one fixed order
int head_bwd_passes(const float* dlogits, const void* y, const float* scale, const float* shift, int64_t npix, int HW, int C,
                    int K, const float* w, void* da, float* dw, float* db, int accumulate, float* partials,
                    int64_t partials_bytes, int dtype, int passes, long pass_aff, hipStream_t s, int* bn_rows) {
    USTRUN_CHECK(dtype_ok(dtype), "head_bwd: dtype %d not built", dtype);
    USTRUN_CHECK(dlogits && y && w && da && dw && db && partials, "head_bwd: null pointer");
    USTRUN_CHECK(C % 4 == 0 && C > 0 && K >= 1 && K <= KMAX, "head_bwd: C=%d K=%d unsupported", C, K);
    USTRUN_CHECK(npix > 0 && HW > 0 && npix < (1LL << 32) && passes >= 1, "head_bwd: bad extent");
    const int LPP = lanes_per_pixel(C / 4);
    int blocks = head_blocks(npix, LPP);
    // batched passes: 1024 partial rows in all (four blocks per CU), as the BatchNorm-backward reduce does -- the finalize that
    // follows walks a pass's rows on eight lanes
    if (passes > 1 && blocks > 1024 / passes) blocks = 1024 / passes > 64 ? 1024 / passes : 64;
    // bn_rows: also form the BatchNorm-backward sums of the layer under the head (scale/shift = its constants): row =
    // [K*C dW | K db | C sum(da mask) | C sum(da mask y)], *bn_rows = rows per pass (rows are dword-aligned only, as they always
    // were: 16-byte global stores take that)
    const int bnr = bn_rows && scale;
    if (bn_rows) *bn_rows = bnr ? blocks : 0;
    const long row = (long)K * C + K + (bnr ? 2 * C : 0);
    USTRUN_CHECK(partials_bytes >= (int64_t)((long)(passes > 1 ? passes * blocks : 1024) * row * 4),
                 "head_bwd: partials too small (%lld bytes for %d passes)", (long long)partials_bytes, passes);
#define USTRUN_HB(E, KT)                                                                                                     \
    hipLaunchKernelGGL((head_bwd_kernel<E, KT>), dim3(blocks, passes), dim3(256), (1024 + 256) * sizeof(float), s, dlogits,    \
                       (const float*)y, scale, shift, (long)npix, HW, C, K, LPP, w, (float*)da, partials, pass_aff, bnr)
    if (dtype == USTRUN_D16) { if (K == 2) USTRUN_HB(2, 2); else if (K == 4) USTRUN_HB(2, 4); else USTRUN_HB(2, 0); }
    else { if (K == 2) USTRUN_HB(4, 2); else if (K == 4) USTRUN_HB(4, 4); else USTRUN_HB(4, 0); }
#undef USTRUN_HB
    USTRUN_LAUNCH_CHECK("head_bwd");
    USTRUN_TRY(reduce_rows(partials, blocks * passes, row, 0, K * C, dw, accumulate, s));
    USTRUN_TRY(reduce_rows(partials, blocks * passes, row, (long)K * C, K, db, accumulate, s));
    return 0;
}
}  // namespace ustrun

extern "C" int ustrun_head_bwd(const float* dlogits, const void* y, const float* scale, const float* shift,
                               int64_t npix, int HW, int C, int K, const float* w, void* da, float* dw, float* db,
                               int accumulate, float* partials, int64_t partials_bytes, int dtype, ustrun_stream_t s) {
    return head_bwd_passes(dlogits, y, scale, shift, npix, HW, C, K, w, da, dw, db, accumulate, partials, partials_bytes, dtype, 1,
                           0, (hipStream_t)s, nullptr);
}
