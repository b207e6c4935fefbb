// loss.hip -- the `ce + dice` term of the step, forward (reductions) and backward (dlogits), and
// the pseudo-label / target-mixing streams.  Everything is HBM-bound: one lane per pixel reads the
// K logit planes (NCHW: coalesced per plane), reductions go wavefront shuffle -> LDS -> one
// partial row per block -> single-block f64 finalize (fixed order: reproducible).
#include "common.h"
#include <cstring>

namespace ustrun {
namespace {

constexpr int KMAX = 8;
constexpr int NS = 1 + 3 * KMAX;   // per-thread running sums: ce, I[k], Z[k], Y[k]
constexpr float SMOOTH = 1e-10f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// softmax mode: target int64 [N,HW], mask f32 [N,HW] or null.  sigmoid mode: target/mask f32 [N,K,HW].
__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const float* __restrict__ logits, const void* __restrict__ target,
                                                          const float* __restrict__ mask, int N, int K, int HW, int mode,
                                                          float* __restrict__ partials) {
    __shared__ float red[4][NS];
    float acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] = 0.f;
    const long npix = (long)N * HW;
    if (mode == USTRUN_LOSS_SOFTMAX) {
        const long long* tgt = (const long long*)target;
        for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
            const long n = (unsigned)p / (unsigned)HW, hw = p - n * HW;   // 32-bit divide: p < 2^32 (checked on the host)
            float l[KMAX], mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (k < K) { l[k] = logits[(n * K + k) * HW + hw]; mx = fmaxf(mx, l[k]); }
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (k < K) { l[k] = expf(l[k] - mx); se += l[k]; }
            const int t = (int)tgt[p];
            const float m = mask ? mask[p] : 1.f;
            const float m1 = mask ? (m == 1.f ? 1.f : 0.f) : 1.f;
            const float inv = 1.f / se;
            float lt = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
                    const float pk = l[k] * inv;
                    const float mk = (k == 0) ? 1.f : m1;          // class-0 mask channel is all ones (Q4)
                    const float tk = (t == k) ? 1.f : 0.f;
                    if (t == k) lt = logits[(n * K + k) * HW + hw] - mx;
                    acc[1 + k] += pk * tk * mk;
                    acc[1 + KMAX + k] += pk * pk * mk;
                    acc[1 + 2 * KMAX + k] += tk * mk;
                }
            acc[0] += (logf(se) - lt) * m;
        }
    } else {
        const float* tgt = (const float*)target;
        const long total = npix * K;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
            const float x = logits[e], t = tgt[e], m = mask ? mask[e] : 1.f;
            const float ea = expf(-fabsf(x));
            acc[0] += (fmaxf(x, 0.f) - x * t + log1pf(ea)) * m;
            const float p = x >= 0.f ? 1.f / (1.f + ea) : ea / (1.f + ea);
            acc[1] += p * t * m;
            acc[1 + KMAX] += p * p * m;
            acc[1 + 2 * KMAX] += t * t * m;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NS; ++i) { const float v = wave_sum(acc[i]); if (lane == 0) red[wave][i] = v; }
    __syncthreads();
    if (threadIdx.x < NS)
        partials[(long)blockIdx.x * NS + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[0] = ce mean, out[1] = dice, out[2] = ce sum, out[3+k] = I, out[3+K+k] = Z, out[3+2K+k] = Y
__global__ void seg_loss_finalize_kernel(const float* __restrict__ partials, int rows, int N, int K, int HW, int mode,
                                         float* __restrict__ out) {
    __shared__ double tot[NS];
    __shared__ double red[32][32];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;      // 32 columns (NS used) x 32 row lanes (1024 threads)
    double v = 0.0;
    if (col < NS) {
        int r = rg;
        for (; r + 96 < rows; r += 128) {                         // four rows in flight, summed in order
            const float a0 = partials[(long)r * NS + col], a1 = partials[(long)(r + 32) * NS + col];
            const float a2 = partials[(long)(r + 64) * NS + col], a3 = partials[(long)(r + 96) * NS + col];
            v += (double)a0; v += (double)a1; v += (double)a2; v += (double)a3;
        }
        for (; r < rows; r += 32) v += (double)partials[(long)r * NS + col];
    }
    red[rg][col] = v;
    __syncthreads();
    if (threadIdx.x < NS) {
        double t = 0.0;
        for (int k = 0; k < 32; ++k) t += red[k][threadIdx.x];
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int kk = mode == USTRUN_LOSS_SOFTMAX ? K : 1;
        const double cnt = mode == USTRUN_LOSS_SOFTMAX ? (double)N * HW : (double)N * K * HW;
        double dice = 0.0;
        for (int k = 0; k < kk; ++k) {
            const double I = tot[1 + k], Z = tot[1 + KMAX + k], Y = tot[1 + 2 * KMAX + k];
            dice += 1.0 - (2.0 * I + (double)SMOOTH) / (Z + Y + (double)SMOOTH);
            out[3 + k] = (float)I; out[3 + kk + k] = (float)Z; out[3 + 2 * kk + k] = (float)Y;
        }
        out[0] = (float)(tot[0] / cnt);
        out[1] = (float)(dice / kk);
        out[2] = (float)tot[0];
    }
}

__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ logits, const void* __restrict__ target,
                                                          const float* __restrict__ mask, int N, int K, int HW, int mode,
                                                          const float* __restrict__ sums, const float* __restrict__ gdev,
                                                          float gscale, float cw, float dw, float* __restrict__ dlogits) {
    // gdev (optional) = upstream gradients of the two outputs {d/d ce, d/d dice} on the device
    const float gs = gscale;
    cw *= gdev ? gdev[0] : 1.f;
    dw *= gdev ? gdev[1] : 1.f;
    const long npix = (long)N * HW;
    if (mode == USTRUN_LOSS_SOFTMAX) {
        const long long* tgt = (const long long*)target;
        float A[KMAX], Bc[KMAX];   // dDice/dp_k = A[k]*t_k*m_k + Bc[k]*p_k*m_k
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < K) {
                const float I = sums[3 + k], D = sums[3 + K + k] + sums[3 + 2 * K + k] + SMOOTH;
                A[k] = -2.f / D / K;
                Bc[k] = 2.f * (2.f * I + SMOOTH) / (D * D) / K;
            }
        const float cen = cw / (float)npix;
        for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
            const long n = (unsigned)p / (unsigned)HW, hw = p - n * HW;   // 32-bit divide: p < 2^32 (checked on the host)
            float l[KMAX], mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (k < K) { l[k] = logits[(n * K + k) * HW + hw]; mx = fmaxf(mx, l[k]); }
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (k < K) { l[k] = expf(l[k] - mx); se += l[k]; }
            const float inv = 1.f / se;
            const int t = (int)tgt[p];
            const float m = mask ? mask[p] : 1.f;
            const float m1 = mask ? (m == 1.f ? 1.f : 0.f) : 1.f;
            float G[KMAX], gp = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
                    l[k] *= inv;
                    const float mk = (k == 0) ? 1.f : m1, tk = (t == k) ? 1.f : 0.f;
                    G[k] = dw * (A[k] * tk * mk + Bc[k] * l[k] * mk);
                    gp += G[k] * l[k];
                }
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
                if (k < K) {
                    const float tk = (t == k) ? 1.f : 0.f;
                    dlogits[(n * K + k) * HW + hw] = gs * (cen * m * (l[k] - tk) + l[k] * (G[k] - gp));
                }
        }
    } else {
        const float* tgt = (const float*)target;
        const long total = npix * K;
        const float I = sums[3], D = sums[4] + sums[5] + SMOOTH;
        const float A = -2.f / D, Bc = 2.f * (2.f * I + SMOOTH) / (D * D);
        const float cen = cw / (float)total;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
            const float x = logits[e], t = tgt[e], m = mask ? mask[e] : 1.f;
            const float ea = expf(-fabsf(x));
            const float p = x >= 0.f ? 1.f / (1.f + ea) : ea / (1.f + ea);
            const float G = dw * (A * t * m + Bc * p * m);
            dlogits[e] = gs * (cen * m * (p - t) + G * p * (1.f - p));
        }
    }
}

__global__ __launch_bounds__(256) void pseudo_label_kernel(const float* __restrict__ logits, int N, int K, int HW,
                                                          float th, int mode, void* __restrict__ label,
                                                          float* __restrict__ mask) {
    const long npix = (long)N * HW;
    if (mode == USTRUN_LOSS_SOFTMAX) {
        long long* lab = (long long*)label;
        for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
            const long n = (unsigned)p / (unsigned)HW, hw = p - n * HW;   // 32-bit divide: p < 2^32 (checked on the host)
            float l[KMAX], mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (k < K) { l[k] = logits[(n * K + k) * HW + hw]; mx = fmaxf(mx, l[k]); }
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) if (k < K) { l[k] = expf(l[k] - mx); se += l[k]; }
            // max over the PROBABILITIES, first index on ties (torch.max semantics)
            int best = 0; float bp = l[0] / se;
#pragma unroll
            for (int k = 1; k < KMAX; ++k) if (k < K) { const float pk = l[k] / se; if (pk > bp) { bp = pk; best = k; } }
            lab[p] = best;
            mask[p] = bp > th ? 1.f : 0.f;
        }
    } else {
        float* lab = (float*)label;
        const long total = npix * K;
        const float lo = 1.f - th;
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
            const float x = logits[e];
            const float p = 1.f / (1.f + expf(-x));
            lab[e] = p >= 0.5f ? 1.f : 0.f;
            mask[e] = (p >= th ? 1.f : 0.f) + (p <= lo ? 1.f : 0.f);
        }
    }
}

// train.py:677-697 on the flattened tensors; cut_label/cut_mask already gathered by `choice`
__global__ __launch_bounds__(256) void mix_targets_kernel(int mode, int N, int K, int HW, const float* __restrict__ box,
                                                         const void* pl_, const float* mask, const void* plwul_,
                                                         const float* mwul, const void* plwlu_, const float* mwlu,
                                                         const void* cutl_, const float* cutm, void* plw_, float* mw,
                                                         void* plul_, float* mul, void* pllu_, float* mlu) {
    const int CH = mode == USTRUN_LOSS_SOFTMAX ? 1 : K;
    const long total = (long)N * CH * HW;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long n = e / ((long)CH * HW), hw = e % HW;
        const float b = box[n * HW + hw], nb = 1.f - b;
        const float m = mask[e];
        float mwv = mwul[e] * nb + mwlu[e] * b;
        if (mode == USTRUN_LOSS_SOFTMAX) {
            const long long pl = ((const long long*)pl_)[e], cl = ((const long long*)cutl_)[e];
            const long long plw = (long long)((float)((const long long*)plwul_)[e] * nb + (float)((const long long*)plwlu_)[e] * b);
            const float ens = (plw == pl ? 1.f : 0.f) * m;
            if (ens == 0.f) mwv = 0.f;
            ((long long*)plw_)[e] = plw;
            ((long long*)plul_)[e] = (long long)((float)pl * nb + (float)cl * b);
            ((long long*)pllu_)[e] = (long long)((float)cl * nb + (float)pl * b);
        } else {
            const float pl = ((const float*)pl_)[e], cl = ((const float*)cutl_)[e];
            const float plw = (float)(long long)(((const float*)plwul_)[e] * nb + ((const float*)plwlu_)[e] * b);
            const float ens = (plw == pl ? 1.f : 0.f) * m;
            if (ens == 0.f) mwv = 0.f;
            ((float*)plw_)[e] = plw;
            ((float*)plul_)[e] = (float)(long long)(pl * nb + cl * b);
            ((float*)pllu_)[e] = (float)(long long)(cl * nb + pl * b);
        }
        mw[e] = mwv;
        mul[e] = b == 1.f ? cutm[e] : m;
        mlu[e] = b == 0.f ? cutm[e] : m;
    }
}

__global__ __launch_bounds__(256) void box_mix_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ box, int N, int C, int HW,
                                                     float* __restrict__ out) {
    const long total = (long)N * C * HW;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long n = e / ((long)C * HW), hw = e % HW;
        const float bx = box[n * HW + hw];
        out[e] = a[e] * (1.f - bx) + b[e] * bx;
    }
}

// blockIdx.x = (sample, class), blockIdx.y = slice of the pixels: exact integer counts {|pred|, |gt|, |pred & gt|}
__global__ __launch_bounds__(256) void dice_counts_kernel(const void* pred, const void* gt, int pi64, int gi64, int K,
                                                         int HW, int by_class, int* __restrict__ counts) {
    __shared__ int red[4][3];
    const int n = blockIdx.x / K, c = blockIdx.x % K;
    const long base = by_class ? (long)n * HW : ((long)n * K + c) * HW;
    int s = 0, g = 0, i = 0;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < HW; e += 256 * gridDim.y) {
        bool pb, gb;
        if (by_class) {
            const long long pv = pi64 ? ((const long long*)pred)[base + e] : (long long)((const float*)pred)[base + e];
            const long long gv = gi64 ? ((const long long*)gt)[base + e] : (long long)((const float*)gt)[base + e];
            pb = pv == c + 1; gb = gv == c + 1;
        } else {
            pb = pi64 ? ((const long long*)pred)[base + e] != 0 : ((const float*)pred)[base + e] != 0.f;
            gb = gi64 ? ((const long long*)gt)[base + e] != 0 : ((const float*)gt)[base + e] != 0.f;
        }
        s += pb; g += gb; i += pb && gb;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); g += __shfl_xor(g, o); i += __shfl_xor(i, o); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = s; red[wave][1] = g; red[wave][2] = i; }
    __syncthreads();
    if (threadIdx.x < 3) {      // integer sums: the order of the blocks' contributions does not matter
        const int v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (gridDim.y == 1) counts[(long)blockIdx.x * 3 + threadIdx.x] = v;
        else if (v) atomicAdd(&counts[(long)blockIdx.x * 3 + threadIdx.x], v);
    }
}

// SGD(momentum, wd) + EMA over flat f32 buffers, 16 B per lane
__global__ __launch_bounds__(256) void sgd_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v,
                                                     float* __restrict__ t, long n, float lr, float mu, float wd, int first,
                                                     float alpha, float gsc) {
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 pv = ((f32x4*)p)[i];
        const f32x4 gv = ((const f32x4*)g)[i] * gsc + wd * pv;
        const f32x4 vv = first ? gv : mu * ((f32x4*)v)[i] + gv;
        ((f32x4*)v)[i] = vv;
        pv = pv - lr * vv;
        ((f32x4*)p)[i] = pv;
        if (t) ((f32x4*)t)[i] = alpha * ((f32x4*)t)[i] + (1.f - alpha) * pv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = n4 * 4 + threadIdx.x;
        float pv = p[i];
        const float gv = g[i] * gsc + wd * pv;
        const float vv = first ? gv : mu * v[i] + gv;
        v[i] = vv; pv -= lr * vv; p[i] = pv;
        if (t) t[i] = alpha * t[i] + (1.f - alpha) * pv;
    }
}

// ---- dynamic loss scale for the IEEE-half path = torch.cuda.amp.GradScaler as the reference uses it (train.py:552,842-845:
// scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()), kept on the device so that no step waits for the host.
// amp_state = {scale, scale, growth_tracker, found_inf, steps skipped, steps seen, -, -}: the first two floats are what ustrun_seg_loss_bwd / ustrun_dice_bwd read
// as the upstream gradients of their two outputs (gscale_dev), so the scale enters the backward there.
__global__ __launch_bounds__(256) void amp_check_kernel(const float* __restrict__ g, long n, float* __restrict__ state) {
    const long n4 = n / 4;
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = ((const f32x4*)g)[i];
        // a finite float has an exponent field below 0xff; the sum of four finite values may overflow, so test each
        bad |= !(__builtin_isfinite(v[0]) && __builtin_isfinite(v[1]) && __builtin_isfinite(v[2]) && __builtin_isfinite(v[3]));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= !__builtin_isfinite(g[n4 * 4 + threadIdx.x]);
    if (__any(bad) && (threadIdx.x & 63) == 0) state[3] = 1.f;          // idempotent store: no atomics, any order
}

// the same update as sgd_ema_kernel with the gradient divided by the loss scale; found_inf set: GradScaler.step skips
// optimizer.step (parameters and momentum stay), the EMA line of the loop still runs (train.py:848-851)
__global__ __launch_bounds__(256) void sgd_ema_scaled_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v,
                                                            float* __restrict__ t, long n, float lr, float mu, float wd, int first,
                                                            float alpha, float gsc, const float* __restrict__ state) {
    const bool skip = state[3] != 0.f;
    gsc = gsc / state[0];
    const long n4 = n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 pv = ((f32x4*)p)[i];
        if (!skip) {
            const f32x4 gv = ((const f32x4*)g)[i] * gsc + wd * pv;
            const f32x4 vv = first ? gv : mu * ((f32x4*)v)[i] + gv;
            ((f32x4*)v)[i] = vv;
            pv = pv - lr * vv;
            ((f32x4*)p)[i] = pv;
        }
        if (t) ((f32x4*)t)[i] = alpha * ((f32x4*)t)[i] + (1.f - alpha) * pv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = n4 * 4 + threadIdx.x;
        float pv = p[i];
        if (!skip) {
            const float gv = g[i] * gsc + wd * pv;
            const float vv = first ? gv : mu * v[i] + gv;
            v[i] = vv; pv -= lr * vv; p[i] = pv;
        }
        if (t) t[i] = alpha * t[i] + (1.f - alpha) * pv;
    }
}

// GradScaler.update(): back off after a skipped step, grow after `interval` clean ones; clears found_inf for the next step
__global__ void amp_update_kernel(float* __restrict__ state, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float scale = state[0], tracker = state[2];
    state[5] += 1.f;                                                    // steps seen / steps skipped: logging and tests
    if (state[3] != 0.f) { scale *= backoff; tracker = 0.f; state[4] += 1.f; }
    else {
        tracker += 1.f;
        if (tracker >= (float)interval) { scale *= growth; tracker = 0.f; }
    }
    state[0] = scale; state[1] = scale; state[2] = tracker; state[3] = 0.f;
}

int stream_blocks(long work_items) {
    long b = (work_items + 256 * 4 - 1) / (256 * 4);
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- DiceLossWithMask.forward in EVERY mode combination (utils/losses.py:236-268): the two the step uses run fused with their
// CE / BCE terms above; this pair serves the rest of the reference signature (class weights, sigmoid per class, softmax + multi,
// raw inputs).  One lane per pixel, all K classes of the pixel in registers.
//   act: 0 none, 1 softmax over classes, 2 sigmoid          multi: one global Dice over [N,K,HW] instead of per class
//   per class: t_k = (target == k) from a class-index map [N,HW] (int64 or f32), m_k = 1 for k = 0, (mask == 1) for k >= 1 (Q4)
//   multi:     t, m elementwise f32 [N,K,HW], or [N,HW] broadcast over the classes (tper / mper = 1)
struct DiceCfg { int N, K, HW, act, multi, t_i64, tper, mper; float w[KMAX]; };

__device__ __forceinline__ void dice_pixel(const DiceCfg& c, const float* __restrict__ logits, const void* __restrict__ target,
                                           const float* __restrict__ mask, long n, long hw, long p, float* pk, float* tk, float* mk) {
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) if (k < c.K) { pk[k] = logits[(n * c.K + k) * c.HW + hw]; mx = fmaxf(mx, pk[k]); }
    if (c.act == 1) {
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < c.K) { pk[k] = expf(pk[k] - mx); se += pk[k]; }
        const float inv = 1.f / se;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < c.K) pk[k] *= inv;
    } else if (c.act == 2) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < c.K) {
                const float ea = expf(-fabsf(pk[k]));
                pk[k] = pk[k] >= 0.f ? 1.f / (1.f + ea) : ea / (1.f + ea);
            }
    }
    if (!c.multi) {
        const float tv = c.t_i64 ? (float)((const long long*)target)[p] : ((const float*)target)[p];
        const float m1 = mask ? (mask[p] == 1.f ? 1.f : 0.f) : 1.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) if (k < c.K) { tk[k] = tv == (float)k ? 1.f : 0.f; mk[k] = k == 0 ? 1.f : m1; }
    } else {
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < c.K) {
                tk[k] = c.tper ? ((const float*)target)[p] : ((const float*)target)[(n * c.K + k) * c.HW + hw];
                mk[k] = !mask ? 1.f : (c.mper ? mask[p] : mask[(n * c.K + k) * c.HW + hw]);
            }
    }
}

__global__ __launch_bounds__(256) void dice_fwd_kernel(const DiceCfg c, const float* __restrict__ logits, const void* __restrict__ target,
                                                      const float* __restrict__ mask, float* __restrict__ partials) {
    __shared__ float red[4][NS];
    float acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] = 0.f;
    const long npix = (long)c.N * c.HW;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        const long n = (unsigned)p / (unsigned)c.HW, hw = p - n * c.HW;
        float pk[KMAX], tk[KMAX], mk[KMAX];
        dice_pixel(c, logits, target, mask, n, hw, p, pk, tk, mk);
        // sum(target * target * mask) runs over the BROADCAST shape of target and mask only (losses.py:229): a per-pixel target with
        // a per-pixel (or no) mask is counted once, not once per class
        const bool yonce = c.multi && c.tper && (c.mper || !mask);
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < c.K) {
                acc[1 + k] += pk[k] * tk[k] * mk[k];
                acc[1 + KMAX + k] += pk[k] * pk[k] * mk[k];
                acc[1 + 2 * KMAX + k] += (yonce && k > 0) ? 0.f : tk[k] * tk[k] * mk[k];
            }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NS; ++i) { const float v = wave_sum(acc[i]); if (lane == 0) red[wave][i] = v; }
    __syncthreads();
    if (threadIdx.x < NS)
        partials[(long)blockIdx.x * NS + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[0] = loss; out[1+k] = I_k, out[1+K+k] = Z_k, out[1+2K+k] = Y_k (multi: the totals in k = 0, zeros behind)
__global__ void dice_finalize_kernel(const DiceCfg c, const float* __restrict__ partials, int rows, float* __restrict__ out) {
    __shared__ double tot[NS];
    if (threadIdx.x < NS) {
        double t = 0.0;
        for (int r = 0; r < rows; ++r) t += (double)partials[(long)r * NS + threadIdx.x];      // fixed order
        tot[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double loss = 0.0;
        if (c.multi) {
            double I = 0, Z = 0, Y = 0;
            for (int k = 0; k < c.K; ++k) { I += tot[1 + k]; Z += tot[1 + KMAX + k]; Y += tot[1 + 2 * KMAX + k]; }
            loss = 1.0 - (2.0 * I + (double)SMOOTH) / (Z + Y + (double)SMOOTH);
            for (int k = 0; k < c.K; ++k) { out[1 + k] = k ? 0.f : (float)I; out[1 + c.K + k] = k ? 0.f : (float)Z; out[1 + 2 * c.K + k] = k ? 0.f : (float)Y; }
        } else {
            for (int k = 0; k < c.K; ++k) {
                const double I = tot[1 + k], Z = tot[1 + KMAX + k], Y = tot[1 + 2 * KMAX + k];
                loss += (double)c.w[k] * (1.0 - (2.0 * I + (double)SMOOTH) / (Z + Y + (double)SMOOTH));
                out[1 + k] = (float)I; out[1 + c.K + k] = (float)Z; out[1 + 2 * c.K + k] = (float)Y;
            }
            loss /= c.K;
        }
        out[0] = (float)loss;
    }
}

__global__ __launch_bounds__(256) void dice_bwd_kernel(const DiceCfg c, const float* __restrict__ logits, const void* __restrict__ target,
                                                      const float* __restrict__ mask, const float* __restrict__ sums,
                                                      const float* __restrict__ gdev, float gscale, float* __restrict__ dlogits) {
    const float gs = gscale * (gdev ? gdev[0] : 1.f);
    float A[KMAX], Bc[KMAX];       // dLoss/dp_k = (A[k] t_k + Bc[k] p_k) m_k
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < c.K) {
            const int kk = c.multi ? 0 : k;
            const float I = sums[1 + kk], D = sums[1 + c.K + kk] + sums[1 + 2 * c.K + kk] + SMOOTH;
            const float wk = c.multi ? 1.f : c.w[k] / c.K;
            A[k] = -2.f / D * wk;
            Bc[k] = 2.f * (2.f * I + SMOOTH) / (D * D) * wk;
        }
    const long npix = (long)c.N * c.HW;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        const long n = (unsigned)p / (unsigned)c.HW, hw = p - n * c.HW;
        float pk[KMAX], tk[KMAX], mk[KMAX], G[KMAX], gp = 0.f;
        dice_pixel(c, logits, target, mask, n, hw, p, pk, tk, mk);
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < c.K) {
                G[k] = (A[k] * tk[k] + Bc[k] * pk[k]) * mk[k];
                gp += G[k] * pk[k];
            }
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
            if (k < c.K) {
                const float d = c.act == 1 ? pk[k] * (G[k] - gp) : c.act == 2 ? G[k] * pk[k] * (1.f - pk[k]) : G[k];
                dlogits[(n * c.K + k) * c.HW + hw] = gs * d;
            }
    }
}


}  // namespace
}  // namespace ustrun

using namespace ustrun;

extern "C" int64_t ustrun_loss_partials_bytes(int N, int K, int HW) {
    (void)N; (void)K; (void)HW;
    return (int64_t)2048 * NS * sizeof(float);
}

extern "C" int ustrun_seg_loss_fwd(const float* logits, const void* target, const float* mask, int N, int K, int HW,
                                   int mode, float* out, float* partials, int64_t partials_bytes, ustrun_stream_t s) {
    USTRUN_CHECK(logits && target && out && partials, "seg_loss_fwd: null pointer");
    USTRUN_CHECK(K >= 1 && K <= KMAX && N > 0 && HW > 0 && (long)N * HW < (1L << 32), "seg_loss_fwd: bad shape N=%d K=%d HW=%d", N, K, HW);
    USTRUN_CHECK(mode == USTRUN_LOSS_SOFTMAX || mode == USTRUN_LOSS_SIGMOID, "seg_loss_fwd: bad mode %d", mode);
    USTRUN_CHECK(partials_bytes >= ustrun_loss_partials_bytes(N, K, HW), "seg_loss_fwd: partials too small");
    const long items = mode == USTRUN_LOSS_SOFTMAX ? (long)N * HW : (long)N * K * HW;
    const int blocks = stream_blocks(items);
    hipLaunchKernelGGL(seg_loss_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, logits, target, mask, N, K, HW,
                       mode, partials);
    USTRUN_LAUNCH_CHECK("seg_loss_fwd");
    hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)s, partials, blocks, N, K, HW, mode, out);
    USTRUN_LAUNCH_CHECK("seg_loss_finalize");
    return 0;
}

extern "C" int ustrun_seg_loss_bwd(const float* logits, const void* target, const float* mask, int N, int K, int HW,
                                   int mode, const float* sums, const float* gscale_dev, float gscale, float ce_weight,
                                   float dice_weight, float* dlogits, ustrun_stream_t s) {
    USTRUN_CHECK(logits && target && sums && dlogits, "seg_loss_bwd: null pointer");
    USTRUN_CHECK(K >= 1 && K <= KMAX && N > 0 && HW > 0 && (long)N * HW < (1L << 32), "seg_loss_bwd: bad shape");
    USTRUN_CHECK(mode == USTRUN_LOSS_SOFTMAX || mode == USTRUN_LOSS_SIGMOID, "seg_loss_bwd: bad mode %d", mode);
    const long items = mode == USTRUN_LOSS_SOFTMAX ? (long)N * HW : (long)N * K * HW;
    hipLaunchKernelGGL(seg_loss_bwd_kernel, dim3(stream_blocks(items)), dim3(256), 0, (hipStream_t)s, logits, target,
                       mask, N, K, HW, mode, sums, gscale_dev, gscale, ce_weight, dice_weight, dlogits);
    USTRUN_LAUNCH_CHECK("seg_loss_bwd");
    return 0;
}


static int dice_cfg(DiceCfg* c, const char* who, const void* logits, const void* target, int N, int K, int HW, int act, int multi,
                    int target_is_i64, int target_per_pixel, int mask_per_pixel, const float* weight_host) {
    USTRUN_CHECK(logits && target, "%s: null pointer", who);
    USTRUN_CHECK(K >= 1 && K <= KMAX && N > 0 && HW > 0 && (long)N * HW < (1L << 32), "%s: bad shape N=%d K=%d HW=%d", who, N, K, HW);
    USTRUN_CHECK(act >= 0 && act <= 2, "%s: activation %d (0 none, 1 softmax, 2 sigmoid)", who, act);
    USTRUN_CHECK(multi || target_per_pixel, "%s: the per-class form takes a class-index target [N,HW]", who);
    USTRUN_CHECK(!multi || !target_is_i64, "%s: the multi form takes float targets", who);
    c->N = N; c->K = K; c->HW = HW; c->act = act; c->multi = multi; c->t_i64 = target_is_i64; c->tper = target_per_pixel; c->mper = mask_per_pixel;
    for (int k = 0; k < KMAX; ++k) c->w[k] = (weight_host && k < K) ? weight_host[k] : 1.f;
    return 0;
}

extern "C" int ustrun_dice_fwd(const float* logits, const void* target, int target_is_i64, int target_per_pixel, const float* mask,
                               int mask_per_pixel, int N, int K, int HW, int act, int multi, const float* weight_host, float* out,
                               float* partials, int64_t partials_bytes, ustrun_stream_t s) {
    DiceCfg c;
    USTRUN_TRY(dice_cfg(&c, "dice_fwd", logits, target, N, K, HW, act, multi, target_is_i64, target_per_pixel, mask_per_pixel, weight_host));
    USTRUN_CHECK(out && partials && partials_bytes >= ustrun_loss_partials_bytes(N, K, HW), "dice_fwd: partials too small");
    const int blocks = stream_blocks((long)N * HW);
    hipLaunchKernelGGL(dice_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, c, logits, target, mask, partials);
    USTRUN_LAUNCH_CHECK("dice_fwd");
    hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, c, partials, blocks, out);
    USTRUN_LAUNCH_CHECK("dice_finalize");
    return 0;
}

extern "C" int ustrun_dice_bwd(const float* logits, const void* target, int target_is_i64, int target_per_pixel, const float* mask,
                               int mask_per_pixel, int N, int K, int HW, int act, int multi, const float* weight_host, const float* sums,
                               const float* gscale_dev, float gscale, float* dlogits, ustrun_stream_t s) {
    DiceCfg c;
    USTRUN_TRY(dice_cfg(&c, "dice_bwd", logits, target, N, K, HW, act, multi, target_is_i64, target_per_pixel, mask_per_pixel, weight_host));
    USTRUN_CHECK(sums && dlogits, "dice_bwd: null pointer");
    hipLaunchKernelGGL(dice_bwd_kernel, dim3(stream_blocks((long)N * HW)), dim3(256), 0, (hipStream_t)s, c, logits, target, mask, sums,
                       gscale_dev, gscale, dlogits);
    USTRUN_LAUNCH_CHECK("dice_bwd");
    return 0;
}

extern "C" int ustrun_pseudo_label(const float* logits, int N, int K, int HW, float threshold, int mode, void* label,
                                   float* mask, ustrun_stream_t s) {
    USTRUN_CHECK(logits && label && mask, "pseudo_label: null pointer");
    USTRUN_CHECK(K >= 1 && K <= KMAX && N > 0 && HW > 0 && (long)N * HW < (1L << 32), "pseudo_label: bad shape");
    const long items = mode == USTRUN_LOSS_SOFTMAX ? (long)N * HW : (long)N * K * HW;
    hipLaunchKernelGGL(pseudo_label_kernel, dim3(stream_blocks(items)), dim3(256), 0, (hipStream_t)s, logits, N, K, HW,
                       threshold, mode, label, mask);
    USTRUN_LAUNCH_CHECK("pseudo_label");
    return 0;
}

extern "C" int ustrun_mix_targets(int mode, int N, int K, int HW, const float* box, const void* pl, const float* mask,
                                  const void* pl_w_ul, const float* mask_w_ul, const void* pl_w_lu,
                                  const float* mask_w_lu, const void* cut_label, const float* cut_mask, void* pl_w,
                                  float* mask_w, void* pl_ul, float* mask_ul, void* pl_lu, float* mask_lu,
                                  ustrun_stream_t s) {
    USTRUN_CHECK(box && pl && mask && pl_w_ul && mask_w_ul && pl_w_lu && mask_w_lu && cut_label && cut_mask && pl_w &&
                     mask_w && pl_ul && mask_ul && pl_lu && mask_lu, "mix_targets: null pointer");
    const long items = (long)N * (mode == USTRUN_LOSS_SOFTMAX ? 1 : K) * HW;
    hipLaunchKernelGGL(mix_targets_kernel, dim3(stream_blocks(items)), dim3(256), 0, (hipStream_t)s, mode, N, K, HW, box,
                       pl, mask, pl_w_ul, mask_w_ul, pl_w_lu, mask_w_lu, cut_label, cut_mask, pl_w, mask_w, pl_ul,
                       mask_ul, pl_lu, mask_lu);
    USTRUN_LAUNCH_CHECK("mix_targets");
    return 0;
}

extern "C" int ustrun_box_mix(const float* a, const float* b, const float* box, int N, int C, int HW, float* out,
                              ustrun_stream_t s) {
    USTRUN_CHECK(a && b && box && out && N > 0 && C > 0 && HW > 0, "box_mix: bad args");
    hipLaunchKernelGGL(box_mix_kernel, dim3(stream_blocks((long)N * C * HW)), dim3(256), 0, (hipStream_t)s, a, b, box, N,
                       C, HW, out);
    USTRUN_LAUNCH_CHECK("box_mix");
    return 0;
}

namespace ustrun { namespace {
struct AsmArgs { ustrun_asm_row_t r[USTRUN_ASM_MAX]; };
// blockIdx.y = output row; 16 bytes per thread and trip
__global__ __launch_bounds__(256) void assemble_kernel(const AsmArgs t, long row16, int HW, float* __restrict__ out) {
    const ustrun_asm_row_t r = t.r[blockIdx.y];
    f32x4* o = (f32x4*)out + (long)blockIdx.y * row16;
    const f32x4* a = (const f32x4*)r.a;
    if (!r.b) {
        for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < row16; e += (long)gridDim.x * 256) o[e] = a[e];
        return;
    }
    const f32x4* b = (const f32x4*)r.b;
    const f32x4* bx = (const f32x4*)r.box;
    const int hw4 = HW / 4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < row16; e += (long)gridDim.x * 256) {
        const f32x4 va = a[e], vb = b[e], x = bx[e % hw4];
        f32x4 v;
        v.x = va.x * (1.f - x.x) + vb.x * x.x; v.y = va.y * (1.f - x.y) + vb.y * x.y;       // (box_mix_kernel's expression)
        v.z = va.z * (1.f - x.z) + vb.z * x.z; v.w = va.w * (1.f - x.w) + vb.w * x.w;
        o[e] = v;
    }
}

__global__ __launch_bounds__(256) void decode_labels_kernel(const float* __restrict__ y, int kind, long npix, int HW, void* out) {
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
        if (kind == 0) {
            const long n = p / HW, hw = p - n * HW;
            const float v = y[p];
            float* o = (float*)out + n * 2 * HW + hw;
            o[0] = v == 0.f ? 1.f : 0.f; o[HW] = v <= 128.f ? 1.f : 0.f;
        } else if (kind == 1) ((long long*)out)[p] = y[p] == 0.f ? 1 : 0;
        else if (kind == 2) ((long long*)out)[p] = y[p] == 255.f ? 1 : 0;
        else {
            const float a = y[3 * p], b = y[3 * p + 1], c = y[3 * p + 2];
            ((long long*)out)[p] = c == 255.f ? 3 : (b == 255.f ? 2 : (a == 255.f ? 1 : 0));
        }
    }
}

struct BboxArgs { const void* p[4]; };
__global__ __launch_bounds__(256) void region_bbox_kernel(const BboxArgs a, int np, int i64bits, int H, int W, int* __restrict__ partial) {
    __shared__ int red[4][4];
    int y0 = H, y1 = -1, x0 = W, x1 = -1;
    const int HW = H * W;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < HW; e += gridDim.x * 256) {
        bool nz = false;
        for (int k = 0; k < np; ++k)
            nz |= ((i64bits >> k) & 1) ? ((const long long*)a.p[k])[e] != 0 : ((const float*)a.p[k])[e] != 0.f;
        if (nz) {
            const int y = e / W, x = e - y * W;
            y0 = min(y0, y); y1 = max(y1, y); x0 = min(x0, x); x1 = max(x1, x);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        y0 = min(y0, __shfl_xor(y0, o)); y1 = max(y1, __shfl_xor(y1, o));
        x0 = min(x0, __shfl_xor(x0, o)); x1 = max(x1, __shfl_xor(x1, o));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w][0] = y0; red[w][1] = y1; red[w][2] = x0; red[w][3] = x1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { y0 = min(y0, red[k][0]); y1 = max(y1, red[k][1]); x0 = min(x0, red[k][2]); x1 = max(x1, red[k][3]); }
        int* o = partial + blockIdx.x * 4;
        o[0] = y0; o[1] = y1; o[2] = x0; o[3] = x1;
    }
}
} }
extern "C" int ustrun_assemble(const ustrun_asm_row_t* rows_host, int nrows, int64_t row_bytes, int HW, void* out, ustrun_stream_t s) {
    USTRUN_CHECK(rows_host && out && nrows > 0 && nrows <= USTRUN_ASM_MAX, "assemble: 1..USTRUN_ASM_MAX rows per call (%d)", nrows);
    USTRUN_CHECK(row_bytes > 0 && row_bytes % 16 == 0 && ((uintptr_t)out & 15) == 0, "assemble: rows of %ld bytes (16-byte units)", (long)row_bytes);
    AsmArgs t;
    memset(&t, 0, sizeof t);
    for (int r = 0; r < nrows; ++r) {
        const ustrun_asm_row_t& q = rows_host[r];
        USTRUN_CHECK(q.a && ((uintptr_t)q.a & 15) == 0 && ((uintptr_t)q.b & 15) == 0 && ((uintptr_t)q.box & 15) == 0, "assemble: row %d pointers", r);
        USTRUN_CHECK(!q.b || (q.box && HW > 0 && HW % 4 == 0 && (row_bytes / 4) % HW == 0), "assemble: row %d mixes with HW=%d", r, HW);
        t.r[r] = q;
    }
    long bx = (row_bytes / 16 + 256 * 4 - 1) / (256 * 4);
    const long cap = (2048 + nrows - 1) / nrows;
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(assemble_kernel, dim3((int)bx, nrows), dim3(256), 0, (hipStream_t)s, t, (long)(row_bytes / 16), HW, (float*)out);
    USTRUN_LAUNCH_CHECK("assemble");
    return 0;
}
extern "C" int ustrun_decode_labels(const float* y, int kind, int N, int HW, void* out, ustrun_stream_t s) {
    USTRUN_CHECK(y && out && N > 0 && HW > 0 && kind >= 0 && kind <= 3, "decode_labels: bad args");
    const long npix = (long)N * HW;
    hipLaunchKernelGGL(decode_labels_kernel, dim3(stream_blocks(npix)), dim3(256), 0, (hipStream_t)s, y, kind, npix, HW, out);
    USTRUN_LAUNCH_CHECK("decode_labels");
    return 0;
}
extern "C" int ustrun_region_bbox(const void* const* planes_host, int nplanes, int is_i64_bits, int H, int W, int32_t* partial,
                                  ustrun_stream_t s) {
    USTRUN_CHECK(planes_host && partial && nplanes >= 1 && nplanes <= 4 && H > 0 && W > 0 && (long)H * W < (1L << 30), "region_bbox: bad args");
    BboxArgs a = {};
    for (int k = 0; k < nplanes; ++k) { USTRUN_CHECK(planes_host[k], "region_bbox: plane %d missing", k); a.p[k] = planes_host[k]; }
    hipLaunchKernelGGL(region_bbox_kernel, dim3(USTRUN_BBOX_BLOCKS), dim3(256), 0, (hipStream_t)s, a, nplanes, is_i64_bits, H, W, (int*)partial);
    USTRUN_LAUNCH_CHECK("region_bbox");
    return 0;
}

/* Small host values reach the device inside the kernel-argument block: no copy engine, no host wait, ordered on the
 * stream like any launch.  (A pinned hipMemcpyAsync on a busy stream cost the step 20 ms here; a pageable copy drains
 * the queue.) */
namespace ustrun { namespace {
struct RectArgs { int r[USTRUN_MAX_RECTS][4]; };
__global__ void rect_masks_kernel(RectArgs a, int N, int H, int W, float* __restrict__ box) {
    const long total = (long)N * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / ((long)H * W));
        const int rem = (int)(i - (long)n * H * W);
        const int y = rem / W, x = rem - y * W;
        box[i] = (y >= a.r[n][0] && y < a.r[n][1] && x >= a.r[n][2] && x < a.r[n][3]) ? 1.f : 0.f;
    }
}
} }
extern "C" int ustrun_rect_masks(const int32_t* rects_host, int N, int H, int W, float* box, ustrun_stream_t s) {
    USTRUN_CHECK(rects_host && box && H > 0 && W > 0, "rect_masks: bad args");
    USTRUN_CHECK(N > 0 && N <= USTRUN_MAX_RECTS, "rect_masks: 1..USTRUN_MAX_RECTS rectangles per call");
    RectArgs a;
    memset(&a, 0, sizeof a);
    memcpy(a.r, rects_host, sizeof(int) * 4 * N);
    hipLaunchKernelGGL(rect_masks_kernel, dim3(stream_blocks((long)N * H * W)), dim3(256), 0, (hipStream_t)s, a, N, H, W,
                       box);
    USTRUN_LAUNCH_CHECK("rect_masks");
    return 0;
}

namespace ustrun { namespace {
struct SmallArgs { uint32_t w[USTRUN_UPLOAD_MAX / 4]; };
__global__ void upload_small_kernel(SmallArgs a, int words, uint32_t* __restrict__ dst) {
    for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = a.w[i];
}
} }
extern "C" int ustrun_upload_small(void* dst, const void* src_host, int nbytes, ustrun_stream_t s) {
    USTRUN_CHECK(dst && src_host, "upload_small: null pointer");
    USTRUN_CHECK(nbytes > 0 && nbytes <= USTRUN_UPLOAD_MAX && nbytes % 4 == 0 && ((uintptr_t)dst & 3) == 0,
                 "upload_small: 4..USTRUN_UPLOAD_MAX bytes, a multiple of 4, to a 4-byte aligned address");
    SmallArgs a;
    memcpy(a.w, src_host, nbytes);
    hipLaunchKernelGGL(upload_small_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, a, nbytes / 4, (uint32_t*)dst);
    USTRUN_LAUNCH_CHECK("upload_small");
    return 0;
}

extern "C" int ustrun_dice_counts(const void* pred, const void* gt, int pred_is_i64, int gt_is_i64, int N, int K,
                                  int HW, int by_class, int32_t* counts, ustrun_stream_t s) {
    USTRUN_CHECK(pred && gt && counts && N > 0 && K > 0 && HW > 0, "dice_counts: bad args");
    // one block per (sample, class) left a 16+16 batch on 32 of 256 CUs (0.2 ms): split the pixels over enough blocks
    int split = 1;
    while ((long)N * K * split < 512 && HW / (split * 2) >= 2048) split *= 2;
    if (split > 1) {
        const hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * 3 * (size_t)N * K, (hipStream_t)s);
        USTRUN_CHECK(e == hipSuccess, "dice_counts: memset failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(dice_counts_kernel, dim3(N * K, split), dim3(256), 0, (hipStream_t)s, pred, gt, pred_is_i64, gt_is_i64, K,
                       HW, by_class, counts);
    USTRUN_LAUNCH_CHECK("dice_counts");
    return 0;
}

extern "C" int ustrun_sgd_ema(float* p, const float* g, float* v, float* t, int64_t n, float lr, float mu, float wd,
                              int first, float alpha, float grad_scale, ustrun_stream_t s) {
    USTRUN_CHECK(p && g && v && n > 0, "sgd_ema: bad args");
    USTRUN_CHECK(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                     (!t || (uintptr_t)t % 16 == 0), "sgd_ema: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(sgd_ema_kernel, dim3(stream_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)s, p, g, v, t, (long)n,
                       lr, mu, wd, first, alpha, grad_scale);
    USTRUN_LAUNCH_CHECK("sgd_ema");
    return 0;
}

extern "C" int ustrun_amp_check(const float* g, int64_t n, float* amp_state, ustrun_stream_t s) {
    USTRUN_CHECK(g && amp_state && n > 0 && (uintptr_t)g % 16 == 0, "amp_check: bad args");
    hipLaunchKernelGGL(amp_check_kernel, dim3(stream_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)s, g, (long)n, amp_state);
    USTRUN_LAUNCH_CHECK("amp_check");
    return 0;
}

extern "C" int ustrun_sgd_ema_scaled(float* p, const float* g, float* v, float* t, int64_t n, float lr, float mu, float wd,
                                     int first, float alpha, float grad_scale, const float* amp_state, ustrun_stream_t s) {
    USTRUN_CHECK(p && g && v && amp_state && n > 0, "sgd_ema_scaled: bad args");
    USTRUN_CHECK(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)v % 16 == 0) &&
                     (!t || (uintptr_t)t % 16 == 0), "sgd_ema_scaled: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(sgd_ema_scaled_kernel, dim3(stream_blocks(n / 4 + 1)), dim3(256), 0, (hipStream_t)s, p, g, v, t, (long)n,
                       lr, mu, wd, first, alpha, grad_scale, amp_state);
    USTRUN_LAUNCH_CHECK("sgd_ema_scaled");
    return 0;
}

extern "C" int ustrun_amp_update(float* amp_state, float growth_factor, float backoff_factor, int growth_interval,
                                 ustrun_stream_t s) {
    USTRUN_CHECK(amp_state && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval >= 1,
                 "amp_update: bad args");
    hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, amp_state, growth_factor, backoff_factor,
                       growth_interval);
    USTRUN_LAUNCH_CHECK("amp_update");
    return 0;
}
