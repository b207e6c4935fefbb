// bn.hip -- train-mode BatchNorm2d statistics, and the BatchNorm+ReLU(+MaxPool) backward passes.
// All kernels here are HBM-bound streams over NHWC activations (4 channels = 16 B per lane) with
// fixed-order two-level reductions (block partials, then an f64 finalize) so results are
// reproducible run to run.
#include "common.h"
#include "loader.h"

namespace ustrun {
namespace {

// ---- forward statistics -------------------------------------------------------------------
// block = 32 channels x 8 row lanes; stat[row][2][C] -> mean, biased var -> scale/shift
// Stage 1 for long statistics tables: block (cb, sp) sums its row range [sp*R, sp*R+R) of its 32 columns
// in f64 and parks the two sums IN PLACE, each as a (hi, lo) float pair, in its OWN cells: row sp*R holds
// the sum (q=0 cell: hi, q=1 cell: lo), row sp*R+1 the sum of squares.  Only this block reads those cells,
// and it does so before the barrier that precedes the writes.
// A shorter TAIL pass may follow the `passes` equal ones (BnTail: its rows start where theirs end; R = 0: too few rows for a first
// stage, the finalize sums them directly -- each pass goes through exactly the arithmetic a call of its own would).
__global__ __launch_bounds__(256) void bn_stat_stage1_kernel(float* stat, int rows, int C, int R, int passes, BnTail tail) {
    __shared__ double red[2][8][32];
    stat += (long)blockIdx.z * rows * 2 * C;            // blockIdx.z = forward pass (its own rows)
    if ((int)blockIdx.z == passes) { rows = tail.rows; R = tail.R; }      // the tail pass: its own rows and split
    if ((int)blockIdx.y * R >= rows) return;            // (grid.y covers the larger of the two split counts)
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl, r0 = blockIdx.y * R, r1 = min(rows, r0 + R);
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        int r = r0 + rg;
        for (; r + 24 < r1; r += 32) {               // four rows' loads issued before the first add, sums in ascending row order
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = stat[((long)(r + 8 * u) * 2 + 0) * C + c]; b[u] = stat[((long)(r + 8 * u) * 2 + 1) * C + c]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s1 += (double)a[u]; s2 += (double)b[u]; }
        }
        for (; r < r1; r += 8) {
            s1 += (double)stat[((long)r * 2 + 0) * C + c];
            s2 += (double)stat[((long)r * 2 + 1) * C + c];
        }
    }
    red[0][rg][cl] = s1; red[1][rg][cl] = s2;
    __syncthreads();
    if (rg < 2 && c < C) {                   // rg 0 -> sum into row r0, rg 1 -> sum of squares into row r0+1
        double v = 0.0;
        for (int k = 0; k < 8; ++k) v += red[rg][k][cl];
        const float hi = (float)v, lo = (float)(v - (double)hi);
        stat[((long)(r0 + rg) * 2 + 0) * C + c] = hi;
        stat[((long)(r0 + rg) * 2 + 1) * C + c] = lo;
    }
}

// Both stages in ONE launch (the U-Net forward: 52 x 2 launches of ~7.5 us per step were launch latency, not work): block
// (cb, sp, g) does stage 1 for its row range, then takes a ticket for its channel block; the block that draws the last
// ticket -- every stage-1 sum of that channel block is then parked and visible (agent-scope release by the writers before
// the ticket, acquire by the reader after it) -- finalizes the channel block's passes one after the other with the very
// summation order of bn_finalize_kernel<true> (eight row lanes per pass, lanes added in ascending order), so the constants
// are bit-identical to the two-launch form.  tickets[cb] returns to 0 for the next launch on the same stream.
__global__ __launch_bounds__(256) void bn_stat_fused_kernel(float* stat, int rows, int C, double count, int R,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* rm, float* rv, int64_t* nbt, float momentum, float eps,
                                                            int update, float* scale, float* shift, float* mean, float* rstd,
                                                            int passes, long astride, unsigned* tickets) {
    constexpr int PPR = 4;                                 // passes finalized per round (their loads in flight together)
    __shared__ double red[2][PPR][8][32];
    __shared__ bool last;
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    {
        float* st = stat + (long)blockIdx.z * rows * 2 * C;
        const int r0 = blockIdx.y * R, r1 = min(rows, r0 + R);
        double s1 = 0.0, s2 = 0.0;
        for (int r = r0 + rg; r < r1; r += 8) {
            s1 += (double)st[((long)r * 2 + 0) * C + c];
            s2 += (double)st[((long)r * 2 + 1) * C + c];
        }
        red[0][0][rg][cl] = s1; red[1][0][rg][cl] = s2;
        __syncthreads();
        if (rg < 2) {
            double v = 0.0;
            for (int k = 0; k < 8; ++k) v += red[rg][0][k][cl];
            const float hi = (float)v, lo = (float)(v - (double)hi);
            // write-through (sc1) stores, drained by every storing wave before the barrier in front of the ticket: no agent
            // release -- that is a write-back of the XCD's L2 (MI355X_MICROARCH.md, fence table), which the convolution has
            // just filled with dirty output lines: with it this kernel cost the step 0.6 ms MORE than the two launches
            __hip_atomic_store(st + ((long)(r0 + rg) * 2 + 0) * C + c, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(st + ((long)(r0 + rg) * 2 + 1) * C + c, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const unsigned total = gridDim.y * gridDim.z;
        const unsigned t = __hip_atomic_fetch_add(tickets + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = t == total - 1;
    }
    __syncthreads();
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");        // (the other blocks' parked sums: not from this CU's L1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          //  and every load of them below is an sc1 load as well)
    __syncthreads();
    double run_m = 0.0, run_v = 0.0;
    if (update && rg == 0) { run_m = (double)rm[c]; run_v = (double)rv[c]; }
    // a row lane holds the splits rg, rg + 8, .. (at most four: the launcher splits a table into <= 32 ranges) of up to PPR
    // passes: every load of the round is issued before the first sum
    for (int g0 = 0; g0 < passes; g0 += PPR) {
        float v[PPR][4][4];
#pragma unroll
        for (int q = 0; q < PPR; ++q) {
            const float* st = stat + (long)(g0 + q < passes ? g0 + q : 0) * rows * 2 * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sp = rg + 8 * k;
                const bool ok = g0 + q < passes && sp * R < rows;
                const long a0 = (long)(sp * R) * 2 * C + c, a1 = (long)(sp * R + 1) * 2 * C + c;
                const long z = ok ? 0 : -1;               // (absent: the block's own first cell, value unused)
                const float x0 = __hip_atomic_load(st + (a0 & ~z) , __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float x1 = __hip_atomic_load(st + ((a0 + C) & ~z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float x2 = __hip_atomic_load(st + (a1 & ~z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float x3 = __hip_atomic_load(st + ((a1 + C) & ~z), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v[q][k][0] = ok ? x0 : 0.f; v[q][k][1] = ok ? x1 : 0.f;
                v[q][k][2] = ok ? x2 : 0.f; v[q][k][3] = ok ? x3 : 0.f;
            }
        }
        __syncthreads();                                       // (red: the previous round's sums are consumed)
#pragma unroll
        for (int q = 0; q < PPR; ++q) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {                      // (x + 0.0 leaves x: absent splits do not change the sums)
                s1 += (double)v[q][k][0] + (double)v[q][k][1];
                s2 += (double)v[q][k][2] + (double)v[q][k][3];
            }
            red[0][q][rg][cl] = s1; red[1][q][rg][cl] = s2;
        }
        __syncthreads();
        if (rg == 0) {
            for (int q = 0; q < PPR && g0 + q < passes; ++q) {
                double t1 = 0.0, t2 = 0.0;
                for (int k = 0; k < 8; ++k) { t1 += red[0][q][k][cl]; t2 += red[1][q][k][cl]; }
                const double m = t1 / count;
                double var = t2 / count - m * m;
                if (var < 0.0) var = 0.0;
                const double rs = 1.0 / sqrt(var + (double)eps);
                const long o = (long)(g0 + q) * astride + c;
                scale[o] = (float)((double)gamma[c] * rs);
                shift[o] = (float)((double)beta[c] - m * (double)gamma[c] * rs);
                mean[o] = (float)m; rstd[o] = (float)rs;
                if (update) {
                    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                    run_m = (double)(float)((1.0 - momentum) * run_m + momentum * m);
                    run_v = (double)(float)((1.0 - momentum) * run_v + momentum * unb);
                }
            }
        }
    }
    if (update && rg == 0) { rm[c] = (float)run_m; rv[c] = (float)run_v; }
    if (update && nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += passes;
    if (threadIdx.x == 0) __hip_atomic_store(tickets + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// `passes` forward passes batched into one launch: pass g owns rows [g*rows, (g+1)*rows) of stat and the constants at
// scale/shift/mean/rstd + g*astride; the running buffers receive the passes' updates one after the other, exactly as
// `passes` separate calls would apply them.
template <bool PRE>
__global__ void bn_finalize_kernel(const float* __restrict__ stat, int rows, int C, double count, int R,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* rm, float* rv, int64_t* nbt, float momentum, float eps, int update,
                                   float* scale, float* shift, float* mean, float* rstd, int passes, long astride, BnTail tail) {
    __shared__ double red[2][32][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;      // 32 channels x 32 row lanes
    const int full = passes;                                     // the equal passes; then the tail, if any
    const int rows_f = rows, R_f = R;
    const double count_f = count;
    if (tail.rows > 0) ++passes;
    const int c = blockIdx.x * 32 + cl;
    double run_m = 0.0, run_v = 0.0;
    if (update && rg == 0 && c < C) { run_m = (double)rm[c]; run_v = (double)rv[c]; }
    // The 32 row lanes work on up to four passes at once, eight lanes per pass (the passes' row sums are independent;
    // only the running-buffer update is sequential): one round of loads instead of one per pass.  A pass always gets
    // eight lanes, so its sums are formed in the same order whether it comes alone or batched with others.
    constexpr int LPP = 8, PPR = 32 / LPP;                        // lanes per pass, passes per round
    const int gl = rg / LPP, lane = rg % LPP;
    for (int g0 = 0; g0 < passes; g0 += PPR) {
        const int g = g0 + gl;
        double s1 = 0.0, s2 = 0.0;
        if (c < C && g < passes) {
            const float* st = stat + (long)g * rows_f * 2 * C;
            rows = g < full ? rows_f : tail.rows; R = g < full ? R_f : tail.R;
            if (PRE && R > 0) {       // stage-1 doubles parked at rows sp*R / sp*R+1, columns of this channel block
                // at most 32 splits (the launcher's R), i.e. four per lane: all sixteen loads issued before the first sum (one
                // split per trip was four dependent memory round trips in a 7 us kernel); x + 0.0 leaves x, so the absent
                // splits' zeros do not change the sums
                float v[4][4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int sp = lane + LPP * k;
                    const bool ok = sp * R < rows;
                    const long a0 = ok ? (long)(sp * R) * 2 * C + c : c, a1 = ok ? (long)(sp * R + 1) * 2 * C + c : c;
                    const float x0 = st[a0], x1 = st[a0 + C], x2 = st[a1], x3 = st[a1 + C];
                    v[k][0] = ok ? x0 : 0.f; v[k][1] = ok ? x1 : 0.f; v[k][2] = ok ? x2 : 0.f; v[k][3] = ok ? x3 : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s1 += (double)v[k][0] + (double)v[k][1];
                    s2 += (double)v[k][2] + (double)v[k][3];
                }
                for (int sp = lane + 4 * LPP; sp * R < rows; sp += LPP) {       // (a table split finer than the launcher does)
                    const long a0 = (long)(sp * R) * 2 * C + c, a1 = (long)(sp * R + 1) * 2 * C + c;
                    s1 += (double)st[a0] + (double)st[a0 + C];
                    s2 += (double)st[a1] + (double)st[a1 + C];
                }
            } else {
#pragma unroll 8
                for (int r = lane; r < rows; r += LPP) {        // (unrolled: the loads of eight rows in flight together)
                    s1 += (double)st[((long)r * 2 + 0) * C + c];
                    s2 += (double)st[((long)r * 2 + 1) * C + c];
                }
            }
        }
        red[0][rg][cl] = s1; red[1][rg][cl] = s2;
        __syncthreads();
        if (rg == 0 && c < C) {
            for (int q = 0; q < PPR && g0 + q < passes; ++q) {     // the round's passes, in order
                double t1 = 0.0, t2 = 0.0;
                for (int k = 0; k < LPP; ++k) { t1 += red[0][q * LPP + k][cl]; t2 += red[1][q * LPP + k][cl]; }
                count = g0 + q < full ? count_f : tail.count;
                const double m = t1 / count;
                double var = t2 / count - m * m;
                if (var < 0.0) var = 0.0;
                const double rs = 1.0 / sqrt(var + (double)eps);
                const long o = (long)(g0 + q) * astride + c;
                scale[o] = (float)((double)gamma[c] * rs);
                shift[o] = (float)((double)beta[c] - m * (double)gamma[c] * rs);
                mean[o] = (float)m; rstd[o] = (float)rs;
                if (update) {    // through float after every pass, as the stored buffer of separate calls would be
                    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
                    run_m = (double)(float)((1.0 - momentum) * run_m + momentum * m);
                    run_v = (double)(float)((1.0 - momentum) * run_v + momentum * unb);
                }
            }
        }
        __syncthreads();
    }
    if (update && rg == 0 && c < C) { rm[c] = (float)run_m; rv[c] = (float)run_v; }
    if (update && nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += passes;
}

__global__ void bn_eval_affine_kernel(int C, const float* gamma, const float* beta, const float* rm,
                                      const float* rv, float eps, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = gamma[c] * (1.0f / sqrtf(rv[c] + eps));
        scale[c] = sc; shift[c] = beta[c] - rm[c] * sc;
    }
}

// eval mode: the 18 layers' constants in ONE launch (blockIdx.y = layer x pass), instead of 18 five-microsecond ones
struct EvalAffineJobs {
    const float *gamma[18], *beta[18], *rm[18], *rv[18];
    float* aff[18];          // layer's table: pass g at aff + g * 4C: {scale[C], shift[C], ...}
    int C[18];
};
__global__ void bn_eval_affine_multi_kernel(const EvalAffineJobs j, float eps, int passes) {
    const int l = blockIdx.y / passes, g = blockIdx.y % passes, C = j.C[l];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        const float sc = j.gamma[l][c] * (1.0f / sqrtf(j.rv[l][c] + eps));
        float* a = j.aff[l] + 4L * C * g;
        a[c] = sc; a[C + c] = j.beta[l][c] - j.rm[l][c] * sc;
    }
}

// a = relu(y*scale+shift) materialised (NHWC -> NHWC or NCHW)
__global__ void bn_relu_apply_kernel(const float* __restrict__ y, int esz, const float* __restrict__ scale,
                                     const float* __restrict__ shift, long npix, int C, int HW,
                                     float* __restrict__ out, int nchw) {
    const long total = npix * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long p = e / C; const int c = (int)(e - p * C);
        float v = ld1(y, e, esz);
        if (scale) v = fmaxf(v * scale[c] + shift[c], 0.f);
        if (nchw) { const long n = p / HW; const long hw = p - n * HW; out[(n * C + c) * HW + hw] = v; }
        else out[e] = v;
    }
}

// a = relu(y*scale+shift) in the 16-bit storage type, NHWC -> NHWC, per-pass constants: the operand of a DoubleConv's second
// convolution written out once (8 channels = 16 bytes per lane, four pixels in flight per thread) where applying BatchNorm + ReLU
// per staged item costs that convolution and its weight gradient more than this pass does (unet.hip: the levels from 256 channels)
__global__ __launch_bounds__(256) void act16_kernel(const elt_t* __restrict__ y, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, int relu, int gN, long gstride,
                                                    long npix, int HW, int C, elt_t* __restrict__ out) {
    typedef __attribute__((ext_vector_type(8))) elt_t bf16x8v;
    const int CV = C / 8, PPB = 256 / CV;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    if (pl >= PPB) return;
    constexpr int U = 4;
    const long stride = (long)gridDim.x * PPB;
    for (long p0 = (long)blockIdx.x * PPB + pl; p0 < npix; p0 += U * stride) {
        bf16x8v r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long p = p0 + u * stride; r[u] = *(const bf16x8v*)(y + (p < npix ? p : npix - 1) * C + cv * 8); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long p = p0 + u * stride;
            if (p >= npix) break;
            const int grp = gN > 0 ? (int)((unsigned)((unsigned)p / (unsigned)HW) / (unsigned)gN) : 0;      // (pixels < 2^32: host check)
            const float* ps = scale + grp * gstride + cv * 8;
            const float* pb = shift + grp * gstride + cv * 8;
            const f32x4 s0 = *(const f32x4*)ps, s1 = *(const f32x4*)(ps + 4), b0 = *(const f32x4*)pb, b1 = *(const f32x4*)(pb + 4);
            bf16x8v o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = (float)r[u][j] * (j < 4 ? s0[j & 3] : s1[j & 3]) + (j < 4 ? b0[j & 3] : b1[j & 3]);
                a = relu ? fmaxf(a, 0.f) : a;
                o[j] = (elt_t)a;
            }
            *(bf16x8v*)(out + p * C + cv * 8) = o;
        }
    }
}

// pooled activation p[n][y][x][c] = max over the 2x2 window of relu(s*y + b): the Down block's MaxPool2d input,
// materialised once (a quarter of y) so that the Down convolution and its weight gradient read a plain tensor by
// LDS-DMA instead of pooling four pixels per staged item in an issue-bound kernel.  8 channels (ESZ = 2) or 4
// (ESZ = 4) per thread; batched passes pick their constants per image.
template <int ESZ>
__global__ __launch_bounds__(256) void pool_act_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, int relu, int gN, long gstride,
                                                      int N, int H, int W, int C, float* __restrict__ out,
                                                      elt_t* __restrict__ act) {
    constexpr int V = ESZ == 2 ? 8 : 4;             // channels per thread: one 16-byte load per window pixel
    typedef __attribute__((ext_vector_type(8))) elt_t bf16x8v;
    const int Hp = H / 2, Wp = W / 2, CV = C / V;
    // a thread keeps its channel group (256 % CV == 0 for the power-of-two channel counts; else it re-derives it)
    const long npix = (long)N * Hp * Wp;
    const int PPB = 256 / CV > 0 ? 256 / CV : 1;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    if (pl >= PPB) return;
    float sc[V], sh[V];
    // the constants are (re)loaded every pixel, unconditionally (L1 hits): a "reload when the pass changes" branch kept the
    // compiler from overlapping one pixel's loads with the next one's
#pragma unroll 2
    for (long p = (long)blockIdx.x * PPB + pl; p < npix; p += (long)gridDim.x * PPB) {
        const unsigned up = (unsigned)p, t = up / (unsigned)Wp;      // 32-bit divides (pooled pixels < 2^32: host check)
        const int px = (int)(up - t * (unsigned)Wp), n = (int)(t / (unsigned)Hp), py = (int)(t - (unsigned)n * (unsigned)Hp);
        const int grp = gN > 0 ? n / gN : 0;
#pragma unroll
        for (int h = 0; h < V / 4; ++h) {
            const f32x4 s4 = scale ? *(const f32x4*)(scale + grp * gstride + cv * V + 4 * h) : (f32x4){1.f, 1.f, 1.f, 1.f};
            const f32x4 b4 = scale ? *(const f32x4*)(shift + grp * gstride + cv * V + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) { sc[4 * h + k] = s4[k]; sh[4 * h + k] = b4[k]; }
        }
        const long base = (((long)n * H + 2 * py) * W + 2 * px) * C + cv * V;
        float v[4][V];
        if (ESZ == 2) {
            bf16x8v r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) r[q] = *(const bf16x8v*)((const elt_t*)y + base + ((q >> 1) * (long)W + (q & 1)) * C);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < V; ++j) v[q][j] = (float)r[q][j % 8];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 r = *(const f32x4*)(y + base + ((q >> 1) * (long)W + (q & 1)) * C);
#pragma unroll
                for (int j = 0; j < V; ++j) v[q][j] = r[j % 4];
            }
        }
        float m[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float a = v[q][j] * sc[j] + sh[j];
                a = relu ? fmaxf(a, 0.f) : a;
                v[q][j] = a;
                m[j] = q == 0 ? a : fmaxf(m[j], a);
            }
        }
        if (ESZ == 2 && act) {                      // the window's four activated pixels, as the concat convolution will read them
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x8v a8;
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[j] = (elt_t)v[q][j % V];
                *(bf16x8v*)(act + base + ((q >> 1) * (long)W + (q & 1)) * C) = a8;
            }
        }
        const long ob = (((long)n * Hp + py) * Wp + px) * C + cv * V;
        if (ESZ == 2) {                                     // one 16-byte store
            bf16x8v o8;
#pragma unroll
            for (int j = 0; j < 8; ++j) o8[j] = (elt_t)m[j % V];
            *(bf16x8v*)((elt_t*)out + ob) = o8;
        } else {
#pragma unroll
            for (int h = 0; h < V / 4; ++h) st4t<ESZ>(out, ob + 4 * h, (f32x4){m[4 * h], m[4 * h + 1], m[4 * h + 2], m[4 * h + 3]});
        }
    }
}

// plain MaxPool2d(2) backward (stand-alone Down block): dx = dp routed to the first maximum of each window (torch's
// tie rule: row-major scan, strict >), zero elsewhere, incl. the odd last row/column the pooling drops
template <int ESZ>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ x, int N, int H,
                                                         int W, int C, float* __restrict__ dx) {
    const int Hp = H / 2, Wp = W / 2;
    const long total = (long)N * H * W * C;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)(e % C); long t = e / C;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H); const int n = (int)(t / H);
        const int py = iy >> 1, px = ix >> 1;
        float g = 0.f;
        if (py < Hp && px < Wp) {
            const long b = (((long)n * H + 2 * py) * W + 2 * px) * C + c;
            int best = 0; float bv = ld1(x, b, ESZ);
            const float v1 = ld1(x, b + C, ESZ), v2 = ld1(x, b + (long)W * C, ESZ), v3 = ld1(x, b + (long)W * C + C, ESZ);
            if (v1 > bv) { bv = v1; best = 1; }
            if (v2 > bv) { bv = v2; best = 2; }
            if (v3 > bv) { bv = v3; best = 3; }
            if (best == ((iy & 1) * 2 + (ix & 1))) g = ld1(dp, (((long)n * Hp + py) * Wp + px) * C + c, ESZ);
        }
        st1(dx, e, g, ESZ);
    }
}

// ---- backward -------------------------------------------------------------------------------
// One "window" = one pixel (POOL = false) or one 2x2 pooling window (POOL = true).  For each
// element: a = relu(s*y+b); da_total = da + (dp routed to the window's first arg-max of a);
// dz = da_total * (a > 0).
template <bool POOL>
struct Win {
    static constexpr int NPX = POOL ? 4 : 1;
};

// EVEN (pooled windows of an even-sized map: every U-Net level): all four pixels and the pooled cell exist, so the validity
// selects (3 per element), the clamped addresses and the integer arg-max index go away -- the routing is three compares per channel
// and mask logic.  The general form below costs 360 VALU instructions per window for 72 loaded bytes: the pooled reduce was issue
// bound at 3.3 TB/s (profiles/r03_bench_bn_bwd.log).
template <bool POOL, int ESZ, bool EVEN = false>
__device__ __forceinline__ void window_dz(const float* da, const float* __restrict__ dp,
                                          const float* __restrict__ y, f32x4 sc, f32x4 sh, long w, int WH, int WW,
                                          int H, int W, int C, int c, f32x4 (&yv)[Win<POOL>::NPX],
                                          f32x4 (&dz)[Win<POOL>::NPX], bool (&ok)[Win<POOL>::NPX]) {
    constexpr int NPX = Win<POOL>::NPX;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 av[NPX];
    if constexpr (POOL && EVEN) {
        const unsigned uw = (unsigned)w, per = (unsigned)(WH * WW);
        const unsigned n = uw / per, rem = uw - n * per, wy = rem / (unsigned)WW, wx = rem - wy * (unsigned)WW;
        const long off0 = (((long)n * H + 2 * wy) * W + 2 * wx) * C + c;
        const long rowo = (long)W * C;
        yv[0] = ld4t<ESZ>(y, off0); yv[1] = ld4t<ESZ>(y, off0 + C); yv[2] = ld4t<ESZ>(y, off0 + rowo); yv[3] = ld4t<ESZ>(y, off0 + rowo + C);
        if (da) {
            dz[0] = ld4t<ESZ>(da, off0); dz[1] = ld4t<ESZ>(da, off0 + C); dz[2] = ld4t<ESZ>(da, off0 + rowo); dz[3] = ld4t<ESZ>(da, off0 + rowo + C);
        } else {
            dz[0] = dz[1] = dz[2] = dz[3] = zero;
        }
        const f32x4 g4 = dp ? ld4t<ESZ>(dp, (long)w * C + c) : zero;       // the pooled map's cell IS window w
#pragma unroll
        for (int q = 0; q < 4; ++q) { ok[q] = true; av[q] = yv[q] * sc + sh; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // first arg-max of relu(a) in the order (0,0) (0,1) (1,0) (1,1): strict > keeps the first
            const float a0 = fmaxf(av[0][j], 0.f), a1 = fmaxf(av[1][j], 0.f), a2 = fmaxf(av[2][j], 0.f), a3 = fmaxf(av[3][j], 0.f);
            const bool m1 = a1 > a0;
            const float b1 = fmaxf(a0, a1);
            const bool m2 = a2 > b1;
            const float b2 = fmaxf(b1, a2);
            const bool m3 = a3 > b2;
            const float g = g4[j];
            const bool e3 = m3, e2 = m2 && !m3, e1 = m1 && !m2 && !m3, e0 = !(m1 || m2 || m3);
            dz[0][j] += e0 ? g : 0.f; dz[1][j] += e1 ? g : 0.f; dz[2][j] += e2 ? g : 0.f; dz[3][j] += e3 ? g : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) dz[q][j] = (av[q][j] > 0.f) ? dz[q][j] : 0.f;
        return;
    }
    // plain: the window IS the pixel, its offset is linear (no division: these kernels were instruction bound on the
    // 64-bit divides, not memory bound); pooled: 32-bit divides (windows per pass < 2^31)
    int n = 0, wy = 0, wx = 0;
    if (POOL) {
        const unsigned uw = (unsigned)w, per = (unsigned)(WH * WW);
        n = (int)(uw / per);
        const unsigned rem = uw - (unsigned)n * per;
        wy = (int)(rem / (unsigned)WW); wx = (int)(rem - (unsigned)wy * (unsigned)WW);
    }
    // every load of the window is issued unconditionally (out-of-range pixels read the window's first pixel, always
    // inside, and are zeroed afterwards): nine independent loads in flight instead of one masked region after another
    const long off0 = POOL ? (((long)n * H + 2 * wy) * W + 2 * wx) * C + c : w * C + c;
#pragma unroll
    for (int q = 0; q < NPX; ++q) {
        const int py = 2 * wy + (q >> 1), px = 2 * wx + (q & 1);
        ok[q] = !POOL || (py < H && px < W);
        const long off = ok[q] ? off0 + ((long)(q >> 1) * W + (q & 1)) * C : off0;
        yv[q] = ld4t<ESZ>(y, off);
        if constexpr (POOL) dz[q] = da ? ld4t<ESZ>(da, off) : zero;      // (pooled layers may have no direct consumer)
        else dz[q] = ld4t<ESZ>(da, off);                                  // plain: da is always there -- no branch around the load
    }
    f32x4 gpool = zero;
    bool gok = false;
    if (POOL && dp) {
        const int PH = H / 2, PW = W / 2;
        gok = wy < PH && wx < PW;
        gpool = ld4t<ESZ>(dp, (((long)n * PH + (gok ? wy : 0)) * PW + (gok ? wx : 0)) * C + c);
    }
    // everything below is selects, not branches (the compiler turned the arg-max routing into ~40 exec-mask regions)
#pragma unroll
    for (int q = 0; q < NPX; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            yv[q][j] = ok[q] ? yv[q][j] : 0.f;
            dz[q][j] = ok[q] ? dz[q][j] : 0.f;
            av[q][j] = ok[q] ? yv[q][j] * sc[j] + sh[j] : 0.f;
        }
    }
    if (POOL) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g = gok ? gpool[j] : 0.f;
            // first arg-max of relu(a) over the window, in the order (0,0) (0,1) (1,0) (1,1): strict > keeps the first
            const float a0 = fmaxf(av[0][j], 0.f), a1 = fmaxf(av[1][j], 0.f), a2 = fmaxf(av[2][j], 0.f), a3 = fmaxf(av[3][j], 0.f);
            const bool m1 = a1 > a0;
            const float b1 = m1 ? a1 : a0;
            const bool m2 = a2 > b1;
            const float b2 = m2 ? a2 : b1;
            const bool m3 = a3 > b2;
            const int best = m3 ? 3 : (m2 ? 2 : (m1 ? 1 : 0));
#pragma unroll
            for (int q = 0; q < NPX; ++q) dz[q][j] += (q == best) ? g : 0.f;
        }
    }
#pragma unroll
    for (int q = 0; q < NPX; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) dz[q][j] = (av[q][j] > 0.f) ? dz[q][j] : 0.f;
}

// grid-stride over windows; thread = (channel quad, window lane).  partials[block][2][C]
// blockIdx.y = forward pass (batched passes: each has its own slice of every tensor, constants and partial rows)
struct PassOff { long act, pool, aff, part, coef; };   // element strides between passes (bytes are esz * act for tensors)

template <bool POOL, int ESZ, bool EVEN = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ da, const float* __restrict__ dp,
                                                           const float* __restrict__ y, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int N, int H, int W, int C,
                                                           int G, float* __restrict__ partials, const PassOff po) {
    constexpr int NPX = Win<POOL>::NPX;
    {
        const long g = blockIdx.y;
        if (da) da = (const float*)((const char*)da + g * po.act * ESZ);
        if (dp) dp = (const float*)((const char*)dp + g * po.pool * ESZ);
        y = (const float*)((const char*)y + g * po.act * ESZ);
        scale += g * po.aff; shift += g * po.aff; partials += g * po.part;
    }
    __shared__ f32x4 red[2][256];
    const int C4 = C / 4, PL = 256 / G;
    const int cq0 = threadIdx.x % G, pl = threadIdx.x / G;
    const int WH = POOL ? (H + 1) / 2 : H, WW = POOL ? (W + 1) / 2 : W;
    const long nwin = (long)N * WH * WW;
    for (int cb = 0; cb < C4; cb += G) {      // uniform trip count: the body holds barriers
        const int cq = cb + cq0;
        const bool active = cq < C4;
        const int c = active ? cq * 4 : 0;
        const f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
        if (active) {
            // plain path: four windows per trip, their eight loads issued before the first use (one window per trip left the
            // memory-level parallelism to occupancy alone: 4.7 -> 5.2 TB/s; eight per trip: no further gain); the sums still run in ascending window order
            constexpr int U = POOL ? 1 : 4;
            const long stride = (long)gridDim.x * PL;
            long w = (long)blockIdx.x * PL + pl;
            for (; w + (U - 1) * stride < nwin; w += U * stride) {
                f32x4 yv[U][NPX], dz[U][NPX]; bool ok[U][NPX];
#pragma unroll
                for (int u = 0; u < U; ++u) window_dz<POOL, ESZ, EVEN>(da, dp, y, sc, sh, w + u * stride, WH, WW, H, W, C, c, yv[u], dz[u], ok[u]);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int q = 0; q < NPX; ++q) { s1 += dz[u][q]; s2 += dz[u][q] * yv[u][q]; }
            }
            for (; w < nwin; w += stride) {
                f32x4 yv[NPX], dz[NPX]; bool ok[NPX];
                window_dz<POOL, ESZ, EVEN>(da, dp, y, sc, sh, w, WH, WW, H, W, C, c, yv, dz, ok);
#pragma unroll
                for (int q = 0; q < NPX; ++q) { s1 += dz[q]; s2 += dz[q] * yv[q]; }
            }
        }
        red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
        __syncthreads();
        if (pl == 0 && active) {
            for (int k = 1; k < PL; ++k) { s1 += red[0][k * G + cq0]; s2 += red[1][k * G + cq0]; }
            *(f32x4*)(partials + ((long)blockIdx.x * 2 + 0) * C + c) = s1;
            *(f32x4*)(partials + ((long)blockIdx.x * 2 + 1) * C + c) = s2;
        }
        __syncthreads();
    }
}

// partials[pass][rows][2][C] -> dgamma, dbeta (the passes' contributions added one after the other, as separate calls
// would), coef[pass][3][C]
template <bool PRE>
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ partials, int rows, int C, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, float* dgamma, float* dbeta, int accumulate,
                                       float* coef, int passes, const PassOff po, long rstride, long roff, int R) {
    __shared__ double red[2][32][32];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;      // 32 channels x 32 row lanes
    const int c = blockIdx.x * 32 + cl;
    float dg_run = 0.f, db_run = 0.f;
    if (rg == 0 && c < C && dgamma && accumulate) { dg_run = dgamma[c]; db_run = dbeta[c]; }
    // up to four passes per round, eight row lanes each (as bn_finalize_kernel: the passes' sums are independent, only the
    // dgamma/dbeta accumulation is sequential; a pass's sums are formed in the same order alone or batched)
    constexpr int LPP = 8, PPR = 32 / LPP;
    const int gl = rg / LPP, lane = rg % LPP;
    for (int g0 = 0; g0 < passes; g0 += PPR) {
        const int g = g0 + gl;
        double s1 = 0.0, s2 = 0.0;
        if (c < C && g < passes) {
            const float* pt = partials + (long)g * po.part + roff + c;
            if constexpr (PRE) {     // rows of a convolution's epilogue, pre-reduced in place by bn_stat_stage1_kernel: f64 sums parked as
                                     // (hi, lo) float pairs at rows sp*R (sum 1) and sp*R+1 (sum 2); at most 32 splits = four per lane
                float v[4][4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int sp = lane + LPP * k;
                    const bool ok = sp * R < rows;
                    const long a0 = ok ? (long)(sp * R) * 2 * C : 0, a1 = ok ? (long)(sp * R + 1) * 2 * C : 0;
                    const float x0 = pt[a0], x1 = pt[a0 + C], x2 = pt[a1], x3 = pt[a1 + C];
                    v[k][0] = ok ? x0 : 0.f; v[k][1] = ok ? x1 : 0.f; v[k][2] = ok ? x2 : 0.f; v[k][3] = ok ? x3 : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { s1 += (double)v[k][0] + (double)v[k][1]; s2 += (double)v[k][2] + (double)v[k][3]; }
            }
            int r = PRE ? rows : lane;
            for (; r + 7 * LPP < rows; r += 8 * LPP) {       // eight rows' loads issued before the first sum (written out: left
                float a[8], b[8];                            // to the unroller, each load was waited for on its own)
#pragma unroll
                for (int u = 0; u < 8; ++u) { const float* q = pt + (long)(r + u * LPP) * rstride; a[u] = q[0]; b[u] = q[C]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { s1 += (double)a[u]; s2 += (double)b[u]; }
            }
            for (; r < rows; r += LPP) {
                s1 += (double)pt[(long)r * rstride];
                s2 += (double)pt[(long)r * rstride + C];
            }
        }
        red[0][rg][cl] = s1; red[1][rg][cl] = s2;
        __syncthreads();
        if (rg == 0 && c < C) {
            for (int q = 0; q < PPR && g0 + q < passes; ++q) {
                double t1 = 0.0, t2 = 0.0;
                for (int k = 0; k < LPP; ++k) { t1 += red[0][q * LPP + k][cl]; t2 += red[1][q * LPP + k][cl]; }
                const int gq = g0 + q;
                const double mu = mean[(long)gq * po.aff + c], rs = rstd[(long)gq * po.aff + c], gm = gamma[c];
                const double dbe = t1, dga = rs * (t2 - mu * t1);
                const double c0 = gm * rs, m1 = dbe / count, m2 = dga / count;
                const double c1 = -c0 * m2 * rs, c2 = -c0 * m1 - c1 * mu;
                float* cf = coef + (long)gq * po.coef;
                cf[c] = (float)c0; cf[C + c] = (float)c1; cf[2 * C + c] = (float)c2;
                if (gq == 0 && !accumulate) { dg_run = (float)dga; db_run = (float)dbe; }
                else { dg_run = dg_run + (float)dga; db_run = db_run + (float)dbe; }
            }
        }
        __syncthreads();
    }
    if (rg == 0 && c < C && dgamma) { dgamma[c] = dg_run; dbeta[c] = db_run; }
}

template <bool POOL, int ESZ, bool EVEN = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* da, const float* __restrict__ dp,
                                                          const float* __restrict__ y, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const float* __restrict__ coef,
                                                          int N, int H, int W, int C, int G, float* dy, const PassOff po) {
    constexpr int NPX = Win<POOL>::NPX;
    {
        const long g = blockIdx.y;
        if (da) da = (const float*)((const char*)da + g * po.act * ESZ);
        if (dp) dp = (const float*)((const char*)dp + g * po.pool * ESZ);
        y = (const float*)((const char*)y + g * po.act * ESZ);
        dy = (float*)((char*)dy + g * po.act * ESZ);
        scale += g * po.aff; shift += g * po.aff; coef += g * po.coef;
    }
    const int C4 = C / 4, PL = 256 / G;
    const int cq0 = threadIdx.x % G, pl = threadIdx.x / G;
    const int WH = POOL ? (H + 1) / 2 : H, WW = POOL ? (W + 1) / 2 : W;
    const long nwin = (long)N * WH * WW;
    for (int cq = cq0; cq < C4; cq += G) {
        const int c = cq * 4;
        const f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
        const f32x4 k0 = *(const f32x4*)(coef + c), k1 = *(const f32x4*)(coef + C + c), k2 = *(const f32x4*)(coef + 2 * C + c);
        // (one window per trip here: with four the loads and the four stores of a trip queue behind each other -- measured
        // 110 -> 125 us, while the store-free reduce gained 13 % from the same change)
        const long stride = (long)gridDim.x * PL;
        long w = (long)blockIdx.x * PL + pl;
        for (; w < nwin; w += stride) {
            f32x4 yv[NPX], dz[NPX]; bool ok[NPX];
            window_dz<POOL, ESZ, EVEN>(da, dp, y, sc, sh, w, WH, WW, H, W, C, c, yv, dz, ok);
            long base = w * C + c;                       // plain: the pixel itself
            if (POOL) {
                const unsigned uw = (unsigned)w, per = (unsigned)(WH * WW);
                const unsigned n = uw / per, rem = uw - n * per, wy = rem / (unsigned)WW, wx = rem - wy * (unsigned)WW;
                base = (((long)n * H + 2 * wy) * W + 2 * wx) * C + c;
            }
#pragma unroll
            for (int q = 0; q < NPX; ++q) {
                if (!ok[q]) continue;
                st4t<ESZ>(dy, base + ((long)(q >> 1) * W + (q & 1)) * C, k0 * dz[q] + k1 * yv[q] + k2);
            }
        }
    }
}

// ---- plain (un-pooled) bf16 passes with 8 channels = 16 bytes per lane (round 3) --------------------------------------------
// The 4-channel kernels above keep 16 bytes per thread in flight (two 8-byte loads, one pixel per trip in the apply pass): at
// their occupancy that is about half of what 8 TB/s x the memory latency asks of a CU, and the plain apply pass ran at
// 4.8-5.1 TB/s where the pooled one (nine loads per thread) reaches 5.5.  Here a lane owns 8 channels of a pixel: thread =
// (octet tid % G8, pixel lane tid / G8), G8 = C / 8 a power of two <= 256; the arithmetic is the same per element.
typedef __attribute__((ext_vector_type(8))) elt_t bf16x8_bn;
struct F8 { f32x4 lo, hi; };
__device__ __forceinline__ F8 up8(bf16x8_bn v) {
    return F8{(f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]}, (f32x4){(float)v[4], (float)v[5], (float)v[6], (float)v[7]}};
}
__device__ __forceinline__ F8 ld8f(const float* p) { return F8{*(const f32x4*)p, *(const f32x4*)(p + 4)}; }

__global__ __launch_bounds__(256) void bn_bwd_apply_x8_kernel(const elt_t* __restrict__ da, const elt_t* __restrict__ y,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ coef, long npix, int C, int G8,
                                                             elt_t* __restrict__ dy, const PassOff po) {
    {
        const long g = blockIdx.y;
        da += g * po.act; y += g * po.act; dy += g * po.act;
        scale += g * po.aff; shift += g * po.aff; coef += g * po.coef;
    }
    const int PL = 256 / G8;
    const int c = 8 * (threadIdx.x % G8), pl = threadIdx.x / G8;
    const F8 sc = ld8f(scale + c), sh = ld8f(shift + c), k0 = ld8f(coef + c), k1 = ld8f(coef + C + c), k2 = ld8f(coef + 2 * C + c);
    const long stride = (long)gridDim.x * PL;
    constexpr int U = 2;
    long w = (long)blockIdx.x * PL + pl;
    auto one = [&](bf16x8_bn yb, bf16x8_bn db) {
        const F8 yv = up8(yb), dv = up8(db);
        bf16x8_bn o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float al = yv.lo[j] * sc.lo[j] + sh.lo[j], ah = yv.hi[j] * sc.hi[j] + sh.hi[j];
            const float zl = al > 0.f ? dv.lo[j] : 0.f, zh = ah > 0.f ? dv.hi[j] : 0.f;
            o[j] = (elt_t)(k0.lo[j] * zl + k1.lo[j] * yv.lo[j] + k2.lo[j]);
            o[4 + j] = (elt_t)(k0.hi[j] * zh + k1.hi[j] * yv.hi[j] + k2.hi[j]);
        }
        return o;
    };
    for (; w + (U - 1) * stride < npix; w += U * stride) {
        bf16x8_bn yb[U], db[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { yb[u] = *(const bf16x8_bn*)(y + (w + u * stride) * C + c); db[u] = *(const bf16x8_bn*)(da + (w + u * stride) * C + c); }
#pragma unroll
        for (int u = 0; u < U; ++u) *(bf16x8_bn*)(dy + (w + u * stride) * C + c) = one(yb[u], db[u]);
    }
    for (; w < npix; w += stride)
        *(bf16x8_bn*)(dy + w * C + c) = one(*(const bf16x8_bn*)(y + w * C + c), *(const bf16x8_bn*)(da + w * C + c));
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_x8_kernel(const elt_t* __restrict__ da, const elt_t* __restrict__ y,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              long npix, int C, int G8, float* __restrict__ partials,
                                                              const PassOff po) {
    {
        const long g = blockIdx.y;
        da += g * po.act; y += g * po.act;
        scale += g * po.aff; shift += g * po.aff; partials += g * po.part;
    }
    __shared__ f32x4 red[4][256];
    const int PL = 256 / G8;
    const int g8 = threadIdx.x % G8, c = 8 * g8, pl = threadIdx.x / G8;
    const F8 sc = ld8f(scale + c), sh = ld8f(shift + c);
    F8 s1 = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, s2 = s1;
    const long stride = (long)gridDim.x * PL;
    constexpr int U = 4;
    long w = (long)blockIdx.x * PL + pl;
    auto acc1 = [&](bf16x8_bn yb, bf16x8_bn db) {
        const F8 yv = up8(yb), dv = up8(db);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float al = yv.lo[j] * sc.lo[j] + sh.lo[j], ah = yv.hi[j] * sc.hi[j] + sh.hi[j];
            const float zl = al > 0.f ? dv.lo[j] : 0.f, zh = ah > 0.f ? dv.hi[j] : 0.f;
            s1.lo[j] += zl; s1.hi[j] += zh;
            s2.lo[j] += zl * yv.lo[j]; s2.hi[j] += zh * yv.hi[j];
        }
    };
    for (; w + (U - 1) * stride < npix; w += U * stride) {       // four pixels' loads in flight, sums in ascending pixel order
        bf16x8_bn yb[U], db[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { yb[u] = *(const bf16x8_bn*)(y + (w + u * stride) * C + c); db[u] = *(const bf16x8_bn*)(da + (w + u * stride) * C + c); }
#pragma unroll
        for (int u = 0; u < U; ++u) acc1(yb[u], db[u]);
    }
    for (; w < npix; w += stride) acc1(*(const bf16x8_bn*)(y + w * C + c), *(const bf16x8_bn*)(da + w * C + c));
    red[0][threadIdx.x] = s1.lo; red[1][threadIdx.x] = s1.hi; red[2][threadIdx.x] = s2.lo; red[3][threadIdx.x] = s2.hi;
    __syncthreads();
    if (pl == 0) {                                              // fixed order over the block's pixel lanes
        for (int k = 1; k < PL; ++k) {
            s1.lo += red[0][k * G8 + g8]; s1.hi += red[1][k * G8 + g8];
            s2.lo += red[2][k * G8 + g8]; s2.hi += red[3][k * G8 + g8];
        }
        float* r0 = partials + ((long)blockIdx.x * 2 + 0) * C + c;
        float* r1 = partials + ((long)blockIdx.x * 2 + 1) * C + c;
        *(f32x4*)r0 = s1.lo; *(f32x4*)(r0 + 4) = s1.hi;
        *(f32x4*)r1 = s2.lo; *(f32x4*)(r1 + 4) = s2.hi;
    }
}

// 8-channel kernels: bf16, no pooling, C / 8 a power of two <= 256 (every layer of the U-Net and of ResNet-50/101);
// ustrun_debug_flags bit 12 (4096) keeps the 4-channel kernels (A/B runs)
static bool x8_ok(int C, bool pool, int dtype) {
    if (g_debug_flags & 4096) return false;
    const int g8 = C / 8;
    return dtype == USTRUN_D16 && !pool && C % 8 == 0 && g8 >= 1 && g8 <= 256 && (g8 & (g8 - 1)) == 0;
}

// ustrun_debug_last_bn_variant: 0x424E0000 ('BN') | pass (0 reduce, 1 apply) << 8 | element bytes << 4 | even windows << 2 |
// pooled << 1 | eight channels per lane; per calling thread
thread_local int g_last_bn_variant = 0;
void note_bn_variant(int pass, int esz, bool even, bool pool, bool x8) {
    g_last_bn_variant = 0x424E0000 | pass << 8 | esz << 4 | (even ? 4 : 0) | (pool ? 2 : 0) | (x8 ? 1 : 0);
}

int group_size(int C4) { int g = 1; while (g < C4 && g < 256) g <<= 1; return g; }

int reduce_blocks(long nwin, int G) {
    const int PL = 256 / G;
    long b = (nwin + (long)PL * 8 - 1) / ((long)PL * 8);   // >= 8 windows per thread
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace
}  // namespace ustrun

using namespace ustrun;

namespace ustrun {
int bn_finalize_passes(float* stat, int mtiles, int passes, int C, int64_t count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                       int update_running, float* scale, float* shift, float* mean, float* rstd, long astride, hipStream_t s,
                       unsigned* tickets, int tail_rows, int64_t tail_count) {
    USTRUN_CHECK(stat && gamma && beta && scale && shift && mean && rstd, "bn_finalize: null pointer");
    USTRUN_CHECK(tail_rows >= 0 && tail_rows <= mtiles && (tail_rows == 0 || tail_count > 0), "bn_finalize: tail of %d rows", tail_rows);
    BnTail tail = {tail_rows, 0, (double)tail_count};
    if (tail_rows >= 96 && C % 32 == 0) {       // the split a call of its own would choose for these rows
        tail.R = cdiv(tail_rows, 32);
        while (tail_rows % tail.R == 1) ++tail.R;
    }
    if (tail_rows > 0) tickets = nullptr;       // (the one-launch form knows no tail)
    USTRUN_CHECK(!update_running || (running_mean && running_var), "bn_finalize: running buffers missing");
    USTRUN_CHECK(mtiles > 0 && passes > 0 && C > 0 && count > 0, "bn_finalize: empty");
    // Two stages from 96 rows up (the first one rewrites `stat` in place): one block per 32 channels pulls the whole table
    // through ONE CU (262 KB for 256 rows x 4 passes: 24 us measured, 368 launches per step), the staged form spreads it over
    // 32 x passes blocks per channel block (6 + 5.5 us).
    if (mtiles >= 96 && C % 32 == 0) {
        int R = cdiv(mtiles, 32);
        while (mtiles % R == 1) ++R;         // every split needs two rows to park its sums in
        if (tickets && cdiv(mtiles, R) <= 32) {       // (zeroed by the caller once; C / 32 <= BN_TICKETS counters, back at zero when the launch retires)
            USTRUN_CHECK(C / 32 <= BN_TICKETS, "bn_finalize: %d channels exceed the ticket table", C);
            hipLaunchKernelGGL(bn_stat_fused_kernel, dim3(C / 32, cdiv(mtiles, R), passes), dim3(256), 0, s, (float*)stat, mtiles,
                               C, (double)count, R, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps,
                               update_running, scale, shift, mean, rstd, passes, astride, tickets);
            USTRUN_LAUNCH_CHECK("bn_stat_fused");
            return 0;
        }
        // the tail's split is its own (fewer rows, smaller R): it can need MORE y-blocks than the equal passes (ADVICE r5:
        // 144 rows -> R 5 -> 29 blocks, a 96-row tail -> R 3 -> 32), so the grid covers both and each block checks its range
        const int ysplits = std::max(cdiv(mtiles, R), tail.R > 0 ? cdiv(tail.rows, tail.R) : 0);
        hipLaunchKernelGGL(bn_stat_stage1_kernel, dim3(C / 32, ysplits, passes + (tail.R > 0 ? 1 : 0)), dim3(256), 0, s,
                           (float*)stat, mtiles, C, R, passes, tail);
        USTRUN_LAUNCH_CHECK("bn_stat_stage1");
        hipLaunchKernelGGL(bn_finalize_kernel<true>, dim3(C / 32), dim3(1024), 0, s, stat, mtiles, C, (double)count, R, gamma,
                           beta, running_mean, running_var, num_batches_tracked, momentum, eps, update_running, scale, shift,
                           mean, rstd, passes, astride, tail);
    } else {
        hipLaunchKernelGGL(bn_finalize_kernel<false>, dim3(cdiv(C, 32)), dim3(1024), 0, s, stat, mtiles, C, (double)count, 0,
                           gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, update_running, scale,
                           shift, mean, rstd, passes, astride, tail);
    }
    USTRUN_LAUNCH_CHECK("bn_finalize");
    return 0;
}
}  // namespace ustrun

extern "C" int ustrun_bn_finalize(float* stat, int mtiles, int C, int64_t count, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var,
                                  int64_t* num_batches_tracked, float momentum, float eps, int update_running,
                                  float* scale, float* shift, float* mean, float* rstd, ustrun_stream_t s) {
    return bn_finalize_passes(stat, mtiles, 1, C, count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum,
                              eps, update_running, scale, shift, mean, rstd, 0, (hipStream_t)s);
}

extern "C" int ustrun_bn_eval_affine(int C, const float* gamma, const float* beta, const float* running_mean,
                                     const float* running_var, float eps, float* scale, float* shift,
                                     ustrun_stream_t s) {
    USTRUN_CHECK(C > 0 && gamma && beta && running_mean && running_var && scale && shift, "bn_eval_affine: bad args");
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)s, C, gamma, beta,
                       running_mean, running_var, eps, scale, shift);
    USTRUN_LAUNCH_CHECK("bn_eval_affine");
    return 0;
}

namespace ustrun {
// used by the U-Net forward in eval mode
int bn_eval_affine_layers(int nlayers, const int* C, const float* const* gamma, const float* const* beta,
                          const float* const* rm, const float* const* rv, float* const* aff, float eps, int passes,
                          hipStream_t s) {
    USTRUN_CHECK(nlayers > 0 && nlayers <= 18 && passes >= 1, "bn_eval_affine_layers: %d layers", nlayers);
    EvalAffineJobs j;
    int cmax = 0;
    for (int l = 0; l < nlayers; ++l) {
        USTRUN_CHECK(gamma[l] && beta[l] && rm[l] && rv[l] && aff[l] && C[l] > 0, "bn_eval_affine_layers: null pointer");
        j.gamma[l] = gamma[l]; j.beta[l] = beta[l]; j.rm[l] = rm[l]; j.rv[l] = rv[l]; j.aff[l] = aff[l]; j.C[l] = C[l];
        cmax = C[l] > cmax ? C[l] : cmax;
    }
    hipLaunchKernelGGL(bn_eval_affine_multi_kernel, dim3(cdiv(cmax, 256), nlayers * passes), dim3(256), 0, s, j, eps, passes);
    USTRUN_LAUNCH_CHECK("bn_eval_affine_layers");
    return 0;
}
}  // namespace ustrun

extern "C" int ustrun_bn_relu_apply(const void* y, const float* scale, const float* shift, int64_t npix, int C,
                                    int HW, float* out, int out_nchw, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(dtype_ok(dtype), "bn_relu_apply: dtype %d not built", dtype);
    USTRUN_CHECK(y && out && npix > 0 && C > 0 && HW > 0, "bn_relu_apply: bad args");
    int blocks = cdiv(npix * C, 256 * 4);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)y, act_esz(dtype), scale,
                       shift, (long)npix, C, HW, out, out_nchw);
    USTRUN_LAUNCH_CHECK("bn_relu_apply");
    return 0;
}

extern "C" int ustrun_act16(const ustrun_src_t* src, int N, void* out, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(dtype == USTRUN_D16, "act16: 16-bit storage only (dtype %d)", dtype);
    USTRUN_CHECK(src && src->ptr && src->scale && src->shift && out && N > 0, "act16: bad args");
    const int C = src->C, H = src->H, W = src->W;
    USTRUN_CHECK(src->sC == 1 && src->sW == C && src->sH == (int64_t)W * C && src->sN == (int64_t)H * W * C && !src->pool &&
                 !src->off_y && !src->off_x && !src->f32, "act16: source must be a plain contiguous NHWC activation");
    USTRUN_CHECK(C % 8 == 0 && C / 8 <= 256, "act16: C=%d unsupported", C);       // (256 / (C/8) pixels per block; spare threads idle)
    const long npix = (long)N * H * W;
    USTRUN_CHECK(npix < (1L << 32), "act16: %ld pixels", npix);
    const int ppb = 256 / (C / 8);
    long nb = (npix + ppb * 8 - 1) / (ppb * 8);
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(act16_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)s, (const elt_t*)src->ptr, src->scale, src->shift,
                       src->relu, src->gN, (long)src->gstride, npix, H * W, C, (elt_t*)out);
    USTRUN_LAUNCH_CHECK("act16");
    return 0;
}

extern "C" int ustrun_pool_act(const ustrun_src_t* src, int N, void* out, int dtype, ustrun_stream_t s) {
    return ustrun_pool_act2(src, N, out, nullptr, dtype, s);
}

extern "C" int ustrun_pool_act2(const ustrun_src_t* src, int N, void* out, void* act, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(dtype_ok(dtype), "pool_act: dtype %d not built", dtype);
    USTRUN_CHECK(src && src->ptr && out && N > 0, "pool_act: bad args");
    USTRUN_CHECK(!act || (dtype == USTRUN_D16 && !(src->H & 1) && !(src->W & 1)),
                 "pool_act: the un-pooled activation needs 16-bit storage and even extents (%dx%d)", src->H, src->W);
    const int C = src->C, H = src->H, W = src->W, V = dtype == USTRUN_D16 ? 8 : 4;
    USTRUN_CHECK(C % V == 0 && H >= 2 && W >= 2, "pool_act: C=%d extent %dx%d unsupported", C, H, W);
    USTRUN_CHECK(src->sC == 1 && src->sW == C && src->sH == (int64_t)W * C && src->sN == (int64_t)H * W * C && !src->pool &&
                 !src->off_y && !src->off_x && !src->f32, "pool_act: source must be a plain contiguous NHWC activation");
    USTRUN_CHECK((src->scale == nullptr) == (src->shift == nullptr), "pool_act: scale/shift must come together");
    USTRUN_CHECK(C / V <= 256, "pool_act: C=%d too wide", C);
    const long npix = (long)N * (H / 2) * (W / 2);
    USTRUN_CHECK(npix < (1L << 32), "pool_act: %ld pooled pixels", npix);
    const int ppb = 256 / (C / V);
    long nb = (npix + ppb * 4 - 1) / (ppb * 4);             // about four pixels per thread
    if (nb > 16384) nb = 16384;
    if (nb < 1) nb = 1;
    const int blocks = (int)nb;
    const int gN = src->scale ? src->gN : 0;
    if (dtype == USTRUN_D16)
        hipLaunchKernelGGL(pool_act_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)src->ptr, src->scale,
                           src->shift, src->relu, gN, (long)src->gstride, N, H, W, C, (float*)out, (elt_t*)act);
    else
        hipLaunchKernelGGL(pool_act_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)src->ptr, src->scale,
                           src->shift, src->relu, gN, (long)src->gstride, N, H, W, C, (float*)out, (elt_t*)nullptr);
    USTRUN_LAUNCH_CHECK("pool_act");
    return 0;
}

extern "C" int ustrun_maxpool_bwd(const void* dp, const void* x, int N, int H, int W, int C, void* dx, int dtype,
                                  ustrun_stream_t s) {
    USTRUN_CHECK(dtype_ok(dtype), "maxpool_bwd: dtype %d not built", dtype);
    USTRUN_CHECK(dp && x && dx && N > 0 && H >= 2 && W >= 2 && C > 0, "maxpool_bwd: bad args");
    const long total = (long)N * H * W * C;
    long nb = (total + 1023) / 1024;
    if (nb > 8192) nb = 8192;
    if (dtype == USTRUN_D16)
        hipLaunchKernelGGL(maxpool_bwd_kernel<2>, dim3((int)nb), dim3(256), 0, (hipStream_t)s, (const float*)dp, (const float*)x, N, H,
                           W, C, (float*)dx);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel<4>, dim3((int)nb), dim3(256), 0, (hipStream_t)s, (const float*)dp, (const float*)x, N, H,
                           W, C, (float*)dx);
    USTRUN_LAUNCH_CHECK("maxpool_bwd");
    return 0;
}

extern "C" int64_t ustrun_bn_bwd_partials_bytes(int64_t npix, int C) {
    (void)npix;
    return (int64_t)1024 * 2 * C * sizeof(float);
}

namespace ustrun {
// `passes` forward passes in one launch each: tensors of pass g start g*act_elems (g*pool_elems for dp) elements
// further, its constants g*aff_stride floats further (scale/shift/mean/rstd), its coefficients at coef + g*3C.
// dgamma/dbeta receive the passes' contributions in order.
int bn_bwd_reduce_passes(const void* da, const void* dp, const void* y, const float* scale, const float* shift,
                         const float* mean, const float* rstd, const float* gamma, int N, int H, int W, int C, float* dgamma,
                         float* dbeta, int accumulate, float* coef, float* partials, int64_t partials_bytes, int dtype,
                         int passes, long act_elems, long pool_elems, long aff_stride, hipStream_t s) {
    USTRUN_CHECK(dtype_ok(dtype), "bn_bwd_reduce: dtype %d not built", dtype);
    USTRUN_CHECK((da || dp) && y && scale && shift && mean && rstd && gamma && coef && partials, "bn_bwd_reduce: null pointer");
    USTRUN_CHECK(C % 4 == 0 && C > 0 && passes >= 1 && passes <= 64, "bn_bwd_reduce: C=%d must be a multiple of 4", C);
    USTRUN_CHECK(partials_bytes >= ustrun_bn_bwd_partials_bytes((int64_t)N * H * W, C), "bn_bwd_reduce: partials too small");
    const bool pool = dp != nullptr;
    const bool x8 = x8_ok(C, pool, dtype) && da;
    const int G = x8 ? C / 8 : group_size(C / 4);
    const long nwin = pool ? (long)N * ((H + 1) / 2) * ((W + 1) / 2) : (long)N * H * W;
    USTRUN_CHECK(nwin < (1L << 31), "bn_bwd_reduce: %ld windows per pass", nwin);
    int blocks = reduce_blocks(nwin, G);
    if (blocks > 1024 / passes) blocks = 1024 / passes;          // all passes' rows share the 1024-row table
    const PassOff po = {act_elems, pool_elems, aff_stride, (long)blocks * 2 * C, 3L * C};
    note_bn_variant(0, act_esz(dtype), pool && !(H & 1) && !(W & 1), pool, x8);
    if (x8) {
        hipLaunchKernelGGL(bn_bwd_reduce_x8_kernel, dim3(blocks, passes), dim3(256), 0, s, (const elt_t*)da, (const elt_t*)y, scale, shift,
                           nwin, C, G, partials, po);
        USTRUN_LAUNCH_CHECK("bn_bwd_reduce");
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<false>, dim3(cdiv(C, 32)), dim3(1024), 0, s, partials, blocks, C, (double)N * H * W, gamma,
                           mean, rstd, dgamma, dbeta, accumulate, coef, passes, po, 2L * C, 0L, 0);
        USTRUN_LAUNCH_CHECK("bn_bwd_finalize");
        return 0;
    }
#define USTRUN_BN_REDUCE(P, E, V)                                                                                      \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<P, E, V>), dim3(blocks, passes), dim3(256), 0, s, (const float*)da,      \
                       (const float*)dp, (const float*)y, scale, shift, N, H, W, C, G, partials, po)
    const bool even = pool && !(H & 1) && !(W & 1);
    if (dtype == USTRUN_D16) { if (even) USTRUN_BN_REDUCE(true, 2, true); else if (pool) USTRUN_BN_REDUCE(true, 2, false); else USTRUN_BN_REDUCE(false, 2, false); }
    else { if (even) USTRUN_BN_REDUCE(true, 4, true); else if (pool) USTRUN_BN_REDUCE(true, 4, false); else USTRUN_BN_REDUCE(false, 4, false); }
#undef USTRUN_BN_REDUCE
    USTRUN_LAUNCH_CHECK("bn_bwd_reduce");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<false>, dim3(cdiv(C, 32)), dim3(1024), 0, s, partials, blocks, C, (double)N * H * W, gamma,
                       mean, rstd, dgamma, dbeta, accumulate, coef, passes, po, 2L * C, 0L, 0);
    USTRUN_LAUNCH_CHECK("bn_bwd_finalize");
    return 0;
}

int bn_bwd_finalize_rows(const float* partials, int rows, long rstride, long roff, int C, int64_t count, const float* gamma,
                         const float* mean, const float* rstd, float* dgamma, float* dbeta, int accumulate, float* coef,
                         int passes, long aff_stride, hipStream_t s) {
    USTRUN_CHECK(partials && rows > 0 && C > 0 && gamma && mean && rstd && coef && passes >= 1, "bn_bwd_finalize_rows: bad args");
    const PassOff po = {0, 0, aff_stride, (long)rows * rstride, 3L * C};
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<false>, dim3(cdiv(C, 32)), dim3(1024), 0, s, partials, rows, C, (double)count, gamma, mean,
                       rstd, dgamma, dbeta, accumulate, coef, passes, po, rstride, roff, 0);
    USTRUN_LAUNCH_CHECK("bn_bwd_finalize");
    return 0;
}

// rows a convolution epilogue wrote (the forward-statistics table format: [pass][rows][2][C]) -> the same finalize; long tables
// go through the forward statistics' in-place stage 1 first
int bn_bwd_finalize_stat(float* stat, int rows, int passes, int C, int64_t count, const float* gamma, const float* mean,
                         const float* rstd, long aff_stride, float* dgamma, float* dbeta, int accumulate, float* coef, hipStream_t s) {
    USTRUN_CHECK(stat && rows > 0 && passes >= 1 && C > 0 && gamma && mean && rstd && coef, "bn_bwd_finalize_stat: bad args");
    const PassOff po = {0, 0, aff_stride, (long)rows * 2 * C, 3L * C};
    if (rows >= 96 && C % 32 == 0) {
        int R = cdiv(rows, 32);
        while (rows % R == 1) ++R;
        hipLaunchKernelGGL(bn_stat_stage1_kernel, dim3(C / 32, cdiv(rows, R), passes), dim3(256), 0, s, stat, rows, C, R, passes, BnTail{0, 0, 0.0});
        USTRUN_LAUNCH_CHECK("bn_stat_stage1");
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<true>, dim3(C / 32), dim3(1024), 0, s, stat, rows, C, (double)count, gamma, mean, rstd,
                           dgamma, dbeta, accumulate, coef, passes, po, 2L * C, 0L, R);
    } else {
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<false>, dim3(cdiv(C, 32)), dim3(1024), 0, s, stat, rows, C, (double)count, gamma, mean,
                           rstd, dgamma, dbeta, accumulate, coef, passes, po, 2L * C, 0L, 0);
    }
    USTRUN_LAUNCH_CHECK("bn_bwd_finalize");
    return 0;
}

int bn_bwd_apply_passes(const void* da, const void* dp, const void* y, const float* scale, const float* shift,
                        const float* coef, int N, int H, int W, int C, void* dy, int dtype, int passes, long act_elems,
                        long pool_elems, long aff_stride, hipStream_t s) {
    USTRUN_CHECK(dtype_ok(dtype), "bn_bwd_apply: dtype %d not built", dtype);
    USTRUN_CHECK((da || dp) && y && scale && shift && coef && dy, "bn_bwd_apply: null pointer");
    USTRUN_CHECK(C % 4 == 0 && C > 0, "bn_bwd_apply: C=%d must be a multiple of 4", C);
    const bool pool = dp != nullptr;
    const bool x8 = x8_ok(C, pool, dtype) && da;
    const int G = x8 ? C / 8 : group_size(C / 4);
    const long nwin = pool ? (long)N * ((H + 1) / 2) * ((W + 1) / 2) : (long)N * H * W;
    USTRUN_CHECK(nwin < (1L << 31), "bn_bwd_apply: %ld windows per pass", nwin);
    const int PL = 256 / G;
    long blocks = (nwin + (long)PL * 4 - 1) / ((long)PL * 4);
    if (blocks > 4096) blocks = 4096;
    const PassOff po = {act_elems, pool_elems, aff_stride, 0, 3L * C};
    note_bn_variant(1, act_esz(dtype), pool && !(H & 1) && !(W & 1), pool, x8);
    if (x8) {
        hipLaunchKernelGGL(bn_bwd_apply_x8_kernel, dim3((int)blocks, passes), dim3(256), 0, s, (const elt_t*)da, (const elt_t*)y, scale, shift,
                           coef, nwin, C, G, (elt_t*)dy, po);
        USTRUN_LAUNCH_CHECK("bn_bwd_apply");
        return 0;
    }
#define USTRUN_BN_APPLY(P, E, V)                                                                                            \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<P, E, V>), dim3((int)blocks, passes), dim3(256), 0, s, (const float*)da,       \
                       (const float*)dp, (const float*)y, scale, shift, coef, N, H, W, C, G, (float*)dy, po)
    const bool even = pool && !(H & 1) && !(W & 1);
    if (dtype == USTRUN_D16) { if (even) USTRUN_BN_APPLY(true, 2, true); else if (pool) USTRUN_BN_APPLY(true, 2, false); else USTRUN_BN_APPLY(false, 2, false); }
    else { if (even) USTRUN_BN_APPLY(true, 4, true); else if (pool) USTRUN_BN_APPLY(true, 4, false); else USTRUN_BN_APPLY(false, 4, false); }
#undef USTRUN_BN_APPLY
    USTRUN_LAUNCH_CHECK("bn_bwd_apply");
    return 0;
}
}  // namespace ustrun

extern "C" int ustrun_bn_bwd_finalize_stat(float* stat, int rows_per_pass, int passes, int C, int64_t count, const float* gamma,
                                           const float* mean, const float* rstd, int64_t aff_stride, float* dgamma, float* dbeta,
                                           int accumulate, float* coef, ustrun_stream_t s) {
    return bn_bwd_finalize_stat(stat, rows_per_pass, passes, C, count, gamma, mean, rstd, (long)aff_stride, dgamma, dbeta, accumulate,
                                coef, (hipStream_t)s);
}

extern "C" int ustrun_bn_bwd_reduce(const void* da, const void* dp, const void* y, const float* scale,
                                    const float* shift, const float* mean, const float* rstd, const float* gamma,
                                    int N, int H, int W, int C, float* dgamma, float* dbeta, int accumulate,
                                    float* coef, float* partials, int64_t partials_bytes, int dtype,
                                    ustrun_stream_t s) {
    return bn_bwd_reduce_passes(da, dp, y, scale, shift, mean, rstd, gamma, N, H, W, C, dgamma, dbeta, accumulate, coef,
                                partials, partials_bytes, dtype, 1, 0, 0, 0, (hipStream_t)s);
}

extern "C" int ustrun_bn_bwd_apply(const void* da, const void* dp, const void* y, const float* scale,
                                   const float* shift, const float* coef, int N, int H, int W, int C, void* dy,
                                   int dtype, ustrun_stream_t s) {
    return bn_bwd_apply_passes(da, dp, y, scale, shift, coef, N, H, W, C, dy, dtype, 1, 0, 0, 0, (hipStream_t)s);
}

extern "C" int ustrun_debug_last_bn_variant(void) { return g_last_bn_variant; }
