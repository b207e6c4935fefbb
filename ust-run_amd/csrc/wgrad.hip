// wgrad.hip -- weight-gradient "TN" GEMM on the exact-f32 matrix cores:
//
//   dW[seg][ci][co] = sum_{p in base grid} A_seg[p][ci] * dY_seg[p][co]
//
// A_seg is the loader's view of the layer input (same on-the-fly BatchNorm/ReLU/pool/concat as the
// forward) at pixel p + tap(seg); dY_seg is the raw output gradient at p (conv3x3) or at
// 2p + (seg/2, seg%2) (ConvTranspose).  The contraction runs over pixels, so both operands are
// consumed in their natural NHWC layout: lanes read consecutive channels of one pixel from LDS.
// Work split: grid = (ci-tile x co-tile x seg) x ksplit; every block (and, for narrow tiles, every
// K-wave inside it) writes its own partial slab, summed in a fixed order by reduce_partials.
#include "common.h"
#include "loader.h"

namespace ustrun {
namespace {

constexpr int KP = 32;  // pixels per stage

template <int TM, int TN, bool POOL>
__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(const WgradArgs a, const int mtn, const int ntn) {
    constexpr int WTM = TM / 64, WTN = TN / 64, KW = 4 / (WTM * WTN);
    constexpr int AQ = TM / 4, APASS = 256 / AQ, AR = KP / APASS;
    constexpr int BQ = TN / 4, BPASS = 256 / BQ, BR = KP / BPASS;
    __shared__ __attribute__((aligned(16))) float As[KP * TM];
    __shared__ __attribute__((aligned(16))) float Bs[KP * TN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kw = wave % KW, wt = wave / KW, wm = wt / WTN, wn = wt % WTN;
    int bid = blockIdx.x;
    const int ntile = bid % ntn; bid /= ntn;
    const int mtile = bid % mtn; bid /= mtn;
    const int seg = bid;
    const int ks = blockIdx.y;
    const int ci0 = mtile * TM, co0 = ntile * TN;

    const int ady = a.d0 + (seg / a.segw) * a.astep, adx = a.d0 + (seg % a.segw) * a.astep;
    const int boy = a.dy_s == 2 ? (seg >> 1) : 0, box = a.dy_s == 2 ? (seg & 1) : 0;
    const long kbeg = (long)ks * a.kchunk;
    const long kend = (kbeg + a.kchunk < a.M) ? kbeg + a.kchunk : a.M;
    const int hw = a.Hb * a.Wb;

    const bool vecA = sources_vectorizable(a.src[0], a.src[1], a.nsrc);
    const bool vecB = (a.Cout & 3) == 0;
    const int a_q = tid % AQ, a_r0 = tid / AQ;
    const int b_q = tid % BQ, b_r0 = tid / BQ;
    const int cg = ci0 + 4 * a_q;
    const bool second = (a.nsrc == 2 && cg >= a.src[0].C);
    const SrcDev S = pick_src(a.src[0], a.src[1], second);
    const int cl = cg - (second ? a.src[0].C : 0);
    f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
    if (vecA && cg < a.Cin && S.scale) { asc = *(const f32x4*)(S.scale + cl); ash = *(const f32x4*)(S.shift + cl); }

    constexpr int NP = POOL ? 4 : 1;
    f32x4 av[AR][NP];
    unsigned aok;
    f32x4 bv[BR];

    auto load_stage = [&](long k0) {
        aok = 0;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const long m = k0 + a_r0 + APASS * i;
            f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            av[i][0] = z4;
            if (m < kend) {
                const int n = (int)(m / hw);
                const int rem = (int)(m - (long)n * hw);
                const int by = rem / a.Wb, bx = rem - by * a.Wb;
                const int iy = (by << a.ashift) + ady, ix = (bx << a.ashift) + adx;
                if (vecA) {
                    const int ly = iy - S.off_y, lx = ix - S.off_x;
                    if (cg < a.Cin && ly >= 0 && ly < S.LH && lx >= 0 && lx < S.LW) {
                        aok |= 1u << i;
                        if (POOL) {
                            const float* p = S.ptr + n * S.sN + (long)(2 * ly) * S.sH + (long)(2 * lx) * S.sW + cl;
                            av[i][0] = *(const f32x4*)p;
                            av[i][1 % NP] = *(const f32x4*)(p + S.sW);
                            av[i][2 % NP] = *(const f32x4*)(p + S.sH);
                            av[i][3 % NP] = *(const f32x4*)(p + S.sH + S.sW);
                        } else {
                            av[i][0] = *(const f32x4*)(S.ptr + n * S.sN + (long)ly * S.sH + (long)lx * S.sW + cl);
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (cg + j < a.Cin) av[i][0][j] = load_elem(a.src[0], a.src[1], a.nsrc, n, iy, ix, cg + j);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const long m = k0 + b_r0 + BPASS * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            const int co = co0 + 4 * b_q;
            if (m < kend && co < a.Cout) {
                const int n = (int)(m / hw);
                const int rem = (int)(m - (long)n * hw);
                const int by = rem / a.Wb, bx = rem - by * a.Wb;
                const int oy = by * a.dy_s + boy, ox = bx * a.dy_s + box;
                const float* p = a.dy + (((long)n * a.dyH + oy) * a.dyW + ox) * a.Cout + co;
                if (vecB) v = *(const f32x4*)p;
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (co + j < a.Cout) v[j] = p[j];
                }
            }
            bv[i] = v;
        }
    };

    auto write_stage = [&]() {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            f32x4 v = av[i][0];
            if (vecA) {
                v = v * asc + ash;
                if (S.relu) v = relu4(v);
                if (POOL) {
#pragma unroll
                    for (int q = 1; q < NP; ++q) {
                        f32x4 t = av[i][q] * asc + ash;
                        if (S.relu) t = relu4(t);
                        v = max4(v, t);
                    }
                }
                if (!((aok >> i) & 1u)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            *(f32x4*)(As + (a_r0 + APASS * i) * TM + 4 * a_q) = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) *(f32x4*)(Bs + (b_r0 + BPASS * i) * TN + 4 * b_q) = bv[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    const float* Ap = As + lh * TM + wm * 64 + l31;
    const float* Bp = Bs + lh * TN + wn * 64 + l31;

    if (kbeg < kend) load_stage(kbeg);
    for (long k0 = kbeg; k0 < kend; k0 += KP) {
        write_stage();
        __syncthreads();
        if (k0 + KP < kend) load_stage(k0 + KP);
#pragma unroll
        for (int t = 0; t < KP / 2 / KW; ++t) {
            const int kk = t * KW + kw;
            const float a0 = Ap[2 * kk * TM], a1 = Ap[2 * kk * TM + 32];
            const float b0 = Bp[2 * kk * TN], b1 = Bp[2 * kk * TN + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }

    // slab (ks*KW + kw): [nseg][Cin][Cout]
    float* slab = a.partials + ((long)(ks * KW + kw) * a.nseg + seg) * a.Cin * a.Cout;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = co0 + wn * 64 + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (ci < a.Cin && co < a.Cout) slab[(long)ci * a.Cout + co] = acc[i][j][r];
            }
    }
}

// v + slab[k0][i] + .. + slab[k1-1][i], added in that order, eight slabs' loads issued before the first add (one slab per trip
// left every 16-byte load waited for on its own: these kernels are 7-11 us of memory round trips, 33 launches per step)
__device__ inline f32x4 sum_slabs(f32x4 v, const float* part, int k0, int k1, long sstride, long i) {
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = ((const f32x4*)(part + (long)(k + u) * sstride))[i];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    if (k + 4 <= k1) {
        f32x4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = ((const f32x4*)(part + (long)(k + u) * sstride))[i];
#pragma unroll
        for (int u = 0; u < 4; ++u) v += t[u];
        k += 4;
    }
    for (; k < k1; ++k) v += ((const f32x4*)(part + (long)k * sstride))[i];
    return v;
}

// In-place pre-reduction of long slab lists: group g (blockIdx.y) sums slabs [g*GRP, g*GRP+GRP) into
// slab g*GRP.  Every thread reads and writes only its own elements, so no ordering is needed.
constexpr int GRP = 16;
__global__ __launch_bounds__(256) void reduce_groups_kernel(float* part, int nslab, long total, long sstride) {
    const int k0 = blockIdx.y * GRP, k1 = min(nslab, k0 + GRP);
    const long n4 = total / 4;
    float* base = part + (long)k0 * sstride;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = ((const f32x4*)base)[i];
        v = sum_slabs(v, part, k0 + 1, k1, sstride, i);
        ((f32x4*)base)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (total & 3)) {
        const long e = n4 * 4 + threadIdx.x;
        float v = 0.f;
        for (int k = k0; k < k1; ++k) v += part[(long)k * sstride + e];
        base[e] = v;
    }
}

// out (+)= sum_k partials[k*sstride ...]  (same layout, 16 B per lane)
__global__ __launch_bounds__(256) void reduce_plain_kernel(const float* __restrict__ part, int nslab, long total, long sstride,
                                                          float* __restrict__ out, int accumulate) {
    const long n4 = total / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = ((const f32x4*)part)[i];
        v = sum_slabs(v, part, 1, nslab, sstride, i);
        if (accumulate) v += ((f32x4*)out)[i];
        ((f32x4*)out)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (total & 3)) {
        const long e = n4 * 4 + threadIdx.x;
        float v = 0.f;
        for (int k = 0; k < nslab; ++k) v += part[(long)k * sstride + e];
        out[e] = accumulate ? out[e] + v : v;
    }
}

// partials [k][seg][ci][co] -> torch conv layout [co][ci][seg] (layout 0) or convT layout [ci][co][seg]
// (layout 1): one 32(ci) x 32(co) x nseg tile per block goes through LDS so that both the reads (along co)
// and the writes (along the destination's inner dimensions) are coalesced.
__global__ __launch_bounds__(256) void reduce_transpose_kernel(const float* __restrict__ part, int nslab, long sstride, int nseg,
                                                              int Cin, int Cout, float* __restrict__ out, int layout,
                                                              int accumulate) {
    __shared__ float tile[9][32][33];
    const int cit = blockIdx.x % ((Cin + 31) / 32), cot = blockIdx.x / ((Cin + 31) / 32);
    const int ci0 = cit * 32, co0 = cot * 32;
    const long total = (long)nseg * Cin * Cout;
    for (int e = threadIdx.x; e < nseg * 32 * 32; e += 256) {
        const int co = e & 31, ci = (e >> 5) & 31, seg = e >> 10;
        float v = 0.f;
        if (ci0 + ci < Cin && co0 + co < Cout) {
            const long idx = ((long)seg * Cin + ci0 + ci) * Cout + co0 + co;
            for (int k = 0; k < nslab; ++k) v += part[(long)k * sstride + idx];
        }
        tile[seg][ci][co] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < nseg * 32 * 32; e += 256) {
        int seg, ci, co;
        if (layout == 0) { seg = e % nseg; ci = (e / nseg) & 31; co = e / (nseg * 32); }   // [co][ci][seg]
        else { seg = e % nseg; co = (e / nseg) & 31; ci = e / (nseg * 32); }               // [ci][co][seg]
        if (ci0 + ci < Cin && co0 + co < Cout) {
            const long o = layout == 0 ? ((long)(co0 + co) * Cin + ci0 + ci) * nseg + seg
                                       : ((long)(ci0 + ci) * Cout + co0 + co) * nseg + seg;
            const float v = tile[seg][ci][co];
            out[o] = accumulate ? out[o] + v : v;
        }
    }
}

template <int TM, int TN, bool POOL>
int launch_cfg(const WgradArgs& a, hipStream_t st) {
    const int mtn = cdiv(a.Cin, TM), ntn = cdiv(a.Cout, TN);
    dim3 grid(mtn * ntn * a.nseg, a.ksplit), block(256);
    hipLaunchKernelGGL((wgrad_f32_kernel<TM, TN, POOL>), grid, block, 0, st, a, mtn, ntn);
    USTRUN_LAUNCH_CHECK("wgrad");
    return 0;
}

}  // namespace

int wgrad_plan(int nseg, int Cin, int Cout, int64_t M, int* ksplit, long* kchunk, int* slabs) {
    const int TM = Cin > 64 ? 128 : 64, TN = Cout > 64 ? 128 : 64;
    const int KW = 4 / ((TM / 64) * (TN / 64));
    const long tiles = (long)cdiv(Cin, TM) * cdiv(Cout, TN) * nseg;
    long ks = (1024 + tiles - 1) / tiles;          // aim at ~4 blocks per CU
    const long maxks = (M + 4 * KP - 1) / (4 * KP); // at least 4 stages per block
    if (ks > maxks) ks = maxks;
    if (ks < 1) ks = 1;
    long chunk = ((M + ks - 1) / ks + 63) / 64 * 64;   // multiple of every kernel's stage depth
    ks = (M + chunk - 1) / chunk;
    *ksplit = (int)ks; *kchunk = chunk; *slabs = (int)ks * KW;
    return 0;
}

int wgrad_launch(const WgradArgs& a, int dtype, hipStream_t st) {
    USTRUN_CHECK(dtype_ok(dtype), "wgrad: dtype %d not built", dtype);
    USTRUN_CHECK(a.M > 0 && a.Cin > 0 && a.Cout > 0, "wgrad: empty problem");
    set_last_wgrad_variant(0);
    int csum = 0;
    for (int i = 0; i < a.nsrc; ++i) csum += a.src[i].C;
    USTRUN_CHECK(csum == a.Cin, "wgrad: source channels %d != Cin %d", csum, a.Cin);
    const bool pool = a.src[0].pool != 0;
    USTRUN_CHECK(!pool || a.nsrc == 1, "wgrad: pooled source cannot be concatenated");
    USTRUN_CHECK(a.kchunk % 64 == 0, "wgrad: K chunk must be a multiple of 64");
    for (int i = 0; i < a.nsrc; ++i)
        USTRUN_CHECK(a.src[i].gN == 0, "wgrad: batched passes reached a kernel without per-pass BatchNorm constants");
    const bool m128 = a.Cin > 64, n128 = a.Cout > 64;
    double in_elems = 0;
    for (int i = 0; i < a.nsrc; ++i) in_elems += (double)a.N * a.src[i].H * a.src[i].W * a.src[i].C;
    prof_begin(1, 2.0 * a.M * a.nseg * a.Cin * a.Cout,
               (dtype == USTRUN_D16 ? 2.0 : 4.0) * (in_elems + (double)a.N * a.dyH * a.dyW * a.Cout) + 4.0 * a.nseg * a.Cin * a.Cout, st);
    struct End { hipStream_t s; ~End() { prof_end(s); } } end_{st};
    if (dtype == USTRUN_D16) return wgrad_launch_bf16(a, st);
    if (pool) {
        if (m128 && n128) return launch_cfg<128, 128, true>(a, st);
        if (m128) return launch_cfg<128, 64, true>(a, st);
        if (n128) return launch_cfg<64, 128, true>(a, st);
        return launch_cfg<64, 64, true>(a, st);
    }
    if (m128 && n128) return launch_cfg<128, 128, false>(a, st);
    if (m128) return launch_cfg<128, 64, false>(a, st);
    if (n128) return launch_cfg<64, 128, false>(a, st);
    return launch_cfg<64, 64, false>(a, st);
}

int reduce_partials(const float* partials, int nslab, int nseg, int Cin, int Cout, float* out, int layout,
                    int accumulate, hipStream_t st) {
    const long total = (long)nseg * Cin * Cout;
    int eblocks = cdiv(total / 4 + 1, 256);
    if (eblocks > 2048) eblocks = 2048;
    long sstride = total;
    // the transposing pass runs one block per 32 x 32 channel tile and walks the slab list serially: with few tiles (<= 256:
    // a 256 -> 256 layer has 64) fold the list down to one slab first, on a full grid
    const bool fold_all = layout != 2 && (long)cdiv(Cin, 32) * cdiv(Cout, 32) <= 256 && nslab > 2;
    while (nslab > (fold_all ? 1 : GRP)) {     // parallel fixed-order tree over long slab lists (in place)
        const int groups = cdiv(nslab, GRP);
        hipLaunchKernelGGL(reduce_groups_kernel, dim3(eblocks, groups), dim3(256), 0, st, const_cast<float*>(partials), nslab,
                           total, sstride);
        USTRUN_LAUNCH_CHECK("reduce_groups");
        nslab = groups; sstride *= GRP;
    }
    if (layout == 2) {
        hipLaunchKernelGGL(reduce_plain_kernel, dim3(eblocks), dim3(256), 0, st, partials, nslab, total, sstride, out, accumulate);
    } else {
        USTRUN_CHECK(nseg <= 9, "reduce_partials: nseg %d > 9", nseg);
        hipLaunchKernelGGL(reduce_transpose_kernel, dim3(cdiv(Cin, 32) * cdiv(Cout, 32)), dim3(256), 0, st, partials, nslab,
                           sstride, nseg, Cin, Cout, out, layout, accumulate);
    }
    USTRUN_LAUNCH_CHECK("reduce_partials");
    return 0;
}

}  // namespace ustrun
