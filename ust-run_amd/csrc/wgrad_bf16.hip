// wgrad_bf16.hip -- the weight-gradient "TN" GEMM of wgrad.hip on the bf16 matrix cores.
// The contraction runs over pixels while both operands are stored pixel-major (NHWC), so each lane
// needs 8 consecutive PIXELS of one channel: the LDS tiles keep the natural [pixel][channel] layout
// (coalesced fills, 8-byte stores) and the fragments are fetched with the transposing LDS read
// ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group, delivered column-major).
#include "common.h"
#include "loader.h"

namespace ustrun {
namespace {

typedef __attribute__((ext_vector_type(8))) elt_t bf16x8;
typedef __attribute__((ext_vector_type(4))) elt_t bf16x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;


// 64-byte segments of a row are XOR-permuted by the row index so that the 4 rows of a transposed
// read fall on different bank segments
template <int RB> __device__ __forceinline__ int seg_swz(int row) { return RB >= 256 ? (row & 3) : ((row >> 1) & 1); }

template <int RB> __device__ __forceinline__ bf16x8 tr_frag(const char* tile, int k0, int col0, int lane) {
    // lane l of the wave: rows k0 + 8*(l>>5) + {0..3 | 4..7}, columns col0 + 16*((l>>4)&1) + ...
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int colb = (col0 + 16 * ((lane >> 4) & 1) + 4 * p) * 2;
    const int r0 = k0 + 8 * (lane >> 5) + q, r1 = r0 + 4;
    const char* a0 = tile + r0 * RB + (colb ^ (seg_swz<RB>(r0) << 6));
    const char* a1 = tile + r1 * RB + (colb ^ (seg_swz<RB>(r1) << 6));
    const bf16x4 lo = USTRUN_DS_READ_TR16((lds_bf16x4*)a0);
    const bf16x4 hi = USTRUN_DS_READ_TR16((lds_bf16x4*)a1);
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

template <int TM, int TN, bool POOL>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const WgradArgs a, const int mtn, const int ntn) {
    constexpr int WTM = TM / 64, WTN = TN / 64, KW = 4 / (WTM * WTN);
    constexpr int KP = 16 * (KW > 2 ? KW : 2);   // pixels per stage: every K-wave owns >= one 16-deep MFMA step
    constexpr int AQ = TM / 4, APASS = 256 / AQ, AR = KP / APASS;
    constexpr int BQ = TN / 4, BPASS = 256 / BQ, BR = KP / BPASS;
    constexpr int RBA = TM * 2, RBB = TN * 2;
    constexpr int NP = POOL ? 4 : 1;
    __shared__ __attribute__((aligned(16))) char As[KP * RBA];
    __shared__ __attribute__((aligned(16))) char Bs[KP * RBB];

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

    const bool vecA = sources_vectorizable(a.src[0], a.src[1], a.nsrc, 2);
    const bool vecB = (a.Cout & 3) == 0;
    const int a_q = tid % AQ, a_r0 = tid / AQ;
    const int b_q = tid % BQ, b_r0 = tid / BQ;
    const int cg = ci0 + 4 * a_q;
    const bool second = (a.nsrc == 2 && cg >= a.src[0].C);
    const SrcDev S = pick_src(a.src[0], a.src[1], second);
    const int cl = cg - (second ? a.src[0].C : 0);
    f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
    if (vecA && cg < a.Cin && S.scale) { asc = *(const f32x4*)(S.scale + cl); ash = *(const f32x4*)(S.shift + cl); }

    f32x4 av[AR][NP];
    unsigned aok;
    f32x4 bv[BR];

    auto load_stage = [&](long k0) {
        aok = 0;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const long m = k0 + a_r0 + APASS * i;
            av[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
                            const long p = n * S.sN + (long)(2 * ly) * S.sH + (long)(2 * lx) * S.sW + cl;
                            av[i][0] = ld4t<2>(S.ptr, p);
                            av[i][1 % NP] = ld4t<2>(S.ptr, p + S.sW);
                            av[i][2 % NP] = ld4t<2>(S.ptr, p + S.sH);
                            av[i][3 % NP] = ld4t<2>(S.ptr, p + S.sH + S.sW);
                        } else {
                            av[i][0] = ld4t<2>(S.ptr, n * S.sN + (long)ly * S.sH + (long)lx * S.sW + cl);
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
                const long p = (((long)n * a.dyH + oy) * a.dyW + ox) * a.Cout + co;
                if (vecB) v = ld4t<2>(a.dy, p);
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (co + j < a.Cout) v[j] = ld1(a.dy, p + j, a.dy_esz);
                }
            }
            bv[i] = v;
        }
    };

    auto to_bf16 = [](f32x4 v) {
        bf16x4 h;
        h[0] = (elt_t)v[0]; h[1] = (elt_t)v[1]; h[2] = (elt_t)v[2]; h[3] = (elt_t)v[3];
        return h;
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
            const int row = a_r0 + APASS * i;
            *(bf16x4*)(As + row * RBA + ((a_q * 8) ^ (seg_swz<RBA>(row) << 6))) = to_bf16(v);
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int row = b_r0 + BPASS * i;
            *(bf16x4*)(Bs + row * RBB + ((b_q * 8) ^ (seg_swz<RBB>(row) << 6))) = to_bf16(bv[i]);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kbeg < kend) load_stage(kbeg);
    for (long k0 = kbeg; k0 < kend; k0 += KP) {
        write_stage();
        __syncthreads();
        if (k0 + KP < kend) load_stage(k0 + KP);
#pragma unroll
        for (int t = 0; t < KP / 16 / KW; ++t) {
            const int kk = t * KW + kw;          // 16-pixel step owned by this wave
            const bf16x8 a0 = tr_frag<RBA>(As, kk * 16, wm * 64, lane);
            const bf16x8 a1 = tr_frag<RBA>(As, kk * 16, wm * 64 + 32, lane);
            const bf16x8 b0 = tr_frag<RBB>(Bs, kk * 16, wn * 64, lane);
            const bf16x8 b1 = tr_frag<RBB>(Bs, kk * 16, wn * 64 + 32, lane);
            acc[0][0] = USTRUN_MFMA_32x32x16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = USTRUN_MFMA_32x32x16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = USTRUN_MFMA_32x32x16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = USTRUN_MFMA_32x32x16(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }

    float* slab = a.partials + ((long)(ks * KW + kw) * a.nseg + seg) * a.Cin * a.Cout;
    const int l31 = lane & 31, lh = lane >> 5;
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

template <int TM, int TN, bool POOL>
int launch_cfg(const WgradArgs& a, hipStream_t st) {
    const int mtn = cdiv(a.Cin, TM), ntn = cdiv(a.Cout, TN);
    dim3 grid(mtn * ntn * a.nseg, a.ksplit), block(256);
    hipLaunchKernelGGL((wgrad_bf16_kernel<TM, TN, POOL>), grid, block, 0, st, a, mtn, ntn);
    USTRUN_LAUNCH_CHECK("wgrad_bf16");
    return 0;
}

}  // namespace

int wgrad_launch_bf16(const WgradArgs& a, hipStream_t st) {
    const bool pool = a.src[0].pool != 0;
    const bool m128 = a.Cin > 64, n128 = a.Cout > 64;
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

}  // namespace ustrun
