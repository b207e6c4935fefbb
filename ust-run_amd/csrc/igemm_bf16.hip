// igemm_bf16.hip -- the implicit-GEMM kernel of igemm.hip on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, f32 accumulate).  Activations stay f32 in HBM this round: the loader
// applies the BatchNorm/ReLU/pool/concat transform in f32, rounds to bf16 (RNE) and stores a
// K-contiguous, XOR-swizzled A tile in LDS; weights are pre-packed in bf16 as [slice][K/8][N][8] so
// that a lane's 8 consecutive K values are one 16-byte LDS read with no conflicts.
#include "common.h"
#include "loader.h"

namespace ustrun {
namespace {

typedef __attribute__((ext_vector_type(8))) elt_t bf16x8;
typedef __attribute__((ext_vector_type(4))) elt_t bf16x4;

struct RowInfo { int n; int yx; };

template <int WM, int WN, bool POOL>
__global__ __launch_bounds__(256, 2) void igemm_bf16_kernel(const IgemmArgs a, const int mt_total, const int nt_total) {
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int BK = POOL ? 32 : 64;
    constexpr int CPR = BK / 4;            // 4-channel groups per row
    constexpr int RPP = 256 / CPR;         // rows per pass
    constexpr int AR = BM / RPP;           // rows per thread per stage
    constexpr int NP = POOL ? 4 : 1;
    constexpr int ROWB = BK * 2;           // bytes per A row in LDS
    constexpr int BCH = (BK / 8) * BN;     // 16-byte chunks in the B tile
    constexpr int BR = BCH / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                        // [BM][BK] bf16, 16-B chunks XOR-swizzled by row
    char* Bs = smem + BM * ROWB;            // [BK/8][BN][8] bf16
    RowInfo* rowinfo = (RowInfo*)(Bs + BCH * 16);

    const int ntiles = mt_total * nt_total;
    int bid = blockIdx.x;
    {
        const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int mtile = bid / nt_total, ntile = bid % nt_total;
    const int z = blockIdx.y;
    const int n0 = ntile * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    for (int r = tid; r < BM; r += 256) {
        long m = (long)mtile * BM + r;
        RowInfo ri;
        if (m < a.M) {
            int hw = a.Hb * a.Wb;
            int n = (int)(m / hw);
            int rem = (int)(m - (long)n * hw);
            int by = rem / a.Wb;
            ri.n = n; ri.yx = (by << 16) | (rem - by * a.Wb);
        } else { ri.n = -1; ri.yx = 0; }
        rowinfo[r] = ri;
    }
    __syncthreads();

    auto swz = [](int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

    const int a_c4 = tid % CPR, a_r0 = tid / CPR;
    const bool vecA = sources_vectorizable(a.src[0], a.src[1], a.nsrc, 2);
    const int nchunk = (a.Cin + BK - 1) / BK;
    const int nstage = a.nseg * nchunk;
    const int K8 = (a.Cin + 7) / 8;         // packed K octets per slice
    const elt_t* Wp = (const elt_t*)a.W;

    f32x4 av[AR][NP];
    f32x4 asc, ash;
    unsigned aok;
    int a_relu;
    bf16x8 bv[BR];

    auto load_stage = [&](int s) {
        const int seg = s / nchunk, c0 = (s - seg * nchunk) * BK;
        const int dy = a.d0 + (seg / a.segw) * a.dstep, dx = a.d0 + (seg % a.segw) * a.dstep;
        const int cg = c0 + 4 * a_c4;
        aok = 0;
        asc = (f32x4){1.f, 1.f, 1.f, 1.f}; ash = (f32x4){0.f, 0.f, 0.f, 0.f}; a_relu = 0;
        if (vecA) {
            const bool second = (a.nsrc == 2 && cg >= a.src[0].C);
            const SrcDev S = pick_src(a.src[0], a.src[1], second);
            const int cl = cg - (second ? a.src[0].C : 0);
            const bool cok = cg < a.Cin;
            if (cok && S.scale) { asc = *(const f32x4*)(S.scale + cl); ash = *(const f32x4*)(S.shift + cl); }
            a_relu = S.relu;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                RowInfo ri = rowinfo[a_r0 + RPP * i];
                const int ly = (ri.yx >> 16) * a.s_in + dy - S.off_y;
                const int lx = (ri.yx & 0xffff) * a.s_in + dx - S.off_x;
                const bool ok = cok && ri.n >= 0 && ly >= 0 && ly < S.LH && lx >= 0 && lx < S.LW;
                if (ok) {
                    aok |= 1u << i;
                    if (POOL) {
                        const long p = ri.n * S.sN + (long)(2 * ly) * S.sH + (long)(2 * lx) * S.sW + cl;
                        av[i][0] = ld4t<2>(S.ptr, p);
                        av[i][1 % NP] = ld4t<2>(S.ptr, p + S.sW);
                        av[i][2 % NP] = ld4t<2>(S.ptr, p + S.sH);
                        av[i][3 % NP] = ld4t<2>(S.ptr, p + S.sH + S.sW);
                    } else {
                        av[i][0] = ld4t<2>(S.ptr, ri.n * S.sN + (long)ly * S.sH + (long)lx * S.sW + cl);
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                RowInfo ri = rowinfo[a_r0 + RPP * i];
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ri.n >= 0) {
                    const int iy = (ri.yx >> 16) * a.s_in + dy, ix = (ri.yx & 0xffff) * a.s_in + dx;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (cg + j < a.Cin) v[j] = load_elem(a.src[0], a.src[1], a.nsrc, ri.n, iy, ix, cg + j);
                }
                av[i][0] = v;
            }
        }
        // B tile: packed bf16 weights [slice][K8][Cout][8]
        const elt_t* wb = Wp + ((long)(seg + z) * K8) * a.Cout * 8;
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int idx = tid + 256 * i, o = idx / BN, n = idx % BN;
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (elt_t)0.f;
            const int ko = c0 / 8 + o;
            if (ko < K8 && n0 + n < a.Cout) v = *(const bf16x8*)(wb + ((long)ko * a.Cout + n0 + n) * 8);
            bv[i] = v;
        }
    };

    auto write_stage = [&]() {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            f32x4 v = av[i][0];
            if (vecA) {
                v = v * asc + ash;
                if (a_relu) v = relu4(v);
                if (POOL) {
#pragma unroll
                    for (int q = 1; q < NP; ++q) {
                        f32x4 t = av[i][q] * asc + ash;
                        if (a_relu) t = relu4(t);
                        v = max4(v, t);
                    }
                }
                if (!((aok >> i) & 1u)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const int row = a_r0 + RPP * i;
            bf16x4 h;
            h[0] = (elt_t)v[0]; h[1] = (elt_t)v[1]; h[2] = (elt_t)v[2]; h[3] = (elt_t)v[3];
            *(bf16x4*)(As + row * ROWB + (((a_c4 >> 1) ^ swz(row)) * 16) + (a_c4 & 1) * 8) = h;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i) *(bf16x8*)(Bs + (tid + 256 * i) * 16) = bv[i];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int l31 = lane & 31, lh = lane >> 5;
    const int rowA = wm * 64 + l31;
    const char* Ap0 = As + rowA * ROWB;
    const char* Ap1 = As + (rowA + 32) * ROWB;
    const int sw0 = swz(rowA), sw1 = swz(rowA + 32);
    const char* Bp = Bs + (lh * BN + wn * 64 + l31) * 16;

    load_stage(0);
    for (int s = 0; s < nstage; ++s) {
        write_stage();
        __syncthreads();
        if (s + 1 < nstage) load_stage(s + 1);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int ch = 2 * ks + lh;
            const bf16x8 a0 = *(const bf16x8*)(Ap0 + ((ch ^ sw0) * 16));
            const bf16x8 a1 = *(const bf16x8*)(Ap1 + ((ch ^ sw1) * 16));
            const bf16x8 b0 = *(const bf16x8*)(Bp + (2 * ks * BN) * 16);
            const bf16x8 b1 = *(const bf16x8*)(Bp + (2 * ks * BN + 32) * 16);
            acc[0][0] = USTRUN_MFMA_32x32x16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = USTRUN_MFMA_32x32x16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = USTRUN_MFMA_32x32x16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = USTRUN_MFMA_32x32x16(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue (identical to the f32 kernel: f32 outputs, BN-statistics partials) ----
    const int oyz = z >> 1, oxz = z & 1;
    const int C1 = a.Cout - a.C0;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const bool cok = col < a.Cout;
        const float bias = (a.bias && cok) ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const RowInfo ri = rowinfo[row];
                if (ri.n >= 0 && cok) {
                    const bool o32 = a.out_esz == 4;                  // (f32 outputs: the DeepLabV2 classifier maps)
                    const float v = o32 ? acc[i][j][r] + bias : rndt<2>(acc[i][j][r] + bias);   // statistics see the stored value
                    const int oy = (ri.yx >> 16) * a.s_out + oyz, ox = (ri.yx & 0xffff) * a.s_out + oxz;
                    if (col < a.C0) {
                        const long oi = (((long)ri.n * a.Ho + oy) * a.Wo + ox) * a.C0 + col;
                        if (o32) st1t<4>(a.out0, oi, v); else st1t<2>(a.out0, oi, v);
                    } else {
                        const int y1 = oy - a.o1y, x1 = ox - a.o1x;
                        if (y1 >= 0 && y1 < a.H1 && x1 >= 0 && x1 < a.W1)
                            st1t<2>(a.out1, (((long)ri.n * a.H1 + y1) * a.W1 + x1) * C1 + (col - a.C0), v);
                    }
                    s1[j] += v; s2[j] += v * v;
                }
            }
        }
    }
    if (a.stat) {
        float* red = (float*)As;   // [WM][2][BN]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            s1[j] += __shfl_xor(s1[j], 32);
            s2[j] += __shfl_xor(s2[j], 32);
            if (lh == 0) {
                red[(wm * 2 + 0) * BN + wn * 64 + j * 32 + l31] = s1[j];
                red[(wm * 2 + 1) * BN + wn * 64 + j * 32 + l31] = s2[j];
            }
        }
        __syncthreads();
        constexpr int HALVES = WM / 2;
        const int stat_rows = (int)((a.M + 127) / 128);
        for (int t = tid; t < HALVES * 2 * BN; t += 256) {
            const int h = t / (2 * BN), q = (t / BN) % 2, c = t % BN;
            const float v = red[((2 * h) * 2 + q) * BN + c] + red[((2 * h + 1) * 2 + q) * BN + c];
            const int srow = mtile * HALVES + h;
            if (srow < stat_rows && n0 + c < a.Cout) a.stat[((long)srow * 2 + q) * a.Cout + n0 + c] = v;
        }
    }
}

template <int WM, int WN, bool POOL>
int launch_cfg(const IgemmArgs& a, hipStream_t st) {
    constexpr int BM = WM * 64, BN = WN * 64, BK = POOL ? 32 : 64;
    const int mt = cdiv(a.M, BM), nt = cdiv(a.Cout, BN);
    const size_t lds = (size_t)BM * BK * 2 + (size_t)BK * BN * 2 + (size_t)BM * sizeof(RowInfo);
    dim3 grid(mt * nt, a.nz), block(256);
    hipLaunchKernelGGL((igemm_bf16_kernel<WM, WN, POOL>), grid, block, lds, st, a, mt, nt);
    USTRUN_LAUNCH_CHECK("igemm_bf16");
    return 0;
}

// torch conv weight [Cout][Cin][taps] (taps = 9) or convT weight [Cin][Cout][taps] (taps = 4)
// -> fwd [tap][ceil(Cin/8)][Cout][8] and dgrad [tap][ceil(Cout/8)][Cin][8], bf16, zero padded
__global__ void pack_bf16_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, int transposed_src,
                                 elt_t* __restrict__ wf, elt_t* __restrict__ wd) {
    const int Ki = (Cin + 7) / 8, Ko = (Cout + 7) / 8;
    const long nf = (long)taps * Ki * Cout * 8, nd = (long)taps * Ko * Cin * 8;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < nf + nd; e += (long)gridDim.x * blockDim.x) {
        int tap, ci, co;
        if (e < nf) {
            const int j = (int)(e & 7); long t = e >> 3;
            co = (int)(t % Cout); t /= Cout;
            ci = (int)(t % Ki) * 8 + j; tap = (int)(t / Ki);
        } else {
            const long f = e - nf;
            const int j = (int)(f & 7); long t = f >> 3;
            ci = (int)(t % Cin); t /= Cin;
            co = (int)(t % Ko) * 8 + j; tap = (int)(t / Ko);
        }
        float v = 0.f;
        if (ci < Cin && co < Cout)
            v = transposed_src ? w[((long)ci * Cout + co) * taps + tap] : w[((long)co * Cin + ci) * taps + tap];
        if (e < nf) wf[e] = (elt_t)v; else if (wd) wd[e - nf] = (elt_t)v;
    }
}

}  // namespace

int igemm_launch_bf16(const IgemmArgs& a, hipStream_t st) {
    bool pool = false;
    for (int i = 0; i < a.nsrc; ++i) pool |= a.src[i].pool != 0;
    if (a.Cout > 64 || pool) return pool ? launch_cfg<2, 2, true>(a, st) : launch_cfg<2, 2, false>(a, st);
    return launch_cfg<4, 1, false>(a, st);
}

// all layers of a model in one launch: blockIdx.y = layer.  A thread owns one (co, ci) pair of the K/N-padded grid:
// it reads the pair's `taps` consecutive source floats (threads run along the source's inner index: coalesced) and
// scatters them to the tap planes of both packed layouts.  (One thread per OUTPUT element re-fetched every source line
// once per tap: 1.3 GB moved for 250 MB.)
__global__ __launch_bounds__(256) void pack_bf16_multi_kernel(const PackJobs jobs) {
    const PackJob j = jobs.j[blockIdx.y];
    const float* __restrict__ w = j.w;
    elt_t* wf = (elt_t*)j.wf;
    elt_t* wd = (elt_t*)j.wd;
    const int Cin = j.Cin, Cout = j.Cout, taps = j.taps;
    // 3x3 conv weights with whole 64-channel input groups (every layer but the first): a block takes 8 output channels
    // x 64 input channels -- 8 contiguous 2304-byte runs of the source -- through LDS and writes both layouts in
    // 16-byte pieces that add up to whole 128-byte (forward) and 1 KB (input-gradient) runs.  The element-per-thread
    // path below wrote 2-byte pieces 16 bytes apart: 1.3 GB of HBM traffic for 248 MB of weights.
    if (!j.transposed_src && taps == 9 && Cin % 64 == 0 && Cout % 8 == 0 && wd) {
        typedef __attribute__((ext_vector_type(8))) elt_t bf16x8;
        __shared__ elt_t t[8][64 * 9 + 2];                 // [co][ci*9 + tap] (+2: rows start 4 bytes apart in the banks)
        const int Ki = Cin / 8, Ko = Cout / 8, tiles_ci = Cin / 64, ntile = Ko * tiles_ci;
        for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
            const int cog = tile / tiles_ci, ci0 = (tile % tiles_ci) * 64;
            __syncthreads();
            for (int e = threadIdx.x; e < 8 * 144; e += 256) {          // 144 float4 per output channel
                const int co = e / 144, q = e % 144;
                const f32x4 v = *(const f32x4*)(w + ((long)(cog * 8 + co) * Cin + ci0) * 9 + 4 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k) t[co][4 * q + k] = (elt_t)v[k];
            }
            __syncthreads();
            for (int e = threadIdx.x; e < 9 * 8 * 8; e += 256) {        // forward: (tap, ci group, co) -> 8 ci
                const int co = e & 7, cig = (e >> 3) & 7, tap = e >> 6;
                bf16x8 v;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = t[co][(cig * 8 + k) * 9 + tap];
                *(bf16x8*)(wf + (((long)tap * Ki + ci0 / 8 + cig) * Cout + cog * 8 + co) * 8) = v;
            }
            for (int e = threadIdx.x; e < 9 * 64; e += 256) {           // input gradient: (tap, ci) -> 8 co
                const int ci = e & 63, tap = e >> 6;
                bf16x8 v;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = t[k][ci * 9 + tap];
                *(bf16x8*)(wd + (((long)tap * Ko + cog) * Cin + ci0 + ci) * 8) = v;
            }
        }
        return;
    }
    const int Ki = (Cin + 7) / 8, Ko = (Cout + 7) / 8;
    const int Pi = Ki * 8, Po = Ko * 8;
    const long total = (long)Pi * Po;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        int ci, co;
        if (j.transposed_src) { co = (int)(e % Po); ci = (int)(e / Po); }       // source [Cin][Cout][taps]
        else { ci = (int)(e % Pi); co = (int)(e / Pi); }                        // source [Cout][Cin][taps]
        const bool in = ci < Cin && co < Cout;
        const float* src = w + (j.transposed_src ? ((long)ci * Cout + co) : ((long)co * Cin + ci)) * taps;
        for (int tap = 0; tap < taps; ++tap) {
            const elt_t v = (elt_t)(in ? src[tap] : 0.f);
            if (co < Cout) wf[(((long)tap * Ki + ci / 8) * Cout + co) * 8 + (ci & 7)] = v;
            if (wd && ci < Cin) wd[(((long)tap * Ko + co / 8) * Cin + ci) * 8 + (co & 7)] = v;
        }
    }
}

int pack_bf16_multi(const PackJobs& jobs, int n, hipStream_t st) {
    hipLaunchKernelGGL(pack_bf16_multi_kernel, dim3(1024, n), dim3(256), 0, st, jobs);   // small layers: most blocks exit at once
    USTRUN_LAUNCH_CHECK("pack_bf16_multi");
    return 0;
}

int pack_bf16(const float* w, int Cout, int Cin, int taps, int transposed_src, void* wf, void* wd, hipStream_t st) {
    const long total = (long)taps * (((Cin + 7) / 8) * (long)Cout + ((Cout + 7) / 8) * (long)Cin) * 8;
    long b = (total + 1023) / 1024;
    if (b > 4096) b = 4096;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3((int)b), dim3(256), 0, st, w, Cout, Cin, taps, transposed_src, (elt_t*)wf,
                       (elt_t*)wd);
    USTRUN_LAUNCH_CHECK("pack_bf16");
    return 0;
}

}  // namespace ustrun
