// x3.hip -- dtype USTRUN_F32X3: f32 tensors everywhere (the layouts, loaders and epilogues of the exact-f32 path), but the 3x3
// convolutions' products run on the BF16 matrix cores with every operand split into three bf16 terms:
//
//     x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)          (24 significand bits in all)
//     a b  ~  a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)                              (six MFMAs, f32 accumulate)
//
// The three dropped products are below 2^-24 |a b|, i.e. at the rounding of one f32 multiply: the result sits within f32 summation
// noise of the v_mfma_f32_32x32x2_f32 path (tests: the reference's full-size logits to 1e-4, arg-max bit-exact outside 1e-4 margins
// -- north_star's tolerance) at 16 / 6 of its matrix rate.  The split is arithmetic, not storage: an f32 value is split when it is
// staged into LDS (after BatchNorm + ReLU were applied in f32), the weights once per optimizer step (ustrun_pack_*: the three
// planes sit behind the f32 pack).
//
//   igemm_x3_kernel   forward / input gradient of the 3x3 convolutions (and the ConvTranspose pair): the generic implicit GEMM of
//                     igemm.hip -- 128 x 128 tile, one (tap, 32-channel chunk) per stage, loader and epilogue unchanged -- with
//                     the activation planes pixel-major in LDS (80-byte rows: the A operand's eight consecutive channels are one
//                     ds_read_b128) and the weight planes packed [slice][plane][K/8][N][8], which IS the B-fragment layout: each
//                     wave loads its fragments straight from L2 into registers while the stage's split runs.
//   wgrad_x3_kernel   weight gradient of the 3x3 convolutions, all nine taps per block (the tiling of wgrad_halo_bf16.hip's
//                     first kernel: a 4 x 16-pixel activation tile against the 6 x 18 dY halo patch, nine accumulators per wave),
//                     both operands split at staging; one block per CU, 216 MFMAs per wave and tile.
// Everything else of this dtype (first convolution, ConvTranspose weight gradient, BatchNorm, head, losses) runs the f32 kernels.
#include "common.h"
#include "loader.h"
#include <type_traits>

namespace ustrun {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 b16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 b16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((address_space(3))) b16x4 lds_b16x4;
#define X3_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16

// three bf16 terms of four f32 values, each plane as two dwords (four bf16)
__device__ __forceinline__ void split4(const f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
    b16x4 h0, h1, h2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        h0[q] = (__bf16)v[q];
        const float r1 = v[q] - (float)h0[q];          // exact: the low 16 bits of v (+ the rounding carry)
        h1[q] = (__bf16)r1;
        const float r2 = r1 - (float)h1[q];            // exact
        h2[q] = (__bf16)r2;
    }
    p0 = __builtin_bit_cast(u32x2, h0); p1 = __builtin_bit_cast(u32x2, h1); p2 = __builtin_bit_cast(u32x2, h2);
}

// the six products of one fragment pair, small terms first
__device__ __forceinline__ f32x16 mfma6(const b16x8 (&a)[3], const b16x8 (&b)[3], f32x16 c) {
    c = X3_MFMA(a[0], b[2], c, 0, 0, 0);
    c = X3_MFMA(a[1], b[1], c, 0, 0, 0);
    c = X3_MFMA(a[2], b[0], c, 0, 0, 0);
    c = X3_MFMA(a[0], b[1], c, 0, 0, 0);
    c = X3_MFMA(a[1], b[0], c, 0, 0, 0);
    c = X3_MFMA(a[0], b[0], c, 0, 0, 0);
    return c;
}

// ---- weights: f32 [S][K][N] -> planes [S][3][K/8][N][8] bf16 ---------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_x3_kernel(const float* __restrict__ w, int S, int K, int N, __bf16* __restrict__ out) {
    const long total = (long)S * (K / 8) * N;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const int n = (int)(e % N);
        const long t = e / N;
        const int k8 = (int)(t % (K / 8)), s = (int)(t / (K / 8));
        const float* src = w + ((long)s * K + 8 * k8) * N + n;
        f32x4 lo, hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) { lo[q] = src[(long)q * N]; hi[q] = src[(long)(4 + q) * N]; }
        u32x2 a0, a1, a2, c0, c1, c2;
        split4(lo, a0, a1, a2);
        split4(hi, c0, c1, c2);
        const long plane = (long)(K / 8) * N * 8;
        __bf16* o = out + (long)s * 3 * plane + ((long)k8 * N + n) * 8;
        *(u32x4*)o = (u32x4){a0[0], a0[1], c0[0], c0[1]};
        *(u32x4*)(o + plane) = (u32x4){a1[0], a1[1], c1[0], c1[1]};
        *(u32x4*)(o + 2 * plane) = (u32x4){a2[0], a2[1], c2[0], c2[1]};
    }
}

// ---- implicit GEMM -----------------------------------------------------------------------------------------------------------
constexpr int XBK = 32;            // channels per stage
constexpr int XAP = 80;            // LDS row pitch of an activation plane: 32 bf16 + 16 bytes of padding
struct RowInfo { int n; int yx; };

template <int WM, int WN>
__global__ __launch_bounds__(256, 2) void igemm_x3_kernel(const IgemmArgs a, const int mt_total, const int nt_total) {
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int AR = BM / 32;         // activation rows per thread per stage
    constexpr int APLANE = BM * XAP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                                    // [3][BM][XAP]
    RowInfo* rowinfo = (RowInfo*)(smem + 3 * APLANE);

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
    const int l31 = lane & 31, lh = lane >> 5;

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

    const int a_c4 = tid & 7, a_r0 = tid >> 3;
    const int nchunk = a.Cin / XBK;
    const int nstage = a.nseg * nchunk;
    const int K8 = a.Cin / 8;
    // the three weight planes sit behind the f32 pack [nseg * nz][Cin][Cout]
    const __bf16* W3 = (const __bf16*)(a.W + (long)a.nseg * a.nz * a.Cin * a.Cout);
    const long plane = (long)K8 * a.Cout * 8;

    f32x4 av[AR];
    f32x4 asc, ash;
    unsigned aok = 0;
    int a_relu = 0;
    // Row geometry -- which input pixel a row reads for this tap, whether it lies inside the source, its element offset -- depends
    // on the tap and on the source only, not on the channel chunk: it is formed when one of the two changes (wave-uniform: the
    // sources' widths are multiples of the chunk) instead of for every stage, which had cost as many VALU instructions as the split.
    long goff[AR];
    int cur_seg = -1, cur_second = -1;
    const float* sptr = nullptr;
    const float* sscale = nullptr;
    const float* sshift = nullptr;
    int cbase = 0;
    auto load_stage = [&](int s) {
        const int seg = s / nchunk, c0 = (s - seg * nchunk) * XBK;
        const int second = (a.nsrc == 2 && c0 >= a.src[0].C) ? 1 : 0;
        if (seg != cur_seg || second != cur_second) {
            cur_seg = seg; cur_second = second;
            const SrcDev S = pick_src(a.src[0], a.src[1], second != 0);
            const int dy = a.d0 + (seg / a.segw) * a.dstep, dx = a.d0 + (seg % a.segw) * a.dstep;
            sptr = S.ptr; sscale = S.scale; sshift = S.shift; a_relu = S.relu;
            cbase = second ? a.src[0].C : 0;
            aok = 0;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const RowInfo ri = rowinfo[a_r0 + 32 * i];
                const int ly = (ri.yx >> 16) * a.s_in + dy - S.off_y;
                const int lx = (ri.yx & 0xffff) * a.s_in + dx - S.off_x;
                const bool ok = ri.n >= 0 && ly >= 0 && ly < S.LH && lx >= 0 && lx < S.LW;
                aok |= (ok ? 1u : 0u) << i;
                goff[i] = ok ? ri.n * S.sN + (long)ly * S.sH + (long)lx * S.sW : 0;      // (a clamped address: the value is masked at the split)
            }
        }
        const int cl = c0 + 4 * a_c4 - cbase;
        asc = (f32x4){1.f, 1.f, 1.f, 1.f}; ash = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (sscale) { asc = *(const f32x4*)(sscale + cl); ash = *(const f32x4*)(sshift + cl); }
#pragma unroll
        for (int i = 0; i < AR; ++i) av[i] = *(const f32x4*)(sptr + goff[i] + cl);
    };
    // this wave's weight fragments of a stage: [k step][column tile][plane], each lane 8 consecutive k of its column
    u32x4 bfr[2][2][3];
    auto load_b = [&](int s) {
        const int seg = s / nchunk, c0 = (s - seg * nchunk) * XBK;
        const __bf16* ws = W3 + (long)(seg + z) * 3 * plane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 64 + j * 32 + l31;
                const long o = ((long)(c0 / 8 + 2 * ks + lh) * a.Cout + (n < a.Cout ? n : 0)) * 8;
#pragma unroll
                for (int p = 0; p < 3; ++p) bfr[ks][j][p] = *(const u32x4*)(ws + p * plane + o);
            }
    };
    auto write_stage = [&]() {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            f32x4 v = av[i] * asc + ash;
            if (a_relu) v = relu4(v);
            if (!((aok >> i) & 1u)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            u32x2 p0, p1, p2;
            split4(v, p0, p1, p2);
            char* dst = As + (a_r0 + 32 * i) * XAP + a_c4 * 8;
            *(u32x2*)dst = p0; *(u32x2*)(dst + APLANE) = p1; *(u32x2*)(dst + 2 * APLANE) = p2;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const char* Ap = As + (wm * 64 + l31) * XAP + lh * 16;
    load_stage(0);
    for (int s = 0; s < nstage; ++s) {
        load_b(s);                                   // in flight under the split below
        __builtin_amdgcn_sched_barrier(0);           // (hipcc otherwise sinks these loads to just in front of their MFMAs: conv3x3_x3_kernel)
        write_stage();
        __syncthreads();
        if (s + 1 < nstage) load_stage(s + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            b16x8 af[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) af[i][p] = __builtin_bit_cast(b16x8, *(const u32x4*)(Ap + p * APLANE + i * 32 * XAP + ks * 32));
            constexpr int TA[6] = {0, 1, 2, 0, 1, 0}, TB[6] = {2, 1, 0, 1, 0, 0};       // mfma6's order, term by term over the four accumulators
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][j] = X3_MFMA(af[i][TA[q]], __builtin_bit_cast(b16x8, bfr[ks][j][TB[q]]), acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue (igemm.hip's): D[row = pixel][col = channel]; col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
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
                    const float v = acc[i][j][r] + bias;
                    const int oy = (ri.yx >> 16) * a.s_out + oyz, ox = (ri.yx & 0xffff) * a.s_out + oxz;
                    if (col < a.C0) {
                        a.out0[(((long)ri.n * a.Ho + oy) * a.Wo + ox) * a.C0 + col] = v;
                    } else {
                        const int y1 = oy - a.o1y, x1 = ox - a.o1x;
                        if (y1 >= 0 && y1 < a.H1 && x1 >= 0 && x1 < a.W1)
                            a.out1[(((long)ri.n * a.H1 + y1) * a.W1 + x1) * C1 + (col - a.C0)] = v;
                    }
                    s1[j] += v; s2[j] += v * v;
                }
            }
        }
    }
    if (a.stat) {
        float* red = (float*)As;  // [WM][2][BN], free after the final barrier of the main loop
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
        constexpr int HALVES = WM / 2;     // one statistics row per 128 pixels (ustrun_conv_mtiles)
        const int stat_rows = (int)((a.M + 127) / 128);
        for (int t = tid; t < HALVES * 2 * BN; t += 256) {
            const int h = t / (2 * BN), q = (t / BN) % 2, c = t % BN;
            const float v = red[((2 * h) * 2 + q) * BN + c] + red[((2 * h + 1) * 2 + q) * BN + c];
            const int srow = mtile * HALVES + h;
            if (srow < stat_rows && n0 + c < a.Cout) a.stat[((long)srow * 2 + q) * a.Cout + n0 + c] = v;
        }
    }
}

// ---- 3x3 convolution, halo-tiled -------------------------------------------------------------------------------------------
// The generic kernel above stages (and splits) an activation once per TAP and per column tile: at 40 VALU instructions per four
// values the split, not the matrix pipe, paced it (4.4 VALU per MFMA; 112 TF/s-equivalent).  Here a block owns an 8 x 16-pixel tile
// and stages the tile's 10 x 18 halo patch of a 32-channel chunk ONCE -- three planes of 80-byte rows -- for all nine taps: a tap is
// a constant byte offset into the patch (forward: (kh - 1, kw - 1); input gradient: the mirrored tap, FLIP), 432 MFMAs per wave and
// chunk against one split of 180 x 32 values.  Weight fragments come straight from L2, one tap ahead.  WM = 2: 128 output columns
// (waves 2 x 2, wave tile 64 px x 64 columns); WM = 4: 64 output columns (waves 4 x 1, wave tile 32 px x 64 columns).
constexpr int CTH = 8, CTW = 16, CPW = CTW + 2, CPP = (CTH + 2) * CPW;     // 180 patch pixels
constexpr int CPLANE = CPP * XAP;                                          // one plane of the patch: 14400 B
constexpr int CIT = (CPP * 8 + 255) / 256;                                 // float4 items per thread and chunk: 6

template <int WM, bool FLIP>
__global__ __launch_bounds__(256, 2) void conv3x3_x3_kernel(const IgemmArgs a, const int tiles_x, const int tiles_y, const int nt_total) {
    constexpr int WN = 4 / WM, RT = 4 / WM;          // column waves; 32-row tiles per wave (2 or 1)
    constexpr int BN = 64 * WN;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [3][180 px][XAP]
    const int mt_total = a.N * tiles_y * tiles_x;
    const int ntiles = mt_total * nt_total;
    int bid = blockIdx.x;
    {
        const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int mtile = bid / nt_total, ntile = bid % nt_total;
    const int img = mtile / (tiles_y * tiles_x), trem = mtile - img * tiles_y * tiles_x;
    const int y0 = (trem / tiles_x) * CTH, x0 = (trem % tiles_x) * CTW;
    const int n0 = ntile * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int nchunk = a.Cin / XBK, K8 = a.Cin / 8;
    const __bf16* W3 = (const __bf16*)(a.W + 9L * a.Cin * a.Cout);
    const long plane = (long)K8 * a.Cout * 8;

    // ---- staging: item i = patch pixel (tid + 256 i) >> 3, 4-channel group tid & 7; geometry per source, not per chunk ----
    const int c4 = tid & 7;
    int goff[CIT];                                  // element offsets inside the source (< 2^31: conv3x3_x3_supported)
    unsigned gok = 0;
    int cur_second = -1, cbase = 0, s_relu = 0;
    const float* sptr = nullptr;
    const float* sscale = nullptr;
    const float* sshift = nullptr;
    auto geometry = [&](int second) {
        cur_second = second;
        const SrcDev S = pick_src(a.src[0], a.src[1], second != 0);
        // batched passes: a tile lies in one image, its pass picks the BatchNorm constants (gN images per pass, gstride floats apart)
        const long gofs = (S.scale && S.gN > 0) ? (long)(img / S.gN) * S.gstride : 0;
        sptr = S.ptr; sscale = S.scale ? S.scale + gofs : nullptr; sshift = S.shift ? S.shift + gofs : nullptr; s_relu = S.relu;
        cbase = second ? a.src[0].C : 0;
        gok = 0;
#pragma unroll
        for (int i = 0; i < CIT; ++i) {
            const int px = (tid + 256 * i) >> 3;
            const int py = px / CPW, pxx = px - py * CPW;
            const int ly = y0 - 1 + py - S.off_y, lx = x0 - 1 + pxx - S.off_x;
            const bool ok = px < CPP && ly >= 0 && ly < S.LH && lx >= 0 && lx < S.LW;
            gok |= (ok ? 1u : 0u) << i;
            goff[i] = ok ? (int)(img * S.sN + (long)ly * S.sH + (long)lx * S.sW) : 0;
        }
    };
    f32x4 av[CIT];
    f32x4 asc, ash;
    auto fetch = [&](int c) {
        const int c0 = c * XBK;
        const int second = (a.nsrc == 2 && c0 >= a.src[0].C) ? 1 : 0;
        if (second != cur_second) geometry(second);
        const int cl = c0 + 4 * c4 - cbase;
        asc = (f32x4){1.f, 1.f, 1.f, 1.f}; ash = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (sscale) { asc = *(const f32x4*)(sscale + cl); ash = *(const f32x4*)(sshift + cl); }
#pragma unroll
        for (int i = 0; i < CIT; ++i) av[i] = *(const f32x4*)(sptr + goff[i] + cl);
    };
    auto write_patch = [&]() {
#pragma unroll
        for (int i = 0; i < CIT; ++i) {
            const int px = (tid + 256 * i) >> 3;
            if (px < CPP) {
                f32x4 v = av[i] * asc + ash;
                if (s_relu) v = relu4(v);
                if (!((gok >> i) & 1u)) v = (f32x4){0.f, 0.f, 0.f, 0.f};      // zero padding is applied after the activation
                u32x2 p0, p1, p2;
                split4(v, p0, p1, p2);
                char* dst = smem + px * XAP + c4 * 8;
                *(u32x2*)dst = p0; *(u32x2*)(dst + CPLANE) = p1; *(u32x2*)(dst + 2 * CPLANE) = p2;
            }
        }
    };
    // weight fragments of step u = 2 tap + (16-channel half) of a chunk: [column tile][plane]
    auto load_b = [&](u32x4 (&b)[2][3], int c, int u) {
        const __bf16* ws = W3 + (long)(u >> 1) * 3 * plane;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + l31;
            const long o = ((long)(c * (XBK / 8) + 2 * (u & 1) + lh) * a.Cout + (n < a.Cout ? n : 0)) * 8;
#pragma unroll
            for (int p = 0; p < 3; ++p) b[j][p] = *(const u32x4*)(ws + p * plane + o);
        }
    };

    f32x16 acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // this lane's A-fragment base: tile row 2 RT wm + 2 i + (l31 >> 4), column l31 & 15, at patch offset (+1, +1); k half lh
    const char* Ap = smem + ((2 * RT * wm + (l31 >> 4) + 1) * CPW + (l31 & 15) + 1) * XAP + lh * 16;
    // A chunk is 18 steps (tap, 16-channel half): 24 MFMAs on the 64-row wave tile, 12 on the 32-row one.  The fragments of a step
    // (24 registers) load NSET - 1 steps ahead into a ring of NSET sets -- one step of lead for the former, two for the latter: the
    // same 24 MFMAs of flight time.  (Two whole taps' worth of registers, 96, put the 64-row build 60 registers past the file.)  18 is
    // a multiple of 2 and of 3: a step's set is a compile-time constant whatever the chunk.
    constexpr int NSET = RT == 2 ? 2 : 3;
    u32x4 bfr[NSET][2][3];
    fetch(0);
#pragma unroll
    for (int u = 0; u < NSET - 1; ++u) load_b(bfr[u], 0, u);
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        if (c) __syncthreads();                       // every wave is done reading the previous chunk's patch
        write_patch();
        __syncthreads();
        if (c + 1 < nchunk) fetch(c + 1);             // registers, in flight under the nine taps below
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            constexpr int dummy = 0; (void)dummy;
            const int un = u + NSET - 1;              // the step whose fragments load under this one's MFMAs
            if (un < 18) load_b(bfr[un % NSET], c, un);
            else if (c + 1 < nchunk) load_b(bfr[un % NSET], c + 1, un - 18);
            // (left alone, hipcc sinks each of these loads to a few MFMAs in front of its use -- shorter live ranges -- and the step waits
            // out an L2 round trip per fragment: the matrix pipe was 51 % busy.  Nothing crosses this line.)
            __builtin_amdgcn_sched_barrier(0);
            const int t = u >> 1, ks = u & 1;
            const int kh = t / 3, kw = t % 3;
            const int dy = FLIP ? 1 - kh : kh - 1, dx = FLIP ? 1 - kw : kw - 1;
            const char* At = Ap + (dy * CPW + dx) * XAP;
            b16x8 af[RT][3];
#pragma unroll
            for (int i = 0; i < RT; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) af[i][p] = __builtin_bit_cast(b16x8, *(const u32x4*)(At + p * CPLANE + 2 * i * CPW * XAP + ks * 32));
            // term by term over all of the wave's accumulators: the six products of one accumulator (small terms first, as in mfma6)
            // stand 2 RT instructions apart instead of back to back
            constexpr int TA[6] = {0, 1, 2, 0, 1, 0}, TB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < RT; ++i)
                        acc[i][j] = X3_MFMA(af[i][TA[q]], __builtin_bit_cast(b16x8, bfr[u % NSET][j][TB[q]]), acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: D[row = pixel][col = channel]; col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the 32-row tile
    const int C1 = a.Cout - a.C0;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const bool cok = col < a.Cout;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int oy = y0 + 2 * RT * wm + 2 * i + (row >> 4), ox = x0 + (row & 15);
                if (oy < a.Hb && ox < a.Wb && cok) {
                    const float v = acc[i][j][r];
                    if (col < a.C0) {
                        a.out0[(((long)img * a.Ho + oy) * a.Wo + ox) * a.C0 + col] = v;
                    } else {
                        const int y1 = oy - a.o1y, x1 = ox - a.o1x;
                        if (y1 >= 0 && y1 < a.H1 && x1 >= 0 && x1 < a.W1)
                            a.out1[(((long)img * a.H1 + y1) * a.W1 + x1) * C1 + (col - a.C0)] = v;
                    }
                    s1[j] += v; s2[j] += v * v;
                }
            }
        }
    }
    if (a.stat) {                                     // one statistics row per tile (fixed order: lane halves, then the row waves)
        __syncthreads();
        float* red = (float*)smem;                    // [WM][2][BN]
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
        for (int t = tid; t < 2 * BN; t += 256) {
            const int q = t / BN, c = t % BN;
            float v = red[q * BN + c];
#pragma unroll
            for (int w = 1; w < WM; ++w) v += red[(w * 2 + q) * BN + c];
            if (n0 + c < a.Cout) a.stat[((long)mtile * 2 + q) * a.Cout + n0 + c] = v;
        }
    }
}

// ---- weight gradient, all nine taps per block ----------------------------------------------------------------------------------
constexpr int TH = 4, TW = 16, HW2 = TW + 2, HP = (TH + 2) * HW2;   // 4 x 16 tile, 6 x 18 = 108 halo pixels
constexpr int RB = 192;                                              // LDS row pitch: 64 bf16 + 64 bytes (conflict-free transposing reads)
constexpr int WAIT = TH * TW * 16 / 256;                             // float4 items per thread: activation 4,
constexpr int WDIT = (HP * 16 + 255) / 256;                          //                          dY 7

__device__ __forceinline__ b16x8 tr_frag(const char* lane_base, int k0) {
    const b16x4 lo = __builtin_bit_cast(b16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ustrun_lds_s16x4*)(lane_base + k0 * RB)));
    const b16x4 hi = __builtin_bit_cast(b16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ustrun_lds_s16x4*)(lane_base + (k0 + 4) * RB)));
    b16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// Consumer / producer waves (as the 64 -> 64 streaming kernel of round 3): waves 0-3 read fragments and multiply, waves 4-7 fetch,
// activate, split and write the NEXT tile into the other of two stage buffers -- one of each per SIMD, so the ~430 VALU instructions of
// splitting a tile run under the other wave's 216 MFMAs instead of in front of them (one block of four waves per CU did both in turn:
// the matrix pipe idled through every split and its two barriers).  Rows are 128 bytes with the 64-byte halves swapped where bit 1 of
// the row is set (the four rows of a transposing read stay on disjoint bank quarters without the 64 bytes of padding per row that
// left LDS for ONE stage only).
constexpr int RBW = 128;
constexpr int WDROWS = (HP * 16 + 255) / 256 * 16;            // 112: every staging item of the dY patch has a row (108 .. 111: slack, never read)
constexpr int WATILE2 = TH * TW * RBW, WDTILE2 = WDROWS * RBW, WSTAGE2 = 3 * (WATILE2 + WDTILE2);

__device__ __forceinline__ b16x8 tr_fragw(const char* lane_base, int k0) {
    const b16x4 lo = __builtin_bit_cast(b16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ustrun_lds_s16x4*)(lane_base + k0 * RBW)));
    const b16x4 hi = __builtin_bit_cast(b16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ustrun_lds_s16x4*)(lane_base + (k0 + 4) * RBW)));
    b16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// grid = (ci tiles * co tiles, ksplit); tiles_per = spatial tiles per split
__global__ __launch_bounds__(512, 2) void wgrad_x3_kernel(const WgradArgs a, const int ntn, const int tiles_x, const int tiles_y,
                                                          const int tiles_per) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 x { [3][64 px][RBW], [3][112 px][RBW] }
    const int lane = threadIdx.x & 63, ptid = threadIdx.x & 255;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool producer = wave8 >= 4;                // wave-uniform
    const int wave = wave8 & 3, wi = wave >> 1, wj = wave & 1;
    const int mtile = blockIdx.x / ntn, ntile = blockIdx.x % ntn;
    const int ci0 = mtile * 64, co0 = ntile * 64;
    const int ttotal = a.N * tiles_y * tiles_x;
    const int tbeg = blockIdx.y * tiles_per;
    const int tend = min(ttotal, tbeg + tiles_per);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (producer) {
        // activation item i = pixel (ptid + 256 i) >> 4 of the tile, 4-channel group ptid & 15 (one source, one set of constants per thread)
        const int c4 = ptid & 15;
        const int cg = ci0 + 4 * c4;
        const bool second = (a.nsrc == 2 && cg >= a.src[0].C);
        const SrcDev S = pick_src(a.src[0], a.src[1], second);
        const int cl = cg - (second ? a.src[0].C : 0);
        f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
        int cur_grp = -1;                              // batched passes: the constants follow the image of the tile being fetched
        auto load_consts = [&](int img) {
            const int grp = S.gN > 0 ? img / S.gN : 0;
            if (S.scale && grp != cur_grp) {
                const long o = (long)grp * (S.gN > 0 ? S.gstride : 0) + cl;
                asc = *(const f32x4*)(S.scale + o); ash = *(const f32x4*)(S.shift + o);
                cur_grp = grp;
            }
        };
        const float a_floor = S.relu ? 0.f : -__builtin_inff();
        const float* sp = S.ptr + cl;
        const float* dyp = a.dy + co0 + 4 * c4;
        f32x4 av[WAIT], dv[WDIT];
        unsigned aok = 0;
        // tile-invariant item geometry: (row, column) inside the tile / the patch and the element offset from the tile's origin --
        // per tile only two compares and a select per item remain (the 64-bit index arithmetic per item and tile was half of the
        // producers' ~840 VALU instructions per tile, which compete with the consumers' MFMAs for the SIMD's issue slots)
        int apy[WAIT], apx[WAIT], aoff[WAIT], dhy[WDIT], dhx[WDIT], doff[WDIT];
#pragma unroll
        for (int i = 0; i < WAIT; ++i) {
            const int px = (ptid + 256 * i) >> 4;
            apy[i] = px >> 4; apx[i] = px & 15;
            aoff[i] = (int)(apy[i] * S.sH + apx[i] * S.sW);
        }
#pragma unroll
        for (int i = 0; i < WDIT; ++i) {
            const int hp = (ptid + 256 * i) >> 4;
            dhy[i] = hp < HP ? hp / HW2 : -(1 << 20);             // (items past the patch: never inside the image)
            dhx[i] = hp - (hp / HW2) * HW2;
            doff[i] = ((hp / HW2) * a.dyW + dhx[i]) * a.Cout;
        }
        auto fetch_tile = [&](int t) {
            const int img = __builtin_amdgcn_readfirstlane(t / (tiles_y * tiles_x));
            const int rem = t - img * tiles_y * tiles_x;
            const int y0 = __builtin_amdgcn_readfirstlane((rem / tiles_x) * TH), x0 = __builtin_amdgcn_readfirstlane((rem % tiles_x) * TW);
            const int ty = y0 - S.off_y, tx = x0 - S.off_x;
            const int arel = (int)(ty * S.sH + tx * S.sW);              // tile origin inside the image (elements; < 2^31: the host checks)
            const float* ab = sp + (long)img * S.sN + arel;            // wave-uniform
            aok = 0;
#pragma unroll
            for (int i = 0; i < WAIT; ++i) {
                const bool ok = (unsigned)(ty + apy[i]) < (unsigned)S.LH && (unsigned)(tx + apx[i]) < (unsigned)S.LW;
                av[i] = *(const f32x4*)(ab + (ok ? aoff[i] : -arel));      // (outside the image: its first pixel, masked at the split)
                aok |= (ok ? 1u : 0u) << i;
            }
            const int drel = ((y0 - 1) * a.dyW + (x0 - 1)) * a.Cout;
            const float* db = dyp + (long)img * a.dyH * a.dyW * a.Cout + drel;
#pragma unroll
            for (int i = 0; i < WDIT; ++i) {
                const bool ok = (unsigned)(y0 - 1 + dhy[i]) < (unsigned)a.dyH && (unsigned)(x0 - 1 + dhx[i]) < (unsigned)a.dyW;
                dv[i] = *(const f32x4*)(db + (ok ? doff[i] : -drel));
                aok |= (ok ? 1u : 0u) << (8 + i);
            }
        };
        // (the constants a tile is split with are those of ITS image: loaded here, at its split, not at its fetch one stage earlier)
        auto write_tile = [&](char* stage, int t) {
            load_consts(t / (tiles_y * tiles_x));
#pragma unroll
            for (int i = 0; i < WAIT; ++i) {
                const int px = (ptid + 256 * i) >> 4;
                f32x4 v = av[i] * asc + ash;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = __builtin_fmaxf(v[q], a_floor);
                if (!((aok >> i) & 1u)) v = (f32x4){0.f, 0.f, 0.f, 0.f};       // padding is applied after the activation
                u32x2 p0, p1, p2;
                split4(v, p0, p1, p2);
                char* dst = stage + px * RBW + ((c4 * 8) ^ (((px >> 1) & 1) << 6));
                *(u32x2*)dst = p0; *(u32x2*)(dst + WATILE2) = p1; *(u32x2*)(dst + 2 * WATILE2) = p2;
            }
#pragma unroll
            for (int i = 0; i < WDIT; ++i) {
                const int hp = (ptid + 256 * i) >> 4;
                const f32x4 v = ((aok >> (8 + i)) & 1u) ? dv[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
                u32x2 p0, p1, p2;
                split4(v, p0, p1, p2);
                char* dst = stage + 3 * WATILE2 + hp * RBW + ((c4 * 8) ^ (((hp >> 1) & 1) << 6));      // (hp >= HP: the slack rows)
                *(u32x2*)dst = p0; *(u32x2*)(dst + WDTILE2) = p1; *(u32x2*)(dst + 2 * WDTILE2) = p2;
            }
        };
        if (tbeg < tend) {
            fetch_tile(tbeg);
            write_tile(smem, tbeg);
            if (tbeg + 1 < tend) fetch_tile(tbeg + 1);
        }
        __syncthreads();
        char* nxt = smem + WSTAGE2;
        for (int t = tbeg; t < tend; ++t) {
            if (t + 1 < tend) {
                write_tile(nxt, t + 1);                   // under the consumers' MFMAs of tile t
                if (t + 2 < tend) fetch_tile(t + 2);      // in flight until the split one stage on
            }
            __syncthreads();
            nxt = smem + (nxt == smem ? WSTAGE2 : 0);
        }
        return;                                           // (the slab is the consumers')
    }

    // ---- consumers.  Fragment bases: rows 8 (lane >> 5) + q (+ 4), columns 32 w + 16 ((lane >> 4) & 1) + 4 (lane & 3); the half-swap of a
    // row depends on bit 1 of (first row of the fragment + the lane's row): activation fragments start on multiples of 16, dY
    // fragments anywhere -- four bases by the start's low two bits
    const int lrow = 8 * (lane >> 5) + ((lane & 15) >> 2), lcolb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    const int abase = lrow * RBW + ((wi * 64 + lcolb) ^ (((lrow >> 1) & 1) << 6));
    int dbase4[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) dbase4[c] = 3 * WATILE2 + lrow * RBW + ((wj * 64 + lcolb) ^ ((((c + lrow) >> 1) & 1) << 6));
    __syncthreads();                                      // the first tile is in place
    const char* cur = smem;
    for (int t = tbeg; t < tend; ++t) {
        b16x8 af[TH][3];
#pragma unroll
        for (int r = 0; r < TH; ++r)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[r][p] = tr_fragw(cur + abase + p * WATILE2, r * TW);
#pragma unroll
        for (int pr = 0; pr < TH + 2; ++pr) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                constexpr int dummy = 0; (void)dummy;
                const int k0 = pr * HW2 + 2 - kw;
                b16x8 b[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) b[p] = tr_fragw(cur + dbase4[k0 & 3] + p * WDTILE2, k0);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {      // tap (kh, kw) pairs pixel row r with dY row r + 2 - kh of the patch
                    const int r = pr + kh - 2;
                    if (r >= 0 && r < TH) acc[kh * 3 + kw] = mfma6(b, af[r], acc[kh * 3 + kw]);      // D[co][ci]
                }
            }
        }
        __syncthreads();                                  // tile t is read, tile t + 1 is written
        cur = smem + (cur == smem ? WSTAGE2 : 0);
    }

    // slab in the torch weight layout [Cout][Cin][3][3]
    float* slab = a.partials + (long)blockIdx.y * 9 * a.Cin * a.Cout;
    const int l31 = lane & 31, lh = lane >> 5;
    const int ci = ci0 + wi * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wj * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float* o = slab + ((long)co * a.Cin + ci) * 9;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) o[tap] = acc[tap][r];
    }
}


// ---- ConvTranspose2d(k = 2, s = 2) weight gradient: dW[tap][ci][co] = sum_p A[p][ci] du[2 p + tap][co] -------------------------------
// The four taps are the four parity classes of the hi-resolution gradient: a block owns a 2 x 16 low-resolution tile (32 pixels), its
// activation tile and the 4 x 32 hi-resolution patch of du de-interleaved into four 32-pixel tap tiles, all split into three planes
// at staging; four accumulators per wave (32 ci x 32 co x 4 taps), 48 MFMAs per wave and tile.
constexpr int UTH = 2, UPX = UTH * TW;                       // 32 low-resolution pixels per tile
constexpr int UATILE = UPX * RB, UDTILE = 4 * UPX * RB;      // one plane: activation tile / the four tap tiles
constexpr int UAIT = UPX * 16 / 256, UDIT = 4 * UPX * 16 / 256;      // float4 items per thread: 2 / 8

__global__ __launch_bounds__(256, 1) void wgradT_x3_kernel(const WgradArgs a, const int ntn, const int tiles_x, const int tiles_y,
                                                           const int tiles_per) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                               // [3][32 px][RB]
    char* Ds = smem + 3 * UATILE;                  // [3][4 taps][32 px][RB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int mtile = blockIdx.x / ntn, ntile = blockIdx.x % ntn;
    const int ci0 = mtile * 64, co0 = ntile * 64;
    const int ttotal = a.N * tiles_y * tiles_x;
    const int tbeg = blockIdx.y * tiles_per;
    const int tend = min(ttotal, tbeg + tiles_per);
    const int c4 = tid & 15;
    const SrcDev& S = a.src[0];
    f32x4 asc = {1.f, 1.f, 1.f, 1.f}, ash = {0.f, 0.f, 0.f, 0.f};
    if (S.scale) { asc = *(const f32x4*)(S.scale + ci0 + 4 * c4); ash = *(const f32x4*)(S.shift + ci0 + 4 * c4); }
    const float* sp = S.ptr + ci0 + 4 * c4;
    const float* dup = a.dy + co0 + 4 * c4;

    f32x4 av[UAIT], dv[UDIT];
    unsigned aok = 0;
    auto fetch_tile = [&](int t) {
        const int img = t / (tiles_y * tiles_x);
        const int rem = t - img * tiles_y * tiles_x;
        const int y0 = (rem / tiles_x) * UTH, x0 = (rem % tiles_x) * TW;
        aok = 0;
#pragma unroll
        for (int i = 0; i < UAIT; ++i) {
            const int px = (tid + 256 * i) >> 4;
            const int ly = y0 + (px >> 4), lx = x0 + (px & 15);
            const bool ok = ly < a.Hb && lx < a.Wb;
            av[i] = *(const f32x4*)(sp + img * S.sN + (long)(ok ? ly : 0) * S.sH + (long)(ok ? lx : 0) * S.sW);
            aok |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < UDIT; ++i) {
            const int hr = (tid + 256 * i) >> 4;                      // hi-resolution pixel of the 4 x 32 patch
            const int hy = hr >> 5, hx = hr & 31;
            const bool ok = y0 + (hy >> 1) < a.Hb && x0 + (hx >> 1) < a.Wb;
            const f32x4 v = *(const f32x4*)(dup + (((long)img * a.dyH + (ok ? 2 * y0 + hy : 0)) * a.dyW + (ok ? 2 * x0 + hx : 0)) * a.Cout);
            dv[i] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < UAIT; ++i) {
            const int px = (tid + 256 * i) >> 4;
            f32x4 v = av[i] * asc + ash;
            if (S.relu) v = relu4(v);
            if (!((aok >> i) & 1u)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
            u32x2 p0, p1, p2;
            split4(v, p0, p1, p2);
            char* dst = As + px * RB + c4 * 8;
            *(u32x2*)dst = p0; *(u32x2*)(dst + UATILE) = p1; *(u32x2*)(dst + 2 * UATILE) = p2;
        }
#pragma unroll
        for (int i = 0; i < UDIT; ++i) {
            const int hr = (tid + 256 * i) >> 4;
            const int hy = hr >> 5, hx = hr & 31;
            const int tap = (hy & 1) * 2 + (hx & 1), px = (hy >> 1) * TW + (hx >> 1);
            u32x2 p0, p1, p2;
            split4(dv[i], p0, p1, p2);
            char* dst = Ds + (tap * UPX + px) * RB + c4 * 8;
            *(u32x2*)dst = p0; *(u32x2*)(dst + UDTILE) = p1; *(u32x2*)(dst + 2 * UDTILE) = p2;
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int lrow = 8 * (lane >> 5) + ((lane & 15) >> 2), lcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const char* Ab = As + lrow * RB + (wi * 32 + lcol) * 2;
    const char* Db = Ds + lrow * RB + (wj * 32 + lcol) * 2;

    if (tbeg < tend) { fetch_tile(tbeg); write_tile(); }
    __syncthreads();
#pragma unroll 1
    for (int t = tbeg; t < tend; ++t) {
        const bool more = t + 1 < tend;
        if (more) fetch_tile(t + 1);
#pragma unroll
        for (int r = 0; r < UTH; ++r) {
            b16x8 af[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) af[p] = tr_frag(Ab + p * UATILE, r * TW);
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                b16x8 b[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) b[p] = tr_frag(Db + p * UDTILE + tap * UPX * RB, r * TW);
                acc[tap] = mfma6(b, af, acc[tap]);      // D[co][ci]
            }
        }
        __syncthreads();
        if (more) write_tile();
        __syncthreads();
    }

    // slab in the torch layout [Cin][Cout][2][2]
    float* slab = a.partials + (long)blockIdx.y * 4 * a.Cin * a.Cout;
    const int l31 = lane & 31, lh = lane >> 5;
    const int ci = ci0 + wi * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + wj * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float* o = slab + ((long)ci * a.Cout + co) * 4;
        *(f32x4*)o = (f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------------
// planes behind an f32 pack [S][K][N] (K a multiple of 8; otherwise no planes: the kernels above do not take such layers)
int pack_x3(const float* w_packed, int S, int K, int N, hipStream_t st) {
    if (K % 8) return 0;
    __bf16* out = (__bf16*)(w_packed + (long)S * K * N);
    long blocks = ((long)S * (K / 8) * N + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_x3_kernel, dim3((int)blocks), dim3(256), 0, st, w_packed, S, K, N, out);
    USTRUN_LAUNCH_CHECK("pack_x3");
    return 0;
}

bool igemm_x3_supported(const IgemmArgs& a) {
    if (g_debug_flags & (1 << 29)) return false;                   // (A/B runs against the f32 matrix-core kernel)
    if (a.Cin % XBK || a.out_esz != 4 || a.M <= 0) return false;
    if (a.nsrc == 2 && a.src[0].C % XBK) return false;             // a chunk never straddles the two sources
    for (int i = 0; i < a.nsrc; ++i) {
        const SrcDev& s = a.src[i];
        if (s.esz != 4 || s.sC != 1 || (s.C & 3) || s.pool) return false;     // (batched passes, gN > 0: the halo-tiled kernel only -- igemm.hip)
        if ((s.sN | s.sH | s.sW) & 3) return false;                // 16-byte loads
    }
    return true;
}

// the halo-tiled kernel: a plain 3x3 convolution (forward, or the input gradient's mirrored taps) without output stride / parity
bool conv3x3_x3_supported(const IgemmArgs& a) {
    if (!igemm_x3_supported(a) || (g_debug_flags & (1 << 30))) return false;      // (bit 30: the generic x3 kernel, for A/B runs)
    if (a.nseg != 9 || a.segw != 3 || a.nz != 1 || a.s_in != 1 || a.s_out != 1 || a.bias) return false;
    if (!((a.d0 == -1 && a.dstep == 1) || (a.d0 == 1 && a.dstep == -1))) return false;
    if (a.Ho != a.Hb || a.Wo != a.Wb) return false;
    for (int i = 0; i < a.nsrc; ++i)
        if ((long)a.N * a.src[i].sN >= (1L << 31) - 64) return false;              // 32-bit element offsets of the staging items
    return true;
}
int conv3x3_x3_stat_rows(const IgemmArgs& a) { return a.N * cdiv(a.Hb, CTH) * cdiv(a.Wb, CTW); }

int igemm_x3_launch(const IgemmArgs& a, hipStream_t st) {
    if (conv3x3_x3_supported(a)) {
        const int tx = cdiv(a.Wb, CTW), ty = cdiv(a.Hb, CTH);
        const bool wide = a.Cout % 128 == 0 || a.Cout > 128;
        const int nt = cdiv(a.Cout, wide ? 128 : 64);
        dim3 grid(a.N * ty * tx * nt), block(256);
        const int lds = 3 * CPLANE;
        const bool flip = a.d0 == 1;
        if (wide) {
            if (flip) hipLaunchKernelGGL((conv3x3_x3_kernel<2, true>), grid, block, lds, st, a, tx, ty, nt);
            else hipLaunchKernelGGL((conv3x3_x3_kernel<2, false>), grid, block, lds, st, a, tx, ty, nt);
        } else {
            if (flip) hipLaunchKernelGGL((conv3x3_x3_kernel<4, true>), grid, block, lds, st, a, tx, ty, nt);
            else hipLaunchKernelGGL((conv3x3_x3_kernel<4, false>), grid, block, lds, st, a, tx, ty, nt);
        }
        USTRUN_LAUNCH_CHECK("conv3x3_x3");
        return 0;
    }
    constexpr int BM = 128, BN = 128;
    const int mt = cdiv(a.M, BM), nt = cdiv(a.Cout, BN);
    const int lds = 3 * BM * XAP + BM * (int)sizeof(RowInfo);
    dim3 grid(mt * nt, a.nz), block(256);
    hipLaunchKernelGGL((igemm_x3_kernel<2, 2>), grid, block, lds, st, a, mt, nt);
    USTRUN_LAUNCH_CHECK("igemm_x3");
    return 0;
}

bool wgrad_x3_supported(const WgradArgs& a) {
    if (g_debug_flags & (1 << 29)) return false;
    if (a.nseg != 9 || a.segw != 3 || a.d0 != -1 || a.astep != 1 || a.dy_s != 1 || a.ashift != 0 || a.dy_esz != 4) return false;
    if (a.Cin % 64 || a.Cout % 64 || a.dyH != a.Hb || a.dyW != a.Wb) return false;
    for (int i = 0; i < a.nsrc; ++i) {
        const SrcDev& s = a.src[i];
        if (s.esz != 4 || s.sC != 1 || (s.C & 3) || s.pool) return false;
        if ((s.sN | s.sH | s.sW) & 3) return false;
        if (s.sN >= (1L << 31) - 64 || (s.H + 8L) * s.sH >= (1L << 31) - 64) return false;      // 32-bit element offsets inside an image
    }
    if ((long)(a.dyH + 8) * a.dyW * a.Cout >= (1L << 31) - 64) return false;
    return true;
}

// split-K over space: one block per CU (98 KB of LDS each), slabs == ksplit
int wgrad_x3_plan(const WgradArgs& a, int* ksplit, int* tiles_per) {
    const long pairs = (long)(a.Cin / 64) * (a.Cout / 64);
    const int ttotal = a.N * cdiv(a.Hb, TH) * cdiv(a.Wb, TW);
    long ks = (256 + pairs - 1) / pairs;
    if (ks > ttotal / 4) ks = ttotal / 4;
    if (ks < 1) ks = 1;
    const int per = cdiv(ttotal, ks);
    *tiles_per = per; *ksplit = cdiv(ttotal, per);
    return 0;
}

bool wgradT_x3_supported(const WgradArgs& a) {
    if (g_debug_flags & (1 << 29)) return false;
    if (a.nseg != 4 || a.segw != 2 || a.dy_s != 2 || a.astep != 0 || a.d0 != 0 || a.ashift != 0 || a.dy_esz != 4 || a.nsrc != 1) return false;
    const SrcDev& s = a.src[0];
    if (s.esz != 4 || s.sC != 1 || s.pool || s.off_y || s.off_x || s.LH != a.Hb || s.LW != a.Wb || s.gN > 0) return false;
    if ((s.sN | s.sH | s.sW) & 3) return false;
    return a.Cin % 64 == 0 && a.Cout % 64 == 0 && a.dyH == 2 * a.Hb && a.dyW == 2 * a.Wb;
}
int wgradT_x3_plan(const WgradArgs& a, int* ksplit, int* tiles_per) {
    const long pairs = (long)(a.Cin / 64) * (a.Cout / 64);
    const int ttotal = a.N * cdiv(a.Hb, UTH) * cdiv(a.Wb, TW);
    long ks = (256 + pairs - 1) / pairs;
    if (ks > ttotal / 8) ks = ttotal / 8;
    if (ks < 1) ks = 1;
    const int per = cdiv(ttotal, ks);
    *tiles_per = per; *ksplit = cdiv(ttotal, per);
    return 0;
}
int wgradT_x3_launch(const WgradArgs& a, int ksplit, int tiles_per, hipStream_t st) {
    const int lds = 3 * (UATILE + UDTILE);
    USTRUN_TRY(ensure_dynamic_lds((const void*)wgradT_x3_kernel, lds, "wgradT_x3"));
    dim3 grid((a.Cin / 64) * (a.Cout / 64), ksplit), block(256);
    hipLaunchKernelGGL(wgradT_x3_kernel, grid, block, lds, st, a, a.Cout / 64, cdiv(a.Wb, TW), cdiv(a.Hb, UTH), tiles_per);
    USTRUN_LAUNCH_CHECK("wgradT_x3");
    return 0;
}

int wgrad_x3_launch(const WgradArgs& a, int ksplit, int tiles_per, hipStream_t st) {
    const int lds = 2 * WSTAGE2;
    USTRUN_TRY(ensure_dynamic_lds((const void*)wgrad_x3_kernel, lds, "wgrad_x3"));
    dim3 grid((a.Cin / 64) * (a.Cout / 64), ksplit), block(512);
    hipLaunchKernelGGL(wgrad_x3_kernel, grid, block, lds, st, a, a.Cout / 64, cdiv(a.Wb, TW), cdiv(a.Hb, TH), tiles_per);
    USTRUN_LAUNCH_CHECK("wgrad_x3");
    return 0;
}

}  // namespace ustrun
