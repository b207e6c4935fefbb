// conv_first.hip -- the network's first convolution (inc.conv0: C = 1..4 input channels, NCHW input,
// 64 outputs) and its weight gradient.  With K = 9*C <= 36 this layer is not a matrix-core problem:
// forward is HBM-bound on its 64-channel output (268 B/pixel written vs 12 B read), the weight gradient
// on reading dY once.
//
//  * forward: direct f32 stencil on the VALU.  8x16-pixel tile per block, lane = pixel, wave = group of
//    16 output channels; the 10x18xC input patch sits in LDS, the 27x16 weights of a wave are wave-uniform
//    (scalar loads, SGPR operands of v_fma).  Epilogue: NHWC f32 store + BatchNorm-statistics partials.
//  * weight gradient (bf16 MFMA): dW[(c,kh,kw)][co] = sum_p x[c][p+(kh,kw)-1] * dY[p][co] as a
//    [32 x pixels] x [pixels x 64] GEMM; the im2col rows are built in LDS from the NCHW patch (three
//    kw-shifted copies keep the 16-byte fragment reads aligned), dY tiles are read transposed.
#include "common.h"
#include "loader.h"
#include <type_traits>

namespace ustrun {
namespace {

typedef __attribute__((ext_vector_type(8))) elt_t bf16x8;
typedef __attribute__((ext_vector_type(4))) elt_t bf16x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int FTH = 8, FTW = 16, FHW = FTW + 2, FHP = (FTH + 2) * FHW;   // 10 x 18 patch
constexpr int CMAX = 4;

// x NCHW [N,C,H,W] (strides given), w = packed forward weights (f32 [9][C][64] or bf16 [9][C/8][64][8]),
// y NHWC [N,H,W,64], stat [tiles][2][64]
template <int ESZ>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const float* __restrict__ x, long sN, long sC, long sH, long sW,
                                                            int C, int H, int W, const void* __restrict__ w, int wbf16,
                                                            float* __restrict__ y, float* __restrict__ stat,
                                                            int tiles_x, int tiles_y) {
    __shared__ float patch[CMAX][FHP];
    __shared__ __attribute__((aligned(16))) float wl[CMAX * 9][64];   // weights as [c*9+t][co]
    __shared__ float red[4][2][16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 16 output channels per wave
    const int img = blockIdx.x / (tiles_y * tiles_x);
    const int rem = blockIdx.x - img * tiles_y * tiles_x;
    const int y0 = (rem / tiles_x) * FTH, x0 = (rem % tiles_x) * FTW;
    for (int t = tid; t < C * FHP; t += 256) {
        const int c = t / FHP, hp = t - c * FHP;
        const int iy = y0 + hp / FHW - 1, ix = x0 + hp % FHW - 1;
        patch[c][hp] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[img * sN + c * sC + iy * sH + ix * sW] : 0.f;
    }
    for (int t = tid; t < C * 9 * 64; t += 256) {                    // packed forward weights -> [c*9+tap][co]
        const int co = t & 63, k = t >> 6, c = k / 9, tap = k - c * 9;
        wl[k][co] = wbf16 ? (float)((const elt_t*)w)[(((long)tap * ((C + 7) / 8) + c / 8) * 64 + co) * 8 + (c & 7)]
                          : ((const float*)w)[((long)tap * C + c) * 64 + co];
    }
    __syncthreads();
    float s1[16], s2[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int g = 0; g < 2; ++g) {                                   // two 4x16 pixel groups per block
        const int py = g * 4 + (lane >> 4), px = lane & 15;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float v = patch[c][(py + t / 3) * FHW + px + t % 3];
                const float* wk = &wl[c * 9 + t][wave * 16];          // wave-uniform address: LDS broadcast
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w4 = *(const f32x4*)(wk + 4 * q);
                    acc[4 * q] = fmaf(v, w4[0], acc[4 * q]); acc[4 * q + 1] = fmaf(v, w4[1], acc[4 * q + 1]);
                    acc[4 * q + 2] = fmaf(v, w4[2], acc[4 * q + 2]); acc[4 * q + 3] = fmaf(v, w4[3], acc[4 * q + 3]);
                }
            }
        const int oy = y0 + py, ox = x0 + px;
        if (oy < H && ox < W) {
            const long o = (((long)img * H + oy) * W + ox) * 64 + wave * 16;
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = rndt<ESZ>(acc[j]);        // statistics see the stored value
#pragma unroll
            for (int q = 0; q < 4; ++q) st4t<ESZ>(y, o + 4 * q, (f32x4){acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]});
#pragma unroll
            for (int j = 0; j < 16; ++j) { s1[j] += acc[j]; s2[j] += acc[j] * acc[j]; }
        }
    }
    if (stat) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float a = s1[j], b = s2[j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
            if (lane == 0) { red[wave][0][j] = a; red[wave][1][j] = b; }
        }
        __syncthreads();
        if (tid < 128) {
            const int q = tid >> 6, c = tid & 63;
            stat[((long)blockIdx.x * 2 + q) * 64 + c] = red[c >> 4][q][c & 15];
        }
    }
}

// ---- forward on the bf16 matrix cores (dtype = bf16) ---------------------------------------------
// The f32 stencil above is VALU-bound (27*64 FMAs per pixel: 134 us at N=16, 256^2) while the layer only has to
// write 134 MB.  As an im2col GEMM [pixels x 9C] x [9C x 64] with K padded to 16*KS it is a handful of MFMAs:
// 8x32-pixel tile per block, the (10 x 34 x C) f32 patch in LDS, every lane gathers the 8 K-values of its pixel
// with ds_read_b32 at per-lane constant offsets (k -> (c, tap)) and rounds them to bf16; the weight fragments are
// block constants kept in registers.  Epilogue as in conv_halo_bf16.hip: per-wave LDS transpose, 16-byte stores,
// BatchNorm-statistics partials of the stored values (one row per block).
constexpr int MTH = 8, MTW = 32, MHW = MTW + 2, MHP = (MTH + 2) * MHW;   // 10 x 34 patch

template <int KS>
__global__ __launch_bounds__(256, 4) void conv_first_fwd_mfma_kernel(const float* __restrict__ x, long sN, long sC, long sH, long sW,
                                                                    int C, int H, int W, const elt_t* __restrict__ w,
                                                                    elt_t* __restrict__ y, float* __restrict__ stat,
                                                                    int tiles_x, int tiles_y) {
    constexpr int EPITCH = 144;
    __shared__ float patch[CMAX * MHP + 4];                 // [c][10][34]; the last word stays zero (K padding)
    __shared__ __attribute__((aligned(16))) elt_t Bw[KS * 2][64][8];
    __shared__ __attribute__((aligned(16))) char eps[4 * 32 * EPITCH];
    __shared__ float red[4][2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int img = blockIdx.x / (tiles_y * tiles_x);
    const int rem = blockIdx.x - img * tiles_y * tiles_x;
    const int y0 = (rem / tiles_x) * MTH, x0 = (rem % tiles_x) * MTW;
    const int K = 9 * C;
    for (int t = tid; t < C * MHP; t += 256) {
        const int c = t / MHP, hp = t - c * MHP;
        const int iy = y0 + hp / MHW - 1, ix = x0 + hp % MHW - 1;
        patch[t] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[img * sN + c * sC + iy * sH + ix * sW] : 0.f;
    }
    if (tid < 4) patch[CMAX * MHP + tid] = 0.f;
    for (int t = tid; t < KS * 2 * 64 * 8; t += 256) {       // k = c*9 + tap; packed forward weights are [tap][1][64][8 (c)]
        const int j = t & 7, co = (t >> 3) & 63, k8 = t >> 9, k = k8 * 8 + j;
        const int c = k / 9, tap = k - c * 9;
        Bw[k8][co][j] = k < K ? w[((long)tap * 64 + co) * 8 + c] : (elt_t)0.f;
    }
    // this lane's K entries: k = 16 ks + 8 lh + j -> patch offset relative to the pixel's (0,0) tap
    int koff[KS][8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * ks + 8 * lh + j, c = k / 9, tap = k - c * 9;
            koff[ks][j] = k < K ? c * MHP + (tap / 3) * MHW + tap % 3 : -1;
        }
    __syncthreads();
    bf16x8 bfr[KS][2];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        bfr[ks][0] = *(const bf16x8*)&Bw[2 * ks + lh][l31][0];
        bfr[ks][1] = *(const bf16x8*)&Bw[2 * ks + lh][32 + l31][0];
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {                            // wave owns tile rows 2*wave + i, 32 pixels each
        const int base = (2 * wave + i) * MHW + l31;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 af;
#pragma unroll
            for (int j = 0; j < 8; ++j) af[j] = (elt_t)patch[koff[ks][j] >= 0 ? base + koff[ks][j] : CMAX * MHP];
            acc[i][0] = USTRUN_MFMA_32x32x16(af, bfr[ks][0], acc[i][0], 0, 0, 0);
            acc[i][1] = USTRUN_MFMA_32x32x16(af, bfr[ks][1], acc[i][1], 0, 0, 0);
        }
    }
    // epilogue
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    char* ep = eps + wave * (32 * EPITCH);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int oy = y0 + 2 * wave + i;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const elt_t hv = (elt_t)acc[i][n][r];
                *(elt_t*)(ep + row * EPITCH + (n * 32 + l31) * 2) = hv;
                if (oy < H && x0 + row < W) {
                    const float v = (float)hv;                 // statistics see the stored value
                    s1[n] += v; s2[n] += v * v;
                }
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = lane + 64 * t, row = idx >> 3, ch = idx & 7;
            const bf16x8 v8 = *(const bf16x8*)(ep + row * EPITCH + ch * 16);
            const int ox = x0 + row;
            if (oy < H && ox < W) *(bf16x8*)(y + (((long)img * H + oy) * W + ox) * 64 + ch * 8) = v8;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (stat) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            s1[n] += __shfl_xor(s1[n], 32);
            s2[n] += __shfl_xor(s2[n], 32);
            if (lh == 0) { red[wave][0][n * 32 + l31] = s1[n]; red[wave][1][n * 32 + l31] = s2[n]; }
        }
        __syncthreads();
        if (tid < 128) {
            const int q = tid >> 6, c = tid & 63;
            stat[((long)blockIdx.x * 2 + q) * 64 + c] = red[0][q][c] + red[1][q][c] + red[2][q][c] + red[3][q][c];
        }
    }
}

// ---- weight gradient -----------------------------------------------------------------------
// block: a range of 8x16 tiles; wave w owns tile rows {2w, 2w+1}: D[32 im2col rows][64 co] per wave,
// waves summed through LDS at the end, one slab per block: slab[(c*9 + t)][co] (rows >= 9C unused)
constexpr int WRB = 192;     // dY LDS row pitch (64 bf16 + pad: conflict-free transposed reads)

template <int ESZ>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(const float* __restrict__ x, long sN, long sC, long sH, long sW,
                                                              int C, int H, int W, const float* __restrict__ dy,
                                                              float* __restrict__ partials, int tiles_x, int tiles_y,
                                                              int ttotal, int tiles_per) {
    // im2col source: xs[c][kw][hy][16] bf16 = patch shifted by kw so that 8 consecutive pixels are 16-B aligned
    __shared__ __attribute__((aligned(16))) elt_t xs[CMAX][3][FTH + 2][FTW];
    // (the end-of-block sum of the four waves' accumulators reuses the dY tile's memory: 36 KB per block instead of 60 -- all
    // 1024 blocks, four per CU, are resident at once; with 60 KB two per CU were, and the kernel ran its grid in two rounds)
    constexpr int DYSB = FTH * FTW * WRB > 4 * 32 * 64 * 4 ? FTH * FTW * WRB : 4 * 32 * 64 * 4;
    __shared__ __attribute__((aligned(16))) char dys[DYSB];
    float (*accs)[32][64] = (float (*)[32][64])dys;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // this lane's im2col row i = l31 -> (c, kh, kw); rows >= 9C are zero
    const int irow = l31, ic = irow / 9, it = irow % 9, ikh = it / 3, ikw = it % 3;
    const bool ivalid = irow < 9 * C;
    const int lrow = 8 * lh + ((lane & 15) >> 2), lcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);

    const int tbeg = blockIdx.x * tiles_per, tend = min(ttotal, tbeg + tiles_per);
    // Register-staged software pipeline: the next tile's dY (16 B per item) and x elements are loaded while the current
    // tile's MFMAs run (the synchronous load -> barrier -> compute loop spent 16 us per 16 KB tile).  The decomposition
    // of this thread's x elements into (channel, kw, patch row, pixel) does not depend on the tile.
    constexpr int XIT = (CMAX * 3 * (FTH + 2) * FTW + 255) / 256;
    constexpr int DIT = FTH * FTW * (ESZ == 2 ? 8 : 16) / 256;       // 16-byte items: 8 bf16 or 4 f32 channels
    int xdec[XIT];
#pragma unroll
    for (int i = 0; i < XIT; ++i) {
        const int e = tid + 256 * i;
        const int px = e % FTW, hy = (e / FTW) % (FTH + 2), kw = (e / (FTW * (FTH + 2))) % 3, c = e / (FTW * (FTH + 2) * 3);
        xdec[i] = e < C * 3 * (FTH + 2) * FTW ? (c << 16) | (kw << 12) | (hy << 6) | px : -1;
    }
    float xr[XIT];
    f32x4 dr[DIT];
    auto prefetch = [&](int t) {
        const int img = t / (tiles_y * tiles_x);
        const int rem = t - img * tiles_y * tiles_x;
        const int y0 = (rem / tiles_x) * FTH, x0 = (rem % tiles_x) * FTW;
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
            const int d = xdec[i];
            const int c = d >> 16, kw = (d >> 12) & 3, hy = (d >> 6) & 63, px = d & 63;
            const int iy = y0 + hy - 1, ix = x0 + px + kw - 1;
            xr[i] = (d >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W) ? x[img * sN + c * sC + iy * sH + ix * sW] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < DIT; ++i) {
            const int e = tid + 256 * i;
            const int p = ESZ == 2 ? e >> 3 : e >> 4, g = ESZ == 2 ? e & 7 : e & 15;
            const int oy = y0 + (p >> 4), ox = x0 + (p & 15);
            dr[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (oy < H && ox < W)
                dr[i] = *(const f32x4*)((const char*)dy + ((((long)img * H + oy) * W + ox) * 64 + (ESZ == 2 ? 8 : 4) * g) * ESZ);
        }
    };
    if (tbeg < tend) prefetch(tbeg);
    for (int t = tbeg; t < tend; ++t) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < XIT; ++i) {
            const int d = xdec[i];
            if (d >= 0) xs[d >> 16][(d >> 12) & 3][(d >> 6) & 63][d & 63] = (elt_t)xr[i];
        }
#pragma unroll
        for (int i = 0; i < DIT; ++i) {
            const int e = tid + 256 * i;
            if (ESZ == 2) {
                *(f32x4*)(dys + (e >> 3) * WRB + (e & 7) * 16) = dr[i];          // 8 bf16, as stored
            } else {
                bf16x4 h;
                h[0] = (elt_t)dr[i][0]; h[1] = (elt_t)dr[i][1]; h[2] = (elt_t)dr[i][2]; h[3] = (elt_t)dr[i][3];
                *(bf16x4*)(dys + (e >> 4) * WRB + (e & 15) * 8) = h;
            }
        }
        if (t + 1 < tend) prefetch(t + 1);
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = wave * 2 + rr;                              // tile row = 16 pixels = one MFMA k-step
            bf16x8 a;
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = (elt_t)0.f;
            if (ivalid) a = *(const bf16x8*)&xs[ic][ikw][r + ikh][8 * lh];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const char* base = dys + (r * FTW + lrow) * WRB + (j * 32 + lcol) * 2;
                const bf16x4 lo = USTRUN_DS_READ_TR16((lds_bf16x4*)base);
                const bf16x4 hi = USTRUN_DS_READ_TR16((lds_bf16x4*)(base + 4 * WRB));
                bf16x8 b;
                b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = lo[3]; b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2]; b[7] = hi[3];
                acc[j] = USTRUN_MFMA_32x32x16(a, b, acc[j], 0, 0, 0);
            }
        }
    }
    // sum the four waves (fixed order) and write the block's slab
    __syncthreads();                              // (every wave is done reading the last dY tile: accs lives in its memory)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][j * 32 + l31] = acc[j][r];
    __syncthreads();
    float* slab = partials + (long)blockIdx.x * 32 * 64;
    for (int e = tid; e < 32 * 64; e += 256) {
        const int i = e >> 6, co = e & 63;
        slab[e] = accs[0][i][co] + accs[1][i][co] + accs[2][i][co] + accs[3][i][co];
    }
}

// ---- weight gradient, streaming form (round 5) -----------------------------------------------------------------------------
// The layer is a 537 MB read of dY (N = 64) beside 17 GFLOP: the tile kernel above moved it at 2.8 TB/s -- 4 MFMAs per wave between
// two block barriers per 16 KB tile, one tile of prefetch.  Here nothing is shared between waves: a wave owns a 16-pixel-wide strip
// of `seg_rows` rows of one image and walks it a row a step; a step's dY row piece (16 px x 64 ch = 2 KB, contiguous) and the eight
// x values of the lane's im2col row (c, kh, kw) -- straight from the NCHW image, which stays in L2 -- are fetched into REGISTERS
// four steps ahead (the compiler counts those waits itself), the dY piece passes through a wave-private LDS slot for the
// transposing fragment reads, two MFMAs, next step.  No barrier until the block's four accumulator sets are summed at the end.
template <int D>
__global__ __launch_bounds__(256) void conv_first_wgrad_stream_kernel(const float* __restrict__ x, int sN, int sC, int sH, int C,
                                                                      int H, int W, int xbytes, const elt_t* __restrict__ dy,
                                                                      int dybytes, float* __restrict__ partials, int strips, int segs,
                                                                      int seg_rows, int items) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    __shared__ __attribute__((aligned(16))) char dys[4 * 2 * FTW * WRB > 4 * 32 * 64 * 4 ? 4 * 2 * FTW * WRB : 4 * 32 * 64 * 4];
    float (*accs)[32][64] = (float (*)[32][64])dys;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int item = blockIdx.x * 4 + wave;
    if (item < items) {
        // (strip fastest: the four waves of a block read four neighbouring 2 KB pieces of the same dY rows)
        const int sx = item % strips, sg = (item / strips) % segs, img = item / (segs * strips);
        const int x0 = sx * FTW, r0 = sg * seg_rows, r1 = min(H, r0 + seg_rows);
        // this lane's im2col row i = l31 -> (c, kh, kw); rows >= 9C are zero
        const int ic = l31 / 9, it = l31 % 9, ikh = it / 3, ikw = it % 3;
        const bool ivalid = l31 < 9 * C;
        // Both tensors are read through ONE buffer resource each, every load issued unconditionally: an offset outside the tensor
        // (one float in front of the first row, a few behind the last) fails the range check and returns zeros; what must not
        // count -- image rows above / below, columns left / right of the image, im2col rows past 9 C, rows past the segment --
        // is masked when the values are USED, four steps later: a select around a load would put its wait right behind the load.
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, xbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, dybytes, 0x00020000);
        const int col0 = x0 + 8 * lh + ikw - 1;                       // first of the lane's eight pixels' x columns
        // (the one lane column that starts a float in FRONT of its row -- x0 = 0, kw = 0 -- would start the tensor's very first row at
        // offset -4: the whole 16-byte load fails the range check, not just its first word.  Those lanes fetch one column to the
        // right and shift the values back when they use them.)
        const bool shl = col0 < 0;
        const int xoff0 = (img * sN + (ivalid ? ic : 0) * sC + col0 + (shl ? 1 : 0)) * 4;
        unsigned cm[8];                                               // per column: all ones where it lies in the image
#pragma unroll
        for (int q = 0; q < 8; ++q) cm[q] = (ivalid && (unsigned)(col0 + q) < (unsigned)W) ? 0xffffffffu : 0u;
        // dY: lane fetches 16-byte piece 64 i + lane of the row's 2 KB: pixel (64 i + lane) >> 3, channel group lane & 7
        const int dpx0 = lane >> 3, dpx1 = 8 + (lane >> 3), dcg = lane & 7;
        const unsigned dm0 = x0 + dpx0 < W ? 0xffffffffu : 0u, dm1 = x0 + dpx1 < W ? 0xffffffffu : 0u;
        const int doff0 = ((img * H * W + x0 + dpx0) * 64 + dcg * 8) * 2, doff1 = doff0 + 8 * 64 * 2;
        char* slot = dys + wave * (2 * FTW * WRB);
        const int lrow = 8 * lh + ((lane & 15) >> 2), lcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        u32x4 xr[D][2], dr[D][2];
        unsigned rm[D];                                               // per slot: all ones where the lane's x row lies in the image
        auto prefetch = [&](int u, int y) {
            const int yy = y + ikh - 1;
            rm[u] = (y < r1 && (unsigned)yy < (unsigned)H) ? 0xffffffffu : 0u;
            const int xo = xoff0 + yy * sH * 4;
            xr[u][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xo, 0, 0));
            xr[u][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xo + 16, 0, 0));
            const int dofs = min(y, r1 - 1) * W * 128;
            dr[u][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, doff0 + dofs, 0, 0));
            dr[u][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, doff1 + dofs, 0, 0));
        };
        auto consume = [&](int u, int par, bool live) {
            char* sl = slot + par * (FTW * WRB);
            u32x4 d0 = dr[u][0], d1 = dr[u][1];
            const unsigned lm = live ? 0xffffffffu : 0u;              // (wave-uniform: rows past the segment add nothing)
#pragma unroll
            for (int q = 0; q < 4; ++q) { d0[q] &= dm0 & lm; d1[q] &= dm1 & lm; }
            *(u32x4*)(sl + dpx0 * WRB + dcg * 16) = d0;
            *(u32x4*)(sl + dpx1 * WRB + dcg * 16) = d1;
            bf16x8 a;
            unsigned v[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = xr[u][0][q]; v[4 + q] = xr[u][1][q]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned t = shl ? (q ? v[q - 1] : 0u) : v[q];
                a[q] = (elt_t)__builtin_bit_cast(float, t & cm[q] & rm[u]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const char* base = sl + lrow * WRB + (j * 32 + lcol) * 2;
                const bf16x4 lo = USTRUN_DS_READ_TR16((lds_bf16x4*)base);
                const bf16x4 hi = USTRUN_DS_READ_TR16((lds_bf16x4*)(base + 4 * WRB));
                bf16x8 b;
                b[0] = lo[0]; b[1] = lo[1]; b[2] = lo[2]; b[3] = lo[3]; b[4] = hi[0]; b[5] = hi[1]; b[6] = hi[2]; b[7] = hi[3];
                acc[j] = USTRUN_MFMA_32x32x16(a, b, acc[j], 0, 0, 0);
            }
        };
#pragma unroll
        for (int u = 0; u < D; ++u) prefetch(u, r0 + u);
        for (int y = r0; y < r1; y += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                consume(u, u & 1, y + u < r1);
                prefetch(u, y + u + D);
            }
        }
    }
    // sum the four waves (fixed order) and write the block's slab
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][j * 32 + l31] = acc[j][r];
    __syncthreads();
    float* slab = partials + (long)blockIdx.x * 32 * 64;
    for (int e = tid; e < 32 * 64; e += 256) {
        const int i = e >> 6, co = e & 63;
        slab[e] = accs[0][i][co] + accs[1][i][co] + accs[2][i][co] + accs[3][i][co];
    }
}

// ---- ... and for dtype USTRUN_F32X3 (f32 dY): the same walk with both operands split into three bf16 terms at use (x3.hip's
// arithmetic: six MFMAs per product, f32-level result): 4 KB of dY per step, three LDS planes per slot.
template <int D>
__global__ __launch_bounds__(256) void conv_first_wgrad_stream_x3_kernel(const float* __restrict__ x, int sN, int sC, int sH, int C,
                                                                         int H, int W, int xbytes, const float* __restrict__ dy,
                                                                         int dybytes, float* __restrict__ partials, int strips, int segs,
                                                                         int seg_rows, int items) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    typedef __attribute__((ext_vector_type(8))) __bf16 b16x8;
    typedef __attribute__((ext_vector_type(4))) __bf16 b16x4;
    constexpr int SLOT = 3 * FTW * WRB;                                // three planes of [16 px][WRB]
    extern __shared__ __attribute__((aligned(16))) char dys[];         // 4 waves x 2 slots (>= 4 x 32 x 64 floats for the final sum)
    float (*accs)[32][64] = (float (*)[32][64])dys;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    auto split1 = [](float v, __bf16& h0, __bf16& h1, __bf16& h2) {
        h0 = (__bf16)v;
        const float r1 = v - (float)h0;
        h1 = (__bf16)r1;
        h2 = (__bf16)(r1 - (float)h1);
    };
    const int item = blockIdx.x * 4 + wave;
    if (item < items) {
        const int sx = item % strips, sg = (item / strips) % segs, img = item / (segs * strips);
        const int x0 = sx * FTW, r0 = sg * seg_rows, r1 = min(H, r0 + seg_rows);
        const int ic = l31 / 9, it = l31 % 9, ikh = it / 3, ikw = it % 3;
        const bool ivalid = l31 < 9 * C;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, xbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, dybytes, 0x00020000);
        const int col0 = x0 + 8 * lh + ikw - 1;
        const bool shl = col0 < 0;                                     // (see the 16-bit kernel: the one lane column that starts in front of its row)
        const int xoff0 = (img * sN + (ivalid ? ic : 0) * sC + col0 + (shl ? 1 : 0)) * 4;
        unsigned cm[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) cm[q] = (ivalid && (unsigned)(col0 + q) < (unsigned)W) ? 0xffffffffu : 0u;
        // dY: lane fetches 16-byte piece 64 i + lane (i < 4) of the row's 4 KB: pixel 4 i + (lane >> 4), 4-channel group lane & 15
        const int dpx = lane >> 4, dcg = lane & 15;
        unsigned dm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dm[i] = x0 + 4 * i + dpx < W ? 0xffffffffu : 0u;
        const int doff0 = ((img * H * W + x0 + dpx) * 64 + dcg * 4) * 4;
        char* slot = dys + wave * (2 * SLOT);
        const int lrow = 8 * lh + ((lane & 15) >> 2), lcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        u32x4 xr[D][2], dr[D][4];
        unsigned rm[D];
        auto prefetch = [&](int u, int y) {
            const int yy = y + ikh - 1;
            rm[u] = (y < r1 && (unsigned)yy < (unsigned)H) ? 0xffffffffu : 0u;
            const int xo = xoff0 + yy * sH * 4;
            xr[u][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xo, 0, 0));
            xr[u][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xo + 16, 0, 0));
            const int dofs = min(y, r1 - 1) * W * 256;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dr[u][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, doff0 + dofs + i * 4 * 256, 0, 0));
        };
        auto consume = [&](int u, int par, bool live) {
            char* sl = slot + par * SLOT;
            const unsigned lm = live ? 0xffffffffu : 0u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                b16x4 h0, h1, h2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __bf16 t0, t1, t2;
                    split1(__builtin_bit_cast(float, dr[u][i][q] & dm[i] & lm), t0, t1, t2);
                    h0[q] = t0; h1[q] = t1; h2[q] = t2;
                }
                char* dst = sl + (4 * i + dpx) * WRB + dcg * 8;
                *(u32x2*)dst = __builtin_bit_cast(u32x2, h0);
                *(u32x2*)(dst + FTW * WRB) = __builtin_bit_cast(u32x2, h1);
                *(u32x2*)(dst + 2 * FTW * WRB) = __builtin_bit_cast(u32x2, h2);
            }
            b16x8 a[3];
            unsigned v[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) { v[q] = xr[u][0][q]; v[4 + q] = xr[u][1][q]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned t = shl ? (q ? v[q - 1] : 0u) : v[q];
                __bf16 t0, t1, t2;
                split1(__builtin_bit_cast(float, t & cm[q] & rm[u]), t0, t1, t2);
                a[0][q] = t0; a[1][q] = t1; a[2][q] = t2;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                b16x8 b[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const char* base = sl + p * FTW * WRB + lrow * WRB + (j * 32 + lcol) * 2;
                    const b16x4 lo = __builtin_bit_cast(b16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ustrun_lds_s16x4*)base));
                    const b16x4 hi = __builtin_bit_cast(b16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ustrun_lds_s16x4*)(base + 4 * WRB)));
                    b[p][0] = lo[0]; b[p][1] = lo[1]; b[p][2] = lo[2]; b[p][3] = lo[3];
                    b[p][4] = hi[0]; b[p][5] = hi[1]; b[p][6] = hi[2]; b[p][7] = hi[3];
                }
                f32x16 c = acc[j];                     // small terms first (x3.hip)
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
                acc[j] = c;
            }
        };
#pragma unroll
        for (int u = 0; u < D; ++u) prefetch(u, r0 + u);
        for (int y = r0; y < r1; y += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                consume(u, u & 1, y + u < r1);
                prefetch(u, y + u + D);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accs[wave][(r & 3) + 8 * (r >> 2) + 4 * lh][j * 32 + l31] = acc[j][r];
    __syncthreads();
    float* slab = partials + (long)blockIdx.x * 32 * 64;
    for (int e = tid; e < 32 * 64; e += 256) {
        const int i = e >> 6, co = e & 63;
        slab[e] = accs[0][i][co] + accs[1][i][co] + accs[2][i][co] + accs[3][i][co];
    }
}

// dw[co][c][t] (+)= sum_k partials[k][c*9+t][co].  Block = 8 outputs (consecutive co of one im2col row: 32 contiguous bytes per
// slab) x 128 slab lanes; a lane's slabs k = lane, lane + 128, .. are ALL loaded before the first add (up to eight in flight: the
// 1024-slab table is one memory round trip, where 32 lanes x four in flight made it eight dependent ones -- 35 us for 8 MB), summed
// in ascending k, then the lanes in ascending order: a fixed order, f64.
__global__ __launch_bounds__(1024) void conv_first_wgrad_reduce_kernel(const float* __restrict__ partials, int nslab, int C,
                                                                     float* __restrict__ dw, int accumulate) {
    __shared__ double red[128][8];
    const int ol = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int e = blockIdx.x * 8 + ol;                   // (im2col row i, co) in i-major order
    const int i = e >> 6, co = e & 63;
    double v = 0.0;
    if (i < 9 * C) {
        const float* p0 = partials + (long)i * 64 + co;
        for (int k0 = sl; k0 < nslab; k0 += 1024) {
            float a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int k = k0 + 128 * u; a[u] = p0[(long)(k < nslab ? k : k0) * 2048]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) v += (k0 + 128 * u < nslab) ? (double)a[u] : 0.0;
        }
    }
    red[sl][ol] = v;
    __syncthreads();
    if (sl == 0 && i < 9 * C) {
        for (int k = 1; k < 128; ++k) v += red[k][ol];
        const int o = co * (C * 9) + i;
        dw[o] = accumulate ? dw[o] + (float)v : (float)v;
    }
}

// ---- forward, streaming form (round 4) -------------------------------------------------------------------------------------
// The layer writes 128 B per pixel and reads 12: it is a store stream with a little arithmetic in front.  The tile-per-block
// kernel above pays its whole setup (weight fragments through LDS, k -> patch offsets with divisions, a patch fill whose loads
// nothing overlaps) for 256 pixels and then stores through 2-byte LDS writes: 3.0-3.2 TB/s of output where the BatchNorm
// passes, which move the same kind of bytes, reach 5.4.  Here a block owns a 32-pixel-wide strip of `seg_rows` rows of one
// image and walks it 8 rows per step: weight fragments (the A operand: rows = output channels) and the k -> patch index
// tables are built once per block and live in registers; the next step's 10 x 34 x C input patch is fetched into registers
// under the current step's work and lands in the other LDS buffer; the product is D[channel][pixel], so four consecutive
// channels of a pixel sit in four consecutive accumulator registers: 2 cvt_pk + ONE ds_write_b64 per group into a per-wave
// scratch, read back as 16-byte pieces (8 channels of one pixel per lane) that go out as whole 128-byte lines; the BatchNorm
// statistics are taken from those pieces (8 channels per lane, 16 accumulators) and leave the block ONCE, as one row per block.
constexpr int SPITCH = 144;          // scratch row: 128 B of channels + 16 B pad (16-byte aligned rows)

template <int C, bool STAT>
__global__ __launch_bounds__(256, 3) void conv_first_fwd_stream_kernel(const float* __restrict__ x, long sN, int sC, int sH, int sW,
                                                                      int H, int W, const elt_t* __restrict__ w,
                                                                      elt_t* __restrict__ y, float* __restrict__ stat,
                                                                      int strips, int segs, int seg_rows, int img_bytes) {
    constexpr int KS = (9 * C + 15) / 16;
    constexpr int PROWS = MTH + 2;                          // patch rows of a step
    // one patch buffer (floats): [c][10][34], then a ZERO AREA that padded k entries read: a lane's k -> patch index table is
    // built for its first row of a step and the second row is the same table + one patch row (an immediate offset), so a
    // padded entry reads ZIDX and ZIDX + MHW
    constexpr int ZIDX = CMAX * MHP;
    constexpr int PSZ = ZIDX + MHW + 6;
    __shared__ float patch[2 * PSZ];
    __shared__ __attribute__((aligned(16))) char scr[4 * 32 * SPITCH];
    __shared__ float red[4][2][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: the stores' row offset is a scalar operand
    const int l31 = lane & 31, lh = lane >> 5;
    const int item = blockIdx.x;
    const int sx = item % strips, sy = (item / strips) % segs, img = item / (strips * segs);
    const int x0 = sx * MTW, r0 = sy * seg_rows, r1 = min(H, r0 + seg_rows);
    const int nsteps = (r1 - r0 + MTH - 1) / MTH;
    constexpr int K = 9 * C;
    // every access to the image and to the output goes through a buffer resource of ONE image: an offset with bit 31 set fails
    // the range check (loads return 0, stores are dropped), so padding and ragged edges need no branch around a memory
    // instruction (hipcc turns a select around a load into a branch with its own wait: the prefetch would serialise)
    constexpr int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + img * sN), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (long)img * H * W * 64), 0, H * W * 128, 0x00020000);
    // weight fragments, A operand of D[co][px] = W[co][k] P[k][px]: lane (row = co, half lh) holds k = 16 ks + 8 lh + j;
    // packed forward weights are [tap][1][64][8 (c)]
    bf16x8 wf[KS][2];
    int kidx[KS][8];                                       // BYTE offset into a patch buffer of k's entry for (row 2 wave, pixel l31)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * ks + 8 * lh + j, c = k / 9, tap = k - c * 9;
            const bool kv = k < K;
            wf[ks][0][j] = kv ? w[((long)tap * 64 + l31) * 8 + c] : (elt_t)0.f;
            wf[ks][1][j] = kv ? w[((long)tap * 64 + 32 + l31) * 8 + c] : (elt_t)0.f;
            kidx[ks][j] = (kv ? c * MHP + (tap / 3 + 2 * wave) * MHW + tap % 3 + l31 : ZIDX) * 4;
        }
    // patch fetch: thread t < 34 C owns patch column (c, px) = (t / 34, t % 34) and fetches its 10 rows -- the row part of the
    // address and its validity are wave-uniform, the column part is a per-thread constant
    const bool loader = tid < C * MHW;
    const int lc = tid / MHW, lpx = tid - lc * MHW, lix = x0 - 1 + lpx;
    const int coloff = (loader && lix >= 0 && lix < W) ? (lc * sC + lix * sW) * 4 : OOB;
    const int lds_col = (lc * MHP + lpx) * 4;
    float pre[PROWS];
    auto load = [&](int s) {
        const int yb = r0 + MTH * s - 1;
#pragma unroll
        for (int i = 0; i < PROWS; ++i) {
            const int iy = yb + i;                          // (uniform)
            const int voff = (iy >= 0 && iy < H) ? coloff : OOB;
            pre[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, voff, iy * sH * 4, 0));
        }
    };
    for (int t = tid; t < MHW + 6; t += 256) { patch[ZIDX + t] = 0.f; patch[PSZ + ZIDX + t] = 0.f; }
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    char* ep = scr + wave * (32 * SPITCH);
    if (loader) load(0);
    auto step = [&](int s, auto buf_c) {
        constexpr int buf = decltype(buf_c)::value;
        const char* P = (const char*)(patch + buf * PSZ);
        if (loader) {
#pragma unroll
            for (int i = 0; i < PROWS; ++i) *(float*)((char*)(patch + buf * PSZ) + lds_col + i * MHW * 4) = pre[i];
        }
        __syncthreads();                                  // (also: every wave is done with the buffer the NEXT step fills)
        if (loader && s + 1 < nsteps) load(s + 1);        // in flight under this step's work
        const int yb = r0 + MTH * s;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = yb + 2 * wave + i;
            f32x16 acc[2];
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (elt_t) * (const float*)(P + kidx[ks][j] + i * MHW * 4);
                acc[0] = USTRUN_MFMA_32x32x16(wf[ks][0], pf, acc[0], 0, 0, 0);
                acc[1] = USTRUN_MFMA_32x32x16(wf[ks][1], pf, acc[1], 0, 0, 0);
            }
            // D row (channel) = (r & 3) + 8 (r >> 2) + 4 lh, column (pixel) = l31: registers 4g .. 4g + 3 = channels 8g + 4lh + 0..3
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 h4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) h4[q] = (elt_t)acc[n][4 * g + q];
                    *(bf16x4*)(ep + l31 * SPITCH + (n * 32 + 8 * g + 4 * lh) * 2) = h4;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const bool rowok = oy < r1;                    // (uniform)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int idx = lane + 64 * t, pr = idx >> 3, ch = idx & 7;        // ch == lane & 7 for every t
                const bf16x8 v8 = *(const bf16x8*)(ep + pr * SPITCH + ch * 16);
                const int ox = x0 + pr;
                const bool ok = rowok && ox < W;
                // (the row offset rides in the VECTOR offset, soffset stays 0: with an SGPR soffset hipcc's hazard recognizer inserts no
                // wait state between a 128-bit buffer store and a VALU write of its data registers -- an SI-era rule -- and on gfx950,
                // with the store pipe saturated, the store then ships the overwritten dword for the last lanes of each 16:
                // 6e-6 of the outputs in the no-statistics build, none in the build whose statistics code sits in between)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v8), ry, ok ? ((oy * W + ox) * 64 + ch * 8) * 2 : OOB, 0, 0);
                if constexpr (STAT) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float v = ok ? (float)v8[j] : 0.f;             // statistics see the stored value
                        s1[j] += v; s2[j] += v * v;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    for (int s = 0; s < nsteps; s += 2) {
        step(s, std::integral_constant<int, 0>{});
        if (s + 1 < nsteps) step(s + 1, std::integral_constant<int, 1>{});
    }
    if constexpr (STAT) {                                  // lanes with equal lane & 7 hold the same 8 channels
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { s1[j] += __shfl_xor(s1[j], o); s2[j] += __shfl_xor(s2[j], o); }
            if (lane < 8) { red[wave][0][lane * 8 + j] = s1[j]; red[wave][1][lane * 8 + j] = s2[j]; }
        }
        __syncthreads();
        if (tid < 128) {
            const int q = tid >> 6, c = tid & 63;
            stat[((long)item * 2 + q) * 64 + c] = red[0][q][c] + red[1][q][c] + red[2][q][c] + red[3][q][c];
        }
    }
}

// rows per block of the streaming forward: 64 where that still gives every CU four blocks, else shorter segments
int stream_seg_rows(int N, int H, int W) {
    int seg = 64;
    while (seg > MTH && (long)N * cdiv(W, MTW) * cdiv(H, seg) < 1024) seg >>= 1;
    return seg;
}

}  // namespace

bool conv_first_supported(const ustrun_src_t& s, int Cout) {
    return s.C <= CMAX && Cout == 64 && !s.pool && !s.scale && !s.relu && s.off_y == 0 && s.off_x == 0;
}
int conv_first_stat_rows(int N, int H, int W, int dtype) {
    if (dtype == USTRUN_D16 && (g_debug_flags & 16384)) return N * cdiv(H, MTH) * cdiv(W, MTW);
    return dtype == USTRUN_D16 ? N * cdiv(H, stream_seg_rows(N, H, W)) * cdiv(W, MTW) : N * cdiv(H, FTH) * cdiv(W, FTW);
}

int conv_first_fwd(const ustrun_src_t& s, const void* w_fwd, int dtype, int N, void* y, float* stat, hipStream_t st) {
    if (dtype == USTRUN_D16 && !(g_debug_flags & 16384)) {       // im2col on the matrix cores, streaming form (round 4)
        const int seg = stream_seg_rows(N, s.H, s.W), strips = cdiv(s.W, MTW), segs = cdiv(s.H, seg);
        USTRUN_CHECK(s.C >= 1 && s.C <= CMAX, "conv_first_fwd: C=%d", s.C);
        const long img_elems = (long)(s.C - 1) * s.sC + (long)(s.H - 1) * s.sH + (long)(s.W - 1) * s.sW + 1;
        USTRUN_CHECK(img_elems * 4 < (1L << 31) && (long)s.H * s.W * 128 < (1L << 31), "conv_first_fwd: an image beyond 2^31 bytes");
        dim3 grid(N * segs * strips), block(256);
#define USTRUN_CFS(CC)                                                                                                              \
    do {                                                                                                                            \
        if (stat) hipLaunchKernelGGL((conv_first_fwd_stream_kernel<CC, true>), grid, block, 0, st, (const float*)s.ptr, (long)s.sN, (int)s.sC, \
                                     (int)s.sH, (int)s.sW, s.H, s.W, (const elt_t*)w_fwd, (elt_t*)y, stat, strips, segs, seg, (int)(img_elems * 4)); \
        else hipLaunchKernelGGL((conv_first_fwd_stream_kernel<CC, false>), grid, block, 0, st, (const float*)s.ptr, (long)s.sN, (int)s.sC,   \
                                (int)s.sH, (int)s.sW, s.H, s.W, (const elt_t*)w_fwd, (elt_t*)y, stat, strips, segs, seg, (int)(img_elems * 4));     \
    } while (0)
        if (s.C == 1) USTRUN_CFS(1); else if (s.C == 2) USTRUN_CFS(2); else if (s.C == 3) USTRUN_CFS(3); else USTRUN_CFS(4);
#undef USTRUN_CFS
        USTRUN_LAUNCH_CHECK("conv_first_fwd_stream");
        return 0;
    }
    if (dtype == USTRUN_D16) {            // the round-1..3 tile-per-block kernel (ustrun_debug_flags bit 14: A/B runs)
        const int tx = cdiv(s.W, MTW), ty = cdiv(s.H, MTH), ks = cdiv(9 * s.C, 16);
        dim3 grid(N * ty * tx), block(256);
#define USTRUN_CF(KS) hipLaunchKernelGGL(conv_first_fwd_mfma_kernel<KS>, grid, block, 0, st, (const float*)s.ptr, (long)s.sN, \
                                         (long)s.sC, (long)s.sH, (long)s.sW, s.C, s.H, s.W, (const elt_t*)w_fwd, (elt_t*)y, stat, tx, ty)
        if (ks == 1) USTRUN_CF(1); else if (ks == 2) USTRUN_CF(2); else USTRUN_CF(3);
#undef USTRUN_CF
        USTRUN_LAUNCH_CHECK("conv_first_fwd_mfma");
        return 0;
    }
    const int tx = cdiv(s.W, FTW), ty = cdiv(s.H, FTH);
    if (false)
        hipLaunchKernelGGL(conv_first_fwd_kernel<2>, dim3(N * ty * tx), dim3(256), 0, st, (const float*)s.ptr, (long)s.sN, (long)s.sC,
                           (long)s.sH, (long)s.sW, s.C, s.H, s.W, w_fwd, 1, (float*)y, stat, tx, ty);
    else
        hipLaunchKernelGGL(conv_first_fwd_kernel<4>, dim3(N * ty * tx), dim3(256), 0, st, (const float*)s.ptr, (long)s.sN, (long)s.sC,
                           (long)s.sH, (long)s.sW, s.C, s.H, s.W, w_fwd, 0, (float*)y, stat, tx, ty);
    USTRUN_LAUNCH_CHECK("conv_first_fwd");
    return 0;
}

int64_t conv_first_wgrad_partials_bytes() { return (int64_t)1024 * 32 * 64 * sizeof(float); }

int conv_first_wgrad(const ustrun_src_t& s, const void* dy, int dy_esz, int N, float* dw, int accumulate, float* partials,
                     int64_t partials_bytes, hipStream_t st, bool x3) {
    USTRUN_CHECK(partials_bytes >= conv_first_wgrad_partials_bytes(), "conv_first_wgrad: partials too small");
    const int tx = cdiv(s.W, FTW), ty = cdiv(s.H, FTH), ttotal = N * ty * tx;
    int blocks = ttotal < 1024 ? ttotal : 1024;
    const int per = cdiv(ttotal, blocks);
    blocks = cdiv(ttotal, per);
    // (ustrun_debug_flags bit 28: the tile kernel of rounds 1-4, for A/B runs)
    if (dy_esz == 4 && x3) {            // dtype USTRUN_F32X3: f32 dY, three-term products
        const long xb = (long)N * s.sN * 4, db = (long)N * s.H * s.W * 64 * 4;
        USTRUN_CHECK(s.sW == 1 && s.f32 && xb < (1L << 31) - 64 && db < (1L << 31) - 64 && s.sN == (int64_t)s.C * s.sC &&
                     s.sC == (int64_t)s.H * s.sH, "conv_first_wgrad: layout outside the streaming kernel's range");
        const int strips = cdiv(s.W, FTW);
        long segs = 4096 / ((long)N * strips);
        if (segs > s.H / 8) segs = s.H / 8;
        if (segs < 1) segs = 1;
        const int seg_rows = cdiv(s.H, segs);
        const int nseg = cdiv(s.H, seg_rows);
        const long items = (long)N * strips * nseg;
        USTRUN_CHECK(items <= 4096, "conv_first_wgrad: %ld strip segments", items);
        blocks = cdiv(items, 4);
        const int lds = 4 * 2 * 3 * FTW * WRB;
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv_first_wgrad_stream_x3_kernel<4>, lds, "conv_first_wgrad_x3"));
        hipLaunchKernelGGL(conv_first_wgrad_stream_x3_kernel<4>, dim3(blocks), dim3(256), lds, st, (const float*)s.ptr, (int)s.sN, (int)s.sC,
                           (int)s.sH, s.C, s.H, s.W, (int)xb, (const float*)dy, (int)db, partials, strips, nseg, seg_rows, (int)items);
        USTRUN_LAUNCH_CHECK("conv_first_wgrad_x3");
        hipLaunchKernelGGL(conv_first_wgrad_reduce_kernel, dim3(cdiv(64 * s.C * 9, 8)), dim3(1024), 0, st, partials, blocks, s.C, dw,
                           accumulate);
        USTRUN_LAUNCH_CHECK("conv_first_wgrad_reduce");
        return 0;
    }
    const long xbytes = (long)N * s.sN * 4, dybytes = (long)N * s.H * s.W * 64 * 2;
    if (dy_esz == 2 && s.sW == 1 && s.f32 && !(g_debug_flags & (1 << 28)) && xbytes < (1L << 31) - 64 && dybytes < (1L << 31) - 64 &&
        s.sN == (int64_t)s.C * s.sC && s.sC == (int64_t)s.H * s.sH) {
        const int strips = cdiv(s.W, FTW);
        long segs = 4096 / ((long)N * strips);
        if (segs > s.H / 8) segs = s.H / 8;
        if (segs < 1) segs = 1;
        const int seg_rows = cdiv(s.H, segs);
        const int nseg = cdiv(s.H, seg_rows);
        const long items = (long)N * strips * nseg;
        if (items <= 4096) {
            blocks = cdiv(items, 4);
            hipLaunchKernelGGL(conv_first_wgrad_stream_kernel<4>, dim3(blocks), dim3(256), 0, st, (const float*)s.ptr, (int)s.sN, (int)s.sC,
                               (int)s.sH, s.C, s.H, s.W, (int)xbytes, (const elt_t*)dy, (int)dybytes, partials, strips, nseg, seg_rows,
                               (int)items);
            USTRUN_LAUNCH_CHECK("conv_first_wgrad_stream");
            hipLaunchKernelGGL(conv_first_wgrad_reduce_kernel, dim3(cdiv(64 * s.C * 9, 8)), dim3(1024), 0, st, partials, blocks, s.C, dw,
                               accumulate);
            USTRUN_LAUNCH_CHECK("conv_first_wgrad_reduce");
            return 0;
        }
    }
    if (dy_esz == 2)
        hipLaunchKernelGGL(conv_first_wgrad_kernel<2>, dim3(blocks), dim3(256), 0, st, (const float*)s.ptr, (long)s.sN, (long)s.sC,
                           (long)s.sH, (long)s.sW, s.C, s.H, s.W, (const float*)dy, partials, tx, ty, ttotal, per);
    else
        hipLaunchKernelGGL(conv_first_wgrad_kernel<4>, dim3(blocks), dim3(256), 0, st, (const float*)s.ptr, (long)s.sN, (long)s.sC,
                           (long)s.sH, (long)s.sW, s.C, s.H, s.W, (const float*)dy, partials, tx, ty, ttotal, per);
    USTRUN_LAUNCH_CHECK("conv_first_wgrad");
    hipLaunchKernelGGL(conv_first_wgrad_reduce_kernel, dim3(cdiv(64 * s.C * 9, 8)), dim3(1024), 0, st, partials, blocks, s.C, dw,
                       accumulate);
    USTRUN_LAUNCH_CHECK("conv_first_wgrad_reduce");
    return 0;
}

}  // namespace ustrun
