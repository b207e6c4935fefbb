// wgrad_tap_bf16.hip -- weight gradient of a 1x1 / dilated / strided k x k convolution on the bf16 matrix cores, one TAP per
// block (the DeepLabV2-ResNet bottlenecks and classifier: reference networks/backbone/resnet.py:78-105, deeplabv2.py:15-17):
//
//   dW[tap][ci][co] = sum_p act(x[stride * p + dilation * (tap - k/2)][ci]) * dy[p][co]          p over the output grid
//
// A "TN" GEMM over pixels whose operands are both pixel-major in HBM.  A block owns 128 (or 64) ci x 128 (or 64) co of one tap and a
// contiguous range of output pixels (split-K over space), 64 pixels per stage, double-buffered; the tiles stay
// [pixel][channel] in LDS (256-byte rows, 64-byte segments XORed by the pixel index) and the MFMA fragments come from the
// transposing LDS read (ds_read_b64_tr_b16), as in wgradT_bf16.hip.  The activation tile goes global -> registers ->
// [BatchNorm affine + ReLU in f32, zero outside the image] -> bf16 -> LDS, the dy tile global -> registers -> LDS: the loads
// of stage s+1 are issued before the MFMAs of stage s and written to the other buffer after them.  The taps of one pixel range run next to each other on one XCD (block order below),
// so the nine shifted reads of x and the nine reads of dy meet in that XCD's L2.  Partial slabs [ksplit][tap][ci][co] f32 are
// summed in a fixed order by reduce_partials (1x1: slabs already in the torch layout, see SWAP).
#include "common.h"
#include "loader.h"

namespace ustrun {
namespace {

typedef __attribute__((ext_vector_type(8))) elt_t bf16x8;
typedef __attribute__((ext_vector_type(4))) elt_t bf16x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int KP = 64;   // pixels per stage; tiles are TM ci x TN co (128 or 64 each), LDS rows TM*2 / TN*2 bytes

// 64-byte segments of a row are XOR-permuted by the row index so that the 4 rows of a transposed read fall on different bank
// segments (256-byte rows: 4 segments, row & 3; 128-byte rows: 2 segments, (row >> 1) & 1)
template <int RB> __device__ __forceinline__ int seg_swz(int row) { return RB >= 256 ? (row & 3) : ((row >> 1) & 1); }

// rows k0 + 8*(l>>5) + {0..3 | 4..7}, columns col0 + 16*((l>>4)&1) + 4*(l&3) .. +3, delivered column-major
template <int RB> __device__ __forceinline__ bf16x8 tr_frag(const char* tile, int k0, int col0, int lane) {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int colb = (col0 + 16 * ((lane >> 4) & 1) + 4 * p) * 2;
    const int r0 = k0 + 8 * (lane >> 5) + q, r1 = r0 + 4;
    const bf16x4 lo = USTRUN_DS_READ_TR16((lds_bf16x4*)(tile + r0 * RB + (colb ^ (seg_swz<RB>(r0) << 6))));
    const bf16x4 hi = USTRUN_DS_READ_TR16((lds_bf16x4*)(tile + r1 * RB + (colb ^ (seg_swz<RB>(r1) << 6))));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// v / d and the remainder for 0 <= v < 2^24 through the float reciprocal (two fix-up steps make it exact)
__device__ __forceinline__ int fdiv(int v, int d, float invd, int& rem) {
    int q = (int)(((float)v + 0.5f) * invd);
    int r = v - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
    rem = r;
    return q;
}

// grid = (ci tiles * co tiles * taps * ksplit)
// SWAP (1x1 convolutions): the MFMA operands trade places, D rows are co and its lanes ci, so the slab comes out as [co][ci] --
// the torch layout of a 1x1 weight -- and the fixed-order streaming sum finishes it without a transposing pass.
// AFF (round 6): the source carries BatchNorm constants / a ReLU.  A finished activation (AFF = false) skips the f32 round trip of
// every staged item -- ~40 VALU instructions per 16-byte item, which each of the Cout / TN column tiles (and each of the nine taps)
// repeated on the same values and which, not the MFMAs, paced the stage (8 VALU per MFMA in the SQ counters).
template <int TM, int TN, bool PLAIN, bool SWAP, bool AFF>
__global__ __launch_bounds__(256, 2) void wgrad_tap_bf16_kernel(const WgradArgs a, const int mtn, const int ntn) {
    constexpr int RBA = TM * 2, RBB = TN * 2;                 // LDS row pitches (bytes)
    constexpr int AQ = TM / 8, AROWS = 256 / AQ, AP = KP / AROWS;      // 16-byte items per row, rows per pass, passes
    constexpr int BQ = TN / 8, BROWS = 256 / BQ, BP = KP / BROWS;
    constexpr int MI = TM / 64, NI = TN / 64;                 // 32 x 32 MFMA tiles per wave (waves 2 x 2)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                      // 2 x [KP][TM] bf16
    char* Bs = smem + 2 * KP * RBA;       // 2 x [KP][TN] bf16

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware order (1-D grid, workgroups go round-robin over the 8 XCDs): every XCD takes a contiguous range of the
    // (slice-major, tile-minor) order, so all (tap, ci, co) tiles of one pixel slice share one XCD's L2
    const int nblk = gridDim.x, tiles = nblk / a.ksplit;
    int lin;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = blockIdx.x % 8, jj = blockIdx.x / 8;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + jj;
    }
    const int ks = lin / tiles;
    int tile = lin - ks * tiles;
    const int ntile = tile % ntn; tile /= ntn;
    const int mtile = tile % mtn;
    const int seg = tile / mtn;
    const int ci0 = mtile * TM, co0 = ntile * TN;
    const int ady = a.d0 + (seg / a.segw) * a.astep, adx = a.d0 + (seg % a.segw) * a.astep;
    const int kbeg = (int)((long)ks * a.kchunk);
    const int kend = (kbeg + a.kchunk < a.M) ? (int)(kbeg + a.kchunk) : (int)a.M;
    const int Wb = a.Wb, Hb = a.Hb;
    const float invW = 1.f / (float)Wb, invH = 1.f / (float)Hb;

    // ---- A items: pixel row tid / AQ + AROWS i, 8-channel group tid % AQ ----
    const SrcDev& S = a.src[0];
    const int c8 = tid % AQ;
    const int cl = ci0 + 8 * c8;
    const bool aff = AFF && S.scale != nullptr;
    f32x4 asc0 = {1.f, 1.f, 1.f, 1.f}, asc1 = asc0, ash0 = {0.f, 0.f, 0.f, 0.f}, ash1 = ash0;
    if (aff) {
        asc0 = *(const f32x4*)(S.scale + cl); asc1 = *(const f32x4*)(S.scale + cl + 4);
        ash0 = *(const f32x4*)(S.shift + cl); ash1 = *(const f32x4*)(S.shift + cl + 4);
    }
    const float floor_ = S.relu ? 0.f : -__builtin_inff();
    const elt_t* sp = (const elt_t*)S.ptr + cl;
    const int arow = tid / AQ;
    const int sN = (int)S.sN, sH = (int)S.sH, sW = (int)S.sW;        // element offsets fit 31 bits (host check)
    bf16x8 av[AP];
    unsigned aok = 0;
    // every load is issued unconditionally from an in-range address (pixel 0 for rows outside the range / the image) and the
    // row is zeroed at the LDS write: a load under a per-row branch gets its own basic block and its own wait
    // pixel coordinates of this thread's four rows, carried from stage to stage (one float-reciprocal decomposition at the start)
    int px[AP], py[AP], pn[AP];
    const bool wide = Wb >= KP;
    if (!PLAIN) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int r = fdiv(kbeg + arow + AROWS * i, Wb, invW, px[i]);
            pn[i] = fdiv(r, Hb, invH, py[i]);
        }
    }
    auto load_A = [&](int k0) {
        aok = 0;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int m = k0 + arow + AROWS * i;
            int ok = m < kend;
            int off = m * sW;                                // PLAIN: 1x1, stride 1, pixel-linear source
            if (!PLAIN) {
                const int ly = (py[i] << a.ashift) + ady - S.off_y, lx = (px[i] << a.ashift) + adx - S.off_x;
                ok &= (int)((unsigned)ly < (unsigned)S.LH) & (int)((unsigned)lx < (unsigned)S.LW);       // bitwise: no exec-mask region
                off = pn[i] * sN + ly * sH + lx * sW;
                // the next stage's pixel, KP further along the row-major order: one branch-free carry when a row holds at least
                // KP pixels, the float-reciprocal decomposition otherwise (wave-uniform choice)
                if (wide) {
                    px[i] += KP;
                    const int cx = px[i] >= Wb;
                    px[i] -= cx ? Wb : 0; py[i] += cx;
                    const int cy = py[i] >= Hb;
                    py[i] -= cy ? Hb : 0; pn[i] += cy;
                } else {
                    const int r = fdiv(m + KP, Wb, invW, px[i]);
                    pn[i] = fdiv(r, Hb, invH, py[i]);
                }
            }
            av[i] = *(const bf16x8*)(sp + (ok ? off : 0));
            aok |= (unsigned)ok << i;
        }
    };
    auto write_A = [&](char* dst) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int row = arow + AROWS * i;
            const bool ok = (aok >> i) & 1u;
            bf16x8 h = av[i];
            if constexpr (AFF) {
                f32x4 lo = (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]} * asc0 + ash0;
                f32x4 hi = (f32x4){(float)h[4], (float)h[5], (float)h[6], (float)h[7]} * asc1 + ash1;
#pragma unroll
                for (int q = 0; q < 4; ++q) { lo[q] = __builtin_fmaxf(lo[q], floor_); hi[q] = __builtin_fmaxf(hi[q], floor_); }
                h[0] = (elt_t)lo[0]; h[1] = (elt_t)lo[1]; h[2] = (elt_t)lo[2]; h[3] = (elt_t)lo[3];
                h[4] = (elt_t)hi[0]; h[5] = (elt_t)hi[1]; h[6] = (elt_t)hi[2]; h[7] = (elt_t)hi[3];
            }
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            u32x4 bits = __builtin_bit_cast(u32x4, h);
#pragma unroll
            for (int q = 0; q < 4; ++q) bits[q] = ok ? bits[q] : 0u;       // rows outside the range / the image are zero
            *(u32x4*)(dst + row * RBA + ((c8 * 16) ^ (seg_swz<RBA>(row) << 6))) = bits;
        }
    };
    // ---- dy tile: the same (row, 8-channel group) items, a pure copy.  It goes through registers like the activation tile
    // and NOT by LDS-DMA: with a DMA in flight hipcc puts s_waitcnt vmcnt(0) in front of the stage's first ds_read (it cannot
    // tell the two LDS buffers apart), which also drains the activation prefetch before the MFMAs instead of after them.
    const int b8 = tid % BQ, brow = tid / BQ;
    const elt_t* dyp = (const elt_t*)a.dy + co0 + 8 * b8;
    const int dyC = a.Cout;
    bf16x8 bv[BP];
    auto load_B = [&](int k0) {
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int m = k0 + brow + BROWS * i;
            // rows past the tensor read pixel 0: any finite value will do, their A rows are zero
            bv[i] = *(const bf16x8*)(dyp + (m < (int)a.M ? m : 0) * dyC);
        }
    };
    auto write_B = [&](char* dst) {
#pragma unroll
        for (int i = 0; i < BP; ++i) {
            const int row = brow + BROWS * i;
            *(bf16x8*)(dst + row * RBB + ((b8 * 16) ^ (seg_swz<RBB>(row) << 6))) = bv[i];
        }
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kbeg < kend) {
        load_A(kbeg);
        load_B(kbeg);
        write_A(As);
        write_B(Bs);
    }
    __syncthreads();
    int buf = 0;
#pragma unroll 1
    for (int k0 = kbeg; k0 < kend; k0 += KP) {
        const bool more = k0 + KP < kend;
        if (more) {
            load_A(k0 + KP);
            load_B(k0 + KP);
        }
        const char* At = As + buf * (KP * RBA);
        const char* Bt = Bs + buf * (KP * RBB);
#pragma unroll
        for (int kk = 0; kk < KP / 16; ++kk) {
            bf16x8 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = tr_frag<RBA>(At, kk * 16, wm * (TM / 2) + 32 * i, lane);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[j] = tr_frag<RBB>(Bt, kk * 16, wn * (TN / 2) + 32 * j, lane);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = SWAP ? USTRUN_MFMA_32x32x16(bf[j], af[i], acc[i][j], 0, 0, 0)
                                     : USTRUN_MFMA_32x32x16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) {
            write_A(As + (buf ^ 1) * (KP * RBA));
            write_B(Bs + (buf ^ 1) * (KP * RBB));
        }
        __syncthreads();
        buf ^= 1;
    }

    float* slab = a.partials + ((long)ks * a.nseg + seg) * a.Cin * a.Cout;
    const int l31 = lane & 31, lh = lane >> 5;
    if (SWAP) {
        // slab [ks][Cout][Cin]: rows of D are co (registers), the 32 lanes of a row are consecutive ci
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ci = ci0 + wm * (TM / 2) + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wn * (TN / 2) + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    slab[(long)co * a.Cin + ci] = acc[i][j][r];
                }
        }
    } else {
        // slab [ks][tap][Cin][Cout]: rows of D are ci (registers), the 32 lanes of a row are consecutive co
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int co = co0 + wn * (TN / 2) + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = ci0 + wm * (TM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    slab[(long)ci * a.Cout + co] = acc[i][j][r];
                }
        }
    }
}

bool pixel_linear(const WgradArgs& a) {
    const SrcDev& s = a.src[0];
    return a.nseg == 1 && a.ashift == 0 && a.d0 == 0 && s.off_y == 0 && s.off_x == 0 && s.LH == a.Hb && s.LW == a.Wb &&
           s.sH == (long)a.Wb * s.sW && s.sN == (long)a.Hb * s.sH;
}

}  // namespace

static int tile_m(const WgradArgs& a) { return a.Cin % 128 == 0 ? 128 : 64; }
static int tile_n(const WgradArgs& a) { return a.Cout % 128 == 0 ? 128 : 64; }

bool wgrad_tap_supported(const WgradArgs& a) {
    if (a.dy_s != 1 || a.dy_esz != 2 || a.nsrc != 1 || a.nseg < 1 || a.nseg > 9 || a.ashift > 1) return false;
    const SrcDev& s = a.src[0];
    if (s.esz != 2 || s.sC != 1 || s.pool || s.gN > 0 || (s.relu && !s.scale)) return false;
    if ((s.sW & 7) || (s.sH & 7) || (s.sN & 7)) return false;                 // 16-byte loads of 8 channels
    if (a.dyH != a.Hb || a.dyW != a.Wb || a.M >= (1L << 24)) return false;    // float-reciprocal pixel decomposition
    if ((long)a.N * s.sN >= (1L << 31) || a.M * a.Cout >= (1L << 31)) return false;      // 32-bit element offsets
    return a.Cin % 64 == 0 && a.Cout % 64 == 0;
}

// split-K plan: at most one resident round of blocks (2 per CU), at least four 64-pixel stages per block
int wgrad_tap_plan(const WgradArgs& a, int* ksplit, long* kchunk) {
    const long tiles = (long)(a.Cin / tile_m(a)) * (a.Cout / tile_n(a)) * a.nseg;
    long ks = 512 / tiles;                 // rounded DOWN: 36 tiles x 15 slices = 540 blocks ran as 512 + a second round of 28
                                           // (0.21 ms for a 0.11 ms job); 14 slices = 504 blocks finish in one round
    if (ks > a.M / (4 * KP)) ks = a.M / (4 * KP);
    if (ks < 1) ks = 1;
    long chunk = (a.M + ks - 1) / ks;
    chunk = (chunk + KP - 1) / KP * KP;
    *kchunk = chunk; *ksplit = (int)((a.M + chunk - 1) / chunk);
    return 0;
}

template <int TM, int TN, bool AFF>
static int launch_tile_aff(const WgradArgs& a, hipStream_t st) {
    const int mtn = a.Cin / TM, ntn = a.Cout / TN;
    dim3 grid(mtn * ntn * a.nseg * a.ksplit), block(256);
    constexpr int lds = 2 * KP * (TM + TN) * 2;
    if (pixel_linear(a)) hipLaunchKernelGGL((wgrad_tap_bf16_kernel<TM, TN, true, true, AFF>), grid, block, lds, st, a, mtn, ntn);
    else if (a.nseg == 1) hipLaunchKernelGGL((wgrad_tap_bf16_kernel<TM, TN, false, true, AFF>), grid, block, lds, st, a, mtn, ntn);
    else hipLaunchKernelGGL((wgrad_tap_bf16_kernel<TM, TN, false, false, AFF>), grid, block, lds, st, a, mtn, ntn);
    USTRUN_LAUNCH_CHECK("wgrad_tap_bf16");
    return 0;
}
template <int TM, int TN>
static int launch_tile(const WgradArgs& a, hipStream_t st) {
    const bool aff = a.src[0].scale != nullptr || a.src[0].relu;
    return aff ? launch_tile_aff<TM, TN, true>(a, st) : launch_tile_aff<TM, TN, false>(a, st);
}

thread_local int g_last_wgrad_variant = 0;     // (per calling thread)
int wgrad_last_variant() { return g_last_wgrad_variant; }
void set_last_wgrad_variant(int v) { g_last_wgrad_variant = v; }

int wgrad_tap_launch_bf16(const WgradArgs& a, hipStream_t st) {
    const int tm = tile_m(a), tn = tile_n(a);
    // 'T' | TM/64 | TN/64 | loader (2 = pixel-linear 1x1, 1 = one tap, 0 = taps) | ksplit   (tests: ustrun_debug_last_wgrad_variant)
    g_last_wgrad_variant = 0x54000000 | (tm / 64) << 20 | (tn / 64) << 16 | (pixel_linear(a) ? 2 : (a.nseg == 1 ? 1 : 0)) << 12 | (a.ksplit & 0xfff);       // (ksplit < 2^11 by the plan)
    if (tm == 128 && tn == 128) return launch_tile<128, 128>(a, st);
    if (tm == 128) return launch_tile<128, 64>(a, st);
    if (tn == 128) return launch_tile<64, 128>(a, st);
    return launch_tile<64, 64>(a, st);
}

}  // namespace ustrun
