// conv_halo_bf16.hip -- 3x3 convolution (forward and input-gradient) on the bf16 matrix cores with
// LDS-staged input HALO tiles: for every 64-channel (32 when pooling) slice of K the block loads its
// (TH+2)x18-pixel input patch ONCE -- BatchNorm affine + ReLU (+2x2 max-pool, concat, zero padding)
// applied in f32 on the way in, rounded to bf16, K-contiguous rows with the 16-byte chunks
// XOR-swizzled by pixel -- and all nine taps read their shifted A fragments from that one patch.
// Weights stream tap by tap as pure copies: bf16 [tap][K/8][N][8] tiles fetched by LDS-DMA
// (global_load_lds_dwordx4, no VGPRs) into a double buffer, one tap ahead of the MFMAs.
//
// Tile: TH x 16 output pixels x BN channels per 256-thread block; every wave owns 4 rows x 16 pixels
// x 64 channels = 2x2 v_mfma_f32_32x32x16_bf16 accumulators (MI = 2), or 8 rows x 16 pixels x 64 channels
// = 4x2 accumulators (MI = 4: 0.75 LDS fragment reads per MFMA instead of 1, half the weight traffic per flop).
#include "common.h"
#include "loader.h"
#include <stdlib.h>

namespace ustrun {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;


// MI = 32-pixel MFMA sub-tiles per wave along M.  TW = 16: a sub-tile is 2 rows x 16 px (wave tile 2*MI rows x 16 px);
// TW = 32: a sub-tile is one row of 32 consecutive pixels (wave tile MI rows x 32 px) -- 32 consecutive patch rows
// per ds_read_b128 make the swizzled A reads conflict-free (with 16-wide tiles 2 of 16 lanes collide).
template <int TH, int TW, int BN, int BK, int MI, bool POOL>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_bf16_kernel(const IgemmArgs a, const int tiles_x, const int tiles_y,
                                                                   const int nt_total) {
    constexpr int HW2 = TW + 2;
    constexpr int SR = 32 / TW;                 // tile rows per 32-pixel sub-tile (2 or 1)
    constexpr int WM = TH / (SR * MI), WN = 4 / WM;
    static_assert(WM * WN == 4 && BN == WN * 64, "4 waves, 64 channels per wave");
    constexpr int HP = (TH + 2) * HW2;          // halo pixels
    constexpr int CPR = BK / 8;                 // 8-channel (16-byte) groups per pixel
    constexpr int AIT = (HP * CPR + 255) / 256; // A items per thread per chunk
    constexpr int NP = POOL ? 4 : 1;
    constexpr int ROWB = BK * 2;
    constexpr int BCH = (BK / 8) * BN;          // 16-byte chunks per B tile
    constexpr int BIT = BCH / 256;
    static_assert(BCH % 256 == 0, "B tile must be a whole number of wave-instructions per wave");

    constexpr int BATCH = (AIT + 8) / 9;        // A items staged per tap (the next patch is spread over the 9 taps)
    constexpr int ABYTES = (HP * ROWB + 15) & ~15;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                            // 2 x [HP][BK] bf16 (current patch / patch being staged)
    char* Bs = smem + 2 * ABYTES;               // 2 x [BK/8][BN][8] bf16

    const int mt_total = a.N * tiles_y * tiles_x;
    const int ntiles = mt_total * nt_total;
    int bid = blockIdx.x;
    {
        const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int mtile = bid / nt_total, ntile = bid % nt_total;
    const int n0 = ntile * BN;
    const int img = mtile / (tiles_y * tiles_x);
    const int trem = mtile - img * tiles_y * tiles_x;
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    auto swz = [](int hp) { return BK == 64 ? ((hp >> 1) & 7) : ((hp >> 2) & 3); };

    const int nchunk = a.Cin / BK;
    const int K8 = a.Cin / 8;
    const __bf16* Wp = (const __bf16*)a.W;

    // ---- A patch staging, BATCH items per call; an item = 8 channels (16 B of bf16) of one patch pixel:
    // global -> registers (raw bf16) ... -> [affine + ReLU (+ 2x2 max) in f32] -> bf16 -> LDS.  Sources without an
    // affine (dY in the input-gradient, the ConvTranspose output in a concat) are copied through untouched.
    bf16x8 av[BATCH][NP];
    f32x4 asc0, asc1, ash0, ash1;
    int a_relu = 0, a_aff = 0;
    unsigned aok = 0;
    const int c8 = tid % CPR;                       // constant per thread: 256 % CPR == 0
    const float* aptr = nullptr;
    long abase = 0;
    int aLH = 0, aLW = 0, aby = 0, abx = 0;
    long asH = 0, asW = 0;
    auto stage_begin = [&](int c) {                 // per-chunk constants: source, affine, origin
        const int cg = c * BK + 8 * c8;
        const bool second = (a.nsrc == 2 && cg >= a.src[0].C);
        const SrcDev S = pick_src(a.src[0], a.src[1], second);
        const int cl = cg - (second ? a.src[0].C : 0);
        a_aff = S.scale != nullptr;
        asc0 = asc1 = (f32x4){1.f, 1.f, 1.f, 1.f}; ash0 = ash1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (a_aff) {
            asc0 = *(const f32x4*)(S.scale + cl); asc1 = *(const f32x4*)(S.scale + cl + 4);
            ash0 = *(const f32x4*)(S.shift + cl); ash1 = *(const f32x4*)(S.shift + cl + 4);
        }
        a_relu = S.relu;
        aptr = S.ptr; abase = img * S.sN + cl;
        aLH = S.LH; aLW = S.LW; asH = S.sH; asW = S.sW;
        aby = y0 - 1 - S.off_y; abx = x0 - 1 - S.off_x;
    };
    auto ld8 = [&](long idx) { return *(const bf16x8*)((const __bf16*)aptr + idx); };
    auto stage_load = [&](int t) {                  // items t*BATCH .. t*BATCH+BATCH-1
        aok = 0;
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int hp = (tid + 256 * (t * BATCH + b)) / CPR;
            if (t * BATCH + b < AIT && hp < HP) {
                const int hy = hp / HW2, hx = hp - hy * HW2;
                const int ly = aby + hy, lx = abx + hx;
                if (ly >= 0 && ly < aLH && lx >= 0 && lx < aLW) {
                    aok |= 1u << b;
                    if (POOL) {
                        const long p = abase + (long)(2 * ly) * asH + (long)(2 * lx) * asW;
                        av[b][0] = ld8(p);
                        av[b][1 % NP] = ld8(p + asW);
                        av[b][2 % NP] = ld8(p + asH);
                        av[b][3 % NP] = ld8(p + asH + asW);
                    } else {
                        av[b][0] = ld8(abase + (long)ly * asH + (long)lx * asW);
                    }
                }
            }
        }
    };
    auto act8 = [&](bf16x8 r, f32x4& lo, f32x4& hi) {     // bf16 raw -> activated f32
        lo = (f32x4){(float)r[0], (float)r[1], (float)r[2], (float)r[3]} * asc0 + ash0;
        hi = (f32x4){(float)r[4], (float)r[5], (float)r[6], (float)r[7]} * asc1 + ash1;
        if (a_relu) { lo = relu4(lo); hi = relu4(hi); }
    };
    auto stage_write = [&](int t, char* Adst) {
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            const int hp = (tid + 256 * (t * BATCH + b)) / CPR;
            if (t * BATCH + b < AIT && hp < HP) {
                bf16x8 h;
#pragma unroll
                for (int q = 0; q < 8; ++q) h[q] = (__bf16)0.f;
                if ((aok >> b) & 1u) {
                    if (a_aff || POOL) {
                        f32x4 lo, hi;
                        act8(av[b][0], lo, hi);
                        if (POOL) {
#pragma unroll
                            for (int q = 1; q < NP; ++q) {
                                f32x4 l2, h2;
                                act8(av[b][q], l2, h2);
                                lo = max4(lo, l2); hi = max4(hi, h2);
                            }
                        }
                        h[0] = (__bf16)lo[0]; h[1] = (__bf16)lo[1]; h[2] = (__bf16)lo[2]; h[3] = (__bf16)lo[3];
                        h[4] = (__bf16)hi[0]; h[5] = (__bf16)hi[1]; h[6] = (__bf16)hi[2]; h[7] = (__bf16)hi[3];
                    } else {
                        h = av[b][0];
                    }
                }
                *(bf16x8*)(Adst + hp * ROWB + ((c8 ^ swz(hp)) * 16)) = h;
            }
        }
    };
    // ---- B tile of stage s = chunk*9 + tap: LDS-DMA, 16 B per lane, lane-linear destination ----
    auto dma_B = [&](int s, int buf) {
        const int c = s / 9, tap = s - c * 9;
        const __bf16* wb = Wp + (((long)tap * K8 + c * (BK / 8)) * a.Cout + n0) * 8;
        char* dst = Bs + buf * (BCH * 16);
#pragma unroll
        for (int i = 0; i < BIT; ++i) {
            const int idx = tid + 256 * i, o = idx / BN, n = idx % BN;
            const __bf16* src = wb + ((long)o * a.Cout + n) * 8;
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(dst + (wave * 64 + 256 * i) * 16), 16, 0, 0);
        }
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // halo pixel (before the tap shift) of this lane's A rows: subtile i covers tile rows
    // wm*2*MI + 2i + (l31>>4), column l31&15
    const int hpb0 = (wm * SR * MI + (TW == 16 ? (l31 >> 4) : 0) + 1) * HW2 + (l31 & (TW - 1)) + 1;
    const int nstage = nchunk * 9;

    dma_B(0, 0);
    stage_begin(0);
    for (int t = 0; t * BATCH < AIT; ++t) { stage_load(t); stage_write(t, As); }
    __syncthreads();
    int buf = 0;
    for (int c = 0; c < nchunk; ++c) {
        const char* Acur = As + (c & 1) * ABYTES;
        char* Anext = As + ((c + 1) & 1) * ABYTES;
        const bool more = c + 1 < nchunk;
        if (more) stage_begin(c + 1);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int s = c * 9 + tap;
            if (s + 1 < nstage) dma_B(s + 1, buf ^ 1);
            const bool stg = more && tap * BATCH < AIT;
            if (stg) stage_load(tap);                      // raw loads of the next patch fly under this tap's MFMAs
            const int dy = a.d0 + (tap / 3) * a.dstep, dx = a.d0 + (tap % 3) * a.dstep;
            const int hp0 = hpb0 + dy * HW2 + dx;
            const char* Bp = Bs + buf * (BCH * 16) + (lh * BN + wn * 64 + l31) * 16;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int ch = 2 * ks + lh;
                const bf16x8 b0 = *(const bf16x8*)(Bp + (2 * ks * BN) * 16);
                const bf16x8 b1 = *(const bf16x8*)(Bp + (2 * ks * BN + 32) * 16);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int hp = hp0 + SR * i * HW2;
                    const bf16x8 af = *(const bf16x8*)(Acur + hp * ROWB + ((ch ^ swz(hp)) * 16));
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b0, acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, b1, acc[i][1], 0, 0, 0);
                }
            }
            if (stg) stage_write(tap, Anext);              // the other patch buffer: no reader until the next chunk
            __syncthreads();                               // B[buf] is free; DMA of B[buf^1] has landed
            buf ^= 1;
        }
    }

    // ---- epilogue: bf16 outputs (NHWC), optional two-destination split, BN-statistics partials.
    // The accumulator layout has lanes along channels and registers along pixels, so a direct store would be
    // 2 bytes per lane (64 store instructions per 32x64 sub-tile).  Each wave instead transposes one 32-pixel
    // sub-tile at a time through its own LDS scratch (rows padded to 144 B: the two half-waves hit disjoint
    // banks) and writes 16 bytes per lane, 128 contiguous bytes per pixel.
    const int C1 = a.Cout - a.C0;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    constexpr int EPITCH = 144;
    char* ep = smem + wave * (32 * EPITCH);         // the A patches are dead after the last barrier
    const float bias0 = a.bias ? a.bias[n0 + wn * 64 + l31] : 0.f, bias1 = a.bias ? a.bias[n0 + wn * 64 + 32 + l31] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int oyb = y0 + wm * SR * MI + SR * i;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;          // 0..31 inside the sub-tile
                const int oy = oyb + (TW == 16 ? (row >> 4) : 0), ox = x0 + (row & (TW - 1));
                const __bf16 hv = (__bf16)(acc[i][j][r] + (j ? bias1 : bias0));
                *(__bf16*)(ep + row * EPITCH + (j * 32 + l31) * 2) = hv;
                if (oy < a.Ho && ox < a.Wo) {
                    const float v = (float)hv;                             // statistics see the stored value
                    s1[j] += v; s2[j] += v * v;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = lane + 64 * t, row = idx >> 3, ch = idx & 7;
            const bf16x8 v8 = *(const bf16x8*)(ep + row * EPITCH + ch * 16);
            const int oy = oyb + (TW == 16 ? (row >> 4) : 0), ox = x0 + (row & (TW - 1));
            const int col = n0 + wn * 64 + ch * 8;
            if (oy < a.Ho && ox < a.Wo) {
                if (col < a.C0) {
                    *(bf16x8*)((__bf16*)a.out0 + (((long)img * a.Ho + oy) * a.Wo + ox) * a.C0 + col) = v8;
                } else {
                    const int y1 = oy - a.o1y, x1 = ox - a.o1x;
                    if (y1 >= 0 && y1 < a.H1 && x1 >= 0 && x1 < a.W1)
                        *(bf16x8*)((__bf16*)a.out1 + (((long)img * a.H1 + y1) * a.W1 + x1) * C1 + (col - a.C0)) = v8;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (a.stat) {
        float* red = (float*)(smem + 4 * 32 * EPITCH);   // [WM][2][BN], behind the waves' transpose scratch
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
        constexpr int SROWS = TH / 8;               // one stat row per 8 tile rows (128 pixels)
        constexpr int WPS = WM / SROWS;             // waves (along M) that share a stat row
        for (int t = tid; t < SROWS * 2 * BN; t += 256) {
            const int h = t / (2 * BN), q = (t / BN) % 2, cc = t % BN;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WPS; ++w) v += red[((h * WPS + w) * 2 + q) * BN + cc];
            a.stat[((long)(mtile * SROWS + h) * 2 + q) * a.Cout + n0 + cc] = v;
        }
    }
}

template <int TH, int TW, int BN, int BK, int MI, bool POOL>
int launch_cfg(const IgemmArgs& a, hipStream_t st) {
    const int tx = cdiv(a.Wb, TW), ty = cdiv(a.Hb, TH), nt = a.Cout / BN;
    const size_t lds = 2 * (size_t)(((TH + 2) * (TW + 2) * BK * 2 + 15) & ~15) + 2 * (size_t)(BK / 8) * BN * 16;
    dim3 grid(a.N * ty * tx * nt), block(256);
    hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<TH, TW, BN, BK, MI, POOL>), grid, block, lds, st, a, tx, ty, nt);
    USTRUN_LAUNCH_CHECK("conv3x3_halo_bf16");
    return 0;
}

}  // namespace

// stat rows the halo kernel writes for an N x H x W output (one per 8 x 16 pixels)
int halo_stat_rows(int N, int H, int W) { return N * cdiv(H, 8) * cdiv(W, 16); }

// can this conv3x3-shaped problem run on the halo kernel?
bool halo_supported(const IgemmArgs& a) {
    if (a.nseg != 9 || a.nz != 1 || a.s_in != 1 || a.s_out != 1 || a.segw != 3) return false;
    bool pool = false;
    if (a.out_esz != 2 || (a.C0 & 7) || ((a.Cout - a.C0) & 7)) return false;
    for (int i = 0; i < a.nsrc; ++i) {
        if (a.src[i].sC != 1 || (a.src[i].C & 7) || a.src[i].esz != 2) return false;
        pool |= a.src[i].pool != 0;
    }
    const int BK = 32;
    if (a.Cin % BK || a.Cout % 64) return false;
    if (a.nsrc == 2 && (a.src[0].C % BK)) return false;
    if (pool && (a.nsrc != 1 || a.Cout % 128)) return false;
    if (a.Hb < 4 || a.Wb < 8) return false;      // tiny extents: the generic kernel wastes less
    return true;
}

// 16-row tiles (256 px x 128 ch per block) read fewer LDS/weight bytes per flop but need >= 2 blocks per CU
bool halo_tall_tile(const IgemmArgs& a) {
    return (long)a.N * cdiv(a.Hb, 16) * cdiv(a.Wb, 16) * (a.Cout / 128) >= 512;
}

// BatchNorm-statistics rows written by the configuration conv3x3_halo_launch_bf16 picks
int halo_stat_rows_used(const IgemmArgs& a) {
    bool pool = false;
    for (int i = 0; i < a.nsrc; ++i) pool |= a.src[i].pool != 0;
    const bool wide = a.Wb >= 32;
    if (pool) return a.N * cdiv(a.Hb, 8) * cdiv(a.Wb, 16);
    if (a.Cout % 128 == 0) {
        if (halo_tall_tile(a)) return wide ? a.N * cdiv(a.Hb, 8) * cdiv(a.Wb, 32) : a.N * cdiv(a.Hb, 16) * 2 * cdiv(a.Wb, 16);
        return a.N * cdiv(a.Hb, 8) * cdiv(a.Wb, 16);
    }
    return wide ? a.N * cdiv(a.Hb, 8) * cdiv(a.Wb, 32) : a.N * cdiv(a.Hb, 16) * 2 * cdiv(a.Wb, 16);
}

int conv3x3_halo_launch_bf16(const IgemmArgs& a, hipStream_t st) {
    bool pool = false;
    for (int i = 0; i < a.nsrc; ++i) pool |= a.src[i].pool != 0;
    const bool wide = a.Wb >= 32;                 // 32-pixel rows: conflict-free A fragment reads
    if (pool) return launch_cfg<8, 16, 128, 32, 2, true>(a, st);
    if (a.Cout % 128 == 0) {
        if (halo_tall_tile(a))                    // 256 px x 128 ch per block, wave tile 128 px x 64 ch
            return wide ? launch_cfg<8, 32, 128, 32, 4, false>(a, st) : launch_cfg<16, 16, 128, 32, 4, false>(a, st);
        if (a.Cin % 64 == 0) return launch_cfg<8, 16, 128, 64, 2, false>(a, st);
        return launch_cfg<8, 16, 128, 32, 2, false>(a, st);
    }
    return wide ? launch_cfg<8, 32, 64, 32, 2, false>(a, st) : launch_cfg<16, 16, 64, 32, 2, false>(a, st);
}

}  // namespace ustrun
