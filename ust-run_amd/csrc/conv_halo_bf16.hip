// conv_halo_bf16.hip -- 3x3 convolution (forward and input-gradient) on the bf16 matrix cores with
// LDS-staged input HALO tiles: for every 64-channel (32 when pooling) slice of K the block loads its
// (TH+2)x18-pixel input patch ONCE -- BatchNorm affine + ReLU (+2x2 max-pool, concat, zero padding)
// applied in f32 on the way in, rounded to bf16, K-contiguous rows with the 16-byte chunks
// XOR-swizzled by pixel -- and all nine taps read their shifted A fragments from that one patch.
// Weights stream tap by tap as pure copies: bf16 [tap][K/8][N][8] tiles fetched by LDS-DMA
// (global_load_lds_dwordx4, no VGPRs) into a double buffer, one tap ahead of the MFMAs.
//
// Tile: TH x 16 output pixels x BN channels per 256-thread block; every wave owns 4 rows x 16 pixels
// x 64 channels = 2x2 v_mfma_f32_32x32x16_bf16 accumulators.  (TH,BN) = (8,128) or (16,64).
#include "common.h"
#include "loader.h"

namespace ustrun {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int TW = 16, HW2 = TW + 2;

template <int TH, int BN, int BK, bool POOL>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_bf16_kernel(const IgemmArgs a, const int tiles_x, const int tiles_y,
                                                                   const int nt_total) {
    constexpr bool EARLY_A = !POOL;             // prefetch the next A patch into registers under tap 8's MFMAs
    constexpr int WM = TH / 4, WN = 4 / WM;
    constexpr int HP = (TH + 2) * HW2;          // halo pixels
    constexpr int CPR = BK / 4;                 // 4-channel groups per pixel
    constexpr int AIT = (HP * CPR + 255) / 256; // A items per thread per chunk
    constexpr int NP = POOL ? 4 : 1;
    constexpr int ROWB = BK * 2;
    constexpr int BCH = (BK / 8) * BN;          // 16-byte chunks per B tile
    constexpr int BIT = BCH / 256;
    static_assert(BCH % 256 == 0, "B tile must be a whole number of wave-instructions per wave");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                            // [HP][BK] bf16
    char* Bs = smem + ((HP * ROWB + 15) & ~15); // 2 x [BK/8][BN][8] bf16

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

    f32x4 av[AIT][NP];
    f32x4 asc, ash;
    unsigned aok;
    int a_relu;

    // ---- A halo: global -> registers (raw), registers -> transform -> bf16 -> LDS ----
    auto load_A = [&](int c) {
        const int c0 = c * BK;
        const int c4 = tid % CPR;                    // constant per thread: 256 % CPR == 0
        const int cg = c0 + 4 * c4;
        const bool second = (a.nsrc == 2 && cg >= a.src[0].C);
        const SrcDev S = pick_src(a.src[0], a.src[1], second);
        const int cl = cg - (second ? a.src[0].C : 0);
        asc = (f32x4){1.f, 1.f, 1.f, 1.f}; ash = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (S.scale) { asc = *(const f32x4*)(S.scale + cl); ash = *(const f32x4*)(S.shift + cl); }
        a_relu = S.relu;
        aok = 0;
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int hp = (tid + 256 * i) / CPR;
            if (hp < HP) {
                const int hy = hp / HW2, hx = hp - hy * HW2;
                const int ly = y0 + hy - 1 - S.off_y, lx = x0 + hx - 1 - S.off_x;
                if (ly >= 0 && ly < S.LH && lx >= 0 && lx < S.LW) {
                    aok |= 1u << i;
                    if (POOL) {
                        const float* p = S.ptr + img * S.sN + (long)(2 * ly) * S.sH + (long)(2 * lx) * S.sW + cl;
                        av[i][0] = *(const f32x4*)p;
                        av[i][1 % NP] = *(const f32x4*)(p + S.sW);
                        av[i][2 % NP] = *(const f32x4*)(p + S.sH);
                        av[i][3 % NP] = *(const f32x4*)(p + S.sH + S.sW);
                    } else {
                        av[i][0] = *(const f32x4*)(S.ptr + img * S.sN + (long)ly * S.sH + (long)lx * S.sW + cl);
                    }
                }
            }
        }
    };
    auto write_A = [&]() {
        const int c4 = tid % CPR;
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int hp = (tid + 256 * i) / CPR;
            if (hp < HP) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((aok >> i) & 1u) {
                    v = av[i][0] * asc + ash;
                    if (a_relu) v = relu4(v);
                    if (POOL) {
#pragma unroll
                        for (int q = 1; q < NP; ++q) {
                            f32x4 t = av[i][q] * asc + ash;
                            if (a_relu) t = relu4(t);
                            v = max4(v, t);
                        }
                    }
                }
                bf16x4 h;
                h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
                *(bf16x4*)(As + hp * ROWB + (((c4 >> 1) ^ swz(hp)) * 16) + (c4 & 1) * 8) = h;
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

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // halo pixel (before the tap shift) of this lane's two A rows: subtile i covers tile rows
    // wm*4 + 2i + (l31>>4), column l31&15
    const int hpb0 = (wm * 4 + (l31 >> 4) + 1) * HW2 + (l31 & 15) + 1;
    const int hpb1 = hpb0 + 2 * HW2;
    const int nstage = nchunk * 9;

    load_A(0);
    dma_B(0, 0);
    write_A();
    __syncthreads();
    int buf = 0;
    for (int c = 0; c < nchunk; ++c) {
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int s = c * 9 + tap;
            if (s + 1 < nstage) dma_B(s + 1, buf ^ 1);
            if (EARLY_A && tap == 8 && c + 1 < nchunk) load_A(c + 1);
            const int dy = a.d0 + (tap / 3) * a.dstep, dx = a.d0 + (tap % 3) * a.dstep;
            const int hp0 = hpb0 + dy * HW2 + dx, hp1 = hpb1 + dy * HW2 + dx;
            const char* Ap0 = As + hp0 * ROWB;
            const char* Ap1 = As + hp1 * ROWB;
            const int sw0 = swz(hp0), sw1 = swz(hp1);
            const char* Bp = Bs + buf * (BCH * 16) + (lh * BN + wn * 64 + l31) * 16;
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                const int ch = 2 * ks + lh;
                const bf16x8 a0 = *(const bf16x8*)(Ap0 + ((ch ^ sw0) * 16));
                const bf16x8 a1 = *(const bf16x8*)(Ap1 + ((ch ^ sw1) * 16));
                const bf16x8 b0 = *(const bf16x8*)(Bp + (2 * ks * BN) * 16);
                const bf16x8 b1 = *(const bf16x8*)(Bp + (2 * ks * BN + 32) * 16);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            }
            __syncthreads();             // B[buf] and (at tap 8) the A patch are free; DMA of B[buf^1] has landed
            if (tap == 8 && c + 1 < nchunk) {
                if (!EARLY_A) load_A(c + 1);
                write_A();
                __syncthreads();
            }
            buf ^= 1;
        }
    }

    // ---- epilogue: f32 outputs (NHWC), optional two-destination split, BN-statistics partials ----
    const int C1 = a.Cout - a.C0;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + l31;
        const float bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;          // 0..31 inside the 2x16 subtile
                const int oy = y0 + wm * 4 + 2 * i + (row >> 4), ox = x0 + (row & 15);
                if (oy < a.Ho && ox < a.Wo) {
                    const float v = acc[i][j][r] + bias;
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
    if (a.stat) {
        float* red = (float*)As;   // [WM][2][BN]; the A patch is dead after the last barrier
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
        constexpr int HALVES = WM / 2;              // one stat row per 8 tile rows (128 pixels)
        for (int t = tid; t < HALVES * 2 * BN; t += 256) {
            const int h = t / (2 * BN), q = (t / BN) % 2, cc = t % BN;
            const float v = red[((2 * h) * 2 + q) * BN + cc] + red[((2 * h + 1) * 2 + q) * BN + cc];
            a.stat[((long)(mtile * HALVES + h) * 2 + q) * a.Cout + n0 + cc] = v;
        }
    }
}

template <int TH, int BN, int BK, bool POOL>
int launch_cfg(const IgemmArgs& a, hipStream_t st) {
    const int tx = cdiv(a.Wb, TW), ty = cdiv(a.Hb, TH), nt = a.Cout / BN;
    const size_t lds = (size_t)(((TH + 2) * HW2 * BK * 2 + 15) & ~15) + 2 * (size_t)(BK / 8) * BN * 16;
    dim3 grid(a.N * ty * tx * nt), block(256);
    hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<TH, BN, BK, POOL>), grid, block, lds, st, a, tx, ty, nt);
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
    for (int i = 0; i < a.nsrc; ++i) {
        if (a.src[i].sC != 1 || (a.src[i].C & 3)) return false;
        pool |= a.src[i].pool != 0;
    }
    const int BK = (pool || a.Cout % 128) ? 32 : 64;
    if (a.Cin % BK || a.Cout % 64) return false;
    if (a.nsrc == 2 && (a.src[0].C % BK)) return false;
    if (pool && (a.nsrc != 1 || a.Cout % 128)) return false;
    if (a.Hb < 4 || a.Wb < 8) return false;      // tiny extents: the generic kernel wastes less
    return true;
}

int conv3x3_halo_launch_bf16(const IgemmArgs& a, hipStream_t st) {
    bool pool = false;
    for (int i = 0; i < a.nsrc; ++i) pool |= a.src[i].pool != 0;
    if (pool) return launch_cfg<8, 128, 32, true>(a, st);
    if (a.Cout % 128 == 0) return launch_cfg<8, 128, 64, false>(a, st);
    return launch_cfg<16, 64, 32, false>(a, st);
}

}  // namespace ustrun
