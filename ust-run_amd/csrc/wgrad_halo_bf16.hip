// wgrad_halo_bf16.hip -- 3x3 weight gradient on the bf16 matrix cores with all nine taps per block:
//
//   dW[tap][ci][co] = sum_{pixels p} A[p + tap][ci] * dY[p][co]        (64 ci x 64 co per block)
//
// Per 4x16-pixel tile the block stages ONE 6x18-pixel input patch (BatchNorm affine + ReLU / concat /
// zero padding applied in f32, rounded to bf16) and ONE dY tile in LDS, both pixel-major; the nine
// taps are nine shifted views of the same patch.  Each wave owns a 32x32 (ci,co) quadrant with nine
// accumulators (one per tap): per 16-pixel row it fetches one dY fragment and nine shifted A fragments
// with the transposing LDS read (ds_read_b64_tr_b16) and issues nine MFMAs.  The block walks a range
// of tiles (split-K over space) and writes one f32 slab already in the torch [Cout][Cin][3][3] layout,
// summed in fixed order by a plain streaming reduce.
#include "common.h"
#include "loader.h"

namespace ustrun {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int TH = 4, TW = 16, HW2 = TW + 2, HP = (TH + 2) * HW2;   // 108 halo pixels
constexpr int RB = 192;   // LDS row pitch: 64 channels x bf16 = 128 B + 64 B pad, so that the 4 rows of a
                          // transposed read fall on disjoint bank quarters and every address is base + immediate

// lane_base = per-lane byte offset ((8*(l>>5) + q) * RB + column bytes), k0 = first pixel row of the fragment
__device__ __forceinline__ bf16x8 tr_frag(const char* lane_base, int k0) {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lane_base + k0 * RB));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(lane_base + (k0 + 4) * RB));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// grid = (ci tiles * co tiles, ksplit); tiles_per = spatial tiles per split
template <bool POOL>
__global__ __launch_bounds__(256, 2) void wgrad_halo_bf16_kernel(const WgradArgs a, const int ntn, const int tiles_x,
                                                                 const int tiles_y, const int tiles_per) {
    constexpr int AIT = (HP * 8 + 255) / 256;      // 16-byte (8-channel) items per thread for the A patch: 4
    constexpr int BIT = (TH * TW * 8) / 256;       // ... and for the dY tile: 2
    __shared__ __attribute__((aligned(16))) char As[HP * RB];
    __shared__ __attribute__((aligned(16))) char Bs[TH * TW * RB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int mtile = blockIdx.x / ntn, ntile = blockIdx.x % ntn;
    const int ci0 = mtile * 64, co0 = ntile * 64;
    const int ttotal = a.N * tiles_y * tiles_x;
    const int tbeg = blockIdx.y * tiles_per;
    const int tend = min(ttotal, tbeg + tiles_per);

    const int c8 = tid & 7;                         // 8-channel group of this thread (same for A and dY)
    const int cg = ci0 + 8 * c8;
    const bool second = (a.nsrc == 2 && cg >= a.src[0].C);
    const SrcDev S = pick_src(a.src[0], a.src[1], second);
    const int cl = cg - (second ? a.src[0].C : 0);
    const bool aff = S.scale != nullptr;
    f32x4 asc0 = {1.f, 1.f, 1.f, 1.f}, asc1 = asc0, ash0 = {0.f, 0.f, 0.f, 0.f}, ash1 = ash0;
    if (aff) {
        asc0 = *(const f32x4*)(S.scale + cl); asc1 = *(const f32x4*)(S.scale + cl + 4);
        ash0 = *(const f32x4*)(S.shift + cl); ash1 = *(const f32x4*)(S.shift + cl + 4);
    }
    const __bf16* sp = (const __bf16*)S.ptr;
    const __bf16* dyp = (const __bf16*)a.dy;

    bf16x8 av[AIT], bv[BIT];
    unsigned aok;
    // patch coordinates of this thread's A items are tile-invariant: (hy << 8) | hx, hp >= HP -> 0xffff
    int hyx[AIT];
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
        const int hp = (tid + 256 * i) >> 3;
        hyx[i] = hp < HP ? (((hp / HW2) << 8) | (hp % HW2)) : 0xffff;
    }
    auto act8 = [&](bf16x8 r, f32x4& lo, f32x4& hi) {
        lo = (f32x4){(float)r[0], (float)r[1], (float)r[2], (float)r[3]} * asc0 + ash0;
        hi = (f32x4){(float)r[4], (float)r[5], (float)r[6], (float)r[7]} * asc1 + ash1;
        if (S.relu) { lo = relu4(lo); hi = relu4(hi); }
    };
    auto pack8 = [](f32x4 lo, f32x4 hi) {
        bf16x8 h;
        h[0] = (__bf16)lo[0]; h[1] = (__bf16)lo[1]; h[2] = (__bf16)lo[2]; h[3] = (__bf16)lo[3];
        h[4] = (__bf16)hi[0]; h[5] = (__bf16)hi[1]; h[6] = (__bf16)hi[2]; h[7] = (__bf16)hi[3];
        return h;
    };
    auto zero8 = []() { bf16x8 h; for (int q = 0; q < 8; ++q) h[q] = (__bf16)0.f; return h; };

    auto load_tile = [&](int t) {
        const int img = t / (tiles_y * tiles_x);
        const int rem = t - img * tiles_y * tiles_x;
        const int y0 = (rem / tiles_x) * TH, x0 = (rem % tiles_x) * TW;
        aok = 0;
        const int by = y0 - 1 - S.off_y, bx = x0 - 1 - S.off_x;
        const long base = img * S.sN + cl;
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            av[i] = zero8();
            const int ly = by + (hyx[i] >> 8), lx = bx + (hyx[i] & 0xff);
            if (hyx[i] != 0xffff && ly >= 0 && ly < S.LH && lx >= 0 && lx < S.LW) {
                aok |= 1u << i;
                if (POOL) {          // 2x2 max of the activated source, evaluated right here (no raw prefetch)
                    const long p = base + (long)(2 * ly) * S.sH + (long)(2 * lx) * S.sW;
                    f32x4 lo, hi, l2, h2;
                    act8(*(const bf16x8*)(sp + p), lo, hi);
                    act8(*(const bf16x8*)(sp + p + S.sW), l2, h2); lo = max4(lo, l2); hi = max4(hi, h2);
                    act8(*(const bf16x8*)(sp + p + S.sH), l2, h2); lo = max4(lo, l2); hi = max4(hi, h2);
                    act8(*(const bf16x8*)(sp + p + S.sH + S.sW), l2, h2); lo = max4(lo, l2); hi = max4(hi, h2);
                    av[i] = pack8(lo, hi);
                } else {
                    av[i] = *(const bf16x8*)(sp + base + (long)ly * S.sH + (long)lx * S.sW);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < BIT; ++i) {
            const int p = (tid + 256 * i) >> 3;              // 0..63 inside the tile
            const int oy = y0 + (p >> 4), ox = x0 + (p & 15);
            bv[i] = zero8();
            if (oy < a.dyH && ox < a.dyW)
                bv[i] = *(const bf16x8*)(dyp + (((long)img * a.dyH + oy) * a.dyW + ox) * a.Cout + co0 + 8 * c8);
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int hp = (tid + 256 * i) >> 3;
            if (hp < HP) {
                bf16x8 h = av[i];
                if (!POOL && aff && ((aok >> i) & 1u)) {     // (out-of-image items stay zero: padding is applied after the activation)
                    f32x4 lo, hi;
                    act8(av[i], lo, hi);
                    h = pack8(lo, hi);
                }
                *(bf16x8*)(As + hp * RB + c8 * 16) = h;
            }
        }
#pragma unroll
        for (int i = 0; i < BIT; ++i) {
            const int p = (tid + 256 * i) >> 3;
            *(bf16x8*)(Bs + p * RB + c8 * 16) = bv[i];
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // per-lane fragment bases: rows 8*(l>>5) + q, columns quadrant + 16*((l>>4)&1) + 4p
    const int lrow = 8 * (lane >> 5) + ((lane & 15) >> 2), lcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const char* Abase = As + lrow * RB + (wi * 32 + lcol) * 2;
    const char* Bbase = Bs + lrow * RB + (wj * 32 + lcol) * 2;

    if (tbeg < tend) load_tile(tbeg);
    for (int t = tbeg; t < tend; ++t) {
        write_tile();
        __syncthreads();
        if (t + 1 < tend) load_tile(t + 1);
#pragma unroll
        for (int r = 0; r < TH; ++r) {
            const bf16x8 b = tr_frag(Bbase, r * TW);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {       // tap (kh,kw) reads the patch at (+kh-1, +kw-1): constant offsets
                const bf16x8 af = tr_frag(Abase, (r + tap / 3) * HW2 + tap % 3);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, af, acc[tap], 0, 0, 0);   // D[co][ci]
            }
        }
        __syncthreads();
    }

    // slab in the torch weight layout [Cout][Cin][3][3]: rows of D are co, lanes are ci, and a lane
    // holds all nine taps of its (co, ci) pairs -> nine consecutive floats
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

}  // namespace

bool wgrad_halo_supported(const WgradArgs& a) {
    if (a.nseg != 9 || a.segw != 3 || a.dy_s != 1 || a.astep != 1 || a.d0 != -1 || a.dy_esz != 2) return false;
    for (int i = 0; i < a.nsrc; ++i)
        if (a.src[i].sC != 1 || a.src[i].esz != 2 || (a.src[i].C % 64) || (a.src[i].relu && !a.src[i].scale && !a.src[i].pool) || (a.src[i].pool && (a.nsrc != 1 || !a.src[i].relu))) return false;
    if (a.Cin % 64 || a.Cout % 64) return false;
    if (a.Hb < 4 || a.Wb < 8) return false;
    return true;
}

// split-K plan of the halo kernel: slabs == ksplit
int wgrad_halo_plan(const WgradArgs& a, int* ksplit, int* tiles_per) {
    const int ttotal = a.N * cdiv(a.Hb, TH) * cdiv(a.Wb, TW);
    const long pairs = (long)(a.Cin / 64) * (a.Cout / 64);
    long ks = (1024 + pairs - 1) / pairs;          // whole waves of 512 resident blocks (2 per CU)
    if (ks > ttotal / 4) ks = ttotal / 4;          // at least four tiles per block
    if (ks < 1) ks = 1;
    const int per = cdiv(ttotal, ks);
    *tiles_per = per; *ksplit = cdiv(ttotal, per);
    return 0;
}

int wgrad_halo_launch_bf16(const WgradArgs& a, int ksplit, int tiles_per, hipStream_t st) {
    dim3 grid((a.Cin / 64) * (a.Cout / 64), ksplit), block(256);
    if (a.src[0].pool)
        hipLaunchKernelGGL(wgrad_halo_bf16_kernel<true>, grid, block, 0, st, a, a.Cout / 64, cdiv(a.Wb, TW), cdiv(a.Hb, TH), tiles_per);
    else
        hipLaunchKernelGGL(wgrad_halo_bf16_kernel<false>, grid, block, 0, st, a, a.Cout / 64, cdiv(a.Wb, TW), cdiv(a.Hb, TH), tiles_per);
    USTRUN_LAUNCH_CHECK("wgrad_halo_bf16");
    return 0;
}

}  // namespace ustrun
