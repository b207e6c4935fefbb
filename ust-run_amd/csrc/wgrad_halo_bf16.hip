// wgrad_halo_bf16.hip -- 3x3 weight gradient on the bf16 matrix cores with all nine taps per block:
//
//   dW[tap][ci][co] = sum_q A[q][ci] * dY[q - (tap - center)][co]        (64 ci x 64 co per block)
//
// i.e. the taps are nine shifted views of the dY HALO patch, not of the activation: per 4x16-pixel tile the block
// stages ONE plain activation tile (BatchNorm affine + ReLU / 2x2 max / concat applied in f32, rounded to bf16 --
// 64 pixels to transform instead of a 108-pixel halo; the kernel is VALU-issue bound on exactly that arithmetic)
// and ONE 6x18-pixel dY patch, which needs no arithmetic and is fetched by LDS-DMA (padding pixels read a zero
// page).  Both are pixel-major with 192-byte rows (128 B of channels + 64 B pad: the 4 rows of a transposing
// read fall on disjoint bank quarters and every fragment address is base + immediate).  Each wave owns a 32x32
// (ci,co) quadrant with nine accumulators: per 16-pixel row one activation fragment and nine shifted dY fragments
// (ds_read_b64_tr_b16), nine MFMAs.  Tiles are double-buffered: one barrier per tile.  The block walks a range of
// tiles (split-K over space) and writes one f32 slab already in the torch [Cout][Cin][3][3] layout, summed in
// fixed order by a plain streaming reduce.
#include "common.h"
#include "loader.h"

namespace ustrun {
namespace {

typedef __attribute__((ext_vector_type(8))) elt_t bf16x8;
typedef __attribute__((ext_vector_type(4))) elt_t bf16x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int TH = 4, TW = 16, HW2 = TW + 2, HP = (TH + 2) * HW2;   // 108 halo pixels
constexpr int RB = 192;                          // LDS row pitch
constexpr int SPP = RB / 16;                     // 16-byte slots per patch pixel (8 data + 4 pad)
constexpr int DSLOTS = (HP * SPP + 63) / 64 * 64;
constexpr int DIT = (DSLOTS + 255) / 256;        // dY DMA items per thread per tile
constexpr int ATILE = TH * TW * RB, DTILE = DSLOTS * 16;

__device__ __attribute__((aligned(16))) const unsigned g_zero16w[4] = {0u, 0u, 0u, 0u};

// lane_base = per-lane byte offset ((8*(l>>5) + q) * RB + column bytes), k0 = first pixel row of the fragment
__device__ __forceinline__ bf16x8 tr_frag(const char* lane_base, int k0) {
    const bf16x4 lo = USTRUN_DS_READ_TR16((lds_bf16x4*)(lane_base + k0 * RB));
    const bf16x4 hi = USTRUN_DS_READ_TR16((lds_bf16x4*)(lane_base + (k0 + 4) * RB));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// grid = (ci tiles * co tiles, ksplit); tiles_per = spatial tiles per split
template <bool POOL>
__global__ __launch_bounds__(256, 2) void wgrad_halo_bf16_kernel(const WgradArgs a, const int ntn, const int tiles_x,
                                                                 const int tiles_y, const int tiles_per) {
    constexpr int AIT = (TH * TW * 8) / 256;       // activation items (8 channels of one pixel) per thread per tile: 2
    constexpr int NP = POOL ? 4 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                               // 2 x activation tile [64 px][RB]
    char* Ds = smem + 2 * ATILE;                   // 2 x dY patch [108 px][RB] (+ slack to whole wave-instructions)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: LDS-DMA bases and role tests stay scalar
    const int wi = wave >> 1, wj = wave & 1;
    const int mtile = blockIdx.x / ntn, ntile = blockIdx.x % ntn;
    const int ci0 = mtile * 64, co0 = ntile * 64;
    const int ttotal = a.N * tiles_y * tiles_x;
    const int tbeg = blockIdx.y * tiles_per;
    const int tend = min(ttotal, tbeg + tiles_per);

    // ---- activation side: item i = pixel (tid + 256 i) >> 3 of the tile, channel group tid & 7 ----
    const int c8 = tid & 7;
    const int cg = ci0 + 8 * c8;
    const bool second = (a.nsrc == 2 && cg >= a.src[0].C);
    const SrcDev S = pick_src(a.src[0], a.src[1], second);
    const int cl = cg - (second ? a.src[0].C : 0);
    const bool aff = S.scale != nullptr;
    const bool xf = aff || S.relu || POOL;
    f32x4 asc0 = {1.f, 1.f, 1.f, 1.f}, asc1 = asc0, ash0 = {0.f, 0.f, 0.f, 0.f}, ash1 = ash0;
    int cur_grp = -1;                              // batched passes: BatchNorm constants follow the tile's image
    auto load_consts = [&](int img) {
        const int grp = S.gN > 0 ? img / S.gN : 0;
        if (aff && grp != cur_grp) {
            const long o = (long)grp * (S.gN > 0 ? S.gstride : 0) + cl;
            asc0 = *(const f32x4*)(S.scale + o); asc1 = *(const f32x4*)(S.scale + o + 4);
            ash0 = *(const f32x4*)(S.shift + o); ash1 = *(const f32x4*)(S.shift + o + 4);
            cur_grp = grp;
        }
    };
    const elt_t* sp = (const elt_t*)S.ptr + cl;
    const elt_t* zsrc = (const elt_t*)g_zero16w;

    // ---- dY side: DMA item i = LDS slot tid + 256 i of the patch image: pixel slot / 12, group slot % 12 (>= 8: pad).
    // Tile-invariant: the relative element offset and the patch coordinates (hy << 8 | hx; 0xffff = no data) ----
    int droff[DIT], dhyx[DIT];
#pragma unroll
    for (int i = 0; i < DIT; ++i) {
        const int slot = tid + 256 * i, hp = slot / SPP, g = slot - hp * SPP;
        const int hy = hp / HW2, hx = hp - hy * HW2;
        const bool v = hp < HP && g < 8;
        droff[i] = v ? (hy * a.dyW + hx) * a.Cout + 8 * g : 0;
        dhyx[i] = v ? ((hy << 8) | hx) : 0xffff;
    }
    const elt_t* dyp = (const elt_t*)a.dy + co0;

    bf16x8 av[AIT][NP];
    unsigned aok = 0;
    auto act8 = [&](bf16x8 r, f32x4& lo, f32x4& hi) {
        lo = (f32x4){(float)r[0], (float)r[1], (float)r[2], (float)r[3]} * asc0 + ash0;
        hi = (f32x4){(float)r[4], (float)r[5], (float)r[6], (float)r[7]} * asc1 + ash1;
        if (S.relu) { lo = relu4(lo); hi = relu4(hi); }
    };
    auto tile_origin = [&](int t, int& img, int& y0, int& x0) {
        img = t / (tiles_y * tiles_x);
        const int rem = t - img * tiles_y * tiles_x;
        y0 = (rem / tiles_x) * TH; x0 = (rem % tiles_x) * TW;
    };
    // top of a stage: the whole next dY patch by DMA, the next activation items into registers
    auto fetch_tile = [&](int t, char* Dbuf) {
        int img, y0, x0;
        tile_origin(t, img, y0, x0);
        load_consts(img);                          // the items fetched below are transformed at this stage's bottom
        const elt_t* dbase = dyp + (((long)img * a.dyH + (y0 - 1)) * a.dyW + (x0 - 1)) * a.Cout;
#pragma unroll
        for (int i = 0; i < DIT; ++i) {
            if (256 * i + wave * 64 < DSLOTS) {                // wave-uniform
                const int ly = y0 - 1 + (dhyx[i] >> 8), lx = x0 - 1 + (dhyx[i] & 0xff);
                const bool ok = dhyx[i] != 0xffff && ly >= 0 && ly < a.dyH && lx >= 0 && lx < a.dyW;
                const elt_t* src = ok ? dbase + droff[i] : zsrc;
                __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(Dbuf + (256 * i + wave * 64) * 16), 16, 0, 0);
            }
        }
        aok = 0;
        const long sbase = img * S.sN;
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int px = (tid + 256 * i) >> 3;
            const int ly = y0 + (px >> 4) - S.off_y, lx = x0 + (px & 15) - S.off_x;
            if (ly >= 0 && ly < S.LH && lx >= 0 && lx < S.LW) {
                aok |= 1u << i;
                if (POOL) {
                    const long p = sbase + (long)(2 * ly) * S.sH + (long)(2 * lx) * S.sW;
                    av[i][0] = *(const bf16x8*)(sp + p);
                    av[i][1 % NP] = *(const bf16x8*)(sp + p + S.sW);
                    av[i][2 % NP] = *(const bf16x8*)(sp + p + S.sH);
                    av[i][3 % NP] = *(const bf16x8*)(sp + p + S.sH + S.sW);
                } else {
                    av[i][0] = *(const bf16x8*)(sp + sbase + (long)ly * S.sH + (long)lx * S.sW);
                }
            }
        }
    };
    // bottom of a stage: activate the fetched items and park them in the other tile buffer
    auto write_tile = [&](char* Abuf) {
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int px = (tid + 256 * i) >> 3;
            bf16x8 h;
#pragma unroll
            for (int q = 0; q < 8; ++q) h[q] = (elt_t)0.f;
            if ((aok >> i) & 1u) {                   // outside the source: zero (padding is applied after the activation)
                if (xf) {
                    f32x4 lo, hi;
                    act8(av[i][0], lo, hi);
#pragma unroll
                    for (int q = 1; q < NP; ++q) {
                        f32x4 l2, h2;
                        act8(av[i][q], l2, h2);
                        lo = max4(lo, l2); hi = max4(hi, h2);
                    }
                    h[0] = (elt_t)lo[0]; h[1] = (elt_t)lo[1]; h[2] = (elt_t)lo[2]; h[3] = (elt_t)lo[3];
                    h[4] = (elt_t)hi[0]; h[5] = (elt_t)hi[1]; h[6] = (elt_t)hi[2]; h[7] = (elt_t)hi[3];
                } else {
                    h = av[i][0];
                }
            }
            *(bf16x8*)(Abuf + px * RB + c8 * 16) = h;
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // per-lane fragment bases: rows 8*(l>>5) + q, columns quadrant + 16*((l>>4)&1) + 4p
    const int lrow = 8 * (lane >> 5) + ((lane & 15) >> 2), lcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int afrag = lrow * RB + (wi * 32 + lcol) * 2;
    const int dfrag = lrow * RB + (wj * 32 + lcol) * 2;

    if (tbeg < tend) {
        fetch_tile(tbeg, Ds);
        write_tile(As);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
#pragma unroll 1
    for (int t = tbeg; t < tend; ++t) {
        const bool more = t + 1 < tend;
        if (more) fetch_tile(t + 1, Ds + (buf ^ 1) * DTILE);
        const char* Ab = As + buf * ATILE + afrag;
        const char* Db = Ds + buf * DTILE + dfrag;
        // every dY fragment (patch row pr, column shift kw) feeds up to three taps: load it once, use it at once
        bf16x8 af[TH];
#pragma unroll
        for (int r = 0; r < TH; ++r) af[r] = tr_frag(Ab, r * TW);
#pragma unroll
        for (int pr = 0; pr < TH + 2; ++pr) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const bf16x8 b = tr_frag(Db, pr * HW2 + 2 - kw);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {      // tap (kh,kw) pairs pixel row r with dY row r + 2 - kh of the patch
                    const int r = pr + kh - 2;
                    if (r >= 0 && r < TH)
                        acc[kh * 3 + kw] = USTRUN_MFMA_32x32x16(b, af[r], acc[kh * 3 + kw], 0, 0, 0);   // D[co][ci]
                }
            }
        }
        if (more) write_tile(As + (buf ^ 1) * ATILE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
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

// ---- non-pooled sources: everything by LDS-DMA, three tile buffers ------------------------------------------------
// The operands of a weight gradient stream from HBM with no reuse in flight, so a transfer issued one tile ahead is
// not there when its tile starts (SQ_WAIT_ANY 44 % of the wave cycles with two buffers).  Here the activation tile and
// the dY patch of tile t+2 are issued at the top of tile t -- straight into LDS, no registers to carry -- and the
// bottom of tile t waits with a COUNTED vmcnt for tile t+1 only.  Rows are 128 B with the two 64-byte halves swapped
// on rows whose bit 1 is set (keeps the four rows of a transposing read on disjoint bank quarters without padding,
// so a DMA wave-instruction is exactly 8 pixels); a thread's items always hold the same channel group, whose BatchNorm
// constants it keeps in registers, and the activation is applied IN PLACE at the bottom of the previous tile.
constexpr int RB3 = 128;
// tile rows T3 = 4: three buffers, transfers two tiles ahead; T3 = 8: two buffers, one (twice as long) tile ahead --
// half the per-tile address/bounds/bookkeeping instructions per MFMA of an issue-bound kernel
template <int T3> struct H3 {
    static constexpr int HPX = (T3 + 2) * HW2;                       // halo pixels
    static constexpr int ASLOTS = T3 * TW * 8, AIT = ASLOTS / 256;   // activation items per thread: 2 or 4
    static constexpr int DSLOTS = (HPX * 8 + 63) / 64 * 64, DIT = (DSLOTS + 255) / 256;
    static constexpr int ATILE = ASLOTS * 16, DTILE = DSLOTS * 16, STAGE = ATILE + DTILE;
    static constexpr int NBUF = T3 == 4 ? 3 : 2;
};

// LDS-DMA issued from inline asm: with the builtin, hipcc puts `s_waitcnt vmcnt(0)` in front of the first transposing
// LDS read that follows (it cannot prove the read does not alias the transfer in flight), which serialises every tile
// on its own prefetch.  The kernel orders reads against landed tiles itself (counted vmcnt + barrier).
__device__ __forceinline__ void dma16(const void* gsrc, const char* lds_wave_base) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(m) : "memory", "m0");
}

__device__ __forceinline__ bf16x8 tr_frag3(const char* lane_base, int k0) {
    const bf16x4 lo = USTRUN_DS_READ_TR16((lds_bf16x4*)(lane_base + k0 * RB3));
    const bf16x4 hi = USTRUN_DS_READ_TR16((lds_bf16x4*)(lane_base + (k0 + 4) * RB3));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

template <int T3>
__global__ __launch_bounds__(256, 2) void wgrad_halo3_bf16_kernel(const WgradArgs a, const int ntn, const int tiles_x,
                                                                  const int tiles_y, const int tiles_per) {
    typedef H3<T3> G;
    constexpr int A3IT = G::AIT, D3IT = G::DIT, D3SLOTS = G::DSLOTS, A3TILE = G::ATILE, STAGE3 = G::STAGE, NBUF = G::NBUF;
    constexpr int HPX = G::HPX;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NBUF x {activation tile [T3*16 px][128 B], dY patch [HPX px][128 B]}
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: LDS-DMA bases and role tests stay scalar
    const int wi = wave >> 1, wj = wave & 1;
    // XCD-aware order (1-D grid, workgroups go round-robin over the 8 XCDs): every XCD takes a contiguous range of the
    // (slice-major, pair-minor) order, so the (ci, co) pairs of one spatial slice run on ONE XCD: the slice's activation
    // and dY tiles come from HBM once and the other pairs re-read them from that XCD's L2 -- with the natural order the
    // Cin/64 x Cout/64 re-reads were spread over all eight L2s.
    const int nblk = gridDim.x, pairs = nblk / a.ksplit;
    int lin;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = blockIdx.x % 8, j = blockIdx.x / 8;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int slice = lin / pairs, pair = lin - slice * pairs;
    const int mtile = pair / ntn, ntile = pair % ntn;
    const int ci0 = mtile * 64, co0 = ntile * 64;
    const int ttotal = a.N * tiles_y * tiles_x;
    const int tbeg = slice * tiles_per;
    const int tend = min(ttotal, tbeg + tiles_per);

    // ---- activation side: item i = LDS slot tid + 256 i: pixel (tid >> 3) + 32 i, slot tid & 7; bit 1 of the pixel is a
    // thread constant, so is the channel group the slot holds ----
    const int apx = tid >> 3;
    const int agl = (tid & 7) ^ (((apx >> 1) & 1) << 2);
    const int cg = ci0 + 8 * agl;
    const bool second = (a.nsrc == 2 && cg >= a.src[0].C);
    const SrcDev S = pick_src(a.src[0], a.src[1], second);
    const int cl = cg - (second ? a.src[0].C : 0);
    const bool aff = S.scale != nullptr;
    const bool xf = aff || S.relu;
    f32x4 asc0 = {1.f, 1.f, 1.f, 1.f}, asc1 = asc0, ash0 = {0.f, 0.f, 0.f, 0.f}, ash1 = ash0;
    int cur_grp = -1;
    auto load_consts = [&](int img) {
        const int grp = S.gN > 0 ? img / S.gN : 0;
        if (aff && grp != cur_grp) {
            const long o = (long)grp * (S.gN > 0 ? S.gstride : 0) + cl;
            asc0 = *(const f32x4*)(S.scale + o); asc1 = *(const f32x4*)(S.scale + o + 4);
            ash0 = *(const f32x4*)(S.shift + o); ash1 = *(const f32x4*)(S.shift + o + 4);
            cur_grp = grp;
        }
    };
    const elt_t* sp = (const elt_t*)S.ptr + cl;
    const elt_t* zsrc = (const elt_t*)g_zero16w;

    // ---- dY side: item i = LDS slot tid + 256 i of the patch: pixel slot >> 3, logical group (slot & 7) ^ swap ----
    int droff[D3IT], dhyx[D3IT];
#pragma unroll
    for (int i = 0; i < D3IT; ++i) {
        const int slot = tid + 256 * i, hp = slot >> 3;
        const int gl = (slot & 7) ^ (((hp >> 1) & 1) << 2);
        const int hy = hp / HW2, hx = hp - hy * HW2;
        const bool v = hp < HPX;
        droff[i] = v ? (hy * a.dyW + hx) * a.Cout + 8 * gl : 0;
        dhyx[i] = v ? ((hy << 8) | hx) : 0xffff;
    }
    const elt_t* dyp = (const elt_t*)a.dy + co0;
    const bool w4 = 256 * (D3IT - 1) + wave * 64 < D3SLOTS;       // does this wave issue the last dY piece?

    auto tile_origin = [&](int t, int& img, int& y0, int& x0) {
        img = t / (tiles_y * tiles_x);
        const int rem = t - img * tiles_y * tiles_x;
        y0 = (rem / tiles_x) * T3; x0 = (rem % tiles_x) * TW;
    };
    // all transfers of tile t -> stage buffer; returns the in-source bits of this thread's two activation items
    auto issue_tile = [&](int t, char* stage) {
        int img, y0, x0;
        tile_origin(t, img, y0, x0);
        const elt_t* dbase = dyp + (((long)img * a.dyH + (y0 - 1)) * a.dyW + (x0 - 1)) * a.Cout;
#pragma unroll
        for (int i = 0; i < D3IT; ++i) {
            if (i < D3IT - 1 || w4) {
                const int ly = y0 - 1 + (dhyx[i] >> 8), lx = x0 - 1 + (dhyx[i] & 0xff);
                const bool ok = dhyx[i] != 0xffff && ly >= 0 && ly < a.dyH && lx >= 0 && lx < a.dyW;
                const elt_t* src = ok ? dbase + droff[i] : zsrc;
                dma16(src, stage + A3TILE + (256 * i + wave * 64) * 16);
            }
        }
        unsigned ok2 = 0;
        const long sbase = img * S.sN;
#pragma unroll
        for (int i = 0; i < A3IT; ++i) {
            const int px = apx + 32 * i;
            const int ly = y0 + (px >> 4) - S.off_y, lx = x0 + (px & 15) - S.off_x;
            const bool ok = ly >= 0 && ly < S.LH && lx >= 0 && lx < S.LW;
            ok2 |= (ok ? 1u : 0u) << i;
            const elt_t* src = ok ? sp + sbase + (long)ly * S.sH + (long)lx * S.sW : zsrc;
            dma16(src, stage + (256 * i + wave * 64) * 16);
        }
        return ok2;
    };
    // BatchNorm affine + ReLU of this thread's two items, in place (items outside the source stay zero)
    auto activate = [&](char* stage, unsigned ok2) {
        if (!xf) return;
        // all items read first (one LDS round trip), selects instead of per-item branches (items outside the source
        // hold zeros and must stay zero: relu(shift) is not)
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const float floor_ = S.relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int i0 = 0; i0 < A3IT; i0 += 2) {              // two items per LDS round trip (four would spill)
        bf16x8 r[A3IT];
#pragma unroll
        for (int i = i0; i < i0 + 2 && i < A3IT; ++i) r[i] = *(const bf16x8*)(stage + (tid + 256 * i) * 16);
#pragma unroll
        for (int i = i0; i < i0 + 2 && i < A3IT; ++i) {
            f32x4 lo = (f32x4){(float)r[i][0], (float)r[i][1], (float)r[i][2], (float)r[i][3]} * asc0 + ash0;
            f32x4 hi = (f32x4){(float)r[i][4], (float)r[i][5], (float)r[i][6], (float)r[i][7]} * asc1 + ash1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lo[q] = __builtin_amdgcn_fmed3f(lo[q], floor_, __builtin_inff());
                hi[q] = __builtin_amdgcn_fmed3f(hi[q], floor_, __builtin_inff());
            }
            bf16x8 h;
            h[0] = (elt_t)lo[0]; h[1] = (elt_t)lo[1]; h[2] = (elt_t)lo[2]; h[3] = (elt_t)lo[3];
            h[4] = (elt_t)hi[0]; h[5] = (elt_t)hi[1]; h[6] = (elt_t)hi[2]; h[7] = (elt_t)hi[3];
            u32x4 u = __builtin_bit_cast(u32x4, h);
            const bool ok = (ok2 >> i) & 1u;
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = ok ? u[q] : 0u;
            *(u32x4*)(stage + (tid + 256 * i) * 16) = u;
        }
        }
    };
    auto tile_img = [&](int t) { return t / (tiles_y * tiles_x); };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // per-lane fragment bases: rows 8*(l>>5) + q; the half swap of a row depends on bit 1 of (k0 + row), i.e. on k0 & 3
    const int lrow = 8 * (lane >> 5) + ((lane & 15) >> 2), lcolb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    int abase, dbase4[4];
    abase = lrow * RB3 + ((wi * 64 + lcolb) ^ (((lrow >> 1) & 1) << 6));
#pragma unroll
    for (int c = 0; c < 4; ++c) dbase4[c] = A3TILE + lrow * RB3 + ((wj * 64 + lcolb) ^ ((((c + lrow) >> 1) & 1) << 6));

    // transfers of one tile, per wave: D3IT (- 1 on the waves without a last dY piece) + A3IT
    auto wait_all_but_one_tile = [&]() {
        if (w4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D3IT + A3IT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D3IT - 1 + A3IT) : "memory");
    };
    unsigned ok_q[3] = {0, 0, 0};                    // in-source bits of the tiles in flight (tile t + k at index k)
    char* st[3] = {smem, smem + STAGE3, smem + (NBUF - 1) * STAGE3};     // buffers of tile t, t+1, (t+2)
    constexpr int PD = NBUF - 1;                     // tiles issued ahead
    if (tbeg < tend) {
        load_consts(tile_img(tbeg));
        ok_q[0] = issue_tile(tbeg, st[0]);
        if (PD == 2 && tbeg + 1 < tend) {
            ok_q[1] = issue_tile(tbeg + 1, st[1]);
            wait_all_but_one_tile();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        activate(st[0], ok_q[0]);
    }
    __syncthreads();
#pragma unroll 1
    for (int t = tbeg; t < tend; ++t) {
        // top: tile t+PD into the buffer tile t-1 used
        const bool issue = t + PD < tend;
        if (issue) ok_q[PD] = issue_tile(t + PD, st[PD]);
#pragma unroll
        for (int h = 0; h < T3 / 4; ++h) {           // four tile rows at a time: their dY fragments are patch rows 4h .. 4h+5
            const char* Ab = st[0] + abase;
            bf16x8 af[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) af[r] = tr_frag3(Ab, (4 * h + r) * TW);
            // the three dY fragments of the next patch row are in flight while this row's MFMAs run (one LDS round trip
            // per row instead of one per fragment)
            bf16x8 bq[2][3];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int k0 = 4 * h * HW2 + 2 - kw;
                bq[0][kw] = tr_frag3(st[0] + dbase4[k0 & 3], k0);
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                if (pr + 1 < 6) {
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int k0 = (4 * h + pr + 1) * HW2 + 2 - kw;
                        bq[(pr + 1) & 1][kw] = tr_frag3(st[0] + dbase4[k0 & 3], k0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {      // tap (kh,kw) pairs tile row r with dY row r + 2 - kh of the patch
                        const int r = pr + kh - 2;
                        if (r >= 0 && r < 4)
                            acc[kh * 3 + kw] = USTRUN_MFMA_32x32x16(bq[pr & 1][kw], af[r], acc[kh * 3 + kw], 0, 0, 0);   // D[co][ci]
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // bottom: tile t+1 has landed once only the transfers of later tiles are outstanding
        __builtin_amdgcn_sched_barrier(0);          // keep the waits behind the MFMAs
        if (t + 1 < tend) {
            if (PD == 2 && issue) wait_all_but_one_tile();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            load_consts(tile_img(t + 1));
            activate(st[1], ok_q[1]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (NBUF == 3) { char* tmp = st[0]; st[0] = st[1]; st[1] = st[2]; st[2] = tmp; ok_q[0] = ok_q[1]; ok_q[1] = ok_q[2]; }
        else { char* tmp = st[0]; st[0] = st[1]; st[1] = tmp; st[2] = st[1]; ok_q[0] = ok_q[1]; }
    }

    float* slab = a.partials + (long)slice * 9 * a.Cin * a.Cout;
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


// ---- round 3: the same tiles with buffer-addressed LDS-DMA ---------------------------------------------------------
// The kernel above spends 463 of its 833 loop instructions per tile (72 MFMAs) on the ten transfers' addresses: 64-bit
// pointer arithmetic, per-item bounds tests behind exec-mask branches and the zero page.  Here the tile's two base
// addresses are wave-uniform (a cursor in SGPRs, no divisions) and go into two buffer resources; an item is a per-lane
// CONSTANT byte offset (tile-invariant), padding is the buffer's own range check (offset bit 31 set -> the transfer
// writes zeros), and tiles whose halo lies inside the image -- most of them -- skip the per-item tests altogether.
// The activation pass drops its masks on such tiles too, and takes ReLU on the rounded bf16 pairs (v_pk_max_i16 against
// zero: the same values as max in f32 before rounding, half the instructions).
__device__ __forceinline__ void dma16_buf(int voff, __amdgpu_buffer_rsrc_t rs, const char* lds_wave_base) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) char*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rs), "s"(m) : "memory", "m0");
}

template <int T3>
__global__ __launch_bounds__(256, 2) void wgrad_halo4_bf16_kernel(const WgradArgs a, const int ntn, const int tiles_x,
                                                                  const int tiles_y, const int tiles_per) {
    typedef H3<T3> G;
    constexpr int A3IT = G::AIT, D3IT = G::DIT, D3SLOTS = G::DSLOTS, A3TILE = G::ATILE, STAGE3 = G::STAGE, NBUF = G::NBUF;
    constexpr int HPX = G::HPX;
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NBUF x {activation tile [T3*16 px][128 B], dY patch [HPX px][128 B]}
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int nblk = gridDim.x, pairs = nblk / a.ksplit;          // XCD-contiguous slices, as above
    int lin;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = blockIdx.x % 8, j = blockIdx.x / 8;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int slice = lin / pairs, pair = lin - slice * pairs;
    const int mtile = pair / ntn, ntile = pair % ntn;
    const int ci0 = mtile * 64, co0 = ntile * 64;
    const int ttotal = a.N * tiles_y * tiles_x;
    const int tbeg = slice * tiles_per;
    const int tend = min(ttotal, tbeg + tiles_per);

    // the 64 input channels of a block come from ONE source (source widths are multiples of 64)
    const bool second = (a.nsrc == 2 && ci0 >= a.src[0].C);
    const SrcDev S = pick_src(a.src[0], a.src[1], second);
    const int cl0 = ci0 - (second ? a.src[0].C : 0);
    const bool aff = S.scale != nullptr;
    const bool xf = aff || S.relu;
    const int apx = tid >> 3;
    const int agl = (tid & 7) ^ (((apx >> 1) & 1) << 2);
    f32x4 asc0 = {1.f, 1.f, 1.f, 1.f}, asc1 = asc0, ash0 = {0.f, 0.f, 0.f, 0.f}, ash1 = ash0;
    int cur_grp = -1;
    auto load_consts = [&](int img) {
        const int grp = S.gN > 0 ? img / S.gN : 0;
        if (aff && grp != cur_grp) {
            const long o = (long)grp * (S.gN > 0 ? S.gstride : 0) + cl0 + 8 * agl;
            asc0 = *(const f32x4*)(S.scale + o); asc1 = *(const f32x4*)(S.scale + o + 4);
            ash0 = *(const f32x4*)(S.shift + o); ash1 = *(const f32x4*)(S.shift + o + 4);
            cur_grp = grp;
        }
    };
    // tile-invariant per-lane byte offsets from the tile's base: activation item i = pixel apx + 32 i (row 2 i + (apx >> 4),
    // column apx & 15), channel group agl; dY item i = slot tid + 256 i of the patch
    const int sH2 = (int)S.sH * 2, sW2 = (int)S.sW * 2;
    int aoff[A3IT];
#pragma unroll
    for (int i = 0; i < A3IT; ++i) aoff[i] = (2 * i + (apx >> 4)) * sH2 + (apx & 15) * sW2 + (cl0 + 8 * agl) * 2;
    int droff[D3IT], dhyx[D3IT];
#pragma unroll
    for (int i = 0; i < D3IT; ++i) {
        const int slot = tid + 256 * i, hp = slot >> 3;
        const int gl = (slot & 7) ^ (((hp >> 1) & 1) << 2);
        const int hy = hp / HW2, hx = hp - hy * HW2;
        const bool v = hp < HPX;
        droff[i] = v ? ((hy * a.dyW + hx) * a.Cout + 8 * gl) * 2 : OOB;
        dhyx[i] = v ? ((hy << 8) | hx) : 0xffff;
    }
    const bool w4 = 256 * (D3IT - 1) + wave * 64 < D3SLOTS;       // does this wave issue the last dY piece?
    const char* const sbytes = (const char*)S.ptr;
    const char* const dbytes = (const char*)a.dy + 2 * co0;
    const long sN2 = S.sN * 2, dN2 = (long)a.dyH * a.dyW * a.Cout * 2;
    const int dW2 = a.dyW * a.Cout * 2, dP2 = a.Cout * 2;

    // cursor of the next tile to issue (wave-uniform)
    // tiles of an image are walked DOWN its columns (y fastest): the two dY halo rows a tile shares with the next one are in L2
    // when that tile asks for them (walking along x left them a whole tile row = MBs of other blocks' traffic apart, and every
    // halo row came from memory twice: 25 % on top of dY; the halo columns now re-read instead are 12.5 %)
    int q_img, q_y0, q_x0;
    {
        q_img = tbeg / (tiles_y * tiles_x);
        const int rem = tbeg - q_img * tiles_y * tiles_x;
        q_x0 = (rem / tiles_y) * TW; q_y0 = (rem % tiles_y) * T3;
    }
    // all transfers of the cursor's tile -> stage buffer; bit i of the result: activation item i lies in the source; bit 8: the
    // whole activation tile does (no masks needed); bits 16..: the tile's image
    auto issue_tile = [&](char* stage) {
        const int img = q_img, y0 = q_y0, x0 = q_x0;
        {
            const char* dbase = dbytes + (long)img * dN2 + ((y0 - 1) * dW2 + (x0 - 1) * dP2);
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dbase, 0, 0x7fffffff, 0x00020000);
            const bool inner = y0 >= 1 && y0 + T3 + 1 <= a.dyH && x0 >= 1 && x0 + TW + 1 <= a.dyW;
            if (inner) {
#pragma unroll
                for (int i = 0; i < D3IT; ++i)
                    if (i < D3IT - 1 || w4) dma16_buf(droff[i], rd, stage + A3TILE + (256 * i + wave * 64) * 16);
            } else {
#pragma unroll
                for (int i = 0; i < D3IT; ++i) {
                    if (i < D3IT - 1 || w4) {
                        const unsigned ly = (unsigned)(y0 - 1 + (dhyx[i] >> 8)), lx = (unsigned)(x0 - 1 + (dhyx[i] & 0xff));
                        const bool ok = ly < (unsigned)a.dyH && lx < (unsigned)a.dyW;      // slots beyond the patch: droff is OOB already
                        dma16_buf(ok ? droff[i] : OOB, rd, stage + A3TILE + (256 * i + wave * 64) * 16);
                    }
                }
            }
        }
        unsigned ok2;
        {
            const int ty = y0 - S.off_y, tx = x0 - S.off_x;
            const char* abase = sbytes + (long)img * sN2 + (ty * sH2 + tx * sW2);
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, 0x7fffffff, 0x00020000);
            const bool inner = ty >= 0 && ty + T3 <= S.LH && tx >= 0 && tx + TW <= S.LW;
            if (inner) {
                ok2 = 0x100u | ((1u << A3IT) - 1u);
#pragma unroll
                for (int i = 0; i < A3IT; ++i) dma16_buf(aoff[i], ra, stage + (256 * i + wave * 64) * 16);
            } else {
                ok2 = 0;
#pragma unroll
                for (int i = 0; i < A3IT; ++i) {
                    const unsigned ly = (unsigned)(ty + 2 * i + (apx >> 4)), lx = (unsigned)(tx + (apx & 15));
                    const bool ok = ly < (unsigned)S.LH && lx < (unsigned)S.LW;
                    ok2 |= (ok ? 1u : 0u) << i;
                    dma16_buf(ok ? aoff[i] : OOB, ra, stage + (256 * i + wave * 64) * 16);
                }
            }
        }
        ok2 |= (unsigned)img << 16;
        q_y0 += T3;
        if (q_y0 >= tiles_y * T3) {
            q_y0 = 0; q_x0 += TW;
            if (q_x0 >= tiles_x * TW) { q_x0 = 0; ++q_img; }
        }
        return ok2;
    };
    // BatchNorm affine + ReLU of this thread's items, in place (items outside the source stay zero)
    auto act_item = [&](u32x4 r, bool relu) {
        const bf16x8 b = __builtin_bit_cast(bf16x8, r);
        f32x4 lo, hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {              // scalar FMAs on purpose: beside another wave's MFMAs a v_pk_fma_f32 costs more than two v_fma_f32
            lo[q] = fma_scalar((float)b[q], asc0[q], ash0[q]);
            hi[q] = fma_scalar((float)b[4 + q], asc1[q], ash1[q]);
        }
        bf16x8 h;
        h[0] = (elt_t)lo[0]; h[1] = (elt_t)lo[1]; h[2] = (elt_t)lo[2]; h[3] = (elt_t)lo[3];
        h[4] = (elt_t)hi[0]; h[5] = (elt_t)hi[1]; h[6] = (elt_t)hi[2]; h[7] = (elt_t)hi[3];
        if (relu) {
            const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, h), z));
        }
        return __builtin_bit_cast(u32x4, h);
    };
    auto activate = [&](char* stage, unsigned ok2) {
        if (!xf) return;
        const bool inner = (ok2 >> 8) & 1u;
#pragma unroll
        for (int i0 = 0; i0 < A3IT; i0 += 2) {              // two items per LDS round trip (four would spill)
            u32x4 r[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) r[i] = *(const u32x4*)(stage + (tid + 256 * (i0 + i)) * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                u32x4 u = S.relu ? act_item(r[i], true) : act_item(r[i], false);
                if (!inner) {
                    const bool ok = (ok2 >> (i0 + i)) & 1u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) u[q] = ok ? u[q] : 0u;
                }
                *(u32x4*)(stage + (tid + 256 * (i0 + i)) * 16) = u;
            }
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int lrow = 8 * (lane >> 5) + ((lane & 15) >> 2), lcolb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    int abase, dbase4[4];
    abase = lrow * RB3 + ((wi * 64 + lcolb) ^ (((lrow >> 1) & 1) << 6));
#pragma unroll
    for (int c = 0; c < 4; ++c) dbase4[c] = A3TILE + lrow * RB3 + ((wj * 64 + lcolb) ^ ((((c + lrow) >> 1) & 1) << 6));

    auto wait_all_but_one_tile = [&]() {
        if (w4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D3IT + A3IT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D3IT - 1 + A3IT) : "memory");
    };
    unsigned ok_q[3] = {0, 0, 0};
    char* st[3] = {smem, smem + STAGE3, smem + (NBUF - 1) * STAGE3};
    constexpr int PD = NBUF - 1;
    if (tbeg < tend) {
        ok_q[0] = issue_tile(st[0]);
        load_consts(ok_q[0] >> 16);
        if (PD == 2 && tbeg + 1 < tend) {
            ok_q[1] = issue_tile(st[1]);
            wait_all_but_one_tile();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        activate(st[0], ok_q[0]);
    }
    __syncthreads();
#pragma unroll 1
    for (int t = tbeg; t < tend; ++t) {
        const bool issue = t + PD < tend;
        if (issue) ok_q[PD] = issue_tile(st[PD]);
#pragma unroll
        for (int h = 0; h < T3 / 4; ++h) {
            const char* Ab = st[0] + abase;
            bf16x8 af[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) af[r] = tr_frag3(Ab, (4 * h + r) * TW);
            bf16x8 bq[2][3];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int k0 = 4 * h * HW2 + 2 - kw;
                bq[0][kw] = tr_frag3(st[0] + dbase4[k0 & 3], k0);
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                if (pr + 1 < 6) {
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int k0 = (4 * h + pr + 1) * HW2 + 2 - kw;
                        bq[(pr + 1) & 1][kw] = tr_frag3(st[0] + dbase4[k0 & 3], k0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const int r = pr + kh - 2;
                        if (r >= 0 && r < 4)
                            acc[kh * 3 + kw] = USTRUN_MFMA_32x32x16(bq[pr & 1][kw], af[r], acc[kh * 3 + kw], 0, 0, 0);   // D[co][ci]
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < tend) {
            if (PD == 2 && issue) wait_all_but_one_tile();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            load_consts(ok_q[1] >> 16);
            activate(st[1], ok_q[1]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (NBUF == 3) { char* tmp = st[0]; st[0] = st[1]; st[1] = st[2]; st[2] = tmp; ok_q[0] = ok_q[1]; ok_q[1] = ok_q[2]; }
        else { char* tmp = st[0]; st[0] = st[1]; st[1] = tmp; st[2] = st[1]; ok_q[0] = ok_q[1]; }
    }

    float* slab = a.partials + (long)slice * 9 * a.Cin * a.Cout;
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


// ---- round 3: two wave groups in opposite phases ("ping-pong") ------------------------------------------------------
// Two 256-thread blocks per CU run the kernel above in step with each other: both multiply at the same time (sharing
// the matrix pipe), both compute addresses / activate at the same time (pipe idle) -- one block per CU alone reaches
// 79 % of the pair's rate (profiles/r03_wgrad_occupancy.log).  Here ONE 512-thread block per CU holds two groups of
// four waves, each with its own accumulators, its own two stage buffers and every other tile of the block's range;
// block-wide barriers separate PHASES, and in every phase one group multiplies its current tile (72 MFMAs per wave)
// while the other issues the transfers of its tile after next, waits for its next tile (issued two phases = one whole
// period earlier) and activates it.  The matrix pipe of a SIMD is fed by one wave at a time, the other wave's VALU /
// SALU / waiting hides behind it.  The groups' accumulators are summed through LDS at the end (group 0 + group 1, a
// fixed order): half as many slabs as before for the same tiles per wave.
__device__ __forceinline__ unsigned long long stamp_pp() {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
// DIAG: phase stamps per wave into dbg[block][wave 0..7][8] (0: multiplying, 1: barrier wait after it, 2: preparing, 3: barrier
// wait after it, 4: phases without work, 5: tiles multiplied, 6: of 2, the wait for the
// tile's transfers, 7: of 2, issuing the next tile's transfers); a development build, never launched by the product
template <bool DIAG>
__global__ __launch_bounds__(512, 1) void wgrad_halo_pp_bf16_kernel(const WgradArgs a, const int ntn, const int tiles_x,
                                                                    const int tiles_y, const int tiles_per,
                                                                    unsigned long long* __restrict__ dbg) {
    constexpr int T3 = 8;
    typedef H3<T3> G;
    constexpr int A3IT = G::AIT, D3IT = G::DIT, D3SLOTS = G::DSLOTS, A3TILE = G::ATILE, STAGE3 = G::STAGE;
    constexpr int HPX = G::HPX;
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 groups x 2 x {activation tile, dY patch}
    const int lane = threadIdx.x & 63, gt = threadIdx.x & 255;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3;
    const int wi = wave >> 1, wj = wave & 1;
    const int nblk = gridDim.x, pairs = nblk / a.ksplit;
    int lin;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = blockIdx.x % 8, j = blockIdx.x / 8;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int slice = lin / pairs, pair = lin - slice * pairs;
    const int mtile = pair / ntn, ntile = pair % ntn;
    const int ci0 = mtile * 64, co0 = ntile * 64;
    const int ttotal = a.N * tiles_y * tiles_x;
    const int tbeg = slice * tiles_per;
    const int tend = min(ttotal, tbeg + tiles_per);
    const int ntl = max(tend - tbeg, 0);
    const int nk = (ntl + 1 - grp) >> 1;                            // this group's tiles: tbeg + grp + 2 k
    const int pend = max(2 * (((ntl + 1) >> 1) - 1), 2 * ((ntl >> 1) - 1) + 1);     // last phase with a multiplication

    const bool second = (a.nsrc == 2 && ci0 >= a.src[0].C);
    const SrcDev S = pick_src(a.src[0], a.src[1], second);
    const int cl0 = ci0 - (second ? a.src[0].C : 0);
    const bool aff = S.scale != nullptr;
    const bool xf = aff || S.relu;
    const int apx = gt >> 3;
    const int agl = (gt & 7) ^ (((apx >> 1) & 1) << 2);
    f32x4 asc0 = {1.f, 1.f, 1.f, 1.f}, asc1 = asc0, ash0 = {0.f, 0.f, 0.f, 0.f}, ash1 = ash0;
    int cur_grp = -1;
    auto load_consts = [&](int img) {
        const int g = S.gN > 0 ? img / S.gN : 0;
        if (aff && g != cur_grp) {
            const long o = (long)g * (S.gN > 0 ? S.gstride : 0) + cl0 + 8 * agl;
            asc0 = *(const f32x4*)(S.scale + o); asc1 = *(const f32x4*)(S.scale + o + 4);
            ash0 = *(const f32x4*)(S.shift + o); ash1 = *(const f32x4*)(S.shift + o + 4);
            cur_grp = g;
        }
    };
    const int sH2 = (int)S.sH * 2, sW2 = (int)S.sW * 2;
    int aoff[A3IT];
#pragma unroll
    for (int i = 0; i < A3IT; ++i) aoff[i] = (2 * i + (apx >> 4)) * sH2 + (apx & 15) * sW2 + (cl0 + 8 * agl) * 2;
    int droff[D3IT], dhyx[D3IT];
#pragma unroll
    for (int i = 0; i < D3IT; ++i) {
        const int slot = gt + 256 * i, hp = slot >> 3;
        const int gl = (slot & 7) ^ (((hp >> 1) & 1) << 2);
        const int hy = hp / HW2, hx = hp - hy * HW2;
        const bool v = hp < HPX;
        droff[i] = v ? ((hy * a.dyW + hx) * a.Cout + 8 * gl) * 2 : OOB;
        dhyx[i] = v ? ((hy << 8) | hx) : 0xffff;
    }
    const bool w4 = 256 * (D3IT - 1) + wave * 64 < D3SLOTS;
    const char* const sbytes = (const char*)S.ptr;
    const char* const dbytes = (const char*)a.dy + 2 * co0;
    const long sN2 = S.sN * 2, dN2 = (long)a.dyH * a.dyW * a.Cout * 2;
    const int dW2 = a.dyW * a.Cout * 2, dP2 = a.Cout * 2;

    int q_img, q_y0, q_x0;                                          // cursor of the group's next tile to issue
    {
        const int t0 = tbeg + grp;                                  // (y fastest, as in the kernel above)
        q_img = t0 / (tiles_y * tiles_x);
        const int rem = t0 - q_img * tiles_y * tiles_x;
        q_x0 = (rem / tiles_y) * TW; q_y0 = (rem % tiles_y) * T3;
    }
    auto issue_tile = [&](char* stage) {
        const int img = q_img, y0 = q_y0, x0 = q_x0;
        {
            const char* dbase = dbytes + (long)img * dN2 + ((y0 - 1) * dW2 + (x0 - 1) * dP2);
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dbase, 0, 0x7fffffff, 0x00020000);
            const bool inner = y0 >= 1 && y0 + T3 + 1 <= a.dyH && x0 >= 1 && x0 + TW + 1 <= a.dyW;
            if (inner) {
#pragma unroll
                for (int i = 0; i < D3IT; ++i)
                    if (i < D3IT - 1 || w4) dma16_buf(droff[i], rd, stage + A3TILE + (256 * i + wave * 64) * 16);
            } else {
#pragma unroll
                for (int i = 0; i < D3IT; ++i) {
                    if (i < D3IT - 1 || w4) {
                        const unsigned ly = (unsigned)(y0 - 1 + (dhyx[i] >> 8)), lx = (unsigned)(x0 - 1 + (dhyx[i] & 0xff));
                        const bool ok = ly < (unsigned)a.dyH && lx < (unsigned)a.dyW;
                        dma16_buf(ok ? droff[i] : OOB, rd, stage + A3TILE + (256 * i + wave * 64) * 16);
                    }
                }
            }
        }
        unsigned ok2;
        {
            const int ty = y0 - S.off_y, tx = x0 - S.off_x;
            const char* abase = sbytes + (long)img * sN2 + (ty * sH2 + tx * sW2);
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)abase, 0, 0x7fffffff, 0x00020000);
            const bool inner = ty >= 0 && ty + T3 <= S.LH && tx >= 0 && tx + TW <= S.LW;
            if (inner) {
                ok2 = 0x100u | ((1u << A3IT) - 1u);
#pragma unroll
                for (int i = 0; i < A3IT; ++i) dma16_buf(aoff[i], ra, stage + (256 * i + wave * 64) * 16);
            } else {
                ok2 = 0;
#pragma unroll
                for (int i = 0; i < A3IT; ++i) {
                    const unsigned ly = (unsigned)(ty + 2 * i + (apx >> 4)), lx = (unsigned)(tx + (apx & 15));
                    const bool ok = ly < (unsigned)S.LH && lx < (unsigned)S.LW;
                    ok2 |= (ok ? 1u : 0u) << i;
                    dma16_buf(ok ? aoff[i] : OOB, ra, stage + (256 * i + wave * 64) * 16);
                }
            }
        }
        ok2 |= (unsigned)img << 16;
#pragma unroll
        for (int s = 0; s < 2; ++s) {                               // the group's next tile is two tiles on
            q_y0 += T3;
            if (q_y0 >= tiles_y * T3) {
                q_y0 = 0; q_x0 += TW;
                if (q_x0 >= tiles_x * TW) { q_x0 = 0; ++q_img; }
            }
        }
        return ok2;
    };
    auto act_item = [&](u32x4 r, bool relu) {
        const bf16x8 b = __builtin_bit_cast(bf16x8, r);
        f32x4 lo, hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {              // scalar FMAs on purpose: beside another wave's MFMAs a v_pk_fma_f32 costs more than two v_fma_f32
            lo[q] = fma_scalar((float)b[q], asc0[q], ash0[q]);
            hi[q] = fma_scalar((float)b[4 + q], asc1[q], ash1[q]);
        }
        bf16x8 h;
        h[0] = (elt_t)lo[0]; h[1] = (elt_t)lo[1]; h[2] = (elt_t)lo[2]; h[3] = (elt_t)lo[3];
        h[4] = (elt_t)hi[0]; h[5] = (elt_t)hi[1]; h[6] = (elt_t)hi[2]; h[7] = (elt_t)hi[3];
        if (relu) {
            const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(s16x8, h), z));
        }
        return __builtin_bit_cast(u32x4, h);
    };
    auto activate = [&](char* stage, unsigned ok2) {
        if (!xf) return;
        const bool inner = (ok2 >> 8) & 1u;
        u32x4 r[A3IT];
#pragma unroll
        for (int i = 0; i < A3IT; ++i) r[i] = *(const u32x4*)(stage + (gt + 256 * i) * 16);       // one LDS round trip
#pragma unroll
        for (int i = 0; i < A3IT; ++i) {
            u32x4 u = S.relu ? act_item(r[i], true) : act_item(r[i], false);
            if (!inner) {
                const bool ok = (ok2 >> i) & 1u;
#pragma unroll
                for (int q = 0; q < 4; ++q) u[q] = ok ? u[q] : 0u;
            }
            *(u32x4*)(stage + (gt + 256 * i) * 16) = u;
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int lrow = 8 * (lane >> 5) + ((lane & 15) >> 2), lcolb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    int abase, dbase4[4];
    abase = lrow * RB3 + ((wi * 64 + lcolb) ^ (((lrow >> 1) & 1) << 6));
#pragma unroll
    for (int c = 0; c < 4; ++c) dbase4[c] = A3TILE + lrow * RB3 + ((wj * 64 + lcolb) ^ ((((c + lrow) >> 1) & 1) << 6));

    auto multiply_tile = [&](const char* stage) {
#pragma unroll
        for (int h = 0; h < T3 / 4; ++h) {
            const char* Ab = stage + abase;
            bf16x8 af[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) af[r] = tr_frag3(Ab, (4 * h + r) * TW);
            bf16x8 bq[2][3];
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int k0 = 4 * h * HW2 + 2 - kw;
                bq[0][kw] = tr_frag3(stage + dbase4[k0 & 3], k0);
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                if (pr + 1 < 6) {
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int k0 = (4 * h + pr + 1) * HW2 + 2 - kw;
                        bq[(pr + 1) & 1][kw] = tr_frag3(stage + dbase4[k0 & 3], k0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const int r = pr + kh - 2;
                        if (r >= 0 && r < 4)
                            acc[kh * 3 + kw] = USTRUN_MFMA_32x32x16(bq[pr & 1][kw], af[r], acc[kh * 3 + kw], 0, 0, 0);   // D[co][ci]
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto wait_all_but_one_tile = [&]() {
        if (w4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D3IT + A3IT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D3IT - 1 + A3IT) : "memory");
    };

    char* stC = smem + grp * 2 * STAGE3;           // the group's current tile (being activated, then multiplied)
    char* stN = stC + STAGE3;                      // the tile after it (in flight)
    unsigned okC = 0, okN = 0;
    if (nk > 0) okC = issue_tile(stC);
    unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, dt0 = 0, dt1 = 0, dsum6 = 0, dsum7 = 0;
#pragma unroll 1
    for (int p = -1; p <= pend; ++p) {
        int role = 4;
        if constexpr (DIAG) dt0 = stamp_pp();
        if ((p & 1) == grp) {
            const int k = p >> 1;
            if (k >= 0 && k < nk) {
                multiply_tile(stC);
                char* tmp = stC; stC = stN; stN = tmp;
                okC = okN;
                role = 0;
                if constexpr (DIAG) ++dsum[5];
            }
        } else {
            const int k = (p + 1) >> 1;            // prepare the group's tile k for the next phase
            if (k < nk) {
                const bool more = k + 1 < nk;
                unsigned long long ta = 0, tb = 0;
                if (more) okN = issue_tile(stN);
                if constexpr (DIAG) ta = stamp_pp();
                if (more) wait_all_but_one_tile();
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if constexpr (DIAG) tb = stamp_pp();
                load_consts(okC >> 16);
                activate(stC, okC);
                role = 2;
                if constexpr (DIAG) { dsum6 += tb - ta; dsum7 += ta - dt0; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (DIAG) { dt1 = stamp_pp(); dsum[role] += dt1 - dt0; }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (DIAG) { if (role != 4) dsum[role + 1] += stamp_pp() - dt1; }
    }
    if constexpr (DIAG) {
        if (lane == 0 && dbg)
            for (int k = 0; k < 8; ++k) dbg[((long)blockIdx.x * 8 + wave8) * 8 + k] = k < 6 ? dsum[k] : k == 6 ? dsum6 : dsum7;
    }

    // group 1's accumulators -> LDS, group 0 adds them to its own and writes the slab
    float* red = (float*)smem;
    if (grp == 1) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((tap * 16 + r) * 4 + wave) * 64 + lane] = acc[tap][r];
    }
    __syncthreads();
    if (grp == 0) {
        float* slab = a.partials + (long)slice * 9 * a.Cin * a.Cout;
        const int l31 = lane & 31, lh = lane >> 5;
        const int ci = ci0 + wi * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wj * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float* o = slab + ((long)co * a.Cin + ci) * 9;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) o[tap] = acc[tap][r] + red[((tap * 16 + r) * 4 + wave) * 64 + lane];
        }
    }
}

}  // namespace

bool wgrad_halo_supported(const WgradArgs& a) {
    if (a.nseg != 9 || a.segw != 3 || a.dy_s != 1 || a.astep != 1 || a.d0 != -1 || a.ashift != 0 || a.dy_esz != 2) return false;
    for (int i = 0; i < a.nsrc; ++i)
        if (a.src[i].sC != 1 || a.src[i].esz != 2 || (a.src[i].C % 64) || (a.src[i].relu && !a.src[i].scale && !a.src[i].pool) || (a.src[i].pool && (a.nsrc != 1 || !a.src[i].relu))) return false;
    if (a.Cin % 64 || a.Cout % 64) return false;
    if (a.Hb < 4 || a.Wb < 8) return false;
    return true;
}

// tile rows of the non-pooled kernel: 8 (two buffers) once the extent has eight rows, else 4 (three buffers)
static int halo3_rows(const WgradArgs& a) { return (!a.src[0].pool && a.Hb >= 8) ? 8 : 4; }

// buffer-addressed transfers (round 3) need 32-bit byte offsets inside one image of either operand
static bool halo_buf_ok(const WgradArgs& a) {
    bool buf = !(g_debug_flags & 64) && (long)a.dyH * a.dyW * a.Cout * 2 < (1L << 31);
    for (int i = 0; i < a.nsrc; ++i) buf = buf && a.src[i].sN * 2 < (1L << 31) && a.src[i].sH * 2 * 16 < (1L << 31);
    return buf;
}
// the two-group kernel: eight-row tiles, at least eight of them per block
static bool halo_pp_ok(const WgradArgs& a) {
    if ((g_debug_flags & 256) || halo3_rows(a) != 8 || !halo_buf_ok(a)) return false;
    // measured (profiles/r03_ab_wgrad_pp_layers.log): ahead by 3-4 % from 256 channels on, behind by 2-3 % on the 64- and 128-channel
    // layers, whose blocks share no operand tiles through L2
    if ((long)(a.Cin / 64) * (a.Cout / 64) < 16) return false;
    return (long)a.N * cdiv(a.Hb, 8) * cdiv(a.Wb, TW) >= 8;
}

// split-K plan of the halo kernel: slabs == ksplit
int wgrad_halo_plan(const WgradArgs& a, int* ksplit, int* tiles_per) {
    const int ttotal = a.N * cdiv(a.Hb, halo3_rows(a)) * cdiv(a.Wb, TW);
    const long pairs = (long)(a.Cin / 64) * (a.Cout / 64);
    const bool pp = halo_pp_ok(a);
    // one resident wave of blocks (2 per CU; the two-group kernel: 1 per CU): every extra block costs a 147 KB f32 slab
    // written and read back -- at 1024 blocks the slab traffic exceeded the layer's own input bytes (measured 527 -> 641
    // TFLOP/s at 512)
    const int resident = pp ? 256 : 512, min_tiles = pp ? 8 : 4;
    long ks = (resident + pairs - 1) / pairs;
    if (ks > ttotal / min_tiles) ks = ttotal / min_tiles;          // at least four tiles per wave group
    if (ks < 1) ks = 1;
    const int per = cdiv(ttotal, ks);
    *tiles_per = per; *ksplit = cdiv(ttotal, per);
    return 0;
}

int wgrad_halo_launch_bf16(const WgradArgs& a, int ksplit, int tiles_per, hipStream_t st) {
    dim3 grid((a.Cin / 64) * (a.Cout / 64), ksplit), block(256);
    set_last_wgrad_variant(0x48000000 | (ksplit & 0xfff));
    if (a.src[0].pool) {
        hipLaunchKernelGGL(wgrad_halo_bf16_kernel<true>, grid, block, 2 * ATILE + 2 * DTILE, st, a, a.Cout / 64, cdiv(a.Wb, TW), cdiv(a.Hb, TH), tiles_per);
    } else {
        WgradArgs b = a;
        b.ksplit = ksplit;
        const bool buf = halo_buf_ok(a);
        if (halo_pp_ok(a)) {
            constexpr int lds = 4 * H3<8>::STAGE;
            set_last_wgrad_variant(0x48200000 | (ksplit & 0xfff));
            unsigned long long* dbg = nullptr;
            USTRUN_TRY(debug_buffer_for((long)grid.x * ksplit, "wgrad_halo_pp", &dbg));
            if (dbg) {                     // ustrun_debug_buffer set: the stamped build
                if (int rc = ensure_dynamic_lds((const void*)wgrad_halo_pp_bf16_kernel<true>, lds, "wgrad_halo")) return rc;
                hipLaunchKernelGGL(wgrad_halo_pp_bf16_kernel<true>, dim3(grid.x * ksplit), dim3(512), lds, st, b, a.Cout / 64, cdiv(a.Wb, TW),
                                   cdiv(a.Hb, 8), tiles_per, dbg);
            } else {
                if (int rc = ensure_dynamic_lds((const void*)wgrad_halo_pp_bf16_kernel<false>, lds, "wgrad_halo")) return rc;
                hipLaunchKernelGGL(wgrad_halo_pp_bf16_kernel<false>, dim3(grid.x * ksplit), dim3(512), lds, st, b, a.Cout / 64, cdiv(a.Wb, TW),
                                   cdiv(a.Hb, 8), tiles_per, (unsigned long long*)nullptr);
            }
            USTRUN_LAUNCH_CHECK("wgrad_halo_bf16");
            return 0;
        }
        if (buf) set_last_wgrad_variant(0x48100000 | (ksplit & 0xfff));
        if (halo3_rows(a) == 8) {
            if (buf && (g_debug_flags & 128))
                if (int rc = ensure_dynamic_lds((const void*)wgrad_halo4_bf16_kernel<8>, H3<8>::NBUF * H3<8>::STAGE + 16384, "wgrad_halo")) return rc;
            if (buf)
                hipLaunchKernelGGL(wgrad_halo4_bf16_kernel<8>, dim3(grid.x * ksplit), block, H3<8>::NBUF * H3<8>::STAGE + ((g_debug_flags & 128) ? 16384 : 0), st, b, a.Cout / 64,
                                   cdiv(a.Wb, TW), cdiv(a.Hb, 8), tiles_per);
            else
                hipLaunchKernelGGL(wgrad_halo3_bf16_kernel<8>, dim3(grid.x * ksplit), block, H3<8>::NBUF * H3<8>::STAGE, st, b, a.Cout / 64,
                                   cdiv(a.Wb, TW), cdiv(a.Hb, 8), tiles_per);
        } else {
            if (buf)
                hipLaunchKernelGGL(wgrad_halo4_bf16_kernel<4>, dim3(grid.x * ksplit), block, H3<4>::NBUF * H3<4>::STAGE, st, b, a.Cout / 64,
                                   cdiv(a.Wb, TW), cdiv(a.Hb, 4), tiles_per);
            else
                hipLaunchKernelGGL(wgrad_halo3_bf16_kernel<4>, dim3(grid.x * ksplit), block, H3<4>::NBUF * H3<4>::STAGE, st, b, a.Cout / 64,
                                   cdiv(a.Wb, TW), cdiv(a.Hb, 4), tiles_per);
        }
    }
    USTRUN_LAUNCH_CHECK("wgrad_halo_bf16");
    return 0;
}

}  // namespace ustrun
