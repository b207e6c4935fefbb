// conv_ws64_bf16.hip -- weight-stationary, row-streaming 3x3 convolution for the 64 -> 64 channel layers at full
// resolution (forward of inc.conv2 / up4.conv2, unet_parts.py:16-21, and their input gradients): the layers the
// north_star's HBM-roofline target is defined on.  bf16 operands, f32 accumulate.
//
// Why a second kernel.  With K = 9 * 64 = 576 the halo-tiled kernel (conv_halo_bf16.hip) spends 15 us per 256-pixel tile
// on 2.7 us of MFMA work: a full prologue (patch + first weights), ten barrier-separated stages that each DMA a weight
// tile, and a transposing epilogue, per tile.  These layers sit at the ridge of the roofline (288 flop/B), so they need
// BOTH the matrix pipe and HBM busy all the time.  Here:
//   * the whole 3x3x64x64 weight set lives in REGISTERS for the lifetime of a block: each wave owns 32 output channels
//     = 36 A fragments (144 registers of the 512 a wave has at one wave per SIMD); no weight traffic, no weight LDS reads;
//   * a block walks a 32-pixel-wide strip of an image downwards, 8 output rows per step, and keeps the activated input
//     rows in an LDS ring (3 banks x 8 rows x 34 px x 144 B): every input row is fetched from HBM once (plus the two halo
//     columns: 34/32), there is no per-tile prologue, and a new item's first rows are staged under the previous item's
//     last steps;
//   * the three taps of a kernel column share their input fragments: 6 LDS reads feed 12 MFMAs (0.5 reads per MFMA);
//   * rows are fetched into REGISTERS one iteration (>= 2 us) ahead, transformed (BatchNorm affine + ReLU, f32) and
//     written to the ring under the MFMAs of the next iteration;
//   * the product is formed as D[channel][pixel] (weights are the A operand): a lane then holds 4 consecutive channels of
//     one pixel per register quad, so the epilogue is pack + ds_write_b64 into a per-wave scratch, and every store
//     instruction writes 16 pixels x 64 contiguous bytes; BatchNorm statistics are taken from the rounded values on their
//     way out (fixed order, one partial row per wave and item).
#include "common.h"
#include "loader.h"
#include <type_traits>

namespace ustrun {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int TW = 32, HW = TW + 2;
constexpr int PITCH = 144;                     // ring pixel: 64 channels + 16 B pad (conflict-free ds_read_b128 at any tap)
constexpr int ROWB = HW * PITCH;               // 4896
constexpr int BANKB = 8 * ROWB;                // 39168: one group of 8 input rows
constexpr int RINGB = 3 * BANKB;               // 117504
constexpr int EPITCH = 80;                     // epilogue scratch: 32 channels of one pixel + 16 B pad
constexpr int EWAVE = 128 * EPITCH;            // 10240 per wave (4 rows x 32 px)
constexpr int DUMMYB = 2048;                   // where threads 128..255 "write" the ninth staging item (2176 = 8.5 x 256)
constexpr int LDSB = RINGB + 4 * EWAVE + DUMMYB;   // 160512 of 163840
constexpr int GITEMS = 8 * HW * 8;             // 16-byte items of a group: 8 rows x 34 px x 8 channel octets = 2176
constexpr int NR = (GITEMS + 255) / 256;       // staging rounds per group: 9 (the last one: waves 0 and 1 only)

__device__ __attribute__((aligned(16))) const unsigned g_zero16w[4] = {0u, 0u, 0u, 0u};

struct WsPlan { int sx, sy, seg, items, ipb; unsigned long long* dbg; };

template <int V> using ic = std::integral_constant<int, V>;

struct Cur {            // one group of 8 input rows of one item (or nothing)
    int valid, item, img, x0, ybeg, S, k;
};

// One iteration u of a block's group sequence (groups = 8 input rows of an item, S + 1 per item):
//   load   group u      -> registers (buffer loads: out-of-image items read as zero without touching memory)
//   write  group u - 1  -> ring bank (u - 1) % 3, transformed, under the MFMAs
//   step   of group u - 2 (if it is not the first of its item): reads banks (u - 3) % 3 and (u - 2) % 3
// FAST = the steady state (all three present, interior tile): one straight-line block the scheduler can interleave;
// everything else (item boundaries, ragged edges, the block's last two iterations) takes the general body.
// DIAG: a development build that stamps the phases of the steady-state iteration with s_memtime and adds the differences
// per wave into p.dbg (read the SHARES, not the run time: the stamps fence the schedule); never launched by the product.
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

template <bool XF, bool STAT, bool DIAG = false>
__global__ __launch_bounds__(256, 1) void conv3x3_ws64_kernel(const IgemmArgs a, const WsPlan p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    char* Ew = smem + RINGB + wave * EWAVE;
    const SrcDev S = a.src[0];
    const __bf16* srcp = (const __bf16*)S.ptr;
    const int H = a.Hb, W = a.Wb;
    const int sH = (int)S.sH, sW = (int)S.sW;
    const unsigned img_bytes = (unsigned)(S.sN * 2);

    // ---- the block's weights: A fragments of v_mfma_f32_32x32x16_bf16, row = output channel 32 wn + l31, k = 8 lh + j
    // of the 16-channel step ks; packed layout [tap][Cin/8][Cout][8] -> one 16-byte load each.  The input gradient
    // walks the taps backwards (a.dstep < 0) over the [tap][Cout/8][Cin][8] pack. ----
    bf16x8 Wr[9][4];
    {
        const __bf16* Wp = (const __bf16*)a.W;
        const bool wflip = a.dstep < 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int wt = wflip ? 8 - tap : tap;
                Wr[tap][ks] = *(const bf16x8*)(Wp + (((long)wt * 8 + 2 * ks + lh) * 64 + 32 * wn + l31) * 8);
            }
        // a use in front of the loop: the compiler otherwise keeps these loads "pending" at the loop head and drains
        // vmcnt(0) -- the row fetches in flight -- before the first MFMA of every iteration
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+a"(Wr[tap][ks]));      // ("a": the fragments live in the accumulator file,
                                                                                      //  which MFMA reads directly; the 256 arch VGPRs stay free)
    }

    // ---- staging geometry (the same for every group): item q = tid + 256 i of the group's [8 rows][34 px][8 octets] ----
    int goffb[NR];         // byte offset from the group's first pixel
    int rp[NR];            // row << 8 | pixel
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int q = tid + 256 * i, hp = q >> 3, r = hp / HW, px = hp - r * HW;
        goffb[i] = (r * sH + px * sW + (tid & 7) * 8) * 2;
        rp[i] = r << 8 | px;
    }
    const int loff0 = (tid >> 3) * PITCH + (tid & 7) * 16;      // ring offset of item i: loff0 + i * 32 * PITCH
    const int afrag0 = l31 * PITCH + lh * 16;

    // ---- cursors over the block's sequence of groups: items [it0, it1), S + 1 groups each ----
    const int it0 = blockIdx.x * p.ipb, it1 = min(it0 + p.ipb, p.items);
    auto decode = [&](int item, int k) __attribute__((always_inline)) {
        Cur c;
        c.valid = item < it1; c.item = item; c.k = k;
        const int per = p.sx * p.sy;
        c.img = item / per;
        const int rem = item - c.img * per;
        const int ys = rem / p.sx;
        c.x0 = (rem - ys * p.sx) * TW;
        c.ybeg = ys * p.seg;
        const int rows = min(p.seg, H - c.ybeg);
        c.S = (rows + 7) >> 3;
        return c;
    };
    auto advance = [&](const Cur& c) __attribute__((always_inline)) {
        if (!c.valid) return c;
        if (c.k < c.S) { Cur n = c; n.k = c.k + 1; return n; }
        return decode(c.item + 1, 0);
    };
    Cur cl = decode(it0, 0), cw, cc;
    cw.valid = cc.valid = 0; cw.item = cc.item = 0; cw.img = cc.img = 0; cw.x0 = cc.x0 = 0; cw.ybeg = cc.ybeg = 0;
    cw.S = cc.S = 0; cw.k = cc.k = 0;

    // ONE register set: item i of the group fetched last iteration is transformed and written to the ring in the second half
    // of this iteration, and the same registers are refilled at once with item i of the next group -- every fetch gets exactly
    // one iteration of flight
    u32x4 stg[NR];
    unsigned okmW = 0;             // in-image mask of the items held in stg
#pragma unroll
    for (int i = 0; i < NR; ++i) stg[i] = (u32x4){0u, 0u, 0u, 0u};
    f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
    const float a_floor = S.relu ? 0.f : -__builtin_inff();
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

    // byte offsets of group c's items inside its image (0x80000000 = beyond num_records: the buffer load returns zero without
    // touching memory) and their in-image mask.  Computed in the shadow of the first half's MFMAs.
    unsigned offL[NR], okmL = 0;
    auto offsets_one = [&](const Cur& c, int i) __attribute__((always_inline)) {
        const int y0g = c.ybeg - 1 + 8 * c.k;
        const int nrows = !c.valid ? 0 : (c.k == c.S ? 2 : 8);  // the item's last group: only its two halo rows are read
        const int gbase = (y0g * sH + (c.x0 - 1) * sW) * 2;
        const int r = rp[i] >> 8, px = rp[i] & 255;
        const unsigned ok = (unsigned)(i < NR - 1 || tid < 128) & (unsigned)(r < nrows) & (unsigned)((unsigned)(y0g + r) < (unsigned)H) &
                            (unsigned)((unsigned)(c.x0 - 1 + px) < (unsigned)W);
        offL[i] = (unsigned)(gbase + goffb[i]) | ((ok ^ 1u) << 31);      // (no select: hipcc turns it into an exec-masked branch)
        okmL = (okmL & ~(1u << i)) | (ok << i);
    };
    auto fetch_one = [&](const __amdgpu_buffer_rsrc_t rs, int i) __attribute__((always_inline)) {          // offL describes the group
        stg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, offL[i], 0, 0);
    };
    // BatchNorm affine + ReLU in f32, back to bf16, zero padding applied after the activation
    auto xform = [&](u32x4 raw, bool ok) __attribute__((always_inline)) {
        if constexpr (!XF) return raw;
        const bf16x8 v = __builtin_bit_cast(bf16x8, raw);
        f32x4 lo = (f32x4){(float)v[0], (float)v[1], (float)v[2], (float)v[3]} * sc0 + sh0;
        f32x4 hi = (f32x4){(float)v[4], (float)v[5], (float)v[6], (float)v[7]} * sc1 + sh1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lo[e] = __builtin_amdgcn_fmed3f(lo[e], a_floor, __builtin_inff());
            hi[e] = __builtin_amdgcn_fmed3f(hi[e], a_floor, __builtin_inff());
        }
        bf16x8 h;
        h[0] = (__bf16)lo[0]; h[1] = (__bf16)lo[1]; h[2] = (__bf16)lo[2]; h[3] = (__bf16)lo[3];
        h[4] = (__bf16)hi[0]; h[5] = (__bf16)hi[1]; h[6] = (__bf16)hi[2]; h[7] = (__bf16)hi[3];
        u32x4 u = __builtin_bit_cast(u32x4, h);
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = ok ? u[e] : 0u;
        return u;
    };
    auto load_consts = [&](const Cur& c) __attribute__((always_inline)) {      // the pass constants of c's image, channel octet tid & 7
        if constexpr (XF) {
            if (S.scale) {
                const long go = S.gN > 0 ? (long)(c.img / S.gN) * S.gstride : 0;
                const float* scp = S.scale + go + 8 * (tid & 7);
                const float* shp = S.shift + go + 8 * (tid & 7);
                sc0 = *(const f32x4*)scp; sc1 = *(const f32x4*)(scp + 4);
                sh0 = *(const f32x4*)shp; sh1 = *(const f32x4*)(shp + 4);
            }
        }
    };

    // ---- the step in two halves of the wave's four rows (HALF 0: rows 0, 1 from patch rows 0..3; HALF 1: rows 2, 3 from
    // patch rows 2..5): the epilogue of one half runs under the MFMAs of the other, across the iteration boundary for
    // half 1.  acc persists across iterations. ----
    f32x16 acc[4];
    int rowaddr[6];
    // fragment rows q = 0..3 of a half for kernel column dx, channel step ks (constants at every call site) ...
    auto frag_read = [&](auto half_c, int dx, int ks, bf16x8* pf) __attribute__((always_inline)) {
        constexpr int HF = decltype(half_c)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) pf[q] = *(const bf16x8*)(ring + rowaddr[2 * HF + q] + dx * PITCH + ks * 32);
    };
    // ... and the 6 MFMAs they feed: rows i = 0, 1 of the half x the three taps of the column
    auto mma6 = [&](auto half_c, int dx, int ks, const bf16x8* pf) __attribute__((always_inline)) {
        constexpr int HF = decltype(half_c)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
                acc[2 * HF + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wr[dy * 3 + dx][ks], pf[i + dy], acc[2 * HF + i], 0, 0, 0);
    };
    auto mma_group = [&](auto half_c, int dx, int ks) __attribute__((always_inline)) {
        bf16x8 pf[4];
        frag_read(half_c, dx, ks, pf);
        mma6(half_c, dx, ks, pf);
    };
    auto zero_half = [&](auto half_c) __attribute__((always_inline)) {
        constexpr int HF = decltype(half_c)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[2 * HF + i][r] = 0.f;
    };
    // epilogue pieces of a half: A = accumulators -> bf16 -> the wave's scratch (lane = pixel, 4 consecutive channels per
    // register quad); B(t) = 16 px x 64 B of one row per store instruction + statistics of the stored values.  LDS executes
    // a wave's instructions in order, so the scratch needs no barrier between A and B.
    auto epi_A = [&](auto half_c) __attribute__((always_inline)) {
        constexpr int HF = decltype(half_c)::value;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * HF + ii;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 h;
                h[0] = (__bf16)acc[i][4 * g]; h[1] = (__bf16)acc[i][4 * g + 1];
                h[2] = (__bf16)acc[i][4 * g + 2]; h[3] = (__bf16)acc[i][4 * g + 3];
                *(bf16x4*)(Ew + (i * 32 + l31) * EPITCH + (8 * g + 4 * lh) * 2) = h;
            }
        }
    };
    auto epi_B = [&](auto half_c, int tt, const Cur& c, auto full_c) __attribute__((always_inline)) {     // tt = 0..3: row 2 HF + tt / 2, pixel half tt & 1
        constexpr int HF = decltype(half_c)::value;
        constexpr bool FULL = decltype(full_c)::value;
        const int yo = c.ybeg + 8 * (c.k - 1) + 4 * wm;           // first output row of this wave
        const int ylim = min(c.ybeg + p.seg, H);
        const int pp = lane >> 2, o = lane & 3;
        __bf16* outp = (__bf16*)a.out0 + (((long)c.img * H + yo) * W + c.x0) * 64 + 32 * wn + 8 * o;
        const int i = 2 * HF + (tt >> 1), px = 16 * (tt & 1) + pp;
        const bf16x8 v = *(const bf16x8*)(Ew + (i * 32 + px) * EPITCH + o * 16);
        const bool inimg = FULL || (yo + i < ylim && c.x0 + px < W);
        if (inimg) *(bf16x8*)(outp + ((long)i * W + px) * 64) = v;
        if constexpr (STAT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = inimg ? (float)v[e] : 0.f;           // statistics see the stored values
                s1[e] += f;
                s2[e] += f * f;
            }
        }
    };
    auto stat_flush = [&](const Cur& c) __attribute__((always_inline)) {       // after the item's last step: one partial row per (item, wm), this wave's 32 channels
        if constexpr (STAT) {
            if (c.k == c.S) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#pragma unroll
                    for (int d = 4; d < 64; d <<= 1) {
                        s1[e] += __shfl_xor(s1[e], d);
                        s2[e] += __shfl_xor(s2[e], d);
                    }
                }
                if (lane < 4) {
                    float* row = a.stat + ((long)(c.item * 2 + wm) * 2) * 64 + 32 * wn + 8 * lane;
                    *(f32x4*)row = (f32x4){s1[0], s1[1], s1[2], s1[3]};
                    *(f32x4*)(row + 4) = (f32x4){s1[4], s1[5], s1[6], s1[7]};
                    *(f32x4*)(row + 64) = (f32x4){s2[0], s2[1], s2[2], s2[3]};
                    *(f32x4*)(row + 68) = (f32x4){s2[4], s2[5], s2[6], s2[7]};
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
            }
        }
    };
    Cur cp = cw;               // the step whose second half's epilogue is still owed (pend)
    bool pend = false;
    auto drain = [&]() __attribute__((always_inline)) {       // the owed half, not overlapped (item boundaries, the block's end)
        if (pend) {
            epi_A(ic<1>{});
#pragma unroll
            for (int t = 0; t < 4; ++t) epi_B(ic<1>{}, t, cp, std::false_type{});
            stat_flush(cp);
            pend = false;
        }
    };

    unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, dt0 = 0, dt1 = 0;
    int m = 0;             // iteration u mod 3: the step reads banks m and m + 1, the write goes to bank m + 2
    auto iteration = [&]() __attribute__((always_inline)) {
        const int bA = m, bB = m == 2 ? 0 : m + 1, bW = m == 0 ? 2 : m - 1;
        const bool do_mma = cc.valid && cc.k >= 1;
        // the step's fragment rows: patch rows 4 wm .. 4 wm + 5 of the 10 (8 in bank A, 2 in bank B)
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int r = 4 * wm + q;
            const int slot = r < 8 ? bA * 8 + r : bB * 8 + (r - 8);
            rowaddr[q] = slot * ROWB + afrag0;
        }
        char* wdst = ring + bW * BANKB + loff0;
        const Cur cn = advance(cl);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(srcp + (long)cl.img * S.sN), 0, (int)img_bytes, 0x00020000);
        const bool fast = do_mma && cw.valid && cl.valid && cc.x0 + TW <= W && cc.ybeg + 8 * cc.k <= min(cc.ybeg + p.seg, H) &&
                          (!pend || (cp.x0 + TW <= W && cp.ybeg + 8 * cp.k <= min(cp.ybeg + p.seg, H)));
        if (fast) {
            // ---- steady state: two straight-line blocks (one per half), each 12 groups of 4 fragment reads + 6 MFMAs with the
            // reads of group g + 1 issued in front of the MFMAs of group g and everything else threaded between the groups ----
            auto body = [&](auto pend_c) __attribute__((always_inline)) {
                constexpr bool PEND = decltype(pend_c)::value;
                bf16x8 pfa[4], pfb[4];
                if constexpr (DIAG) dt0 = stamp();
                zero_half(ic<0>{});
                // half 0 (rows 0, 1)  ||  the owed epilogue of the previous step's half 1, this iteration's fetch offsets
                frag_read(ic<0>{}, 0, 0, pfa);
                frag_read(ic<0>{}, 0, 1, pfb); mma6(ic<0>{}, 0, 0, pfa); if constexpr (PEND) epi_A(ic<1>{});
                frag_read(ic<0>{}, 0, 2, pfa); mma6(ic<0>{}, 0, 1, pfb); offsets_one(cl, 0);
                frag_read(ic<0>{}, 0, 3, pfb); mma6(ic<0>{}, 0, 2, pfa); offsets_one(cl, 1);
                frag_read(ic<0>{}, 1, 0, pfa); mma6(ic<0>{}, 0, 3, pfb); if constexpr (PEND) epi_B(ic<1>{}, 0, cp, std::true_type{});
                frag_read(ic<0>{}, 1, 1, pfb); mma6(ic<0>{}, 1, 0, pfa); offsets_one(cl, 2);
                frag_read(ic<0>{}, 1, 2, pfa); mma6(ic<0>{}, 1, 1, pfb); if constexpr (PEND) epi_B(ic<1>{}, 1, cp, std::true_type{});
                frag_read(ic<0>{}, 1, 3, pfb); mma6(ic<0>{}, 1, 2, pfa); offsets_one(cl, 3);
                frag_read(ic<0>{}, 2, 0, pfa); mma6(ic<0>{}, 1, 3, pfb); if constexpr (PEND) epi_B(ic<1>{}, 2, cp, std::true_type{});
                frag_read(ic<0>{}, 2, 1, pfb); mma6(ic<0>{}, 2, 0, pfa); offsets_one(cl, 4); offsets_one(cl, 5);
                frag_read(ic<0>{}, 2, 2, pfa); mma6(ic<0>{}, 2, 1, pfb); if constexpr (PEND) epi_B(ic<1>{}, 3, cp, std::true_type{});
                frag_read(ic<0>{}, 2, 3, pfb); mma6(ic<0>{}, 2, 2, pfa); offsets_one(cl, 6); offsets_one(cl, 7);
                mma6(ic<0>{}, 2, 3, pfb); offsets_one(cl, 8);
                if constexpr (PEND) stat_flush(cp);           // (between the halves: half 0 of THIS step adds to the sums next)
                if constexpr (DIAG) { dt1 = stamp(); dsum[1] += dt1 - dt0; dt0 = dt1; }
                zero_half(ic<1>{});
                // half 1 (rows 2, 3)  ||  the epilogue of half 0; item i: transform + ring write of the previous group's, then the
                // same registers take this group's
                char* wd8 = tid < 128 ? wdst + 8 * (32 * PITCH) : smem + RINGB + 4 * EWAVE + (tid - 128) * 16;
                auto stage = [&](int i) __attribute__((always_inline)) {
                    *(u32x4*)(i < NR - 1 ? wdst + i * (32 * PITCH) : wd8) = xform(stg[i], (okmW >> i) & 1u);
                    fetch_one(rs, i);
                };
                frag_read(ic<1>{}, 0, 0, pfa);
                frag_read(ic<1>{}, 0, 1, pfb); mma6(ic<1>{}, 0, 0, pfa); epi_A(ic<0>{});
                frag_read(ic<1>{}, 0, 2, pfa); mma6(ic<1>{}, 0, 1, pfb); stage(0);
                frag_read(ic<1>{}, 0, 3, pfb); mma6(ic<1>{}, 0, 2, pfa); stage(1);
                frag_read(ic<1>{}, 1, 0, pfa); mma6(ic<1>{}, 0, 3, pfb); epi_B(ic<0>{}, 0, cc, std::true_type{});
                frag_read(ic<1>{}, 1, 1, pfb); mma6(ic<1>{}, 1, 0, pfa); stage(2);
                frag_read(ic<1>{}, 1, 2, pfa); mma6(ic<1>{}, 1, 1, pfb); epi_B(ic<0>{}, 1, cc, std::true_type{}); stage(3);
                frag_read(ic<1>{}, 1, 3, pfb); mma6(ic<1>{}, 1, 2, pfa); stage(4);
                frag_read(ic<1>{}, 2, 0, pfa); mma6(ic<1>{}, 1, 3, pfb); epi_B(ic<0>{}, 2, cc, std::true_type{}); stage(5);
                frag_read(ic<1>{}, 2, 1, pfb); mma6(ic<1>{}, 2, 0, pfa); stage(6);
                frag_read(ic<1>{}, 2, 2, pfa); mma6(ic<1>{}, 2, 1, pfb); epi_B(ic<0>{}, 3, cc, std::true_type{}); stage(7);
                frag_read(ic<1>{}, 2, 3, pfb); mma6(ic<1>{}, 2, 2, pfa); stage(8);
                mma6(ic<1>{}, 2, 3, pfb);
                okmW = okmL;
                load_consts(cl);          // for the group fetched above (queued behind its rows; used next iteration)
                if constexpr (DIAG) { dt1 = stamp(); dsum[2] += dt1 - dt0; dt0 = dt1; dsum[5] += 1; }
            };
            if (pend) body(std::true_type{}); else body(std::false_type{});
            cp = cc; pend = true;
        } else {
            // ---- general body: item boundaries (no step), ragged edges, the block's last iterations ----
            drain();
            if (cw.valid) {
#pragma unroll
                for (int i = 0; i < NR; ++i)
                    if (i < NR - 1 || wave < 2) *(u32x4*)(wdst + i * (32 * PITCH)) = xform(stg[i], (okmW >> i) & 1u);
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) offsets_one(cl, i);
#pragma unroll
            for (int i = 0; i < NR; ++i) fetch_one(rs, i);
            okmW = okmL;
            if (cl.valid) load_consts(cl);
            if (do_mma) {
                zero_half(ic<0>{}); zero_half(ic<1>{});
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) { mma_group(ic<0>{}, dx, ks); mma_group(ic<1>{}, dx, ks); }
                epi_A(ic<0>{});
#pragma unroll
                for (int t = 0; t < 4; ++t) epi_B(ic<0>{}, t, cc, std::false_type{});
                cp = cc; pend = true;
                drain();
            }
        }
        // the ring bank written above is read from the next iteration on; the scratch is private to the wave
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (DIAG) { if (fast) { dt1 = stamp(); dsum[4] += dt1 - dt0; } }
        cc = cw; cw = cl; cl = cn;
        m = m == 2 ? 0 : m + 1;
    };
    load_consts(cl);
    while (cl.valid || cw.valid || cc.valid) iteration();
    drain();
    if constexpr (DIAG) {
        if (lane == 0 && p.dbg) {
#pragma unroll
            for (int k = 0; k < 6; ++k) p.dbg[((long)blockIdx.x * 4 + wave) * 8 + k] = dsum[k];
        }
    }
}

// segments per strip: whole waves of blocks over the 256 CUs, few bubbles (one staging-only iteration per item)
WsPlan ws_plan(const IgemmArgs& a) {
    WsPlan p;
    p.sx = cdiv(a.Wb, TW);
    const int steps = cdiv(a.Hb, 8);
    double best = 1e30;
    p.sy = 1;
    for (int sy = 1; sy <= steps; ++sy) {
        const int per = cdiv(steps, sy);
        if (cdiv(steps, per) != sy) continue;                 // (no empty segments)
        const long items = (long)a.N * p.sx * sy;
        const long ipb = (items + 255) / 256;
        const double cost = (double)ipb * (per + 1.5);
        if (cost < best - 1e-9) { best = cost; p.sy = sy; }
    }
    p.seg = cdiv(steps, p.sy) * 8;
    p.items = a.N * p.sx * p.sy;
    p.ipb = (p.items + 255) / 256;
    p.dbg = nullptr;
    return p;
}

}  // namespace

void* g_ws64_dbg = nullptr;       // ustrun_debug_buffer: when set, the DIAG build runs and writes [block][wave][8] u64 there
void ws64_set_debug_buffer(void* p) { g_ws64_dbg = p; }

bool ws64_supported(const IgemmArgs& a) {
    if (a.nseg != 9 || a.nz != 1 || a.s_in != 1 || a.s_out != 1 || a.segw != 3 || a.nsrc != 1) return false;
    if (a.Cin != 64 || a.Cout != 64 || a.C0 != 64 || a.out_esz != 2 || a.bias) return false;
    const SrcDev& s = a.src[0];
    if (s.sC != 1 || s.esz != 2 || s.pool || s.off_y || s.off_x || s.H != a.Hb || s.W != a.Wb || s.sW != 64) return false;
    if (a.Ho != a.Hb || a.Wo != a.Wb || a.Wb < 32 || a.Hb < 16) return false;
    const WsPlan p = ws_plan(a);
    return p.items >= 192;              // smaller problems (the batch-1 forward): the tiled kernel fills the chip better
}

int ws64_stat_rows(const IgemmArgs& a) { return ws_plan(a).items * 2; }

int conv3x3_ws64_launch_bf16(const IgemmArgs& a, hipStream_t st) {
    WsPlan p = ws_plan(a);
    p.dbg = (unsigned long long*)g_ws64_dbg;
    const int grid = cdiv(p.items, p.ipb);
    bool xf = false;
    set_last_variant(0x57530000 | ((a.src[0].scale != nullptr || a.src[0].relu != 0) ? 1 : 0));     // 'WS' | XF
    xf |= a.src[0].scale != nullptr || a.src[0].relu != 0;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3x3_ws64_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        (void)hipFuncSetAttribute((const void*)conv3x3_ws64_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        (void)hipFuncSetAttribute((const void*)conv3x3_ws64_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        (void)hipFuncSetAttribute((const void*)conv3x3_ws64_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        attr_done = true;
    }
    const bool stat = a.stat != nullptr;
    if (p.dbg) {
        (void)hipFuncSetAttribute((const void*)conv3x3_ws64_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        (void)hipFuncSetAttribute((const void*)conv3x3_ws64_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        if (xf) hipLaunchKernelGGL((conv3x3_ws64_kernel<true, true, true>), dim3(grid), dim3(256), LDSB, st, a, p);
        else hipLaunchKernelGGL((conv3x3_ws64_kernel<false, false, true>), dim3(grid), dim3(256), LDSB, st, a, p);
        USTRUN_LAUNCH_CHECK("conv3x3_ws64_bf16 (diag)");
        return 0;
    }
    if (xf && stat) hipLaunchKernelGGL((conv3x3_ws64_kernel<true, true>), dim3(grid), dim3(256), LDSB, st, a, p);
    else if (xf) hipLaunchKernelGGL((conv3x3_ws64_kernel<true, false>), dim3(grid), dim3(256), LDSB, st, a, p);
    else if (stat) hipLaunchKernelGGL((conv3x3_ws64_kernel<false, true>), dim3(grid), dim3(256), LDSB, st, a, p);
    else hipLaunchKernelGGL((conv3x3_ws64_kernel<false, false>), dim3(grid), dim3(256), LDSB, st, a, p);
    USTRUN_LAUNCH_CHECK("conv3x3_ws64_bf16");
    return 0;
}

}  // namespace ustrun
