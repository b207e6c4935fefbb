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

typedef __attribute__((ext_vector_type(8))) elt_t bf16x8;
typedef __attribute__((ext_vector_type(4))) elt_t bf16x4;
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

// Uniform plan: every strip is cut into sy segments of seg rows, item = (image, segment, strip), ipb items per block.
// Flat plan (L > 0; the consumer / producer build only): the strips' 8-row steps form ONE sequence of N * sx * steps steps and
// block b takes steps [b L, (b + 1) L) of it, whatever strips they fall in -- equal work per block at ANY image count (81 images of
// 256^2: 648 strips over 256 CUs are 3 rounds of items with the last one half empty, but 81 steps per block exactly).  L >= steps,
// so a strip is cut at most once: item slot = 2 strip + (the piece does not start at the strip's first row).
struct WsPlan { int sx, sy, seg, items, ipb, L, steps; unsigned long long* dbg; };

template <int V> using ic = std::integral_constant<int, V>;

struct Cur {            // one group of 8 input rows of one item (or nothing)
    int valid, item, img, x0, ybeg, S, k;
    int left;           // flat plan: steps of the block's range behind this item
};

// One iteration u of a block's group sequence (groups = 8 input rows of an item, S + 1 per item):
//   fetch  group u      -> registers (buffer loads: out-of-image items read as zero without touching memory)
//   write  group u - 1  -> ring bank (u - 1) % 3, transformed, under the MFMAs
//   step   of group u - 2 (if it is not the first of its item): reads banks (u - 3) % 3 and (u - 2) % 3
// The iteration is ONE straight-line body that always runs in full: what does not apply (no step at an item's first group,
// nothing left to fetch, pixels beyond a ragged edge) is switched off per lane through the buffer instructions' range check
// (an offset with bit 31 set is dropped by the hardware) and through selects -- never through branches.  Alternative bodies
// for the special cases were tried first: the register allocator then gives the loop-carried values (fetches in flight,
// accumulators of the half whose epilogue is owed) different registers per path and the copies at the joins wait for the
// fetches, serialising memory and MFMAs.  The price: an item's first iteration runs its 144 MFMAs on stale data for nothing
// (one in S + 1).
// DIAG: a development build that stamps the two halves and the barrier with s_memtime and adds the differences per wave into
// p.dbg (read the SHARES, not the run time: the stamps fence the schedule); never launched by the product.
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
    const elt_t* srcp = (const elt_t*)S.ptr;
    const int H = a.Hb, W = a.Wb;
    const int sH = (int)S.sH, sW = (int)S.sW;
    // one buffer descriptor per tensor (ws64_supported keeps them under 2 GiB); the image goes into the scalar offset
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)srcp, 0, (int)min((long)a.N * S.sN * 2, 0x7fffffffL), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)a.out0, 0, (int)min((long)a.N * H * W * 128, 0x7fffffffL), 0x00020000);
    const int img_bytes = (int)(S.sN * 2), out_bytes = H * W * 128;

    // ---- the block's weights: A fragments of v_mfma_f32_32x32x16_bf16, row = output channel 32 wn + l31, k = 8 lh + j
    // of the 16-channel step ks; packed layout [tap][Cin/8][Cout][8] -> one 16-byte load each.  The input gradient
    // walks the taps backwards (a.dstep < 0) over the [tap][Cout/8][Cin][8] pack. ----
    bf16x8 Wr[9][4];
    {
        const elt_t* Wp = (const elt_t*)a.W;
        const bool wflip = a.dstep < 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int wt = wflip ? 8 - tap : tap;
                Wr[tap][ks] = *(const bf16x8*)(Wp + (((long)wt * 8 + 2 * ks + lh) * 64 + 32 * wn + l31) * 8);
            }
        // A use in front of the loop: the compiler otherwise keeps these loads "pending" at the loop head and drains vmcnt(0)
        // -- the row fetches in flight -- before the first MFMA of every iteration.  Constraint "a": the fragments live in
        // the accumulator half of the register file, which MFMA reads directly; the 256 arch VGPRs stay free.
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+a"(Wr[tap][ks]));
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
    // the ninth item exists for threads 0..127 only (2176 = 8.5 x 256): the others park theirs in a dummy region
    const int wd8_dummy = RINGB + 4 * EWAVE + (tid & 127) * 16;
    // epilogue: lane (pp, o) stores pixel pp (+16) of a row, channel octet o of the wave's 32 channels
    const int pp = lane >> 2, o = lane & 3;
    const int st_lane = (pp * 64 + 32 * wn + 8 * o) * 2;

    // ---- cursors over the block's sequence of groups: items [it0, it1), S + 1 groups each ----
    const int it0 = blockIdx.x * p.ipb, it1 = min(it0 + p.ipb, p.items);
    auto decode = [&](int item, int k) __attribute__((always_inline)) {
        Cur c;
        c.valid = item < it1; c.k = k;
        item = min(item, it1 - 1);                 // (geometry stays inside the tensor when there is nothing left)
        c.item = item;
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
    Cur cl = decode(it0, 0);
    Cur cw = cl, cc = cl, cp = cl;       // write / step / owed-half cursors, nothing there yet
    cw.valid = cc.valid = cp.valid = 0;

    // ONE register set: item i of the group fetched last iteration is transformed and written to the ring in the second half
    // of this iteration, and the same registers are refilled at once with item i of the next group -- every fetch gets exactly
    // one iteration of flight
    u32x4 stg[NR];
    unsigned okmW = 0;             // in-image mask of the items held in stg
#pragma unroll
    for (int i = 0; i < NR; ++i) stg[i] = (u32x4){0u, 0u, 0u, 0u};
    f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
    const short a_floor16 = S.relu ? (short)0 : (short)0x8000;     // ReLU on the rounded bf16 pairs (act8_bf16)
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

    // byte offsets of group c's items inside its image (bit 31 set = beyond num_records: the buffer load returns zero without
    // touching memory) and their in-image mask; computed in the shadow of the first half's MFMAs
    unsigned offL[NR], okmL = 0;
    unsigned xokm = 0;             // per item: its pixel column lies inside the image (changes with the item's strip only)
    int xok_x0 = -1 << 20;
    auto offsets_one = [&](const Cur& c, int i) __attribute__((always_inline)) {
        const int y0g = c.ybeg - 1 + 8 * c.k;
        const int nrows = !c.valid ? 0 : (c.k == c.S ? 2 : 8);  // the item's last group: only its two halo rows are read
        // rows of the group that exist: r < nrows and 0 <= y0g + r < H  (scalar: one 8-bit mask per group)
        const int lo = max(0, -y0g), hi = min(nrows, H - y0g);
        const unsigned ymask = hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
        const int gbase = (y0g * sH + (c.x0 - 1) * sW) * 2;
        const unsigned ok = (ymask >> (rp[i] >> 8)) & (xokm >> i) & 1u;
        offL[i] = (unsigned)(gbase + goffb[i]) | ((ok ^ 1u) << 31);      // (no select: hipcc turns it into an exec-masked branch)
        okmL = (okmL & ~(1u << i)) | (ok << i);
    };
    auto xok_update = [&](const Cur& c) __attribute__((always_inline)) {       // when the fetch cursor moves to another strip
        if (c.x0 != xok_x0) {
            xok_x0 = c.x0;
            xokm = 0;
#pragma unroll
            for (int i = 0; i < NR; ++i)
                xokm |= ((unsigned)(i < NR - 1 || tid < 128) & (unsigned)((unsigned)(c.x0 - 1 + (rp[i] & 255)) < (unsigned)W)) << i;
        }
    };
    // BatchNorm affine + ReLU in f32, back to bf16, zero padding applied after the activation
    auto xform = [&](u32x4 raw, bool ok) __attribute__((always_inline)) {
        if constexpr (!XF) return raw;
        u32x4 u = act8_bf16(raw, sc0, sc1, sh0, sh1, a_floor16);
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
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
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
        for (int i = 0; i < 2; ++i)                // (alternating the two accumulators instead measured 2-4 % slower)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
                acc[2 * HF + i] = USTRUN_MFMA_32x32x16(Wr[dy * 3 + dx][ks], pf[i + dy], acc[2 * HF + i], 0, 0, 0);
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
                h[0] = (elt_t)acc[i][4 * g]; h[1] = (elt_t)acc[i][4 * g + 1];
                h[2] = (elt_t)acc[i][4 * g + 2]; h[3] = (elt_t)acc[i][4 * g + 3];
                *(bf16x4*)(Ew + (i * 32 + l31) * EPITCH + (8 * g + 4 * lh) * 2) = h;
            }
        }
    };
    // tt = 0..3: row 2 HF + tt / 2 of the wave, pixel half tt & 1; `live`: the step exists (wave-uniform)
    auto epi_B = [&](auto half_c, int tt, const Cur& c, bool live) __attribute__((always_inline)) {
        constexpr int HF = decltype(half_c)::value;
        const int i = 2 * HF + (tt >> 1), px = 16 * (tt & 1) + pp;
        const int y = c.ybeg + 8 * (c.k - 1) + 4 * wm + i;
        const int ylim = min(c.ybeg + p.seg, H);
        u32x4 u = *(const u32x4*)(Ew + (i * 32 + px) * EPITCH + o * 16);
        const bool inimg = live & (y < ylim) & (c.x0 + px < W);
        // (the row offset is added into the VECTOR offset and soffset stays 0: for a 128-bit buffer store with an SGPR soffset hipcc's
        // hazard recognizer inserts no wait state in front of a VALU write of the store's data registers -- the selects right below --
        // and gfx950 then ships the overwritten dword under store pressure: found in conv_first.hip, round 4)
        const unsigned voff = inimg ? (unsigned)(st_lane + (tt & 1) * 2048 + __builtin_amdgcn_readfirstlane(c.img * out_bytes + (y * W + c.x0) * 128)) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(u, ro, voff, 0, 0);
        if constexpr (STAT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = inimg ? u[e] : 0u;
            const bf16x8 v = __builtin_bit_cast(bf16x8, u);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];           // statistics see the stored values
                s1[e] = add_scalar(s1[e], f);
                s2[e] = fma_scalar(f, f, s2[e]);
            }
        }
    };
    auto stat_flush = [&](const Cur& c) __attribute__((always_inline)) {       // one partial row per (item, wm), this wave's 32 channels
        if constexpr (STAT) {
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
    };

    unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, dt0 = 0, dt1 = 0;
    bool pend = false;     // the previous iteration ran a step: the epilogue of its second half is owed (cursor cp)
    load_consts(cl);
    // The iteration, with the ring phase m = u mod 3 a compile-time constant (the loop below is unrolled three times): the step
    // reads banks m and m + 1, the write goes to bank m + 2, and every LDS address is a lane constant plus an immediate.
    // Patch rows 4 wm .. 4 wm + 5 of the step's 10 (8 in bank A, 2 in bank B): wave row wm = 0 reads bank A rows 0..5, wm = 1
    // bank A rows 4..7 and bank B rows 0, 1 -- two lane bases, the row picked by an immediate.
    const int frA = afrag0 + 4 * wm * ROWB;                       // + bank A * BANKB + q * ROWB          (q < 4 or wm == 0)
    const int frB = afrag0 + (4 * wm - 8) * ROWB;                 // + bank B * BANKB + q * ROWB          (q >= 4 and wm == 1)
    const bool hiB = wm == 1;
    auto iteration = [&](auto m_c) __attribute__((always_inline)) {
        constexpr int bA = decltype(m_c)::value, bB = bA == 2 ? 0 : bA + 1, bW = bA == 0 ? 2 : bA - 1;
        const bool live = cc.valid && cc.k >= 1;
        rowaddr[0] = frA + bA * BANKB; rowaddr[1] = rowaddr[0] + ROWB; rowaddr[2] = rowaddr[0] + 2 * ROWB; rowaddr[3] = rowaddr[0] + 3 * ROWB;
        rowaddr[4] = hiB ? frB + bB * BANKB + 4 * ROWB : frA + bA * BANKB + 4 * ROWB;
        rowaddr[5] = rowaddr[4] + ROWB;
        char* wdst = ring + bW * BANKB + loff0;
        char* wd8 = tid < 128 ? wdst + 8 * (32 * PITCH) : smem + wd8_dummy;
        const Cur cn = advance(cl);
        xok_update(cl);
        const int in_soff = __builtin_amdgcn_readfirstlane(cl.img * img_bytes);
        // 24 groups (half, dx, ks) of 4 fragment reads + 6 MFMAs in one stream; the reads run TWO groups ahead of their MFMAs
        // (one group = 192 MFMA cycles is less than an LDS round trip with four waves reading and writing)
        bf16x8 pf[3][4];
        auto stage = [&](int i) __attribute__((always_inline)) {
            *(u32x4*)(i < NR - 1 ? wdst + i * (32 * PITCH) : wd8) = xform(stg[i], (okmW >> i) & 1u);
            stg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, offL[i], in_soff, 0);
        };
        auto read_group = [&](int gg) __attribute__((always_inline)) {       // gg = 0..23, folds to a constant
            if (gg < 12) frag_read(ic<0>{}, gg / 4, gg % 4, pf[gg % 3]);
            else if (gg < 24) frag_read(ic<1>{}, (gg - 12) / 4, (gg - 12) % 4, pf[gg % 3]);
        };
        if constexpr (DIAG) dt0 = stamp();
        zero_half(ic<0>{});
        read_group(0); read_group(1);
        // ---- half 0 (rows 0, 1)  ||  the owed epilogue of the previous step's half 1, this iteration's fetch offsets ----
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            read_group(g + 2);
            mma6(ic<0>{}, g / 4, g % 4, pf[g % 3]);
            if (g == 0) epi_A(ic<1>{});
            if (g == 1) offsets_one(cl, 0);
            if (g == 2) offsets_one(cl, 1);
            if (g == 3) epi_B(ic<1>{}, 0, cp, pend);
            if (g == 4) offsets_one(cl, 2);
            if (g == 5) epi_B(ic<1>{}, 1, cp, pend);
            if (g == 6) offsets_one(cl, 3);
            if (g == 7) epi_B(ic<1>{}, 2, cp, pend);
            if (g == 8) { offsets_one(cl, 4); offsets_one(cl, 5); }
            if (g == 9) epi_B(ic<1>{}, 3, cp, pend);
            if (g == 10) { offsets_one(cl, 6); offsets_one(cl, 7); }
            if (g == 11) offsets_one(cl, 8);
        }
        // between the halves: half 0 of THIS step adds to the sums next
        if (pend && cp.k == cp.S) stat_flush(cp);
        if constexpr (DIAG) { dt1 = stamp(); dsum[1] += dt1 - dt0; dt0 = dt1; }
        zero_half(ic<1>{});
        // ---- half 1 (rows 2, 3)  ||  the epilogue of half 0; item i: transform + ring write of the previous group's, then the
        // same registers take this group's ----
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            read_group(12 + g + 2);
            mma6(ic<1>{}, g / 4, g % 4, pf[(12 + g) % 3]);
            if (g == 0) epi_A(ic<0>{});
            if (g == 1) stage(0);
            if (g == 2) stage(1);
            if (g == 3) epi_B(ic<0>{}, 0, cc, live);
            if (g == 4) stage(2);
            if (g == 5) { epi_B(ic<0>{}, 1, cc, live); stage(3); }
            if (g == 6) stage(4);
            if (g == 7) { epi_B(ic<0>{}, 2, cc, live); stage(5); }
            if (g == 8) stage(6);
            if (g == 9) { epi_B(ic<0>{}, 3, cc, live); stage(7); }
            if (g == 10) stage(8);
        }
        okmW = okmL;
        load_consts(cl);          // for the group fetched above (queued behind its rows; used next iteration)
        if constexpr (DIAG) { dt1 = stamp(); dsum[2] += dt1 - dt0; dt0 = dt1; dsum[5] += 1; }
        // the ring bank written above is read from the next iteration on; the scratch is private to the wave
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (DIAG) { dt1 = stamp(); dsum[4] += dt1 - dt0; }
        cp = cc; pend = live;
        cc = cw; cw = cl; cl = cn;
    };
    while (true) {
        if (!(cl.valid | cw.valid | cc.valid | (int)pend)) break;
        iteration(ic<0>{});
        if (!(cl.valid | cw.valid | cc.valid | (int)pend)) break;
        iteration(ic<1>{});
        if (!(cl.valid | cw.valid | cc.valid | (int)pend)) break;
        iteration(ic<2>{});
    }
    if constexpr (DIAG) {
        if (lane == 0 && p.dbg) {
#pragma unroll
            for (int k = 0; k < 6; ++k) p.dbg[((long)blockIdx.x * 4 + wave) * 8 + k] = dsum[k];
        }
    }
}


// ======================================================================================================================
// Round 3: the same streaming structure on EIGHT waves (two per SIMD) -- conv3x3_ws64x8_kernel.
//
// What limited the four-wave kernel above: one wave per SIMD (its 144 weight registers leave
// no room for a second) issues ~750 instructions per 144 MFMAs, a lone wave pays >= 4 cycles per instruction of any kind, and
// the instructions bunch around the ring writes and stores: the matrix pipe was 62 % busy.  Here each wave owns SIXTEEN output
// channels on v_mfma_f32_16x16x32_bf16 (A = weights: 9 taps x 2 k-steps x 4 registers = 72 per wave), so a 512-thread block
// puts two waves on every SIMD and one wave's address arithmetic, transform, ring write or store runs in the other's MFMA
// shadow.  Wave w: channel group cg = w & 3 (16 channels), row half wm = w >> 2 (4 of the step's 8 output rows); SIMD partners
// (w, w + 4) share cg and differ in wm.  Per step and wave: 8 pixel tiles (4 rows x 2 halves of the 32-px strip) x 9 taps x 2
// k-steps = 144 MFMAs of 16 cycles; the three taps of a kernel column still share their input fragments (6 reads -> 12 MFMAs).
//   * ring pixel pitch 160 B: the 16x16x32 B-operand read (lane l: pixel l & 15, 16-byte channel chunk l >> 4) is
//     conflict-free for ds_read_b128's four lane groups at that pitch (144 B, the pitch of the 32x32x16 kernel, is 2-way);
//   * the product is D[channel][pixel] again; a lane holds 4 consecutive channels of ONE pixel per accumulator, so the
//     epilogue needs no LDS at all: 2 cvt_pk + one 8-byte buffer store per tile (a pixel's 32 bytes from 4 lanes, 16 pixels
//     per instruction; the four channel groups' 32-byte pieces of a 128-byte line meet in L2), statistics from the rounded
//     values in registers (4 channels per lane, 16 lanes per channel reduced by DPP-able shuffles at the item's end);
//   * no deferred half-epilogue: with a partner wave to cover it the step is one run of 12 read/MFMA groups, then the
//     staging of the next group and the stores.  Still ONE straight-line iteration body, everything irregular switched off
//     through the buffer range check or selects.
constexpr int PITCH8 = 160;
constexpr int ROWB8 = HW * PITCH8;               // 5440
constexpr int BANKB8 = 8 * ROWB8;                // 43520
constexpr int RINGB8 = 3 * BANKB8;               // 130560
constexpr int NR8 = (GITEMS + 511) / 512;        // staging rounds per group with 512 threads: 5 (the last one: threads 0..127)
constexpr int DUMMYB8 = 384 * 16;                // where threads 128..511 "write" the fifth staging item
constexpr int LDSB8 = RINGB8 + DUMMYB8;          // 136704

template <bool XF, bool STAT, bool DIAG = false>
__global__ __launch_bounds__(512, 2) void conv3x3_ws64x8_kernel(const IgemmArgs a, const WsPlan p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, wm = wave >> 2;
    const int lp = lane & 15, lq = lane >> 4;
    const SrcDev S = a.src[0];
    const elt_t* srcp = (const elt_t*)S.ptr;
    const int H = a.Hb, W = a.Wb;
    const int sH = (int)S.sH, sW = (int)S.sW;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)srcp, 0, (int)min((long)a.N * S.sN * 2, 0x7fffffffL), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)a.out0, 0, (int)min((long)a.N * H * W * 128, 0x7fffffffL), 0x00020000);
    const int img_bytes = (int)(S.sN * 2), out_bytes = H * W * 128;

    // ---- the wave's weights: A fragments of v_mfma_f32_16x16x32_bf16, row = output channel 16 cg + lp, k = 8 lq + j of the
    // 32-channel step ks; packed layout [tap][Cin/8][Cout][8] -> one 16-byte load each (input gradient: taps backwards over
    // the [tap][Cout/8][Cin][8] pack) ----
    bf16x8 Wr[9][2];
    {
        const elt_t* Wp = (const elt_t*)a.W;
        const bool wflip = a.dstep < 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int wt = wflip ? 8 - tap : tap;
                Wr[tap][ks] = *(const bf16x8*)(Wp + (((long)wt * 8 + 4 * ks + lq) * 64 + 16 * cg + lp) * 8);
            }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(Wr[tap][ks]));       // (loaded before the loop, not pending at its head)
    }

    // ---- staging geometry: item q = tid + 512 i of the group's [8 rows][34 px][8 octets] ----
    int goffb[NR8], rp[NR8];
#pragma unroll
    for (int i = 0; i < NR8; ++i) {
        const int q = tid + 512 * i, hp = q >> 3, r = hp / HW, px = hp - r * HW;
        goffb[i] = (r * sH + px * sW + (tid & 7) * 8) * 2;
        rp[i] = r << 8 | px;
    }
    const int loff0 = (tid >> 3) * PITCH8 + (tid & 7) * 16;      // ring offset of item i: loff0 + i * 64 * PITCH8
    const int afrag0 = lp * PITCH8 + lq * 16;
    const int wd4_dummy = RINGB8 + (tid >= 128 ? (tid - 128) * 16 : 0);
    const int st_lane = lp * 128 + 32 * cg + 8 * lq;             // output byte offset of the lane's 4 channels inside a 16-px run

    const int it0 = blockIdx.x * p.ipb, it1 = min(it0 + p.ipb, p.items);
    auto decode = [&](int item, int k) __attribute__((always_inline)) {
        Cur c;
        c.valid = item < it1; c.k = k;
        item = min(item, it1 - 1);
        c.item = item;
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
    Cur cl = decode(it0, 0);
    Cur cw = cl, cc = cl;                // write / step cursors, nothing there yet
    cw.valid = cc.valid = 0;

    u32x4 stg[NR8];
    unsigned okmW = 0;
#pragma unroll
    for (int i = 0; i < NR8; ++i) stg[i] = (u32x4){0u, 0u, 0u, 0u};
    f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
    const short a_floor16 = S.relu ? (short)0 : (short)0x8000;     // ReLU on the rounded bf16 pairs (act8_bf16)
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};

    unsigned offL[NR8], okmL = 0;
    unsigned xokm = 0;
    int xok_x0 = -(1 << 20);
    auto offsets_one = [&](const Cur& c, int i) __attribute__((always_inline)) {
        const int y0g = c.ybeg - 1 + 8 * c.k;
        const int nrows = !c.valid ? 0 : (c.k == c.S ? 2 : 8);
        const int lo = max(0, -y0g), hi = min(nrows, H - y0g);
        const unsigned ymask = hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
        const int gbase = (y0g * sH + (c.x0 - 1) * sW) * 2;
        const unsigned ok = (ymask >> (rp[i] >> 8)) & (xokm >> i) & 1u;
        offL[i] = (unsigned)(gbase + goffb[i]) | ((ok ^ 1u) << 31);
        okmL = (okmL & ~(1u << i)) | (ok << i);
    };
    auto xok_update = [&](const Cur& c) __attribute__((always_inline)) {
        if (c.x0 != xok_x0) {
            xok_x0 = c.x0;
            xokm = 0;
#pragma unroll
            for (int i = 0; i < NR8; ++i)
                xokm |= ((unsigned)(i < NR8 - 1 || tid < 128) & (unsigned)((unsigned)(c.x0 - 1 + (rp[i] & 255)) < (unsigned)W)) << i;
        }
    };
    auto xform = [&](u32x4 raw, bool ok) __attribute__((always_inline)) {
        if constexpr (!XF) return raw;
        u32x4 u = act8_bf16(raw, sc0, sc1, sh0, sh1, a_floor16);
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = ok ? u[e] : 0u;
        return u;
    };
    auto load_consts = [&](const Cur& c) __attribute__((always_inline)) {
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

    f32x4 acc[4][2];                     // [output row of the wave][pixel half]: 16 channels x 16 pixels each
    int rowaddr[6];
    // fragments of patch rows q = 0..5 for kernel column dx, channel step ks, pixel half ph (constants at every call site) ...
    auto frag_read = [&](int dx, int ks, int ph, bf16x8* pf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 6; ++q) pf[q] = *(const bf16x8*)(ring + rowaddr[q] + (16 * ph + dx) * PITCH8 + ks * 64);
    };
    // ... and the 12 MFMAs they feed: the wave's 4 output rows x the three taps of the column
    auto mma12 = [&](int dx, int ks, int ph, const bf16x8* pf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
                acc[i][ph] = USTRUN_MFMA_16x16x32(Wr[dy * 3 + dx][ks], pf[i + dy], acc[i][ph], 0, 0, 0);
    };
    // epilogue of tile (row i, pixel half ph): 4 channels of one pixel per lane -> 8 bytes; statistics of the stored values
    auto epi = [&](int i, int ph, const Cur& c, bool live) __attribute__((always_inline)) {
        const int y = c.ybeg + 8 * (c.k - 1) + 4 * wm + i;
        const int ylim = min(c.ybeg + p.seg, H);
        bf16x4 h;
        h[0] = (elt_t)acc[i][ph][0]; h[1] = (elt_t)acc[i][ph][1]; h[2] = (elt_t)acc[i][ph][2]; h[3] = (elt_t)acc[i][ph][3];
        u32x2 u = __builtin_bit_cast(u32x2, h);
        const bool inimg = live & (y < ylim) & (c.x0 + 16 * ph + lp < W);
        const unsigned voff = (unsigned)(st_lane + ph * 2048) | (inimg ? 0u : 0x80000000u);
        __builtin_amdgcn_raw_buffer_store_b64(u, ro, voff, __builtin_amdgcn_readfirstlane(c.img * out_bytes + (y * W + c.x0) * 128), 0);
        if constexpr (STAT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f = inimg ? (float)h[e] : 0.f;
                s1[e] = add_scalar(s1[e], f);
                s2[e] = fma_scalar(f, f, s2[e]);
            }
        }
    };
    auto stat_flush = [&](const Cur& c) __attribute__((always_inline)) {       // one partial row per (item, wm), this wave's 16 channels
        if constexpr (STAT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {
                    s1[e] += __shfl_xor(s1[e], d);
                    s2[e] += __shfl_xor(s2[e], d);
                }
            }
            if (lp == 0) {
                float* row = a.stat + ((long)(c.item * 2 + wm) * 2) * 64 + 16 * cg + 4 * lq;
                *(f32x4*)row = (f32x4){s1[0], s1[1], s1[2], s1[3]};
                *(f32x4*)(row + 64) = (f32x4){s2[0], s2[1], s2[2], s2[3]};
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
        }
    };

    unsigned long long dsum[6] = {0, 0, 0, 0, 0, 0}, dt0 = 0, dt1 = 0;
    load_consts(cl);
    const int frA = afrag0 + 4 * wm * ROWB8;                      // + bank A * BANKB8 + q * ROWB8          (q < 4 or wm == 0)
    const int frB = afrag0 + (4 * wm - 8) * ROWB8;                // + bank B * BANKB8 + q * ROWB8          (q >= 4 and wm == 1)
    const bool hiB = wm == 1;
    auto iteration = [&](auto m_c) __attribute__((always_inline)) {
        constexpr int bA = decltype(m_c)::value, bB = bA == 2 ? 0 : bA + 1, bW = bA == 0 ? 2 : bA - 1;
        const bool live = cc.valid && cc.k >= 1;
        rowaddr[0] = frA + bA * BANKB8; rowaddr[1] = rowaddr[0] + ROWB8; rowaddr[2] = rowaddr[0] + 2 * ROWB8; rowaddr[3] = rowaddr[0] + 3 * ROWB8;
        rowaddr[4] = hiB ? frB + bB * BANKB8 + 4 * ROWB8 : frA + bA * BANKB8 + 4 * ROWB8;
        rowaddr[5] = rowaddr[4] + ROWB8;
        char* wdst = ring + bW * BANKB8 + loff0;
        char* wd4 = tid < 128 ? wdst + 4 * (64 * PITCH8) : smem + wd4_dummy;
        const Cur cn = advance(cl);
        xok_update(cl);
        const int in_soff = __builtin_amdgcn_readfirstlane(cl.img * img_bytes);
        bf16x8 pf[2][6];
        auto stage = [&](int i) __attribute__((always_inline)) {
            *(u32x4*)(i < NR8 - 1 ? wdst + i * (64 * PITCH8) : wd4) = xform(stg[i], (okmW >> i) & 1u);
            stg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, offL[i], in_soff, 0);
        };
        // 12 groups (ks, dx, ph) of 6 fragment reads + 12 MFMAs; the reads run one group (192 MFMA cycles) ahead of their MFMAs
        // -- the partner wave covers what that leaves of the LDS latency, and a third fragment set spills at 256 registers
        auto read_group = [&](int gg) __attribute__((always_inline)) {
            if (gg < 12) frag_read((gg >> 1) % 3, gg / 6, gg & 1, pf[gg & 1]);
        };
        if constexpr (DIAG) dt0 = stamp();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) acc[i][ph] = (f32x4){0.f, 0.f, 0.f, 0.f};
        read_group(0);
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            read_group(g + 1);
            mma12((g >> 1) % 3, g / 6, g & 1, pf[g & 1]);
            if (g == 1) offsets_one(cl, 0);
            if (g == 2) offsets_one(cl, 1);
            if (g == 3) offsets_one(cl, 2);
            if (g == 4) offsets_one(cl, 3);
            if (g == 5) offsets_one(cl, 4);
            if (g == 6) stage(0);
            if (g == 7) stage(1);
            if (g == 8) stage(2);
            if (g == 9) stage(3);
            if (g == 10) stage(4);
        }
        okmW = okmL;
        load_consts(cl);
        if constexpr (DIAG) { dt1 = stamp(); dsum[1] += dt1 - dt0; dt0 = dt1; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) epi(i, ph, cc, live);
        if (live && cc.k == cc.S) stat_flush(cc);
        if constexpr (DIAG) { dt1 = stamp(); dsum[2] += dt1 - dt0; dt0 = dt1; dsum[5] += 1; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (DIAG) { dt1 = stamp(); dsum[4] += dt1 - dt0; }
        cc = cw; cw = cl; cl = cn;
    };
    while (true) {
        if (!(cl.valid | cw.valid | cc.valid)) break;
        iteration(ic<0>{});
        if (!(cl.valid | cw.valid | cc.valid)) break;
        iteration(ic<1>{});
        if (!(cl.valid | cw.valid | cc.valid)) break;
        iteration(ic<2>{});
    }
    if constexpr (DIAG) {
        if (lane == 0 && p.dbg) {
#pragma unroll
            for (int k = 0; k < 6; ++k) p.dbg[((long)blockIdx.x * 8 + wave) * 8 + k] = dsum[k];
        }
    }
}


// ======================================================================================================================
// Round 3, second form: CONSUMER and PRODUCER waves -- conv3x3_ws64cp_kernel.
//
// The counters of the two kernels above say the same thing: the matrix pipe is ~53 % busy because each wave issues ~4 other
// instructions per MFMA (fetch offsets, transform, ring writes, stores, statistics, cursors), a SIMD issues roughly one
// instruction per 4 cycles whoever it comes from, and splitting the TILE over more waves multiplies that work.  Here the WORK
// KINDS are split instead (the ring-gemm arrangement of the programming guide): a 512-thread block, waves 0-3 multiply, waves
// 4-7 do everything else; partner waves (w, w + 4) share a SIMD.
//   consumers (one per SIMD): the four-wave kernel's multiply -- 32 output channels x 4 rows x 32 px per step on
//     v_mfma_f32_32x32x16_bf16, the wave's 3x3x64x32 weights in 144 registers -- and nothing else: per half-step 12 groups of
//     4 fragment reads + 6 MFMAs, then 16 cvt_pk + 8 ds_write_b64 that park the half's bf16 results in the wave's scratch.
//     ~2.3 other instructions per MFMA instead of ~4.2.  Accumulators: ONE half-step (32 registers): with the parked results
//     handed over at a barrier nothing has to survive a half-step.
//   producers: the row fetch one iteration ahead (buffer loads, range-checked), BatchNorm affine + ReLU, ring writes, the
//     cursor arithmetic -- and the consumers' epilogue: 16-byte-per-lane stores of the parked rows and the BatchNorm
//     statistics of the stored values (fixed order: half 0 then half 1 of a step, one partial row per (item, wave row)).
//   Two workgroup barriers per step: B1 after half 0 (its parked rows -> producers; they store them during half 1), B2 after
//     half 1 (those rows are stored during the next step's half 0; the ring bank written this step becomes readable).
// Same ring (3 banks x 8 rows x 34 px x 144 B), same per-wave scratch (4 rows x 32 px x 80 B), same work decomposition and
// statistics rows as the four-wave kernel: results are bit-identical to it.
constexpr int LDSBCP = RINGB + 4 * EWAVE + DUMMYB;

// BNS (round 4): the launch is the input gradient that writes da of a BatchNorm + ReLU layer (inc.conv1 / up4.conv1 under their second
// convolution); the producers, which store the parked rows anyway, also form that layer's backward sums sum(da mask) and
// sum(da mask y) in place of the forward statistics (same rows, same flush).  The 16 bytes of y beside each stored piece are
// fetched like the input rows: one buffer load per piece, issued right after the piece of the PREVIOUS step was consumed, i.e. half
// an iteration (~2 us) or more before its use -- eight pieces (32 registers) in flight per producer lane.
template <bool XF, bool STAT, bool DIAG = false, bool BNS = false>
__global__ __launch_bounds__(512, 2) void conv3x3_ws64cp_kernel(const IgemmArgs a, const WsPlan p) {
    static_assert(!BNS || (!XF && !STAT), "BatchNorm-backward sums: plain source, no forward statistics");
    unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dt0 = 0, dt1 = 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave8 >= 4;
    const int wave = wave8 & 3;                      // consumer w parks into scratch w; producer w + 4 stores scratch w
    const int tp = tid & 255;                        // thread index inside the role
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    char* Ew = smem + RINGB + wave * EWAVE;
    const SrcDev S = a.src[0];
    const elt_t* srcp = (const elt_t*)S.ptr;
    const int H = a.Hb, W = a.Wb;
    const int sH = (int)S.sH, sW = (int)S.sW;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)srcp, 0, (int)min((long)a.N * S.sN * 2, 0x7fffffffL), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)a.out0, 0, (int)min((long)a.N * H * W * 128, 0x7fffffffL), 0x00020000);
    const int img_bytes = (int)(S.sN * 2), out_bytes = H * W * 128;

    // ---- cursors over the block's sequence of groups (all eight waves keep them: the trip count must agree) ----
    const bool flat = p.L > 0;
    const int it0 = blockIdx.x * p.ipb, it1 = min(it0 + p.ipb, p.items);
    auto decode = [&](int item, int k) __attribute__((always_inline)) {
        Cur c;
        c.k = k; c.left = 0;
        if (flat) {             // the block's first piece: wherever step blockIdx.x * L falls
            const int total = a.N * p.sx * p.steps;
            const int pos = blockIdx.x * p.L, end = min(pos + p.L, total);
            const int strip = pos / p.steps, st = pos - strip * p.steps;
            c.valid = pos < end;
            c.img = strip / p.sx;
            c.x0 = (strip - c.img * p.sx) * TW;
            c.ybeg = st * 8;
            c.S = max(min(p.steps - st, end - pos), 1);
            c.left = max(end - pos - c.S, 0);
            c.item = strip * 2 + (st != 0 ? 1 : 0);
            return c;
        }
        c.valid = item < it1;
        item = min(item, it1 - 1);
        c.item = item;
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
    // the next group: same item, or the next item by counting (strip, segment, image) up -- no divisions on the producers'
    // path between two barriers (the division-based decode runs once, for the block's first item)
    auto advance = [&](const Cur& c) __attribute__((always_inline)) {
        Cur n = c;
        const bool same = c.k < c.S;
        if (flat) {             // the next piece starts at the next strip's first row and ends with the strip or with the block's range
            const bool has = c.left > 0;
            int x0 = c.x0 + TW, img = c.img;
            const bool wrapx = x0 >= p.sx * TW;
            x0 = wrapx ? 0 : x0;
            img = wrapx ? img + 1 : img;
            const int S = min(p.steps, c.left);
            if (same) n.k = c.k + 1;
            else if (has) { n.item = (c.item | 1) + 1; n.img = img; n.x0 = x0; n.ybeg = 0; n.S = S; n.left = c.left - S; n.k = 0; }
            else { n.valid = 0; n.k = 0; }
            return c.valid ? n : c;
        }
        const int item = min(c.item + 1, it1 - 1);
        const bool has = c.item + 1 < it1;
        int x0 = c.x0 + TW, ybeg = c.ybeg, img = c.img;
        const bool wrapx = x0 >= p.sx * TW;
        x0 = wrapx ? 0 : x0;
        ybeg = wrapx ? ybeg + p.seg : ybeg;
        const bool wrapy = ybeg >= p.sy * p.seg;
        ybeg = wrapy ? 0 : ybeg;
        img = wrapy ? img + 1 : img;
        const int rows = min(p.seg, H - ybeg);
        if (same) n.k = c.k + 1;
        else if (has) { n.item = item; n.img = img; n.x0 = x0; n.ybeg = ybeg; n.S = (rows + 7) >> 3; n.k = 0; }
        else { n.valid = 0; n.k = 0; }
        return c.valid ? n : c;
    };
    Cur cl = decode(it0, 0);
    Cur cw = cl, cc = cl, cp = cl;
    cw.valid = cc.valid = cp.valid = 0;
    bool pend = false;

    const int frA = l31 * PITCH + lh * 16 + 4 * wm * ROWB;
    const int frB = l31 * PITCH + lh * 16 + (4 * wm - 8) * ROWB;
    const bool hiB = wm == 1;

    if (!producer) {
        // =================================== consumer ===================================
        bf16x8 Wr[9][4];
        {
            const elt_t* Wp = (const elt_t*)a.W;
            const bool wflip = a.dstep < 0;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int wt = wflip ? 8 - tap : tap;
                    Wr[tap][ks] = *(const bf16x8*)(Wp + (((long)wt * 8 + 2 * ks + lh) * 64 + 32 * wn + l31) * 8);
                }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(Wr[tap][ks]));
        }
        int rowaddr[6];
        f32x16 acc[2];
        // (reading the next half-step's first two fragment groups across the barrier -- legal: they only touch ring rows complete
        // at the last B2 -- shortened the stamped half-steps by 2-5 % and LENGTHENED the un-stamped launch by 8 %: not kept)
        auto half = [&](auto half_c) __attribute__((always_inline)) {
            constexpr int HF = decltype(half_c)::value;
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            bf16x8 pf[3][4];
            auto read_group = [&](int gg) __attribute__((always_inline)) {      // (dx, ks) = (gg / 4, gg % 4)
                if (gg < 12) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) pf[gg % 3][q] = *(const bf16x8*)(ring + rowaddr[2 * HF + q] + (gg / 4) * PITCH + (gg % 4) * 32);
                }
            };
            // the reads run two groups (384 MFMA cycles) ahead: one group is less than an LDS round trip under this load
            read_group(0); read_group(1);
#pragma unroll
            for (int g = 0; g < 12; ++g) {
                read_group(g + 2);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)       // (the first product of an accumulator takes the constant 0 as C: no zeroing pass)
                        acc[i] = USTRUN_MFMA_32x32x16(Wr[dy * 3 + g / 4][g % 4], pf[g % 3][i + dy], (g == 0 && dy == 0) ? zero16 : acc[i], 0, 0, 0);
            }
            // park the half's rows (bf16) in the wave's scratch: lane = pixel, 4 consecutive channels per register quad
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * HF + ii;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 h;
                    h[0] = (elt_t)acc[ii][4 * g]; h[1] = (elt_t)acc[ii][4 * g + 1];
                    h[2] = (elt_t)acc[ii][4 * g + 2]; h[3] = (elt_t)acc[ii][4 * g + 3];
                    *(bf16x4*)(Ew + (i * 32 + l31) * EPITCH + (8 * g + 4 * lh) * 2) = h;
                }
            }
        };
        auto iteration = [&](auto m_c) __attribute__((always_inline)) {
            constexpr int bA = decltype(m_c)::value, bB = bA == 2 ? 0 : bA + 1;
            rowaddr[0] = frA + bA * BANKB; rowaddr[1] = rowaddr[0] + ROWB; rowaddr[2] = rowaddr[0] + 2 * ROWB; rowaddr[3] = rowaddr[0] + 3 * ROWB;
            rowaddr[4] = hiB ? frB + bB * BANKB + 4 * ROWB : frA + bA * BANKB + 4 * ROWB;
            rowaddr[5] = rowaddr[4] + ROWB;
            const Cur cn = advance(cl);
            if constexpr (DIAG) dt0 = stamp();
            half(ic<0>{});
            if constexpr (DIAG) { dt1 = stamp(); dsum[0] += dt1 - dt0; dt0 = dt1; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // B1
            asm volatile("" ::: "memory");
            if constexpr (DIAG) { dt1 = stamp(); dsum[1] += dt1 - dt0; dt0 = dt1; }
            half(ic<1>{});
            if constexpr (DIAG) { dt1 = stamp(); dsum[2] += dt1 - dt0; dt0 = dt1; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // B2
            asm volatile("" ::: "memory");
            if constexpr (DIAG) { dt1 = stamp(); dsum[3] += dt1 - dt0; dsum[5] += 1; }
            cp = cc; pend = cc.valid && cc.k >= 1;
            cc = cw; cw = cl; cl = cn;
        };
        while (true) {
            if (!(cl.valid | cw.valid | cc.valid | (int)pend)) break;
            iteration(ic<0>{});
            if (!(cl.valid | cw.valid | cc.valid | (int)pend)) break;
            iteration(ic<1>{});
            if (!(cl.valid | cw.valid | cc.valid | (int)pend)) break;
            iteration(ic<2>{});
        }
        if constexpr (DIAG) {
            if (lane == 0 && p.dbg) {
#pragma unroll
                for (int k = 0; k < 8; ++k) p.dbg[((long)blockIdx.x * 8 + wave8) * 8 + k] = dsum[k];
            }
        }
        return;
    }

    // =================================== producer ===================================
    int goffb[NR], rp[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int q = tp + 256 * i, hp = q >> 3, r = hp / HW, px = hp - r * HW;
        goffb[i] = (r * sH + px * sW + (tp & 7) * 8) * 2;
        rp[i] = r << 8 | px;
    }
    const int loff0 = (tp >> 3) * PITCH + (tp & 7) * 16;
    const int wd8_dummy = RINGB + 4 * EWAVE + (tp & 127) * 16;
    const int pp = lane >> 2, o = lane & 3;
    const int st_lane = (pp * 64 + 32 * wn + 8 * o) * 2;
    u32x4 stg[NR];
    unsigned okmW = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) stg[i] = (u32x4){0u, 0u, 0u, 0u};
    f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
    const short a_floor16 = S.relu ? (short)0 : (short)0x8000;     // ReLU on the rounded bf16 pairs (act8_bf16)
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    // BNS: y pieces in flight ([0..3]: half 1 of the step stored first in an iteration, [4..7]: half 0 of the next), the layer's
    // constants for the lane's 8 channels under the two cursors an iteration stores for
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(BNS ? a.bny : a.out0), 0, (int)min((long)a.N * H * W * 128, 0x7fffffffL), 0x00020000);
    u32x4 ypre[8];
    float bsc[2][8], bsh[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ypre[i] = (u32x4){0u, 0u, 0u, 0u};
    auto bns_consts = [&](const Cur& c, int set) __attribute__((always_inline)) {
        if constexpr (BNS) {
            int grp = 0;
            if (a.bn_gN > 0) {
#pragma unroll
                for (int q = 1; q < 8; ++q) grp += (c.img >= q * a.bn_gN) ? 1 : 0;      // (passes <= 8: the launcher checks; no division here)
            }
            const float* ps = a.bnsc + (long)grp * a.bn_gstride + 32 * wn + 8 * o;
            const float* pb = a.bnsh + (long)grp * a.bn_gstride + 32 * wn + 8 * o;
            const f32x4 x0 = *(const f32x4*)ps, x1 = *(const f32x4*)(ps + 4), b0 = *(const f32x4*)pb, b1 = *(const f32x4*)(pb + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bsc[set][e] = x0[e]; bsc[set][4 + e] = x1[e]; bsh[set][e] = b0[e]; bsh[set][4 + e] = b1[e]; }
        }
    };
    unsigned offL[NR], okmL = 0;
    unsigned xokm = 0;
    int xok_x0 = -(1 << 20);
    auto offsets_one = [&](const Cur& c, int i) __attribute__((always_inline)) {
        const int y0g = c.ybeg - 1 + 8 * c.k;
        const int nrows = !c.valid ? 0 : (c.k == c.S ? 2 : 8);
        const int lo = max(0, -y0g), hi = min(nrows, H - y0g);
        const unsigned ymask = hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
        const int gbase = (y0g * sH + (c.x0 - 1) * sW) * 2;
        const unsigned ok = (ymask >> (rp[i] >> 8)) & (xokm >> i) & 1u;
        offL[i] = (unsigned)(gbase + goffb[i]) | ((ok ^ 1u) << 31);
        okmL = (okmL & ~(1u << i)) | (ok << i);
    };
    auto xok_update = [&](const Cur& c) __attribute__((always_inline)) {
        if (c.x0 != xok_x0) {
            xok_x0 = c.x0;
            xokm = 0;
#pragma unroll
            for (int i = 0; i < NR; ++i)
                xokm |= ((unsigned)(i < NR - 1 || tp < 128) & (unsigned)((unsigned)(c.x0 - 1 + (rp[i] & 255)) < (unsigned)W)) << i;
        }
    };
    auto xform = [&](u32x4 raw, bool ok) __attribute__((always_inline)) {
        if constexpr (!XF) return raw;
        u32x4 u = act8_bf16(raw, sc0, sc1, sh0, sh1, a_floor16);
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = ok ? u[e] : 0u;
        return u;
    };
    auto load_consts = [&](const Cur& c) __attribute__((always_inline)) {
        if constexpr (XF) {
            if (S.scale) {
                const long go = S.gN > 0 ? (long)(c.img / S.gN) * S.gstride : 0;
                const float* scp = S.scale + go + 8 * (tp & 7);
                const float* shp = S.shift + go + 8 * (tp & 7);
                sc0 = *(const f32x4*)scp; sc1 = *(const f32x4*)(scp + 4);
                sh0 = *(const f32x4*)shp; sh1 = *(const f32x4*)(shp + 4);
            }
        }
    };
    // the parked rows of a half -> global (16 px x 64 B per store instruction) + statistics of the stored values
    // byte offset of piece tt of half HF of cursor c's step in the output (and in y: same layout); bit 31 set outside the image
    auto piece_off = [&](int HF, int tt, const Cur& c, bool live) __attribute__((always_inline)) {
        const int i = 2 * HF + (tt >> 1), px = 16 * (tt & 1) + pp;
        const int y = c.ybeg + 8 * (c.k - 1) + 4 * wm + i;
        const int ylim = min(c.ybeg + 8 * c.S, H);        // the item's rows (uniform plan: seg = 8 S but for an image's last segment)
        const bool inimg = live & (y < ylim) & (c.x0 + px < W);
        return inimg ? (unsigned)(st_lane + (tt & 1) * 2048 + __builtin_amdgcn_readfirstlane(c.img * out_bytes + (y * W + c.x0) * 128)) : 0x80000000u;
    };
    auto epi_B = [&](int HF, int tt, const Cur& c, bool live, int set = 0) __attribute__((always_inline)) {
        const int i = 2 * HF + (tt >> 1), px = 16 * (tt & 1) + pp;
        const int y = c.ybeg + 8 * (c.k - 1) + 4 * wm + i;
        const int ylim = min(c.ybeg + 8 * c.S, H);        // the item's rows (uniform plan: seg = 8 S but for an image's last segment)
        u32x4 u = *(const u32x4*)(Ew + (i * 32 + px) * EPITCH + o * 16);
        const bool inimg = live & (y < ylim) & (c.x0 + px < W);
        // (row offset in the vector offset, soffset 0: see the four-wave kernel's epi_B)
        const unsigned voff = inimg ? (unsigned)(st_lane + (tt & 1) * 2048 + __builtin_amdgcn_readfirstlane(c.img * out_bytes + (y * W + c.x0) * 128)) : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b128(u, ro, voff, 0, 0);
        if constexpr (BNS) {         // (the y piece was fetched from the same offset one step ago: zeros outside the image, like u below)
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = inimg ? u[e] : 0u;
            const bf16x8 v = __builtin_bit_cast(bf16x8, u), y8 = __builtin_bit_cast(bf16x8, ypre[(HF ? 0 : 4) + tt]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float yf = (float)y8[e];
                const float dz = fma_scalar(yf, bsc[set][e], bsh[set][e]) > 0.f ? (float)v[e] : 0.f;
                s1[e] = add_scalar(s1[e], dz);
                s2[e] = fma_scalar(dz, yf, s2[e]);
            }
        }
        if constexpr (STAT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = inimg ? u[e] : 0u;
            const bf16x8 v = __builtin_bit_cast(bf16x8, u);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                s1[e] = add_scalar(s1[e], f);
                s2[e] = fma_scalar(f, f, s2[e]);
            }
        }
    };
    auto stat_flush = [&](const Cur& c) __attribute__((always_inline)) {
        if constexpr (STAT || BNS) {
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
                if (flat && c.ybeg == 0 && c.S == p.steps) {        // a strip nobody cut: its second slot's rows are zeros, written here
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    float* row1 = row + 2 * 2 * 64;
                    *(f32x4*)row1 = z; *(f32x4*)(row1 + 4) = z; *(f32x4*)(row1 + 64) = z; *(f32x4*)(row1 + 68) = z;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
        }
    };

    auto iteration = [&](auto m_c) __attribute__((always_inline)) {
        constexpr int bA = decltype(m_c)::value, bW = bA == 0 ? 2 : bA - 1;
        const bool live = cc.valid && cc.k >= 1;
        // the pass constants of the group whose rows sit in stg (fetched last iteration): loaded HERE and used below, inside one
        // iteration.  Loaded at the end of the previous iteration (as the four-wave kernel does) they are loop-carried registers,
        // and the copies hipcc places at the loop's back edge wait vmcnt(0) -- for the row fetches just issued: ~2000 cycles per
        // iteration that no stamp inside the segments showed
        load_consts(cw);
        bns_consts(cp, 0); bns_consts(cc, 1);
        char* wdst = ring + bW * BANKB + loff0;
        char* wd8 = tp < 128 ? wdst + 8 * (32 * PITCH) : smem + wd8_dummy;
        const Cur cn = advance(cl);
        xok_update(cl);
        const int in_soff = __builtin_amdgcn_readfirstlane(cl.img * img_bytes);
        auto stage = [&](int i) __attribute__((always_inline)) {
            *(u32x4*)(i < NR - 1 ? wdst + i * (32 * PITCH) : wd8) = xform(stg[i], (okmW >> i) & 1u);
            stg[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, offL[i], in_soff, 0);
        };
        // ---- consumers' half 0: the owed stores of the previous step's half 1, then the staging of the next group ----
        if constexpr (DIAG) dt0 = stamp();
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) epi_B(1, tt, cp, pend, 0);
        if (pend && cp.k == cp.S) stat_flush(cp);
        if constexpr (BNS) {         // half 1 of THIS step is stored at the top of the next iteration (its cp = cc, its pend = live)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) ypre[tt] = __builtin_amdgcn_raw_buffer_load_b128(ry, piece_off(1, tt, cc, live), 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) offsets_one(cl, i);
#pragma unroll
        for (int i = 0; i < 5; ++i) stage(i);
        if constexpr (DIAG) { dt1 = stamp(); dsum[0] += dt1 - dt0; dt0 = dt1; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // B1: half 0 of this step is parked
        asm volatile("" ::: "memory");
        if constexpr (DIAG) { dt1 = stamp(); dsum[1] += dt1 - dt0; dt0 = dt1; }
        // ---- consumers' half 1: the stores of half 0, the rest of the staging ----
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) epi_B(0, tt, cc, live, 1);
        if constexpr (BNS) {         // half 0 of the NEXT step (its cc = cw) is stored in the next iteration's second half
            const bool live_n = cw.valid && cw.k >= 1;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) ypre[4 + tt] = __builtin_amdgcn_raw_buffer_load_b128(ry, piece_off(0, tt, cw, live_n), 0, 0);
        }
#pragma unroll
        for (int i = 5; i < NR; ++i) stage(i);
        okmW = okmL;
        if constexpr (DIAG) { dt1 = stamp(); dsum[2] += dt1 - dt0; dt0 = dt1; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // B2: half 1 parked; the ring bank written above is complete
        asm volatile("" ::: "memory");
        if constexpr (DIAG) { dt1 = stamp(); dsum[3] += dt1 - dt0; dsum[5] += 1; }
        cp = cc; pend = live;
        cc = cw; cw = cl; cl = cn;
    };
    while (true) {
        if (!(cl.valid | cw.valid | cc.valid | (int)pend)) break;
        iteration(ic<0>{});
        if (!(cl.valid | cw.valid | cc.valid | (int)pend)) break;
        iteration(ic<1>{});
        if (!(cl.valid | cw.valid | cc.valid | (int)pend)) break;
        iteration(ic<2>{});
    }
    if constexpr (DIAG) {
        if (lane == 0 && p.dbg) {
#pragma unroll
            for (int k = 0; k < 8; ++k) p.dbg[((long)blockIdx.x * 8 + wave8) * 8 + k] = dsum[k];
        }
    }
}

// which build serves a launch (conv3x3_ws64_launch_bf16): the consumer / producer waves unless a debug flag forces another
static bool ws_cp_build() { return (g_debug_flags & 32) ? true : (g_debug_flags & (16 | 4 | 2)) ? false : true; }

// segments per strip: whole waves of blocks over the 256 CUs, few bubbles (one staging-only iteration per item) -- or, where that
// leaves a round of items half empty, the flat plan (WsPlan): equal step counts per block
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
    p.L = 0; p.steps = steps;
    p.dbg = nullptr;
    // the flat plan: L steps per block, its range touching at most cdiv(L - 1, steps) + 1 strips (ustrun_debug_flags2 bit 2 keeps
    // the uniform plan: A/B runs)
    const long total = (long)a.N * p.sx * steps;
    const int L = (int)((total + 255) / 256);
    if (ws_cp_build() && !(g_debug_flags2 & 4) && L >= steps && total < (1L << 30)) {
        const double cost = L + 1.5 * (cdiv(L - 1, steps) + 1);
        if (cost < 0.97 * best) p.L = L;
    }
    return p;
}
static int ws_grid(const IgemmArgs& a, const WsPlan& p) {
    return p.L > 0 ? (int)cdiv((long)a.N * p.sx * p.steps, (long)p.L) : cdiv(p.items, p.ipb);
}

}  // namespace


bool ws64_supported(const IgemmArgs& a) {
    if (a.nseg != 9 || a.nz != 1 || a.s_in != 1 || a.s_out != 1 || a.segw != 3 || a.nsrc != 1) return false;
    if (!((a.d0 == -1 && a.dstep == 1) || (a.d0 == 1 && a.dstep == -1))) return false;
    if (a.Cin != 64 || a.Cout != 64 || a.C0 != 64 || a.out_esz != 2 || a.bias) return false;
    const SrcDev& s = a.src[0];
    if (s.sC != 1 || s.esz != 2 || s.pool || s.off_y || s.off_x || s.H != a.Hb || s.W != a.Wb || s.sW != 64) return false;
    if (a.Ho != a.Hb || a.Wo != a.Wb || a.Wb < 32 || a.Hb < 16) return false;
    if ((long)a.N * s.sN * 2 >= 0x7fffffffL || (long)a.N * a.Hb * a.Wb * 128 >= 0x7fffffffL) return false;     // one buffer descriptor each
    const WsPlan p = ws_plan(a);
    return p.items >= 192;              // smaller problems (the batch-1 forward): the tiled kernel fills the chip better
}

// two rows (one per consumer-wave pair) per item; flat plan: per slot, two slots per strip
int ws64_stat_rows(const IgemmArgs& a) { const WsPlan p = ws_plan(a); return p.L > 0 ? a.N * p.sx * 4 : p.items * 2; }

// can the streaming kernel's input gradient also form the BatchNorm-backward sums of the layer whose da it writes?
bool ws64_bnsum_supported(const IgemmArgs& a) {
    if (!ws64_supported(a) || a.src[0].scale || a.src[0].relu) return false;
    if (g_debug_flags & (1 | 2 | 4 | 16)) return false;              // (another build of the kernel, or the tiled kernel, is forced)
    if (g_dbg.p) return false;              // (a stamp buffer selects the DIAG build, which forms no sums: the caller runs the reduce pass)
    return a.bn_gN == 0 || a.N <= 8 * a.bn_gN;                       // at most eight passes (the producers count them without a division)
}

int conv3x3_ws64_launch_bf16(const IgemmArgs& a, hipStream_t st) {
    WsPlan p = ws_plan(a);
    const int grid = ws_grid(a, p);
    USTRUN_TRY(debug_buffer_for(grid, "conv3x3_ws64_bf16", &p.dbg));     // set: the DIAG build runs and writes [block][wave][8] u64 there
    bool xf = false;
    set_last_variant(0x57530000 | ((a.src[0].scale != nullptr || a.src[0].relu != 0) ? 1 : 0));     // 'WS' | XF
    xf |= a.src[0].scale != nullptr || a.src[0].relu != 0;
    USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64_kernel<true, true>, LDSB, "conv3x3_ws64_bf16"));
    USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64_kernel<true, false>, LDSB, "conv3x3_ws64_bf16"));
    USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64_kernel<false, true>, LDSB, "conv3x3_ws64_bf16"));
    USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64_kernel<false, false>, LDSB, "conv3x3_ws64_bf16"));
    const bool stat = a.stat != nullptr;
    // Which build (measured in one process, N = 64 images of 256^2, profiles/r03_ab_ws64_8waves.log): the eight-wave kernel wins on the
    // plain-source launches (input gradients: 0.320 vs 0.333 ms) and loses where the loader transforms and statistics are taken
    // (forward: 0.394 vs 0.339 ms) -- its per-wave address / cursor / wait instructions double per SIMD while a 16x16x32 MFMA
    // hides half as many of them.  debug flag bit 1 forces four waves everywhere, bit 2 eight waves everywhere.
    const bool eight = (g_debug_flags & 4) ? true : (g_debug_flags & 2) ? false : !(xf || a.stat);
    // consumer / producer waves (round 3, second form): debug flag bit 4 forces it off, bit 5 on everywhere
    const bool cpw = ws_cp_build();
    USTRUN_CHECK(cpw || p.L == 0, "conv3x3_ws64: the flat plan belongs to the consumer / producer build");
    const int flatbit = p.L > 0 ? 0x800 : 0;          // variant code: | 0x800 = flat plan
    if (cpw && p.dbg) {          // stamped build: [block][wave 0..7][8] u64 (0/2: work of the two segments, 1/3: waits at B1 / B2, 5: iterations)
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64cp_kernel<true, true, true>, LDSBCP, "conv3x3_ws64cp_bf16 (diag)"));
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64cp_kernel<false, false, true>, LDSBCP, "conv3x3_ws64cp_bf16 (diag)"));
        if (xf) hipLaunchKernelGGL((conv3x3_ws64cp_kernel<true, true, true>), dim3(grid), dim3(512), LDSBCP, st, a, p);
        else hipLaunchKernelGGL((conv3x3_ws64cp_kernel<false, false, true>), dim3(grid), dim3(512), LDSBCP, st, a, p);
        USTRUN_LAUNCH_CHECK("conv3x3_ws64cp_bf16 (diag)");
        return 0;
    }
    if (a.bny) {                  // input gradient + BatchNorm-backward sums: the consumer / producer build only (ws64_bnsum_supported)
        USTRUN_CHECK(cpw && !p.dbg && !xf && a.stat && a.bnsc && a.bnsh, "conv3x3_ws64: BatchNorm-backward sums need the plain consumer / producer build");
        set_last_variant(0x57530000 | 0x200 | 0x400 | flatbit);                 // 'WS' | consumer/producer | sums
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64cp_kernel<false, false, false, true>, LDSBCP, "conv3x3_ws64cp_bf16"));
        hipLaunchKernelGGL((conv3x3_ws64cp_kernel<false, false, false, true>), dim3(grid), dim3(512), LDSBCP, st, a, p);
        USTRUN_LAUNCH_CHECK("conv3x3_ws64cp_bf16");
        return 0;
    }
    if (cpw && !p.dbg) {
        set_last_variant(0x57530000 | 0x200 | flatbit | (xf ? 1 : 0));       // 'WS' | consumer/producer | XF
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64cp_kernel<true, true>, LDSBCP, "conv3x3_ws64cp_bf16"));
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64cp_kernel<true, false>, LDSBCP, "conv3x3_ws64cp_bf16"));
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64cp_kernel<false, true>, LDSBCP, "conv3x3_ws64cp_bf16"));
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64cp_kernel<false, false>, LDSBCP, "conv3x3_ws64cp_bf16"));
        const bool stat_ = a.stat != nullptr;
        if (xf && stat_) hipLaunchKernelGGL((conv3x3_ws64cp_kernel<true, true>), dim3(grid), dim3(512), LDSBCP, st, a, p);
        else if (xf) hipLaunchKernelGGL((conv3x3_ws64cp_kernel<true, false>), dim3(grid), dim3(512), LDSBCP, st, a, p);
        else if (stat_) hipLaunchKernelGGL((conv3x3_ws64cp_kernel<false, true>), dim3(grid), dim3(512), LDSBCP, st, a, p);
        else hipLaunchKernelGGL((conv3x3_ws64cp_kernel<false, false>), dim3(grid), dim3(512), LDSBCP, st, a, p);
        USTRUN_LAUNCH_CHECK("conv3x3_ws64cp_bf16");
        return 0;
    }
    if (eight && p.dbg) {                        // stamped build of the eight-wave kernel: [block][wave 0..7][8] u64
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64x8_kernel<true, true, true>, LDSB8, "conv3x3_ws64x8_bf16 (diag)"));
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64x8_kernel<false, false, true>, LDSB8, "conv3x3_ws64x8_bf16 (diag)"));
        if (xf) hipLaunchKernelGGL((conv3x3_ws64x8_kernel<true, true, true>), dim3(grid), dim3(512), LDSB8, st, a, p);
        else hipLaunchKernelGGL((conv3x3_ws64x8_kernel<false, false, true>), dim3(grid), dim3(512), LDSB8, st, a, p);
        USTRUN_LAUNCH_CHECK("conv3x3_ws64x8_bf16 (diag)");
        return 0;
    }
    if (eight) {                                 // round 3: eight waves, two per SIMD
        set_last_variant(0x57530000 | 0x100 | (xf ? 1 : 0));       // 'WS' | 8 waves | XF
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64x8_kernel<true, true>, LDSB8, "conv3x3_ws64x8_bf16"));
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64x8_kernel<true, false>, LDSB8, "conv3x3_ws64x8_bf16"));
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64x8_kernel<false, true>, LDSB8, "conv3x3_ws64x8_bf16"));
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64x8_kernel<false, false>, LDSB8, "conv3x3_ws64x8_bf16"));
        if (xf && stat) hipLaunchKernelGGL((conv3x3_ws64x8_kernel<true, true>), dim3(grid), dim3(512), LDSB8, st, a, p);
        else if (xf) hipLaunchKernelGGL((conv3x3_ws64x8_kernel<true, false>), dim3(grid), dim3(512), LDSB8, st, a, p);
        else if (stat) hipLaunchKernelGGL((conv3x3_ws64x8_kernel<false, true>), dim3(grid), dim3(512), LDSB8, st, a, p);
        else hipLaunchKernelGGL((conv3x3_ws64x8_kernel<false, false>), dim3(grid), dim3(512), LDSB8, st, a, p);
        USTRUN_LAUNCH_CHECK("conv3x3_ws64x8_bf16");
        return 0;
    }
    if (p.dbg) {
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64_kernel<true, true, true>, LDSB, "conv3x3_ws64_bf16 (diag)"));
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_ws64_kernel<false, false, true>, LDSB, "conv3x3_ws64_bf16 (diag)"));
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
