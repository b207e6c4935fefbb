// conv_halo_bf16.hip -- 3x3 convolution (forward and input-gradient; unet_parts.py:16,19 and their autograd) on the
// bf16 matrix cores with LDS-staged input HALO tiles: for every 32-channel chunk of K the block loads its
// (TH+2) x (TW+2)-pixel input patch ONCE -- BatchNorm affine + ReLU (+2x2 max-pool, concat, zero padding) applied in f32
// on the way in, rounded to bf16, one 80-byte row per pixel (64 B of channels + 16 B pad: conflict-free fragment reads
// with compile-time offsets) -- and all nine taps read their shifted A fragments from that one patch.
// Weights stream tap by tap as pure copies: bf16 [tap][K/8][N][8] tiles fetched by LDS-DMA (global_load_lds_dwordx4,
// no VGPRs) into a double buffer, one stage ahead of the MFMAs.
//
// Tile: TH x TW output pixels x BN channels per 256-thread block (8x32 or 16x16 pixels x 128 channels with 4x2
// v_mfma_f32_32x32x16_bf16 accumulators per wave, MI = 4: 0.75 LDS fragment reads per MFMA; 8x16 x 128 or 8x32 / 16x16 x
// 64 with 2x2 accumulators, MI = 2, for small grids and 64-channel outputs).  Two blocks per CU.
#include "common.h"
#include "loader.h"
#include <stdlib.h>
#include <type_traits>

namespace ustrun {
namespace {

typedef __attribute__((ext_vector_type(8))) elt_t bf16x8;
typedef __attribute__((ext_vector_type(4))) elt_t bf16x4;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;


__device__ __attribute__((aligned(16))) const unsigned g_zero16[4] = {0u, 0u, 0u, 0u};   // source of padding pixels

template <int V> using ic = std::integral_constant<int, V>;
template <class T> struct more_value { static constexpr bool value = true; };      // runtime bool: not used as a constant
template <bool B> struct more_value<std::integral_constant<bool, B>> { static constexpr bool value = B; };

// MI = 32-pixel MFMA sub-tiles per wave along M.  TW = 16: a sub-tile is 2 rows x 16 px (wave tile 2*MI rows x 16 px);
// TW = 32: a sub-tile is one row of 32 consecutive pixels (wave tile MI rows x 32 px).
//
// The kernel is instruction-issue bound, not MFMA- or LDS-bound (SQ counters: at ~210 VALU+SALU instructions per
// 16 MFMAs the two waves of a SIMD keep its issue port 80 % busy), so the stage body is built to need almost no
// address arithmetic:
//   * patch rows are padded to 80 bytes instead of XOR-swizzled (16 consecutive rows still cover all 64 banks once),
//     so every A fragment of every tap is ONE per-chunk base register plus a compile-time immediate; the nine taps
//     are unrolled, the input-gradient walks the weight slices backwards instead of mirroring the geometry;
//   * the patch pixel, bounds test and source offset of every staging item are tile constants computed once per
//     source, not once per stage.
// Pipeline (one stage = NT taps of one K chunk, NT*2*MI*BK/16 MFMAs per wave, one barrier):
//   top    : LDS-DMA of the next stage's NT weight tiles (two buffers of NT tiles); LDS-DMA of BATCH items (8 channels
//            of one patch pixel per thread) of the NEXT chunk's patch -- straight into the other patch buffer when the
//            source needs no arithmetic (dY, ConvTranspose outputs), else into a private raw slot
//            (XF: BatchNorm affine / ReLU / pooling on load), one stage ahead of their transform
//   middle : all fragment reads, the raw items the PREVIOUS stage fetched, then the MFMAs of the stage's taps with the
//            items' BatchNorm affine + ReLU (+2x2 max) in f32 in their shadow (one basic block: branch-free transform,
//            out-of-patch items land in a trash slot), written to the other patch buffer
//   bottom : s_waitcnt vmcnt(0): this top's transfers have landed; barrier.
// A wave's stage used to be DMA issue -> fragment reads -> 16 MFMAs -> wait -> transform -> barrier in sequence, ~700
// cycles outside the 512 MFMA cycles, more than the SIMD's other wave can cover.
// DIL: dilation of the 3x3 taps (1 everywhere in the U-Net; 2 and 4 for the dilated bottlenecks of DeepLabV2-ResNet, reference
// networks/backbone/resnet.py:8-10,193-200): the halo is DIL pixels wide and tap (ty, tx) reads DIL * (ty, tx) into the patch.
// BREG (round 3): the weight fragments do not pass through LDS.  Each wave loads the B fragments of the NEXT tap straight into
// registers (one 16-byte buffer load per fragment: the packed layout [tap][K/8][N][8] IS the fragment layout, 32 columns x 16 B
// contiguous per half-wave) one stage ahead of their MFMAs.  That removes the two weight LDS-DMAs and the four B-fragment LDS
// reads of a stage -- and, with nothing shared between the waves inside a chunk any more, the per-tap workgroup barrier: the
// waves meet once per K chunk (when the patch buffers swap) instead of nine times, and drift apart in between.
// M16 (round 4, plain sources with register-fed weights only): the same tile on v_mfma_f32_16x16x32 instead of 32x32x16 -- one
// instruction spans the whole 32-channel chunk; a 32-pixel sub-tile is two 16-pixel halves, the wave's 64 columns four groups of
// 16; operands swapped (A = weights, B = pixels) so that the product is D[channel][pixel]: four consecutive channels of a pixel
// in the four registers of a tile -> 2 cvt_pk + one ds_write_b64 in the epilogue instead of four 2-byte writes.  Same fragment
// bytes from LDS, same weight loads, same accumulator count; what differs is the clock the chip holds (MI355X_MICROARCH.md,
// "DVFS give-back" item 7).
// LINW (round 5; maps whose sides the rectangular tiles do not divide: 18 / 36 / 72 pixels of the 288 x 288 workload, 24 of the
// 384 x 384 one -- an 8 x 16 tile covers an 18 x 18 map at 42 % useful MFMAs): the M dimension is the FLAT PADDED space of one
// pass -- image stride (H + 1)(W + 1), row pitch RP = W + 1: one zero column behind every row and one zero row behind every
// image serve as right AND left, lower AND upper padding of the neighbours -- cut into tiles of TH x TW = 256 consecutive
// positions, whatever image or row they fall in.  Tap (ty, tx) is then the constant shift ty RP + tx for every position, so the
// patch is the 256 + 2 RP + 2 positions around the tile, a fragment is still one base register + a compile-time immediate, and
// nothing in the stage body changes; what changes is where a patch position comes from (src_setup: position -> image, row,
// column, once per tile) and where an output position goes (epilogue: the pad positions are computed and dropped).  Useful
// MFMAs: H W / ((H + 1)(W + 1)) = 90 % at 18 x 18, 95 % at 36 x 36.  tiles_x = tiles per full pass, tiles_y = images per pass
// (BatchNorm constants are per pass: a tile never crosses one; a shorter last pass has its own tile count).
template <int TH, int TW, int BN, int BK, int MI, bool POOL, int NT, bool XF, int DIL = 1, bool BREG = false, bool M16 = false, bool BNS = false,
          int LINW = 0>
__global__ __launch_bounds__(256, (TH * TW * BN >= 512 * 64 && BN == 64 ? 1 : 2)) void conv3x3_halo_bf16_kernel(const IgemmArgs a, const int tiles_x, const int tiles_y,
                                                                   const int nt_total) {
    static_assert(BK == 32, "80-byte patch rows hold one 32-channel chunk");
    constexpr bool LIN = LINW > 0;
    constexpr int RP = LINW + 1;                // LIN: row pitch of the flat padded space
    constexpr int TM = TH * TW;
    static_assert(!LIN || (DIL == 1 && !POOL && TW == 32 && BREG && RP > 8), "linear tiles: un-dilated, un-pooled, 32-position sub-tiles");
    static_assert(!M16 || (BREG && !XF), "the 16x16x32 build: plain sources, weights through registers");
    static_assert(!BNS || M16, "BatchNorm-backward sums ride on the 16x16x32 epilogue (8 channels of a pixel per lane)");
    static_assert(!BREG || NT <= 2, "register-fed weights: 16 registers per tap and set");
    constexpr int HW2 = LIN ? RP : TW + 2 * DIL;
    constexpr int SR = 32 / TW;                 // tile rows per 32-pixel sub-tile (2 or 1)
    constexpr int WM = TH / (SR * MI), WN = 4 / WM;
    static_assert(WM * WN == 4 && BN == WN * 64, "4 waves, 64 channels per wave");
    constexpr int HP = LIN ? TM + 2 * RP + 2 : (TH + 2 * DIL) * HW2;    // halo pixels
    constexpr int PITCH = 80;                   // patch row: 4 x 16 B of channels + 16 B pad
    constexpr int DSLOTS = (HP * 5 + 63) / 64 * 64;   // direct staging: 16-byte LDS slots incl. the pad slots, whole waves
    constexpr int XSLOTS = (HP * 4 + 63) / 64 * 64;   // transform staging: (pixel, channel group) items, whole waves
    constexpr int AIT = (DSLOTS + 255) / 256;   // staging steps per chunk (taps 0 .. AIT-1), one item per thread each
    constexpr int NSTG = (9 + NT - 1) / NT;     // stages per chunk: NT taps (weight tiles) per barrier
    constexpr int BATCH = (AIT + NSTG - 2) / (NSTG - 1);   // patch items per stage; the chunk's last stage carries the constants
    static_assert(BATCH * (NSTG - 1) >= AIT, "staging fits the chunk");
    constexpr int NP = POOL ? 4 : 1;
    // BREG: every wave issues every staging step (its transfer counts are then compile-time constants, see the counted wait):
    // the slots past the patch image are a few hundred bytes of trash behind it
    constexpr int ABYTES = (BREG ? AIT * 256 : DSLOTS) * 16;
    constexpr int BCH = (BK / 8) * BN;          // 16-byte chunks per B tile
    constexpr int BIT = BCH / 256;
    static_assert(BCH % 256 == 0, "B tile must be a whole number of wave-instructions per wave");
    constexpr int RAWB = XF ? BATCH * NP * 4096 : 0;   // a raw slot: BATCH x NP x 16 B per thread
    // SKEW: the items a stage fetches are transformed under the NEXT stage's MFMAs (two raw slots, by stage parity) --
    // where a second slot still leaves room for two blocks per CU; otherwise after the stage's own MFMAs
    constexpr int BBYTES = BREG ? 0 : 2 * NT * BCH * 16;
    constexpr bool SKEW = XF && 2 * ABYTES + BBYTES + 2 * RAWB + 512 <= 81920;
    constexpr int NRAW = SKEW ? 2 : 1;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                            // 2 x patch [HP][80 B] (current / being staged)
    char* Bs = smem + 2 * ABYTES;               // 2 x NT x [BK/8][BN][8] bf16 (none with BREG)
    char* Raw = Bs + BBYTES;                    // NRAW x raw slot
    char* Cst = Raw + NRAW * RAWB;              // 2 x {32 scales, 32 shifts} f32 of a chunk (kept out of vmcnt's way)

    const int IS = LIN ? (a.Ho + 1) * RP : 1;                                   // LIN: positions per image
    const int lin_full = LIN ? a.N / tiles_y : 0;                               //      full passes
    const int mt_total = LIN ? lin_full * tiles_x + ((a.N - lin_full * tiles_y) * IS + TM - 1) / TM : a.N * tiles_y * tiles_x;
    const int ntiles = mt_total * nt_total;
    int bid = blockIdx.x;
    {
        const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int mtile = bid / nt_total, ntile = bid % nt_total;
    const int n0 = ntile * BN;
    const int lpass = LIN ? min(mtile / tiles_x, lin_full) : 0;
    const int img = LIN ? lpass * tiles_y : mtile / (tiles_y * tiles_x);        // LIN: the first image of the tile's pass,
    const int nimg = LIN ? min(tiles_y, a.N - img) : 1;                         //      the images of that pass,
    const int P0 = LIN ? (mtile - lpass * tiles_x) * TM : 0;                    //      the tile's first position in it
    const int trem = LIN ? 0 : mtile - img * tiles_y * tiles_x;
    const int y0 = LIN ? 0 : (trem / tiles_x) * TH, x0 = LIN ? 0 : (trem % tiles_x) * TW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: LDS-DMA bases and role tests stay scalar
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    const int nchunk = a.Cin / BK;
    const int K8 = a.Cin / 8;
    const elt_t* Wp = (const elt_t*)a.W;
    const elt_t* zsrc = (const elt_t*)g_zero16;
    const bool wflip = a.dstep < 0;             // input-gradient: geometric tap g pairs with weight slice 8 - g

    // ---- A patch staging.  Tile constants per SOURCE (recomputed when a concat switches source):
    //   direct   : item i = LDS slot q = tid + 256 i of the patch image: pixel q / 5, 16-byte group q % 5 (4 = pad)
    //   transform: item i = (pixel (tid + 256 i) / 4, channel group tid % 4): the thread keeps one group, its
    //              scale/shift stay in registers
    // aoff[i] = BYTE offset of the item from the chunk base (image base of the source + first channel of the chunk), bit 31
    // set when the item does not read the image: the buffer-addressed transfer below then writes zeros by its own range
    // check (no zero page, no 64-bit pointer arithmetic per item and chunk: that was 12 VALU instructions x 7 items of
    // the 195 a chunk carried beside its 144 MFMAs); bit i of aokm = the item reads the image. ----
    constexpr int OOB = (int)0x80000000;
    int aoff[AIT];
    unsigned aokm = 0;
    __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)g_zero16, 0, 0, 0x00020000);     // current source, this image
    int asoff = 0;                              // byte offset of the chunk's first channel
    int cur_src = -1;
    float a_floor = 0.f;                        // ReLU as max(x, floor): 0, or -inf for a source without ReLU
    const elt_t* aptr = nullptr;               // chunk base: source + image + first channel of the chunk
    long dq1 = 0, dq2 = 0;                      // pooled source: element offsets of the right / lower neighbour
    f32x4 asc0, asc1, ash0, ash1;
    const int p8 = tid & 3;
    auto src_setup = [&](const SrcDev& S) {
        a_floor = S.relu ? 0.f : -__builtin_inff();
        dq1 = S.sW; dq2 = S.sH;
        const int by = y0 - DIL - S.off_y, bx = x0 - DIL - S.off_x;
        aokm = 0;
        // (re)derived from the thread index on every call: hoisted out of the chunk loop, the per-item pixel coordinates are
        // 3 x AIT registers held across it for a branch taken once per source -- the opaque copy keeps them out of the loop
        int tq = tid;
        asm volatile("" : "+v"(tq));
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int q = tq + 256 * i;
            const int hp = XF ? (q >> 2) : q / 5;
            const int g = XF ? (tq & 3) : q - 5 * hp;
            if constexpr (LIN) {     // patch position hp = flat position P0 - RP - 1 + hp of the pass: input pixel (r, c) of image il
                const int F = P0 - RP - 1 + hp, Fc = F < 0 ? 0 : F;
                const int il = Fc / IS, rem = Fc - il * IS, r = rem / RP, c = rem - r * RP;
                const int ly = r - S.off_y, lx = c - S.off_x;
                const bool ok = hp < HP && g < 4 && F >= 0 && il < nimg && r < a.Ho && c < LINW && ly >= 0 && ly < S.LH && lx >= 0 && lx < S.LW;
                aoff[i] = ok ? 2 * ((int)((long)il * S.sN + (long)ly * S.sH + (long)lx * S.sW) + 8 * g) : OOB;
                aokm |= (ok ? 1u : 0u) << i;
                continue;
            }
            const int hy = hp / HW2, hx = hp - hy * HW2;
            const int ly = by + hy, lx = bx + hx;
            const bool ok = hp < HP && g < 4 && ly >= 0 && ly < S.LH && lx >= 0 && lx < S.LW;
            aoff[i] = ok ? 2 * ((int)((POOL ? 2 : 1) * ((long)ly * S.sH + (long)lx * S.sW)) + 8 * g) : OOB;
            aokm |= (ok ? 1u : 0u) << i;
        }
        ars = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)S.ptr + (long)img * S.sN * 2), 0, 0x7fffffff, 0x00020000);
    };
    auto stage_begin = [&](int c, bool from_lds = true) {   // per-chunk state, set one chunk ahead of its use
        const int cg = c * BK;
        const int second = (a.nsrc == 2 && cg >= a.src[0].C) ? 1 : 0;
        const SrcDev S = pick_src(a.src[0], a.src[1], second != 0);
        const int cl = cg - (second ? a.src[0].C : 0);
        if (second != cur_src) { src_setup(S); cur_src = second; }
        aptr = (const elt_t*)S.ptr + (img * S.sN + cl);      // (the prologue of the transforming variants loads through it)
        asoff = 2 * cl;
        asc0 = asc1 = (f32x4){1.f, 1.f, 1.f, 1.f}; ash0 = ash1 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (S.scale && !from_lds) {                  // prologue: straight from memory, in flight with the patch loads
            const long goff = S.gN > 0 ? (long)(img / S.gN) * S.gstride : 0;
            const float* sc = S.scale + goff + cl + 8 * p8;
            const float* sh = S.shift + goff + cl + 8 * p8;
            asc0 = *(const f32x4*)sc; asc1 = *(const f32x4*)(sc + 4);
            ash0 = *(const f32x4*)sh; ash1 = *(const f32x4*)(sh + 4);
        } else if (S.scale) {                        // staged in LDS two chunks ahead by dma_consts
            const char* cs = Cst + (c & 1) * 256 + p8 * 32;
            asc0 = *(const f32x4*)cs; asc1 = *(const f32x4*)(cs + 16);
            ash0 = *(const f32x4*)(cs + 128); ash1 = *(const f32x4*)(cs + 144);
        }
    };
    // BatchNorm scale/shift of chunk c -> LDS (wave 0, 16 lanes x 16 B): a plain global load here would make the
    // compiler drain vmcnt(0) -- every DMA in flight -- at each use
    auto dma_consts = [&](int c) {
        const int cg = c * BK;
        const bool second = (a.nsrc == 2 && cg >= a.src[0].C);
        const SrcDev S = pick_src(a.src[0], a.src[1], second);
        const int cl = cg - (second ? a.src[0].C : 0);
        if (S.scale && wave == 0 && lane < 16) {
            const long goff = S.gN > 0 ? (long)(img / S.gN) * S.gstride : 0;        // this image's pass
            const float* src = (lane < 8 ? S.scale : S.shift) + goff + cl + 4 * (lane & 7);
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(Cst + (c & 1) * 256), 16, 0, 0);
        }
    };
    // does this wave own a slot of staging step i?  (wave-uniform: the counted waits depend on it)
    auto wave_has = [&](int i) { return BREG || 256 * i + wave * 64 < (XF ? XSLOTS : DSLOTS); };
    // item i of the chunk, b-th item of its stage (both fold to constants: the callers are fully unrolled)
    auto issue_one = [&](int i, int b, char* Adst, char* raw, bool live = true) {
        const int vo = live ? aoff[i] : OOB;
        if constexpr (XF) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int d = 2 * (int)((q & 1 ? dq1 : 0) + (q & 2 ? dq2 : 0));       // wave-uniform: rides in the scalar offset
                dma16_buf_s(vo, ars, asoff + d, raw + ((b * NP + q) * 256 + wave * 64) * 16);
            }
        } else {
            dma16_buf_s(vo, ars, asoff, Adst + (256 * i + wave * 64) * 16);
        }
    };
    auto act8 = [&](bf16x8 r, f32x4& lo, f32x4& hi) {     // bf16 raw -> activated f32
        lo = (f32x4){(float)r[0], (float)r[1], (float)r[2], (float)r[3]} * asc0 + ash0;
        hi = (f32x4){(float)r[4], (float)r[5], (float)r[6], (float)r[7]} * asc1 + ash1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            lo[q] = __builtin_amdgcn_fmed3f(lo[q], a_floor, __builtin_inff());
            hi[q] = __builtin_amdgcn_fmed3f(hi[q], a_floor, __builtin_inff());
        }
    };
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    auto xform8 = [&](const bf16x8* r, bool inimg) {      // NP raw pieces -> one activated (pooled) bf16 group; no branches
        f32x4 lo, hi;
        act8(r[0], lo, hi);
#pragma unroll
        for (int q = 1; q < NP; ++q) {
            f32x4 l2, h2;
            act8(r[q], l2, h2);
            lo = max4(lo, l2); hi = max4(hi, h2);
        }
        bf16x8 h;
        h[0] = (elt_t)lo[0]; h[1] = (elt_t)lo[1]; h[2] = (elt_t)lo[2]; h[3] = (elt_t)lo[3];
        h[4] = (elt_t)hi[0]; h[5] = (elt_t)hi[1]; h[6] = (elt_t)hi[2]; h[7] = (elt_t)hi[3];
        u32x4 u = __builtin_bit_cast(u32x4, h);
#pragma unroll
        for (int q = 0; q < 4; ++q) u[q] = inimg ? u[q] : 0u;      // padding is applied after the activation
        return __builtin_bit_cast(bf16x8, u);
    };
    const int xw0 = (tid >> 2) * PITCH + p8 * 16;         // transform item i lands at xw0 + i * 64 * PITCH
    auto raw_read = [&](int b, const char* raw, bf16x8* r) {
#pragma unroll
        for (int q = 0; q < NP; ++q) r[q] = *(const bf16x8*)(raw + ((b * NP + q) * 256 + tid) * 16);
    };
    // item i -> patch buffer; items past the patch (the rounded-up tail) and a chunk without successor are written back
    // over the thread's own raw slot, which is dead once read
    auto xform_store = [&](int i, const bf16x8* r, char* Adst, bool live, const char* raw) {
        const bool in_patch = live && (tid >> 2) + 64 * i < HP;
        char* dst = in_patch ? Adst + xw0 + i * 64 * PITCH : (char*)raw + tid * 16;
        *(bf16x8*)dst = xform8(r, (aokm >> i) & 1u);
    };
    // ---- B tile of (chunk c, geometric tap): LDS-DMA, 16 B per lane, lane-linear destination ----
    const elt_t* wthr[BIT];
#pragma unroll
    for (int i = 0; i < BIT; ++i) {
        const int idx = tid + 256 * i, o = idx / BN, n = idx % BN;
        wthr[i] = Wp + ((long)o * a.Cout + n0 + n) * 8;
    }
    auto dma_B = [&](int c, int tap, int slot) {      // slot = tile index inside Bs
        const int wt = wflip ? 8 - tap : tap;
        const long woff = ((long)wt * K8 + c * (BK / 8)) * a.Cout * 8;
        char* dst = Bs + slot * (BCH * 16);
#pragma unroll
        for (int i = 0; i < BIT; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t*)(wthr[i] + woff), (lptr_t*)(dst + (wave * 64 + 256 * i) * 16), 16, 0, 0);
    };

    // BREG: the fragments of (chunk c, tap) for this wave's 64 columns, k half lh: [ks][column half].  The loads are inline
    // asm: beside an LDS-DMA in flight hipcc waits vmcnt(0) for any load it knows about (every other stage drained the
    // B loads it had just issued AND the patch transfer); hidden from it, they are counted by hand below.
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, (int)min((long)9 * K8 * a.Cout * 16, 0x7fffffffL), 0x00020000);
    const int bvoff0 = (lh * a.Cout + n0 + wn * 64 + l31) * 16, bvoff1 = bvoff0 + 2 * a.Cout * 16;      // ks = 0, 1; + 512: columns 32..63
    // M16: lane (column l & 15, k group l >> 4) of column group g: the 16-byte item [tap][4 c + (l >> 4)][n0 + 64 wn + 16 g + (l & 15)]
    const int bvoff16 = ((lane >> 4) * a.Cout + n0 + wn * 64 + (lane & 15)) * 16;
    bf16x8 bnx[NT][BK / 16][2];                       // (M16: the four column groups, [g >> 1][g & 1])
    auto load_B = [&](int c, int tap, int k) {       // k: slot of the stage (folds to a constant)
        const int wt = wflip ? 8 - tap : tap;
        const int soff = __builtin_amdgcn_readfirstlane((int)(((long)wt * K8 + c * (BK / 8)) * a.Cout * 16));
        if constexpr (M16)
            asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %4, %5, %6 offen\n\tbuffer_load_dwordx4 %1, %4, %5, %6 offen offset:256\n\t"
                         "buffer_load_dwordx4 %2, %4, %5, %6 offen offset:512\n\tbuffer_load_dwordx4 %3, %4, %5, %6 offen offset:768"
                         : "=&v"(bnx[k][0][0]), "=&v"(bnx[k][0][1]), "=&v"(bnx[k][1][0]), "=&v"(bnx[k][1][1])
                         : "v"(bvoff16), "s"(wrs), "s"(soff) : "memory");
        else
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %4, %6, %7 offen\n\tbuffer_load_dwordx4 %1, %4, %6, %7 offen offset:512\n\t"
                     "buffer_load_dwordx4 %2, %5, %6, %7 offen\n\tbuffer_load_dwordx4 %3, %5, %6, %7 offen offset:512"
                     : "=&v"(bnx[k][0][0]), "=&v"(bnx[k][0][1]), "=&v"(bnx[k][1][0]), "=&v"(bnx[k][1][1])
                     : "v"(bvoff0), "v"(bvoff1), "s"(wrs), "s"(soff) : "memory");
    };

    f32x16 acc[M16 ? 1 : MI][2];
    f32x4 acc16[M16 ? MI : 1][2][4];                  // [sub-tile][16-pixel half][column group]: D[channel][pixel]
#pragma unroll
    for (int i = 0; i < (M16 ? 1 : MI); ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < (M16 ? MI : 1); ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc16[i][h][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // this lane's fragment bases: patch pixel of sub-tile 0 at tap (0,0), K half lh; weight column wn*64 + l31
    const int afrag0 = LIN ? (M16 ? (wm * MI * 32 + (lane & 15)) * PITCH + (lane >> 4) * 16 : (wm * MI * 32 + l31) * PITCH + lh * 16)
                     : M16 ? ((wm * SR * MI) * HW2 + (lane & 15)) * PITCH + (lane >> 4) * 16      // pixel l & 15 of a half, k group l >> 4
                           : ((wm * SR * MI + (TW == 16 ? (l31 >> 4) : 0)) * HW2 + (l31 & (TW - 1))) * PITCH + lh * 16;
    const int bfrag0 = (lh * BN + wn * 64 + l31) * 16;
    const int nstage = nchunk * NSTG;

    // ---- prologue: the first stage's weight tiles and the whole first patch, all transfers in flight together ----
    if constexpr (BREG) {
#pragma unroll
        for (int k = 0; k < NT; ++k) load_B(0, k, k);
    } else {
#pragma unroll
        for (int k = 0; k < NT; ++k) dma_B(0, k, k);
    }
    if (nchunk > 1) dma_consts(1);
    stage_begin(0, false);                        // one memory round trip for constants, patch and weights together
    if constexpr (XF) {
        bf16x8 pv[AIT][NP];
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const bool ok = (aokm >> i) & 1u;
            const elt_t* src = ok ? (const elt_t*)((const char*)aptr + aoff[i]) : zsrc;
#pragma unroll
            for (int q = 0; q < NP; ++q) pv[i][q] = *(const bf16x8*)(src + (ok ? (q & 1 ? dq1 : 0) + (q & 2 ? dq2 : 0) : 0));
        }
#pragma unroll
        for (int i = 0; i < AIT; ++i)
        {   // no branch per item: items past the patch (the rounded-up tail) land in the raw slot, unused until the loop
            const bool in_patch = (tid >> 2) + 64 * i < HP;
            *(bf16x8*)(in_patch ? As + xw0 + i * 64 * PITCH : Raw + tid * 16) = xform8(pv[i], (aokm >> i) & 1u);
        }
    } else {
#pragma unroll
        for (int i = 0; i < AIT; ++i)
            if (wave_has(i)) issue_one(i, 0, As, nullptr);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    bf16x8 acar[MI];                              // BREG, plain source: the A fragments of the next half-stage (read under this one's MFMAs)
    constexpr bool PEEL = BREG && !XF;
    // CT (transforming variants with register-fed weights): the transfer counts are made compile-time WITHOUT a second copy of the
    // body -- in the last chunk the staging steps and the last stage's weight loads are still issued, as dummies (the 16 zero
    // bytes of g_zero16 into the raw slot nobody transforms into a patch; the same weights again) -- so the counted wait is one
    // instruction instead of a ladder of scalar branches and a stage is one basic block
    constexpr bool CT = BREG && XF && SKEW;
    // (pipelining the fragment reads of the transforming variants the same way measured flat on the forward layers -- 5.86 vs
    // 5.74 ms over the U-Net's layers at N = 64, one box -- their stage is paced by the transform's VALU work: plain sources only)
    constexpr bool APIPE = BREG && !XF;
    int ni_prev = 0;                              // BREG without PEEL: patch pieces the previous stage issued (runtime count)
    // one K chunk.  PEEL: `more` (a successor exists) is a compile-time constant -- the last chunk is a second copy of the
    // body -- so that every transfer count is known when the waits are written and a stage stays ONE basic block (the
    // transforming variants sit at the 256-register limit: two copies of their body spill, they keep the runtime form)
    auto chunk = [&](const int c, auto more_c) __attribute__((always_inline)) {
        const bool more = more_c;
        constexpr bool morec = more_value<decltype(more_c)>::value;       // meaningful with PEEL only
        const char* Afrag = As + (c & 1) * ABYTES + afrag0;
        char* Anext = As + ((c + 1) & 1) * ABYTES;
        if (more) stage_begin(c + 1);
        auto stage = [&](auto ic_j) {
            constexpr int j = decltype(ic_j)::value;          // stage of the chunk: taps j*NT .. j*NT + nt - 1
            constexpr int t0 = j * NT, nt = (9 - t0 < NT) ? 9 - t0 : NT;
            constexpr int jn = (j + 1) % NSTG, tn0 = jn * NT, nn = (9 - tn0 < NT) ? 9 - tn0 : NT;
            const int s = c * NSTG + j;
            // top: next stage's weights, one item of the next patch
            bf16x8 bcur[NT][BK / 16][2];
            const bool has_next = PEEL ? (morec || j < NSTG - 1) : (CT || s + 1 < nstage);  // a stage follows (CT: or a dummy reload)
            const int nb_new = (BREG && has_next) ? 4 * nn : 0;        // B loads this stage issues (wave-uniform)
            if constexpr (BREG) {
#pragma unroll
                for (int k = 0; k < NT; ++k)
#pragma unroll
                    for (int ks = 0; ks < BK / 16; ++ks) { bcur[k][ks][0] = bnx[k][ks][0]; bcur[k][ks][1] = bnx[k][ks][1]; }
                if (has_next) {
#pragma unroll
                    for (int k = 0; k < nn; ++k) load_B(jn == 0 ? (CT && !more ? c : c + 1) : c, tn0 + k, k);
                }
            } else if (s + 1 < nstage) {
#pragma unroll
                for (int k = 0; k < nn; ++k) dma_B(jn == 0 ? c + 1 : c, tn0 + k, ((s + 1) & 1) * NT + k);
            }
            if constexpr (j == NSTG - 1) { if (c + 2 < nchunk) dma_consts(c + 2); }
            char* rawW = Raw + (SKEW ? (s & 1) * RAWB : 0);
            const char* rawR = Raw + (SKEW ? ((s & 1) ^ 1) * RAWB : 0);
            int ni = 0;                                       // patch transfers this wave issues in this stage (wave-uniform)
            if constexpr (j < NSTG - 1) {
                if (CT || more) {
#pragma unroll
                    for (int b = 0; b < BATCH; ++b) {
                        constexpr int dummy = 0; (void)dummy;
                        if (j * BATCH + b < AIT && wave_has(j * BATCH + b)) { issue_one(j * BATCH + b, b, Anext, rawW, more); ni += NP; }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (BREG) {
                // this stage's fragments were loaded a stage ago; younger than them: that stage's patch pieces (XF: consumed by
                // this stage's raw reads, so they must have landed too), then this stage's B loads and patch pieces
                if constexpr (PEEL || CT) {  // all compile-time: every wave issues every staging step; `more` is a template value (PEEL)
                                             // or does not change the counts (CT)
                    constexpr auto pieces = [](int jj) constexpr {
                        int n = 0;
                        if ((CT || morec) && jj >= 0 && jj < NSTG - 1)
                            for (int b = 0; b < BATCH; ++b) n += (jj * BATCH + b < AIT) ? NP : 0;
                        return n;
                    };
                    constexpr int younger = ((CT || morec || j < NSTG - 1) ? 4 * nn : 0) + pieces(j) + (XF ? 0 : pieces(j - 1));
                    static_assert(younger < 64, "vmcnt is a 6-bit counter");
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(younger) : "memory");
                } else {
                    const int younger = nb_new + ni + (XF ? 0 : ni_prev);
                    switch (younger < 12 ? younger : 12) {
                        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
                        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    }
                    ni_prev = ni;
                }
#pragma unroll
                for (int k = 0; k < NT; ++k)
#pragma unroll
                    for (int ks = 0; ks < BK / 16; ++ks) asm volatile("" : "+v"(bcur[k][ks][0]), "+v"(bcur[k][ks][1]));   // (no MFMA above the wait)
                __builtin_amdgcn_sched_barrier(0);
            }
            // middle: every fragment is base + immediate
            auto afrag = [&](int tap, int ks, int i) __attribute__((always_inline)) {
                if constexpr (LIN)       // position 32 i (+ 16 ks) of the wave's run, shifted by the tap
                    return *(const bf16x8*)(Afrag + (32 * i + (M16 ? 16 * ks : 0) + (tap / 3) * RP + tap % 3) * PITCH + (M16 ? 0 : ks * 32));
                else if constexpr (M16)       // ks = the 16-pixel half: the right half of a 32-pixel row, or the second row of a 2 x 16 sub-tile
                    return *(const bf16x8*)(Afrag + ((DIL * (tap / 3) + SR * i + (TW == 16 ? ks : 0)) * HW2 + DIL * (tap % 3) + (TW == 16 ? 0 : 16 * ks)) * PITCH);
                else
                return *(const bf16x8*)(Afrag + ((DIL * (tap / 3) + SR * i) * HW2 + DIL * (tap % 3)) * PITCH + ks * 32);
            };
            if constexpr (APIPE) {
                // register-fed weights: nothing is shared between the waves inside a chunk, so the A fragments
                // of the NEXT half-stage (8 or 4 MFMAs ahead, same patch buffer) are read under this half-stage's MFMAs:
                // same 2 x MI fragment registers as reading a whole stage up front, and the LDS latency leaves the
                // critical path everywhere but at a chunk's first half-stage
                if constexpr (j == 0) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) acar[i] = afrag(t0, 0, i);
                }
                if constexpr (M16) {
                    // 16x16x32: a fragment feeds four MFMAs (the four column groups), so it is re-read IN PLACE for the next
                    // half-stage right behind them -- 12 MFMAs before its next use -- instead of into a second register set:
                    // MI fragments live, not 2 MI (the second set put this build 12 registers past the file: scratch reloads
                    // with a full vmcnt drain at the top of every chunk)
#pragma unroll
                    for (int k = 0; k < nt; ++k)
#pragma unroll
                        for (int ks = 0; ks < BK / 16; ++ks) {
                            constexpr int dummy = 0; (void)dummy;
                            const int hn = (k * (BK / 16) + ks + 1);
                            const bool in_stage = hn < nt * (BK / 16);
#pragma unroll
                            for (int i = 0; i < MI; ++i) {
#pragma unroll
                                for (int g = 0; g < 4; ++g)
                                    acc16[i][ks][g] = USTRUN_MFMA_16x16x32(bcur[k][g >> 1][g & 1], acar[i], acc16[i][ks][g], 0, 0, 0);
                                if (in_stage) acar[i] = afrag(t0 + hn / (BK / 16), hn % (BK / 16), i);
                                else if (j < NSTG - 1) acar[i] = afrag(t0 + nt, 0, i);
                            }
                        }
                } else
#pragma unroll
                for (int k = 0; k < nt; ++k)
#pragma unroll
                    for (int ks = 0; ks < BK / 16; ++ks) {
                        bf16x8 anx[MI];
                        constexpr int dummy = 0; (void)dummy;
                        const int hn = (k * (BK / 16) + ks + 1);                 // next half-stage of this stage, if any
                        const bool in_stage = hn < nt * (BK / 16);
                        if (in_stage) {
#pragma unroll
                            for (int i = 0; i < MI; ++i) anx[i] = afrag(t0 + hn / (BK / 16), hn % (BK / 16), i);
                        } else if (j < NSTG - 1) {                               // first half-stage of the next stage (same chunk)
#pragma unroll
                            for (int i = 0; i < MI; ++i) anx[i] = afrag(t0 + nt, 0, i);
                        }
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
                            if constexpr (M16) {
#pragma unroll
                                for (int g = 0; g < 4; ++g)
                                    acc16[i][ks][g] = USTRUN_MFMA_16x16x32(bcur[k][g >> 1][g & 1], acar[i], acc16[i][ks][g], 0, 0, 0);
                            } else {
                            acc[i][0] = USTRUN_MFMA_32x32x16(acar[i], bcur[k][ks][0], acc[i][0], 0, 0, 0);
                            acc[i][1] = USTRUN_MFMA_32x32x16(acar[i], bcur[k][ks][1], acc[i][1], 0, 0, 0);
                            }
                        }
                        if (in_stage || j < NSTG - 1) {
#pragma unroll
                            for (int i = 0; i < MI; ++i) acar[i] = anx[i];
                        }
                    }
            }
            const char* Bp = Bs + (s & 1) * NT * (BCH * 16) + bfrag0;
            bf16x8 bf[NT][BK / 16][2], af[NT][BK / 16][MI];
            if constexpr (!APIPE) {
#pragma unroll
            for (int k = 0; k < nt; ++k) {
                constexpr int dummy = 0; (void)dummy;
                const int tap = t0 + k;
#pragma unroll
                for (int ks = 0; ks < BK / 16; ++ks) {
                    if constexpr (BREG) { bf[k][ks][0] = bcur[k][ks][0]; bf[k][ks][1] = bcur[k][ks][1]; }
                    else {
                        bf[k][ks][0] = *(const bf16x8*)(Bp + k * (BCH * 16) + (2 * ks * BN) * 16);
                        bf[k][ks][1] = *(const bf16x8*)(Bp + k * (BCH * 16) + (2 * ks * BN + 32) * 16);
                    }
#pragma unroll
                    for (int i = 0; i < MI; ++i) af[k][ks][i] = afrag(tap, ks, i);
                }
            }
            }
            // SKEW: the items the previous stage fetched (landed: that stage ended on vmcnt(0)), transformed under the MFMAs
            constexpr int jx = SKEW ? j - 1 : j;              // the stage whose items are transformed here
            constexpr bool XNOW = XF && jx >= 0 && jx < NSTG - 1;
            constexpr int nitem = !XNOW ? 0 : (jx * BATCH + BATCH <= AIT ? BATCH : (AIT > jx * BATCH ? AIT - jx * BATCH : 0));
            bf16x8 rr[BATCH][NP];
            if constexpr (SKEW && nitem > 0) {
#pragma unroll
                for (int b = 0; b < nitem; ++b) raw_read(b, rawR, rr[b]);
            }
            if constexpr (!SKEW && !APIPE) __builtin_amdgcn_sched_barrier(0);
            if constexpr (!APIPE) {
#pragma unroll
            for (int k = 0; k < nt; ++k)
#pragma unroll
                for (int ks = 0; ks < BK / 16; ++ks)
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        if constexpr (M16) {          // ks = the 16-pixel half; the stage's four column groups live in bf[k][g >> 1][g & 1]
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                acc16[i][ks][g] = USTRUN_MFMA_16x16x32(bf[k][g >> 1][g & 1], af[k][ks][i], acc16[i][ks][g], 0, 0, 0);
                        } else {
                        acc[i][0] = USTRUN_MFMA_32x32x16(af[k][ks][i], bf[k][ks][0], acc[i][0], 0, 0, 0);
                        acc[i][1] = USTRUN_MFMA_32x32x16(af[k][ks][i], bf[k][ks][1], acc[i][1], 0, 0, 0);
                        }
                    }
            }
            if constexpr (SKEW) {
#pragma unroll
                for (int b = 0; b < nitem; ++b) xform_store(jx * BATCH + b, rr[b], Anext, more, rawR);
            }
            constexpr int nmfma = nt * (BK / 16) * MI * (M16 ? 4 : 2);
            constexpr int vpm = !SKEW ? 0 : (nitem * (NP * 26 + 14) + nmfma - 1) / nmfma;
            if constexpr (APIPE) {
                // pinned (left alone, hipcc re-fetches each fragment just in time into ONE register set: read -> wait -> 2 MFMAs):
                // the raw items (and a chunk's first fragments) up front; per half-stage two MFMAs, the next half-stage's reads,
                // the other MFMAs -- its fragments are 2 MI - 2 MFMAs old when a half-stage starts; the transform's VALU work in
                // the MFMAs' shadow, its writes last
                if constexpr ((j == 0 ? MI : 0) + nitem * NP > 0) __builtin_amdgcn_sched_group_barrier(0x100, (j == 0 ? MI : 0) + nitem * NP, 0);
                if constexpr (M16) {
#pragma unroll
                    for (int h = 0; h < nt * (BK / 16); ++h) {
                        const bool next_read = h + 1 < nt * (BK / 16) || j < NSTG - 1;
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                            if (next_read) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                    }
                } else
#pragma unroll
                for (int h = 0; h < nt * (BK / 16); ++h) {
                    const bool next_read = h + 1 < nt * (BK / 16) || j < NSTG - 1;
#pragma unroll
                    for (int m = 0; m < (M16 ? 4 : 2) * MI; ++m) {
                        if (m == 2 && next_read) __builtin_amdgcn_sched_group_barrier(0x100, MI, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if constexpr (vpm > 0) __builtin_amdgcn_sched_group_barrier(0x002, vpm, 0);
                    }
                }
                if constexpr (nitem > 0) __builtin_amdgcn_sched_group_barrier(0x200, nitem, 0);
            } else if constexpr (SKEW) {
                // scheduling: every LDS read up front (the compiler otherwise fetches fragments pair by pair, each behind
                // a full wait), then the MFMAs with the transform's VALU work in their shadow, the writes last
                constexpr int nread = nt * (BK / 16) * ((BREG ? 0 : 2) + MI) + nitem * NP;
                __builtin_amdgcn_sched_group_barrier(0x100, nread, 0);
#pragma unroll
                for (int m = 0; m < nmfma; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (vpm > 0) __builtin_amdgcn_sched_group_barrier(0x002, vpm, 0);
                }
                if constexpr (nitem > 0) __builtin_amdgcn_sched_group_barrier(0x200, nitem, 0);
            }
            // bottom: this top's transfers have landed.  (The scheduling barrier keeps the wait BEHIND the MFMAs: an asm
            // statement only orders against memory operations, and hipcc otherwise hoists it above fifteen of them.)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (BREG && j < NSTG - 1 && (!XF || SKEW)) {
                // nothing to wait for here: the next stage's own counted wait covers its fragments and raw pieces
                // (without SKEW the raw pieces are transformed right below: the full wait of the last branch stays)
            } else if constexpr (!XF && j < NSTG - 1) {
                // plain source: the patch items go straight into the next chunk's buffer, which nobody reads before the
                // chunk's last stage -- and they come from HBM, not L2 like the weights.  They were issued after the
                // weights, so "all but the newest ni" = the weights of the next stage and every earlier item: an item
                // gets two stages to land instead of one (the full wait cost the input-gradient kernels ~8 %).
                if (ni == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (ni == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if constexpr (!SKEW && nitem > 0) {
                if (more) {
#pragma unroll
                    for (int b = 0; b < nitem; ++b) {
                        if (256 * (jx * BATCH + b) + wave * 64 < XSLOTS) {
                            raw_read(b, rawR, rr[b]);
                            xform_store(jx * BATCH + b, rr[b], Anext, true, rawR);
                        }
                    }
                }
            }
            if constexpr (!BREG || j == NSTG - 1) {            // BREG: the waves only share the patch, which swaps once per chunk
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            asm volatile("" ::: "memory");
        };
        stage(ic<0>{}); stage(ic<1>{}); stage(ic<2>{}); stage(ic<3>{}); stage(ic<4>{});
        if constexpr (NSTG > 5) { stage(ic<5>{}); stage(ic<6>{}); stage(ic<7>{}); stage(ic<8>{}); }
    };
    if constexpr (PEEL) {
        for (int c = 0; c + 1 < nchunk; ++c) chunk(c, std::true_type{});
        chunk(nchunk - 1, std::false_type{});
    } else {
        for (int c = 0; c < nchunk; ++c) chunk(c, c + 1 < nchunk);
    }

    // ---- epilogue: bf16 outputs (NHWC), optional two-destination split, BN-statistics partials.
    // The accumulator layout has lanes along channels and registers along pixels, so a direct store would be
    // 2 bytes per lane (64 store instructions per 32x64 sub-tile).  Each wave instead transposes one 32-pixel
    // sub-tile at a time through its own LDS scratch (rows padded to 144 B: the two half-waves hit disjoint
    // banks) and writes 16 bytes per lane, 128 contiguous bytes per pixel.
    const int C1 = a.Cout - a.C0;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) elt_t bf16x2;
    f32x2 s1v[2] = {{0.f, 0.f}, {0.f, 0.f}}, s2v[2] = {{0.f, 0.f}, {0.f, 0.f}};    // two pixel rows at a time (packed f32 math)
    float s1p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // M16: from the stored pieces
    auto stat_piece = [&](const bf16x8 v8, bool ok) {      // the lane's 8 channels (lane & 7) of one pixel: statistics see the stored values
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float f = ok ? (float)v8[q] : 0.f;
            s1p[q] += f; s2p[q] += f * f;
        }
    };
    // BNS (round 4): this launch is the input gradient that PRODUCES da of a BatchNorm + ReLU layer; the layer's backward needs
    // sum(da mask) and sum(da mask y) per channel (mask = y scale + shift > 0).  The lane holds the stored da piece anyway: it loads
    // the 16 bytes of y beside it and forms the two sums in place of the forward statistics -- same rows, same reduction below --
    // instead of a reduce pass reading both tensors again (4 B per element at 5.2 TB/s).
    float bsc[8], bsh[8];
    if constexpr (BNS) {
        const long go = a.bn_gN > 0 ? (long)(img / a.bn_gN) * a.bn_gstride : 0;
        const float* ps = a.bnsc + go + n0 + wn * 64 + (lane & 7) * 8;
        const float* pb = a.bnsh + go + n0 + wn * 64 + (lane & 7) * 8;
        const f32x4 s0 = *(const f32x4*)ps, s1_ = *(const f32x4*)(ps + 4), b0 = *(const f32x4*)pb, b1 = *(const f32x4*)(pb + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { bsc[q] = s0[q]; bsc[4 + q] = s1_[q]; bsh[q] = b0[q]; bsh[4 + q] = b1[q]; }
    }
    auto bns_piece = [&](const bf16x8 v8, const bf16x8 y8, bool ok) {      // (v8: da as stored; y8: the layer's pre-BatchNorm output)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float yf = (float)y8[q];
            const float dz = (ok && yf * bsc[q] + bsh[q] > 0.f) ? (float)v8[q] : 0.f;
            s1p[q] += dz; s2p[q] += dz * yf;
        }
    };
    constexpr int EPITCH = 144;
    char* ep = smem + wave * (32 * EPITCH);         // the A patches are dead after the last barrier
    const bool full = !LIN && y0 + TH <= a.Ho && x0 + TW <= a.Wo;     // interior tile: no per-pixel masks
    // LIN: (image of the pass, row, column) of the position the lane stores next -- P0 + 128 wm + (lane >> 3), then 8 further per
    // store step (RP > 8: one carry) -- and, for the statistics taken from the accumulators, the sub-tile's valid positions as a mask
    int e_il = 0, e_r = 0, e_c = 0;
    unsigned vmask = 0xffffffffu;
    auto lin_pos = [&](int P, int& il, int& r, int& c) { il = P / IS; const int rem = P - il * IS; r = rem / RP; c = rem - r * RP; };
    if constexpr (LIN) lin_pos(P0 + wm * MI * 32 + (lane >> 3), e_il, e_r, e_c);
    // sub-tile i -> LDS scratch (bf16) + statistics; MASK = false on interior tiles (a wave-uniform branch, not selects)
    auto park = [&](int i, auto mask_c, auto stat_c) {
        constexpr bool MASK = decltype(mask_c)::value, STAT = decltype(stat_c)::value;
        const int oyb = y0 + wm * SR * MI + SR * i;
        if constexpr (M16) {     // D[channel][pixel]: registers 0..3 of tile (half h, group g) = channels 16 g + 4 (l >> 4) + 0..3 of pixel 16 h + (l & 15)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = acc16[M16 ? i : 0][h][g];
                    bf16x4 h4;
                    h4[0] = (elt_t)v[0]; h4[1] = (elt_t)v[1]; h4[2] = (elt_t)v[2]; h4[3] = (elt_t)v[3];
                    *(bf16x4*)(ep + (16 * h + (lane & 15)) * EPITCH + (16 * g + 4 * (lane >> 4)) * 2) = h4;
                }
        } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {       // registers r, r+1 are pixel rows row, row+1 of the sub-tile
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const f32x2 v = {acc[M16 ? 0 : i][j][r], acc[M16 ? 0 : i][j][r + 1]};
                const bf16x2 h = __builtin_convertvector(v, bf16x2);
                *(elt_t*)(ep + row * EPITCH + (j * 32 + l31) * 2) = h[0];
                *(elt_t*)(ep + (row + 1) * EPITCH + (j * 32 + l31) * 2) = h[1];
                if (!STAT) continue;
                f32x2 f = __builtin_convertvector(h, f32x2);               // statistics see the stored values
                if (MASK && LIN) {
                    if (!((vmask >> row) & 1u)) f[0] = 0.f;
                    if (!((vmask >> (row + 1)) & 1u)) f[1] = 0.f;
                } else if (MASK) {
                    const int oy0 = oyb + (TW == 16 ? (row >> 4) : 0), ox0 = x0 + (row & (TW - 1));
                    const int oy1 = oyb + (TW == 16 ? ((row + 1) >> 4) : 0), ox1 = x0 + ((row + 1) & (TW - 1));
                    if (!(oy0 < a.Ho && ox0 < a.Wo)) f[0] = 0.f;
                    if (!(oy1 < a.Ho && ox1 < a.Wo)) f[1] = 0.f;
                }
                s1v[j] += f;
                s2v[j] += f * f;
            }
        }
        }
    };
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int oyb = y0 + wm * SR * MI + SR * i;
        if constexpr (LIN && !M16) {
            if (a.stat) {
                int il, r, c;
                lin_pos(P0 + wm * MI * 32 + 32 * i + l31, il, r, c);
                vmask = (unsigned)__builtin_amdgcn_ballot_w64(il < nimg && r < a.Ho && c < LINW);
            }
        }
        if constexpr (M16) park(i, std::false_type{}, std::false_type{});
        else {
        if (!a.stat) park(i, std::false_type{}, std::false_type{});        // input-gradient: no statistics
        else if (full) park(i, std::false_type{}, std::true_type{});
        else park(i, std::true_type{}, std::true_type{});
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (full && n0 + wn * 64 + 64 <= a.C0) {
            // interior tile, one destination (wave-uniform): four plain stores off one base -- the general path below costs
            // three nested exec-mask regions and a 64-bit address chain per store
            const int r0 = lane >> 3, ch = lane & 7;
            elt_t* base = (elt_t*)a.out0 + (((long)img * a.Ho + oyb) * a.Wo + x0) * a.C0 + n0 + wn * 64 + ch * 8;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = r0 + 8 * t;
                const bf16x8 v8 = *(const bf16x8*)(ep + row * EPITCH + ch * 16);
                const long off = TW == 16 ? ((long)(row >> 4) * a.Wo + (row & 15)) * a.C0 : (long)row * a.C0;
                *(bf16x8*)(base + off) = v8;
                if constexpr (BNS) bns_piece(v8, *(const bf16x8*)((const elt_t*)a.bny + (base - (elt_t*)a.out0) + off), true);
                else if constexpr (M16) { if (a.stat) stat_piece(v8, true); }
            }
        } else
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int idx = lane + 64 * t, row = idx >> 3, ch = idx & 7;
            const bf16x8 v8 = *(const bf16x8*)(ep + row * EPITCH + ch * 16);
            const int oy = LIN ? e_r : oyb + (TW == 16 ? (row >> 4) : 0), ox = LIN ? e_c : x0 + (row & (TW - 1));
            const int oimg = LIN ? img + e_il : img;
            const bool inb = LIN ? (e_il < nimg && e_r < a.Ho && e_c < LINW) : (oy < a.Ho && ox < a.Wo);
            if constexpr (LIN) {     // the next step's position: 8 further
                e_c += 8;
                if (e_c >= RP) { e_c -= RP; if (++e_r > a.Ho) { e_r = 0; ++e_il; } }
            }
            const int col = n0 + wn * 64 + ch * 8;
            if constexpr (BNS) {     // (single destination: the launcher admits nothing else)
                const bool ok = inb && col < a.C0;
                const long eo = ok ? (((long)oimg * a.Ho + oy) * a.Wo + ox) * a.C0 + col : 0;
                bns_piece(v8, *(const bf16x8*)((const elt_t*)a.bny + eo), ok);
            } else if constexpr (M16) { if (a.stat) stat_piece(v8, inb); }
            if (inb) {
                if (col < a.C0) {
                    *(bf16x8*)((elt_t*)a.out0 + (((long)oimg * a.Ho + oy) * a.Wo + ox) * a.C0 + col) = v8;
                } else {
                    const int y1 = oy - a.o1y, x1 = ox - a.o1x;
                    if (y1 >= 0 && y1 < a.H1 && x1 >= 0 && x1 < a.W1)
                        *(bf16x8*)((elt_t*)a.out1 + (((long)oimg * a.H1 + y1) * a.W1 + x1) * C1 + (col - a.C0)) = v8;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    float s1[2] = {s1v[0][0] + s1v[0][1], s1v[1][0] + s1v[1][1]}, s2[2] = {s2v[0][0] + s2v[0][1], s2v[1][0] + s2v[1][1]};
    if (a.stat) {
        float* red = (float*)(smem + 4 * 32 * EPITCH);   // [WM][2][BN], behind the waves' transpose scratch
        if constexpr (M16) {                        // lanes with equal lane & 7 hold the same 8 channels
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) { s1p[q] += __shfl_xor(s1p[q], o); s2p[q] += __shfl_xor(s2p[q], o); }
                if (lane < 8) {
                    red[(wm * 2 + 0) * BN + wn * 64 + lane * 8 + q] = s1p[q];
                    red[(wm * 2 + 1) * BN + wn * 64 + lane * 8 + q] = s2p[q];
                }
            }
        } else
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

thread_local int g_last_variant = 0;    // (per calling thread) TH<<24 | TW<<16 | BN<<8 | MI<<4 | NT<<2 | POOL<<1 | XF of the last launch (tests: ustrun_debug_last_conv_variant)
                                        // (linear tiles: bit 30 | W << 16 | ...)

// linear tiles (LINW): passes and tiles of a launch.  A pass = the images that share BatchNorm constants (the sources' gN, or the
// gN of the BatchNorm-backward sums the launch forms; the whole batch without either); tiles never cross one.
struct LinPlan { int gN, tpp, mtiles, last_rows; };
static LinPlan lin_plan(const IgemmArgs& a) {
    int g = a.pass_gN > 0 ? a.pass_gN : 0;
    for (int i = 0; i < a.nsrc && !g; ++i)
        if (a.src[i].gN > 0) g = a.src[i].gN;
    if (!g && a.bn_gN > 0) g = a.bn_gN;
    if (g <= 0 || g > a.N) g = a.N;
    const long IS = (long)(a.Ho + 1) * (a.Wo + 1);
    const int tpp = (int)cdiv(g * IS, 256L), full = a.N / g, tailN = a.N - full * g, tail = (int)cdiv(tailN * IS, 256L);
    return {g, tpp, full * tpp + tail, tailN ? tail : (full > 1 ? tpp : -1)};
}

template <int TH, int TW, int BN, int BK, int MI, bool POOL, int NT, bool XF, int DIL = 1, bool BREG = false, bool M16 = false, bool BNS = false,
          int LINW = 0>
int launch_xf(const IgemmArgs& a, hipStream_t st) {
    g_last_variant = (LINW ? (1 << 30 | LINW << 16) : (TH << 24 | TW << 16)) | BN << 8 | (M16 ? 0x80 : 0) | MI << 4 | NT << 2 | (POOL ? 2 : 0) | (XF ? 1 : 0);
    const LinPlan lp = LINW ? lin_plan(a) : LinPlan{0, 0, 0, -1};
    const int tx = LINW ? lp.tpp : cdiv(a.Wb, TW), ty = LINW ? lp.gN : cdiv(a.Hb, TH), nt = a.Cout / BN;
    constexpr int DSLOTS = ((LINW ? TH * TW + 2 * (LINW + 1) + 2 : (TH + 2 * DIL) * (TW + 2 * DIL)) * 5 + 63) / 64 * 64;
    constexpr int AIT = (DSLOTS + 255) / 256, NSTG = (9 + NT - 1) / NT, BATCH = (AIT + NSTG - 2) / (NSTG - 1);
    const size_t rawb = XF ? (size_t)BATCH * (POOL ? 4 : 1) * 4096 : 0;
    const size_t fixed = 2 * (size_t)(BREG ? AIT * 256 : DSLOTS) * 16 + (BREG ? 0 : 2 * NT * (size_t)(BK / 8) * BN * 16) + 512;
    const size_t lds = fixed + (XF && fixed + 2 * rawb <= 81920 ? 2 : 1) * rawb;
    dim3 grid((LINW ? lp.mtiles : a.N * ty * tx) * nt), block(256);
    // above the 64 KB default (the dilation-4 patch pair: 93 KB; the 16 x 16 x 128 transforming tile: 66 KB): raise the limit
    // whatever the dilation is (the un-dilated 16 x 16 tile used to launch without the attribute)
    if (lds > 64 * 1024)
        USTRUN_TRY(ensure_dynamic_lds((const void*)conv3x3_halo_bf16_kernel<TH, TW, BN, BK, MI, POOL, NT, XF, DIL, BREG, M16, BNS, LINW>, (int)lds, "conv3x3_halo_bf16"));
    hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<TH, TW, BN, BK, MI, POOL, NT, XF, DIL, BREG, M16, BNS, LINW>), grid, block, lds, st, a, tx, ty, nt);
    USTRUN_LAUNCH_CHECK("conv3x3_halo_bf16");
    return 0;
}

// XF: some source needs arithmetic on load (BatchNorm affine, ReLU, pooling); plain sources (dY, ConvTranspose outputs)
// then pass through the same path with scale 1, shift 0, floor -inf
template <int TH, int TW, int BN, int BK, int MI, bool POOL, int NT = 1>
int launch_cfg(const IgemmArgs& a, hipStream_t st) {
    bool xf = POOL;
    for (int i = 0; i < a.nsrc; ++i) xf |= a.src[i].scale != nullptr || a.src[i].relu != 0;
    if constexpr (POOL) return launch_xf<TH, TW, BN, BK, MI, POOL, NT, true>(a, st);
    else {
        {                                              // weights through registers, one barrier per chunk (debug flag bit 3: off)
            // the 16x16x32 build (round 4): measured ahead on the input gradients of every layer (profiles/r04_ab_halo_m16.log:
            // 512 -> 512 at 32 x 32 0.220 -> 0.202 ms, 1024 -> 1024 0.217 -> 0.201, 128 -> 128 -1.5 %): the default for plain
            // sources on the 256-pixel tiles.  ustrun_debug_flags bit 15 (32768): every plain-source tile on it (A/B runs of
            // the smaller tiles); bit 21 (2097152): none.
            {
                const bool m16 = !(g_debug_flags & 8) && !(g_debug_flags & 2097152) && ((g_debug_flags & 32768) || (!xf && MI == 4 && NT == 1));
                if constexpr (MI == 4 && NT == 1 && BN == 128) {      // (halo_bnsum_supported admits these two tiles only)
                    if (a.bny) {
                        USTRUN_CHECK(m16 && !xf && a.stat && a.bnsc && a.bnsh, "conv3x3_halo: BatchNorm-backward sums need a plain source on the 16x16x32 build");
                        return launch_xf<TH, TW, BN, BK, MI, POOL, NT, false, 1, true, true, true>(a, st);
                    }
                }
                if (m16 && !xf)      // (the transforming variants on it sit 160 registers past the file: not built)
                    return launch_xf<TH, TW, BN, BK, MI, POOL, NT, false, 1, true, true>(a, st);
            }
            if (!(g_debug_flags & 8))
                return xf ? launch_xf<TH, TW, BN, BK, MI, POOL, NT, true, 1, true>(a, st) : launch_xf<TH, TW, BN, BK, MI, POOL, NT, false, 1, true>(a, st);
        }
        return xf ? launch_xf<TH, TW, BN, BK, MI, POOL, NT, true>(a, st) : launch_xf<TH, TW, BN, BK, MI, POOL, NT, false>(a, st);
    }
}

}  // namespace

int halo_last_variant() { return g_last_variant; }
static int halo_dilation(const IgemmArgs& a) { return a.dstep < 0 ? -a.dstep : a.dstep; }
void set_last_variant(int v) { g_last_variant = v; }

// stat rows the halo kernel writes for an N x H x W output (one per 8 x 16 pixels)
int halo_stat_rows(int N, int H, int W) { return N * cdiv(H, 8) * cdiv(W, 16); }

// can this conv3x3-shaped problem run on the halo kernel?
bool halo_supported(const IgemmArgs& a) {
    if (a.nseg != 9 || a.nz != 1 || a.s_in != 1 || a.s_out != 1 || a.segw != 3) return false;
    const int dil = a.dstep < 0 ? -a.dstep : a.dstep;
    if (a.d0 != -a.dstep || !(dil == 1 || dil == 2 || dil == 4)) return false;      // (other rates: the generic kernel)
    bool pool = false;
    if (a.out_esz != 2 || (a.C0 & 7) || ((a.Cout - a.C0) & 7) || a.bias) return false;     // (3x3 convs here carry no bias)
    for (int i = 0; i < a.nsrc; ++i) {
        if (a.src[i].sC != 1 || (a.src[i].C & 7) || a.src[i].esz != 2) return false;
        pool |= a.src[i].pool != 0;
    }
    const int BK = 32;
    if (a.Cin % BK || a.Cout % 64) return false;
    if (a.nsrc == 2 && (a.src[0].C % BK)) return false;
    if (pool && (a.nsrc != 1 || a.Cout % 128)) return false;
    if (a.Hb < 4 || a.Wb < 8) return false;      // tiny extents: the generic kernel wastes less
    if (dil > 1 && (pool || a.nsrc != 1)) return false;
    return true;
}

int halo_tile128(const IgemmArgs& a);
static double halo_rect_score(const IgemmArgs& a);

// Linear tiles (LINW builds: 18, 24, 36, 72-pixel rows): the row length if this launch runs on them, else 0.  They serve
// 128-column outputs of un-pooled, un-dilated sources when (a) the rectangular tiles would waste more of their MFMAs on padding --
// useful fraction x the tiles' measured relative rates, as in halo_tile128 -- and (b) the launch still has a block per CU.
// (ustrun_debug_flags2 bit 0: off, for A/B runs)
int halo_linear_w(const IgemmArgs& a) {
    if ((g_debug_flags2 & 1) || (g_debug_flags & (8 | 2097152 | 32768)) || ((g_debug_flags >> 10) & 3)) return 0;
    if (!halo_supported(a) || halo_dilation(a) != 1 || a.Cout % 128 || a.Hb != a.Ho || a.Wb != a.Wo) return 0;
    const int w = a.Wo;
    if (!(w == 18 || w == 24 || w == 36 || w == 72)) return 0;
    int g = a.pass_gN > 0 ? a.pass_gN : 0;
    for (int i = 0; i < a.nsrc; ++i) {
        if (a.src[i].pool) return 0;
        if (a.src[i].gN > 0) { if (g && g != a.src[i].gN) return 0; g = a.src[i].gN; }
    }
    if (a.bny && a.bn_gN > 0 && g && g != a.bn_gN) return 0;
    const LinPlan lp = lin_plan(a);
    for (int i = 0; i < a.nsrc; ++i)
        if ((long)lp.gN * a.src[i].sN * 2 >= (1L << 31) - 4096) return 0;          // 32-bit byte offsets inside a pass
    if ((long)lp.mtiles * (a.Cout / 128) < 160) return 0;
    const double lin = (double)a.N * a.Ho * a.Wo / (lp.mtiles * 256.0);
    return lin > 1.05 * halo_rect_score(a) ? w : 0;
}
// statistics rows of the last pass of a linear-tile launch (-1: not one, or a single pass)
int halo_linear_last_pass_rows(const IgemmArgs& a) { return halo_linear_w(a) ? lin_plan(a).last_rows : -1; }

// can this input-gradient launch also form the BatchNorm-backward sums of the layer whose da it writes?  (one plain source, one
// destination, the two 256-pixel x 128-channel tiles on the 16x16x32 build)
bool halo_bnsum_supported(const IgemmArgs& a) {
    if (!halo_supported(a) || a.nsrc != 1 || a.out1 || a.C0 != a.Cout || a.Cout % 128) return false;
    if (a.src[0].scale || a.src[0].relu || a.src[0].pool) return false;
    if ((g_debug_flags & (8 | 2097152))) return false;
    if (halo_dilation(a) != 1) return a.bn_gN == 0;      // (round 6) the dilated tiles of DeepLabV2-ResNet: launch_dilated's 16x16x32 builds
    if (halo_linear_w(a)) return true;
    const int t = halo_tile128(a);
    return t == 0 || t == 1;
}

// 16-row tiles (256 px x 128 ch per block) read fewer LDS/weight bytes per flop but need >= 2 blocks per CU
bool halo_tall_tile(const IgemmArgs& a) {
    return (long)a.N * cdiv(a.Hb, 16) * cdiv(a.Wb, 16) * (a.Cout / 128) >= 512;
}

// Tile of the 128-column, un-pooled, un-dilated case: 0 = 8 x 32 px (MI 4), 1 = 16 x 16 px (MI 4), 2 = 8 x 16 px (MI 2, 128 columns),
// 3 = 8 x 16 px, 64 columns (less than one block per CU).  ustrun_debug_flags bits 10-11 force 0..2 (+1) for A/B runs.
// useful MFMA fraction x relative rate of the rectangular tile conv3x3_halo_launch_bf16 would pick (128-column, un-pooled case)
static double halo_rect_score(const IgemmArgs& a) {
    const double h8 = cdiv(a.Hb, 8) * 8.0, h16 = cdiv(a.Hb, 16) * 16.0, w16 = cdiv(a.Wb, 16) * 16.0, w32 = cdiv(a.Wb, 32) * 32.0;
    const double hw = (double)a.Hb * a.Wb;
    switch (halo_tile128(a)) {
        case 0: return hw / (h8 * w32);
        case 1: return 0.93 * hw / (h16 * w16);
        default: return 0.80 * hw / (h8 * w16);
    }
}

int halo_tile128(const IgemmArgs& a) {
    const int force = (g_debug_flags >> 10) & 3;
    const bool wide = a.Wb >= 32;
    if (force) return (force == 1 && !wide) ? 1 : force - 1;
    if (halo_tall_tile(a)) {
        // enough blocks for any tile: the one that wastes the fewest MFMAs on padding, weighted by the tiles' measured relative
        // rates on full tiles (8 x 32: 1, 16 x 16: 0.93, 8 x 16 with two sub-tiles per wave: 0.80 -- profiles/r03_ab_halo_tiles.log:
        // 512 -> 512 at 48 x 48: 0.346 -> 0.290 ms on 16 x 16 tiles; 1024 -> 1024 at 18 x 18 and 24 x 24: -9 % on 8 x 16).  Maps
        // whose sides are multiples of 32 (every level of the 256 x 256 workloads) keep the 8 x 32 tile.
        const double h8 = cdiv(a.Hb, 8) * 8.0, h16 = cdiv(a.Hb, 16) * 16.0, w16 = cdiv(a.Wb, 16) * 16.0, w32 = cdiv(a.Wb, 32) * 32.0;
        const double s0 = wide ? 1.0 / (h8 * w32) : 0.0, s1 = 0.93 / (h16 * w16), s2 = 0.80 / (h8 * w16);
        return (s0 >= s1 && s0 >= s2) ? 0 : (s1 >= s2 ? 1 : 2);
    }
    if ((long)a.N * cdiv(a.Hb, 8) * cdiv(a.Wb, 16) * (a.Cout / 128) < 256) return 3;      // less than one block per CU
    return 2;
}

// BatchNorm-statistics rows written by the configuration conv3x3_halo_launch_bf16 picks
int halo_stat_rows_used(const IgemmArgs& a) {
    if (halo_linear_w(a)) return lin_plan(a).mtiles;          // one row per 256-position tile
    if (halo_dilation(a) == 4 && a.Cout % 128 == 0) return a.N * cdiv(a.Hb, 16) * 2 * cdiv(a.Wb, 16);       // 16 x 16 tiles, two rows each
    if (halo_dilation(a) > 1) return a.N * cdiv(a.Hb, 8) * cdiv(a.Wb, 16);
    bool pool = false;
    for (int i = 0; i < a.nsrc; ++i) pool |= a.src[i].pool != 0;
    const bool wide = a.Wb >= 32;
    if (pool) return a.N * cdiv(a.Hb, 8) * cdiv(a.Wb, 16);
    if (a.Cout % 128 == 0) {
        const int t = halo_tile128(a);
        return t == 0 ? a.N * cdiv(a.Hb, 8) * cdiv(a.Wb, 32) : t == 1 ? a.N * cdiv(a.Hb, 16) * 2 * cdiv(a.Wb, 16) : a.N * cdiv(a.Hb, 8) * cdiv(a.Wb, 16);
    }
    if (wide && (g_debug_flags & 8192) && a.Hb >= 16) return a.N * cdiv(a.Hb, 16) * 2 * cdiv(a.Wb, 32);
    return wide ? a.N * cdiv(a.Hb, 8) * cdiv(a.Wb, 32) : a.N * cdiv(a.Hb, 16) * 2 * cdiv(a.Wb, 16);
}

// dilated 3x3 (DeepLabV2-ResNet layer3 / layer4): 8 x 16-pixel tiles keep the (8 + 2d) x (16 + 2d) patch pair small at rate 2
template <int DIL>
static int launch_dilated(const IgemmArgs& a, hipStream_t st) {
    bool xf = false;
    for (int i = 0; i < a.nsrc; ++i) xf |= a.src[i].scale != nullptr || a.src[i].relu != 0;
    const bool breg = !(g_debug_flags & 8);          // weights through registers, one barrier per chunk (round 3)
    if (a.bny) {        // input gradient + BatchNorm-backward sums (halo_bnsum_supported): the 16x16x32 build of the rate's tile
        USTRUN_CHECK(breg && !xf && a.stat && a.bnsc && a.bnsh && a.Cout % 128 == 0, "conv3x3_halo: BatchNorm-backward sums need a plain source on the 16x16x32 build");
        if (DIL == 4) return launch_xf<16, 16, 128, 32, 4, false, 1, false, DIL, true, true, true>(a, st);
        return launch_xf<8, 16, 128, 32, 2, false, 2, false, DIL, true, true, true>(a, st);
    }
    if (a.Cout % 128 == 0) {
        // rate 4: the (8 + 8) x (16 + 8) patch of an 8 x 16 tile is three times its output and its pair of buffers leaves room
        // for ONE block per CU anyway -- a 16 x 16 tile (24 x 24 patch: 2.25 x) with the 4 x 2 wave tile does twice the MFMA
        // work per stage in that one block: 0.49 -> 0.39 ms on layer4's 512 -> 512 convolutions (627 -> 786 TF/s).  Rate 2
        // keeps the small tile (two blocks per CU; the large one measured 0.089 -> 0.108 ms there).
        if (DIL == 4) {
            if (breg) return xf ? launch_xf<16, 16, 128, 32, 4, false, 1, true, DIL, true>(a, st) : launch_xf<16, 16, 128, 32, 4, false, 1, false, DIL, true>(a, st);
            return xf ? launch_xf<16, 16, 128, 32, 4, false, 1, true, DIL>(a, st) : launch_xf<16, 16, 128, 32, 4, false, 1, false, DIL>(a, st);
        }
        if (breg) return xf ? launch_xf<8, 16, 128, 32, 2, false, 2, true, DIL, true>(a, st) : launch_xf<8, 16, 128, 32, 2, false, 2, false, DIL, true>(a, st);
        return xf ? launch_xf<8, 16, 128, 32, 2, false, 2, true, DIL>(a, st) : launch_xf<8, 16, 128, 32, 2, false, 2, false, DIL>(a, st);
    }
    if (breg) return xf ? launch_xf<8, 16, 64, 32, 1, false, 2, true, DIL, true>(a, st) : launch_xf<8, 16, 64, 32, 1, false, 2, false, DIL, true>(a, st);
    return xf ? launch_xf<8, 16, 64, 32, 1, false, 2, true, DIL>(a, st) : launch_xf<8, 16, 64, 32, 1, false, 2, false, DIL>(a, st);
}

template <int W>
static int launch_linear(const IgemmArgs& a, hipStream_t st) {
    bool xf = false;
    for (int i = 0; i < a.nsrc; ++i) xf |= a.src[i].scale != nullptr || a.src[i].relu != 0;
    if (a.bny) {
        USTRUN_CHECK(!xf && a.stat && a.bnsc && a.bnsh, "conv3x3_halo: BatchNorm-backward sums need a plain source");
        return launch_xf<8, 32, 128, 32, 4, false, 1, false, 1, true, true, true, W>(a, st);
    }
    return xf ? launch_xf<8, 32, 128, 32, 4, false, 1, true, 1, true, false, false, W>(a, st)
              : launch_xf<8, 32, 128, 32, 4, false, 1, false, 1, true, true, false, W>(a, st);
}

int conv3x3_halo_launch_bf16(const IgemmArgs& a, hipStream_t st) {
    if (halo_dilation(a) == 2) return launch_dilated<2>(a, st);
    if (halo_dilation(a) == 4) return launch_dilated<4>(a, st);
    switch (halo_linear_w(a)) {
        case 18: return launch_linear<18>(a, st);
        case 24: return launch_linear<24>(a, st);
        case 36: return launch_linear<36>(a, st);
        case 72: return launch_linear<72>(a, st);
        default: break;
    }
    bool pool = false;
    for (int i = 0; i < a.nsrc; ++i) pool |= a.src[i].pool != 0;
    const bool wide = a.Wb >= 32;                 // 32-pixel rows: conflict-free A fragment reads
    // MI = 2 tiles run two taps per barrier (16 MFMAs per wave and stage, like the MI = 4 tiles)
    if (pool) return launch_cfg<8, 16, 128, 32, 2, true, 2>(a, st);
    if (a.Cout % 128 == 0) {
        // 256 px x 128 ch per block, wave tile 128 px x 64 ch (tiles 0, 1); less than one block per CU (the batch-1 forward,
        // validation at test_bs 1): twice the blocks at 64 channels each (tile 3: 20-25 % faster per layer at batch 1)
        switch (halo_tile128(a)) {
            case 0: return launch_cfg<8, 32, 128, 32, 4, false>(a, st);
            case 1: return launch_cfg<16, 16, 128, 32, 4, false>(a, st);
            case 3: return launch_cfg<8, 16, 64, 32, 1, false, 2>(a, st);
            default: return launch_cfg<8, 16, 128, 32, 2, false, 2>(a, st);
        }
    }
    // (ustrun_debug_flags bit 13 = 8192: the 512-pixel x 64-channel tile, wave tile 128 px x 64 ch, one block per CU -- A/B runs)
    if (wide && (g_debug_flags & 8192) && a.Hb >= 16) return launch_cfg<16, 32, 64, 32, 4, false, 1>(a, st);
    if (wide) return launch_cfg<8, 32, 64, 32, 2, false, 2>(a, st);
    return launch_cfg<16, 16, 64, 32, 2, false, 2>(a, st);
}

}  // namespace ustrun
