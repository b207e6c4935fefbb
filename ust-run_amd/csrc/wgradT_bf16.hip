// wgradT_bf16.hip -- ConvTranspose2d(k=2, s=2) weight gradient on the bf16 matrix cores (the autograd backward
// of nn.ConvTranspose2d in Up, reference networks/unet_parts.py:50-52):
//
//   dW[tap][ci][co] = sum_m act(a[m][ci]) * du[hi(m, tap)][co]        hi(m, tap) = 4m - 2x + 2*dy*W + dx
//
// One GEMM over the low-resolution pixels m with the four taps folded into the COLUMN axis: a block owns
// 128 ci x (4 taps x 32 co) columns and a contiguous range of pixels (split-K over space), 64 pixels per
// stage, double-buffered.  Both operands stay pixel-major in LDS ([pixel][channel], 64-byte segments XORed by
// the pixel index) and the fragments come from the transposing LDS read, as in wgrad_bf16.hip.  The activation
// tile goes global -> registers -> [BatchNorm affine + ReLU, f32] -> bf16 -> LDS; the du tile is a pure copy and
// is fetched by LDS-DMA (the swizzle is applied on the global side: every lane fetches the 16 bytes that belong
// at its LDS position).  The du fragments are gathered so that the 32 accumulator lanes are (8 co) x (4 taps):
// the f32 slab comes out directly in the torch [Cin][Cout][2][2] layout with fully coalesced stores and is summed
// in fixed order by the streaming reduce.  The bias gradient rides along: blocks of the first ci tile add one
// MFMA with an all-ones A fragment per du fragment, whose result is the column sum of du.
#include "common.h"
#include "loader.h"

namespace ustrun {
namespace {

typedef __attribute__((ext_vector_type(8))) elt_t bf16x8;
typedef __attribute__((ext_vector_type(4))) elt_t bf16x4;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int TM = 128, TN = 128, KP = 64, RB = 256;   // tile, pixels per stage, LDS row pitch (bytes)

// rows k0 + 8*(l>>5) + {0..3 | 4..7}, columns col0 + 16*((l>>4)&1) + 4*(l&3) .. +3, delivered column-major
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int k0, int col0, int lane) {
    const int q = (lane & 15) >> 2, p = lane & 3;
    const int colb = (col0 + 16 * ((lane >> 4) & 1) + 4 * p) * 2;
    const int r0 = k0 + 8 * (lane >> 5) + q, r1 = r0 + 4;      // r1 & 3 == r0 & 3
    const int off = colb ^ ((r0 & 3) << 6);
    const bf16x4 lo = USTRUN_DS_READ_TR16((lds_bf16x4*)(tile + r0 * RB + off));
    const bf16x4 hi = USTRUN_DS_READ_TR16((lds_bf16x4*)(tile + r1 * RB + off));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

// du fragment of 8-co group t: lane (q, p, g) reads row q, tap p, channels 8t + 4g .. +3 of the [tap][32 co] row,
// so after the transpose lane 16g + 4*tap + c of the MFMA holds column (co = 8t + 4g + c, tap)
__device__ __forceinline__ bf16x8 tr_frag_du(const char* tile, int k0, int t, int lane) {
    const int q = (lane & 15) >> 2, p = lane & 3, g = (lane >> 4) & 1;
    const int r0 = k0 + 8 * (lane >> 5) + q, r1 = r0 + 4;
    const int off = ((p ^ (r0 & 3)) << 6) + 16 * t + 8 * g;
    const bf16x4 lo = USTRUN_DS_READ_TR16((lds_bf16x4*)(tile + r0 * RB + off));
    const bf16x4 hi = USTRUN_DS_READ_TR16((lds_bf16x4*)(tile + r1 * RB + off));
    bf16x8 f;
    f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
    f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
    return f;
}

__device__ __forceinline__ int wrap_add(int x, int inc, int W, float invW) {   // (x + inc) mod W, x < W < 2^15, inc <= 64
    const int v = x + inc;
    const int q = (int)(((float)v + 0.5f) * invW);
    int r = v - q * W;
    if (r < 0) r += W;
    if (r >= W) r -= W;
    return r;
}

// grid = (ci tiles * column tiles, ksplit)
__global__ __launch_bounds__(256, 2) void wgradT_bf16_kernel(const WgradArgs a, const int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                      // 2 x [KP][TM] bf16
    char* Bs = smem + 2 * KP * RB;        // 2 x [KP][TN] bf16

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: LDS-DMA bases and role tests stay scalar
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware order (1-D grid, workgroups go round-robin over the 8 XCDs): every XCD takes a contiguous range of the
    // (slice-major, tile-minor) order, so all (ci, column) tiles of one pixel slice share one XCD's L2: the slice's
    // activations and du come from HBM once instead of once per tile (the kernel is HBM-bound: 5.7 TB/s measured).
    const int nblk = gridDim.x, tiles = nblk / a.ksplit;
    int lin;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = blockIdx.x % 8, jj = blockIdx.x / 8;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + jj;
    }
    const int ks = lin / tiles, tile = lin - ks * tiles;
    const int mtile = tile / ntn, ntile = tile % ntn;
    const int ci0 = mtile * TM, co0 = ntile * 32;
    const long kbeg = (long)ks * a.kchunk;
    const long kend = (kbeg + a.kchunk < a.M) ? kbeg + a.kchunk : a.M;
    const int W = a.Wb;
    const float invW = 1.f / (float)W;

    // ---- A items: pixel row (tid >> 4) + 16 i, 8-channel group tid & 15; the source is pixel-linear (m * sW) ----
    const SrcDev& S = a.src[0];
    const int c8 = tid & 15;
    const int cl = ci0 + 8 * c8;
    const bool aff = S.scale != nullptr;
    f32x4 asc0 = {1.f, 1.f, 1.f, 1.f}, asc1 = asc0, ash0 = {0.f, 0.f, 0.f, 0.f}, ash1 = ash0;
    int cur_grp = -1;                     // batched passes: a 64-pixel stage never straddles two passes (host check)
    const long gpix = S.gN > 0 ? (long)S.gN * a.Hb * a.Wb : 0;
    auto load_consts = [&](long k0) {
        const int grp = gpix > 0 ? (int)(k0 / gpix) : 0;
        if (aff && grp != cur_grp) {
            const long o = (long)grp * (gpix > 0 ? S.gstride : 0) + cl;
            asc0 = *(const f32x4*)(S.scale + o); asc1 = *(const f32x4*)(S.scale + o + 4);
            ash0 = *(const f32x4*)(S.shift + o); ash1 = *(const f32x4*)(S.shift + o + 4);
            cur_grp = grp;
        }
    };
    const elt_t* sp = (const elt_t*)S.ptr + cl;
    const int arow = tid >> 4;
    bf16x8 av[4];
    auto load_A = [&](long k0) {
        load_consts(k0);                  // write_A of this stage runs before the next load_A
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = k0 + arow + 16 * i;
#pragma unroll
            for (int q = 0; q < 8; ++q) av[i][q] = (elt_t)0.f;
            if (m < kend) av[i] = *(const bf16x8*)(sp + m * S.sW);
        }
    };
    auto write_A = [&](long k0, char* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = arow + 16 * i;
            bf16x8 h = av[i];
            if (aff && k0 + row < kend) {       // rows past the range stay zero
                f32x4 lo = (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]} * asc0 + ash0;
                f32x4 hi = (f32x4){(float)h[4], (float)h[5], (float)h[6], (float)h[7]} * asc1 + ash1;
                if (S.relu) { lo = relu4(lo); hi = relu4(hi); }
                h[0] = (elt_t)lo[0]; h[1] = (elt_t)lo[1]; h[2] = (elt_t)lo[2]; h[3] = (elt_t)lo[3];
                h[4] = (elt_t)hi[0]; h[5] = (elt_t)hi[1]; h[6] = (elt_t)hi[2]; h[7] = (elt_t)hi[3];
            }
            *(bf16x8*)(dst + row * RB + ((c8 * 16) ^ ((row & 3) << 6))) = h;
        }
    };
    // ---- du tile by LDS-DMA: wave-instruction i fills rows 16 i + 4 wave .. +3 (1 KB, lane-linear).  The lane at
    // LDS position `pos` of row r fetches logical 16-byte item ((pos >> 2) ^ (r & 3)) << 2 | (pos & 3).
    const int brl = lane >> 4, pos = lane & 15;
    const int item = ((((pos >> 2) ^ brl) << 2) | (pos & 3));      // (4 wave + brl) & 3 == brl
    const int tap = item >> 2, co = co0 + 8 * (item & 3);          // LDS row = [tap][32 co]
    const long tapoff = (long)(tap >> 1) * 2 * W + (tap & 1);
    const elt_t* dup = (const elt_t*)a.dy + co;
    int bx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bx[i] = (int)((kbeg + 16 * i + 4 * wave + brl) % W);
    auto dma_B = [&](long k0, char* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = k0 + 16 * i + 4 * wave + brl;
            // rows past the tensor read pixel 0: any finite value will do, their A rows are zero
            const long hp = m < a.M ? 4 * m - 2 * bx[i] + tapoff : 0;
            __builtin_amdgcn_global_load_lds((gptr_t*)(dup + hp * a.Cout), (lptr_t*)(dst + (16 * i + 4 * wave) * RB), 16, 0, 0);
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) bx[i] = wrap_add(bx[i], KP, W, invW);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool do_bias = a.bias_partials != nullptr && mtile == 0 && wm == 0;      // wave-uniform
    f32x16 accb[2];
    bf16x8 ones;
#pragma unroll
    for (int q = 0; q < 8; ++q) ones[q] = (elt_t)1.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[j][r] = 0.f;

    if (kbeg < kend) {
        dma_B(kbeg, Bs);
        load_A(kbeg);
        write_A(kbeg, As);
    }
    __syncthreads();
    int buf = 0;
#pragma unroll 1
    for (long k0 = kbeg; k0 < kend; k0 += KP) {
        const bool more = k0 + KP < kend;
        if (more) {
            advance();
            dma_B(k0 + KP, Bs + (buf ^ 1) * (KP * RB));
            load_A(k0 + KP);
        }
        const char* At = As + buf * (KP * RB);
        const char* Bt = Bs + buf * (KP * RB);
#pragma unroll
        for (int kk = 0; kk < KP / 16; ++kk) {
            const bf16x8 a0 = tr_frag(At, kk * 16, wm * 64, lane);
            const bf16x8 a1 = tr_frag(At, kk * 16, wm * 64 + 32, lane);
            const bf16x8 b0 = tr_frag_du(Bt, kk * 16, 2 * wn, lane);
            const bf16x8 b1 = tr_frag_du(Bt, kk * 16, 2 * wn + 1, lane);
            if (do_bias) {
                // this lane's 8 K entries are pixels k0 + 16 kk + 8 (lane >> 5) + 0..7: only those inside the range count
                const long rem = kend - k0 - 16 * kk - 8 * (lane >> 5);
                bf16x8 on = ones;
                if (rem < 8) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) on[q] = (elt_t)(q < rem ? 1.f : 0.f);
                }
                accb[0] = USTRUN_MFMA_32x32x16(on, b0, accb[0], 0, 0, 0);
                accb[1] = USTRUN_MFMA_32x32x16(on, b1, accb[1], 0, 0, 0);
            }
            acc[0][0] = USTRUN_MFMA_32x32x16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = USTRUN_MFMA_32x32x16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = USTRUN_MFMA_32x32x16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = USTRUN_MFMA_32x32x16(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) write_A(k0 + KP, As + (buf ^ 1) * (KP * RB));
        __syncthreads();
        buf ^= 1;
    }

    // slab in the torch layout [Cin][Cout][2][2]: rows of D are ci (registers); lane 16g + 4*tap + c is column
    // (co = co0 + 8t + 4g + c, tap) -> the 32 lanes of a row write 32 consecutive floats
    float* slab = a.partials + (long)ks * 4 * a.Cin * a.Cout;
    const int l31 = lane & 31, lh = lane >> 5;
    const int lg = l31 >> 4, ltap = (l31 >> 2) & 3, lc = l31 & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int co = co0 + 8 * (2 * wn + j) + 4 * lg + lc;
        float* o = slab + (long)co * 4 + ltap;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                o[(long)ci * a.Cout * 4] = acc[i][j][r];
            }
        if (do_bias) {      // every row of accb is the column sum; fold the four taps (lane bits 2..3)
            float v = accb[j][0];
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if (lh == 0 && ltap == 0) a.bias_partials[(long)ks * a.Cout + co] = v;
        }
    }
}


// ---- round 5: the same GEMM on larger tiles, every transfer by buffer-addressed LDS-DMA three stages deep ---------------------
// The kernel above issues the next stage's transfers with the LDS-DMA builtin, and hipcc then drains `vmcnt(0)` in front of the
// CURRENT stage's first transposing read (it cannot prove the read does not alias the transfer in flight): every 64-pixel stage
// waited out its own prefetch -- 16 MFMAs per wave between two full memory round trips, 435 TF/s on up1.up where the all-taps 3x3
// kernel runs at 1.25 PF/s.  And a 128 x 128 tile has a quarter of that kernel's flops per operand byte (4 taps on the column axis
// instead of 9 in the accumulators): at full rate it would pull 64 B per clock and CU out of L2.  Here:
//  * tile = TM ci x 256 columns (64 co x 4 taps), wave tile 64 ci x 128 columns (eight accumulators + four for the bias sums), two
//    waves per SIMD: TM = 256: 512 threads = waves 4 x 2, one block per CU; TM = 128 (up4.up): 256 threads = waves 2 x 2, two blocks;
//  * stage = 32 pixels; rows are 512 B (256 B for 128 ci) with the 64-byte segments XORed by the row's low bits, as above; the du row
//    is [co half][tap][32 co] so that a wave's fragments sit in one 256-byte half;
//  * BOTH operands arrive by `buffer_load ... lds` issued from inline asm TWO stages ahead into three stage buffers: an item is a
//    per-lane constant byte offset against a buffer resource whose base is the stage's first pixel (wave-uniform, 64-bit in SGPRs);
//    the du row of low-resolution pixel m0 + j starts (2 j + 2 W w_j) hi-resolution pixels behind that of m0, w_j = the image rows
//    the run has wrapped by pixel j -- two compares per item and stage; rows past the block's range take the out-of-range offset
//    and come back as zeros;
//  * the bottom of stage s waits with a COUNTED vmcnt for stage s + 1, applies BatchNorm + ReLU to it IN PLACE (a thread's items
//    always hold one channel group: its constants stay in registers), one barrier per stage.
constexpr int KP2 = 32, BRB2 = 512;
// the bias gradient = column sums of du: each lane adds the eight K entries of its column with four packed dot products against
// (1, 1) -- four registers and 16 VALU instructions per 16 pixels where an all-ones MFMA per fragment took 64 registers
typedef __attribute__((ext_vector_type(2))) elt_t elt2_t;
#ifdef USTRUN_ELT_F16
#define USTRUN_DOT2_ONES(pair, acc) __builtin_amdgcn_fdot2(pair, (elt2_t){(elt_t)1.f, (elt_t)1.f}, acc, false)
#else
#define USTRUN_DOT2_ONES(pair, acc) __builtin_amdgcn_fdot2_f32_bf16(pair, (elt2_t){(elt_t)1.f, (elt_t)1.f}, acc, false)
#endif
template <int TM> struct T2 {
    static constexpr int ARB = TM * 2;                       // activation row pitch (bytes)
    static constexpr int ATILE = KP2 * ARB, BTILE = KP2 * BRB2, STAGE = ATILE + BTILE;
    static constexpr int NTH = 2 * TM;                       // threads: 512 / 256
    static constexpr int AIT = KP2 * (TM / 8) / NTH;         // 16-byte activation items per thread and stage: 2
    static constexpr int BIT = KP2 * 32 / NTH;               // du items: 2 / 4
    static constexpr int WM = TM / 64, WN = 2;               // waves along ci / along the columns (one 32-co half each)
    static constexpr int NT = 4;                             // du fragments (8 co x 4 taps each) per wave
};

// DIAG (a development build, launched only while ustrun_debug_buffer is set; tools/diag_wgradT.py): phase stamps per wave into
// dbg[block][wave 0..7][8] -- cycles (s_memtime) summed over the stages: 0 issuing the transfers two stages ahead, 1 fragment reads +
// MFMAs, 2 the counted wait for the next stage, 3 its activation in place, 4 the barrier; 5 = stages.  Measured (1024 -> 512 at 16 x
// 16, N = 64, per 32-pixel stage and wave): 646 / 1197 / 103 / 722 / 387 cycles -- the 16 MFMAs' phase is pipe-bound (two waves per
// SIMD: 1024), everything else runs with the waves in step and nothing multiplying.  Tried on that evidence, both without effect
// (0.568 ms over the four layers either way): the stage's transfers issued one by one between the MFMAs; two 128-ci blocks per CU.
__device__ __forceinline__ unsigned long long stamp_t2() {
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
template <int TM, bool DIAG = false>
__global__ __launch_bounds__(2 * TM, 2) void wgradT2_bf16_kernel(const WgradArgs a, const int ntn, unsigned long long* __restrict__ dbg = nullptr) {
    typedef T2<TM> G;
    constexpr int ARB = G::ARB, ATILE = G::ATILE, STAGE = G::STAGE, AIT = G::AIT, BIT = G::BIT, WN = G::WN, NT = G::NT, NTH = G::NTH;
    constexpr int OOB = (int)0x80000000;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 3 x {activation [32 px][TM], du [32 px][2 halves][4 taps][32 co]}
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nblk = gridDim.x, tiles = nblk / a.ksplit;          // XCD-contiguous (slice-major) order, as above
    int lin;
    {
        const int q = nblk / 8, r = nblk % 8, xcd = blockIdx.x % 8, jj = blockIdx.x / 8;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + jj;
    }
    const int ks = lin / tiles, tile = lin - ks * tiles;
    const int mtile = tile / ntn, ntile = tile % ntn;
    const int ci0 = mtile * TM, co0 = ntile * 64;
    const long kbeg = (long)ks * a.kchunk;
    const long kend = (kbeg + a.kchunk < a.M) ? kbeg + a.kchunk : a.M;
    const int W = a.Wb;

    // ---- this thread's transfer items (tile-invariant): linear 16-byte slot L = tid + NTH i of the stage's A / du tile ----
    const SrcDev& S = a.src[0];
    const bool aff = S.scale != nullptr;
    int aoff[AIT], arow[AIT];
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
        const int L = tid + NTH * i, row = L / (TM / 8), pos = L % (TM / 8);
        const int sl = pos ^ ((row & 3) << 2);                    // logical channel group at this LDS position
        arow[i] = row;
        aoff[i] = row * (int)S.sW * 2 + (ci0 + 8 * sl) * 2;
    }
    // (row & 3 is the same for all of a thread's items: one channel group per thread)
    const int acg = ((tid % (TM / 8)) ^ (((tid / (TM / 8)) & 3) << 2));
    int boff[BIT], brow[BIT];
    const int P = a.Cout * 2;                                     // bytes per hi-resolution pixel of du
#pragma unroll
    for (int i = 0; i < BIT; ++i) {
        const int L = tid + NTH * i, row = L >> 5, pos = L & 31;
        const int sl = pos ^ ((row & 3) << 2);
        const int half = sl >> 4, tap = (sl >> 2) & 3, cg = sl & 3;
        brow[i] = row;
        boff[i] = (2 * row + (tap >> 1) * 2 * W + (tap & 1)) * P + (co0 + 32 * half + 8 * cg) * 2;
    }
    const int wrapB = 2 * W * P;                                  // one more image row wrapped: + 2 W hi-resolution pixels
    f32x4 asc0 = {1.f, 1.f, 1.f, 1.f}, asc1 = asc0, ash0 = {0.f, 0.f, 0.f, 0.f}, ash1 = ash0;
    int cur_grp = -1;
    const long gpix = S.gN > 0 ? (long)S.gN * a.Hb * a.Wb : 0;
    auto load_consts = [&](long k0) {
        const int grp = gpix > 0 ? (int)(k0 / gpix) : 0;
        if (aff && grp != cur_grp) {
            const long o = (long)grp * (gpix > 0 ? S.gstride : 0) + ci0 + 8 * acg;
            asc0 = *(const f32x4*)(S.scale + o); asc1 = *(const f32x4*)(S.scale + o + 4);
            ash0 = *(const f32x4*)(S.shift + o); ash1 = *(const f32x4*)(S.shift + o + 4);
            // the loads complete HERE, inside the rare branch: left pending, hipcc puts `s_waitcnt vmcnt(0)` in front of their first
            // use -- the activation of every stage -- and that drains the transfers this kernel keeps two stages in flight
            asm volatile("" : "+v"(asc0), "+v"(asc1), "+v"(ash0), "+v"(ash1));
            cur_grp = grp;
        }
    };
    const char* const sbytes = (const char*)S.ptr;
    const char* const dbytes = (const char*)a.dy;
    const long sW2 = (long)S.sW * 2;

    // cursor of the next stage to issue (wave-uniform): first pixel and its column
    long q_m = kbeg;
    int q_x = (int)(kbeg % W);
    auto issue_stage = [&](char* stage) {
        const long m0 = q_m;
        const int x0 = q_x;
        const bool full = m0 + KP2 <= kend;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(sbytes + m0 * sW2), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dbytes + (4 * m0 - 2 * x0) * (long)P), 0, 0x7fffffff, 0x00020000);
        const int rem = (int)(kend - m0);                        // rows [0, rem) lie inside the range
#pragma unroll
        for (int i = 0; i < AIT; ++i)
            dma16_buf_s((full || arow[i] < rem) ? aoff[i] : OOB, ra, 0, stage + (NTH * i + 64 * wave) * 16);
#pragma unroll
        for (int i = 0; i < BIT; ++i) {
            const int t = x0 + brow[i];
            const int off = boff[i] + (t >= W ? wrapB : 0) + (t >= 2 * W ? wrapB : 0);
            dma16_buf_s((full || brow[i] < rem) ? off : OOB, rd, 0, stage + ATILE + (NTH * i + 64 * wave) * 16);
        }
        q_m += KP2;
        q_x += KP2;
        while (q_x >= W) q_x -= W;
    };
    // BatchNorm affine + ReLU of this thread's items of a landed stage, in place (rows past the range stay zero)
    auto activate = [&](char* stage, long m0) {
        if (!aff) return;
        const int rem = (int)((kend - m0 < KP2) ? kend - m0 : KP2);
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            u32x4* cell = (u32x4*)(stage + (tid + NTH * i) * 16);
            u32x4 u = act8_bf16(*cell, asc0, asc1, ash0, ash1, S.relu ? (short)0 : (short)0x8000);
            if (arow[i] >= rem) u = (u32x4){0u, 0u, 0u, 0u};
            *cell = u;
        }
    };

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool do_bias = a.bias_partials != nullptr && mtile == 0 && wm == 0;      // wave-uniform
    float accb[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) accb[j] = 0.f;

    // per-lane fragment bases (k0 = 0): rows 8 (lane >> 5) + q (+ 4), activation columns 64 wm + 32 i + 16 ((lane >> 4) & 1) + 4 p,
    // du: tap p's 64-byte segment of this wave's 256-byte half, 8-co group t (+ 4 g)
    const int fq = (lane & 15) >> 2, fp = lane & 3, fg = (lane >> 4) & 1, frow = 8 * (lane >> 5) + fq;
    int abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) abase[i] = frow * ARB + (((wm * 64 + 32 * i + 16 * fg + 4 * fp) * 2) ^ (fq << 6));
    const int half = wn, t0 = 0;
    const int bbase = ATILE + frow * BRB2 + 256 * half + ((fp ^ fq) << 6) + 8 * fg + 16 * t0;
    auto fragA = [&](const char* st, int i, int k0) {
        const char* p = st + abase[i] + k0 * ARB;
        const bf16x4 lo = USTRUN_DS_READ_TR16((lds_bf16x4*)p);
        const bf16x4 hi = USTRUN_DS_READ_TR16((lds_bf16x4*)(p + 4 * ARB));
        bf16x8 f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    };
    auto fragB = [&](const char* st, int t, int k0) {
        const char* p = st + bbase + 16 * t + k0 * BRB2;
        const bf16x4 lo = USTRUN_DS_READ_TR16((lds_bf16x4*)p);
        const bf16x4 hi = USTRUN_DS_READ_TR16((lds_bf16x4*)(p + 4 * BRB2));
        bf16x8 f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        return f;
    };

    auto wait_all_but_one_stage = [&]() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(AIT + BIT) : "memory"); };
    char* st[3] = {smem, smem + STAGE, smem + 2 * STAGE};
    const int nstage = (int)((kend - kbeg + KP2 - 1) / KP2);
    if (nstage > 0) {
        load_consts(kbeg);
        issue_stage(st[0]);
        if (nstage > 1) { issue_stage(st[1]); wait_all_but_one_stage(); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        activate(st[0], kbeg);
    }
    __syncthreads();
    unsigned long long dsum[5] = {0, 0, 0, 0, 0}, d0 = 0, d1 = 0;
#pragma unroll 1
    for (int s = 0; s < nstage; ++s) {
        const bool issue = s + 2 < nstage;
        if constexpr (DIAG) d0 = stamp_t2();
        if (issue) issue_stage(st[2]);
        if constexpr (DIAG) { d1 = stamp_t2(); dsum[0] += d1 - d0; d0 = d1; }
#pragma unroll
        for (int kk = 0; kk < KP2 / 16; ++kk) {
            bf16x8 af[2], bf[NT];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = fragA(st[0], i, 16 * kk);
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = fragB(st[0], j, 16 * kk);
            if (do_bias) {       // (rows past the range arrived as zeros)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) accb[j] = USTRUN_DOT2_ONES(((elt2_t){bf[j][2 * q], bf[j][2 * q + 1]}), accb[j]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = USTRUN_MFMA_32x32x16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DIAG) { d1 = stamp_t2(); dsum[1] += d1 - d0; d0 = d1; }
        if (s + 1 < nstage) {
            if (issue) wait_all_but_one_stage();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (DIAG) { d1 = stamp_t2(); dsum[2] += d1 - d0; d0 = d1; }
            const long m1 = kbeg + (long)(s + 1) * KP2;
            load_consts(m1);
            activate(st[1], m1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (DIAG) { d1 = stamp_t2(); dsum[3] += d1 - d0; d0 = d1; }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (DIAG) { d1 = stamp_t2(); dsum[4] += d1 - d0; }
        { char* tmp = st[0]; st[0] = st[1]; st[1] = st[2]; st[2] = tmp; }
    }
    if constexpr (DIAG) {
        if (lane == 0 && dbg)
            for (int k = 0; k < 8; ++k) dbg[((long)blockIdx.x * 8 + wave) * 8 + k] = k < 5 ? dsum[k] : (k == 5 ? (unsigned long long)nstage : 0ull);
    }

    // slab in the torch layout [Cin][Cout][2][2]: rows of D are ci (registers); lane 16 g + 4 tap + c is column (co, tap)
    float* slab = a.partials + (long)ks * 4 * a.Cin * a.Cout;
    const int l31 = lane & 31, lh = lane >> 5;
    const int lg = l31 >> 4, ltap = (l31 >> 2) & 3, lc = l31 & 3;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = co0 + 32 * half + 8 * (t0 + j) + 4 * lg + lc;
        float* o = slab + (long)co * 4 + ltap;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = ci0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                o[(long)ci * a.Cout * 4] = acc[i][j][r];
            }
        if (do_bias) {      // lane l holds the sum of its column over K entries 8 (l >> 5) .. + 7 of every 16: fold the halves, then the four taps
            float v = accb[j];
            v += __shfl_xor(v, 32);
            v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8);
            if (lh == 0 && ltap == 0) a.bias_partials[(long)ks * a.Cout + co] = v;
        }
    }
}

}  // namespace

static bool wgradT_base_ok(const WgradArgs& a) {
    if (a.nseg != 4 || a.segw != 2 || a.dy_s != 2 || a.astep != 0 || a.d0 != 0 || a.ashift != 0 || a.dy_esz != 2 || a.nsrc != 1) return false;
    const SrcDev& s = a.src[0];
    if (s.esz != 2 || s.sC != 1 || s.pool || s.off_y || s.off_x || s.LH != a.Hb || s.LW != a.Wb) return false;
    if ((s.relu && !s.scale) || (s.sW & 7)) return false;
    if (s.sH != (long)a.Wb * s.sW || s.sN != (long)a.Hb * s.sH) return false;      // pixel-linear source
    if (a.dyH != 2 * a.Hb || a.dyW != 2 * a.Wb || a.Wb >= 32768) return false;
    return true;
}

// the round-5 kernel: whole 64-co column tiles, 128 / 256-ci row tiles, rows of >= 16 pixels (a 32-pixel stage wraps at most two
// image rows), 32-bit item offsets; ustrun_debug_flags bit 27: never (A/B runs against the round-1 kernel)
static int wgradT2_tm(const WgradArgs& a) {
    if (!wgradT_base_ok(a) || (g_debug_flags & (1 << 27))) return 0;
    const SrcDev& s = a.src[0];
    if (a.Cout % 64 || a.Cin % 128 || a.Wb < 16) return 0;
    if (s.gN > 0 && ((long)s.gN * a.Hb * a.Wb) % KP2) return 0;         // stages must not straddle passes
    if ((long)KP2 * s.sW * 2 + (long)a.Cin * 2 >= (1L << 30)) return 0;
    if ((2L * KP2 + 6L * a.Wb + 2) * a.Cout * 2 >= (1L << 30)) return 0;
    return a.Cin % 256 == 0 ? 256 : 128;
}

bool wgradT_supported(const WgradArgs& a) {
    if (!wgradT_base_ok(a)) return false;
    if (wgradT2_tm(a)) return true;
    const SrcDev& s = a.src[0];
    if (s.gN > 0 && ((long)s.gN * a.Hb * a.Wb) % KP) return false;        // stages must not straddle passes
    return a.Cin % 128 == 0 && a.Cout % 32 == 0;
}

// split-K plan: one resident wave of blocks (2 per CU), at least four 64-pixel stages per block
int wgradT_plan(int Cin, int Cout, long M, int* ksplit, long* kchunk) {
    const long tiles = (long)(Cin / TM) * (Cout / 32);
    long ks = (512 + tiles - 1) / tiles;
    if (ks > M / (4 * KP)) ks = M / (4 * KP);
    if (ks < 1) ks = 1;
    long chunk = (M + ks - 1) / ks;
    chunk = (chunk + KP - 1) / KP * KP;
    *kchunk = chunk; *ksplit = (int)((M + chunk - 1) / chunk);
    return 0;
}
// ... of the round-5 kernel: one resident round of blocks (one 256-ci block or two 128-ci blocks per CU), at least eight 32-pixel
// stages per block.  (Cin, Cout alone decide the tile: the partials bound ustrun_wgrad_partials_bytes publishes takes the larger
// of the two plans.)
int wgradT2_plan(int Cin, int Cout, long M, int* ksplit, long* kchunk) {
    const int tm = Cin % 256 == 0 ? 256 : 128;
    const long tiles = (long)(Cin / tm) * (Cout / 64);
    long ks = ((tm == 256 ? 256 : 512) + tiles - 1) / tiles;
    if (ks > M / (8 * KP2)) ks = M / (8 * KP2);
    if (ks < 1) ks = 1;
    long chunk = (M + ks - 1) / ks;
    chunk = (chunk + KP2 - 1) / KP2 * KP2;
    *kchunk = chunk; *ksplit = (int)((M + chunk - 1) / chunk);
    return 0;
}
int wgradT_plan_for(const WgradArgs& a, int* ksplit, long* kchunk) {
    return wgradT2_tm(a) ? wgradT2_plan(a.Cin, a.Cout, a.M, ksplit, kchunk) : wgradT_plan(a.Cin, a.Cout, a.M, ksplit, kchunk);
}

int wgradT_launch_bf16(const WgradArgs& a, hipStream_t st) {
    const int tm = wgradT2_tm(a);
    if (tm) {
        dim3 grid((a.Cin / tm) * (a.Cout / 64) * a.ksplit), block(2 * tm);
        set_last_wgrad_variant(0x54320000 | tm);                   // 'T2' | ci tile
        unsigned long long* dbg = nullptr;
        USTRUN_TRY(debug_buffer_for((long)grid.x, "wgradT2", &dbg));
        if (tm == 256 && dbg) {                  // ustrun_debug_buffer set: the stamped build (tools/diag_wgradT.py)
            USTRUN_TRY(ensure_dynamic_lds((const void*)wgradT2_bf16_kernel<256, true>, 3 * T2<256>::STAGE, "wgradT2_bf16"));
            hipLaunchKernelGGL((wgradT2_bf16_kernel<256, true>), grid, block, 3 * T2<256>::STAGE, st, a, a.Cout / 64, dbg);
        } else if (tm == 256) {
            USTRUN_TRY(ensure_dynamic_lds((const void*)wgradT2_bf16_kernel<256>, 3 * T2<256>::STAGE, "wgradT2_bf16"));
            hipLaunchKernelGGL((wgradT2_bf16_kernel<256>), grid, block, 3 * T2<256>::STAGE, st, a, a.Cout / 64, (unsigned long long*)nullptr);
        } else {
            USTRUN_TRY(ensure_dynamic_lds((const void*)wgradT2_bf16_kernel<128>, 3 * T2<128>::STAGE, "wgradT2_bf16"));
            hipLaunchKernelGGL((wgradT2_bf16_kernel<128>), grid, block, 3 * T2<128>::STAGE, st, a, a.Cout / 64, (unsigned long long*)nullptr);
        }
        USTRUN_LAUNCH_CHECK("wgradT2_bf16");
        return 0;
    }
    set_last_wgrad_variant(0x54310000);                            // 'T1'
    dim3 grid((a.Cin / TM) * (a.Cout / 32) * a.ksplit), block(256);
    hipLaunchKernelGGL(wgradT_bf16_kernel, grid, block, 4 * KP * RB, st, a, a.Cout / 32);
    USTRUN_LAUNCH_CHECK("wgradT_bf16");
    return 0;
}

}  // namespace ustrun
