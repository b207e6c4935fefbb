// convT_bf16.hip -- ConvTranspose2d(k=2, s=2) forward and input-gradient on the bf16 matrix cores.
// Replaces nn.ConvTranspose2d in Up (reference networks/unet_parts.py:50-52) and its autograd backward.
//
// With stride == kernel the four taps never overlap, so both directions are plain GEMMs over the
// LOW-resolution pixel grid m = (n, y, x):
//   forward   u[n, 2y+dy, 2x+dx, co] = bias[co] + sum_ci act(a[m, ci]) w[ci, co, dy, dx]
//             -> [M x Cin] x [Cin x 4*Cout], columns ordered (tap, co), scatter epilogue
//   dgrad     da[m, ci] = sum_{tap, co} du[n, 2y+dy, 2x+dx, co] w[ci, co, dy, dx]
//             -> [M x 4*Cout] x [4*Cout x Cin], the K axis ordered (tap, co), gather loader
// A rows are 128 consecutive pixels (no halo, no tile waste at any extent), staged global -> registers ->
// [BatchNorm affine + ReLU in f32] -> bf16 -> XOR-swizzled LDS one chunk ahead of the MFMAs; weights are
// [tap][K/8][N][8] bf16 tiles fetched by LDS-DMA.  The epilogue transposes through per-wave LDS scratch and
// stores 16 bytes per lane (see conv_halo_bf16.hip).  The high-resolution pixel of (m, tap) is
// 4m - 2x + 2*dy*W + dx, so only x = m mod W is ever needed.
#include "common.h"
#include "loader.h"

namespace ustrun {
namespace {

typedef __attribute__((ext_vector_type(8))) elt_t bf16x8;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

__device__ __forceinline__ int fastmod(int v, int d, float inv) {      // v mod d for 0 <= v < 2^22, d < 2^16
    const int q = (int)(((float)v + 0.5f) * inv);
    int r = v - q * d;
    if (r < 0) r += d;
    if (r >= d) r -= d;
    return r;
}

// MODE 0: forward (a.src[0] = activation source, a.Cin = Cin, a.Cout = Cout, out0 = u, bias)
// MODE 1: dgrad   (a.src[0] = du [N,2H,2W,Cout] contiguous, a.Cin = Cout, a.Cout = Cin, out0 = da)
// MODE 2: a plain 1x1 convolution (round 2: the bottleneck GEMMs of DeepLabV2-ResNet, reference networks/backbone/resnet.py:
//         12-13,66,70): [M x Cin] x [Cin x Cout] with the forward's loader (producer's BatchNorm + ReLU on load), the
//         dgrad's plain epilogue, optional bias, and BatchNorm-statistics partials (one row per 128-pixel tile) of the
//         rounded outputs
// BREG (round 3, as in conv_halo_bf16.hip): the packed weight layout [K/8][N][8] IS the B-fragment layout, so each wave loads the
// fragments of its 64 columns straight from L2 into registers one stage (two k-steps = 4 MI MFMAs) ahead -- no weight
// tile in LDS, no LDS-DMA beside the compiler-visible activation loads (which made hipcc drain vmcnt(0) at every chunk) -- and the
// waves share only the activation tile: one barrier per 64-channel chunk.
// JOIN (round 6, MODE 2 only): the 1x1 GEMM is the input gradient of a bottleneck's conv1 (DeepLabV2-ResNet backward, reference
// networks/backbone/resnet.py:87-105 under autograd) and its epilogue performs the residual join that followed it as a pass of its
// own: stored = (product + join_add) * (join_ref > 0) -- the gradient of the previous block's pre-ReLU sum -- and, with BNS, the two
// BatchNorm-backward sums of that block's bn3 (sum(g), sum(g y3): a.bnsc = 0, a.bnsh = 1 make the mask all ones) from the stored
// pieces.  7 tensor passes over the widest maps of the network become 4.
template <int BN, int BK, int MODE, bool BREG, bool BNS = false, bool JOIN = false>
__global__ __launch_bounds__(256, 2) void convT_bf16_kernel(const IgemmArgs a, const int mt_total, const int nt_total) {
    constexpr bool DG = MODE == 1, PLAIN = MODE == 2;
    static_assert(!BNS || DG || JOIN, "BatchNorm-backward sums ride on an input gradient's epilogue");
    static_assert(!JOIN || PLAIN, "the residual join rides on the 1x1 GEMM");
    static_assert(!BREG || BK == 64, "register-fed weights: two stages of two k-steps per chunk");
    constexpr int BM = 128;
    constexpr int WN = BN / 64, WM = 4 / WN, MI = BM / (32 * WM);
    constexpr int CPR = BK / 8;                 // 16-byte groups per pixel per chunk
    constexpr int AIT = BM * CPR / 256;         // A items per thread per chunk
    constexpr int ROWB = BK * 2;
    constexpr int ABYTES = BM * ROWB;
    constexpr int BCH = CPR * BN;               // 16-byte chunks per B tile
    constexpr int BIT = BCH / 256;
    static_assert(BCH % 256 == 0 && 256 % BN == 0, "B tile mapping: a thread keeps one column");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + 2 * ABYTES;

    const int ntiles = mt_total * nt_total;
    int bid = blockIdx.x;
    {
        const int q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, j = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int mtile = bid / nt_total, ntile = bid % nt_total;
    const int n0 = ntile * BN;
    const int m0 = mtile * BM;                  // (M < 2^29: the launcher checks)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: LDS-DMA bases and role tests stay scalar
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;
    auto swz = [](int p) { return BK == 64 ? ((p >> 1) & 7) : ((p >> 2) & 3); };

    const int W = a.Wb, HWb = a.Hb * a.Wb;
    const float invW = 1.f / (float)W;
    const int Ktot = DG ? 4 * a.Cin : a.Cin;
    const int nchunk = Ktot / BK;
    const elt_t* Wp = (const elt_t*)a.W;
    const int x0r = m0 % W;                     // x of the tile's first pixel

    // ---- A items: pixel = (tid + 256 i) / CPR, 8-channel group c8 = tid % CPR.
    // Round 4: 32-bit, TILE-RELATIVE addressing through a buffer resource whose base is the tile's first (gradient: lowest) source
    // row -- the per-tile setup of this kernel was ~2300 instructions per wave in front of 64 MFMAs, most of them 64-bit index
    // arithmetic and divisions (profiles/r04_pmc_convT_up4.txt).  The source is dense NHWC (the launchers check), so a forward item
    // is px * Cin elements behind the tile base; a gradient item is (4 px - 2 x + 2 W) * Cin behind the base (4 m0 - 2 W) * Cin,
    // which keeps the offset non-negative; rows past M fall behind the tensor's end and read zeros by the range check.
    const int c8 = tid % CPR;
    constexpr int OOBV = (int)0x80000000;
    int avoff[AIT];                             // byte offsets behind the tile base
    const long a_total = DG ? (long)a.M * 4 * a.Cin : (long)a.M * a.Cin;               // source elements
    const long a_base = DG ? ((long)4 * m0 - 2 * W) * a.Cin : (long)m0 * a.Cin;        // (negative for the first gradient tile: never dereferenced)
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
        const int px = (tid + 256 * i) / CPR;
        if (DG) {
            const int x = fastmod(x0r + px, W, invW);
            avoff[i] = ((4 * px - 2 * x + 2 * W) * a.Cin + 8 * c8) * 2;
        } else {
            avoff[i] = (px * a.Cin + 8 * c8) * 2;
        }
        if (m0 + px >= a.M) avoff[i] = OOBV;    // (the gradient's rows past M end up behind the tensor anyway; the forward's last tile needs it
    }                                           //  only when another pass's rows follow in memory -- cheap either way)
    const long a_left = (a_total - a_base) * 2;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)((const elt_t*)a.src[0].ptr + a_base), 0,
                                                                          (int)(a_left < 0x7fffffffL ? a_left : 0x7fffffffL), 0x00020000);
    bf16x8 av[AIT];
    f32x4 asc0, asc1, ash0, ash1;
    const bool a_aff = !DG && a.src[0].scale != nullptr;
    // batched passes: a tile never straddles two passes (checked on the host), its pass picks the BatchNorm constants
    const long goff = (!DG && a.src[0].gN > 0) ? (long)(m0 / (a.src[0].gN * HWb)) * a.src[0].gstride : 0;
    const bool a_relu = !DG && a.src[0].relu;
    auto load_A = [&](int c) {
        int koffb;                              // wave-uniform byte offset of the chunk: rides in the scalar offset
        if (DG) {
            const int k0 = c * BK, tap = k0 / a.Cin, kk = k0 - tap * a.Cin;
            koffb = (((tap >> 1) * 2 * W + (tap & 1)) * a.Cin + kk) * 2;
        } else {
            koffb = c * BK * 2;
            if (a_aff) {
                const int koff = c * BK + 8 * c8;
                asc0 = *(const f32x4*)(a.src[0].scale + goff + koff); asc1 = *(const f32x4*)(a.src[0].scale + goff + koff + 4);
                ash0 = *(const f32x4*)(a.src[0].shift + goff + koff); ash1 = *(const f32x4*)(a.src[0].shift + goff + koff + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < AIT; ++i) av[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff[i], koffb, 0));
    };
    auto write_A = [&](char* Adst) {
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int px = (tid + 256 * i) / CPR;
            bf16x8 h = av[i];
            if (a_aff) {                                  // wave-uniform
                const bf16x8 r = av[i];
                f32x4 lo = (f32x4){(float)r[0], (float)r[1], (float)r[2], (float)r[3]} * asc0 + ash0;
                f32x4 hi = (f32x4){(float)r[4], (float)r[5], (float)r[6], (float)r[7]} * asc1 + ash1;
                if (a_relu) { lo = relu4(lo); hi = relu4(hi); }
                h[0] = (elt_t)lo[0]; h[1] = (elt_t)lo[1]; h[2] = (elt_t)lo[2]; h[3] = (elt_t)lo[3];
                h[4] = (elt_t)hi[0]; h[5] = (elt_t)hi[1]; h[6] = (elt_t)hi[2]; h[7] = (elt_t)hi[3];
            }
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
            u32x4 u = __builtin_bit_cast(u32x4, h);
            const bool ok = avoff[i] != OOBV;             // a select, not a branch (rows past M: the affine of a zero is not zero)
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = ok ? u[q] : 0u;
            *(bf16x8*)(Adst + px * ROWB + ((c8 ^ swz(px)) * 16)) = __builtin_bit_cast(bf16x8, u);
        }
    };
    // ---- B tile of chunk c: rows c*CPR .. +CPR of the [K/8][ncols][8] matrix; this thread's column is fixed ----
    long bcol;                                   // element offset of the thread's column in K-row 0
    {
        const int n = n0 + (tid % BN);
        if (DG || PLAIN) {
            bcol = (long)n * 8;
        } else {
            const int tap = n / a.Cout, co = n - tap * a.Cout;
            bcol = ((long)tap * (a.Cin / 8) * a.Cout + co) * 8;
        }
    }
    const long brow = (long)a.Cout * 8;          // elements per K/8 row
    auto dma_B = [&](int c, int buf) {
        char* dst = Bs + buf * (BCH * 16);
#pragma unroll
        for (int i = 0; i < BIT; ++i) {
            const int o = (tid + 256 * i) / BN;
            const elt_t* src = Wp + bcol + (long)(c * CPR + o) * brow;
            __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(dst + (wave * 64 + 256 * i) * 16), 16, 0, 0);
        }
    };

    // BREG: fragments (k-step ks of the stage, column half j) of this wave's 64 columns: K-row c*CPR + 2 ks + lh, column
    // colw + 32 j + l31.  Inline asm (hipcc must not fold them into its own vmcnt bookkeeping); counted by hand below.
    bf16x8 bnx[2][2];
    int bvoff = 0;
    __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, 0x7fffffff, 0x00020000);
    if constexpr (BREG) {
        const int n = n0 + wn * 64 + l31;
        long bc;
        if (DG || PLAIN) bc = (long)n * 8;
        else { const int tp = n / a.Cout, co = n - tp * a.Cout; bc = ((long)tp * (a.Cin / 8) * a.Cout + co) * 8; }
        bvoff = (int)((bc + (long)lh * brow) * 2);
    }
    auto load_Breg = [&](int c, int stg) {       // stage stg (k-steps 2 stg, 2 stg + 1) of chunk c -> bnx
        const int s0 = __builtin_amdgcn_readfirstlane((int)(((long)(c * CPR + 4 * stg) * brow) * 2));
        const int s1 = __builtin_amdgcn_readfirstlane((int)(((long)(c * CPR + 4 * stg + 2) * brow) * 2));
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %4, %5, %6 offen\n\tbuffer_load_dwordx4 %1, %4, %5, %6 offen offset:512\n\t"
                     "buffer_load_dwordx4 %2, %4, %5, %7 offen\n\tbuffer_load_dwordx4 %3, %4, %5, %7 offen offset:512"
                     : "=&v"(bnx[0][0]), "=&v"(bnx[0][1]), "=&v"(bnx[1][0]), "=&v"(bnx[1][1])
                     : "v"(bvoff), "s"(wrs), "s"(s0), "s"(s1) : "memory");
    };

    // The product is D[column][pixel] (weights are the MFMA's A operand): lane = pixel l31 of a 32-pixel sub-tile, registers 4g .. 4g + 3
    // of accumulator j = columns 32 j + 8 g + 4 lh + 0..3 -- four consecutive channels of one pixel, so the epilogue packs them with two
    // cvt_pk and parks them with ONE ds_write_b64 (round 4: the kernel spent ~1800 VALU instructions per wave and tile beside its 64
    // MFMAs -- SQ counters in profiles/r04_pmc_convT_up4.txt -- a quarter of them 2-byte LDS scatter of the other orientation).  The
    // bias of a column is a property of the REGISTER now: the accumulators start from it instead of from zero.
    f32x16 acc[MI][2];
    {
        float* bsc = (float*)smem + wave * 64;              // this wave's 64 bias values (the A tiles are not in use yet)
        const int colw0 = n0 + wn * 64;
        float bv = 0.f;
        if (!DG && a.bias) {
            int co0 = colw0;
            if (!PLAIN) { const int tp = colw0 / a.Cout; co0 = colw0 - tp * a.Cout; }
            bv = a.bias[co0 + lane];
        }
        bsc[lane] = bv;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *(const f32x4*)(bsc + j * 32 + 8 * g + 4 * lh);
#pragma unroll
                for (int i = 0; i < MI; ++i) { acc[i][j][4 * g] = b4[0]; acc[i][j][4 * g + 1] = b4[1]; acc[i][j][4 * g + 2] = b4[2]; acc[i][j][4 * g + 3] = b4[3]; }
            }
        __syncthreads();                                    // (the bias scratch aliases the first activation tile)
    }

    if constexpr (BREG) load_Breg(0, 0); else dma_B(0, 0);
    load_A(0);
    write_A(As);
    if constexpr (BREG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    const int prow0 = wm * 32 * MI + l31;        // this lane's A row in sub-tile 0
    if constexpr (BREG) {
#pragma unroll 1
        for (int c = 0; c < nchunk; ++c) {
            const bool more = c + 1 < nchunk;
            const char* Acur = As + buf * ABYTES;
#pragma unroll
            for (int stg = 0; stg < 2; ++stg) {
                bf16x8 bcur[2][2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) { bcur[ks][0] = bnx[ks][0]; bcur[ks][1] = bnx[ks][1]; }
                // top: the next stage's weights (stage 1 of this chunk, or stage 0 of the next one); behind them, in stage 0, the next
                // chunk's activation items -- so that the wait for THIS stage's weights never covers a load from HBM
                if (stg == 0) { load_Breg(c, 1); if (more) load_A(c + 1); }
                else if (more) load_Breg(c + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                // this stage's fragments were issued one stage ago; younger than them: stage 0: the 4 loads just issued + the AIT
                // activation loads; stage 1: the activation loads issued in stage 0 + the 4 loads just issued
                if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 + AIT) : "memory");
                else if (stg == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(bcur[ks][0]), "+v"(bcur[ks][1]));      // (no MFMA above the wait)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int ch = 2 * (2 * stg + ks) + lh;
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int p = prow0 + 32 * i;
                        const bf16x8 af = *(const bf16x8*)(Acur + p * ROWB + ((ch ^ swz(p)) * 16));
                        acc[i][0] = USTRUN_MFMA_32x32x16(bcur[ks][0], af, acc[i][0], 0, 0, 0);      // weights as the A operand: D[column][pixel]
                        acc[i][1] = USTRUN_MFMA_32x32x16(bcur[ks][1], af, acc[i][1], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) write_A(As + (buf ^ 1) * ABYTES);
            __syncthreads();
            buf ^= 1;
        }
    }
#pragma unroll 1
    for (int c = 0; c < (BREG ? 0 : nchunk); ++c) {
        const bool more = c + 1 < nchunk;
        if (more) { dma_B(c + 1, buf ^ 1); load_A(c + 1); }
        const char* Acur = As + buf * ABYTES;
        const char* Bp = Bs + buf * (BCH * 16) + (lh * BN + wn * 64 + l31) * 16;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int ch = 2 * ks + lh;
            const bf16x8 b0 = *(const bf16x8*)(Bp + (2 * ks * BN) * 16);
            const bf16x8 b1 = *(const bf16x8*)(Bp + (2 * ks * BN + 32) * 16);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int p = prow0 + 32 * i;
                const bf16x8 af = *(const bf16x8*)(Acur + p * ROWB + ((ch ^ swz(p)) * 16));
                acc[i][0] = USTRUN_MFMA_32x32x16(b0, af, acc[i][0], 0, 0, 0);
                acc[i][1] = USTRUN_MFMA_32x32x16(b1, af, acc[i][1], 0, 0, 0);
            }
        }
        if (more) write_A(As + (buf ^ 1) * ABYTES);
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue: bf16, per-wave LDS transpose (ds_write_b64: four channels of a pixel), 16-byte stores of 64 contiguous channels ----
    constexpr int EPITCH = 144;
    char* ep = smem + wave * (32 * EPITCH);
    const int colw = n0 + wn * 64;               // first of this wave's 64 columns
    int tap = 0, co0 = colw;
    if (!DG && !PLAIN) { tap = colw / a.Cout; co0 = colw - tap * a.Cout; }
    float st1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, st2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // PLAIN: statistics of the lane's 8 channels (lane & 7)
    // stores: 32-bit offsets behind the wave's own base through a buffer resource (soffset 0: see conv_first.hip on why no
    // register soffset rides on a 128-bit store) -- forward: base = pixel 4 m0 - 2 W + tap offset, channel co0; a piece of row d sits
    // (4 d - 2 x + 2 W) * Cout + 8 ch behind it; gradient / plain: base = pixel m0, column colw.  Rows past M fall behind the tensor.
    const long tapoff = (long)(tap >> 1) * 2 * W + (tap & 1);
    const long o_total = (DG || PLAIN) ? (long)a.M * a.Cout : (long)a.M * 4 * a.Cout;
    const long o_base = (DG || PLAIN) ? (long)m0 * a.Cout + colw : ((long)4 * m0 - 2 * W + tapoff) * a.Cout + co0;
    const long o_left = (o_total - o_base) * 2;
    const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc((void*)((elt_t*)a.out0 + o_base), 0,
                                                                          (int)(o_left < 0x7fffffffL ? o_left : 0x7fffffffL), 0x00020000);
    typedef __attribute__((ext_vector_type(4))) elt_t bf16x4;
    // this lane's pieces of a sub-tile: pixel rows r0 + 8 t (t = 0..3), channel group ch -- x advances by 8 per piece
    const int r0 = lane >> 3, ch = lane & 7;
    // BNS (round 4): da written here is the gradient at a BatchNorm + ReLU layer's output (the conv2 under this ConvTranspose); its
    // backward sums sum(da mask), sum(da mask y) are formed from the stored pieces and the 16 bytes of y beside each, as one
    // statistics row per 128-pixel tile (the tile lies inside one pass: the launcher checks) -- conv_halo_bf16.hip, same idea
    float bsc[8], bsh[8];
    __amdgpu_buffer_rsrc_t yrs = ors;
    if constexpr (BNS) {
        if (JOIN && !a.bnsc) {          // a BatchNorm that no ReLU follows (bn3): every position counts
#pragma unroll
            for (int q = 0; q < 8; ++q) { bsc[q] = 0.f; bsh[q] = 1.f; }
        } else {
            const long go = a.bn_gN > 0 ? (long)(m0 / (a.bn_gN * a.Hb * a.Wb)) * a.bn_gstride : 0;
            const float* ps = a.bnsc + go + colw + ch * 8;
            const float* pb = a.bnsh + go + colw + ch * 8;
            const f32x4 s0 = *(const f32x4*)ps, s1_ = *(const f32x4*)(ps + 4), b0 = *(const f32x4*)pb, b1 = *(const f32x4*)(pb + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) { bsc[q] = s0[q]; bsc[4 + q] = s1_[q]; bsh[q] = b0[q]; bsh[4 + q] = b1[q]; }
        }
        yrs = __builtin_amdgcn_make_buffer_rsrc((void*)((elt_t*)a.bny + o_base), 0, (int)(o_left < 0x7fffffffL ? o_left : 0x7fffffffL), 0x00020000);
    }
    // JOIN: the other contribution and the ReLU reference beside every stored piece (same layout as the output; a null tensor
    // reads through a zero-length resource: zeros, and "no reference" means no mask)
    __amdgpu_buffer_rsrc_t jars = ors, jrrs = ors;
    if constexpr (JOIN) {
        const int left = (int)(o_left < 0x7fffffffL ? o_left : 0x7fffffffL);
        jars = __builtin_amdgcn_make_buffer_rsrc((void*)(a.join_add ? (elt_t*)a.join_add + o_base : (elt_t*)a.out0), 0, a.join_add ? left : 0, 0x00020000);
        jrrs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.join_ref ? (elt_t*)a.join_ref + o_base : (elt_t*)a.out0), 0, a.join_ref ? left : 0, 0x00020000);
    }
    const bool j_mask = JOIN && a.join_ref != nullptr;
    __syncthreads();                              // every wave is done with the activation tiles the scratch aliases
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int d0 = wm * 32 * MI + 32 * i + r0;
        // The tensors read beside the stored pieces (join: the other contribution and the ReLU reference; BNS: y) do not depend on
        // the product: all of a sub-tile's loads (up to twelve 16-byte pieces per lane) are issued HERE, in front of the LDS
        // transposition, instead of one piece at a time between the stores -- hipcc kept each load behind the store before it
        // (it cannot tell the resources apart) and every piece waited out a memory round trip of its own (16 per wave and tile).
        // (Issued one sub-tile further ahead -- 24 pieces in flight, 256 registers -- measured the same.)
        u32x4_t pb[4], pr[4], py[4];
        if constexpr (JOIN || BNS) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int d = d0 + 8 * t;
                const int vo = (m0 + d < a.M) ? (d * a.Cout + ch * 8) * 2 : OOBV;
                if constexpr (JOIN) {
                    pb[t] = __builtin_amdgcn_raw_buffer_load_b128(jars, vo, 0, 0);
                    pr[t] = __builtin_amdgcn_raw_buffer_load_b128(jrrs, vo, 0, 0);
                }
                if constexpr (BNS) py[t] = __builtin_amdgcn_raw_buffer_load_b128(yrs, vo, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 h4;
#pragma unroll
                for (int q = 0; q < 4; ++q) h4[q] = (elt_t)acc[i][j][4 * g + q];
                *(bf16x4*)(ep + l31 * EPITCH + (j * 32 + 8 * g + 4 * lh) * 2) = h4;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int x = (DG || PLAIN) ? 0 : fastmod(x0r + d0, W, invW);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = r0 + 8 * t;
            bf16x8 v8 = *(const bf16x8*)(ep + row * EPITCH + ch * 16);
            const int d = d0 + 8 * t;
            const bool ok = m0 + d < a.M;
            const int vo = (DG || PLAIN) ? (d * a.Cout + ch * 8) * 2 : ((4 * d - 2 * x + 2 * W) * a.Cout + ch * 8) * 2;
            if constexpr (JOIN) {       // (the unfused pair rounds the product to the storage type, adds in f32 and rounds again: so does this)
                const bf16x8 b8 = __builtin_bit_cast(bf16x8, pb[t]);
                const bf16x8 r8 = __builtin_bit_cast(bf16x8, pr[t]);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float g = (float)v8[q] + (float)b8[q];
                    v8[q] = (elt_t)((!j_mask || (float)r8[q] > 0.f) ? g : 0.f);
                }
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v8), ors, ok ? vo : OOBV, 0, 0);
            if constexpr (BNS) {
                const bf16x8 y8 = __builtin_bit_cast(bf16x8, py[t]);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float yf = (float)y8[q];
                    const float dz = (ok && yf * bsc[q] + bsh[q] > 0.f) ? (float)v8[q] : 0.f;
                    st1[q] += dz; st2[q] += dz * yf;
                }
            }
            if constexpr (PLAIN && !BNS) {          // statistics see the stored values; rows past M are not counted
                if (a.stat) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const float f = ok ? (float)v8[q] : 0.f; st1[q] += f; st2[q] += f * f; }
                }
            }
            if (!(DG || PLAIN)) {                   // the next piece is 8 pixels on: x advances without a division from W = 8 up
                if (W >= 8) { x += 8; x = x >= W ? x - W : x; }
                else x = fastmod(x0r + d + 8, W, invW);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if constexpr (PLAIN || BNS) {
        if (a.stat) {       // fixed order: the lanes that hold the same channels (lane & 7), then the WM waves that share these columns
            __syncthreads();
            float* red = (float*)smem;            // [WM][2][BN]
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) { st1[q] += __shfl_xor(st1[q], o); st2[q] += __shfl_xor(st2[q], o); }
                if (lane < 8) {
                    red[(wm * 2 + 0) * BN + wn * 64 + lane * 8 + q] = st1[q];
                    red[(wm * 2 + 1) * BN + wn * 64 + lane * 8 + q] = st2[q];
                }
            }
            __syncthreads();
            for (int t = tid; t < 2 * BN; t += 256) {
                const int q = t / BN, cc = t % BN;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) v += red[(w * 2 + q) * BN + cc];
                a.stat[((long)mtile * 2 + q) * a.Cout + n0 + cc] = v;
            }
        }
    }
}

template <int BN, int BK, int MODE, bool BREG = false, bool BNS = false, bool JOIN = false>
int launch_cfg(const IgemmArgs& a, hipStream_t st) {
    const int ncols = MODE == 0 ? 4 * a.Cout : a.Cout;
    const int mt = cdiv(a.M, 128), nt = ncols / BN;
    const size_t lds_a = 2 * (size_t)128 * BK * 2, lds_ep = 4 * 32 * 144;
    const size_t lds = BREG ? (lds_a > lds_ep ? lds_a : lds_ep) : lds_a + 2 * (size_t)(BK / 8) * BN * 16;
    dim3 grid(mt * nt), block(256);
    set_last_variant(0x43540000 | (JOIN ? 0x4000 : 0) | (JOIN && BNS ? 0x2000 : 0) | (BREG ? 0x1000 : 0) | (BN / 32) << 8 | (BK / 32) << 4 | MODE);   // 'CT' | join | sums | register-fed weights | BN/32 | BK/32 | MODE (tests)
    hipLaunchKernelGGL((convT_bf16_kernel<BN, BK, MODE, BREG, BNS, JOIN>), grid, block, lds, st, a, mt, nt);
    USTRUN_LAUNCH_CHECK("convT_bf16");
    return 0;
}

bool common_ok(const IgemmArgs& a) {
    if (a.nsrc != 1 || a.out_esz != 2 || a.out1) return false;
    const SrcDev& s = a.src[0];
    if (s.esz != 2 || s.sC != 1 || s.pool || s.off_y || s.off_x) return false;
    if (a.M >= (1l << 29) || a.Wb >= 32768) return false;
    return true;
}
// the forward / 1x1 loaders address their source pixel-linearly: dense NHWC only (strided views take the generic kernel)
bool dense_src(const IgemmArgs& a) {
    const SrcDev& s = a.src[0];
    return s.C == a.Cin && s.sW == s.C && s.sH == (long)s.W * s.C && s.sN == (long)s.H * s.W * s.C;
}

}  // namespace

// ConvTranspose forward as built by ustrun_convT2x2_fwd (nz = 4 parity classes, one segment)
bool convT_fwd_supported(const IgemmArgs& a) {
    if (a.nz != 4 || a.nseg != 1 || a.s_out != 2 || a.s_in != 1 || a.stat) return false;
    if (!common_ok(a)) return false;
    const SrcDev& s = a.src[0];
    if (s.LH != a.Hb || s.LW != a.Wb || !dense_src(a)) return false;
    if (s.gN > 0 && ((long)s.gN * a.Hb * a.Wb) % 128) return false;     // 128-pixel tiles must not straddle passes
    return a.Cin % 64 == 0 && a.Cout % 64 == 0 && a.C0 == a.Cout;
}
// ConvTranspose input-gradient as built by ustrun_convT2x2_dgrad (4 segments reading du at stride 2)
bool convT_dgrad_supported(const IgemmArgs& a) {
    if (a.nz != 1 || a.nseg != 4 || a.segw != 2 || a.s_in != 2 || a.s_out != 1 || (a.stat && !a.bny) || a.bias) return false;
    if (!common_ok(a)) return false;
    const SrcDev& s = a.src[0];
    if (s.scale || s.relu || s.H != 2 * a.Hb || s.W != 2 * a.Wb) return false;
    if (s.sW != s.C || s.sH != (long)s.W * s.C || s.sN != (long)s.H * s.W * s.C) return false;
    return a.Cin % 64 == 0 && a.Cout % 128 == 0 && a.C0 == a.Cout;
}

// (ustrun_debug_flags bit 9 = 512: the round-2 builds with weight tiles in LDS, for A/B runs and the tests that pin them)
int convT_fwd_launch_bf16(const IgemmArgs& a, hipStream_t st) {
    if (g_debug_flags & 512) return launch_cfg<256, 32, 0>(a, st);
    return launch_cfg<256, 64, 0, true>(a, st);
}
// a 1x1 convolution at stride 1 over one bf16 NHWC source (ustrun_conv2d_fwd, k = 1)
bool conv1x1_supported(const IgemmArgs& a) {
    if (a.nz != 1 || a.nseg != 1 || a.s_out != 1 || a.s_in != 1 || a.d0 != 0) return false;
    if (!common_ok(a)) return false;
    const SrcDev& s = a.src[0];
    if (s.LH != a.Hb || s.LW != a.Wb || !dense_src(a) || s.gN > 0) return false;
    return a.Cin % 64 == 0 && a.Cout % 64 == 0 && a.C0 == a.Cout && a.Ho == a.Hb && a.Wo == a.Wb;
}
int conv1x1_launch_bf16(const IgemmArgs& a, hipStream_t st) {
    const bool old = g_debug_flags & 512;
    if (a.Cout % 256 == 0 && (long)cdiv(a.M, 128) * (a.Cout / 256) >= 256) return old ? launch_cfg<256, 32, 2>(a, st) : launch_cfg<256, 64, 2, true>(a, st);
    if (a.Cout % 128 == 0) return old ? launch_cfg<128, 64, 2>(a, st) : launch_cfg<128, 64, 2, true>(a, st);
    return old ? launch_cfg<64, 64, 2>(a, st) : launch_cfg<64, 64, 2, true>(a, st);
}
// the 1x1 GEMM with the residual join in its epilogue (and, with a.bny, the BatchNorm-backward sums of the stored gradient): a plain
// source, 128-column tiles, one statistics row per 128-pixel tile
bool conv1x1_join_supported(const IgemmArgs& a) {
    if (!conv1x1_supported(a) || (g_debug_flags & 512) || a.Cout % 128 || a.bias) return false;
    if (a.src[0].scale || a.src[0].relu) return false;
    return (long)a.M * a.Cout < (1L << 30);          // (32-bit byte offsets behind a tile base)
}
int conv1x1_join_launch_bf16(const IgemmArgs& a, hipStream_t st) {
    USTRUN_CHECK(conv1x1_join_supported(a) && (a.join_add || a.join_ref || a.bny), "conv1x1_join: unsupported shape");
    USTRUN_CHECK((a.bnsc == nullptr) == (a.bnsh == nullptr), "conv1x1_join: scale / shift come together");
    USTRUN_CHECK(!a.bny || a.stat, "conv1x1_join: BatchNorm-backward sums need their statistics rows");
    const bool wide = a.Cout % 256 == 0 && (long)cdiv(a.M, 128) * (a.Cout / 256) >= 256;
    if (a.bny) return wide ? launch_cfg<256, 64, 2, true, true, true>(a, st) : launch_cfg<128, 64, 2, true, true, true>(a, st);
    return wide ? launch_cfg<256, 64, 2, true, false, true>(a, st) : launch_cfg<128, 64, 2, true, false, true>(a, st);
}
// the input gradient that also forms the BatchNorm-backward sums of the layer whose da it writes (a.bny set): one statistics row per
// 128-pixel tile; the register-fed builds only, tiles inside one pass
bool convT_dgrad_bnsum_supported(const IgemmArgs& a) {
    if (!convT_dgrad_supported(a) || (g_debug_flags & 512)) return false;
    return a.bn_gN == 0 || ((long)a.bn_gN * a.Hb * a.Wb) % 128 == 0;
}
int convT_dgrad_launch_bf16(const IgemmArgs& a, hipStream_t st) {
    if (a.bny) {
        USTRUN_CHECK(a.stat && a.bnsc && a.bnsh && convT_dgrad_bnsum_supported(a), "convT_dgrad: BatchNorm-backward sums on an unsupported shape");
        if (a.Cout % 256 == 0 && (long)cdiv(a.M, 128) * (a.Cout / 256) >= 512) return launch_cfg<256, 64, 1, true, true>(a, st);
        return launch_cfg<128, 64, 1, true, true>(a, st);
    }
    const bool old = g_debug_flags & 512;
    if (a.Cout % 256 == 0 && (long)cdiv(a.M, 128) * (a.Cout / 256) >= 512) return old ? launch_cfg<256, 32, 1>(a, st) : launch_cfg<256, 64, 1, true>(a, st);
    return old ? launch_cfg<128, 64, 1>(a, st) : launch_cfg<128, 64, 1, true>(a, st);
}

}  // namespace ustrun
