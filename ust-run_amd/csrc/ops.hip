// ops.hip -- C-ABI wrappers that phrase each U-Net operator as one launch of the generic
// implicit-GEMM / weight-gradient kernels, plus the weight packing kernels.
#include "common.h"
#include "loader.h"
#include <stdlib.h>

namespace ustrun {
namespace {

// torch conv weight [Cout][Cin][3][3] -> fwd [9][Cin][Cout], dgrad [9][Cout][Cin]
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, int Cout, int Cin, float* __restrict__ wf,
                                    float* __restrict__ wd) {
    const long total = (long)Cout * Cin * 9;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        // e indexes the fwd layout so that the stores are coalesced
        const int co = (int)(e % Cout);
        const long t = e / Cout;
        const int ci = (int)(t % Cin), tap = (int)(t / Cin);
        const float v = w[((long)co * Cin + ci) * 9 + tap];
        wf[e] = v;
        if (wd) wd[((long)tap * Cout + co) * Cin + ci] = v;
    }
}

// torch convT weight [Cin][Cout][2][2] -> fwd [4][Cin][Cout], dgrad [4][Cout][Cin]
__global__ void pack_convT_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ wf,
                                  float* __restrict__ wd) {
    const long total = (long)Cin * Cout * 4;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int co = (int)(e % Cout);
        const long t = e / Cout;
        const int ci = (int)(t % Cin), ij = (int)(t / Cin);
        const float v = w[((long)ci * Cout + co) * 4 + ij];
        wf[e] = v;
        if (wd) wd[((long)ij * Cout + co) * Cin + ci] = v;
    }
}

int pack_blocks(long total) { long b = (total + 1023) / 1024; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

}  // namespace
}  // namespace ustrun

using namespace ustrun;

extern "C" int ustrun_pack_conv3x3(const float* w, int Cout, int Cin, void* w_fwd, void* w_dgrad, int dtype,
                                   ustrun_stream_t s) {
    USTRUN_CHECK(dtype_ok(dtype), "pack_conv3x3: dtype %d not built", dtype);
    USTRUN_CHECK(w && w_fwd && Cout > 0 && Cin > 0, "pack_conv3x3: bad args");
    if (dtype == USTRUN_D16) return pack_bf16(w, Cout, Cin, 9, 0, w_fwd, w_dgrad, (hipStream_t)s);
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(pack_blocks((long)Cout * Cin * 9)), dim3(256), 0, (hipStream_t)s, w, Cout,
                       Cin, (float*)w_fwd, (float*)w_dgrad);
    USTRUN_LAUNCH_CHECK("pack_conv3x3");
    if (dtype == USTRUN_F32X3) {          // the three bf16 planes behind each f32 pack (x3.hip)
        USTRUN_TRY(pack_x3((const float*)w_fwd, 9, Cin, Cout, (hipStream_t)s));
        if (w_dgrad) USTRUN_TRY(pack_x3((const float*)w_dgrad, 9, Cout, Cin, (hipStream_t)s));
    }
    return 0;
}

extern "C" int ustrun_pack_convT2x2(const float* w, int Cin, int Cout, void* w_fwd, void* w_dgrad, int dtype,
                                    ustrun_stream_t s) {
    USTRUN_CHECK(dtype_ok(dtype), "pack_convT2x2: dtype %d not built", dtype);
    USTRUN_CHECK(w && w_fwd && Cout > 0 && Cin > 0, "pack_convT2x2: bad args");
    if (dtype == USTRUN_D16) return pack_bf16(w, Cout, Cin, 4, 1, w_fwd, w_dgrad, (hipStream_t)s);
    hipLaunchKernelGGL(pack_convT_kernel, dim3(pack_blocks((long)Cout * Cin * 4)), dim3(256), 0, (hipStream_t)s, w, Cin,
                       Cout, (float*)w_fwd, (float*)w_dgrad);
    USTRUN_LAUNCH_CHECK("pack_convT2x2");
    if (dtype == USTRUN_F32X3) {
        USTRUN_TRY(pack_x3((const float*)w_fwd, 4, Cin, Cout, (hipStream_t)s));
        if (w_dgrad) USTRUN_TRY(pack_x3((const float*)w_dgrad, 4, Cout, Cin, (hipStream_t)s));
    }
    return 0;
}

// upper bound of the BatchNorm-statistics rows over every kernel that may serve the launch; the
// wrapper zero-fills the buffer when the chosen kernel writes fewer (extents not multiples of 16).
// Three families write rows: the linear kernels (one per 128 pixels), the halo-tiled kernels (at most one per 8 x 16
// pixels <= 2 per 16 x 16 tile) and the streaming 64 -> 64 kernel (two per strip segment: 2 N cdiv(W, 32) sy with sy up to
// cdiv(H, 8) segments per strip -- round 2 left this family out of the bound and a 72 x 72, N = 8 layer wrote 432 rows into
// 400).  Every entry point that takes `stat` checks its kernel's row count against this bound before it launches.
extern "C" int ustrun_conv_mtiles(int N, int H, int W, int Cout) {
    const int lin = igemm_mtiles((int64_t)N * H * W, Cout), tiled = N * cdiv(H, 16) * 2 * cdiv(W, 16);
    const int stream = Cout == 64 ? 2 * N * cdiv(W, 32) * cdiv(H, 8) : 0;
    const int m = lin > tiled ? lin : tiled;
    return m > stream ? m : stream;
}

// the published bound is a contract: a kernel that would write more rows than callers were told to allocate must not launch
static int check_stat_rows(int used, int N, int H, int W, int Cout, const char* who) {
    const int cap = ustrun_conv_mtiles(N, H, W, Cout);
    USTRUN_CHECK(used <= cap, "%s: the kernel for N=%d %dx%d Cout=%d writes %d statistics rows, ustrun_conv_mtiles promises %d", who, N, H,
                 W, Cout, used, cap);
    return 0;
}
namespace ustrun { int stat_rows_within_bound(int used, int N, int H, int W, int Cout, const char* who) { return check_stat_rows(used, N, H, W, Cout, who); } }

// Host-only: the number of statistics rows the kernel serving a k x k convolution of a dense NHWC source would write (no
// launch, no device access) -- tests sweep shapes with it against ustrun_conv_mtiles without a GPU.
extern "C" int ustrun_debug_conv_stat_rows(int N, int Ho, int Wo, int Cin, int Cout, int k, int stride, int dilation, int pooled,
                                           int dtype) {
    IgemmArgs a = {};
    ustrun_src_t sd = {};
    const int Hs = (pooled ? 2 : 1) * ((Ho - 1) * stride + 1), Ws = (pooled ? 2 : 1) * ((Wo - 1) * stride + 1);
    sd.ptr = (const void*)16; sd.C = Cin; sd.H = Hs; sd.W = Ws;
    sd.sC = 1; sd.sW = Cin; sd.sH = (int64_t)Ws * Cin; sd.sN = (int64_t)Hs * Ws * Cin;
    sd.pool = pooled; sd.relu = pooled;
    a.nsrc = 1; a.src[0] = make_src(sd, dtype); a.Cin = Cin;
    a.W = (const float*)16; a.Cout = Cout;
    a.N = N; a.Hb = Ho; a.Wb = Wo; a.M = N * Ho * Wo;
    a.s_in = stride; a.nseg = k * k; a.segw = k; a.d0 = -dilation * (k / 2); a.dstep = dilation;
    a.nz = 1; a.s_out = 1;
    a.out0 = (float*)16; a.C0 = Cout; a.Ho = Ho; a.Wo = Wo; a.out_esz = act_esz(dtype);
    a.stat = (float*)16;
    if (k == 3 && stride == 1 && dilation == 1 && Cin <= 4 && Cout == 64 && !pooled) return conv_first_stat_rows(N, Ho, Wo, dtype);
    return igemm_stat_rows_used(a, dtype);
}

static int check_srcs(const ustrun_src_t* srcs, int nsrc, const char* who) {
    USTRUN_CHECK(srcs && (nsrc == 1 || nsrc == 2), "%s: nsrc must be 1 or 2", who);
    for (int i = 0; i < nsrc; ++i) {
        USTRUN_CHECK(srcs[i].ptr && srcs[i].C > 0 && srcs[i].H > 0 && srcs[i].W > 0, "%s: bad source %d", who, i);
        USTRUN_CHECK((srcs[i].scale == nullptr) == (srcs[i].shift == nullptr), "%s: scale/shift must come together", who);
        USTRUN_CHECK(!srcs[i].pool || srcs[i].relu, "%s: pooled sources must be ReLU-activated", who);
    }
    return 0;
}

// ---- several forward passes batched into one call (ustrun_src_t::gN): the fast bf16 kernels pick the BatchNorm
// constants per image; every other kernel is run once per pass on the corresponding slices -------------------------
namespace ustrun { thread_local bool g_short_last_pass = false; }
static int src_groups(const ustrun_src_t* srcs, int nsrc, int N, int* gN) {
    int g = 0;
    for (int i = 0; i < nsrc; ++i)
        if (srcs[i].gN > 0) {       // (a source without scale/shift still tells where the passes -- the statistics groups -- end)
            if (g && g != srcs[i].gN) return -1;
            g = srcs[i].gN;
        }
    *gN = g;
    if (g == 0) return 1;
    // a shorter last pass exists only where the caller DECLARED one (ustrun_unet_desc_t::tail -> ShortLastPass in unet.hip, which
    // sized a (G+1)-th constants table for it); a mis-sized batch through the operator API is an error, not a tail
    if (N % g != 0 && !g_short_last_pass) return -1;
    return (N + g - 1) / g;
}
static inline int pass_images(int g, int gN, int N) { return N - g * gN < gN ? N - g * gN : gN; }
static ustrun_src_t src_slice(const ustrun_src_t& s, int g, int gN, int dtype) {
    ustrun_src_t t = s;
    const int esz = (s.f32 || dtype != USTRUN_D16) ? 4 : 2;
    t.ptr = (const char*)s.ptr + (int64_t)g * gN * s.sN * esz;
    if (s.scale && s.gN > 0) { t.scale = s.scale + (int64_t)g * s.gstride; t.shift = s.shift + (int64_t)g * s.gstride; }
    t.gN = 0; t.gstride = 0;
    return t;
}

// rows the LAST pass of the calling thread's latest ustrun_conv3x3_fwd_rows wrote when it ran one launch per pass (-1: one launch over
// the whole batch, every image the same number of rows): how the U-Net plan splits the rows of a batch with a shorter tail pass
namespace ustrun { thread_local int g_last_pass_rows = -1; int conv_last_pass_rows() { return g_last_pass_rows; } }
extern "C" int ustrun_debug_last_conv_variant(void) { return halo_last_variant(); }
extern "C" int ustrun_debug_last_wgrad_variant(void) { return wgrad_last_variant(); }
// (USTRUN_DEBUG_FLAGS in the environment presets the flags: A/B runs of whole programs on one box)
namespace ustrun {
static const int g_env_debug_flags = getenv("USTRUN_DEBUG_FLAGS") ? atoi(getenv("USTRUN_DEBUG_FLAGS")) : 0;     // read once, at load
thread_local int g_debug_flags = g_env_debug_flags;
static const int g_env_debug_flags2 = getenv("USTRUN_DEBUG_FLAGS2") ? atoi(getenv("USTRUN_DEBUG_FLAGS2")) : 0;
thread_local int g_debug_flags2 = g_env_debug_flags2;       // the second word (ustrun_debug_flags2)
int env_debug_flags() { return g_env_debug_flags; }     // the process-wide constant: for switches that shape a plan two calls share
thread_local DebugBuf g_dbg = {nullptr, 0};
int debug_buffer_for(long blocks, const char* who, unsigned long long** out) {
    *out = nullptr;
    if (!g_dbg.p) return 0;
    USTRUN_CHECK(blocks * 64 <= g_dbg.n_u64, "%s: the stamped build writes 64 u64 per workgroup: %ld workgroups need %ld u64, "
                 "ustrun_debug_buffer was given %ld", who, blocks, blocks * 64, g_dbg.n_u64);
    *out = g_dbg.p;
    return 0;
}
}
extern "C" int ustrun_debug_buffer(void* device_u64, int64_t n_u64) {
    USTRUN_CHECK(!device_u64 || n_u64 >= 64, "debug_buffer: %ld u64 is smaller than one workgroup's stamps", (long)n_u64);
    g_dbg.p = (unsigned long long*)device_u64; g_dbg.n_u64 = device_u64 ? (long)n_u64 : 0;
    return 0;
}
extern "C" int ustrun_debug_flags(int flags) { const int old = g_debug_flags; g_debug_flags = flags; return old; }
extern "C" int ustrun_debug_flags2(int flags) { const int old = g_debug_flags2; g_debug_flags2 = flags; return old; }
extern "C" int ustrun_short_last_pass(int allow) { const int old = g_short_last_pass; g_short_last_pass = allow != 0; return old; }

// ---- clock probe (tools/clock_probe.py; MI355X_MICROARCH.md "DVFS give-back" item 6): every workgroup records the shader-clock
// counter (s_memtime: one tick per shader cycle) and the constant 100 MHz counter (s_memrealtime) together with where it ran.
// Two probes on one stream around a long series of back-to-back launches of a kernel give the clock the chip HELD while it ran
// them: d(memtime) / d(memrealtime) x 100 MHz per XCD -- without a stamped build of that kernel.
namespace ustrun { namespace {
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* __restrict__ out) {
    unsigned long long t, r;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(r) :: "memory");
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)" : "=s"(xcc), "=s"(hw));
    if (threadIdx.x == 0) {
        unsigned long long* o = out + (long)blockIdx.x * 4;
        o[0] = t; o[1] = r; o[2] = xcc & 0xf; o[3] = hw;
    }
}
} }
extern "C" int ustrun_debug_clock_probe(void* device_u64, int blocks, ustrun_stream_t s) {
    USTRUN_CHECK(device_u64 && blocks > 0 && blocks <= 65536, "debug_clock_probe: bad args");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)s, (unsigned long long*)device_u64);
    USTRUN_LAUNCH_CHECK("debug_clock_probe");
    return 0;
}

extern "C" int ustrun_conv3x3_fwd(const ustrun_src_t* srcs, int nsrc, const void* w_fwd, int N, int H, int W, int Cout,
                                  void* y, float* stat, int dtype, ustrun_stream_t s) {
    return ustrun_conv3x3_fwd_rows(srcs, nsrc, w_fwd, N, H, W, Cout, y, stat, nullptr, dtype, s);
}

extern "C" int ustrun_conv3x3_fwd_rows(const ustrun_src_t* srcs, int nsrc, const void* w_fwd, int N, int H, int W, int Cout,
                                       void* y, float* stat, int* stat_rows, int dtype, ustrun_stream_t s) {
    USTRUN_TRY(check_srcs(srcs, nsrc, "conv3x3_fwd"));
    USTRUN_CHECK(w_fwd && y && N > 0 && H > 0 && W > 0 && Cout > 0, "conv3x3_fwd: bad args");
    IgemmArgs a = {};
    a.nsrc = nsrc; a.Cin = 0;
    for (int i = 0; i < nsrc; ++i) { a.src[i] = make_src(srcs[i], dtype); a.Cin += srcs[i].C; }
    a.W = (const float*)w_fwd; a.Cout = Cout;
    a.N = N; a.Hb = H; a.Wb = W; a.M = N * H * W;
    a.s_in = 1; a.nseg = 9; a.segw = 3; a.d0 = -1; a.dstep = 1;
    a.nz = 1; a.s_out = 1;
    a.out0 = (float*)y; a.out1 = nullptr; a.C0 = Cout; a.Ho = H; a.Wo = W;
    a.bias = nullptr; a.stat = stat; a.out_esz = act_esz(dtype);
    const bool first = nsrc == 1 && conv_first_supported(srcs[0], Cout) && srcs[0].H == H && srcs[0].W == W &&
                       (srcs[0].f32 || dtype != USTRUN_D16);
    int gN = 0;
    const int G = src_groups(srcs, nsrc, N, &gN);
    USTRUN_CHECK(G >= 1, "conv3x3_fwd: inconsistent pass groups");
    a.pass_gN = gN;
    // one launch per pass -- unless the kernel picks the pass's constants per image (the 16-bit halo kernel, the halo-tiled x3 kernel)
    if (G > 1 && !(dtype == USTRUN_D16 && halo_supported(a)) && !(dtype == USTRUN_F32X3 && !first && conv3x3_x3_supported(a) && !(g_debug_flags2 & 2))) {      // (ustrun_debug_flags2 bit 1: x3 per pass, A/B runs)
        USTRUN_CHECK(!stat || stat_rows, "conv3x3_fwd: batched passes need ustrun_conv3x3_fwd_rows");
        int total = 0;
        for (int g = 0; g < G; ++g) {
            ustrun_src_t sl[2];
            for (int i = 0; i < nsrc; ++i) sl[i] = src_slice(srcs[i], g, gN, dtype);
            int rows = 0;
            USTRUN_TRY(ustrun_conv3x3_fwd_rows(sl, nsrc, w_fwd, pass_images(g, gN, N), H, W, Cout,
                                               (char*)y + (int64_t)g * gN * H * W * Cout * act_esz(dtype),
                                               stat ? stat + (int64_t)total * 2 * Cout : nullptr, &rows, dtype, s));
            total += rows;
            g_last_pass_rows = rows;
        }
        if (stat_rows) *stat_rows = total;
        return 0;
    }
    // one launch: rows per image are uniform -- or, on the halo kernel's linear tiles, per pass with the last pass's count here
    g_last_pass_rows = (dtype == USTRUN_D16 && !first && G > 1) ? halo_linear_last_pass_rows(a) : -1;
    if (stat) {
        const int rows = ustrun_conv_mtiles(N, H, W, Cout);
        const int used = first ? conv_first_stat_rows(N, H, W, dtype) : igemm_stat_rows_used(a, dtype);
        USTRUN_TRY(check_stat_rows(used, N, H, W, Cout, "conv3x3_fwd"));
        if (stat_rows) *stat_rows = used;                 // the caller finalizes exactly these rows
        else if (used < rows) {
            hipError_t e = hipMemsetAsync(stat, 0, (size_t)rows * 2 * Cout * sizeof(float), (hipStream_t)s);
            USTRUN_CHECK(e == hipSuccess, "conv3x3_fwd: memset failed: %s", hipGetErrorString(e));
        }
    }
    if (first) {     // C <= 4 input channels: direct f32 stencil, HBM-bound on its output
        // algorithmic bytes: the f32 image in, the activation out at ITS element size (bf16 storage: 2 bytes -- this used to count
        // 4, which overstated the layer's GB/s by 1.9x), the weights
        prof_begin(0, 2.0 * a.M * Cout * 9 * a.Cin,
                   (double)a.M * a.Cin * (srcs[0].f32 ? 4.0 : act_esz(dtype)) + (double)a.M * Cout * act_esz(dtype) + 36.0 * a.Cin * Cout, (hipStream_t)s);
        const int rc = conv_first_fwd(srcs[0], w_fwd, dtype, N, y, stat, (hipStream_t)s);
        prof_end((hipStream_t)s);
        return rc;
    }
    return igemm_launch(a, dtype, (hipStream_t)s);
}

extern "C" int ustrun_conv3x3_dgrad(const void* dy, const void* w_dgrad, int N, int H, int W, int Cout, int Cin,
                                    void* da0, int C0, void* da1, int H1, int W1, int o1y, int o1x, int dtype,
                                    ustrun_stream_t s) {
    USTRUN_CHECK(dy && w_dgrad && da0 && N > 0 && H > 0 && W > 0 && Cout > 0 && Cin > 0, "conv3x3_dgrad: bad args");
    USTRUN_CHECK(C0 > 0 && C0 <= Cin && (C0 == Cin || da1), "conv3x3_dgrad: bad channel split %d/%d", C0, Cin);
    IgemmArgs a = {};
    ustrun_src_t sd = {};
    sd.ptr = dy; sd.C = Cout; sd.H = H; sd.W = W;
    sd.sC = 1; sd.sW = Cout; sd.sH = (int64_t)W * Cout; sd.sN = (int64_t)H * W * Cout;
    a.nsrc = 1; a.src[0] = make_src(sd, dtype); a.Cin = Cout;
    a.W = (const float*)w_dgrad; a.Cout = Cin;
    a.N = N; a.Hb = H; a.Wb = W; a.M = N * H * W;
    a.s_in = 1; a.nseg = 9; a.segw = 3; a.d0 = 1; a.dstep = -1;   // da[q] = sum_t dy[q - (t-1)] W[t]^T
    a.nz = 1; a.s_out = 1;
    a.out0 = (float*)da0; a.C0 = C0; a.Ho = H; a.Wo = W;
    a.out1 = (float*)da1; a.H1 = H1; a.W1 = W1; a.o1y = o1y; a.o1x = o1x; a.out_esz = act_esz(dtype);
    return igemm_launch(a, dtype, (hipStream_t)s);
}

/* the same input gradient, single destination, which ALSO forms the BatchNorm-backward sums of the layer whose da it writes:
 * y = that layer's pre-BatchNorm output [N,H,W,Cin] (the layout of da), aff = its scale / shift (ustrun_src_t conventions for
 * batched passes: gN images per pass, gstride floats between the passes' constants); stat receives *stat_rows rows of
 * [2][Cin] = {sum(da mask), sum(da mask y)} per row, mask = y scale + shift > 0, over the stored da values.  *stat_rows = 0 and
 * NO launch when this shape is not one the fused epilogue covers (the caller then runs the plain input gradient and
 * ustrun_bn_bwd_reduce). */
extern "C" int ustrun_conv3x3_dgrad_bnsum(const void* dy, const void* w_dgrad, int N, int H, int W, int Cout, int Cin, void* da,
                                          const void* y, const float* scale, const float* shift, int gN, int64_t gstride,
                                          float* stat, int* stat_rows, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(dy && w_dgrad && da && y && scale && shift && stat && stat_rows && N > 0 && H > 0 && W > 0 && Cout > 0 && Cin > 0,
                 "conv3x3_dgrad_bnsum: bad args");
    *stat_rows = 0;
    if (dtype != USTRUN_D16) return 0;
    IgemmArgs a = {};
    ustrun_src_t sd = {};
    sd.ptr = dy; sd.C = Cout; sd.H = H; sd.W = W;
    sd.sC = 1; sd.sW = Cout; sd.sH = (int64_t)W * Cout; sd.sN = (int64_t)H * W * Cout;
    a.nsrc = 1; a.src[0] = make_src(sd, dtype); a.Cin = Cout;
    a.W = (const float*)w_dgrad; a.Cout = Cin;
    a.N = N; a.Hb = H; a.Wb = W; a.M = N * H * W;
    a.s_in = 1; a.nseg = 9; a.segw = 3; a.d0 = 1; a.dstep = -1;
    a.nz = 1; a.s_out = 1;
    a.out0 = (float*)da; a.C0 = Cin; a.Ho = H; a.Wo = W; a.out_esz = 2;
    a.bn_gN = gN; a.bn_gstride = (long)gstride;
    const bool ws = ws64_supported(a) && !(g_debug_flags & 1);        // (the 64 -> 64 full-resolution layers run the streaming kernel)
    if (ws ? !ws64_bnsum_supported(a) : !halo_bnsum_supported(a)) return 0;
    a.bny = y; a.bnsc = scale; a.bnsh = shift; a.stat = stat;
    const int used = ws ? ws64_stat_rows(a) : halo_stat_rows_used(a);
    USTRUN_TRY(check_stat_rows(used, N, H, W, Cin, "conv3x3_dgrad_bnsum"));
    USTRUN_TRY(igemm_launch(a, dtype, (hipStream_t)s));          // (profiled with the conv class; dispatches to the halo kernel)
    *stat_rows = used;
    return 0;
}

extern "C" int ustrun_convT2x2_fwd(const ustrun_src_t* src, const void* w_fwd, const float* bias, int N, int H, int W,
                                   int Cout, void* u, int dtype, ustrun_stream_t s) {
    USTRUN_TRY(check_srcs(src, 1, "convT2x2_fwd"));
    USTRUN_CHECK(w_fwd && u && N > 0 && H > 0 && W > 0 && Cout > 0, "convT2x2_fwd: bad args");
    IgemmArgs a = {};
    a.nsrc = 1; a.src[0] = make_src(*src, dtype); a.Cin = src->C;
    USTRUN_CHECK(a.src[0].LH == H && a.src[0].LW == W, "convT2x2_fwd: source extent %dx%d != %dx%d", a.src[0].LH, a.src[0].LW, H, W);
    a.W = (const float*)w_fwd; a.Cout = Cout;
    a.N = N; a.Hb = H; a.Wb = W; a.M = N * H * W;
    a.s_in = 1; a.nseg = 1; a.segw = 1; a.d0 = 0; a.dstep = 0;
    a.nz = 4; a.s_out = 2;                                          // u[2p + (i,j)] = a[p] W[ij] + bias
    a.out0 = (float*)u; a.C0 = Cout; a.Ho = 2 * H; a.Wo = 2 * W;
    a.bias = bias; a.out_esz = act_esz(dtype);
    {
        int gN = 0;
        const int G = src_groups(src, 1, N, &gN);
        USTRUN_CHECK(G >= 1, "convT2x2_fwd: inconsistent pass groups");
        if (G > 1 && !(dtype == USTRUN_D16 && convT_fwd_supported(a))) {
            for (int g = 0; g < G; ++g) {
                const ustrun_src_t sl = src_slice(*src, g, gN, dtype);
                USTRUN_TRY(ustrun_convT2x2_fwd(&sl, w_fwd, bias, pass_images(g, gN, N), H, W, Cout,
                                               (char*)u + (int64_t)g * gN * 4 * H * W * Cout * act_esz(dtype), dtype, s));
            }
            return 0;
        }
    }
    return igemm_launch(a, dtype, (hipStream_t)s);
}

extern "C" int ustrun_convT2x2_dgrad(const void* du, const void* w_dgrad, int N, int H, int W, int Cout, int Cin,
                                     void* da, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(du && w_dgrad && da && N > 0 && H > 0 && W > 0 && Cout > 0 && Cin > 0, "convT2x2_dgrad: bad args");
    IgemmArgs a = {};
    ustrun_src_t sd = {};
    sd.ptr = du; sd.C = Cout; sd.H = 2 * H; sd.W = 2 * W;
    sd.sC = 1; sd.sW = Cout; sd.sH = (int64_t)2 * W * Cout; sd.sN = (int64_t)4 * H * W * Cout;
    a.nsrc = 1; a.src[0] = make_src(sd, dtype); a.Cin = Cout;
    a.W = (const float*)w_dgrad; a.Cout = Cin;
    a.N = N; a.Hb = H; a.Wb = W; a.M = N * H * W;
    a.s_in = 2; a.nseg = 4; a.segw = 2; a.d0 = 0; a.dstep = 1;     // da[p] = sum_ij du[2p+(i,j)] W[ij]^T
    a.nz = 1; a.s_out = 1;
    a.out0 = (float*)da; a.C0 = Cin; a.Ho = H; a.Wo = W; a.out_esz = act_esz(dtype);
    return igemm_launch(a, dtype, (hipStream_t)s);
}

/* ... and with the BatchNorm-backward sums of the layer whose da it writes (ustrun_conv3x3_dgrad_bnsum's contract; one row per 128
 * pixels) */
extern "C" int ustrun_convT2x2_dgrad_bnsum(const void* du, const void* w_dgrad, int N, int H, int W, int Cout, int Cin, void* da,
                                           const void* y, const float* scale, const float* shift, int gN, int64_t gstride,
                                           float* stat, int* stat_rows, int dtype, ustrun_stream_t s) {
    USTRUN_CHECK(du && w_dgrad && da && y && scale && shift && stat && stat_rows && N > 0 && H > 0 && W > 0 && Cout > 0 && Cin > 0,
                 "convT2x2_dgrad_bnsum: bad args");
    *stat_rows = 0;
    if (dtype != USTRUN_D16) return 0;
    IgemmArgs a = {};
    ustrun_src_t sd = {};
    sd.ptr = du; sd.C = Cout; sd.H = 2 * H; sd.W = 2 * W;
    sd.sC = 1; sd.sW = Cout; sd.sH = (int64_t)2 * W * Cout; sd.sN = (int64_t)4 * H * W * Cout;
    a.nsrc = 1; a.src[0] = make_src(sd, dtype); a.Cin = Cout;
    a.W = (const float*)w_dgrad; a.Cout = Cin;
    a.N = N; a.Hb = H; a.Wb = W; a.M = N * H * W;
    a.s_in = 2; a.nseg = 4; a.segw = 2; a.d0 = 0; a.dstep = 1;
    a.nz = 1; a.s_out = 1;
    a.out0 = (float*)da; a.C0 = Cin; a.Ho = H; a.Wo = W; a.out_esz = 2;
    a.bny = y; a.bnsc = scale; a.bnsh = shift; a.bn_gN = gN; a.bn_gstride = (long)gstride; a.stat = stat;
    if (!convT_dgrad_bnsum_supported(a)) return 0;
    const int used = cdiv(a.M, 128);
    USTRUN_TRY(check_stat_rows(used, N, H, W, Cin, "convT2x2_dgrad_bnsum"));
    USTRUN_TRY(igemm_launch(a, dtype, (hipStream_t)s));
    *stat_rows = used;
    return 0;
}

extern "C" int64_t ustrun_wgrad_partials_bytes(int nseg, int Cin, int Cout, int64_t npix) {
    int ks, slabs; long chunk;
    wgrad_plan(nseg, Cin, Cout, npix, &ks, &chunk, &slabs);
    if (nseg == 9 && Cin % 64 == 0 && Cout % 64 == 0) {          // the all-taps bf16 kernel may split further
        const long pairs = (long)(Cin / 64) * (Cout / 64);
        const int halo = (int)((1024 + pairs - 1) / pairs) + 1;
        if (halo > slabs) slabs = halo;
    }
    if (Cin % 64 == 0 && Cout % 64 == 0 && nseg <= 9) {          // the one-tap-per-block bf16 kernel (1x1 / dilated / strided)
        WgradArgs t = {};
        t.Cin = Cin; t.Cout = Cout; t.nseg = nseg; t.M = npix;
        int kt; long ct;
        wgrad_tap_plan(t, &kt, &ct);
        if (kt > slabs) slabs = kt;
    }
    if (nseg == 4 && Cin % 64 == 0 && Cout % 64 == 0) {          // dtype USTRUN_F32X3's ConvTranspose kernel: at most one block per CU and pair
        const long pairs = (long)(Cin / 64) * (Cout / 64);
        const int kx = (int)((256 + pairs - 1) / pairs);
        if (kx > slabs) slabs = kx;
    }
    if (nseg == 4 && Cin % 128 == 0 && Cout % 32 == 0) {         // the ConvTranspose all-taps bf16 kernels
        int kt; long ct;
        wgradT_plan(Cin, Cout, npix, &kt, &ct);
        if (kt > slabs) slabs = kt;
        if (Cout % 64 == 0) {
            wgradT2_plan(Cin, Cout, npix, &kt, &ct);
            if (kt > slabs) slabs = kt;
        }
    }
    int64_t b = (int64_t)slabs * nseg * Cin * Cout * sizeof(float);
    if (nseg == 4) b += (int64_t)slabs * Cout * sizeof(float);    // bias column sums ride behind the slabs
    if (nseg == 9 && Cin <= 4 && Cout == 64 && conv_first_wgrad_partials_bytes() > b) b = conv_first_wgrad_partials_bytes();
    return b;
}

extern "C" int ustrun_conv3x3_wgrad(const ustrun_src_t* srcs, int nsrc, const void* dy, int N, int H, int W, int Cout,
                                    float* dw, int accumulate, float* partials, int64_t partials_bytes, int dtype,
                                    ustrun_stream_t s) {
    USTRUN_TRY(check_srcs(srcs, nsrc, "conv3x3_wgrad"));
    USTRUN_CHECK(dy && dw && partials && N > 0 && H > 0 && W > 0 && Cout > 0, "conv3x3_wgrad: bad args");
    WgradArgs a = {};
    a.nsrc = nsrc; a.Cin = 0;
    for (int i = 0; i < nsrc; ++i) { a.src[i] = make_src(srcs[i], dtype); a.Cin += srcs[i].C; }
    a.dy = (const float*)dy; a.Cout = Cout; a.dy_esz = act_esz(dtype);
    a.N = N; a.Hb = H; a.Wb = W; a.M = (long)N * H * W;
    a.nseg = 9; a.segw = 3; a.d0 = -1; a.astep = 1; a.dy_s = 1; a.dyH = H; a.dyW = W;
    int slabs;
    a.partials = partials;
    {
        int gN = 0;
        const int G = src_groups(srcs, nsrc, N, &gN);
        USTRUN_CHECK(G >= 1, "conv3x3_wgrad: inconsistent pass groups");
        if (G > 1 && !(dtype == USTRUN_D16 && wgrad_halo_supported(a)) && !(dtype == USTRUN_F32X3 && wgrad_x3_supported(a) && !(g_debug_flags2 & 2))) {
            for (int g = 0; g < G; ++g) {
                ustrun_src_t sl[2];
                for (int i = 0; i < nsrc; ++i) sl[i] = src_slice(srcs[i], g, gN, dtype);
                USTRUN_TRY(ustrun_conv3x3_wgrad(sl, nsrc, (const char*)dy + (int64_t)g * gN * H * W * Cout * act_esz(dtype),
                                                pass_images(g, gN, N), H, W, Cout, dw, g == 0 ? accumulate : 1, partials, partials_bytes, dtype, s));
            }
            return 0;
        }
    }
    // (its im2col tile holds 32 rows = 9 taps of at most three channels: a 4-channel input -- 36 rows -- takes the generic kernel)
    if (dtype == USTRUN_D16 && nsrc == 1 && conv_first_supported(srcs[0], Cout) && srcs[0].H == H && srcs[0].W == W && srcs[0].f32 &&
        9 * srcs[0].C <= 32)
        return conv_first_wgrad(srcs[0], dy, act_esz(dtype), N, dw, accumulate, partials, partials_bytes, (hipStream_t)s);
    if (dtype == USTRUN_D16 && wgrad_halo_supported(a)) {
        int per;
        wgrad_halo_plan(a, &slabs, &per);
        USTRUN_CHECK(partials_bytes >= (int64_t)slabs * 9 * a.Cin * Cout * 4, "conv3x3_wgrad: partials too small");
        prof_begin(1, 2.0 * a.M * 9 * a.Cin * Cout, 2.0 * ((double)a.M * a.Cin + (double)a.M * Cout) + 36.0 * a.Cin * Cout, (hipStream_t)s);
        const int rc = wgrad_halo_launch_bf16(a, slabs, per, (hipStream_t)s);
        prof_end((hipStream_t)s);
        USTRUN_TRY(rc);
        return reduce_partials(partials, slabs, 9, a.Cin, Cout, dw, 2, accumulate, (hipStream_t)s);   // slabs are in torch layout
    }
    // (dtype USTRUN_F32X3's first convolution: the streaming kernel with three-term products)
    if (dtype == USTRUN_F32X3 && nsrc == 1 && conv_first_supported(srcs[0], Cout) && srcs[0].H == H && srcs[0].W == W && srcs[0].f32 &&
        9 * srcs[0].C <= 32 && srcs[0].sW == 1 && srcs[0].sN == (int64_t)srcs[0].C * srcs[0].sC && srcs[0].sC == (int64_t)H * srcs[0].sH &&
        (int64_t)N * H * W * 256 < (1LL << 31) - 64 && (int64_t)N * srcs[0].sN * 4 < (1LL << 31) - 64 && (int64_t)N * cdiv(W, 16) <= 4096 &&
        !(g_debug_flags & (1 << 29)))
        return conv_first_wgrad(srcs[0], dy, 4, N, dw, accumulate, partials, partials_bytes, (hipStream_t)s, true);
    if (dtype == USTRUN_F32X3 && wgrad_x3_supported(a)) {        // all nine taps per block, operands split at staging (x3.hip)
        int per;
        wgrad_x3_plan(a, &slabs, &per);
        USTRUN_CHECK(partials_bytes >= (int64_t)slabs * 9 * a.Cin * Cout * 4, "conv3x3_wgrad: partials too small");
        prof_begin(1, 2.0 * a.M * 9 * a.Cin * Cout, 4.0 * ((double)a.M * a.Cin + (double)a.M * Cout) + 36.0 * a.Cin * Cout, (hipStream_t)s);
        const int rc = wgrad_x3_launch(a, slabs, per, (hipStream_t)s);
        prof_end((hipStream_t)s);
        USTRUN_TRY(rc);
        return reduce_partials(partials, slabs, 9, a.Cin, Cout, dw, 2, accumulate, (hipStream_t)s);   // slabs are in torch layout
    }
    wgrad_plan(9, a.Cin, Cout, a.M, &a.ksplit, &a.kchunk, &slabs);
    USTRUN_CHECK(partials_bytes >= (int64_t)slabs * 9 * a.Cin * Cout * 4, "conv3x3_wgrad: partials too small");
    USTRUN_TRY(wgrad_launch(a, dtype, (hipStream_t)s));
    return reduce_partials(partials, slabs, 9, a.Cin, Cout, dw, 0, accumulate, (hipStream_t)s);
}

namespace ustrun {
namespace {
// db[co] = sum over all pixels of du[p][co]: block partials then reduce_rows.  thread = (4-channel group,
// pixel lane); fixed-order LDS combine.  C % 4 == 0.
template <int ESZ>
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ du, long npix, int C,
                                                       float* __restrict__ partials) {
    __shared__ f32x4 red[256];
    const int C4 = C / 4;
    int G = 1; while (G < C4 && G < 256) G <<= 1;
    const int PL = 256 / G, g = threadIdx.x % G, pl = threadIdx.x / G;
    for (int cb = 0; cb < C4; cb += G) {
        const int cq = cb + g;
        const bool active = cq < C4;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (active)
            for (long p = (long)blockIdx.x * PL + pl; p < npix; p += (long)gridDim.x * PL) s += ld4t<ESZ>(du, p * C + cq * 4);
        red[threadIdx.x] = s;
        __syncthreads();
        if (pl == 0 && active) {
            for (int k = 1; k < PL; ++k) s += red[k * G + g];
            *(f32x4*)(partials + (long)blockIdx.x * C + cq * 4) = s;
        }
        __syncthreads();
    }
}
}  // namespace
}  // namespace ustrun

extern "C" int ustrun_convT2x2_wgrad(const ustrun_src_t* src, const void* du, int N, int H, int W, int Cout, float* dw,
                                     float* db, int accumulate, float* partials, int64_t partials_bytes, int dtype,
                                     ustrun_stream_t s) {
    USTRUN_TRY(check_srcs(src, 1, "convT2x2_wgrad"));
    USTRUN_CHECK(du && dw && partials && N > 0 && H > 0 && W > 0 && Cout > 0, "convT2x2_wgrad: bad args");
    WgradArgs a = {};
    a.nsrc = 1; a.src[0] = make_src(*src, dtype); a.Cin = src->C;
    a.dy = (const float*)du; a.Cout = Cout; a.dy_esz = act_esz(dtype);
    a.N = N; a.Hb = H; a.Wb = W; a.M = (long)N * H * W;
    a.nseg = 4; a.segw = 2; a.d0 = 0; a.astep = 0; a.dy_s = 2; a.dyH = 2 * H; a.dyW = 2 * W;
    int slabs;
    a.partials = partials;
    {
        int gN = 0;
        const int G = src_groups(src, 1, N, &gN);
        USTRUN_CHECK(G >= 1, "convT2x2_wgrad: inconsistent pass groups");
        if (G > 1 && !(dtype == USTRUN_D16 && wgradT_supported(a))) {
            for (int g = 0; g < G; ++g) {
                const ustrun_src_t sl = src_slice(*src, g, gN, dtype);
                USTRUN_TRY(ustrun_convT2x2_wgrad(&sl, (const char*)du + (int64_t)g * gN * 4 * H * W * Cout * act_esz(dtype),
                                                 pass_images(g, gN, N), H, W, Cout, dw, db, g == 0 ? accumulate : 1, partials, partials_bytes, dtype, s));
            }
            return 0;
        }
    }
    if (dtype == USTRUN_D16 && wgradT_supported(a)) {       // all four taps (and the bias) in one GEMM: wgradT_bf16.hip
        wgradT_plan_for(a, &a.ksplit, &a.kchunk);
        slabs = a.ksplit;
        const int64_t slab_bytes = (int64_t)slabs * 4 * a.Cin * Cout * 4;
        USTRUN_CHECK(partials_bytes >= slab_bytes + (db ? (int64_t)slabs * Cout * 4 : 0), "convT2x2_wgrad: partials too small");
        a.bias_partials = db ? partials + slab_bytes / 4 : nullptr;
        prof_begin(1, 2.0 * a.M * 4 * a.Cin * Cout, 2.0 * (a.M * (double)a.Cin + 4.0 * a.M * Cout) + 16.0 * a.Cin * Cout, (hipStream_t)s);
        const int rc = wgradT_launch_bf16(a, (hipStream_t)s);
        prof_end((hipStream_t)s);
        USTRUN_TRY(rc);
        if (db) USTRUN_TRY(reduce_rows(a.bias_partials, slabs, Cout, 0, Cout, db, accumulate, (hipStream_t)s));
        return reduce_partials(partials, slabs, 4, a.Cin, Cout, dw, 2, accumulate, (hipStream_t)s);
    }
    if (dtype == USTRUN_F32X3 && wgradT_x3_supported(a)) {       // the four parity classes as four accumulators, three-term products (x3.hip)
        int per;
        wgradT_x3_plan(a, &slabs, &per);
        USTRUN_CHECK(partials_bytes >= (int64_t)slabs * 4 * a.Cin * Cout * 4, "convT2x2_wgrad: partials too small");
        prof_begin(1, 2.0 * a.M * 4 * a.Cin * Cout, 4.0 * (a.M * (double)a.Cin + 4.0 * a.M * Cout) + 16.0 * a.Cin * Cout, (hipStream_t)s);
        const int rc = wgradT_x3_launch(a, slabs, per, (hipStream_t)s);
        prof_end((hipStream_t)s);
        USTRUN_TRY(rc);
        USTRUN_TRY(reduce_partials(partials, slabs, 4, a.Cin, Cout, dw, 2, accumulate, (hipStream_t)s));      // slabs are in torch layout
    } else {
    wgrad_plan(4, a.Cin, Cout, a.M, &a.ksplit, &a.kchunk, &slabs);
    USTRUN_CHECK(partials_bytes >= (int64_t)slabs * 4 * a.Cin * Cout * 4, "convT2x2_wgrad: partials too small");
    USTRUN_TRY(wgrad_launch(a, dtype, (hipStream_t)s));
    USTRUN_TRY(reduce_partials(partials, slabs, 4, a.Cin, Cout, dw, 1, accumulate, (hipStream_t)s));
    }
    if (db) {
        const long npix = (long)N * 4 * H * W;
        int blocks = cdiv(npix, 512);
        if (blocks > 512) blocks = 512;
        USTRUN_CHECK(partials_bytes >= (int64_t)blocks * Cout * 4, "convT2x2_wgrad: partials too small for bias");
        USTRUN_CHECK(Cout % 4 == 0, "convT2x2_wgrad: Cout %d must be a multiple of 4", Cout);
        if (dtype == USTRUN_D16)
            hipLaunchKernelGGL(bias_grad_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)du, npix, Cout, partials);
        else
            hipLaunchKernelGGL(bias_grad_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const float*)du, npix, Cout, partials);
        USTRUN_LAUNCH_CHECK("bias_grad");
        USTRUN_TRY(reduce_rows(partials, blocks, Cout, 0, Cout, db, accumulate, (hipStream_t)s));
    }
    return 0;
}
