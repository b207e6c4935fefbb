// Shared host/device helpers for libustrun (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ustrun.h"

namespace ustrun {

void set_error(const char* fmt, ...);

#define USTRUN_CHECK(cond, ...)                      \
    do {                                             \
        if (!(cond)) {                               \
            ::ustrun::set_error(__VA_ARGS__);        \
            return 1;                                \
        }                                            \
    } while (0)

#define USTRUN_LAUNCH_CHECK(name)                                                   \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            ::ustrun::set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return 2;                                                               \
        }                                                                           \
    } while (0)

#define USTRUN_TRY(expr)          \
    do {                          \
        int rc_ = (expr);         \
        if (rc_) return rc_;      \
    } while (0)

// errors.hip: raise a kernel's dynamic-LDS limit once per (kernel, device); nonzero + message on failure
int ensure_dynamic_lds(const void* fn, int bytes, const char* who);

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// ---- element-wise arithmetic that runs BESIDE MFMAs (activation on load, statistics) ------------------------------------
// Scalar v_fma_f32 / v_add_f32 on purpose: next to a stream of MFMAs (the same wave's or the partner wave's on the SIMD) one
// v_pk_fma_f32 costs about 22 cycles more than two v_fma_f32 (MI355X_MICROARCH.md, "price of one filler beside MFMAs"), and
// hipcc packs every adjacent pair of f32 operations it sees.  The asm statements are opaque to that.
__device__ __forceinline__ float fma_scalar(float x, float s, float h) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(s), "v"(h));
    return r;
}
__device__ __forceinline__ float add_scalar(float x, float y) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// One LDS-DMA piece (64 lanes x 16 B) addressed through a buffer resource: lane address = base + soff + voff, an offset with
// bit 31 set (beyond num_records) writes zeros.  Inline asm: hipcc does not see it -- the kernels count vmcnt by hand.
__device__ __forceinline__ void dma16_buf_s(int voff, __amdgpu_buffer_rsrc_t rs, int soff, const char* lds_wave_base) {
    // (a generic pointer into LDS is aperture base (high half) | LDS byte offset (low half): no address-space cast, whose null
    // test hipcc mis-selects on gfx950 when the pointer comes out of a select)
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    const int so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" :: "v"(voff), "s"(rs), "s"(m), "s"(so) : "memory", "m0");
}
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
// 8 raw bf16 values (one 16-byte group) -> BatchNorm affine in f32 -> bf16 -> ReLU on the rounded pairs: signed 16-bit max
// against floor16 (0: ReLU -- the same values as max in f32 before rounding, rounding is monotonic and keeps the sign;
// 0x8000 = the most negative int16: no ReLU)
__device__ __forceinline__ u32x4_t act8_bf16(u32x4_t raw, f32x4 sc0, f32x4 sc1, f32x4 sh0, f32x4 sh1, short floor16) {
    const bf16x8_t b = __builtin_bit_cast(bf16x8_t, raw);
    bf16x8_t h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        h[q] = (__bf16)fma_scalar((float)b[q], sc0[q], sh0[q]);
        h[4 + q] = (__bf16)fma_scalar((float)b[4 + q], sc1[q], sh1[q]);
    }
    const s16x8_t z = {floor16, floor16, floor16, floor16, floor16, floor16, floor16, floor16};
    return __builtin_bit_cast(u32x4_t, __builtin_elementwise_max(__builtin_bit_cast(s16x8_t, h), z));
}
#endif

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---- device-side view of an activation source (ustrun_src_t + derived logical extent) ----
struct SrcDev {
    const float* ptr;
    const float* scale;
    const float* shift;
    int C, H, W;
    long sN, sH, sW, sC;
    int relu, pool, off_y, off_x;
    int LH, LW;  // logical extent (H/2 when pooled)
    int esz;     // element size of the stored tensor: 4 (f32) or 2 (bf16); strides are in elements
    int gN;      // > 0: image n takes scale/shift + (n / gN) * gstride (several passes batched, BatchNorm per pass)
    long gstride;
};

static inline int act_esz(int dtype) { return dtype == USTRUN_BF16 ? 2 : 4; }

static inline SrcDev make_src(const ustrun_src_t& s, int dtype) {
    SrcDev d;
    d.esz = (s.f32 || dtype != USTRUN_BF16) ? 4 : 2;
    d.ptr = (const float*)s.ptr; d.scale = s.scale; d.shift = s.shift;
    d.C = s.C; d.H = s.H; d.W = s.W;
    d.sN = s.sN; d.sH = s.sH; d.sW = s.sW; d.sC = s.sC;
    d.relu = s.relu; d.pool = s.pool; d.off_y = s.off_y; d.off_x = s.off_x;
    d.LH = s.pool ? s.H / 2 : s.H;
    d.LW = s.pool ? s.W / 2 : s.W;
    d.gN = s.scale ? s.gN : 0; d.gstride = s.gstride;
    return d;
}

// ---- generic implicit GEMM:  out[map_out(m)][n] = sum_seg sum_c A_seg[map_in(m,seg)][c] * W[seg][c][n]
struct IgemmArgs {
    SrcDev src[2];
    int nsrc, Cin;
    const float* W;          // [nslice][Cin][Cout]
    int Cout;
    int N, Hb, Wb, M;        // base grid; M = N*Hb*Wb
    int s_in;                // input pixel = base*s_in + (dy,dx)
    int nseg, segw, d0, dstep; // segment s: (dy,dx) = d0 + (s/segw, s%segw)*dstep, weight slice s
    int nz;                  // grid.z: output parity classes z: out offset (z/2, z%2), weight slice z
    int s_out;               // out pixel = base*s_out + (z/2, z%2)
    float* out0; float* out1;
    int C0;                  // channels [0,C0) -> out0, [C0,Cout) -> out1
    int Ho, Wo;              // out0 extent
    int H1, W1, o1y, o1x;    // out1 extent / offset
    const float* bias;
    float* stat;             // [mtiles][2][Cout] or null
    int out_esz;             // element size of out0/out1 (4 or 2)
};
int igemm_mtiles(int64_t M, int Cout);
int stat_rows_within_bound(int used, int N, int H, int W, int Cout, const char* who);   // ops.hip: used <= ustrun_conv_mtiles, or error
int igemm_stat_rows_used(const IgemmArgs& a, int dtype);
int igemm_launch(const IgemmArgs& a, int dtype, hipStream_t st);
int igemm_launch_bf16(const IgemmArgs& a, hipStream_t st);
bool halo_supported(const IgemmArgs& a);
bool halo_tall_tile(const IgemmArgs& a);
int halo_stat_rows_used(const IgemmArgs& a);
int conv3x3_halo_launch_bf16(const IgemmArgs& a, hipStream_t st);
int halo_stat_rows(int N, int H, int W);
int halo_last_variant();
void set_last_variant(int v);
// weight-stationary row-streaming kernel for 64 -> 64 channels at full resolution (conv_ws64_bf16.hip)
bool ws64_supported(const IgemmArgs& a);
int ws64_stat_rows(const IgemmArgs& a);
int conv3x3_ws64_launch_bf16(const IgemmArgs& a, hipStream_t st);
// Test / tuning state is PER CALLING THREAD (the header promises a library without process-wide mutable state): a thread that
// sets flags or a stamp buffer changes kernel selection for its own launches only; every thread starts from the value the
// environment variable USTRUN_DEBUG_FLAGS had when the library was loaded (read once, never on a launch path).
extern thread_local int g_debug_flags;     // ustrun_debug_flags: bit 0 = keep the 64 -> 64 layers on the tiled kernel ...
struct DebugBuf { unsigned long long* p; long n_u64; };
extern thread_local DebugBuf g_dbg;        // ustrun_debug_buffer: development builds with phase stamps write [block][8 waves][8] u64 here
// the stamp buffer for a launch of `blocks` workgroups: null when none is set; error (rc != 0) when the one set is too small
int debug_buffer_for(long blocks, const char* who, unsigned long long** out);
bool convT_fwd_supported(const IgemmArgs& a);
bool convT_dgrad_supported(const IgemmArgs& a);
bool conv1x1_supported(const IgemmArgs& a);
int conv1x1_launch_bf16(const IgemmArgs& a, hipStream_t st);
int convT_fwd_launch_bf16(const IgemmArgs& a, hipStream_t st);
int convT_dgrad_launch_bf16(const IgemmArgs& a, hipStream_t st);
int pack_bf16(const float* w, int Cout, int Cin, int taps, int transposed_src, void* wf, void* wd, hipStream_t st);
int bn_finalize_passes(float* stat, int mtiles, int passes, int C, int64_t count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                       int update_running, float* scale, float* shift, float* mean, float* rstd, long astride, hipStream_t s);
int head_bwd_passes(const float* dlogits, const void* y, const float* scale, const float* shift, int64_t npix, int HW, int C,
                    int K, const float* w, void* da, float* dw, float* db, int accumulate, float* partials,
                    int64_t partials_bytes, int dtype, int passes, long pass_aff, hipStream_t s);
int bn_eval_affine_layers(int nlayers, const int* C, const float* const* gamma, const float* const* beta,
                          const float* const* rm, const float* const* rv, float* const* aff, float eps, int passes,
                          hipStream_t s);
int bn_bwd_reduce_passes(const void* da, const void* dp, const void* y, const float* scale, const float* shift,
                         const float* mean, const float* rstd, const float* gamma, int N, int H, int W, int C, float* dgamma,
                         float* dbeta, int accumulate, float* coef, float* partials, int64_t partials_bytes, int dtype,
                         int passes, long act_elems, long pool_elems, long aff_stride, hipStream_t s);
int bn_bwd_apply_passes(const void* da, const void* dp, const void* y, const float* scale, const float* shift,
                        const float* coef, int N, int H, int W, int C, void* dy, int dtype, int passes, long act_elems,
                        long pool_elems, long aff_stride, hipStream_t s);
struct PackJob { const float* w; void* wf; void* wd; int Cout, Cin, taps, transposed_src; };
struct PackJobs { PackJob j[24]; };
int pack_bf16_multi(const PackJobs& jobs, int n, hipStream_t st);
static inline bool dtype_ok(int dtype) { return dtype == USTRUN_F32 || dtype == USTRUN_BF16; }

// ---- generic "TN" weight-gradient GEMM: dW[seg][ci][co] = sum_p A_seg[p][ci] * dY_seg[p][co]
struct WgradArgs {
    SrcDev src[2];
    int nsrc, Cin;
    const float* dy; int Cout; int dy_esz;
    int N, Hb, Wb; long M;   // base grid = pixels summed over
    int nseg, segw, d0, astep; // A side: in pixel = base + d0 + (s/segw, s%segw)*astep
    int dy_s;                  // dY pixel = base*dy_s + (dy_s == 2 ? (s/2, s%2) : (0,0))
    int ashift;                // strided convolution: in pixel = (base << ashift) + d0 + ... (generic kernels only)
    int dyH, dyW;              // dY extent
    float* partials;           // [ksplit][nseg][Cin][Cout]
    float* bias_partials;      // wgradT only: [ksplit][Cout] column sums of dY, or null
    int ksplit; long kchunk;   // block-level K splits; pixels per split (multiple of 32)
};
bool wgradT_supported(const WgradArgs& a);
int wgradT_plan(int Cin, int Cout, long M, int* ksplit, long* kchunk);
int wgradT_launch_bf16(const WgradArgs& a, hipStream_t st);
// slabs = total partial slabs written (ksplit x in-block K waves)
int wgrad_plan(int nseg, int Cin, int Cout, int64_t M, int* ksplit, long* kchunk, int* slabs);
int wgrad_launch(const WgradArgs& a, int dtype, hipStream_t st);
int wgrad_launch_bf16(const WgradArgs& a, hipStream_t st);
bool wgrad_tap_supported(const WgradArgs& a);      // one tap per block: 1x1 / dilated / strided convolutions (wgrad_tap_bf16.hip)
int wgrad_tap_plan(const WgradArgs& a, int* ksplit, long* kchunk);
int wgrad_tap_launch_bf16(const WgradArgs& a, hipStream_t st);
int wgrad_last_variant();
void set_last_wgrad_variant(int v);
bool wgrad_halo_supported(const WgradArgs& a);
int wgrad_halo_plan(const WgradArgs& a, int* ksplit, int* tiles_per);
int wgrad_halo_launch_bf16(const WgradArgs& a, int ksplit, int tiles_per, hipStream_t st);
// reduce partial slabs [ksplit][rows] -> out (permuted): layout 0: conv3x3 torch [Cout][Cin][3][3];
// layout 1: convT torch [Cin][Cout][2][2]; layout 2: plain [rows]
int reduce_partials(const float* partials, int ksplit, int nseg, int Cin, int Cout, float* out,
                    int layout, int accumulate, hipStream_t st);

// ---- optional per-launch timing with HIP events on the launch stream (bench.py roofline leg) ----
// kind 0: implicit-GEMM (conv3x3 fwd/dgrad, convT fwd/dgrad), kind 1: weight-gradient GEMM
void prof_begin(int kind, double flops, double bytes, hipStream_t st);
void prof_end(hipStream_t st);
// layer tag of the launches that follow (unet.hip): conv i forward = i, ConvTranspose j forward = 20 + j, input
// gradients + 100, weight gradients + 200; n = images in the launch; -1 = untagged (operator-level calls)
void prof_set_tag(int tag, int n);

// first convolution (C <= 4 input channels, 64 outputs): conv_first.hip
bool conv_first_supported(const ustrun_src_t& s, int Cout);
int conv_first_stat_rows(int N, int H, int W, int dtype);
int conv_first_fwd(const ustrun_src_t& s, const void* w_fwd, int dtype, int N, void* y, float* stat, hipStream_t st);
int64_t conv_first_wgrad_partials_bytes();
int conv_first_wgrad(const ustrun_src_t& s, const void* dy, int dy_esz, int N, float* dw, int accumulate, float* partials,
                     int64_t partials_bytes, hipStream_t st);

int reduce_rows(const float* part, int nslab, long stride, long offset, int count, float* out, int accumulate,
                hipStream_t st);

}  // namespace ustrun
