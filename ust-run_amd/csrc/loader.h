// Device-side activation loader shared by the implicit-GEMM kernels: one 4-channel group of one
// logical pixel, with the producer's BatchNorm affine + ReLU, 2x2 max-pool, the concat of two
// sources and zero padding evaluated on the fly.
#pragma once
#include "common.h"

namespace ustrun {

// field-wise select between the two kernel-argument sources (a runtime index into a kernarg
// array would be spilled to scratch)
__device__ __forceinline__ SrcDev pick_src(const SrcDev& s0, const SrcDev& s1, bool second) {
    SrcDev d;
    d.ptr = second ? s1.ptr : s0.ptr;
    d.scale = second ? s1.scale : s0.scale;
    d.shift = second ? s1.shift : s0.shift;
    d.C = second ? s1.C : s0.C; d.H = second ? s1.H : s0.H; d.W = second ? s1.W : s0.W;
    d.sN = second ? s1.sN : s0.sN; d.sH = second ? s1.sH : s0.sH;
    d.sW = second ? s1.sW : s0.sW; d.sC = second ? s1.sC : s0.sC;
    d.relu = second ? s1.relu : s0.relu; d.pool = second ? s1.pool : s0.pool;
    d.off_y = second ? s1.off_y : s0.off_y; d.off_x = second ? s1.off_x : s0.off_x;
    d.LH = second ? s1.LH : s0.LH; d.LW = second ? s1.LW : s0.LW;
    return d;
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    return v;
}
__device__ __forceinline__ f32x4 max4(f32x4 a, f32x4 b) {
    a[0] = fmaxf(a[0], b[0]); a[1] = fmaxf(a[1], b[1]); a[2] = fmaxf(a[2], b[2]); a[3] = fmaxf(a[3], b[3]);
    return a;
}

// Scalar (any stride, any channel count) evaluation of logical element (n, iy, ix, c) of the
// concatenated sources.  (iy, ix) are coordinates on the consumer's input grid.
__device__ __forceinline__ float load_elem(const SrcDev& s0, const SrcDev& s1, int nsrc, int n, int iy, int ix, int c) {
    const bool second = (nsrc == 2 && c >= s0.C);
    const SrcDev S = pick_src(s0, s1, second);
    const int cl = c - (second ? s0.C : 0);
    const int ly = iy - S.off_y, lx = ix - S.off_x;
    if (ly < 0 || ly >= S.LH || lx < 0 || lx >= S.LW) return 0.f;
    const float sc = S.scale ? S.scale[cl] : 1.f, sh = S.scale ? S.shift[cl] : 0.f;
    if (S.pool) {
        const float* p = S.ptr + n * S.sN + (long)(2 * ly) * S.sH + (long)(2 * lx) * S.sW + cl * S.sC;
        float t0 = p[0] * sc + sh, t1 = p[S.sW] * sc + sh, t2 = p[S.sH] * sc + sh, t3 = p[S.sH + S.sW] * sc + sh;
        if (S.relu) { t0 = fmaxf(t0, 0.f); t1 = fmaxf(t1, 0.f); t2 = fmaxf(t2, 0.f); t3 = fmaxf(t3, 0.f); }
        return fmaxf(fmaxf(t0, t1), fmaxf(t2, t3));
    }
    float r = S.ptr[n * S.sN + (long)ly * S.sH + (long)lx * S.sW + cl * S.sC] * sc + sh;
    return S.relu ? fmaxf(r, 0.f) : r;
}

__device__ __forceinline__ bool sources_vectorizable(const SrcDev& s0, const SrcDev& s1, int nsrc) {
    return (s0.sC == 1 && (s0.C & 3) == 0) && (nsrc == 1 || (s1.sC == 1 && (s1.C & 3) == 0));
}

}  // namespace ustrun
