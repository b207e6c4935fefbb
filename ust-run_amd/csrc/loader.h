// Device-side activation loader shared by the implicit-GEMM kernels: one 4-channel group of one
// logical pixel, with the producer's BatchNorm affine + ReLU, 2x2 max-pool, the concat of two
// sources and zero padding evaluated on the fly.
#pragma once
#include "common.h"

namespace ustrun {

typedef __attribute__((ext_vector_type(4))) elt_t bf16x4_t;

// element-type-agnostic accessors: `base` is the tensor base, idx an ELEMENT index, esz 4 (f32) or 2 (bf16)
__device__ __forceinline__ f32x4 ld4(const float* base, long idx, int esz) {
    if (esz == 4) return *(const f32x4*)(base + idx);
    const bf16x4_t h = *(const bf16x4_t*)((const elt_t*)base + idx);
    return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
__device__ __forceinline__ float ld1(const float* base, long idx, int esz) {
    return esz == 4 ? base[idx] : (float)((const elt_t*)base)[idx];
}
__device__ __forceinline__ void st4(float* base, long idx, f32x4 v, int esz) {
    if (esz == 4) { *(f32x4*)(base + idx) = v; return; }
    bf16x4_t h;
    h[0] = (elt_t)v[0]; h[1] = (elt_t)v[1]; h[2] = (elt_t)v[2]; h[3] = (elt_t)v[3];
    *(bf16x4_t*)((elt_t*)base + idx) = h;
}
__device__ __forceinline__ void st1(float* base, long idx, float v, int esz) {
    if (esz == 4) base[idx] = v; else ((elt_t*)base)[idx] = (elt_t)v;
}
// value as it will read back from a tensor of element size esz
__device__ __forceinline__ float rnd(float v, int esz) { return esz == 4 ? v : (float)(elt_t)v; }

// compile-time element size: no branch around the memory instruction (a runtime select makes hipcc put
// every load in its own basic block with its own wait -- cdna_hip_programming.md "Three .s-level traps" (c))
template <int ESZ> __device__ __forceinline__ f32x4 ld4t(const float* base, long idx) {
    if constexpr (ESZ == 4) return *(const f32x4*)(base + idx);
    else {
        const bf16x4_t h = *(const bf16x4_t*)((const elt_t*)base + idx);
        return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    }
}
template <int ESZ> __device__ __forceinline__ void st4t(float* base, long idx, f32x4 v) {
    if constexpr (ESZ == 4) *(f32x4*)(base + idx) = v;
    else {
        bf16x4_t h;
        h[0] = (elt_t)v[0]; h[1] = (elt_t)v[1]; h[2] = (elt_t)v[2]; h[3] = (elt_t)v[3];
        *(bf16x4_t*)((elt_t*)base + idx) = h;
    }
}
template <int ESZ> __device__ __forceinline__ void st1t(float* base, long idx, float v) {
    if constexpr (ESZ == 4) base[idx] = v; else ((elt_t*)base)[idx] = (elt_t)v;
}
template <int ESZ> __device__ __forceinline__ float rndt(float v) {
    if constexpr (ESZ == 4) return v; else return (float)(elt_t)v;
}

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
    d.esz = second ? s1.esz : s0.esz;
    d.gN = second ? s1.gN : s0.gN; d.gstride = second ? s1.gstride : s0.gstride;
    return d;
}

// max(v, 0) as ONE v_med3_f32 (fmaxf costs a canonicalising v_max plus the max itself)
__device__ __forceinline__ float relu1(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()); }
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v[0] = relu1(v[0]); v[1] = relu1(v[1]); v[2] = relu1(v[2]); v[3] = relu1(v[3]);
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
        const long p = n * S.sN + (long)(2 * ly) * S.sH + (long)(2 * lx) * S.sW + cl * S.sC;
        float t0 = ld1(S.ptr, p, S.esz) * sc + sh, t1 = ld1(S.ptr, p + S.sW, S.esz) * sc + sh,
              t2 = ld1(S.ptr, p + S.sH, S.esz) * sc + sh, t3 = ld1(S.ptr, p + S.sH + S.sW, S.esz) * sc + sh;
        if (S.relu) { t0 = fmaxf(t0, 0.f); t1 = fmaxf(t1, 0.f); t2 = fmaxf(t2, 0.f); t3 = fmaxf(t3, 0.f); }
        return fmaxf(fmaxf(t0, t1), fmaxf(t2, t3));
    }
    float r = ld1(S.ptr, n * S.sN + (long)ly * S.sH + (long)lx * S.sW + cl * S.sC, S.esz) * sc + sh;
    return S.relu ? fmaxf(r, 0.f) : r;
}

__device__ __forceinline__ bool sources_vectorizable(const SrcDev& s0, const SrcDev& s1, int nsrc) {
    return (s0.sC == 1 && (s0.C & 3) == 0) && (nsrc == 1 || (s1.sC == 1 && (s1.C & 3) == 0));
}
// ... and every source stored with element size esz (the vector paths are compiled for one element size)
__device__ __forceinline__ bool sources_vectorizable(const SrcDev& s0, const SrcDev& s1, int nsrc, int esz) {
    return sources_vectorizable(s0, s1, nsrc) && s0.esz == esz && (nsrc == 1 || s1.esz == esz);
}

}  // namespace ustrun
