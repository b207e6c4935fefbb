"""One measured experiment on changing the arithmetic: Winograd F(2x2, 3x3) on ONE MFMA-bound layer,
512 -> 512 channels at 32 x 32, N = 64 images, bf16.

F(2x2, 3x3) turns the 3x3 convolution into 16 independent [tiles x Cin] x [Cin x Cout] products over 4 x 4 input tiles at
stride 2 (2.25x fewer multiply-adds).  Composed form measured here, from parts this library already has:

    input transform   V_xi = (B^T d B)_xi          per 4x4 tile and channel       (torch on the device: bf16 in, f32 math, bf16 out)
    16 GEMMs          M_xi = V_xi U_xi             the library's bf16 1x1 GEMM kernel (ustrun_conv2d_fwd, k = 1), one launch each
    output transform  Y    = A^T M A               per tile and output channel    (torch on the device)

Timed: the 16 GEMM launches alone (the part a fused kernel could at best approach -- its transforms add VALU work and LDS
traffic, they remove no MFMA), the transforms' HBM bytes priced at 5.4 TB/s, against the direct halo-tiled kernel on the same
layer.  Accuracy: rel-L2 of both against the f64 convolution of the SAME bf16-rounded inputs and weights.

    python tools/exp_winograd.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd"))
from ustrun import _lib as l  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    lib = l.lib()
    dev, bf = "cuda", torch.bfloat16
    n, ci, co, hw = 64, 512, 512, 32
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, ci, hw, hw, generator=g).to(bf).float()          # an activation-like input, bf16-representable
    w = (torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)).to(bf).float()
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)

    # ---- direct: the library's halo-tiled kernel (plain source, no statistics) ----
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev).to(bf)
    wf, wd = torch.zeros(9 * ci * co, dtype=bf, device=dev), torch.zeros(9 * ci * co, dtype=bf, device=dev)
    wg = w.to(dev)
    l.check(lib.ustrun_pack_conv3x3(wg.data_ptr(), co, ci, wf.data_ptr(), wd.data_ptr(), 1, None))
    src = l.nhwc_src(xg.data_ptr(), ci, hw, hw)
    y = torch.empty(n, hw, hw, co, device=dev, dtype=bf)
    direct = lambda: l.check(lib.ustrun_conv3x3_fwd(C.byref(src), 1, wf.data_ptr(), n, hw, hw, co, y.data_ptr(), None, 1, None))
    t_direct = timed(direct)
    e_direct = float((y.float().cpu().permute(0, 3, 1, 2).double() - ref).norm() / ref.norm())

    # ---- Winograd, composed ----
    T = hw // 2                                                     # tiles per side
    xp = F.pad(x, (1, 1, 1, 1)).to(dev)
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                      # [n, ci, T, T, 4, 4]
    BTd = BT.to(dev)
    V = torch.einsum("ij,nctujk,lk->nctuil", BTd, tiles, BTd)       # B^T d B
    V = V.permute(4, 5, 0, 2, 3, 1).reshape(16, n * T * T, ci).contiguous().to(bf)     # [xi][tile][ci], rounded to bf16
    U = torch.einsum("ij,ocjk,lk->ocil", G.to(dev), wg, G.to(dev))  # G g G^T: [co, ci, 4, 4]
    U = U.permute(2, 3, 0, 1).reshape(16, co, ci).contiguous()      # [xi][co][ci] f32 (packed to bf16 below)
    M = torch.empty(16, n * T * T, co, device=dev, dtype=bf)
    packs, srcs = [], []
    rows = n * T * T
    for xi in range(16):
        p = torch.zeros(lib.ustrun_pack_conv_elems(co, ci, 1), dtype=bf, device=dev)
        uw = U[xi].reshape(co, ci, 1, 1).contiguous()
        l.check(lib.ustrun_pack_conv(uw.data_ptr(), co, ci, 1, p.data_ptr(), 1, None))
        packs.append(p)
        srcs.append(l.nhwc_src(V[xi].data_ptr(), ci, rows // 128, 128))          # any 2-D factorisation of the rows: a 1x1 conv does not care

    def gemms():
        for xi in range(16):
            l.check(lib.ustrun_conv2d_fwd(C.byref(srcs[xi]), 1, packs[xi].data_ptr(), None, 1, rows // 128, 128, co, 1, 1, 1,
                                          M[xi].data_ptr(), 0, None, None, 1, None))
    t_gemm = timed(gemms)
    Mf = M.float().reshape(4, 4, n, T, T, co)
    ATd = AT.to(dev)
    Y = torch.einsum("ij,jkntuc,lk->ntiulc", ATd, Mf, ATd).reshape(n, hw, hw, co)      # A^T M A, tiles back to pixels
    e_wino = float((Y.cpu().permute(0, 3, 1, 2).double() - ref).norm() / ref.norm())

    fl_direct = 2.0 * 9 * ci * co * n * hw * hw
    fl_wino = 2.0 * 16 * ci * co * rows
    by_tr = 2.0 * (n * hw * hw * ci + 16 * rows * ci) + 2.0 * (16 * rows * co + n * hw * hw * co)     # transforms as separate passes
    print(f"layer {ci} -> {co} at {hw} x {hw}, N = {n}, bf16")
    print(f"direct halo-tiled kernel      : {t_direct:.4f} ms  {fl_direct / t_direct / 1e9:6.0f} TF/s (direct flops)   rel-L2 vs f64 {e_direct:.3e}")
    print(f"Winograd F(2x2,3x3), 16 GEMMs : {t_gemm:.4f} ms  {fl_wino / t_gemm / 1e9:6.0f} TF/s over {fl_wino / 1e9:.0f} GFLOP "
          f"({fl_direct / fl_wino:.2f}x fewer than direct)   rel-L2 vs f64 {e_wino:.3e}")
    print(f"  + transforms as separate passes: {by_tr / 1e6:.0f} MB -> {by_tr / 5.4e9:.4f} ms at 5.4 TB/s  => composed {t_gemm + by_tr / 5.4e9:.4f} ms")
    print(f"  GEMM part alone vs direct: x{t_direct / t_gemm:.2f};  error ratio Winograd / direct: x{e_wino / e_direct:.1f}")
    print("decision rule (VERDICT r3 next 9): product wiring only if >= 1.25x at <= 2x the direct kernel's error")


if __name__ == "__main__":
    main()
