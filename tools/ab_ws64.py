#!/usr/bin/env python3
"""GPU box: A/B of the 64 -> 64 full-resolution convolution (forward with BatchNorm affine + ReLU on load and statistics;
input gradient) on the halo-tiled kernel (ustrun_debug_flags(1)) against the weight-stationary row-streaming kernel, in
interleaved rounds inside one process (cdna_hip_programming.md rule 24).  Prints median ms, TFLOP/s and algorithmic GB/s.

    python tools/ab_ws64.py [--hw 256] [--ns 16,48,64] [--rounds 7] [--reps 10]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd"))
from ustrun import _lib as l  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, default=256)
    ap.add_argument("--ns", default="16,48,64")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--diag", action="store_true", help="phase stamps of the streaming kernel's steady-state iteration")
    a = ap.parse_args()
    lib = l.lib()
    dev, bf = "cuda", torch.bfloat16
    ci = co = 64
    h = w = a.hw
    for n in [int(v) for v in a.ns.split(",")]:
        wt = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
        wf, wd = torch.zeros(9 * ci * co, dtype=bf, device=dev), torch.zeros(9 * ci * co, dtype=bf, device=dev)
        l.check(lib.ustrun_pack_conv3x3(wt.data_ptr(), co, ci, wf.data_ptr(), wd.data_ptr(), 1, None))
        sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
        x = torch.randn(n, h, w, ci, device=dev).to(bf)
        pad0 = torch.empty(69632 * 3, dtype=torch.uint8, device=dev)
        y = torch.empty(n, h, w, co, device=dev, dtype=bf)
        pad1 = torch.empty(69632 * 5, dtype=torch.uint8, device=dev)
        dy = torch.randn(n, h, w, co, device=dev).to(bf)
        pad2 = torch.empty(69632 * 7, dtype=torch.uint8, device=dev)
        da = torch.empty(n, h, w, ci, device=dev, dtype=bf)
        stat = torch.zeros(lib.ustrun_conv_mtiles(n, h, w, co), 2, co, device=dev)
        src = l.nhwc_src(x.data_ptr(), ci, h, w, scale=sc.data_ptr(), shift=sh.data_ptr(), relu=1)
        fwd = lambda: l.check(lib.ustrun_conv3x3_fwd(C.byref(src), 1, wf.data_ptr(), n, h, w, co, y.data_ptr(), stat.data_ptr(), 1, None))
        dgr = lambda: l.check(lib.ustrun_conv3x3_dgrad(dy.data_ptr(), wd.data_ptr(), n, h, w, co, ci, da.data_ptr(), ci, None, 0, 0, 0, 0, 1, None))
        res = {(f, op): [] for f in (1, 2, 4, 32) for op in ("fwd", "dgrad")}
        outs = {}
        for r in range(a.rounds):
            for f in (1, 2, 4, 32):
                lib.ustrun_debug_flags(f)
                res[(f, "fwd")].append(timed(fwd, a.reps))
                res[(f, "dgrad")].append(timed(dgr, a.reps))
                if r == 0:
                    outs[f] = (y.clone(), da.clone(), stat[:, :, :].sum(0).clone())
        lib.ustrun_debug_flags(0)
        same = all(torch.equal(outs[32][0], outs[f][0]) and torch.equal(outs[32][1], outs[f][1]) for f in (1, 2))
        fl = 2.0 * 9 * ci * co * n * h * w
        by = 2.0 * (2 * n * h * w * 64) + 2.0 * 9 * 64 * 64
        for op in ("fwd", "dgrad"):
            t1, t2, t0 = float(np.median(res[(1, op)])), float(np.median(res[(2, op)])), float(np.median(res[(4, op)]))
            print(f"N={n:3d} {h}x{w} {op:5s}: tiled {t1:.4f} ms {fl / t1 / 1e9:6.0f} TF/s | streaming, 4 waves {t2:.4f} ms {fl / t2 / 1e9:6.0f} TF/s "
                  f"{by / t2 / 1e6:6.0f} GB/s ({by / t2 / 1e6 / 8000:.3f}) | streaming, 8 waves {t0:.4f} ms "
                  f"{fl / t0 / 1e9:6.0f} TF/s {by / t0 / 1e6:6.0f} GB/s ({by / t0 / 1e6 / 8000:.3f} of 8 TB/s) | x{t1 / t0:.2f}   min {min(res[(4, op)]):.4f}", flush=True)
            tc = float(np.median(res[(32, op)]))
            print(f"       {op:5s}: consumer / producer waves {tc:.4f} ms {fl / tc / 1e9:6.0f} TF/s {by / tc / 1e6:6.0f} GB/s ({by / tc / 1e6 / 8000:.3f} of 8 TB/s)   "
                  f"min {min(res[(32, op)]):.4f}   x{t2 / tc:.2f} over four waves", flush=True)
        if a.diag:
            dbg = torch.zeros(2048 * 64, dtype=torch.int64, device=dev)      # 64 u64 per workgroup; the library refuses grids beyond the buffer
            l.check(lib.ustrun_debug_buffer(dbg.data_ptr(), dbg.numel()))
            lib.ustrun_debug_flags(4)
            for name, fn in (("fwd", fwd), ("dgrad", dgr)):
                dbg.zero_()
                fn()
                torch.cuda.synchronize()
                t_diag = timed(fn, 5)                      # wall time of the stamped build itself -> the in-kernel clock
                d = dbg[:256 * 64].view(256, 8, 8).double()
                it = d[..., 5].clamp(min=1)
                ph = [float((d[..., k] / it).mean()) for k in range(5)]
                tot = d[..., 1] + d[..., 2] + d[..., 4]
                print(f"       diag {name} (8 waves): cycles per iteration: MFMA groups + staging {ph[1]:.0f} | epilogue (stores, statistics) {ph[2]:.0f} | "
                      f"lgkmcnt + barrier {ph[4]:.0f} | sum {sum(ph):.0f} (MFMA floor at two waves per SIMD 4608); iterations per wave {float(it.mean()):.1f}; "
                      f"stamped launch {t_diag:.4f} ms -> in-kernel clock >= {float(tot.max()) / (t_diag * 1e-3) / 1e9:.2f} GHz; "
                      f"per wave half (0-3 / 4-7): groups {float((d[:, :4, 1] / it[:, :4]).mean()):.0f} / {float((d[:, 4:, 1] / it[:, 4:]).mean()):.0f}, "
                      f"barrier {float((d[:, :4, 4] / it[:, :4]).mean()):.0f} / {float((d[:, 4:, 4] / it[:, 4:]).mean()):.0f}", flush=True)
            lib.ustrun_debug_flags(32)
            for name, fn in (("fwd", fwd), ("dgrad", dgr)):
                dbg.zero_()
                fn()
                torch.cuda.synchronize()
                d = dbg[:256 * 64].view(256, 8, 8).double()
                it = d[..., 5].clamp(min=1)
                f = lambda lo, hi, k: float((d[:, lo:hi, k] / it[:, lo:hi]).mean())
                print(f"       diag {name} (consumer / producer waves), cycles per iteration -- consumers: half 0 {f(0, 4, 0):.0f}, wait B1 {f(0, 4, 1):.0f}, "
                      f"half 1 {f(0, 4, 2):.0f}, wait B2 {f(0, 4, 3):.0f} | producers: segment A {f(4, 8, 0):.0f}, wait B1 {f(4, 8, 1):.0f}, segment B "
                      f"{f(4, 8, 2):.0f}, wait B2 {f(4, 8, 3):.0f} (MFMA floor per half 2304)", flush=True)
            lib.ustrun_debug_buffer(None, 0)
            lib.ustrun_debug_flags(0)
        print(f"       outputs bit-identical between the kernels on this random float data (different summation orders; the exact-integer "
              f"tests are tests/test_gpu_production_tiles.py): {same}; stat sums rel diff "
              f"{float((outs[32][2] - outs[1][2]).abs().max() / outs[1][2].abs().max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
