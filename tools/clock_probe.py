"""The shader clock the chip HOLDS while it runs a kernel back to back (MI355X_MICROARCH.md "DVFS give-back"
item 6): `ustrun_debug_clock_probe` records s_memtime (one tick per shader cycle) and s_memrealtime (100 MHz) per workgroup; a
probe before and one after >= 2 s of back-to-back launches of ONE kernel give d(memtime) / d(memrealtime) x 100 MHz per XCD --
the product kernels themselves, no stamped build.  Reported next to the achieved rate, for the kernels DESIGN 8.8 reasons about:
the halo-tiled 3x3 convolution (forward with BatchNorm + ReLU on load and statistics; input gradient), the all-taps weight
gradient (two wave groups), the 64 -> 64 streaming kernel (consumer / producer waves), at N = 64 images; plus two HBM-bound
passes for contrast.

    python tools/clock_probe.py [--seconds 2.0] [--n 64]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd"))
from ustrun import _lib as l  # noqa: E402

NB = 1024


def probe(lib):
    buf = torch.zeros(NB * 4, dtype=torch.int64, device="cuda")
    l.check(lib.ustrun_debug_clock_probe(buf.data_ptr(), NB, None))
    return buf


def clock_mhz(a, b):
    """per-XCD d(memtime) / d(memrealtime) x 100 MHz from two probes (any workgroup of an XCD speaks for it: the counters are
    read in one instruction pair); returns (median over XCDs, min, max)"""
    a, b = a.view(NB, 4).cpu().numpy(), b.view(NB, 4).cpu().numpy()
    out = []
    for x in range(8):
        ia, ib = np.nonzero(a[:, 2] == x)[0], np.nonzero(b[:, 2] == x)[0]
        if len(ia) == 0 or len(ib) == 0:
            continue
        dt = float(np.median(b[ib, 0])) - float(np.median(a[ia, 0]))
        dr = float(np.median(b[ib, 1])) - float(np.median(a[ia, 1]))
        out.append(dt / dr * 100.0)
    return float(np.median(out)), float(min(out)), float(max(out)), len(out)


def measure(lib, name, fn, flops, nbytes, seconds):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    per = e0.elapsed_time(e1) / 20
    reps = max(50, int(seconds * 1e3 / per))
    for _ in range(reps // 4):                   # warm: the clock settles under the load before the first probe
        fn()
    pa = probe(lib)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    pb = probe(lib)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    mhz, lo, hi, nx = clock_mhz(pa, pb)
    sustained = mhz * 1e6 * 4096 * 256 / 1e12            # bf16 MFMA peak at that clock (TF/s): 4096 flop / clk / CU x 256 CUs
    tf = flops / ms / 1e9
    print(f"{name:58s} {ms:7.4f} ms  {tf:7.1f} TF/s  {nbytes / ms / 1e6:7.0f} GB/s | clock {mhz:6.0f} MHz ({lo:.0f}-{hi:.0f} over {nx} XCDs) "
          f"-> MFMA peak at that clock {sustained:6.0f} TF/s, achieved / that = {tf / sustained:.3f} (of 2500: {tf / 2500:.3f})", flush=True)
    return mhz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--n", type=int, default=64)
    a = ap.parse_args()
    lib = l.lib()
    dev, bf = "cuda", torch.bfloat16
    n = a.n

    def conv_case(ci, co, hw):
        wt = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
        nel = 9 * ci * co
        wf, wd = torch.zeros(nel, dtype=bf, device=dev), torch.zeros(nel, dtype=bf, device=dev)
        l.check(lib.ustrun_pack_conv3x3(wt.data_ptr(), co, ci, wf.data_ptr(), wd.data_ptr(), 1, None))
        sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
        x = torch.randn(n, hw, hw, ci, device=dev).to(bf)
        pad0 = torch.empty(69632 * 3, dtype=torch.uint8, device=dev)
        y = torch.empty(n, hw, hw, co, device=dev, dtype=bf)
        pad1 = torch.empty(69632 * 5, dtype=torch.uint8, device=dev)
        dy = torch.randn(n, hw, hw, co, device=dev).to(bf)
        pad2 = torch.empty(69632 * 7, dtype=torch.uint8, device=dev)
        da = torch.empty(n, hw, hw, ci, device=dev, dtype=bf)
        stat = torch.zeros(lib.ustrun_conv_mtiles(n, hw, hw, co), 2, co, device=dev)
        srcs = (l.Src * 1)()
        srcs[0] = l.nhwc_src(x.data_ptr(), ci, hw, hw, scale=sc.data_ptr(), shift=sh.data_ptr(), relu=1)
        nb = lib.ustrun_wgrad_partials_bytes(9, ci, co, n * hw * hw)
        part = torch.empty(nb // 4, device=dev)
        dw = torch.empty(co, ci, 3, 3, device=dev)
        keep = [wt, wf, wd, sc, sh, x, y, dy, da, stat, part, dw, pad0, pad1, pad2, srcs]
        fl = 2.0 * 9 * ci * co * n * hw * hw
        by = 2.0 * n * hw * hw * (ci + co) + 2.0 * nel
        fwd = lambda: l.check(lib.ustrun_conv3x3_fwd(srcs, 1, wf.data_ptr(), n, hw, hw, co, y.data_ptr(), stat.data_ptr(), 1, None))
        dgr = lambda: l.check(lib.ustrun_conv3x3_dgrad(dy.data_ptr(), wd.data_ptr(), n, hw, hw, co, ci, da.data_ptr(), ci, None, 0, 0, 0, 0, 1, None))
        wgr = lambda: l.check(lib.ustrun_conv3x3_wgrad(srcs, 1, dy.data_ptr(), n, hw, hw, co, dw.data_ptr(), 0, part.data_ptr(), nb, 1, None))
        return fl, by, fwd, dgr, wgr, keep

    print(f"N = {n}, >= {a.seconds} s of back-to-back launches per row; random data", flush=True)
    t0 = time.time()
    fl, by, fwd, dgr, wgr, keep = conv_case(512, 512, 32)
    measure(lib, "halo 3x3 512->512 @32x32 forward (BN+ReLU on load, stats)", fwd, fl, by, a.seconds)
    measure(lib, "halo 3x3 512->512 @32x32 input gradient", dgr, fl, by, a.seconds)
    measure(lib, "all-taps weight gradient 512x512 @32x32 (two wave groups)", wgr, fl, by + 4.0 * 9 * 512 * 512, a.seconds)
    del keep
    fl, by, fwd, dgr, wgr, keep = conv_case(64, 64, 256)
    measure(lib, "streaming 64->64 @256x256 forward (consumer/producer waves)", fwd, fl, by, a.seconds)
    measure(lib, "streaming 64->64 @256x256 input gradient", dgr, fl, by, a.seconds)
    measure(lib, "all-taps weight gradient 64x64 @256x256", wgr, fl, by, a.seconds)
    del keep
    fl, by, fwd, dgr, wgr, keep = conv_case(128, 128, 128)
    measure(lib, "halo 3x3 128->128 @128x128 forward", fwd, fl, by, a.seconds)
    measure(lib, "halo 3x3 128->128 @128x128 input gradient", dgr, fl, by, a.seconds)
    del keep
    # an HBM-bound pass for contrast: the fused SGD + EMA update over 31 M parameters (28 B / parameter)
    P = 31037698
    p, g, v, t = (torch.randn(P, device=dev) for _ in range(4))
    upd = lambda: l.check(lib.ustrun_sgd_ema(p.data_ptr(), g.data_ptr(), v.data_ptr(), t.data_ptr(), P, 0.01, 0.9, 1e-4, 0, 0.99, 1.0, None))
    measure(lib, "sgd_ema over 31 M parameters (HBM-bound, no MFMA)", upd, 0.0, 28.0 * P, a.seconds)
    print(f"total {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
