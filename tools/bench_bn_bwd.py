#!/usr/bin/env python3
"""GPU box: the BatchNorm(+ReLU, +MaxPool routing) backward passes through the C ABI at the U-Net's shapes (bf16): time and
algorithmic GB/s of ustrun_bn_bwd_reduce (reads y, da [, dp]) and ustrun_bn_bwd_apply (the same reads + one write).

    python tools/bench_bn_bwd.py [--n 64] [--reps 10]
"""
import argparse
import os
import sys

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
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    lib = l.lib()
    dev, bf = "cuda", torch.bfloat16
    tot = [0.0, 0.0]
    for c, hw, pool in [(64, 256, False), (64, 256, True), (128, 128, False), (128, 128, True), (256, 64, False), (256, 64, True),
                        (512, 32, False), (512, 32, True), (1024, 16, False)]:
        n = a.n
        y = torch.randn(n, hw, hw, c, device=dev).to(bf)
        pad0 = torch.empty(69632 * 3, dtype=torch.uint8, device=dev)
        da = torch.randn(n, hw, hw, c, device=dev).to(bf)
        pad1 = torch.empty(69632 * 5, dtype=torch.uint8, device=dev)
        dp = torch.randn(n, hw // 2, hw // 2, c, device=dev).to(bf) if pool else None
        dz = torch.empty(n, hw, hw, c, device=dev, dtype=bf)
        sc, sh = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
        mean, rstd, gamma = torch.randn(c, device=dev) * 0.1, torch.rand(c, device=dev) + 0.5, torch.rand(c, device=dev) + 0.5
        dg, db, coef = torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.zeros(3 * c, device=dev)
        pb = lib.ustrun_bn_bwd_partials_bytes(n * hw * hw, c)
        part = torch.empty(pb // 4, device=dev)
        dpp = dp.data_ptr() if pool else None
        red = lambda: l.check(lib.ustrun_bn_bwd_reduce(da.data_ptr(), dpp, y.data_ptr(), sc.data_ptr(), sh.data_ptr(), mean.data_ptr(),
                                                       rstd.data_ptr(), gamma.data_ptr(), n, hw, hw, c, dg.data_ptr(), db.data_ptr(), 0,
                                                       coef.data_ptr(), part.data_ptr(), pb, 1, None))
        app = lambda: l.check(lib.ustrun_bn_bwd_apply(da.data_ptr(), dpp, y.data_ptr(), sc.data_ptr(), sh.data_ptr(), coef.data_ptr(),
                                                      n, hw, hw, c, dz.data_ptr(), 1, None))
        tr, ta = timed(red, a.reps), timed(app, a.reps)
        t = n * hw * hw * c * 2.0
        br = t * (2.25 if pool else 2.0)
        ba = br + t
        tot[0] += tr; tot[1] += ta
        print(f"C={c:4d} {hw:3d}x{hw:<3d} N={n} {'pool ' if pool else 'plain'}: reduce (+ finalize) {tr * 1e3:7.1f} us {br / tr / 1e6:6.0f} GB/s | "
              f"apply {ta * 1e3:7.1f} us {ba / ta / 1e6:6.0f} GB/s", flush=True)
    print(f"total: reduce {tot[0]:.3f} ms, apply {tot[1]:.3f} ms")


if __name__ == "__main__":
    main()
