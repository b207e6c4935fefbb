"""A/B of the first convolution's forward (C = 3 -> 64 at 256 x 256, NCHW f32 input, bf16 output + BatchNorm statistics): the
streaming kernel (default) against the tile-per-block kernel of rounds 1-3 (ustrun_debug_flags bit 14), interleaved rounds in one
process; algorithmic bytes = 12 B read + 128 B written per pixel.    python tools/ab_first.py [--n 64] [--c 3] [--hw 256]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd"))
from ustrun import _lib as l  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--c", type=int, default=3)
    ap.add_argument("--hw", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    lib = l.lib()
    n, c, hw = a.n, a.c, a.hw
    x = torch.randn(n, c, hw, hw, device="cuda")
    wt = torch.randn(64, c, 3, 3, device="cuda") / 5
    wf, wd = torch.zeros(9 * 8 * 64, dtype=torch.bfloat16, device="cuda"), torch.zeros(9 * 8 * 64, dtype=torch.bfloat16, device="cuda")
    l.check(lib.ustrun_pack_conv3x3(wt.data_ptr(), 64, c, wf.data_ptr(), wd.data_ptr(), 1, None))
    src = l.nchw_src(x.data_ptr(), c, hw, hw)
    y = torch.empty(n, hw, hw, 64, device="cuda", dtype=torch.bfloat16)
    stat = torch.zeros(lib.ustrun_conv_mtiles(n, hw, hw, 64), 2, 64, device="cuda")
    fn = lambda: l.check(lib.ustrun_conv3x3_fwd(C.byref(src), 1, wf.data_ptr(), n, hw, hw, 64, y.data_ptr(), stat.data_ptr(), 1, None))
    by = n * hw * hw * (4.0 * c + 128.0)
    res = {0: [], 16384: []}
    outs = {}
    for r in range(a.rounds):
        for f in (0, 16384):
            lib.ustrun_debug_flags(f)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[f].append(e0.elapsed_time(e1) / 20)
            outs[f] = y.clone()
    lib.ustrun_debug_flags(0)
    for f, name in ((16384, "tile per block (rounds 1-3)"), (0, "streaming (round 4)")):
        t = float(np.median(res[f]))
        print(f"{name:30s} N={n} C={c} {hw}x{hw}: {t:.4f} ms (min {min(res[f]):.4f})  {by / t / 1e6:7.0f} GB/s = {by / t / 1e6 / 8000:.3f} of 8 TB/s")
    print("outputs bit-identical:", bool(torch.equal(outs[0], outs[16384])))


if __name__ == "__main__":
    main()
