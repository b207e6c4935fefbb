"""A/B of the first convolution's weight gradient (C = 3 -> 64 at 256 x 256, NCHW f32 input, 16-bit dY): the streaming kernel
(default) against the tile kernel of rounds 1-4 (ustrun_debug_flags bit 28), interleaved rounds in one process; algorithmic bytes =
128 B of dY + 12 B of x per pixel.    python tools/ab_first_wgrad.py [--n 64] [--c 3] [--hw 256]"""
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
    dy = torch.randn(n, hw, hw, 64, device="cuda").bfloat16()
    src = l.nchw_src(x.data_ptr(), c, hw, hw)
    nb = lib.ustrun_wgrad_partials_bytes(9, c, 64, n * hw * hw)
    part = torch.empty(nb // 4, device="cuda")
    dw = torch.empty(64, c, 3, 3, device="cuda")
    res = {0: [], 1 << 28: []}
    for r in range(a.rounds):
        for f in res:
            lib.ustrun_debug_flags(f)
            for _ in range(3):
                l.check(lib.ustrun_conv3x3_wgrad(C.byref(src), 1, dy.data_ptr(), n, hw, hw, 64, dw.data_ptr(), 0, part.data_ptr(), nb, 1, None))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                l.check(lib.ustrun_conv3x3_wgrad(C.byref(src), 1, dy.data_ptr(), n, hw, hw, 64, dw.data_ptr(), 0, part.data_ptr(), nb, 1, None))
            e1.record()
            torch.cuda.synchronize()
            res[f].append(e0.elapsed_time(e1) / 20)
    lib.ustrun_debug_flags(0)
    by = n * hw * hw * (128 + 4 * c)
    for f, name in ((1 << 28, "tile kernel (rounds 1-4)"), (0, "streaming kernel")):
        m = float(np.median(res[f]))
        print(f"{name:28s} {m * 1e3:8.1f} us per launch incl. the slab reduce   {by / m / 1e9:7.2f} TB/s algorithmic")


if __name__ == "__main__":
    main()
