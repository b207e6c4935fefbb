#!/usr/bin/env python3
"""GPU box: the ConvTranspose layers of the U-Net (forward / input gradient / weight gradient through the C ABI) under several
ustrun_debug_flags values, interleaved in one process.    python tools/ab_convT.py --flags 0,512,1024 [--n 64]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_layers import convT_layer, l  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", default="0,512")
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    lib = l.lib()
    flags = [int(v) for v in a.flags.split(",")]
    tot = {f: np.zeros(3) for f in flags}
    for name, ci, co, hw in [("up1.up 1024->512 @16", 1024, 512, 16), ("up2.up 512->256 @32", 512, 256, 32),
                             ("up3.up 256->128 @64", 256, 128, 64), ("up4.up 128->64 @128", 128, 64, 128)]:
        res = {f: [] for f in flags}
        for r in range(3):
            for f in flags:
                lib.ustrun_debug_flags(f)
                res[f].append(convT_layer(lib, a.n, ci, co, hw, hw, a.reps))
        lib.ustrun_debug_flags(0)
        cells = []
        for f in flags:
            fl = res[f][0][0]
            med = np.array([float(np.median([x[k] for x in res[f]])) for k in (1, 2, 3)])
            tot[f] += med
            cells.append(" / ".join(f"{t:.3f} ({fl / t / 1e9:4.0f})" for t in med))
        print(name.ljust(24) + " | " + " | ".join(cells), flush=True)
    print("total ms".ljust(24) + " | " + " | ".join(" / ".join(f"{v:.3f}" for v in tot[f]) for f in flags))


if __name__ == "__main__":
    main()
