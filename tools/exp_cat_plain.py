"""What a per-chunk switch between transforming and direct staging could buy on the four concat convolutions: the
production form (skip half through BatchNorm + ReLU on load, ConvTranspose half plain -- every chunk pays the transform) against the
same layer with BOTH halves plain (no chunk pays it; the mixed kernel would sit in between), forward with statistics, N = 64.

    python tools/exp_cat_plain.py [--n 64] [--reps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd"))
from ustrun import _lib as l  # noqa: E402
from bench_layers import timed, spacer, _spacers  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lib = l.lib()
    dev, bf = "cuda", torch.bfloat16
    n = a.n
    for name, ci, co, hw in (("up1.conv1", 1024, 512, 32), ("up2.conv1", 512, 256, 64), ("up3.conv1", 256, 128, 128), ("up4.conv1", 128, 64, 256)):
        _spacers.clear()
        wt = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
        wf, wd = torch.zeros(9 * ci * co, dtype=bf, device=dev), torch.zeros(9 * ci * co, dtype=bf, device=dev)
        l.check(lib.ustrun_pack_conv3x3(wt.data_ptr(), co, ci, wf.data_ptr(), wd.data_ptr(), 1, None))
        c0 = ci // 2
        sc, sh = torch.rand(c0, device=dev) + 0.5, torch.randn(c0, device=dev) * 0.1
        a0 = torch.randn(n, hw, hw, c0, device=dev).to(bf); spacer()
        a1 = torch.randn(n, hw, hw, c0, device=dev).to(bf); spacer()
        y = torch.empty(n, hw, hw, co, device=dev, dtype=bf)
        stat = torch.zeros(lib.ustrun_conv_mtiles(n, hw, hw, co), 2, co, device=dev)
        res = []
        for kind in ("xf+plain", "plain+plain"):
            srcs = (l.Src * 2)()
            srcs[0] = (l.nhwc_src(a0.data_ptr(), c0, hw, hw, scale=sc.data_ptr(), shift=sh.data_ptr(), relu=1) if kind == "xf+plain"
                       else l.nhwc_src(a0.data_ptr(), c0, hw, hw))
            srcs[1] = l.nhwc_src(a1.data_ptr(), c0, hw, hw)
            for flags in (0, 32768):
                old = lib.ustrun_debug_flags(flags)
                t = timed(lambda: l.check(lib.ustrun_conv3x3_fwd(srcs, 2, wf.data_ptr(), n, hw, hw, co, y.data_ptr(), stat.data_ptr(), 1, None)), a.reps)
                v = lib.ustrun_debug_last_conv_variant()
                lib.ustrun_debug_flags(old)
                res.append(f"{kind} flags={flags}: {t:.4f} ms (variant {v:#x})")
        print(f"{name} {ci}->{co} @{hw}: " + " | ".join(res), flush=True)


if __name__ == "__main__":
    main()
