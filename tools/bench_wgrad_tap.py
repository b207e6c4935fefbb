#!/usr/bin/env python3
"""The DeepLabV2-ResNet101 weight-gradient shapes (BASELINE.json configs[4]: 512^2 -> 65 x 65 maps at output stride 8) through
ustrun_conv2d_wgrad, one at a time: ms and TFLOP/s per shape, weighted by how often the network runs it.  Development tool; under
rocprofv3 --pmc (tools/pmc_wgrad_tap.sh) with --only it gives the SQ counters of exactly one shape.

    python3 tools/bench_wgrad_tap.py [--n 16] [--reps 5] [--only NAME] [--plain 0|1]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd"))
from ustrun import _lib as l  # noqa: E402

# (name, Cin, Cout, k, dilation, H = W, count in ResNet-101 with layer3 / layer4 dilated)
SHAPES = [("l3.conv1 1024->256 1x1", 1024, 256, 1, 1, 65, 22), ("l3.conv2 256->256 3x3 d2", 256, 256, 3, 2, 65, 23),
          ("l3.conv3 256->1024 1x1", 256, 1024, 1, 1, 65, 23), ("l4.conv1 2048->512 1x1", 2048, 512, 1, 1, 65, 2),
          ("l4.conv2 512->512 3x3 d4", 512, 512, 3, 4, 65, 3), ("l4.conv3 512->2048 1x1", 512, 2048, 1, 1, 65, 3),
          ("l2.conv1 512->128 1x1", 512, 128, 1, 1, 65, 3), ("l2.conv3 128->512 1x1", 128, 512, 1, 1, 65, 4),
          ("l1.conv1 256->64 1x1", 256, 64, 1, 1, 129, 2), ("l1.conv3 64->256 1x1", 64, 256, 1, 1, 129, 3)]


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
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--plain", type=int, default=0, help="1: the source is a finished activation (no BatchNorm + ReLU on load)")
    a = ap.parse_args()
    lib = l.lib()
    bf = torch.bfloat16
    tot_ms = tot_fl = 0.0
    print(f"{'shape':28s} {'GF':>7s} {'ms':>7s} {'TF/s':>6s} {'x':>3s}  variant")
    for name, ci, co, k, d, hw, cnt in SHAPES:
        if a.only and a.only not in name:
            continue
        x = torch.randn(a.n, hw, hw, ci, device="cuda").to(bf)
        dy = torch.randn(a.n, hw, hw, co, device="cuda").to(bf)
        sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
        src = l.nhwc_src(x.data_ptr(), ci, hw, hw) if a.plain else l.nhwc_src(x.data_ptr(), ci, hw, hw, sc.data_ptr(), sh.data_ptr(), relu=1)
        pb = lib.ustrun_wgrad_partials_bytes(k * k, ci, co, a.n * hw * hw)
        part = torch.empty(pb // 4, device="cuda")
        dw = torch.empty(co, ci, k, k, device="cuda")
        fn = lambda: l.check(lib.ustrun_conv2d_wgrad(C.byref(src), 1, dy.data_ptr(), a.n, hw, hw, co, k, 1, d, dw.data_ptr(), 0, part.data_ptr(),
                                                     pb, 1, None), "wgrad")
        ms = timed(fn, a.reps)
        fl = 2.0 * a.n * hw * hw * ci * co * k * k
        print(f"{name:28s} {fl / 1e9:7.1f} {ms:7.3f} {fl / ms / 1e9:6.0f} {cnt:3d}  {lib.ustrun_debug_last_wgrad_variant():#x}", flush=True)
        tot_ms += ms * cnt
        tot_fl += fl * cnt
    if tot_ms:
        print(f"{'network total (weighted)':28s} {tot_fl / 1e9:7.1f} {tot_ms:7.3f} {tot_fl / tot_ms / 1e9:6.0f}")


if __name__ == "__main__":
    main()
