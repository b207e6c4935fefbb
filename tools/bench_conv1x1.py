#!/usr/bin/env python3
"""The DeepLabV2-ResNet101 1x1 convolutions (BASELINE.json configs[4]: 512^2, 65 x 65 maps at output stride 8) through
ustrun_conv2d_fwd (train mode: statistics rows out) and ustrun_conv1x1_dgrad_join, one shape at a time: ms, TFLOP/s and the
algorithmic GB/s of each launch beside the two roofs.  Development tool.

    python3 tools/bench_conv1x1.py [--n 16] [--reps 10] [--only NAME] [--ops "fwd plain,dgrad join"]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd"))
from ustrun import _lib as l  # noqa: E402

# (name, Cin, Cout, H = W, count per forward in ResNet-101)
SHAPES = [("l3.conv1 1024->256", 1024, 256, 65, 22), ("l3.conv3 256->1024", 256, 1024, 65, 23),
          ("l4.conv1 2048->512", 2048, 512, 65, 2), ("l4.conv3 512->2048", 512, 2048, 65, 3),
          ("l2.conv1 512->128", 512, 128, 65, 3), ("l2.conv3 128->512", 128, 512, 65, 4),
          ("l1.conv1 256->64", 256, 64, 129, 2), ("l1.conv3 64->256", 64, 256, 129, 3)]


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
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    ap.add_argument("--ops", default="", help="comma-separated subset of: fwd affine, fwd plain, dgrad plain, dgrad join")
    a = ap.parse_args()
    lib = l.lib()
    want_ops = [o.strip() for o in a.ops.split(",") if o.strip()]
    bf = torch.bfloat16
    dt = l.BF16
    print(f"{'shape':22s} {'op':>12s} {'GF':>6s} {'MB':>6s} {'ms':>7s} {'TF/s':>6s} {'GB/s':>6s} {'x':>3s}")
    tot = {}
    for name, ci, co, hw, cnt in SHAPES:
        if a.only and a.only not in name:
            continue
        N = a.n
        M = N * hw * hw
        x = torch.randn(N, hw, hw, ci, device="cuda").to(bf)
        w = torch.randn(co, ci, 1, 1, device="cuda") / ci ** 0.5
        ne = lib.ustrun_pack_conv_elems(co, ci, 1)
        wf = torch.zeros(ne, dtype=bf, device="cuda")
        l.check(lib.ustrun_pack_conv(w.data_ptr(), co, ci, 1, wf.data_ptr(), dt, None), "pack")
        wt = w.permute(1, 0, 2, 3).contiguous()                   # the input gradient's GEMM: [ci][co]
        ne2 = lib.ustrun_pack_conv_elems(ci, co, 1)
        wd = torch.zeros(ne2, dtype=bf, device="cuda")
        l.check(lib.ustrun_pack_conv(wt.data_ptr(), ci, co, 1, wd.data_ptr(), dt, None), "pack")
        sc, sh = torch.rand(ci, device="cuda") + 0.5, torch.randn(ci, device="cuda") * 0.1
        y = torch.empty(N, hw, hw, co, dtype=bf, device="cuda")
        rows = lib.ustrun_conv_mtiles(N, hw, hw, co)
        stat = torch.empty(rows, 2, co, device="cuda")
        used = C.c_int(0)
        fl = 2.0 * M * ci * co
        for op in ("fwd affine", "fwd plain"):
            if want_ops and op not in want_ops:
                continue
            src = l.nhwc_src(x.data_ptr(), ci, hw, hw, sc.data_ptr(), sh.data_ptr(), relu=1) if op == "fwd affine" else l.nhwc_src(x.data_ptr(), ci, hw, hw)
            fn = lambda: l.check(lib.ustrun_conv2d_fwd(C.byref(src), 1, wf.data_ptr(), None, N, hw, hw, co, 1, 1, 1, y.data_ptr(), 0,
                                                       stat.data_ptr(), C.byref(used), dt, None), "fwd")
            ms = timed(fn, a.reps)
            by = 2.0 * M * (ci + co) + 2.0 * ci * co
            print(f"{name:22s} {op:>12s} {fl / 1e9:6.1f} {by / 1e6:6.0f} {ms:7.3f} {fl / ms / 1e9:6.0f} {by / ms / 1e6:6.0f} {cnt:3d}", flush=True)
            tot[op] = tot.get(op, 0.0) + ms * cnt
        # input gradient dx[M][ci] = dy[M][co] . W: plain, and with the residual join + bn3 sums (conv1 of a bottleneck)
        dy = torch.randn(N, hw, hw, co, device="cuda").to(bf)
        dx = torch.empty(N, hw, hw, ci, dtype=bf, device="cuda")
        add = torch.randn(N, hw, hw, ci, device="cuda").to(bf)
        ref = torch.randn(N, hw, hw, ci, device="cuda").to(bf)
        y3 = torch.randn(N, hw, hw, ci, device="cuda").to(bf)
        rows2 = lib.ustrun_conv_mtiles(N, hw, hw, ci)
        stat2 = torch.empty(rows2, 2, ci, device="cuda")
        u2, fused = C.c_int(0), C.c_int(0)
        dsrc = l.nhwc_src(dy.data_ptr(), co, hw, hw)
        for op in ("dgrad plain", "dgrad join"):
            if want_ops and op not in want_ops:
                continue
            if op == "dgrad join":
                fn = lambda: l.check(lib.ustrun_conv1x1_dgrad_join(dy.data_ptr(), wd.data_ptr(), N, hw, hw, co, ci, add.data_ptr(), ref.data_ptr(),
                                                                   dx.data_ptr(), y3.data_ptr(), None, None, stat2.data_ptr(), C.byref(u2),
                                                                   C.byref(fused), dt, None), "dgrad")
            else:
                fn = lambda: l.check(lib.ustrun_conv2d_fwd(C.byref(dsrc), 1, wd.data_ptr(), None, N, hw, hw, ci, 1, 1, 1, dx.data_ptr(), 0, None, None,
                                                           dt, None), "dgrad")
            ms = timed(fn, a.reps)
            if op == "dgrad join" and not fused.value:
                print(f"{name:22s} {op:>12s} -- not fused")
                continue
            by = 2.0 * M * (ci + co) + 2.0 * ci * co + (3 * 2.0 * M * ci if op == "dgrad join" else 0)
            print(f"{name:22s} {op:>12s} {fl / 1e9:6.1f} {by / 1e6:6.0f} {ms:7.3f} {fl / ms / 1e9:6.0f} {by / ms / 1e6:6.0f} {cnt:3d}", flush=True)
            tot[op] = tot.get(op, 0.0) + ms * cnt
    for k, v in tot.items():
        print(f"network total (weighted) {k}: {v:.3f} ms")


if __name__ == "__main__":
    main()
