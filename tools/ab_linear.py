"""GPU box: the 3x3 layers of the maps no rectangular tile divides (configs[2] / configs[3]: 24 px of prostate 384^2; 72 / 36 / 18 px
of M&Ms 288^2), forward (BatchNorm + ReLU on load, statistics) and input gradient, on the halo kernel's linear tiles (round 5) and
-- ustrun_debug_flags2 bit 0 -- on the rectangular tile the padding rule picks.  Same buffers, alternating.

    python tools/ab_linear.py [--b 8] [--reps 20]
"""
import argparse
import ctypes as C
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
    ap.add_argument("--b", type=int, default=8, help="images per pass (the student's call: 5 passes + 1 image forward, 4 passes backward)")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    lib = l.lib()
    bf, dev = torch.bfloat16, "cuda"
    shapes = [(512, 1024, 18), (1024, 1024, 18), (256, 512, 36), (512, 512, 36), (1024, 512, 36), (128, 256, 72), (256, 256, 72),
              (512, 256, 72), (512, 1024, 24), (1024, 1024, 24)]
    tot = {0: [0.0, 0.0], 1: [0.0, 0.0]}
    for ci, co, hw in shapes:
        nf, nb = 5 * a.b + 1, 4 * a.b
        wt = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
        wf, wd = torch.zeros(9 * ci * co, dtype=bf, device=dev), torch.zeros(9 * ci * co, dtype=bf, device=dev)
        l.check(lib.ustrun_pack_conv3x3(wt.data_ptr(), co, ci, wf.data_ptr(), wd.data_ptr(), 1, None))
        x = torch.randn(nf, hw, hw, ci, device=dev).to(bf)
        aff = torch.rand(6, 4, ci, device=dev) + 0.5
        src = l.nhwc_src(x.data_ptr(), ci, hw, hw, aff.data_ptr(), aff.data_ptr() + 4 * ci, relu=1, gN=a.b, gstride=4 * ci)
        y = torch.empty(nf, hw, hw, co, device=dev, dtype=bf)
        dy = torch.randn(nb, hw, hw, co, device=dev).to(bf)
        da = torch.empty(nb, hw, hw, ci, device=dev, dtype=bf)
        stat = torch.zeros(lib.ustrun_conv_mtiles(nf, hw, hw, co), 2, co, device=dev)
        rows = C.c_int(0)
        fwd = lambda: l.check(lib.ustrun_conv3x3_fwd_rows(C.byref(src), 1, wf.data_ptr(), nf, hw, hw, co, y.data_ptr(), stat.data_ptr(),
                                                          C.byref(rows), 1, None))
        dgr = lambda: l.check(lib.ustrun_conv3x3_dgrad(dy.data_ptr(), wd.data_ptr(), nb, hw, hw, co, ci, da.data_ptr(), ci, None, 0, 0, 0, 0, 1, None))
        res, outs = {}, {}
        for rect in (1, 0, 1, 0):
            old = lib.ustrun_debug_flags2(1 if rect else 0)
            tf, vf = timed(fwd, a.reps), lib.ustrun_debug_last_conv_variant()
            yo = y.clone()
            td, vd = timed(dgr, a.reps), lib.ustrun_debug_last_conv_variant()
            lib.ustrun_debug_flags2(old)
            res[rect] = (min(tf, res[rect][0]) if rect in res else tf, min(td, res[rect][1]) if rect in res else td, vf, vd)
            outs[rect] = (yo, da.clone())
        same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        flf, flb = 2.0 * nf * hw * hw * 9 * ci * co, 2.0 * nb * hw * hw * 9 * ci * co
        for rect in (1, 0):
            tot[rect][0] += res[rect][0]; tot[rect][1] += res[rect][1]
        print(f"{ci:5d}->{co:4d} @{hw:2d}: fwd n={nf} rect {res[1][0]:.3f} ms ({flf / res[1][0] / 1e9:5.0f} TF/s) -> linear {res[0][0]:.3f} ({flf / res[0][0] / 1e9:5.0f})"
              f" [{res[0][2]:#x}]   dgrad n={nb} {res[1][1]:.3f} ({flb / res[1][1] / 1e9:5.0f}) -> {res[0][1]:.3f} ({flb / res[0][1] / 1e9:5.0f}) [{res[0][3]:#x}]"
              f"   bitwise equal: {same}", flush=True)
    print(f"sum: fwd {tot[1][0]:.3f} -> {tot[0][0]:.3f} ms, dgrad {tot[1][1]:.3f} -> {tot[0][1]:.3f} ms")


if __name__ == "__main__":
    main()
