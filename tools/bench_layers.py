"""Per-layer timing of the conv operators through the C ABI (bf16 path), at the bench workload's shapes
(fundus 256x256, base 64, N images per launch).  Prints TFLOP/s per layer for forward (with BN statistics
and the producer's affine+ReLU on load), input gradient and weight gradient, plus the ConvTranspose trio.
Development tool: tells which layer shapes fall short of the class average that bench.py reports.

    python tools/bench_layers.py [--n 16] [--reps 10] [--hw 256]
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


_spacers = []


def spacer():
    """Tensor sizes here are powers of two; two tensors that are a power of two apart and streamed in lockstep by one
    kernel share HBM channels (measured: 0.98 instead of 0.80 ms for the 128->64 concat conv).  The U-Net plan staggers its
    tensors; this tool does the same between its allocations."""
    _spacers.append(torch.empty((2 * (len(_spacers) % 8) + 1) * 69632, dtype=torch.uint8, device="cuda"))


def conv_layer(lib, n, ci, co, h, w, pool, cat, reps):
    """pool: the source is the 2x-resolution tensor pooled on load; cat: two sources of ci/2 channels."""
    dev = "cuda"
    bf = torch.bfloat16
    _spacers.clear()
    wt = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
    nel = 9 * ci * co
    wf, wd = torch.zeros(nel, dtype=bf, device=dev), torch.zeros(nel, dtype=bf, device=dev)
    l.check(lib.ustrun_pack_conv3x3(wt.data_ptr(), co, ci, wf.data_ptr(), wd.data_ptr(), 1, None))
    sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    keep = []
    if cat:
        c0 = ci // 2
        a0 = torch.randn(n, h, w, c0, device=dev).to(bf)
        spacer()
        a1 = torch.randn(n, h, w, ci - c0, device=dev).to(bf)
        srcs = (l.Src * 2)()
        srcs[0] = l.nhwc_src(a0.data_ptr(), c0, h, w, scale=sc.data_ptr(), shift=sh.data_ptr(), relu=1)
        srcs[1] = l.nhwc_src(a1.data_ptr(), ci - c0, h, w)
        nsrc = 2
        keep += [a0, a1]
    else:
        sh_, sw_ = (2 * h, 2 * w) if pool else (h, w)
        a0 = torch.randn(n, sh_, sw_, ci, device=dev).to(bf)
        srcs = (l.Src * 1)()
        srcs[0] = l.nhwc_src(a0.data_ptr(), ci, sh_, sw_, scale=sc.data_ptr(), shift=sh.data_ptr(), relu=1, pool=int(pool))
        nsrc = 1
        keep += [a0]
    spacer()
    y = torch.empty(n, h, w, co, device=dev, dtype=bf)
    spacer()
    dy = torch.randn(n, h, w, co, device=dev).to(bf)
    stat = torch.zeros(lib.ustrun_conv_mtiles(n, h, w, co), 2, co, device=dev)
    spacer()
    da = torch.empty(n, h, w, ci, device=dev, dtype=bf)
    nb = lib.ustrun_wgrad_partials_bytes(9, ci, co, n * h * w)
    part = torch.empty(nb // 4, device=dev)
    dw = torch.empty(co, ci, 3, 3, device=dev)
    fl = 2.0 * 9 * ci * co * n * h * w
    t_f = timed(lambda: l.check(lib.ustrun_conv3x3_fwd(srcs, nsrc, wf.data_ptr(), n, h, w, co, y.data_ptr(), stat.data_ptr(), 1, None)), reps)
    t_d = timed(lambda: l.check(lib.ustrun_conv3x3_dgrad(dy.data_ptr(), wd.data_ptr(), n, h, w, co, ci, da.data_ptr(), ci, None, 0, 0, 0, 0, 1, None)), reps)
    t_w = timed(lambda: l.check(lib.ustrun_conv3x3_wgrad(srcs, nsrc, dy.data_ptr(), n, h, w, co, dw.data_ptr(), 0, part.data_ptr(), nb, 1, None)), reps)
    return fl, t_f, t_d, t_w


def convT_layer(lib, n, ci, co, h, w, reps):
    dev, bf = "cuda", torch.bfloat16
    wt = torch.randn(ci, co, 2, 2, device=dev) / (2 * ci ** 0.5)
    b = torch.randn(co, device=dev)
    nel = 4 * ci * co
    wf, wd = torch.zeros(nel, dtype=bf, device=dev), torch.zeros(nel, dtype=bf, device=dev)
    l.check(lib.ustrun_pack_convT2x2(wt.data_ptr(), ci, co, wf.data_ptr(), wd.data_ptr(), 1, None))
    sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    a = torch.randn(n, h, w, ci, device=dev).to(bf)
    src = l.nhwc_src(a.data_ptr(), ci, h, w, scale=sc.data_ptr(), shift=sh.data_ptr(), relu=1)
    u = torch.empty(n, 2 * h, 2 * w, co, device=dev, dtype=bf)
    du = torch.randn(n, 2 * h, 2 * w, co, device=dev).to(bf)
    da = torch.empty(n, h, w, ci, device=dev, dtype=bf)
    nb = max(lib.ustrun_wgrad_partials_bytes(4, ci, co, n * h * w), 512 * co * 4)
    part = torch.empty(nb // 4, device=dev)
    dw, db = torch.empty(ci, co, 2, 2, device=dev), torch.empty(co, device=dev)
    fl = 2.0 * 4 * ci * co * n * h * w
    t_f = timed(lambda: l.check(lib.ustrun_convT2x2_fwd(C.byref(src), wf.data_ptr(), b.data_ptr(), n, h, w, co, u.data_ptr(), 1, None)), reps)
    t_d = timed(lambda: l.check(lib.ustrun_convT2x2_dgrad(du.data_ptr(), wd.data_ptr(), n, h, w, co, ci, da.data_ptr(), 1, None)), reps)
    t_w = timed(lambda: l.check(lib.ustrun_convT2x2_wgrad(C.byref(src), du.data_ptr(), n, h, w, co, dw.data_ptr(), db.data_ptr(), 0, part.data_ptr(), nb, 1, None)), reps)
    return fl, t_f, t_d, t_w


def ab_flags(a, lib):
    """the conv3x3 layers under each debug-flag value, interleaved (rule 24): median of --rounds rounds per (layer, flag)"""
    import numpy as np
    flags = a.flags.split(",")          # each "F" or "F:F2" = (ustrun_debug_flags, ustrun_debug_flags2)
    pair = lambda f: (int(f.split(":")[0]), int(f.split(":")[1]) if ":" in f else 0)
    S = a.hw
    layers = [("inc.2   64->64", 64, 64, S, False, False)]
    c, s = 64, S
    for i in range(4):
        s //= 2
        layers.append((f"down{i+1}.1 {c}->{2*c} pool", c, 2 * c, s, True, False))
        layers.append((f"down{i+1}.2 {2*c}->{2*c}", 2 * c, 2 * c, s, False, False))
        c *= 2
    for i in range(4):
        s *= 2
        layers.append((f"up{i+1}.1 cat {c}->{c//2}", c, c // 2, s, False, True))
        layers.append((f"up{i+1}.2 {c//2}->{c//2}", c // 2, c // 2, s, False, False))
        c //= 2
    tot = {f: [0.0, 0.0, 0.0] for f in flags}
    print("layer".ljust(28) + " | " + " | ".join(f"flags={f}: fwd / dgrad / wgrad ms (TF/s)" for f in flags))
    for name, ci, co, hw, pool, cat in layers:
        if a.only and a.only not in name:
            continue
        res = {f: [] for f in flags}
        for r in range(3):
            for f in flags:
                lib.ustrun_debug_flags(pair(f)[0]); lib.ustrun_debug_flags2(pair(f)[1])
                res[f].append(conv_layer(lib, a.n, ci, co, hw, hw, pool, cat, a.reps))
        lib.ustrun_debug_flags(0); lib.ustrun_debug_flags2(0)
        cells = []
        for f in flags:
            fl = res[f][0][0]
            med = [float(np.median([x[k] for x in res[f]])) for k in (1, 2, 3)]
            for k in range(3):
                tot[f][k] += med[k]
            cells.append(" / ".join(f"{m:.3f} ({fl / m / 1e9:4.0f})" for m in med))
        print(name.ljust(28) + " | " + " | ".join(cells), flush=True)
    print("total ms".ljust(28) + " | " + " | ".join(" / ".join(f"{v:.3f}" for v in tot[f]) for f in flags))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--hw", type=int, default=256)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--flags", type=str, default="", help="comma-separated ustrun_debug_flags values: A/B the builds they select, "
                    "interleaved per layer inside this process (e.g. 8,0)")
    a = ap.parse_args()
    lib = l.lib()
    if a.flags:
        return ab_flags(a, lib)
    S = a.hw
    layers = [("inc.2   64->64", 64, 64, S, False, False)]
    c, s = 64, S
    for i in range(4):
        s //= 2
        layers.append((f"down{i+1}.1 {c}->{2*c} pool", c, 2 * c, s, True, False))
        layers.append((f"down{i+1}.2 {2*c}->{2*c}", 2 * c, 2 * c, s, False, False))
        c *= 2
    for i in range(4):
        s *= 2
        layers.append((f"up{i+1}.1 cat {c}->{c//2}", c, c // 2, s, False, True))
        layers.append((f"up{i+1}.2 {c//2}->{c//2}", c // 2, c // 2, s, False, False))
        c //= 2
    tot = [0.0, 0.0, 0.0, 0.0]
    print(f"{'layer':28s} {'GF':>7s} | {'fwd ms':>7s} {'TF/s':>6s} | {'dgrad':>7s} {'TF/s':>6s} | {'wgrad':>7s} {'TF/s':>6s}")
    for name, ci, co, hw, pool, cat in layers:
        if a.only and a.only not in name:
            continue
        fl, tf, td, tw = conv_layer(lib, a.n, ci, co, hw, hw, pool, cat, a.reps)
        tot[0] += fl; tot[1] += tf; tot[2] += td; tot[3] += tw
        print(f"{name:28s} {fl/1e9:7.1f} | {tf:7.3f} {fl/tf/1e9:6.0f} | {td:7.3f} {fl/td/1e9:6.0f} | {tw:7.3f} {fl/tw/1e9:6.0f}", flush=True)
    if tot[0]:
        print(f"{'conv3x3 total':28s} {tot[0]/1e9:7.1f} | {tot[1]:7.3f} {tot[0]/tot[1]/1e9:6.0f} | {tot[2]:7.3f} {tot[0]/tot[2]/1e9:6.0f} | {tot[3]:7.3f} {tot[0]/tot[3]/1e9:6.0f}")
    c, s = 1024, S // 16
    tt = [0.0, 0.0, 0.0, 0.0]
    for i in range(4):
        name = f"up{i+1}.up convT {c}->{c//2} @{s}"
        if not a.only or a.only in name:
            fl, tf, td, tw = convT_layer(lib, a.n, c, c // 2, s, s, a.reps)
            tt[0] += fl; tt[1] += tf; tt[2] += td; tt[3] += tw
            print(f"{name:28s} {fl/1e9:7.1f} | {tf:7.3f} {fl/tf/1e9:6.0f} | {td:7.3f} {fl/td/1e9:6.0f} | {tw:7.3f} {fl/tw/1e9:6.0f}", flush=True)
        c //= 2
        s *= 2
    if tt[0]:
        print(f"{'convT total':28s} {tt[0]/1e9:7.1f} | {tt[1]:7.3f} {tt[0]/tt[1]/1e9:6.0f} | {tt[2]:7.3f} {tt[0]/tt[2]/1e9:6.0f} | {tt[3]:7.3f} {tt[0]/tt[3]/1e9:6.0f}")


if __name__ == "__main__":
    main()
