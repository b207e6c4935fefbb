#!/usr/bin/env python3
"""GPU box: phase stamps of the two-group 3x3 weight-gradient kernel (wgrad_halo_pp_bf16_kernel<true>) on one layer shape:
cycles per tile a wave spends multiplying, preparing (transfers + activation) and waiting at the phase barriers.

    python tools/diag_wgrad.py [ci co hw n]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ustrun import _lib as l  # noqa: E402


def main():
    ci, co, hw, n = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (128, 128, 128, 64))]
    lib = l.lib()
    dev, bf = "cuda", torch.bfloat16
    sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    x = torch.randn(n, hw, hw, ci, device=dev).to(bf)
    pad = torch.empty(69632 * 3, dtype=torch.uint8, device=dev)
    dy = torch.randn(n, hw, hw, co, device=dev).to(bf)
    src = (l.Src * 1)()
    src[0] = l.nhwc_src(x.data_ptr(), ci, hw, hw, scale=sc.data_ptr(), shift=sh.data_ptr(), relu=1)
    nb = lib.ustrun_wgrad_partials_bytes(9, ci, co, n * hw * hw)
    part = torch.empty(nb // 4, device=dev)
    dw = torch.empty(co, ci, 3, 3, device=dev)
    run = lambda: l.check(lib.ustrun_conv3x3_wgrad(src, 1, dy.data_ptr(), n, hw, hw, co, dw.data_ptr(), 0, part.data_ptr(), nb, 1, None))

    def timed(reps=10):
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    t = timed()
    fl = 2.0 * 9 * ci * co * n * hw * hw
    print(f"{ci}->{co} {hw}x{hw} N={n}: {t:.4f} ms {fl / t / 1e9:.0f} TF/s (kernel + slab reduce), variant {lib.ustrun_debug_last_wgrad_variant():#x}")
    dbg = torch.zeros(2048 * 64, dtype=torch.int64, device=dev)      # 64 u64 per workgroup; the library refuses grids beyond the buffer
    l.check(lib.ustrun_debug_buffer(dbg.data_ptr(), dbg.numel()))
    run()
    torch.cuda.synchronize()
    td = timed(5)
    lib.ustrun_debug_buffer(None, 0)
    d = dbg[:256 * 64].view(256, 8, 8).double()
    tiles = d[..., 5].clamp(min=1)
    f = lambda k: float((d[..., k] / tiles).mean())
    tot = d[..., 0] + d[..., 1] + d[..., 2] + d[..., 3] + d[..., 4]
    print(f"   stamped build {td:.4f} ms; per tile and wave (cycles; 72 MFMAs = 2304): multiplying {f(0):.0f}, barrier after it {f(1):.0f}, "
          f"preparing {f(2):.0f} (issuing the next tile's transfers {f(7):.0f}, waiting for this tile's {f(6):.0f}, activating {f(2) - f(6) - f(7):.0f}), barrier after it {f(3):.0f}, phases without work {f(4):.0f}; tiles per wave {float(tiles.mean()):.1f}; "
          f"in-kernel clock >= {float(tot.max()) / (td * 1e-3) / 1e9:.2f} GHz")


if __name__ == "__main__":
    main()
