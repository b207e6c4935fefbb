"""GPU box: phase stamps of the ConvTranspose weight-gradient kernel (wgradT2_bf16_kernel<256, true>, launched while
ustrun_debug_buffer is set) on one layer shape: cycles per 32-pixel stage a wave spends issuing transfers, reading fragments +
multiplying (16 MFMAs = 512 cycles of matrix pipe), waiting for the next stage, activating it in place, and at the barrier.

    python tools/diag_wgradT.py [ci co hw n]        (ci % 256 == 0: the 256-ci tile)
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd"))
from ustrun import _lib as l  # noqa: E402


def main():
    ci, co, hw, n = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (1024, 512, 16, 64))]
    lib = l.lib()
    dev, bf = "cuda", torch.bfloat16
    for aff in (True, False):
        sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
        x = torch.randn(n, hw, hw, ci, device=dev).to(bf)
        du = torch.randn(n, 2 * hw, 2 * hw, co, device=dev).to(bf)
        src = l.nhwc_src(x.data_ptr(), ci, hw, hw, scale=sc.data_ptr() if aff else None, shift=sh.data_ptr() if aff else None, relu=1 if aff else 0)
        nb = max(lib.ustrun_wgrad_partials_bytes(4, ci, co, n * hw * hw), 512 * co * 4)
        part = torch.empty(nb // 4, device=dev)
        dw, db = torch.empty(ci, co, 2, 2, device=dev), torch.empty(co, device=dev)
        run = lambda: l.check(lib.ustrun_convT2x2_wgrad(C.byref(src), du.data_ptr(), n, hw, hw, co, dw.data_ptr(), db.data_ptr(), 0, part.data_ptr(), nb, 1, None))

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
        fl = 2.0 * 4 * ci * co * n * hw * hw
        print(f"{ci}->{co} x4 {hw}x{hw} N={n} {'BatchNorm + ReLU on load' if aff else 'plain source'}: {t:.4f} ms {fl / t / 1e9:.0f} TF/s (kernel + reduces), "
              f"variant {lib.ustrun_debug_last_wgrad_variant():#x}")
        dbg = torch.zeros(1024 * 64, dtype=torch.int64, device=dev)
        l.check(lib.ustrun_debug_buffer(dbg.data_ptr(), dbg.numel()))
        run()
        torch.cuda.synchronize()
        td = timed(5)
        lib.ustrun_debug_buffer(None, 0)
        nblk = int((dbg.view(-1, 8, 8)[:, 0, 5] > 0).sum())
        d = dbg[:nblk * 64].view(nblk, 8, 8).double()
        st = d[..., 5].clamp(min=1)
        f = lambda k: float((d[..., k] / st).mean())
        print(f"   stamped build {td:.4f} ms, {nblk} workgroups; per stage and wave (cycles): issuing transfers {f(0):.0f}, fragments + 16 MFMAs {f(1):.0f}, "
              f"wait for the next stage {f(2):.0f}, activation {f(3):.0f}, barrier {f(4):.0f}; sum {sum(f(k) for k in range(5)):.0f}; stages per wave {float(st.mean()):.1f}")


if __name__ == "__main__":
    main()
