"""GPU box, under rocprofv3 --pmc (or alone: prints ms and TFLOP/s-equivalent): a few launches of one 3x3 layer under dtype f32x3 --
forward with BatchNorm + ReLU on load, input gradient, weight gradient (csrc/x3.hip).

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU -d gpurun_out/pmc_x3 -- python3 tools/pmc_x3.py [ci co hw n reps]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd"))
from ustrun import _lib as l  # noqa: E402

ci, co, hw, n, reps = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (512, 512, 32, 64, 3))]
lib = l.lib()
dev = "cuda"
wt = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
nel = 9 * ci * co
wf, wd = torch.zeros(3 * nel, device=dev), torch.zeros(3 * nel, device=dev)
l.check(lib.ustrun_pack_conv3x3(wt.data_ptr(), co, ci, wf.data_ptr(), wd.data_ptr(), 3, None))
x = torch.randn(n, hw, hw, ci, device=dev)
sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
src = l.nhwc_src(x.data_ptr(), ci, hw, hw, scale=sc.data_ptr(), shift=sh.data_ptr(), relu=1)
y = torch.empty(n, hw, hw, co, device=dev)
dy = torch.randn(n, hw, hw, co, device=dev)
da = torch.empty(n, hw, hw, ci, device=dev)
mt = lib.ustrun_conv_mtiles(n, hw, hw, co)
stat = torch.zeros(mt, 2, co, device=dev)
nb = lib.ustrun_wgrad_partials_bytes(9, ci, co, n * hw * hw)
part = torch.empty(nb // 4, device=dev)
dw = torch.empty(co, ci, 3, 3, device=dev)
fl = 2.0 * n * hw * hw * 9 * ci * co


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tf = timed(lambda: l.check(lib.ustrun_conv3x3_fwd(C.byref(src), 1, wf.data_ptr(), n, hw, hw, co, y.data_ptr(), stat.data_ptr(), 3, None)))
td = timed(lambda: l.check(lib.ustrun_conv3x3_dgrad(dy.data_ptr(), wd.data_ptr(), n, hw, hw, co, ci, da.data_ptr(), ci, None, 0, 0, 0, 0, 3, None)))
tw = timed(lambda: l.check(lib.ustrun_conv3x3_wgrad(C.byref(src), 1, dy.data_ptr(), n, hw, hw, co, dw.data_ptr(), 0, part.data_ptr(), nb, 3, None)))
print(f"{ci}->{co} @{hw} n={n}: fwd {tf:.3f} ms ({fl / tf / 1e9:.0f} TF/s-eq)  dgrad {td:.3f} ({fl / td / 1e9:.0f})  wgrad {tw:.3f} ({fl / tw / 1e9:.0f})")
