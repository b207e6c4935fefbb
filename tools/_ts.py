import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd"))
from ustrun import _lib as l
import bench_layers as B
lib = l.lib()
h = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd", "ustrun", "libustrun.so"))
def run(ci, co, hw, n, which):
    dev, bf = "cuda", torch.bfloat16
    wt = torch.randn(co, ci, 3, 3, device=dev) / (3 * ci ** 0.5)
    nel = 9 * ci * co
    wf, wd = torch.zeros(nel, dtype=bf, device=dev), torch.zeros(nel, dtype=bf, device=dev)
    l.check(lib.ustrun_pack_conv3x3(wt.data_ptr(), co, ci, wf.data_ptr(), wd.data_ptr(), 1, None))
    sc, sh = torch.rand(ci, device=dev) + 0.5, torch.randn(ci, device=dev) * 0.1
    a0 = torch.randn(n, hw, hw, ci, device=dev).to(bf)
    srcs = (l.Src * 1)()
    srcs[0] = l.nhwc_src(a0.data_ptr(), ci, hw, hw, scale=sc.data_ptr(), shift=sh.data_ptr(), relu=1)
    y = torch.empty(n, hw, hw, co, device=dev, dtype=bf)
    dy = torch.randn(n, hw, hw, co, device=dev).to(bf)
    da = torch.empty(n, hw, hw, ci, device=dev, dtype=bf)
    stat = torch.zeros(lib.ustrun_conv_mtiles(n, hw, hw, co), 2, co, device=dev)
    for _ in range(3):
        if which == "fwd":
            l.check(lib.ustrun_conv3x3_fwd(srcs, 1, wf.data_ptr(), n, hw, hw, co, y.data_ptr(), stat.data_ptr(), 1, None))
        else:
            l.check(lib.ustrun_conv3x3_dgrad(dy.data_ptr(), wd.data_ptr(), n, hw, hw, co, ci, da.data_ptr(), ci, None, 0, 0, 0, 0, 1, None))
        torch.cuda.synchronize()
    ts = np.zeros((32768, 8), dtype=np.uint64)
    h.ustrun_debug_halo_ts.argtypes = [C.c_void_p, C.c_long]
    assert h.ustrun_debug_halo_ts(ts.ctypes.data, ts.nbytes) == 0
    nb = n * (hw // 8) * (hw // 32 if hw >= 32 else 1) * max(1, (ci if which == "dgrad" else co) // 128)
    nb = min(nb, 32768)
    t = ts[:nb].astype(np.float64)
    ns = t[:, 6].mean()
    print(f"{which} {ci}->{co} @{hw} n={n}: blocks {nb}, stages/tile {ns:.0f}; cycles per stage (wave 1):")
    names = ["top (DMA issue)", "reads+MFMA issue", "vmcnt wait", "transform+lgkm", "barrier"]
    for q in range(5):
        print("   %-18s %7.0f" % (names[q], (t[:, q] / t[:, 6]).mean()))
    print("   %-18s %7.0f   (MFMA pipe time of one wave's stage: 512)" % ("loop total/stage", (t[:, 5] / t[:, 6]).mean()))
for args in ((64, 64, 256, 64, "fwd"), (64, 64, 256, 64, "dgrad"), (256, 256, 64, 64, "fwd"), (256, 256, 64, 64, "dgrad"), (128, 128, 128, 64, "fwd")):
    run(*args)
