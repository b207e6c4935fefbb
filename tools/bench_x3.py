"""Forward + backward of the U-Net at configs[1]'s shape under the exact-f32 dtype and under f32x3 (three-term bf16 products,
csrc/x3.hip), same weights and input: time per pass and how far the two are apart (logits, gradients).  Development tool.

    python tools/bench_x3.py [--n 16] [--hw 256] [--reps 3]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "ust-run_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--hw", type=int, default=256)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import copy
    from networks.unet_model import UNet
    torch.manual_seed(1337)
    m32 = UNet(3, 2, dtype="f32").cuda().train()
    mx3 = copy.deepcopy(m32)
    mx3.compute_dtype = "f32x3"
    x = (torch.randint(0, 256, (a.n, 3, a.hw, a.hw)).float() / 127.5 - 1).cuda()
    res = {}
    for name, m in (("f32", m32), ("f32x3", mx3)):
        for it in range(a.reps + 1):
            for p in m.parameters():
                p.grad = None
            if it == 1:
                torch.cuda.synchronize()
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                tf = tb = 0.0
            if it >= 1:
                e0.record()
            lg = m(x)
            if it >= 1:
                e1.record()
            lg.square().mean().backward()
            if it >= 1:
                e2.record()
                torch.cuda.synchronize()
                tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
        res[name] = (lg.detach(), [p.grad.clone() for p in m.parameters()], tf / a.reps, tb / a.reps)
        print(f"{name:6s} forward {tf / a.reps:8.2f} ms   backward {tb / a.reps:8.2f} ms   ({a.n} images of {a.hw}^2)")
    l32, g32, _, _ = res["f32"]
    lx3, gx3, _, _ = res["f32x3"]
    rel = lambda u, v: float((u.double() - v.double()).norm() / (v.double().norm() + 1e-30))
    print(f"logits f32x3 vs f32: rel-L2 {rel(lx3, l32):.3e}, arg-max flips {int((lx3.argmax(1) != l32.argmax(1)).sum())} of {l32[:, 0].numel()}")
    errs = [rel(u, v) for u, v in zip(gx3, g32)]
    print(f"gradients f32x3 vs f32: median rel-L2 {sorted(errs)[len(errs) // 2]:.3e}, worst {max(errs):.3e}")
    print(f"speed-up: forward {res['f32'][2] / res['f32x3'][2]:.2f}x, backward {res['f32'][3] / res['f32x3'][3]:.2f}x")


if __name__ == "__main__":
    main()
