"""Where does the host spend its time inside SSLTrainer.step?  Prints the mean host-side time between the
trainer's timeline marks (no device synchronisation added) and the synchronised wall time per step, so that
launch-bound or host-blocked sections show up next to the GPU-bound ones.  Development tool.

    python tools/host_timeline.py [--steps 10] [--dtype bf16]
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "ust-run_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--dataset", default="fundus")
    ap.add_argument("--bs", type=int, default=16)
    a = ap.parse_args()
    from networks.unet_model import UNet
    from ustrun import synthetic
    from ustrun.trainer import DATASETS, SSLTrainer
    dev = torch.device("cuda:0")
    C, H, K = DATASETS[a.dataset][:3]
    torch.manual_seed(1337)
    model, ema = UNet(C, K, dtype=a.dtype).to(dev), UNet(C, K, dtype=a.dtype).to(dev)
    tr = SSLTrainer(a.dataset, model, ema, fft="device")
    random.seed(1212); np.random.seed(1337)
    batches = [[t.to(dev) for t in synthetic.batch(a.dataset, a.bs, C, H, 1337 + i)] for i in range(4)]
    for s in range(3):
        tr.step(*batches[s % 4], epoch_start=(s == 0))
    torch.cuda.synchronize()
    acc, order = {}, []
    t0 = time.perf_counter()
    for s in range(a.steps):
        tr.timeline = []
        tr.step(*batches[(3 + s) % 4])
        tl = tr.timeline
        for (l0, a0), (l1, a1) in zip(tl[:-1], tl[1:]):
            if l1 not in acc:
                acc[l1] = 0.0
                order.append(l1)
            acc[l1] += a1 - a0
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    for l in order:
        print(f"{l:28s} {acc[l] / a.steps * 1e3:8.3f} ms")
    print(f"{'host issue total / step':28s} {t_issue / a.steps * 1e3:8.3f} ms")
    print(f"{'synchronised wall / step':28s} {t_wall / a.steps * 1e3:8.3f} ms")


if __name__ == "__main__":
    main()
