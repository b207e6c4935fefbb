#!/usr/bin/env python3
"""Training-trajectory parity harness (GPU box): N fixed-seed steps of the HIP trainer in bf16 against
the HIP trainer in f32 (the parity-pinned exact path), same synthetic batches, same RNG streams.
Reports the pseudo-label Dice of the last steps and the loss curves; the bar of BASELINE.json's
north_star is |Dice_bf16 - Dice_f32| <= 1e-3 after 200 steps.

    python tools/parity_200.py [--steps 200] [--bs 4] [--dataset fundus] [--oracle-steps 0]
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ust-run_amd")]
import numpy as np
import torch


def run(dtype, a, C, H, K):
    from networks.unet_model import UNet
    from ustrun import synthetic
    from ustrun.trainer import SSLTrainer
    torch.manual_seed(1337)
    model, ema = UNet(C, K, dtype=dtype).cuda(), UNet(C, K, dtype=dtype).cuda()
    tr = SSLTrainer(a.dataset, model, ema, fft="device", max_iterations=a.max_iter)
    random.seed(1212); np.random.seed(1337)
    hist = []
    t0 = time.time()
    for s in range(a.steps):
        b = synthetic.batch(a.dataset, a.bs, C, H, 5000 + s)
        tr.step(*[t.cuda() for t in b], epoch_start=(s == 0))
        if s % 10 == 9 or s == a.steps - 1:
            sc = tr.scalars()
            hist.append((s + 1, sc["loss"], sc["ulb_dice"], sc["mask_ratio"]))
            print(f"[{dtype}] step {s + 1}: loss {sc['loss']:.4f} ulb_dice {['%.4f' % v for v in sc['ulb_dice']]} mask {sc['mask_ratio']:.3f}"
                  f"  ({time.time() - t0:.0f}s)", flush=True)
    return hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--dataset", default="fundus")
    ap.add_argument("--max_iter", type=int, default=2000, help="shortened schedule so that the consistency weight ramps within the run")
    a = ap.parse_args()
    from ustrun.trainer import DATASETS
    C, H, K = DATASETS[a.dataset][:3]
    h32 = run("f32", a, C, H, K)
    h16 = run("bf16", a, C, H, K)
    tail = lambda h: np.mean([np.mean(d) for _, _, d, _ in h[-3:]])
    res = {"dataset": a.dataset, "steps": a.steps, "bs": a.bs, "dice_f32": float(tail(h32)), "dice_bf16": float(tail(h16)),
           "abs_diff": float(abs(tail(h32) - tail(h16))), "loss_f32": h32[-1][1], "loss_bf16": h16[-1][1]}
    print("PARITY " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
