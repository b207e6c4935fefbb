#!/usr/bin/env python3
"""GPU box: run the trajectory experiment of tools/traj_common.py with the HIP trainers (f32 and bf16) on several
synthetic task difficulties and print, per task, the validation Dice after 200 steps and the f32-vs-bf16 spread --
used once to pick a task that does not saturate before the CPU oracle spends an hour on it.

    python tools/calib_task.py [--tasks default,hard,...] [--steps 200]
"""
import argparse
import json
import random
import sys
import time

import numpy as np
import torch

import traj_common as T


def run(dtype, task, steps, C, H, K, perturb=0.0, kind="unet", base_lr=0.03):
    from networks.unet_model import UNet
    from oracle import unet_ref as U
    from ustrun import evaluate
    from ustrun.trainer import SSLTrainer
    torch.manual_seed(T.MODEL_SEED)
    if kind == "unet":
        sd_s, sd_t = U.make_state_dict(C, K), U.make_state_dict(C, K)
        model, ema = UNet(C, K, dtype=dtype), UNet(C, K, dtype=dtype)
    else:                               # deeplabv2-<arch>: BASELINE.json configs[4]'s model under the same experiment
        from networks.deeplabv2 import DeepLabV2
        from oracle import deeplab_ref as D
        arch = kind.split("-")[1]
        sd_s, sd_t = D.make_state_dict(arch, K, T.MODEL_SEED), D.make_state_dict(arch, K, T.MODEL_SEED + 1)
        model, ema = DeepLabV2(arch, K, pretrained=False, dtype=dtype), DeepLabV2(arch, K, pretrained=False, dtype=dtype)
    if perturb:
        for k in U.param_keys(sd_s):
            sd_s[k] = sd_s[k] * (1 + perturb)
    model.load_state_dict({k: v.clone() for k, v in sd_s.items()})
    ema.load_state_dict({k: v.clone() for k, v in sd_t.items()})
    tr = SSLTrainer(T.DATASET, model.cuda(), ema.cuda(), fft="device", max_iterations=T.MAX_ITER, num_eval_iter=T.NUM_EVAL_ITER,
                    patch_size=H, base_lr=base_lr)
    random.seed(T.PY_SEED); np.random.seed(T.NP_SEED)
    hist = []
    for s in range(steps):
        b = T.batch(s, task, C, H)
        tr.step(*[t.cuda() for t in b], epoch_start=(s % T.NUM_EVAL_ITER == 0))
        if s % T.LOG_EVERY == T.LOG_EVERY - 1:
            sc = tr.scalars()
            hist.append((s + 1, sc["loss"], float(np.mean(sc["ulb_dice"])), sc["mask_ratio"]))
    loaders = [[(x.cuda(), y.cuda()) for x, y in dom] for dom in T.val_loaders(task, C, H)]
    vs, _ = evaluate.validate(T.DATASET, model, loaders, log=None)
    vt, _ = evaluate.validate(T.DATASET, ema, loaders, log=None)
    return hist, vs, vt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", default="default,hard")
    ap.add_argument("--steps", type=int, default=T.STEPS)
    ap.add_argument("--model", default="unet", help="unet | deeplabv2-resnet50 | deeplabv2-resnet101")
    ap.add_argument("--size", type=int, default=0, help="patch extent (0: the dataset's)")
    ap.add_argument("--base_lr", type=float, default=0.03)
    a = ap.parse_args()
    from ustrun import synthetic
    from ustrun.trainer import DATASETS
    C, H, K = DATASETS[T.DATASET][:3]
    H = a.size or H
    for name in a.tasks.split(","):
        if name in synthetic.TASKS:
            task = synthetic.TASKS[name]
        else:                      # contrast:noise:rmin:rspan
            c, n, r0, rs = (float(v) for v in name.split(":"))
            task = dict(contrast=c, noise=n, rmin=r0, rspan=rs)
        t0 = time.time()
        res = {}
        for tag, dt, pert in (("f32", "f32", 0.0), ("f32p", "f32", 1e-6), ("bf16", "bf16", 0.0)):
            hist, vs, vt = run(dt, task, a.steps, C, H, K, pert, a.model, a.base_lr)
            res[tag] = {"val_student": vs, "val_teacher": vt, "loss_end": hist[-1][1], "pl_dice_tail": [h[2] for h in hist[-3:]],
                        "loss_curve": [round(h[1], 4) for h in hist]}
        print("CALIB " + json.dumps({"model": a.model, "size": H, "base_lr": a.base_lr, "task": name, "cfg": task, "secs": round(time.time() - t0, 1), **res}), flush=True)


if __name__ == "__main__":
    main()
