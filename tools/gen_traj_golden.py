#!/usr/bin/env python3
"""Build container only (about an hour on 8 cores, once): the CPU oracle's 200-step training trajectory ->
tests/golden/g9_traj_fundus_200.npz.

Runs oracle/step_ref.py::RefTrainer -- the restatement of the reference loop train.py:577-858, every op of which is
pinned to the reference by G1-G8 -- on the experiment tools/traj_common.py defines (fundus 256^2, B = 4+4, f32, fixed
seeds, the non-saturating "medium" synthetic task), and stores what tests/test_gpu_trajectory.py compares the HIP trainers
against: the loss terms and pseudo-label Dice every 10 steps, the validation Dice (oracle/eval_ref.py, the reference's
test() of train.py:253-395) of student and EMA teacher on a fixed 24-image set at steps 100 and 200, and the final
per-parameter norms.  The fixture is data (scalars and norms); no reference source is stored.

    python3 tools/gen_traj_golden.py [--task hard] [--steps 200] [--threads 6]
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

import traj_common as T


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="medium")
    ap.add_argument("--steps", type=int, default=T.STEPS)
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(T.ROOT, "tests", "golden", "g9_traj_fundus_200.npz"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from oracle import eval_ref as E
    from oracle import unet_ref as U
    from oracle.step_ref import DATASETS, RefTrainer
    from ustrun import synthetic
    C, H, K = DATASETS[T.DATASET][:3]
    task = synthetic.TASKS[a.task]
    torch.manual_seed(T.MODEL_SEED)
    sd_s, sd_t = U.make_state_dict(C, K), U.make_state_dict(C, K)
    tr = RefTrainer(T.DATASET, sd_s, max_iterations=T.MAX_ITER, num_eval_iter=T.NUM_EVAL_ITER)
    tr.set_teacher(sd_t)
    random.seed(T.PY_SEED); np.random.seed(T.NP_SEED)
    loaders = T.val_loaders(task, C, H)
    keys = ("loss", "sup", "ul", "lu", "s", "w", "mask_ratio")
    log = {k: [] for k in keys}
    log["step"], log["ulb_dice"] = [], []
    val = {}
    t0 = time.time()
    for s in range(a.steps):
        out = tr.step(*T.batch(s, task, C, H), epoch_start=(s % T.NUM_EVAL_ITER == 0))
        if s % T.LOG_EVERY == T.LOG_EVERY - 1:
            log["step"].append(s + 1)
            for k in keys:
                log[k].append(out[k])
            log["ulb_dice"].append(out["ulb_dice"])
            print(f"step {s + 1}: loss {out['loss']:.4f} sup {out['sup']:.4f} ulb_dice {out['ulb_dice']} mask {out['mask_ratio']:.3f} "
                  f"({time.time() - t0:.0f} s)", flush=True)
        if (s + 1) % 100 == 0:
            vs, _ = E.validate(T.DATASET, tr.student, loaders)
            vt, _ = E.validate(T.DATASET, tr.teacher, loaders)
            val[s + 1] = (vs, vt)
            print(f"step {s + 1}: val student {vs} teacher {vt}", flush=True)
    rec = {k: np.array(v, dtype=np.float64) for k, v in log.items()}
    for st, (vs, vt) in val.items():
        rec[f"val_student_{st}"], rec[f"val_teacher_{st}"] = np.array(vs), np.array(vt)
    pk = U.param_keys(tr.student)
    rec["student_norms"] = np.array([float(tr.student[k].detach().double().norm()) for k in pk])
    rec["teacher_norms"] = np.array([float(tr.teacher[k].detach().double().norm()) for k in pk])
    rec["student_sums"] = np.array([float(tr.student[k].detach().double().sum()) for k in pk])
    rec["task"] = np.array(a.task)
    rec["task_cfg"] = np.array([task["contrast"], task["noise"], task["rmin"], task["rspan"]])
    rec["config"] = np.array([a.steps, T.BS, T.MAX_ITER, T.NUM_EVAL_ITER, T.MODEL_SEED, T.PY_SEED, T.NP_SEED, T.BATCH_SEED0, T.VAL_SEED,
                              T.VAL_BATCHES, T.VAL_BS])
    rec["_torch_version"] = np.array(torch.__version__)
    np.savez_compressed(a.out, **rec)
    print(f"wrote {a.out} ({os.path.getsize(a.out) / 1024:.1f} KiB) in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
