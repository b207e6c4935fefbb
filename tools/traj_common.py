"""Shared definition of the 200-step trajectory gate (north_star: "Dice within 1e-3 of reference after 200 fixed-seed
steps"; loop: reference train.py:577-858): which batches, which seeds, which schedule, which validation set.  Used by
tools/gen_traj_golden.py (CPU oracle -> tests/golden/g9_traj_*.npz), tools/calib_task.py and
tests/test_gpu_trajectory.py (HIP trainers), so that all three run the SAME experiment."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ust-run_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

DATASET = "fundus"
STEPS = 200
BS = 4                     # label_bs = unlabel_bs = 4: the reference's own batch (train.py:408-409)
MAX_ITER = 2000            # shortened schedule: the consistency weight ramps and the LR decays within the run
NUM_EVAL_ITER = 50         # "epoch" length: hardness ranking and the low-quality-sample forward switch on at step 50
MODEL_SEED = 1337
PY_SEED, NP_SEED = 1212, 1337
BATCH_SEED0 = 5000         # batch of step s: synthetic.batch(..., seed = BATCH_SEED0 + s, task)
VAL_SEED, VAL_BATCHES, VAL_BS = 9000, 6, 4   # fixed validation set: 24 images, one "domain"
LOG_EVERY = 10


def batch(s, task, C, H):
    from ustrun import synthetic
    return synthetic.batch(DATASET, BS, C, H, BATCH_SEED0 + s, task)


def val_loaders(task, C, H):
    from ustrun import synthetic
    return synthetic.test_loaders(DATASET, 1, VAL_BATCHES, VAL_BS, C, H, VAL_SEED, task)
