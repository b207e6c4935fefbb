#!/usr/bin/env python3
"""Build container only (~10 min on 8 cores, ~25 GB of host memory, once): two steps of the CPU oracle at BASELINE.json configs[1]'s
REAL shape -> tests/golden/g11_config1_step.npz.

Runs oracle/step_ref.py::RefTrainer (the restatement of the reference loop train.py:577-858, every op of which is pinned to the
reference by G1-G8) on fundus 256^2 with label_bs = unlabel_bs = 16 and the reference's channel plan: step 1 opens an epoch, step 2
carries the low-quality-sample forward (train.py:740).  tests/test_gpu_step.py::test_ssl_step_at_config1_shape_matches_oracle rebuilds
the same initial weights and batches from the same seeds (CONFIG below) and compares the HIP trainer's loss terms, pseudo-label Dice,
BatchNorm running statistics (all of them) and a strided sample of every parameter tensor (<= 1024 values each, plus the tensor's
norm) of student and EMA teacher after the two steps.  On the GPU box the same two oracle steps took 244 s of the test run: hence a
fixture.  The fixture is data (scalars, statistics, samples); no reference source is stored.

    python3 tools/gen_config1_golden.py [--threads 8]
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ust-run_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

# what the test must reproduce: dataset, C, K, H, B, model seed, batch seed of step 0, python / numpy seed, steps
CONFIG = dict(dataset="fundus", C=3, K=2, H=256, B=16, model_seed=11, batch_seed0=900, rng_seed=9, steps=2)
KW = dict(max_iterations=300, threshold=0.52, num_eval_iter=2)
SAMPLE = 1024


def sample(t):
    """<= SAMPLE values of a tensor at a fixed stride over its flattened elements"""
    f = t.detach().reshape(-1)
    return f[::max(1, f.numel() // SAMPLE)][:SAMPLE].clone()


def setup():
    """-> (student state_dict, teacher state_dict, batches): shared by this script and the test"""
    from oracle import unet_ref as U
    from ustrun import synthetic
    c = CONFIG
    torch.manual_seed(c["model_seed"])
    sd_s, sd_t = U.make_state_dict(c["C"], c["K"]), U.make_state_dict(c["C"], c["K"])
    batches = [synthetic.batch(c["dataset"], c["B"], c["C"], c["H"], c["batch_seed0"] + s) for s in range(c["steps"])]
    return sd_s, sd_t, batches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "g11_config1_step.npz"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from oracle.step_ref import RefTrainer
    sd_s, sd_t, batches = setup()
    ref = RefTrainer(CONFIG["dataset"], sd_s, **KW)
    ref.set_teacher(sd_t)
    random.seed(CONFIG["rng_seed"]); np.random.seed(CONFIG["rng_seed"])
    out = {"config": np.array([CONFIG[k] for k in ("C", "K", "H", "B", "model_seed", "batch_seed0", "rng_seed", "steps")]),
           "torch_version": np.array(torch.__version__)}
    for s, b in enumerate(batches):
        t0 = time.time()
        r = ref.step(*b, epoch_start=(s == 0))
        print(f"step {s}: {time.time() - t0:.0f} s  " + ", ".join(f"{k} {r[k]:.6f}" for k in ("sup", "ul", "lu", "s", "loss", "w")), flush=True)
        for k in ("sup", "ul", "lu", "s", "loss", "w"):
            out[f"step{s}.{k}"] = np.float64(r[k])
        out[f"step{s}.ulb_dice"] = np.asarray(r["ulb_dice"], dtype=np.float64)
    out["iter_num"], out["lr"] = np.int64(ref.iter_num), np.float64(ref.lr)
    for name, sd in (("student", ref.student), ("teacher", ref.teacher)):
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                out[f"{name}.{k}"] = np.int64(int(v))
            elif "running_" in k:
                out[f"{name}.{k}"] = v.detach().numpy().astype(np.float32)
            else:
                out[f"{name}.{k}.sample"] = sample(v).numpy().astype(np.float32)
                out[f"{name}.{k}.norm"] = np.float64(float(v.detach().double().norm()))
    np.savez_compressed(a.out, **out)
    print("wrote", a.out, os.path.getsize(a.out), "bytes")


if __name__ == "__main__":
    main()
