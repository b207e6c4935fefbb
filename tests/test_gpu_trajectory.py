"""The north_star's trajectory gate, against the CPU path: "Dice within 1e-3 of reference after 200 fixed-seed steps"
(loop: reference train.py:577-858).

tests/golden/g9_traj_fundus_200.npz holds the CPU oracle's run (tools/gen_traj_golden.py: oracle/step_ref.py, fundus
256^2, B = 4+4, f32, the non-saturating "medium" synthetic task of ustrun/synthetic.py); here the HIP trainers run the
same 200 steps on the same seeds -- f32 (the exact path) and bf16 (the production kernels, BASELINE.json configs[1]'s
dtype) -- and must land on the oracle's validation Dice (fixed 24-image set, eval mode) and follow its loss curve.

Trajectories that differ only by rounding separate chaotically (ReLU kinks, the 0.95 pseudo-label threshold), so every
bound is set from the measured spread of two f32 HIP runs whose initial weights differ by 1e-6 (tools/calib_task.py,
profiles/r02_calib_task.log), on the hardest task where a 1e-3 gate is still readable: EMA-teacher validation Dice
(the model the reference validates first and reports, train.py:913-935) twins 1.5e-4 apart -> gate 1e-3 (north_star);
student validation Dice twins 2.1e-3 apart (cup 4e-3) -> 1e-2; loss at the logged steps twins <= 1.1e-3 apart -> 1e-2
absolute after step 50, 2 % relative before."""
import os
import random
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden

sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

DICE_TOL = 1e-3            # north_star, on the EMA teacher
STUDENT_TOL = 1e-2         # the student's own Dice moves by 4e-3 between rounding-equivalent runs
LOSS_RTOL_EARLY = 2e-2     # steps <= 50: trajectories still rounding-close (f32, f16)
LOSS_RTOL_EARLY_BF16 = 2e-2   # bf16: the same bar again.  Round 4 had widened it to 3e-2 after one build landed 2.1 % from the oracle at
                              # step 30; with this round's step (profiles/r05_traj_loss_deviation.log) bf16 sits within 0.3 % at every
                              # logged step <= 50 (f32x3: 0.2 %), so the original bar holds with a wide margin
LOSS_ATOL_LATE = 1e-2      # later: same basin, different rounding path


_BATCHES = {}      # the 200 seeded synthetic batches (0.2 s of host generation each) are the same for every dtype: made once per session


def _batch(T, s, task, C, H):
    key = (s, tuple(sorted(task.items())), C, H)
    if key not in _BATCHES:
        _BATCHES[key] = T.batch(s, task, C, H)
    return _BATCHES[key]


def _run(dtype, g):
    import traj_common as T
    from networks.unet_model import UNet
    from oracle import unet_ref as U
    from ustrun import evaluate, synthetic
    from ustrun.trainer import DATASETS, SSLTrainer
    C, H, K = DATASETS[T.DATASET][:3]
    cfg = g["task_cfg"]
    task = dict(contrast=float(cfg[0]), noise=float(cfg[1]), rmin=float(cfg[2]), rspan=float(cfg[3]))
    assert [int(v) for v in g["config"]] == [T.STEPS, T.BS, T.MAX_ITER, T.NUM_EVAL_ITER, T.MODEL_SEED, T.PY_SEED, T.NP_SEED,
                                             T.BATCH_SEED0, T.VAL_SEED, T.VAL_BATCHES, T.VAL_BS], "fixture made for another experiment"
    torch.manual_seed(T.MODEL_SEED)
    sd_s, sd_t = U.make_state_dict(C, K), U.make_state_dict(C, K)
    model, ema = UNet(C, K, dtype=dtype), UNet(C, K, dtype=dtype)
    model.load_state_dict({k: v.clone() for k, v in sd_s.items()})
    ema.load_state_dict({k: v.clone() for k, v in sd_t.items()})
    tr = SSLTrainer(T.DATASET, model.cuda(), ema.cuda(), fft="host" if dtype in ("f32", "f32x3") else "device", max_iterations=T.MAX_ITER,
                    num_eval_iter=T.NUM_EVAL_ITER)
    random.seed(T.PY_SEED); np.random.seed(T.NP_SEED)
    loaders = [[(x.cuda(), y.cuda()) for x, y in dom] for dom in T.val_loaders(task, C, H)]
    loss, dice, val = [], [], {}
    for s in range(T.STEPS):
        tr.step(*[t.cuda() for t in _batch(T, s, task, C, H)], epoch_start=(s % T.NUM_EVAL_ITER == 0))
        if s % T.LOG_EVERY == T.LOG_EVERY - 1:
            sc = tr.scalars()
            loss.append(sc["loss"]); dice.append(sc["ulb_dice"])
        if (s + 1) % 100 == 0:
            val[s + 1] = (evaluate.validate(T.DATASET, model, loaders, log=None)[0], evaluate.validate(T.DATASET, ema, loaders, log=None)[0])
    norms = np.array([float(p.detach().double().norm()) for p in model.parameters()])
    if tr.scaler is not None:
        skipped, seen = tr.scaler.skipped_steps()
        print(f"[{dtype}] loss scale {tr.scaler.get_scale():.0f}, {skipped} of {seen} steps skipped")
        assert seen == T.STEPS and skipped == 0
    return np.array(loss), np.array(dice), val, norms


@pytest.mark.parametrize("dtype", ["f32", "f32x3", "bf16", "f16"])
def test_200_step_trajectory_lands_on_the_oracle(dtype):
    """(f16 = `--amp 1` of the reference: IEEE-half kernels + the device-side GradScaler; a skipped step would be a lost update
    against the f32 oracle, so the run also asserts that the default scale 65536 never overflowed here)"""
    g = load_golden("g9_traj_fundus_200")
    loss, dice, val, norms = _run(dtype, g)
    steps = g["step"].astype(int)
    print(f"[{dtype}] loss  hip {np.round(loss, 4).tolist()}\n[{dtype}] loss  ref {np.round(g['loss'], 4).tolist()}")
    for st in (100, 200):
        vs, vt = val[st]
        ds = float(np.mean(vs)) - float(np.mean(g[f"val_student_{st}"]))
        dt_ = float(np.mean(vt)) - float(np.mean(g[f"val_teacher_{st}"]))
        print(f"[{dtype}] step {st}: val Dice student {np.mean(vs):.5f} (oracle {np.mean(g[f'val_student_{st}']):.5f}, d {ds:+.1e}) "
              f"teacher {np.mean(vt):.5f} (oracle {np.mean(g[f'val_teacher_{st}']):.5f}, d {dt_:+.1e})")
    # the gate: mean validation Dice of the EMA teacher (the model the reference reports and keeps, train.py:913-935) and
    # of the student after 200 steps
    ref_t, ref_s = float(np.mean(g["val_teacher_200"])), float(np.mean(g["val_student_200"]))
    assert 0.3 < ref_t < 0.99, "the task must not saturate"
    assert abs(float(np.mean(val[200][1])) - ref_t) <= DICE_TOL
    assert abs(float(np.mean(val[200][0])) - ref_s) <= STUDENT_TOL
    early = steps <= 50
    dev_early = np.abs(loss[early] - g["loss"][early]) / np.abs(g["loss"][early])
    print(f"[{dtype}] early-loss relative deviation per logged step {np.round(dev_early, 4).tolist()}, mean {dev_early.mean():.4f}")
    np.testing.assert_allclose(loss[early], g["loss"][early], rtol=LOSS_RTOL_EARLY_BF16 if dtype == "bf16" else LOSS_RTOL_EARLY)
    assert float(dev_early.mean()) <= LOSS_RTOL_EARLY
    assert float(np.abs(loss - g["loss"]).max()) <= LOSS_ATOL_LATE
    for st in (100,):                                       # half way: rounding-equivalent runs are still 1e-4 apart
        assert abs(float(np.mean(val[st][1])) - float(np.mean(g[f"val_teacher_{st}"]))) <= DICE_TOL
        assert abs(float(np.mean(val[st][0])) - float(np.mean(g[f"val_student_{st}"]))) <= DICE_TOL
    # final parameters, tensor by tensor (BatchNorm biases have norms of 0.02: absolute floor)
    np.testing.assert_allclose(norms, g["student_norms"], rtol=2e-2, atol=2e-3)
