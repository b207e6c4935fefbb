#!/usr/bin/env python3
"""bench.py -- training-step throughput of the UST-RUN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one semi-supervised iteration (reference train.py:578-858 minus data loading and
logging: 3 teacher + 5 student forwards (+1 low-quality-sample forward), 4 backwards, losses,
SGD + EMA) on a synthetic batch resident in HBM.  Metric: train images/sec =
(label_bs + unlabel_bs) * world / step time.  For N > 1 launch with torch.distributed.run; every
rank trains its own shard (weak scaling) and gradients are summed with one RCCL all-reduce.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ust-run_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dataset", default="fundus", choices=["fundus", "prostate", "BUSI", "MNMS"])
    ap.add_argument("--label_bs", type=int, default=16)
    ap.add_argument("--unlabel_bs", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16"],
                    help="bf16 = bf16 matrix-core operands, f32 accumulate/statistics (configs[1]); f32 = the exact parity path")
    ap.add_argument("--fft", default="device", choices=["host", "device"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP-event roofline leg")
    return ap.parse_args()


def cpu_baseline(dataset):
    """The oracle's step (CPU restatement of the reference) timed on this host: config[0]
    (B = 4+4, fp32) -- a reported baseline, not the target."""
    from oracle import unet_ref as U
    from oracle.step_ref import DATASETS, RefTrainer
    from ustrun import synthetic
    C, H, K = DATASETS[dataset][:3]
    cores = min(len(os.sched_getaffinity(0)), 16)     # the GPU box gives one GPU a 16-CPU share
    torch.set_num_threads(cores)
    torch.manual_seed(1337)
    sd = U.make_state_dict(C, K)
    tr = RefTrainer(dataset, sd)
    random.seed(1212); np.random.seed(1337)
    b = synthetic.batch(dataset, 4, C, H, 1337)
    print(f"[bench] cpu_baseline: timing one oracle step on {cores} threads ...", file=sys.stderr, flush=True)
    t0 = time.time()
    tr.step(*b, epoch_start=True)
    dt = time.time() - t0
    return {"value": round(8 / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"1 step of config[0]: {dataset} {H}x{H}, label_bs=unlabel_bs=4, fp32, oracle/step_ref.py on torch-CPU ({dt:.1f} s)"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        from ustrun import ddp
        ddp.init("nccl", device=dev)              # "nccl" is RCCL on ROCm

    from networks.unet_model import UNet
    from ustrun import _lib, synthetic
    from ustrun.trainer import DATASETS, SSLTrainer
    C, H, K = DATASETS[a.dataset][:3]
    torch.manual_seed(1337)                       # identical initial weights on every rank
    model = UNet(C, K, dtype=a.dtype).to(dev)
    ema = UNet(C, K, dtype=a.dtype).to(dev)

    # one SUM all-reduce of the flat 124 MB gradient buffer per step; 1/world is folded into the SGD kernel
    from ustrun.ddp import make_grad_allreduce
    tr = SSLTrainer(a.dataset, model, ema, grad_allreduce=make_grad_allreduce(world), world_size=world, fft=a.fft)
    random.seed(1212 + rank); np.random.seed(1337 + rank)
    nb = 4                                        # a few distinct resident batches, cycled
    batches = []
    for i in range(nb):
        lb = synthetic.batch(a.dataset, a.label_bs, C, H, 1337 + 1000 * rank + i)
        batches.append([t.to(dev) for t in lb])

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for s in range(a.warmup):
        tr.step(*batches[s % nb], epoch_start=(s == 0))
    lib = _lib.lib()
    sync()
    prof = not a.no_profile
    # the HIP-event pairs around every conv launch cost ~5 % of a step, so they sample the LAST min(2, K) steps of the
    # timed region rather than all of it (the whole region is still what `value` is computed from)
    nprof = min(2, a.steps) if prof else 0
    t0 = time.perf_counter()
    for s in range(a.steps):
        if s == a.steps - nprof:
            lib.ustrun_profile_enable(1)
        tr.step(*batches[(a.warmup + s) % nb])
    sync()
    dt = time.perf_counter() - t0
    lib.ustrun_profile_enable(0)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    roof = None
    if prof:
        ms, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        lib.ustrun_profile_collect(0, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(n))
        peak = 157.3 if a.dtype == "f32" else 2500.0
        ach = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        roof = {"kernel": "DoubleConv convolutions: conv3x3 forward + input-gradient (halo-tiled implicit GEMM) and ConvTranspose",
                "bound": "mfma", "achieved": round(ach, 2),
                "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                "launches_per_step": n.value // max(nprof, 1), "sampled_steps": nprof,
                "avg_launch_ms": round(ms.value / max(n.value, 1), 4),
                "alg_flops_per_launch": fl.value / max(n.value, 1), "alg_bytes_per_launch": by.value / max(n.value, 1),
                "alg_gbps": round(by.value / (ms.value * 1e-3) / 1e9, 1) if ms.value > 0 else 0.0,
                "time_share_of_step": round(ms.value * 1e-3 / (dt * nprof / a.steps), 3)}
        ms2, fl2, n2 = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        lib.ustrun_profile_collect(1, ctypes.byref(ms2), ctypes.byref(fl2), None, ctypes.byref(n2))
        if ms2.value > 0:
            roof["wgrad"] = {"achieved": round(fl2.value / (ms2.value * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                             "frac": round(fl2.value / (ms2.value * 1e-3) / 1e12 / peak, 4),
                             "time_share_of_step": round(ms2.value * 1e-3 / (dt * nprof / a.steps), 3)}
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), measured offline with
        # rocprofv3 --pmc on this same command and committed under profiles/ (bench.py cannot run under two profilers)
        tpath = os.path.join(ROOT, "profiles", f"traffic_{a.dtype}.json")
        if os.path.exists(tpath):
            try:
                roof["traffic"] = json.load(open(tpath))["hbm_bytes_per_launch"]
            except Exception:
                pass
    if rank == 0:
        imgs = (a.label_bs + a.unlabel_bs) * world * a.steps
        out = {"metric": "train images/sec (256x256 U-Net, mixed lb+ulb batch)", "value": round(imgs / dt, 3),
               "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"{a.dataset} {H}x{H}, {K}-class U-Net (31.0M params), batch={a.label_bs}+{a.unlabel_bs} per GPU, "
                                      f"SSL step = 3 teacher + 5(+1) student forwards, 4 backwards, CE+Dice, SGD+EMA",
                          "global_batch": (a.label_bs + a.unlabel_bs) * world, "parallelism": f"dp{world}", "fft_mix": a.fft},
               "roofline": roof, "cpu_baseline": None}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.dataset)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
