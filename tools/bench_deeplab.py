#!/usr/bin/env python3
"""GPU box: forward throughput of DeepLabV2-ResNet101 at BASELINE.json configs[4]'s shape (BUSI 512x512, 2 classes) through
libustrun.so -- images/s, achieved TFLOP/s of the convolution launches (HIP-event pairs, algorithmic flops) and the share of
time by kernel class.  Forward only (what this round builds of SURVEY.md 8f row 4).

    python tools/bench_deeplab.py [--n 16] [--hw 512] [--dtype bf16] [--mode train|eval] [--reps 5] [--arch resnet101]
"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ust-run_amd")]
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16)
    ap.add_argument("--hw", type=int, default=512)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--mode", default="train")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--arch", default="resnet101")
    a = ap.parse_args()
    from networks.deeplabv2 import DeepLabV2
    from ustrun import _lib
    lib = _lib.lib()
    torch.manual_seed(0)
    m = DeepLabV2(a.arch, 2, pretrained=False, dtype=a.dtype).cuda()
    m.train(a.mode == "train")
    x = torch.randn(a.n, 3, a.hw, a.hw, device="cuda")
    with torch.no_grad():
        m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            m(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
        lib.ustrun_profile_enable(1)
        m(x)
        torch.cuda.synchronize()
        lib.ustrun_profile_enable(0)
    buf = (_lib.ProfRec * 4096)()
    n = lib.ustrun_profile_records(buf, 4096)
    ms = sum(r.ms for r in buf[:n])
    fl = sum(r.flops for r in buf[:n])
    by = sum(r.bytes for r in buf[:n])
    if os.environ.get("USTRUN_BENCH_VERBOSE"):
        rows = sorted(((r.ms, r.flops, r.bytes) for r in buf[:n]), reverse=True)
        for ms_, fl_, by_ in rows[:24]:
            print(f"   {ms_:8.3f} ms  {fl_ / 1e9:9.1f} GF  {fl_ / ms_ / 1e9:7.0f} TF/s  {by_ / 1e6:8.1f} MB  {by_ / ms_ / 1e6:7.0f} GB/s  AI {fl_ / by_:6.0f}")
    msd, fld, byd, nd = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    lib.ustrun_profile_collect(0, ctypes.byref(msd), ctypes.byref(fld), ctypes.byref(byd), ctypes.byref(nd))
    print(f"DeepLabV2-{a.arch} {a.mode}-mode forward, N={a.n} {a.hw}x{a.hw}, {a.dtype}: {dt * 1e3:.2f} ms = {a.n / dt:.1f} images/s; "
          f"{n} convolution launches: {ms:.2f} ms, {fl / 1e9 / a.n:.1f} GFLOP/image algorithmic, {fl / ms / 1e9:.0f} TFLOP/s, "
          f"{by / ms / 1e6:.0f} GB/s algorithmic; convolutions are {ms / (dt * 1e3) * 100:.0f} % of the forward")


if __name__ == "__main__":
    main()
