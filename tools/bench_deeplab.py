#!/usr/bin/env python3
"""GPU box: forward throughput of DeepLabV2-ResNet101 at BASELINE.json configs[4]'s shape (BUSI 512x512, 2 classes) through
libustrun.so -- images/s, achieved TFLOP/s of the convolution launches (HIP-event pairs, algorithmic flops) and the share of
time by kernel class; --backward times forward + backward (every parameter's gradient, ustrun.resnet_engine.DeepLabFn).

    python tools/bench_deeplab.py [--n 16] [--hw 512] [--dtype bf16] [--mode train|eval] [--reps 5] [--arch resnet101] [--backward]
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
    ap.add_argument("--backward", action="store_true")
    ap.add_argument("--ssl", action="store_true", help="the whole semi-supervised step (ustrun.trainer.SSLTrainer) around the model: "
                    "BUSI shapes, label_bs = unlabel_bs = --n")
    a = ap.parse_args()
    if a.ssl:
        return bench_ssl(a)
    from networks.deeplabv2 import DeepLabV2
    from ustrun import _lib
    lib = _lib.lib()
    torch.manual_seed(0)
    m = DeepLabV2(a.arch, 2, pretrained=False, dtype=a.dtype).cuda()
    m.train(a.mode == "train")
    x = torch.randn(a.n, 3, a.hw, a.hw, device="cuda")
    if a.backward:
        return bench_backward(a, m, x, lib, _lib)
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
        import collections
        cls = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for r in buf[:n]:
            c = cls[(round(r.flops / 1e9, 1), round(r.bytes / 1e6, 1))]
            c[0] += 1; c[1] += r.ms; c[2] += r.flops
        if os.environ.get("USTRUN_BENCH_SEQ"):
            print("   launch order:", " ".join(f"{r.flops / 1e9:.0f}:{r.ms * 1e3:.0f}" for r in buf[:n]))
        print("   by launch class (GFLOP, MB): launches, total ms, TF/s")
        for k, (cnt, ms_, fl_) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
            print(f"   {k[0]:8.1f} GF {k[1]:8.1f} MB  x{cnt:3d}  {ms_:7.3f} ms  {fl_ / ms_ / 1e9:6.0f} TF/s")
    msd, fld, byd, nd = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    lib.ustrun_profile_collect(0, ctypes.byref(msd), ctypes.byref(fld), ctypes.byref(byd), ctypes.byref(nd))
    print(f"DeepLabV2-{a.arch} {a.mode}-mode forward, N={a.n} {a.hw}x{a.hw}, {a.dtype}: {dt * 1e3:.2f} ms = {a.n / dt:.1f} images/s; "
          f"{n} convolution launches: {ms:.2f} ms, {fl / 1e9 / a.n:.1f} GFLOP/image algorithmic, {fl / ms / 1e9:.0f} TFLOP/s, "
          f"{by / ms / 1e6:.0f} GB/s algorithmic; convolutions are {ms / (dt * 1e3) * 100:.0f} % of the forward")


def bench_ssl(a):
    """BASELINE.json configs[4]: BUSI 512x512, 2-class DeepLabV2-ResNet, the reference's SSL step (3 teacher + 5(+1) student
    forwards, 4 backwards, CE + Dice, fused SGD + EMA) -- images/s = (label_bs + unlabel_bs) / step time."""
    import random
    import numpy as np
    from networks.deeplabv2 import DeepLabV2
    from ustrun import synthetic
    from ustrun.trainer import SSLTrainer
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    stu = DeepLabV2(a.arch, 2, pretrained=False, dtype=a.dtype).cuda()
    tea = DeepLabV2(a.arch, 2, pretrained=False, dtype=a.dtype).cuda()
    trn = SSLTrainer("BUSI", stu, tea, base_lr=1e-6, patch_size=a.hw, fft="device")
    pool = [[t.cuda() for t in synthetic.batch("BUSI", a.n, 1, a.hw, 77 + i)] for i in range(2)]
    for i in range(2):
        trn.step(*pool[i % 2], epoch_start=(i == 0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.reps):
        trn.step(*pool[i % 2])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    sc = trn.scalars()
    print(f"DeepLabV2-{a.arch} SSL step, BUSI {a.hw}x{a.hw}, label_bs = unlabel_bs = {a.n}, {a.dtype}: {dt * 1e3:.1f} ms/step = "
          f"{2 * a.n / dt:.1f} images/s; loss {sc['loss']:.4f}; peak device memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


def phase_times(step):
    """torch.cuda.Event pairs around the engine's backward helpers for one step: {helper: ms} (the events serialise nothing,
    the helpers already run back to back on one stream)"""
    from ustrun import resnet_engine as E
    names = ["_bn_backward", "_bn_apply", "_conv_wgrad", "_conv_dgrad", "_conv3_dgrad_bn2", "_conv2_dgrad_bn1", "_conv1_dgrad_join", "_finalize_fused_sums", "_join",
             "_zero_insert", "_head_backward", "_stem_backward", "conv_bn", "stem", "bottleneck"]
    recs, orig = {k: [] for k in names}, {k: getattr(E, k) for k in names}

    def wrap(k):
        def f(*args, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig[k](*args, **kw)
            e1.record()
            recs[k].append((e0, e1))
            return r
        return f

    for k in names:
        setattr(E, k, wrap(k))
    side, E._WGRAD_SIDE_STREAM = E._WGRAD_SIDE_STREAM, False      # (event pairs on one stream: the weight gradients back on it for this step)
    try:
        step()
        torch.cuda.synchronize()
    finally:
        E._WGRAD_SIDE_STREAM = side
        for k in names:
            setattr(E, k, orig[k])
    return {k: sum(e0.elapsed_time(e1) for e0, e1 in v) for k, v in recs.items()}


def bench_backward(a, m, x, lib, _lib):
    dl = torch.randn(a.n, 2, a.hw, a.hw, device="cuda")

    def step():
        for p in m.parameters():
            p.grad = None
        m(x).backward(dl)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    lib.ustrun_profile_enable(1)
    step()
    torch.cuda.synchronize()
    lib.ustrun_profile_enable(0)
    buf = (_lib.ProfRec * 8192)()
    n = lib.ustrun_profile_records(buf, 8192)
    out = []
    for kind, name in ((0, "forward + input-gradient convolutions"), (1, "weight-gradient GEMMs")):
        rs = [r for r in buf[:n] if r.kind == kind]
        ms, fl = sum(r.ms for r in rs), sum(r.flops for r in rs)
        out.append(f"{len(rs)} {name}: {ms:.2f} ms, {fl / max(ms, 1e-9) / 1e9:.0f} TFLOP/s")
        if os.environ.get("USTRUN_BENCH_VERBOSE"):
            for r in sorted(rs, key=lambda r: -r.ms)[:16]:
                print(f"   kind {kind} {r.ms:8.3f} ms  {r.flops / 1e9:9.1f} GF  {r.flops / r.ms / 1e9:7.0f} TF/s  {r.bytes / 1e6:8.1f} MB")
    for k in (0, 1):
        msd, fld, byd, nd = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        lib.ustrun_profile_collect(k, ctypes.byref(msd), ctypes.byref(fld), ctypes.byref(byd), ctypes.byref(nd))
    print(f"DeepLabV2-{a.arch} train-mode forward + backward, N={a.n} {a.hw}x{a.hw}, {a.dtype}: {dt * 1e3:.2f} ms = {a.n / dt:.1f} images/s; "
          + "; ".join(out) + f"; peak device memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    ph = phase_times(step)
    print("   stream time by engine helper (ms; bottleneck = the whole forward block, conv_bn inside it; _conv_dgrad includes _zero_insert; _conv3_dgrad_bn2 / _conv1_dgrad_join include their _bn_apply / _finalize_fused_sums / fallbacks): "
          + ", ".join(f"{k} {v:.2f}" for k, v in ph.items()))


if __name__ == "__main__":
    main()
