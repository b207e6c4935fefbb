#!/usr/bin/env python3
"""bench.py -- training-step throughput of the UST-RUN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one semi-supervised iteration (reference train.py:578-858 minus data loading and
logging: 3 teacher + 5 student forwards (+1 low-quality-sample forward), 4 backwards, losses,
SGD + EMA) on a synthetic batch resident in HBM.  Metric: train images/sec =
(label_bs + unlabel_bs) * world / step time.  For N > 1 either launch under torch.distributed.run
(one rank per GPU) or run `python bench.py --gpus N` plainly: it then starts the N ranks itself before
touching the GPU.  Every rank trains its own shard (weak scaling) and gradients are summed with one RCCL
all-reduce.  Prints ONE JSON line on rank 0.
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
    ap.add_argument("--dtype", default="bf16", choices=["f32", "f32x3", "bf16", "f16"],
                    help="bf16 = bf16 matrix-core operands, f32 accumulate/statistics (configs[1]); f32 = the exact parity path")
    ap.add_argument("--fft", default="device", choices=["host", "device"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP-event roofline leg")
    ap.add_argument("--dump-layers", default="", help="write every tagged launch class (layer, op, images) of the sampled steps to this JSON file")
    ap.add_argument("--dump-full", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"),
                    help="the FULL result (per-layer roofline tables, every secondary field) goes to this file; stdout carries "
                         "only the compact line (\"\" = no file)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` entries (the other BASELINE.json shapes, the f32 parity path, DeepLabV2-ResNet101 @512^2)")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(dataset):
    """The oracle's step (CPU restatement of the reference, oracle/step_ref.py) timed on this host's cores: config[0]
    (B = 4+4, fp32), 1 warm-up + 3 timed steps at all cores of the GPU's CPU share, plus a 1-thread
    figure from a bounded sample (one image forward + backward = 3 of the step's 65 image-forward units).  A reported
    baseline, not the target."""
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
    batches = [synthetic.batch(dataset, 4, C, H, 1337 + i) for i in range(4)]
    print(f"[bench] cpu_baseline: 1 warm-up + 3 timed oracle steps on {cores} threads ...", file=sys.stderr, flush=True)
    tr.step(*batches[0], epoch_start=True)
    times = []
    for i in range(1, 4):
        t0 = time.time()
        tr.step(*batches[i])
        times.append(time.time() - t0)
        print(f"[bench] cpu_baseline: step {i}: {times[-1]:.1f} s", file=sys.stderr, flush=True)
    dt = sum(times) / len(times)
    # 1 thread: one image forward + backward of the same network (3 F of a step's (16 B + 1) F = 65 F at B = 4)
    torch.set_num_threads(1)
    sd1 = U.clone_sd(sd, requires_grad=True)
    x1 = batches[0][0][:1]
    U.unet_forward(x1, sd1, train=True).square().mean().backward()          # warm-up
    t0 = time.time()
    U.unet_forward(x1, sd1, train=True).square().mean().backward()
    t1 = time.time() - t0
    torch.set_num_threads(cores)
    step_1t = t1 * (16 * 4 + 1) / 3.0
    return {"value": round(8 / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "torch": torch.__version__, "step_seconds": [round(t, 2) for t in times],
            "one_thread": {"value": round(8 / step_1t, 4), "unit": "images/sec", "cores": 1, "kind": "extrapolated",
                           "sample": f"NOT a timed step: 1 image forward+backward ({t1:.1f} s) scaled by 65/3 to a config[0] step"},
            "sample": f"3 steps after 1 warm-up of config[0]: {dataset} {H}x{H}, label_bs=unlabel_bs=4, fp32, "
                      f"oracle/step_ref.py on torch-CPU ({dt:.1f} s/step)"}


LAYER_NAMES = ["inc.conv1", "inc.conv2", "down1.conv1", "down1.conv2", "down2.conv1", "down2.conv2", "down3.conv1", "down3.conv2",
               "down4.conv1", "down4.conv2", "up1.conv1", "up1.conv2", "up2.conv1", "up2.conv2", "up3.conv1", "up3.conv2",
               "up4.conv1", "up4.conv2"]
HBM_PEAK, HBM_UNIT = 8000.0, "GB/s"


def tag_name(tag):
    op = {0: "fwd", 1: "dgrad", 2: "wgrad"}[tag // 100]
    t = tag % 100
    return (LAYER_NAMES[t] if t < 18 else f"up{t - 19}.up"), op


def layer_table(lib, mfma_peak):
    """Per-layer roofline rows from the HIP-event pairs of the sampled steps (ustrun_profile_records): for every tagged
    launch class (layer, op, images-in-launch) the mean duration, the launch's ALGORITHMIC flops and bytes, both achieved
    rates, both roof fractions and the roof that binds it (the larger of flops/peak_mfma and bytes/peak_hbm)."""
    from ustrun import _lib
    buf = (_lib.ProfRec * 4096)()
    n = lib.ustrun_profile_records(buf, 4096)
    agg = {}
    for r in buf[:max(n, 0)]:
        if r.tag < 0:
            continue
        k = (r.tag, r.n)
        a = agg.setdefault(k, [0, 0.0, r.flops, r.bytes])
        a[0] += 1
        a[1] += r.ms
    rows = []
    for (tag, nimg), (cnt, ms, fl, by) in sorted(agg.items()):
        name, op = tag_name(tag)
        t = ms / cnt * 1e-3
        tf, gb = fl / t / 1e12, by / t / 1e9
        rows.append({"layer": name, "op": op, "images": nimg, "launches": cnt, "ms": round(ms / cnt, 4), "alg_flops": fl, "alg_bytes": by,
                     "tflops": round(tf, 1), "gbps": round(gb, 1), "frac_mfma": round(tf / mfma_peak, 4),
                     "frac_hbm": round(gb / HBM_PEAK, 4),
                     "bound": "mfma" if fl / (mfma_peak * 1e12) >= by / (HBM_PEAK * 1e9) else "hbm"})
    return rows


def _collect(lib, kind):
    ms, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    lib.ustrun_profile_collect(kind, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(n))
    return ms.value, fl.value, by.value, n.value


def _class_roofline(lib, peak, sampled):
    """conv / weight-gradient class rows from the HIP-event pairs recorded since the last collect (same accounting as the
    headline's `roofline`: algorithmic flops / summed launch time)"""
    out = {}
    for kind, name in ((0, "conv"), (1, "wgrad")):
        ms, fl, by, n = _collect(lib, kind)
        if ms > 0:
            tf = fl / (ms * 1e-3) / 1e12
            out[name] = {"achieved": round(tf, 1), "unit": "TFLOP/s", "peak": peak, "frac": round(tf / peak, 4), "bound": "mfma",
                         "launches_per_step": n // max(sampled, 1), "ms_per_step": round(ms / max(sampled, 1), 3),
                         "alg_gbps": round(by / (ms * 1e-3) / 1e9, 1)}
    return out


def secondary_unet(dev, dataset, lb, dtype, steps, warmup, lib):
    """The SSL step at another BASELINE.json shape (per-rank batch of the multi-GPU configs) or in the f32 parity path."""
    from networks.unet_model import UNet
    from ustrun import synthetic
    from ustrun.trainer import DATASETS, SSLTrainer
    C, H, K = DATASETS[dataset][:3]
    torch.manual_seed(1337)
    model, ema = UNet(C, K, dtype=dtype).to(dev), UNet(C, K, dtype=dtype).to(dev)
    tr = SSLTrainer(dataset, model, ema, fft="device")
    random.seed(1212); np.random.seed(1337)
    batches = [[t.to(dev) for t in synthetic.batch(dataset, lb, C, H, 1337 + i)] for i in range(2)]
    for s in range(warmup):
        tr.step(*batches[s % 2], epoch_start=(s == 0))
    lib.ustrun_profile_stream(torch.cuda.current_stream(dev).cuda_stream, 1)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for s in range(steps):
        tr.step(*batches[(warmup + s) % 2])
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    # the per-launch HIP events cost time: the profiled step runs AFTER the clock has stopped
    lib.ustrun_profile_enable(1)
    tr.step(*batches[(warmup + steps) % 2])
    torch.cuda.synchronize(dev)
    lib.ustrun_profile_enable(0)
    # (f32x3: six bf16 MFMAs per algorithmic multiply-add -> its roof is the bf16 matrix peak / 6 = 416.7 TFLOP/s of algorithmic work)
    peak = 157.3 if dtype == "f32" else (416.7 if dtype == "f32x3" else 2500.0)
    return {"workload": f"{dataset} {H}x{H}, {K}-class U-Net, SSL step, batch={lb}+{lb}", "dtype": dtype, "steps": steps,
            "ms_per_step": round(dt * 1e3, 3), "images_per_s": round(2 * lb / dt, 2), "roofline": _class_roofline(lib, peak, 1)}


def secondary_deeplab(dev, n, hw, dtype, lib, ssl, steps):
    """BASELINE.json configs[4]: DeepLabV2-ResNet101 on 512 x 512 (BUSI): train-mode forward + backward of n images, or the
    whole SSL step with label_bs = unlabel_bs = n."""
    from networks.deeplabv2 import DeepLabV2
    from ustrun import synthetic
    from ustrun.trainer import SSLTrainer
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    stu = DeepLabV2("resnet101", 2, pretrained=False, dtype=dtype).to(dev)
    if ssl:
        tea = DeepLabV2("resnet101", 2, pretrained=False, dtype=dtype).to(dev)
        trn = SSLTrainer("BUSI", stu, tea, base_lr=1e-6, patch_size=hw, fft="device")
        pool = [[t.to(dev) for t in synthetic.batch("BUSI", n, 1, hw, 77 + i)] for i in range(2)]
        step = lambda i: trn.step(*pool[i % 2], epoch_start=(i == 0 and not step.warm))
        step.warm = False
        imgs, what = 2 * n, f"SSL step, batch={n}+{n}"
    else:
        stu.train()
        x = torch.randn(n, 3, hw, hw, device=dev)
        dl = torch.randn(n, 2, hw, hw, device=dev)

        def step(i):
            for p in stu.parameters():
                p.grad = None
            stu(x).backward(dl)
        imgs, what = n, f"train-mode forward + backward, batch={n}"
    step(0)
    if ssl:
        step.warm = True
        step(1)
    lib.ustrun_profile_stream(torch.cuda.current_stream(dev).cuda_stream, 1)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    # profiled step outside the timed window -- and with the weight gradients back on the main stream: the timed steps
    # overlap them with the input-gradient chain (ustrun/resnet_engine.py), which makes every co-running kernel slower and the step
    # faster; the class rates below are the kernels' own
    from ustrun import resnet_engine
    side, resnet_engine._WGRAD_SIDE_STREAM = resnet_engine._WGRAD_SIDE_STREAM, False
    lib.ustrun_profile_enable(1)
    try:
        step(steps)
        torch.cuda.synchronize(dev)
    finally:
        lib.ustrun_profile_enable(0)
        resnet_engine._WGRAD_SIDE_STREAM = side
    return {"workload": f"BUSI {hw}x{hw}, 2-class DeepLabV2-ResNet101, {what}", "dtype": dtype, "steps": steps,
            "ms_per_step": round(dt * 1e3, 3), "images_per_s": round(imgs / dt, 2), "roofline": _class_roofline(lib, 2500.0, 1),
            "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1)}


def secondary_runs(dev, lib):
    """Every other workload this repository claims a number for, under the same clock as the headline.
    Each entry frees its memory before the next; a failing entry is reported, not fatal."""
    import gc
    jobs = [("unet", dict(dataset="fundus", lb=16, dtype="f32", steps=3, warmup=1)),
            # f32 tensors, the convolutions' products as six bf16 MFMAs over three-term operand splits (csrc/x3.hip): the
            # north_star's 1e-4 / bit-exact arg-max tolerance at a multiple of the f32 matrix rate
            ("unet", dict(dataset="fundus", lb=16, dtype="f32x3", steps=5, warmup=2)),
            # the reference's own mixed-precision mode (--amp 1: IEEE half + GradScaler) on configs[1]'s shape
            ("unet", dict(dataset="fundus", lb=16, dtype="f16", steps=10, warmup=3)),
            ("unet", dict(dataset="prostate", lb=8, dtype="bf16", steps=10, warmup=2)),
            ("unet", dict(dataset="MNMS", lb=8, dtype="bf16", steps=10, warmup=2)),
            ("deeplab", dict(n=16, hw=512, dtype="bf16", ssl=False, steps=5)),
            ("deeplab", dict(n=16, hw=512, dtype="bf16", ssl=True, steps=3))]
    out = []
    for kind, kw in jobs:
        t0 = time.time()
        print(f"[bench] secondary: {kind} {kw} ...", file=sys.stderr, flush=True)
        try:
            out.append(secondary_unet(dev, lib=lib, **kw) if kind == "unet" else secondary_deeplab(dev, lib=lib, **kw))
        except Exception as e:          # noqa: BLE001 -- the headline line must still be printed
            out.append({"workload": f"{kind} {kw}", "error": f"{type(e).__name__}: {e}"[:300]})
        _collect(lib, 0); _collect(lib, 1)
        gc.collect(); torch.cuda.empty_cache()
        print(f"[bench] secondary: done in {time.time() - t0:.1f} s", file=sys.stderr, flush=True)
    return out


def spawn_ranks(a):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as children (one process per GPU, RCCL)
    BEFORE this process touches the GPU, relay rank 0's JSON line, exit with their status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


PARITY_NOTE = {
    "bf16": "bf16 operands + stored activations, f32 accumulate: logits <= 3e-2 rel-L2, <= 2 % arg-max flips vs the reference "
            "fixtures; the north_star's 1e-4 / bit-exact arg-max bar is met by f32 and f32x3 (secondary[0,1]); secondary[2] = "
            "f16 + loss scale, the reference's --amp arithmetic (5e-3, <= 0.3 %)",
    "f16": "f16 operands + stored activations, f32 accumulate, GradScaler's loss-scale schedule (the reference's --amp): logits "
           "<= 5e-3 rel-L2, <= 0.3 % arg-max flips vs the reference fixtures",
    "f32": "the exact parity path (1e-4 rel, arg-max bit-exact outside 1e-6 margins)",
    "f32x3": "f32 tensors, products as six bf16 MFMAs over three-term splits: reference logits to 3e-6 rel-L2, arg-max flips only "
             "at margins < 3e-6 -- the north_star tolerance"}


LINE_LIMIT = 6000          # characters of the ONE stdout line (the driver parses a bounded tail of stdout: VERDICT r5, weak 1)


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def compact_line(out):
    """The ONE JSON line the driver parses, from the full result dict: headline fields, `config`, `roofline` (scalars, the
    weight-gradient class, the two DoubleConv blocks the north_star names, the traffic fields), `cpu_baseline`, a short
    `parity_note`, and `secondary` reduced to workload / dtype / ms / images per s / class fractions.  The per-layer tables
    (`roofline.layers`, `layers_bwd`) never travel on this line: they go to --dump-full / --dump-layers files and stderr."""
    o = {k: v for k, v in out.items() if k not in ("roofline", "cpu_baseline", "secondary", "parity_note")}
    r = out.get("roofline")
    if r is not None:
        c = _pick(r, ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "launches_per_step", "avg_launch_ms",
                      "alg_flops_per_launch", "alg_bytes_per_launch", "time_share_of_step", "clock_mhz", "sustained_peak",
                      "frac_sustained", "traffic_per_step", "alg_bytes_per_step_all_streams", "traffic_ratio_by_totals"))
        for k in ("alg_flops_per_launch", "alg_bytes_per_launch", "alg_bytes_per_step_all_streams"):
            if k in c:
                c[k] = float(f"{c[k]:.5g}")
        if r.get("wgrad"):
            c["wgrad"] = _pick(r["wgrad"], ("achieved", "unit", "frac", "traffic", "time_share_of_step", "traffic_ratio_by_totals"))
        if r.get("doubleconv"):
            c["doubleconv"] = [_pick(b, ("block", "op", "images", "ms", "gbps", "frac_hbm", "frac_mfma")) for b in r["doubleconv"]]
        if r.get("traffic_note"):
            c["traffic_note"] = "profiles/traffic file is of another library build: dropped"
        if r.get("layers_file"):
            c["layers_file"] = r["layers_file"]
        o["roofline"] = c
    else:
        o["roofline"] = None
    b = out.get("cpu_baseline")
    if b is not None:
        c = _pick(b, ("value", "unit", "cores", "kind", "sample", "cpu", "step_seconds"))
        if b.get("one_thread"):
            c["one_thread_extrapolated"] = b["one_thread"].get("value")
        o["cpu_baseline"] = c
    else:
        o["cpu_baseline"] = None
    o["parity_note"] = out.get("parity_note")
    if "secondary" in out:
        sec = []
        for s in out["secondary"]:
            e = _pick(s, ("workload", "dtype", "ms_per_step", "images_per_s", "peak_mem_gib"))
            if "error" in s:
                e["error"] = s["error"][:120]
            for cls in ("conv", "wgrad"):
                if cls in s.get("roofline", {}):
                    e[cls + "_frac"] = s["roofline"][cls]["frac"]
            sec.append(e)
        o["secondary"] = sec
    line = json.dumps(o, allow_nan=False)
    if len(line) > LINE_LIMIT:         # never print a line the driver cannot parse: shed the optional parts, largest first
        for k in ("secondary", "parity_note"):
            o.pop(k, None)
            line = json.dumps(o, allow_nan=False)
            if len(line) <= LINE_LIMIT:
                break
    return line


def held_clock_mhz(pa, pb, nb):
    """median over the XCDs of d(s_memtime) / d(s_memrealtime) x 100 MHz between two ustrun_debug_clock_probe outputs"""
    a, b = pa.view(nb, 4).cpu().numpy(), pb.view(nb, 4).cpu().numpy()
    out = []
    for x in range(8):
        ia, ib = np.nonzero(a[:, 2] == x)[0], np.nonzero(b[:, 2] == x)[0]
        if len(ia) and len(ib):
            d_clk = float(np.median(b[ib, 0])) - float(np.median(a[ia, 0]))
            d_ref = float(np.median(b[ib, 1])) - float(np.median(a[ia, 1]))
            if d_ref > 0:
                out.append(d_clk / d_ref * 100.0)
    return float(np.median(out)) if out else None


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)                                # never returns; nothing above has initialised the GPU
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    rccl_ranks, backend = 1, None
    if world > 1:
        import torch.distributed as dist
        from ustrun import ddp
        ddp.init("nccl", device=dev)              # "nccl" is RCCL on ROCm
        rccl_ranks, backend = dist.get_world_size(), dist.get_backend()

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
    lib.ustrun_profile_stream(torch.cuda.current_stream(dev).cuda_stream, 1)   # not the batch-1 forward's side stream
    sync()
    prof = not a.no_profile
    # the HIP-event pairs around every conv launch cost ~5 % of a step, so they sample the LAST min(2, K) steps of the
    # timed region rather than all of it (the whole region is still what `value` is computed from)
    nprof = min(2, a.steps) if prof else 0
    # the shader clock the chip HOLDS over the timed steps: two probes (s_memtime / s_memrealtime per workgroup, a ~5 us
    # launch each) on the step's stream around the region -> d(memtime) / d(memrealtime) x 100 MHz, median over the XCDs
    NPB = 1024
    pa = torch.zeros(NPB * 4, dtype=torch.int64, device=dev)
    pb = torch.zeros(NPB * 4, dtype=torch.int64, device=dev)
    cur = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.ustrun_debug_clock_probe(pa.data_ptr(), NPB, cur))
    t0 = time.perf_counter()
    for s in range(a.steps):
        if s == a.steps - nprof:
            lib.ustrun_profile_enable(1)
        tr.step(*batches[(a.warmup + s) % nb])
    _lib.check(lib.ustrun_debug_clock_probe(pb.data_ptr(), NPB, cur))
    sync()
    dt = time.perf_counter() - t0
    lib.ustrun_profile_enable(0)
    clock_mhz = held_clock_mhz(pa, pb, NPB)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    roof = None
    if prof:
        layers = layer_table(lib, 157.3 if a.dtype == "f32" else (416.7 if a.dtype == "f32x3" else 2500.0))
        ms, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        lib.ustrun_profile_collect(0, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(n))
        peak = 157.3 if a.dtype == "f32" else (416.7 if a.dtype == "f32x3" else 2500.0)
        ach = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        roof = {"kernel": "DoubleConv convolutions: conv3x3 forward + input-gradient (halo-tiled implicit GEMM) and ConvTranspose",
                "bound": "mfma", "achieved": round(ach, 2),
                "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                "launches_per_step": n.value // max(nprof, 1), "sampled_steps": nprof,
                "avg_launch_ms": round(ms.value / max(n.value, 1), 4),
                "alg_flops_per_launch": fl.value / max(n.value, 1), "alg_bytes_per_launch": by.value / max(n.value, 1),
                "alg_gbps": round(by.value / (ms.value * 1e-3) / 1e9, 1) if ms.value > 0 else 0.0,
                "time_share_of_step": round(ms.value * 1e-3 / (dt * nprof / a.steps), 3)}
        # `peak` above is the data-sheet figure (2.4 GHz); the chip does not hold 2.4 GHz under this load, so the same achieved
        # rate is also priced against the MFMA peak AT THE CLOCK IT HELD over these steps (4096 flop / clk / CU x 256 CUs in
        # 16-bit; the f32 figure scaled the same way) -- a whole-step average: MFMA-dense kernels run below it, HBM passes above
        if clock_mhz:
            sus = peak * clock_mhz / 2400.0
            roof.update({"clock_mhz": round(clock_mhz, 0), "sustained_peak": round(sus, 1),
                         "frac_sustained": round(ach / sus, 4)})
        ms2, fl2, by2, n2 = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        lib.ustrun_profile_collect(1, ctypes.byref(ms2), ctypes.byref(fl2), ctypes.byref(by2), ctypes.byref(n2))
        if ms2.value > 0:
            roof["wgrad"] = {"achieved": round(fl2.value / (ms2.value * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                             "frac": round(fl2.value / (ms2.value * 1e-3) / 1e12 / peak, 4),
                             "alg_bytes_per_launch": by2.value / max(n2.value, 1), "traffic": None,
                             "alg_gbps": round(by2.value / (ms2.value * 1e-3) / 1e9, 1),
                             "time_share_of_step": round(ms2.value * 1e-3 / (dt * nprof / a.steps), 3)}
        # the north_star's HBM target is defined on the 64-channel full-resolution DoubleConv layers:
        # per-layer rows of the largest launches (the student's four batched passes), both roofs, binding roof named
        if a.dump_layers and rank == 0:
            json.dump(layers, open(a.dump_layers, "w"), indent=0)
        big = max((r["images"] for r in layers), default=0)
        roof["layers"] = [r for r in layers if r["images"] == big and r["op"] == "fwd"]
        # (the backward covers the gradient passes only: fewer images than the forward since the leading and tail passes ride along)
        big_bwd = max((r["images"] for r in layers if r["op"] != "fwd"), default=0)
        roof["layers_bwd"] = [r for r in layers if r["images"] == big_bwd and r["op"] != "fwd"]
        # ... and the two DoubleConv BLOCKS that target names (inc, up4.conv), forward: both convolutions' algorithmic bytes
        # (read x, write y1, read y1, write y2) over both launches' time, against the HBM roof
        dc = []
        for blk, names in (("inc", ("inc.conv1", "inc.conv2")), ("up4.conv", ("up4.conv1", "up4.conv2"))):
            rows = [r for r in roof["layers"] if r["layer"] in names]
            if len(rows) == 2:
                ms = sum(r["ms"] for r in rows)
                by, fl = sum(r["alg_bytes"] for r in rows), sum(r["alg_flops"] for r in rows)
                dc.append({"block": blk, "op": "fwd", "images": big, "ms": round(ms, 4), "alg_bytes": by, "alg_flops": fl,
                           "gbps": round(by / ms / 1e6, 1), "frac_hbm": round(by / ms / 1e6 / HBM_PEAK, 4),
                           "tflops": round(fl / ms / 1e9, 1), "frac_mfma": round(fl / ms / 1e9 / peak, 4)})
        roof["doubleconv"] = dc
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), measured offline with
        # rocprofv3 --pmc on this same command and committed under profiles/ (bench.py cannot run under two profilers)
        tpath = os.path.join(ROOT, "profiles", f"traffic_{a.dtype}.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # the counters belong to ONE build of the library: a file read from another build says nothing about this run
                import hashlib
                so = os.path.join(ROOT, "ust-run_amd", "ustrun", "libustrun.so")
                sha = hashlib.sha256(open(so, "rb").read()).hexdigest()[:16]
                if tj.get("library_sha16") != sha:
                    roof["traffic_note"] = (f"profiles/traffic_{a.dtype}.json was measured on library build {tj.get('library_sha16')}, "
                                            f"this run loaded {sha}: traffic dropped (re-run tools/profile_round.sh)")
                    raise KeyError("stale traffic file")
                roof["traffic"] = tj["hbm_bytes_per_launch"]
                if "wgrad" in roof:
                    roof["wgrad"]["traffic"] = tj.get("wgrad_hbm_bytes_per_launch")
                # the counters see EVERY launch of the class (the batch-1 forward on its side stream too) while `achieved` is
                # timed on the step's stream only, so the ratio is taken by TOTALS over one launch set: the class's HBM bytes per
                # step from the counters over its algorithmic bytes per step on all streams (one extra, untimed step below)
                roof["traffic_per_step"] = tj.get("hbm_bytes_per_step")
                if "wgrad" in roof:
                    roof["wgrad"]["traffic_per_step"] = tj.get("wgrad_hbm_bytes_per_step")
            except Exception:
                roof.setdefault("traffic", None)
    if prof and roof is not None:
        lib.ustrun_profile_stream(torch.cuda.current_stream(dev).cuda_stream, 0)      # every stream: bytes only, times unused
        lib.ustrun_profile_enable(1)
        tr.step(*batches[0])
        sync()
        lib.ustrun_profile_enable(0)
        for kind, r in ((0, roof), (1, roof.get("wgrad"))):
            if r is None:
                continue
            ms, fl, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
            lib.ustrun_profile_collect(kind, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by), ctypes.byref(n))
            r["alg_bytes_per_step_all_streams"] = by.value
            r["launches_per_step_all_streams"] = n.value
            if r.get("traffic_per_step") and by.value > 0:
                r["traffic_ratio_by_totals"] = round(r["traffic_per_step"] / by.value, 3)
    if rank == 0:
        imgs = (a.label_bs + a.unlabel_bs) * world * a.steps
        out = {"metric": "train images/sec (256x256 U-Net, mixed lb+ulb batch)", "value": round(imgs / dt, 3),
               "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
               "config": {"workload": f"{a.dataset} {H}x{H}, {K}-class U-Net (31.0M params), batch={a.label_bs}+{a.unlabel_bs} per GPU, "
                                      f"SSL step = 3 teacher + 5(+1) student forwards, 4 backwards, CE+Dice, SGD+EMA",
                          "global_batch": (a.label_bs + a.unlabel_bs) * world, "parallelism": f"dp{world}", "fft_mix": a.fft},
               "rccl_ranks": rccl_ranks, "collective_backend": backend,
               "roofline": roof, "cpu_baseline": None,
               "parity_note": PARITY_NOTE[a.dtype]}
        if world == 1 and not a.no_secondary:
            del tr, model, ema, batches
            import gc
            gc.collect(); torch.cuda.empty_cache()
            out["secondary"] = secondary_runs(dev, lib)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.dataset)
        if a.dump_full:
            try:
                os.makedirs(os.path.dirname(os.path.abspath(a.dump_full)), exist_ok=True)
                with open(a.dump_full, "w") as f:
                    json.dump(out, f)
                if out["roofline"] is not None:
                    out["roofline"]["layers_file"] = os.path.relpath(a.dump_full, ROOT)
            except OSError as e:
                print(f"[bench] --dump-full {a.dump_full}: {e}", file=sys.stderr, flush=True)
        print(compact_line(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
