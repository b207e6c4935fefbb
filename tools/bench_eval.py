"""Throughput of the validation path (eval-mode forward + device prediction + Dice counts), Fundus 256x256.

    python tools/bench_eval.py [--dtype bf16] [--bs 1 16 64] [--batches 20]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ust-run_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--bs", type=int, nargs="+", default=[1, 16, 64])
    ap.add_argument("--batches", type=int, default=20)
    ap.add_argument("--dataset", default="fundus")
    ap.add_argument("--coalesce", type=int, default=64)
    a = ap.parse_args()
    from networks.unet_model import UNet
    from ustrun import synthetic
    from ustrun.evaluate import validate
    from ustrun.trainer import DATASETS
    C, H, K = DATASETS[a.dataset][:3]
    torch.manual_seed(0)
    model = UNet(n_channels=C, n_classes=K, dtype=a.dtype).cuda()
    for bs in a.bs:
        loaders = [[(x.cuda(), y.cuda()) for x, y in dom] for dom in synthetic.test_loaders(a.dataset, 1, a.batches, bs, C, H, 3)]
        validate(a.dataset, model, loaders, log=None, coalesce=a.coalesce)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        validate(a.dataset, model, loaders, log=None, coalesce=a.coalesce)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{a.dataset} {a.dtype} test_bs={bs:3d}: {bs * a.batches / dt:9.1f} images/s  ({dt / a.batches * 1e3:.2f} ms per batch)", flush=True)


if __name__ == "__main__":
    main()
