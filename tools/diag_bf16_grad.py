#!/usr/bin/env python3
"""GPU box: per-parameter rel-L2 of the bf16 path's gradients against the f32 path's at BASELINE configs[1]'s shape, in
network order, for two losses -- tells accumulated rounding noise (grows smoothly with depth) from a kernel slip (jumps at a
layer)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ust-run_amd")]
import numpy as np
import torch
from networks.unet_model import UNet
from oracle import unet_ref as U

def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))

torch.manual_seed(1337)
sd = U.make_state_dict(3, 2)
gen = torch.Generator().manual_seed(16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = (torch.randint(0, 256, (n, 3, 256, 256), generator=gen).float() / 127.5 - 1).cuda()
dl = torch.randn(n, 2, 256, 256, generator=gen).cuda() / (n * 2 * 256 * 256)
for lossname in ("square_mean", "random_dl"):
    out = {}
    for dt in ("f32", "bf16", "f32b"):          # f32b: a second f32 run on inputs rounded to bf16 (the input-rounding share)
        m = UNet(3, 2, dtype="f32" if dt == "f32b" else dt)
        m.load_state_dict({k: v.clone() for k, v in sd.items()})
        m = m.cuda().train()
        xi = x.bfloat16().float() if dt == "f32b" else x
        lg = m(xi)
        if lossname == "square_mean":
            lg.square().mean().backward()
        else:
            lg.backward(dl)
        out[dt] = (lg.detach().cpu(), [p.grad.detach().cpu() for p in m.parameters()], [k for k, _ in m.named_parameters()])
        del m, lg
    print(lossname, "logits rel", rel(out["bf16"][0], out["f32"][0]), "f32b", rel(out["f32b"][0], out["f32"][0]))
    for k, g16, g32, gb in zip(out["f32"][2], out["bf16"][1], out["f32"][1], out["f32b"][1]):
        print("  %-45s bf16 %.3e   f32(bf16-rounded input) %.3e   |g| %.3e" % (k, rel(g16, g32), rel(gb, g32), float(g32.norm())))
