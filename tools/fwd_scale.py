"""GPU box: U-Net forward time against the number of images in a call (one pass, or several passes batched with BatchNorm per
pass): how much the 16-image student weak-view forward and the 48-image teacher call lose to partly filled deep-layer grids
(measured: 7.5 / 9.1 / 9.2 images per ms at 16 / 48 / 64).

    python tools/fwd_scale.py"""
import sys, time, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ust-run_amd')]
from networks.unet_model import UNet
torch.manual_seed(0)
m=UNet(3,2,dtype='bf16').cuda().train()
for n,passes in ((16,1),(16,3),(16,4),(32,1),(48,1),(64,1)):
    xs=[torch.randn(n,3,256,256,device='cuda') for _ in range(passes)]
    with torch.no_grad():
        f=(lambda: m.forward_passes(xs)) if passes>1 else (lambda: m(xs[0]))
        for _ in range(3): f()
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(10): f()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print(f"N={n} x {passes} passes: {dt*1e3:.3f} ms  {n*passes/dt/1e3:.2f} img/ms")
