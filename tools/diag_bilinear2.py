"""dev: the bilinear adjoint at the shapes of the odd-extent case"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ust-run_amd")]
import torch, torch.nn.functional as TF
from ustrun.blocks import _BilinearFn
g = torch.Generator().manual_seed(4)
for n, c, h, w in ((4, 16, 24, 36), (4, 32, 12, 18), (4, 64, 6, 9), (4, 128, 3, 4), (2, 8, 16, 16)):
    x = torch.randn(n, c, h, w, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = TF.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy)
    xg = x.cuda().requires_grad_(True)
    yg = _BilinearFn.apply(xg)
    ef = float((yg.permute(0, 3, 1, 2).cpu() - yr.detach()).norm() / yr.detach().norm())
    yg.backward(dy.permute(0, 2, 3, 1).contiguous().cuda())
    eb = float((xg.grad.cpu() - xr.grad).norm() / xr.grad.norm())
    d = (xg.grad.cpu() - xr.grad)
    print(f"{n}x{c}x{h}x{w}: fwd rel {ef:.2e} bwd rel {eb:.2e}; bwd abs err by column (max over rest): {[round(float(v), 4) for v in d.abs().amax((0, 1, 2))][:40]}")
