"""dev: UNet(bilinear=True) fused plan and block modules against the CPU oracle: per-parameter gradient rel-L2 at an odd-extent shape"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ust-run_amd")]
import torch
from networks.unet_model import UNet
from oracle import unet_ref as U
base, n, h, w = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (16, 4, 48, 72))]
torch.manual_seed(11)
sd = U.make_state_dict(3, 2, bilinear=True, base=base)
x = torch.randn(n, 3, h, w)
ref_sd = U.clone_sd(sd, requires_grad=True)
ref = U.unet_forward(x, ref_sd, train=True, bilinear=True)
ref.square().mean().backward()
res = {}
for name in ("fused", "blocks"):
    m = UNet(3, 2, bilinear=True, base_channels=base)
    m.load_state_dict({k: v.detach().clone() for k, v in sd.items()})
    m = m.cuda().train()
    out = m(x.cuda()) if name == "fused" else m._forward_blocks(x.cuda())
    print(name, "logits rel-L2", float((out.detach().cpu() - ref.detach()).norm() / ref.detach().norm()))
    out.square().mean().backward()
    res[name] = {k: float((p.grad.double().cpu() - ref_sd[k].grad.double()).norm() / (ref_sd[k].grad.double().norm() + 1e-30)) for k, p in m.named_parameters()}
for k in res["fused"]:
    print(f"{k:34s} fused {res['fused'][k]:.2e}  blocks {res['blocks'][k]:.2e}")
