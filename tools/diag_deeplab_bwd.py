"""Conditioning of the DeepLabV2 train-mode gradient check in bf16 (diagnostic behind tests/test_gpu_deeplab_bwd.py): cosine of the
HIP gradient field against the float64 oracle for several damping factors of the residual branches' last BatchNorm (gamma of bn3
and of the projection shortcut's BatchNorm scaled by `damp`), batch sizes and extents.

    python tools/diag_deeplab_bwd.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ust-run_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import deeplab_ref as D
from test_gpu_deeplab_bwd import _oracle_grads, _field_stats, rel
from networks.deeplabv2 import DeepLabV2


def run(dtype, damp, n, h, w, seed=23, rounded=False, ref=None):
    sd = D.make_state_dict("resnet50", 2, seed)
    for k in sd:
        if k.endswith("bn3.weight"):
            sd[k] = sd[k] * damp
    sd0 = sd
    if rounded:                       # yardstick: the f32 path's response to a bf16 rounding of its parameters and input
        sd = {k: (v.bfloat16().float() if v.is_floating_point() and v.dim() == 4 else v) for k, v in sd.items()}
    torch.manual_seed(seed)
    m = DeepLabV2("resnet50", 2, pretrained=False, dtype=dtype)
    m.load_state_dict(sd)
    m = m.cuda().train()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, 3, h, w, generator=g)
    R = torch.randn(n, 2, h, w, generator=g)
    ref_out, ref = _oracle_grads(x, sd0, "resnet50", R, torch.float64)
    if rounded:
        x = x.bfloat16().float()
    out = m(x.cuda())
    (out * R.cuda()).sum().backward()
    got = {k: p.grad.cpu() for k, p in m.named_parameters()}
    w_, wk, cos, ratio = _field_stats(got, ref)
    errs = sorted(rel(got[k], ref[k]) for k in ref)
    print(f"{'yardstick ' if rounded else ''}{dtype} damp {damp} n{n} {h}x{w}: logits rel {rel(out.detach().cpu(), ref_out):.2e}  grad cosine {cos:.4f} ratio {ratio:.3f} "
          f"median rel {errs[len(errs) // 2]:.2e} worst {w_:.2e} ({wk})", flush=True)


if __name__ == "__main__":
    for damp, n, h, w in ((1.0, 2, 96, 80), (0.25, 2, 96, 80), (0.1, 4, 128, 96)):
        for dtype in ("f32", "bf16"):
            run(dtype, damp, n, h, w)
        run("f32", damp, n, h, w, rounded=True)
