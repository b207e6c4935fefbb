"""dev: f32x3 batched passes -- one launch per convolution (default) vs one launch per pass (ustrun_debug_flags2 bit 1): logits, buffers, gradients"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ust-run_amd")]
import torch
from networks.unet_model import UNet
from ustrun import _lib
lib = _lib.lib()
torch.manual_seed(1)
for (n, hw, passes, lead, tail) in [(4, 256, 5, 1, 1), (2, 64, 3, 0, 1), (4, 256, 4, 0, 0)]:
    m1 = UNet(3, 2, dtype="f32x3").cuda().train()
    m2 = copy.deepcopy(m1)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n * passes + tail, 3, hw, hw, generator=g).cuda()
    dl = torch.randn(n * passes, 2, hw, hw, generator=g).cuda()
    lib.ustrun_debug_flags2(0)
    a = m1.forward_batched(x, passes, tail=tail, lead=lead); a.backward(dl)
    torch.cuda.synchronize()
    lib.ustrun_debug_flags2(2)
    b = m2.forward_batched(x, passes, tail=tail, lead=lead); b.backward(dl)
    torch.cuda.synchronize()
    lib.ustrun_debug_flags2(0)
    print(f"n={n} hw={hw} passes={passes} lead={lead} tail={tail}: logits equal {torch.equal(a, b)} max|d| {float((a - b).abs().max()):.3e}")
    for (k, b1), (_, b2) in zip(m1.named_buffers(), m2.named_buffers()):
        if not torch.equal(b1, b2):
            print("   buffer differs", k, float((b1.double() - b2.double()).abs().max())); break
    worst = (0.0, "")
    for (k, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        e = float((p1.grad - p2.grad).norm() / (p2.grad.norm() + 1e-30))
        worst = max(worst, (e, k))
    print("   worst gradient rel-L2", worst)
