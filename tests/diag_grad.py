"""Diagnostic (run by hand, not collected by pytest): where a BatchNorm beta gradient of the deep layers differs between
the HIP path and the CPU oracle -- conditioning of the problem, not a kernel error.  Lives under tests/ because it uses
the oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ust-run_amd"), os.path.join(ROOT, "tests")]
import torch, torch.nn.functional as F
from oracle import unet_ref as U
import ustrun.engine as E
from test_gpu_unet import rel_l2
# capture oracle conv outputs
ys = []
xs = []
orig = F.conv2d
def rec(x, w, b=None, s=1, p=0):
    y = orig(x, w, b, s, p)
    if w.shape[-1] == 3:
        y.retain_grad(); ys.append(y)
        if x.requires_grad and not x.is_leaf: x.retain_grad()
        xs.append(x)
    return y
F.conv2d = rec
# keep the scratch buffer alive for inspection
orig_empty = torch.empty
keep = {}
from test_gpu_unet import run_pair
model, logits, ref_sd, ref_logits, sd64, l64 = run_pair(1, 2, 2, 32, 32, 64, seed=5, want64=True)
ys32 = ys[:18]
ref_logits.square().mean().backward()
class Hook:
    pass
def patched_empty(*a, **k):
    t = orig_empty(*a, **k)
    if k.get("dtype") == torch.uint8 and "scratch" not in keep and t.numel() > 1e6 and keep.get("arm"):
        keep["scratch"] = t
    return t
torch.empty = patched_empty
keep["arm"] = True
wsave = logits.grad_fn.ws
from ustrun import _lib
_lib.lib().ustrun_debug_flags((6 + 1) << 16)        # stop the backward before layer 6 (bits 16-20 = layer + 1)
logits.square().mean().backward()
torch.cuda.synchronize()
sc = keep["scratch"].view(torch.float32)
ws = wsave.view(torch.float32)
# forward workspace layout (unet.hip make_plan)
yo, ao, o = [], [], 0
for y in ys32:
    n, c, h, w = y.shape
    yo.append(o); o = (o + n*c*h*w + 63)//64*64
    ao.append(o); o = (o + 4*c + 63)//64*64
do, o = [], 0
for y in ys32:
    n, c, h, w = y.shape
    do.append(o); o = (o + n*c*h*w + 63)//64*64
i = 6
n, c, h, w = ys32[i].shape
y6 = ws[yo[i]:yo[i]+n*c*h*w].view(n,h,w,c).permute(0,3,1,2).cpu()
print("y6 err", rel_l2(y6, ys32[i].detach()))
aff = ws[ao[i]:ao[i]+4*c].cpu().view(4, c)
yr = ys32[i].detach()
var, mean = torch.var_mean(yr, dim=(0,2,3), unbiased=False)
rstd = torch.rsqrt(var+1e-5)
g = ref_sd["down3.maxpool_conv.1.double_conv.1.weight"].detach(); b = ref_sd["down3.maxpool_conv.1.double_conv.1.bias"].detach()
print("scale err", rel_l2(aff[0], g*rstd), "shift err", rel_l2(aff[1], b-mean*g*rstd), "mean err", rel_l2(aff[2], mean), "rstd err", rel_l2(aff[3], rstd))
print("mean/std ratio max", float((mean.abs()*rstd).max()))
da = sc[do[i]:do[i]+n*c*h*w].view(n,h,w,c).permute(0,3,1,2).cpu()
ref = xs[7].grad
print("da6 rel err %.3e" % rel_l2(da, ref))
# CPU evaluation of the BN backward from these very tensors
a = yr*aff[0][None,:,None,None] + aff[1][None,:,None,None]
dz = da * (a > 0)
dbeta = dz.sum((0,2,3)); 
print("dbeta (cpu from hip tensors) vs oracle", rel_l2(dbeta, ref_sd["down3.maxpool_conv.1.double_conv.1.bias"].grad))
print("oracle dbeta norm", float(ref_sd["down3.maxpool_conv.1.double_conv.1.bias"].grad.norm()), "sum|dz| per ch mean", float(dz.abs().sum((0,2,3)).mean()))
# run the op-level kernels on the very same device tensors
lib = _lib.lib()
daG = sc[do[i]:do[i]+n*c*h*w]
yG = ws[yo[i]:yo[i]+n*c*h*w]
affG = ws[ao[i]:ao[i]+4*c]
gG = model.down3.maxpool_conv[1].double_conv[1].weight.detach()
dgam, dbet, coef = torch.empty(c, device="cuda"), torch.empty(c, device="cuda"), torch.empty(3*c, device="cuda")
nb = lib.ustrun_bn_bwd_partials_bytes(n*h*w, c)
part = torch.empty(nb//4, device="cuda")
_lib.check(lib.ustrun_bn_bwd_reduce(daG.data_ptr(), None, yG.data_ptr(), affG.data_ptr(), affG.data_ptr()+4*c, affG.data_ptr()+8*c, affG.data_ptr()+12*c,
                                   gG.data_ptr(), n, h, w, c, dgam.data_ptr(), dbet.data_ptr(), 0, coef.data_ptr(), part.data_ptr(), nb, 0, None))
torch.cuda.synchronize()
print("op-level dbeta vs oracle", rel_l2(dbet.cpu(), ref_sd["down3.maxpool_conv.1.double_conv.1.bias"].grad), "vs cpu-from-hip", rel_l2(dbet.cpu(), dbeta))
print("model grad dbeta vs oracle", rel_l2(model.down3.maxpool_conv[1].double_conv[1].bias.grad.cpu() if model.down3.maxpool_conv[1].double_conv[1].bias.grad is not None else dbet.cpu()*0, ref_sd["down3.maxpool_conv.1.double_conv.1.bias"].grad))
d = (dbet.cpu()-dbeta).abs(); print("worst channels", d.topk(5))
