"""Device-side low-frequency amplitude mix (ustrun_freq_mix)."""
import math

import torch

from . import _lib as L
from .engine import stream_ptr


def freq_mix_device(src, trg, LB, ratios, h2d=None):
    """src/trg: normalised [n,C,H,W] HIP tensors; ratios: n python floats drawn by the caller."""
    lib = L.lib()
    src, trg = src.contiguous(), trg.contiguous()
    n, C, H, W = src.shape
    b = int(math.floor(min(H, W) * LB))
    if h2d is not None:
        r = h2d(ratios)
    else:
        r = torch.tensor(ratios, dtype=torch.float32).to(src.device, non_blocking=True)
    nb = lib.ustrun_freq_mix_work_bytes(n, C, b)
    work = torch.empty(nb, dtype=torch.uint8, device=src.device)
    out = torch.empty_like(src)
    L.check(lib.ustrun_freq_mix(src.data_ptr(), trg.data_ptr(), r.data_ptr(), n, C, H, W, b, out.data_ptr(),
                                work.data_ptr(), nb, stream_ptr()), "ustrun_freq_mix")
    return out
