"""Loss modules of the hot path -- MI355X build.

DiceLossWithMask keeps the reference's constructor and forward signature
(utils/losses.py:194-268); on HIP tensors it evaluates through ustrun_seg_loss_fwd/_bwd.
The SSL4MIS leftovers of the reference's losses.py (:8-192, :271-295) are never called by any
script and are out of scope.
"""
import torch
import torch.nn as nn


class DiceLossWithMask(nn.Module):
    def __init__(self, n_classes):
        super(DiceLossWithMask, self).__init__()
        self.n_classes = n_classes

    def forward(self, inputs, target, mask=None, weight=None, softmax=False, sigmoid=False, multi=False):
        from ustrun import functional as F
        if sigmoid and softmax:
            assert (0)
        if weight is not None and any(w != 1 for w in weight):
            raise NotImplementedError("class weights other than 1 are not used by the reference's step")
        if not inputs.is_cuda:
            raise RuntimeError("ust-run_amd runs on MI355X (HIP) tensors only; there is no CPU fallback.")
        if softmax and not multi:
            _, dice = F.seg_loss(inputs, target.squeeze(1), mask, "softmax", ce_weight=0.0)
            return dice
        if sigmoid and multi:
            _, dice = F.seg_loss(inputs, target.squeeze(1), mask, "sigmoid", ce_weight=0.0)
            return dice
        raise NotImplementedError("only the two mode combinations the reference's step uses are built: "
                                  "softmax=True (per-class) and sigmoid=True, multi=True (train.py:515-521)")
