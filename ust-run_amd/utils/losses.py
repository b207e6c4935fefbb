"""Loss modules of the hot path -- MI355X build.

DiceLossWithMask keeps the reference's constructor and forward signature
(utils/losses.py:194-268); on HIP tensors it evaluates through ustrun_seg_loss_fwd/_bwd (the two
mode combinations the step uses, fused with their CE / BCE pass) or ustrun_dice_fwd/_bwd (every
other combination: class weights, sigmoid per class, softmax + multi, raw inputs).
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
        if not inputs.is_cuda:
            raise RuntimeError("ust-run_amd runs on MI355X (HIP) tensors only; there is no CPU fallback.")
        unit = weight is None or all(w == 1 for w in weight)
        # the two combinations the reference's step evaluates (train.py:515-521,817,830-836): fused with their CE / BCE pass
        if softmax and not multi and unit:
            _, dice = F.seg_loss(inputs, target.squeeze(1), mask, "softmax", ce_weight=0.0)
            return dice
        if sigmoid and multi:
            _, dice = F.seg_loss(inputs, target.squeeze(1), mask, "sigmoid", ce_weight=0.0)
            return dice
        # every other combination of the reference signature (losses.py:236-268): ustrun_dice_fwd/_bwd
        if sigmoid:
            target = target.squeeze(1)                                   # losses.py:241
        act = "sigmoid" if sigmoid else ("softmax" if softmax else "none")
        if not multi:
            # losses.py:250-253: the one-hot of `target` (concatenated over dim 1) must have the shape of `inputs`
            if target.dim() != inputs.dim() or target.shape[1] != 1 or target.shape[0] != inputs.shape[0] or \
                    tuple(target.shape[2:]) != tuple(inputs.shape[2:]):
                raise AssertionError("predict & target shape do not match")
            if weight is not None and len(weight) != self.n_classes:
                raise IndexError("weight needs one entry per class")
            if inputs.shape[1] != self.n_classes:
                raise AssertionError("predict & target shape do not match")
        return F.dice_general(inputs, target, mask, act=act, multi=multi, weight=None if multi else weight)
