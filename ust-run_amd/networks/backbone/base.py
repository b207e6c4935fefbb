"""BaseNet -- MI355X build of the reference's networks/backbone/base.py:8-45.

`forward(x)` is `base_forward(x)` (the subclass's network on libustrun.so).  `forward(x, tta=True)` is the reference's multi-scale
+ mirror test-time augmentation (base.py:24-45): class probabilities of ten views -- five scales, each plain and mirrored --
brought back to the input extent and summed in the reference's order; it is composition of `base_forward` calls with torch
resizes, so it stays on the host side."""
from itertools import product

import torch.nn.functional as F
from torch import nn

from .resnet import resnet50, resnet101

_ZOO = {'resnet50': resnet50, 'resnet101': resnet101}


def _resize(t, size):
    return F.interpolate(t, size=size, mode='bilinear', align_corners=True)


class BaseNet(nn.Module):
    tta_scales = (0.5, 0.75, 1.0, 1.5, 2.0)

    def __init__(self, backbone, pretrained=True, dtype="f32"):
        super(BaseNet, self).__init__()
        self.backbone = _ZOO[backbone](pretrained=pretrained, dtype=dtype)

    def _view_probabilities(self, x, scale, mirrored, size):
        """softmax of one augmented view, un-mirrored, at the input extent (mirror first, then resize, as base.py:38-40)"""
        view = _resize(x, (int(size[0] * scale), int(size[1] * scale)))
        if mirrored:
            view = view.flip(3)
        prob = F.softmax(self.base_forward(view), dim=1)
        return _resize(prob.flip(3) if mirrored else prob, size)

    def forward(self, x, tta=False):
        if not tta:
            return self.base_forward(x)
        size = tuple(x.shape[-2:])
        total = None
        for scale, mirrored in product(self.tta_scales, (False, True)):
            prob = self._view_probabilities(x, scale, mirrored, size)
            total = prob if total is None else total + prob
        return total
