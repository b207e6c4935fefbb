"""CPU restatement of the reference's DeepLabV2-ResNet forward (test infrastructure, see oracle/__init__.py).

Functional form over a plain ``state_dict`` with the reference's keys (``backbone.conv1.weight`` ...
``classifier.3.bias``), pinned by tests/golden/g10_deeplabv2_*.npz which tools/gen_goldens.py captures from the
reference modules themselves.  Follows networks/deeplabv2.py:22-33, networks/backbone/resnet.py:78-105,159-171;
BatchNorm as oracle/unet_ref.bn_relu (nn.BatchNorm2d defaults).  Forward only, like the product path this round.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .unet_ref import bn_relu

ARCH = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}


def _bottleneck(x, p, sd, train, stride, dilation):
    """resnet.py:78-105: 1x1 -> BN/ReLU -> 3x3(stride, dilation) -> BN/ReLU -> 1x1 -> BN; + identity (projected when a
    `downsample` exists); ReLU."""
    out = bn_relu(F.conv2d(x, sd[p + ".conv1.weight"]), p + ".bn1", sd, train)
    out = bn_relu(F.conv2d(out, sd[p + ".conv2.weight"], None, stride, dilation, dilation), p + ".bn2", sd, train)
    out = bn_relu(F.conv2d(out, sd[p + ".conv3.weight"]), p + ".bn3", sd, train, relu=False)
    idn = x
    if p + ".downsample.0.weight" in sd:
        idn = bn_relu(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride), p + ".downsample.1", sd, train, relu=False)
    return torch.relu(out + idn)


def backbone_features(x, sd, arch, train, prefix="backbone."):
    """resnet.py:159-171 with replace_stride_with_dilation=[False, True, True] (resnet.py:193-200): output stride 8."""
    y = F.conv2d(x, sd[prefix + "conv1.weight"], None, 2, 3)
    a = F.max_pool2d(bn_relu(y, prefix + "bn1", sd, train), 3, 2, 1)
    feats = []
    rate = 1
    for li, (nblk, stride, dilate) in enumerate(zip(ARCH[arch], (1, 2, 2, 2), (False, False, True, True)), 1):
        first_rate = rate                      # resnet.py:154-158: the first block keeps the previous rate
        if dilate:
            rate, stride = rate * stride, 1
        for b in range(nblk):
            a = _bottleneck(a, f"{prefix}layer{li}.{b}", sd, train, stride if b == 0 else 1, first_rate if b == 0 else rate)
        feats.append(a)
    return feats


def deeplabv2_forward(x, sd, arch, train):
    """deeplabv2.py:22-33: sum of the four dilated classifier convolutions on c4, bilinear (align_corners) to the input size."""
    h, w = x.shape[-2:]
    c4 = backbone_features(x, sd, arch, train)[-1]
    out = None
    for i, d in enumerate((6, 12, 18, 24)):
        o = F.conv2d(c4, sd[f"classifier.{i}.weight"], sd[f"classifier.{i}.bias"], 1, d, d)
        out = o if out is None else out + o
    return F.interpolate(out, size=(h, w), mode="bilinear", align_corners=True)


def make_state_dict(arch, nclass, seed):
    """The reference's initial weights for this seed, through the product's parameter-owning mirror (checked key by key and
    value by value against the reference's own construction in tests/test_oracle_golden.py via the golden's weight sums)."""
    from networks.deeplabv2 import DeepLabV2
    torch.manual_seed(seed)
    m = DeepLabV2(arch, nclass, pretrained=False)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}
