"""DeepLabV2 -- MI355X build of the reference's networks/deeplabv2.py:10-33 (BASELINE.json configs[4]).

Same constructor surface (`DeepLabV2(backbone, nclass)`), the same `backbone` / `classifier` members and state_dict keys;
`base_forward` = dilated ResNet features -> four dilated 3x3 classifier convolutions (rates 6, 12, 18, 24, with bias) summed
-> bilinear resize (align_corners=True) to the input extent, all in libustrun.so.  `pretrained=False` and `dtype` are
additive keywords (the reference always loads ../../checkpoints/pretrained/<arch>.pth, base.py:12).  Under autograd (train mode)
the call is differentiable with respect to the parameters: ustrun.resnet_engine.DeepLabFn runs the backward in libustrun.so.
"""
from networks.backbone.base import BaseNet

from torch import nn


class DeepLabV2(BaseNet):
    def __init__(self, backbone, nclass, pretrained=True, dtype="f32"):
        super(DeepLabV2, self).__init__(backbone, pretrained=pretrained, dtype=dtype)
        self.compute_dtype = dtype
        self.classifier = nn.ModuleList()
        for dilation in [6, 12, 18, 24]:
            self.classifier.append(
                nn.Conv2d(2048, nclass, kernel_size=3, stride=1, padding=dilation, dilation=dilation, bias=True))
        for m in self.classifier:
            m.weight.data.normal_(0, 0.01)

    def base_forward(self, x):
        from ustrun import resnet_engine as E
        return E.deeplabv2_apply(self, x)
