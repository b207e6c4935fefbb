"""ResNet backbone -- MI355X build of the reference's networks/backbone/resnet.py:55-212.

Same constructor signatures, attributes and submodule names as the reference, registered in the same order -- hence the same
state_dict keys and, under the same torch seed, the same initial weights (tests/test_abi.py-style check in
tests/test_host_logic.py) -- built from a stage table instead of the reference's hand-written members.  `base_forward` runs on
libustrun.so through ustrun.resnet_engine: every convolution is an implicit GEMM on the matrix cores with the producer's
BatchNorm + ReLU applied on load, the residual join is one fused pass; the backward runs through DeepLabV2.forward
(ustrun.resnet_engine.DeepLabFn), the stand-alone feature path records no autograd graph.  BasicBlock nets (resnet18/34) are on no path of the reference's DeepLabV2 (base.py:12) and are not built.
"""
import torch
import torch.nn as nn

__all__ = ['ResNet', 'Bottleneck', 'resnet50', 'resnet101']


class Bottleneck(nn.Module):
    """resnet.py:55-105 -- 1x1 reduce, 3x3 (stride / dilation), 1x1 expand, each followed by BatchNorm; residual join.
    Holds parameters only: the arithmetic is ustrun.resnet_engine.bottleneck."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1,
                 base_width=64, dilation=1, norm_layer=None):
        super().__init__()
        if groups != 1:
            raise NotImplementedError("grouped convolutions are not on the reference's DeepLabV2 path")
        norm = norm_layer or nn.BatchNorm2d
        width = int(planes * (base_width / 64.)) * groups
        # (cin, cout, kernel, stride, dilation) of conv1..3; registered conv_i, bn_i alternately like the reference
        table = ((inplanes, width, 1, 1, 1), (width, width, 3, stride, dilation), (width, planes * self.expansion, 1, 1, 1))
        for i, (cin, cout, k, s, d) in enumerate(table, 1):
            setattr(self, "conv%d" % i, nn.Conv2d(cin, cout, kernel_size=k, stride=s, padding=d * (k // 2), dilation=d, bias=False))
            setattr(self, "bn%d" % i, norm(cout))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("Bottleneck runs inside ResNet.base_forward (HIP kernels); it has no stand-alone forward")


class ResNet(nn.Module):

    def __init__(self, block, layers, zero_init_residual=False, groups=1,
                 width_per_group=64, replace_stride_with_dilation=None, norm_layer=None, dtype="f32"):
        super().__init__()
        if block is not Bottleneck:
            raise NotImplementedError("only Bottleneck nets (resnet50 / resnet101) are on the reference's DeepLabV2 path")
        dil = [False, False, False] if replace_stride_with_dilation is None else list(replace_stride_with_dilation)
        if len(dil) != 3:
            raise ValueError("replace_stride_with_dilation should be None "
                             "or a 3-element tuple, got {}".format(replace_stride_with_dilation))
        self.compute_dtype = dtype
        self.channels = [w * block.expansion for w in (64, 128, 256, 512)]
        self._norm_layer = norm_layer or nn.BatchNorm2d
        self.groups, self.base_width = groups, width_per_group
        self.inplanes, self.dilation = 64, 1
        # stem (resnet.py:124-127)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = self._norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        # stages (resnet.py:128-134): (width, blocks, stride, trade the stride for dilation)
        stages = [(64, layers[0], 1, False)] + [(w, n, 2, d) for w, n, d in zip((128, 256, 512), layers[1:], dil)]
        for i, spec in enumerate(stages, 1):
            setattr(self, "layer%d" % i, self._stage(block, *spec))
        # initialisation (resnet.py:136-149), in module order: the RNG stream of the reference
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def _stage(self, block, planes, blocks, stride, dilate):
        """resnet.py:151-174: the first block carries the stride (or, dilated, the previous rate) and the projection
        shortcut; the others run at the stage's rate."""
        first_rate = self.dilation
        if dilate:
            self.dilation, stride = self.dilation * stride, 1
        out = planes * block.expansion
        shortcut = None
        if stride != 1 or self.inplanes != out:
            shortcut = nn.Sequential(nn.Conv2d(self.inplanes, out, kernel_size=1, stride=stride, bias=False), self._norm_layer(out))
        seq = [block(self.inplanes, planes, stride, shortcut, self.groups, self.base_width, first_rate, self._norm_layer)]
        self.inplanes = out
        seq += [block(out, planes, groups=self.groups, base_width=self.base_width, dilation=self.dilation,
                      norm_layer=self._norm_layer) for _ in range(blocks - 1)]
        return nn.Sequential(*seq)

    def base_forward(self, x):
        """resnet.py:159-171: (c1, c2, c3, c4), NCHW float32 like the reference's."""
        from ustrun import resnet_engine as E
        return tuple(E.to_nchw(t) for t in E.backbone_features(self, x))

    def forward(self, x):
        return self.base_forward(x)


def _resnet(arch, layers, pretrained, **kwargs):
    model = ResNet(Bottleneck, layers, replace_stride_with_dilation=[False, True, True], **kwargs)
    if pretrained:      # resnet.py:179-181: the reference's checkpoint location; the state_dict keys interchange
        model.load_state_dict(torch.load("../../checkpoints/pretrained/%s.pth" % arch), strict=False)
    return model


def resnet50(pretrained=False, **kw):
    return _resnet('resnet50', [3, 4, 6, 3], pretrained, **kw)


def resnet101(pretrained=False, **kw):
    return _resnet('resnet101', [3, 4, 23, 3], pretrained, **kw)
