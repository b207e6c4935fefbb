""" Parts of the U-Net model -- MI355X build.

Same class names, constructor signatures, submodule layout and state_dict keys as the reference
(networks/unet_parts.py:8-76): parameters live in real nn.Conv2d / nn.BatchNorm2d /
nn.ConvTranspose2d members, so reference checkpoints load unchanged and the default
initialisation consumes the torch RNG in the same order.  The arithmetic does NOT go through
these modules' ATen forward: UNet.forward hands the parameter pointers to libustrun.so (HIP,
gfx950), where BatchNorm+ReLU, MaxPool2d, F.pad and torch.cat are folded into the consuming
convolution's loads.  Called on their own, the blocks run through the same C ABI
(ustrun.blocks), on MI355X only; there is no CPU path.
"""
import torch
import torch.nn as nn


def _hip_only(x):
    if not x.is_cuda:
        raise RuntimeError("ust-run_amd runs on MI355X (HIP) tensors only; there is no CPU fallback. "
                           "Move the module and its inputs to the GPU.")


class DoubleConv(nn.Module):
    """(convolution => [BN] => ReLU) * 2"""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        if not mid_channels:
            mid_channels = out_channels
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True)
        )

    def forward(self, x):
        _hip_only(x)
        from ustrun import blocks
        return blocks.double_conv(self, x, pool=False)


class Down(nn.Module):
    """Downscaling with maxpool then double conv"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(
            nn.MaxPool2d(2),
            DoubleConv(in_channels, out_channels)
        )

    def forward(self, x):
        _hip_only(x)
        from ustrun import blocks
        return blocks.double_conv(self.maxpool_conv[1], x, pool=True)


class Up(nn.Module):
    """Upscaling then double conv"""

    def __init__(self, in_channels, out_channels, bilinear=True):
        super().__init__()
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2):
        _hip_only(x1)
        from ustrun import blocks
        return blocks.up(self, x1, x2)


class OutConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super(OutConv, self).__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        _hip_only(x)
        from ustrun import blocks
        return blocks.out_conv(self, x)
