""" Full assembly of the parts to form the complete network -- MI355X build.

Drop-in for the reference's networks/unet_model.py:6-39: same constructor, attributes, submodule
names (inc, down1..4, up1..4, outc), parameter order and forward(x, feature=False) contract.
forward runs the whole network as one call into libustrun.so (ustrun_unet_forward) and registers
one autograd node whose backward is ustrun_unet_backward.
"""
from .unet_parts import *
from .unet_parts import _hip_only


class UNet(nn.Module):
    def __init__(self, n_channels, n_classes, bilinear=False, base_channels=64, dtype="f32"):
        super(UNet, self).__init__()
        self.n_channels = n_channels
        self.n_classes = n_classes
        self.bilinear = bilinear
        self.base_channels = base_channels       # 64 in the reference; smaller only for cheap tests
        self.compute_dtype = dtype
        b = base_channels

        self.inc = DoubleConv(n_channels, b)
        self.down1 = Down(b, 2 * b)
        self.down2 = Down(2 * b, 4 * b)
        self.down3 = Down(4 * b, 8 * b)
        factor = 2 if bilinear else 1
        self.down4 = Down(8 * b, 16 * b // factor)
        self.up1 = Up(16 * b, 8 * b // factor, bilinear)
        self.up2 = Up(8 * b, 4 * b // factor, bilinear)
        self.up3 = Up(4 * b, 2 * b // factor, bilinear)
        self.up4 = Up(2 * b, b, bilinear)
        self.outc = OutConv(b, n_classes)

    def forward(self, x, feature=False):
        _hip_only(x)
        if self.bilinear:
            raise NotImplementedError("bilinear=True is not on the reference's hot path (every call site uses "
                                      "the ConvTranspose2d default, train.py:499); not built in the HIP path yet")
        from ustrun import engine
        return engine.unet_forward(self, x, feature)

    def forward_passes(self, xs, feature=False):
        """Additive API (not in the reference): run several forward passes of equal shape as ONE batched call.
        Equivalent to `[self(x) for x in xs]` -- BatchNorm batch statistics and running-buffer updates stay per
        pass, in order -- but every kernel sees the concatenated batch (better GPU fill on the deep, small layers,
        a third of the launches, one weight-gradient reduction for all passes).  Returns the logits of the
        concatenated batch; split them with `.split(len(xs[0]))`."""
        import torch
        xs = list(xs)
        if len({tuple(x.shape) for x in xs}) != 1:
            raise RuntimeError("forward_passes: all passes must have the same shape")
        x = torch.cat(xs, 0) if len(xs) > 1 else xs[0]
        _hip_only(x)
        if self.bilinear:
            raise NotImplementedError("bilinear=True is not built in the HIP path yet")
        from ustrun import engine
        return engine.unet_forward(self, x, feature, groups=len(xs))
