"""Block-level entry points (DoubleConv / Down / Up / OutConv called on their own).

The reference's scripts never call the blocks directly -- only `UNet.forward` does
(train.py:643-702) -- and inside `UNet.forward` this build fuses BatchNorm+ReLU, MaxPool, pad and cat
into the consuming convolution's loads, so a block boundary is not a materialisation point on the hot
path.  Stand-alone block calls would need the activated tensor materialised at every block edge; they
are listed under "next" in DESIGN.md 7 and refuse loudly until built (no silent ATen fallback).
"""


def _refuse(name):
    raise NotImplementedError(
        f"{name} called on its own is not built yet: on MI355X the blocks run fused inside UNet.forward "
        "(ustrun_unet_forward). Use networks.unet_model.UNet, or the op-level C ABI (include/ustrun.h).")


def double_conv(module, x, pool=False):
    _refuse("Down" if pool else "DoubleConv")


def up(module, x1, x2):
    _refuse("Up")


def out_conv(module, x):
    _refuse("OutConv")
