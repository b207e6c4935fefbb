"""CPU restatement of the reference 2D U-Net (test infrastructure, see oracle/__init__.py).

Functional form over a plain ``state_dict`` (same 118 keys as the reference model,
SURVEY.md 8b) so that the HIP path and the oracle can be fed identical weights.
Train-mode BatchNorm is written out explicitly (batch mean / biased variance for the
normalisation, running stats updated with the unbiased variance and momentum 0.1).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def bn_relu(y, prefix, sd, train, relu=True):
    """BatchNorm2d (+ ReLU) as the reference's nn.BatchNorm2d/nn.ReLU pair applies it.

    reference: networks/unet_parts.py:17-18,20-21 (nn.BatchNorm2d defaults eps=1e-5,
    momentum=0.1, affine, track_running_stats).  In train mode the batch statistics
    normalise and the running buffers are updated in place in ``sd``.
    """
    g, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if train:
        n = y.numel() // y.shape[1]
        var, mean = torch.var_mean(y, dim=(0, 2, 3), unbiased=False)
        with torch.no_grad():
            rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
            rm.mul_(1 - BN_MOMENTUM).add_(mean.detach(), alpha=BN_MOMENTUM)
            unb = var.detach() * (n / max(n - 1, 1))
            rv.mul_(1 - BN_MOMENTUM).add_(unb, alpha=BN_MOMENTUM)
            sd[prefix + ".num_batches_tracked"] += 1
    else:
        mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    scale = g * torch.rsqrt(var + BN_EPS)
    shift = b - mean * scale
    z = y * scale[None, :, None, None] + shift[None, :, None, None]
    return torch.relu(z) if relu else z


def double_conv(x, prefix, sd, train):
    """(conv3x3 pad 1 no bias -> BN -> ReLU) twice.  reference: networks/unet_parts.py:8-25."""
    y = F.conv2d(x, sd[prefix + ".double_conv.0.weight"], None, 1, 1)
    a = bn_relu(y, prefix + ".double_conv.1", sd, train)
    y = F.conv2d(a, sd[prefix + ".double_conv.3.weight"], None, 1, 1)
    return bn_relu(y, prefix + ".double_conv.4", sd, train)


def down(x, prefix, sd, train):
    """max_pool2d(2) then DoubleConv.  reference: networks/unet_parts.py:28-39."""
    return double_conv(F.max_pool2d(x, 2), prefix + ".maxpool_conv.1", sd, train)


def up(x1, x2, prefix, sd, train, bilinear=False):
    """Upsample x1, zero-pad to x2's size, cat([x2, x1]) (skip FIRST), DoubleConv.

    reference: networks/unet_parts.py:42-68.  Default path is ConvTranspose2d(k=2,s=2)
    with bias; bilinear path is Upsample(scale 2, align_corners=True).
    """
    if bilinear:
        x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)
    else:
        x1 = F.conv_transpose2d(x1, sd[prefix + ".up.weight"], sd[prefix + ".up.bias"], stride=2)
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    x1 = F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return double_conv(torch.cat([x2, x1], dim=1), prefix + ".conv", sd, train)


def out_conv(x, prefix, sd):
    """1x1 conv with bias.  reference: networks/unet_parts.py:71-76."""
    return F.conv2d(x, sd[prefix + ".conv.weight"], sd[prefix + ".conv.bias"])


def unet_forward(x, sd, train=True, bilinear=False, feature=False):
    """reference: networks/unet_model.py:25-39."""
    x1 = double_conv(x, "inc", sd, train)
    x2 = down(x1, "down1", sd, train)
    x3 = down(x2, "down2", sd, train)
    x4 = down(x3, "down3", sd, train)
    x5 = down(x4, "down4", sd, train)
    y = up(x5, x4, "up1", sd, train, bilinear)
    y = up(y, x3, "up2", sd, train, bilinear)
    y = up(y, x2, "up3", sd, train, bilinear)
    y = up(y, x1, "up4", sd, train, bilinear)
    logits = out_conv(y, "outc", sd)
    return (logits, y) if feature else logits


# ---------------------------------------------------------------------------------
# state_dict construction with PyTorch's default initialisation, in the reference's
# module-construction order (so torch.manual_seed(s) gives the reference's weights).
# ---------------------------------------------------------------------------------

def _conv_init(cout, cin, k, bias):
    conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=bias)
    return conv.weight.detach(), (conv.bias.detach() if bias else None)


def _dc_entries(sd, prefix, cin, cout, mid=None):
    mid = mid or cout
    for idx, (ci, co) in ((0, (cin, mid)), (3, (mid, cout))):
        w, _ = _conv_init(co, ci, 3, False)
        sd[f"{prefix}.double_conv.{idx}.weight"] = w
        bn = f"{prefix}.double_conv.{idx + 1}"
        sd[bn + ".weight"] = torch.ones(co)
        sd[bn + ".bias"] = torch.zeros(co)
        sd[bn + ".running_mean"] = torch.zeros(co)
        sd[bn + ".running_var"] = torch.ones(co)
        sd[bn + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def make_state_dict(n_channels, n_classes, bilinear=False, base=64):
    """Fresh parameters/buffers with the reference's key set.

    ``base`` (=64 in the reference, networks/unet_model.py:13-22) may be lowered in
    tests to get a cheap full-network case; key names are unchanged.
    """
    sd = {}
    c = [base, base * 2, base * 4, base * 8, base * 16]
    factor = 2 if bilinear else 1
    _dc_entries(sd, "inc", n_channels, c[0])
    _dc_entries(sd, "down1.maxpool_conv.1", c[0], c[1])
    _dc_entries(sd, "down2.maxpool_conv.1", c[1], c[2])
    _dc_entries(sd, "down3.maxpool_conv.1", c[2], c[3])
    _dc_entries(sd, "down4.maxpool_conv.1", c[3], c[4] // factor)
    ups = [(c[4], c[3] // factor), (c[3], c[2] // factor), (c[2], c[1] // factor), (c[1], c[0])]
    for i, (cin, cout) in enumerate(ups, start=1):
        if bilinear:
            _dc_entries(sd, f"up{i}.conv", cin, cout, cin // 2)
        else:
            ct = torch.nn.ConvTranspose2d(cin, cin // 2, 2, 2)
            sd[f"up{i}.up.weight"] = ct.weight.detach()
            sd[f"up{i}.up.bias"] = ct.bias.detach()
            _dc_entries(sd, f"up{i}.conv", cin, cout)
    w, b = _conv_init(n_classes, c[0], 1, True)
    sd["outc.conv.weight"], sd["outc.conv.bias"] = w, b
    return sd


def param_keys(sd):
    """Keys that are nn.Parameters in the reference (everything but BN buffers), in order."""
    return [k for k in sd if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]


def clone_sd(sd, requires_grad=False):
    out = {}
    pk = set(param_keys(sd))
    for k, v in sd.items():
        t = v.detach().clone()
        if requires_grad and k in pk:
            t.requires_grad_(True)
        out[k] = t
    return out
