""" Full assembly of the parts to form the complete network -- MI355X build.

Drop-in for the reference's networks/unet_model.py:6-39: same constructor, attributes, submodule
names (inc, down1..4, up1..4, outc), parameter order and forward(x, feature=False) contract.
forward runs the whole network as one call into libustrun.so (ustrun_unet_forward) and registers
one autograd node whose backward is ustrun_unet_backward.
"""
from .unet_parts import *
from .unet_parts import _hip_only


class UNet(nn.Module):
    def __init__(self, n_channels, n_classes, bilinear=False, base_channels=64, dtype="f32", num_domains=0):
        super(UNet, self).__init__()
        self.num_domains = num_domains           # > 0 (additive): every BatchNorm2d becomes a DomainSpecificBatchNorm2d (networks/dsbn.py)
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
        if num_domains:
            from .dsbn import DomainSpecificBatchNorm2d
            for dc in (self.inc, self.down1.maxpool_conv[1], self.down2.maxpool_conv[1], self.down3.maxpool_conv[1],
                       self.down4.maxpool_conv[1], self.up1.conv, self.up2.conv, self.up3.conv, self.up4.conv):
                for k in (1, 4):                 # BatchNorm2d draws nothing from the RNG: the convolutions' initial weights are unchanged
                    bn = dc.double_conv[k]
                    dc.double_conv[k] = DomainSpecificBatchNorm2d(bn.num_features, num_domains, bn.eps, bn.momentum)
            self._ustrun_domain = 0

    def forward(self, x, feature=False, domain_label=None):
        """domain_label (networks with num_domains > 0 only): the batch's domain; its FIRST entry selects the BatchNorm2d of every
        DomainSpecificBatchNorm2d for this call (reference networks/dsbn.py:24-27) -- batch statistics, running buffers and
        gradients all belong to that domain's members."""
        _hip_only(x)
        if self.num_domains:
            if domain_label is None:
                raise RuntimeError("UNet(num_domains > 0): forward needs domain_label")
            self._ustrun_domain = int(domain_label[0]) if hasattr(domain_label, "__getitem__") else int(domain_label)
        elif domain_label is not None:
            raise RuntimeError("UNet: domain_label given to a network without domain-specific BatchNorm (num_domains = 0)")
        from ustrun import engine
        return engine.unet_forward(self, x, feature)

    def _forward_blocks(self, x, feature=False):
        """unet_model.py:25-39 composed from the block modules (each one HIP kernels through the operator-level ABI,
        f32): the parity surface of the blocks, and what the bilinear variant ran on before the fused plan took it
        (round 6: ustrun_unet_desc_t::bilinear, every storage dtype).  Kept for the tests that compare the two."""
        if self.compute_dtype != "f32":
            raise RuntimeError("the block modules run in f32; construct the network with dtype='f32'")
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        y = self.up1(x5, x4)
        y = self.up2(y, x3)
        y = self.up3(y, x2)
        y = self.up4(y, x1)
        logits = self.outc(y)
        return (logits, y) if feature else logits

    def forward_batched(self, x, groups, tail=0, feature=False, lead=0):
        """`forward_passes` for a batch the caller has already laid out end to end: x = `groups` equal passes followed by a
        shorter tail pass of `tail` images (0: none; output discarded, see `forward_passes`).  `lead`: the first `lead` passes
        are forward-only (`with torch.no_grad(): self(x_k)` in front of the others): their logits come back with the rest, the
        backward skips them."""
        import torch
        _hip_only(x)
        n = (len(x) - tail) // max(groups, 1)
        if groups < 1 or n * groups + tail != len(x) or (tail and tail >= n):
            raise RuntimeError(f"forward_batched: {len(x)} images are not {groups} equal passes + a shorter tail of {tail}")
        if lead and feature:
            parts = list(x[:n * groups].split(n))
            with torch.no_grad():           # (with `feature` every pass returns (logits, feat): concatenated per component)
                head = [self(t, feature) for t in parts[:lead]]
            rest = self.forward_passes(parts[lead:], feature, tail=x[n * groups:] if tail else None)
            if not lead:
                return rest
            if feature:
                return tuple(torch.cat([h[k] for h in head] + [rest[k]], 0) for k in range(2))
            return torch.cat(head + [rest], 0)
        from ustrun import engine
        return engine.unet_forward(self, x, feature, groups=groups, tail=tail, lead=lead)

    def forward_passes(self, xs, feature=False, tail=None):
        """Additive API (not in the reference): run several forward passes of equal shape as ONE batched call.
        Equivalent to `[self(x) for x in xs]` -- BatchNorm batch statistics and running-buffer updates stay per
        pass, in order -- but every kernel sees the concatenated batch (better GPU fill on the deep, small layers,
        a third of the launches, one weight-gradient reduction for all passes).  Returns the logits of the
        concatenated batch; split them with `.split(len(xs[0]))`.
        `tail`: a shorter batch that goes through the network as one more pass AFTER `xs` and whose output is discarded --
        `self(tail)` for its side effect on the BatchNorm running statistics only (the reference's low-quality-sample
        forward, train.py:740); it gets no gradient."""
        import torch
        xs = list(xs)
        if len({tuple(x.shape) for x in xs}) != 1:
            raise RuntimeError("forward_passes: all passes must have the same shape")
        if tail is not None and (tuple(tail.shape[1:]) != tuple(xs[0].shape[1:]) or not 0 < len(tail) < len(xs[0])):
            raise RuntimeError("forward_passes: the tail pass must have the passes' image shape and fewer images")
        x = torch.cat(xs + ([tail] if tail is not None else []), 0) if len(xs) > 1 or tail is not None else xs[0]
        _hip_only(x)
        from ustrun import engine
        return engine.unet_forward(self, x, feature, groups=len(xs), tail=0 if tail is None else len(tail))
