"""DeepLabV2-ResNet forward through the C ABI (reference networks/deeplabv2.py:22-33, networks/backbone/resnet.py:78-105,
159-171).  PyTorch owns memory and the stream; every FLOP runs in libustrun.so.

Internal tensors are NHWC in the compute dtype.  Only raw convolution outputs `y` (pre-BatchNorm) and the block outputs are
materialised: BatchNorm + ReLU of a producer is applied by its consumer's loader, the residual join
relu(bn3(y3) + identity) is one pass (identity = the block input, or bn_d(y_d) of the projection shortcut, evaluated in the
same pass).  Train mode takes batch statistics from the convolution epilogues (f64 finalize, running buffers updated in
place); eval mode uses the running buffers.  Forward only.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .engine import stream_ptr

_DT = {"f32": L.F32, "bf16": L.BF16}


class Act:
    """An NHWC tensor plus how to read it: raw conv output with (scale, shift) + ReLU pending, or a finished activation."""
    __slots__ = ("t", "aff", "relu", "N", "H", "W", "C")

    def __init__(self, t, N, H, W, Cc, aff=None, relu=0):
        self.t, self.N, self.H, self.W, self.C, self.aff, self.relu = t, N, H, W, Cc, aff, relu

    def src(self):
        if self.aff is None:
            return L.nhwc_src(self.t.data_ptr(), self.C, self.H, self.W)
        return L.nhwc_src(self.t.data_ptr(), self.C, self.H, self.W, self.aff[0].data_ptr(), self.aff[1].data_ptr(), relu=self.relu)


def _tdtype(dt):
    return torch.bfloat16 if dt == L.BF16 else torch.float32


def _packed(conv, dt):
    """forward pack of a conv's weight, cached on the module until the weight changes"""
    key = (conv.weight.data_ptr(), conv.weight._version, dt)
    if getattr(conv, "_ustrun_pack_key", None) != key:
        lib = L.lib()
        co, ci, kh, kw = conv.weight.shape
        n = lib.ustrun_pack_conv_elems(co, ci, kh * kw)
        buf = torch.zeros(n, dtype=_tdtype(dt), device=conv.weight.device)
        w = conv.weight.detach().contiguous()
        L.check(lib.ustrun_pack_conv(w.data_ptr(), co, ci, kh * kw, buf.data_ptr(), dt, stream_ptr()), "ustrun_pack_conv")
        conv._ustrun_pack, conv._ustrun_pack_key = buf, key
    return conv._ustrun_pack


def _out_extent(h, k, s, d):
    return (h + 2 * (d * (k // 2)) - d * (k - 1) - 1) // s + 1


def conv_bn(src_act, x_nchw, conv, bn, dt, train):
    """conv (+ BatchNorm constants of its output): returns Act(raw y, aff=[scale, shift, mean, rstd], relu pending)."""
    lib = L.lib()
    co, ci, k, _ = conv.weight.shape
    s, d = conv.stride[0], conv.dilation[0]
    if x_nchw is not None:
        N, _, H, W = x_nchw.shape
        src = L.nchw_src(x_nchw.data_ptr(), ci, H, W)
    else:
        N, H, W = src_act.N, src_act.H, src_act.W
        src = src_act.src()
    Ho, Wo = _out_extent(H, k, s, d), _out_extent(W, k, s, d)
    dev = conv.weight.device
    y = torch.empty(N, Ho, Wo, co, dtype=_tdtype(dt), device=dev)
    aff = torch.empty(4, co, device=dev)
    wf = _packed(conv, dt)
    if train:
        rows = lib.ustrun_conv_mtiles(N, Ho, Wo, co)         # upper bound over every kernel that may serve the launch
        stat = torch.empty(rows, 2, co, device=dev)
        used = C.c_int(0)
        L.check(lib.ustrun_conv2d_fwd(C.byref(src), 1, wf.data_ptr(), None, N, Ho, Wo, co, k, s, d, y.data_ptr(), 0, stat.data_ptr(),
                                      C.byref(used), dt, stream_ptr()), "ustrun_conv2d_fwd")
        L.check(lib.ustrun_bn_finalize(stat.data_ptr(), used.value, co, N * Ho * Wo, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                       bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr(),
                                       float(bn.momentum), float(bn.eps), 1, aff[0].data_ptr(), aff[1].data_ptr(), aff[2].data_ptr(),
                                       aff[3].data_ptr(), stream_ptr()), "ustrun_bn_finalize")
    else:
        L.check(lib.ustrun_conv2d_fwd(C.byref(src), 1, wf.data_ptr(), None, N, Ho, Wo, co, k, s, d, y.data_ptr(), 0, None, None, dt,
                                      stream_ptr()), "ustrun_conv2d_fwd")
        L.check(lib.ustrun_bn_eval_affine(co, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                          bn.running_var.data_ptr(), float(bn.eps), aff[0].data_ptr(), aff[1].data_ptr(), stream_ptr()),
                "ustrun_bn_eval_affine")
    return Act(y, N, Ho, Wo, co, aff=aff, relu=1)


def stem(net, x, dt, train):
    """resnet.py:124-126,160-162: the 7x7 / stride-2 / padding-3 convolution of the 3-channel NCHW input + BatchNorm constants.
    The input is re-laid as zero-padded NHWC in the compute dtype (tensor plumbing), where the 7 pixels x 3 channels of one
    kernel row are 21 contiguous elements: the convolution runs as 7 row segments of a 24-wide window (ustrun_conv_rowwin_fwd)
    instead of 49 taps of 3 channels."""
    lib = L.lib()
    conv, bn = net.conv1, net.bn1
    N, Cin, H, W = x.shape
    k, st, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    win = (k * Cin + 7) // 8 * 8                                                   # 24
    extra = (win - k * Cin + Cin - 1) // Cin                                       # whole pixels the rounded-up window reaches past
    xp = torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, 0, pad, pad + extra, pad, pad)).to(_tdtype(dt)).contiguous()
    Hp, Wp = H + 2 * pad, W + 2 * pad + extra
    key = (conv.weight.data_ptr(), conv.weight._version, dt, "rowwin")
    if getattr(conv, "_ustrun_pack_key", None) != key:
        co = conv.weight.shape[0]
        w = conv.weight.detach().permute(0, 2, 3, 1).reshape(co, k, k * Cin)       # [co][ky][kx*Cin + ci]
        w = torch.nn.functional.pad(w, (0, win - k * Cin)).permute(0, 2, 1).contiguous()     # [co][win][ky]: "Cin" = win, taps = ky
        buf = torch.zeros(lib.ustrun_pack_conv_elems(co, win, k), dtype=_tdtype(dt), device=x.device)
        L.check(lib.ustrun_pack_conv(w.data_ptr(), co, win, k, buf.data_ptr(), dt, stream_ptr()), "ustrun_pack_conv")
        conv._ustrun_pack, conv._ustrun_pack_key = buf, key
    co = conv.weight.shape[0]
    src = L.Src(xp.data_ptr(), None, None, win, Hp, Wp - (win + Cin - 1) // Cin + 1, Hp * Wp * Cin, Wp * Cin, Cin, 1, 0, 0, 0, 0, 0, 0, 0)
    y = torch.empty(N, Ho, Wo, co, dtype=_tdtype(dt), device=x.device)
    aff = torch.empty(4, co, device=x.device)
    stat = torch.empty(lib.ustrun_conv_mtiles(N, Ho, Wo, co), 2, co, device=x.device) if train else None
    used = C.c_int(0)
    L.check(lib.ustrun_conv_rowwin_fwd(C.byref(src), conv._ustrun_pack.data_ptr(), N, Ho, Wo, co, k, st, y.data_ptr(),
                                       stat.data_ptr() if train else None, C.byref(used), dt, stream_ptr()), "ustrun_conv_rowwin_fwd")
    if train:
        L.check(lib.ustrun_bn_finalize(stat.data_ptr(), used.value, co, N * Ho * Wo, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                       bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr(),
                                       float(bn.momentum), float(bn.eps), 1, aff[0].data_ptr(), aff[1].data_ptr(), aff[2].data_ptr(),
                                       aff[3].data_ptr(), stream_ptr()), "ustrun_bn_finalize")
    else:
        L.check(lib.ustrun_bn_eval_affine(co, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                          bn.running_var.data_ptr(), float(bn.eps), aff[0].data_ptr(), aff[1].data_ptr(), stream_ptr()),
                "ustrun_bn_eval_affine")
    return Act(y, N, Ho, Wo, co, aff=aff, relu=1)


def bottleneck(blk, x, dt, train):
    """resnet.py:78-105 on a finished activation x -> the block's finished activation."""
    lib = L.lib()
    y1 = conv_bn(x, None, blk.conv1, blk.bn1, dt, train)
    y2 = conv_bn(y1, None, blk.conv2, blk.bn2, dt, train)
    y3 = conv_bn(y2, None, blk.conv3, blk.bn3, dt, train)
    if blk.downsample is not None:
        yd = conv_bn(x, None, blk.downsample[0], blk.downsample[1], dt, train)
        idn, isc, ish = yd.t, yd.aff[0].data_ptr(), yd.aff[1].data_ptr()
    else:
        idn, isc, ish = x.t, None, None
    out = torch.empty_like(y3.t)
    L.check(lib.ustrun_bn_add_relu(y3.t.data_ptr(), y3.aff[0].data_ptr(), y3.aff[1].data_ptr(), idn.data_ptr(), isc, ish,
                                   y3.N * y3.H * y3.W, y3.C, out.data_ptr(), dt, stream_ptr()), "ustrun_bn_add_relu")
    return Act(out, y3.N, y3.H, y3.W, y3.C)


def _check_input(net, x):
    if not x.is_cuda:
        raise RuntimeError("ResNet / DeepLabV2 run on an MI355X through libustrun.so: there is no CPU fallback for this path")
    if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32:
        raise RuntimeError(f"expected a float32 input [N,3,H,W], got {x.dtype} {tuple(x.shape)}")
    if torch.is_grad_enabled() and net.training and any(p.requires_grad for p in net.parameters()):
        raise NotImplementedError("DeepLabV2-ResNet is forward-only in this build: call it under torch.no_grad() "
                                  "(train-mode BatchNorm statistics are computed and the running buffers updated)")


def backbone_features(net, x):
    """resnet.py:159-171 -> [c1, c2, c3, c4] as finished NHWC activations."""
    _check_input(net, x)
    lib = L.lib()
    dt = _DT[net.compute_dtype]
    train = net.training
    x = x.contiguous()
    y0 = stem(net, x, dt, train)
    Hp, Wp = (y0.H + 1) // 2, (y0.W + 1) // 2
    p = torch.empty(y0.N, Hp, Wp, y0.C, dtype=y0.t.dtype, device=x.device)
    L.check(lib.ustrun_maxpool3x3s2(y0.t.data_ptr(), y0.aff[0].data_ptr(), y0.aff[1].data_ptr(), y0.N, y0.H, y0.W, y0.C, p.data_ptr(),
                                    dt, stream_ptr()), "ustrun_maxpool3x3s2")
    a = Act(p, y0.N, Hp, Wp, y0.C)
    feats = []
    for stage in (net.layer1, net.layer2, net.layer3, net.layer4):
        for blk in stage:
            a = bottleneck(blk, a, dt, train)
        feats.append(a)
    return feats


def to_nchw(a):
    return a.t.float().permute(0, 3, 1, 2).contiguous()


def _classifier_gemm(net, dt):
    """The four dilated classifier convolutions as one 1x1 weight [nrates*9*K, 2048, 1, 1] (row (r*9+tap)*K+k) and the sum of
    their biases; cached until a classifier parameter changes."""
    key = tuple((c.weight.data_ptr(), c.weight._version, c.bias._version) for c in net.classifier) + (dt,)
    if getattr(net, "_ustrun_cls_key", None) != key:
        lib = L.lib()
        K, Cin = net.classifier[0].weight.shape[:2]
        wall = torch.stack([c.weight.detach().reshape(K, Cin, 9).permute(2, 0, 1) for c in net.classifier], 0)    # [r, tap, k, c]
        wall = wall.reshape(len(net.classifier) * 9 * K, Cin).contiguous()
        n = lib.ustrun_pack_conv_elems(wall.shape[0], Cin, 1)
        buf = torch.zeros(n, dtype=_tdtype(dt), device=wall.device)
        L.check(lib.ustrun_pack_conv(wall.data_ptr(), wall.shape[0], Cin, 1, buf.data_ptr(), dt, stream_ptr()), "ustrun_pack_conv")
        net._ustrun_cls = (buf, sum(c.bias.detach() for c in net.classifier).contiguous(), wall.shape[0])
        net._ustrun_cls_key = key
    return net._ustrun_cls


def deeplabv2_forward(net, x):
    """deeplabv2.py:22-33: logits NCHW float32 [N, nclass, H, W].  The four dilated 3x3 branches (Cout = nclass = 2: a
    64-column MFMA tile would be 97 % idle, 2.1-2.6 ms each at N = 8) run as ONE 1x1 GEMM with nrates*9*nclass columns followed
    by a shifted add (a convolution is linear in its taps)."""
    lib = L.lib()
    N, _, H, W = x.shape
    c4 = backbone_features(net.backbone, x)[-1]
    dt = _DT[net.backbone.compute_dtype]
    K = net.classifier[0].weight.shape[0]
    wf, bias_sum, ZC = _classifier_gemm(net, dt)
    src = c4.src()
    z = torch.empty(N, c4.H, c4.W, ZC, device=x.device)
    L.check(lib.ustrun_conv2d_fwd(C.byref(src), 1, wf.data_ptr(), None, N, c4.H, c4.W, ZC, 1, 1, 1, z.data_ptr(), 1, None, None, dt,
                                  stream_ptr()), "ustrun_conv2d_fwd")
    low = torch.empty(N, c4.H, c4.W, K, device=x.device)
    rates = (C.c_int * 4)(*[c.dilation[0] for c in net.classifier])
    L.check(lib.ustrun_aspp_gather(z.data_ptr(), N, c4.H, c4.W, K, len(net.classifier), rates, bias_sum.data_ptr(), low.data_ptr(),
                                   stream_ptr()), "ustrun_aspp_gather")
    out = torch.empty(N, K, H, W, device=x.device)
    arr = (C.c_void_p * 4)(low.data_ptr(), None, None, None)
    L.check(lib.ustrun_sum_resize_bilinear(arr, 1, N, c4.H, c4.W, K, H, W, out.data_ptr(), stream_ptr()), "ustrun_sum_resize_bilinear")
    return out
