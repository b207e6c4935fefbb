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
    y0 = conv_bn(None, x, net.conv1, net.bn1, dt, train)
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


def deeplabv2_forward(net, x):
    """deeplabv2.py:22-33: logits NCHW float32 [N, nclass, H, W]."""
    lib = L.lib()
    N, _, H, W = x.shape
    c4 = backbone_features(net.backbone, x)[-1]
    dt = _DT[net.backbone.compute_dtype]
    K = net.classifier[0].weight.shape[0]
    maps = []
    src = c4.src()
    for conv in net.classifier:
        m = torch.empty(N, c4.H, c4.W, K, device=x.device)
        L.check(lib.ustrun_conv2d_fwd(C.byref(src), 1, _packed(conv, dt).data_ptr(), conv.bias.data_ptr(), N, c4.H, c4.W, K, 3, 1,
                                      conv.dilation[0], m.data_ptr(), 1, None, None, dt, stream_ptr()), "ustrun_conv2d_fwd")
        maps.append(m)
    out = torch.empty(N, K, H, W, device=x.device)
    arr = (C.c_void_p * 4)(*[m.data_ptr() for m in maps])
    L.check(lib.ustrun_sum_resize_bilinear(arr, 4, N, c4.H, c4.W, K, H, W, out.data_ptr(), stream_ptr()), "ustrun_sum_resize_bilinear")
    return out
