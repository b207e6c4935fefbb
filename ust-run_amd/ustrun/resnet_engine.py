"""DeepLabV2-ResNet forward through the C ABI (reference networks/deeplabv2.py:22-33, networks/backbone/resnet.py:78-105,
159-171).  PyTorch owns memory and the stream; every FLOP runs in libustrun.so.

Internal tensors are NHWC in the compute dtype.  Only raw convolution outputs `y` (pre-BatchNorm) and the block outputs are
materialised: BatchNorm + ReLU of a producer is applied by its consumer's loader, the residual join
relu(bn3(y3) + identity) is one pass (identity = the block input, or bn_d(y_d) of the projection shortcut, evaluated in the
same pass).  Train mode takes batch statistics from the convolution epilogues (f64 finalize, running buffers updated in
place); eval mode uses the running buffers.

Backward (train mode): `deeplabv2_forward(..., tape=[])` records what each operator kept, `deeplabv2_backward` walks the tape
in reverse -- BatchNorm backward as reduce + apply (ustrun_bn_bwd_*), weight gradients on the TN GEMMs (ustrun_conv2d_wgrad),
input gradients as convolutions of dy with the flipped / transposed weights through ustrun_conv2d_fwd (the two stride-2
convolutions through a zero-inserted dy), the join / max-pool / resize / shifted-add adjoints of resnet_ops.hip.  `DeepLabFn`
is the torch.autograd.Function through which `DeepLabV2.forward` is differentiable with respect to its parameters (the input
image receives no gradient, as in the reference's training loops).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from .engine import stream_ptr

_DT = {"f32": L.F32, "bf16": L.BF16, "f16": L.F16}


class Act:
    """An NHWC tensor plus how to read it: raw conv output with (scale, shift) + ReLU pending, or a finished activation."""
    __slots__ = ("t", "aff", "relu", "N", "H", "W", "C", "keep")

    def __init__(self, t, N, H, W, Cc, aff=None, relu=0):
        self.t, self.N, self.H, self.W, self.C, self.aff, self.relu = t, N, H, W, Cc, aff, relu
        self.keep = None

    def src(self):
        if self.aff is None:
            return L.nhwc_src(self.t.data_ptr(), self.C, self.H, self.W)
        return L.nhwc_src(self.t.data_ptr(), self.C, self.H, self.W, self.aff[0].data_ptr(), self.aff[1].data_ptr(), relu=self.relu)


def _tdtype(dt):
    return torch.bfloat16 if dt == L.BF16 else torch.float16 if dt == L.F16 else torch.float32


_GEN = [0]
# A/B switch (tools/bench_deeplab.py): weight gradients read BatchNorm + ReLU sources through a materialised activation
_MATERIALISE_WGRAD_OPERAND = os.environ.get("USTRUN_DEEPLAB_WGRAD_ONLOAD", "0") != "1"
_SPACE_TO_BATCH_WGRAD = os.environ.get("USTRUN_DEEPLAB_WGRAD_TAPS", "0") != "1"      # "1": dilated 3x3 weight gradients tap by tap (rounds 2-5)


def invalidate_packs():
    """Parameters were rewritten behind autograd's back (the fused SGD + EMA kernel updates the flat buffer in place and leaves
    `weight._version` alone): every cached weight pack is stale.  ustrun.engine.invalidate_packed calls this."""
    _GEN[0] += 1


def _packed(conv, dt):
    """forward pack of a conv's weight, cached on the module until the weight changes"""
    key = (conv.weight.data_ptr(), conv.weight._version, dt, _GEN[0])
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
    kernel row are 21 contiguous elements; `ustrun_rowwin_patches` writes, per output pixel, its 7 row windows of 24 elements as
    one GEMM row of 192 columns (168 + 24 zeros), and the convolution is ONE 1x1 GEMM over those "channels" on the fast kernel
    (0.53 ms as seven 24-wide segments on the generic kernel -> patches + GEMM; the weight gradient reuses the patches:
    1.2 ms -> a plain TN GEMM).  Returns Act(raw y, BatchNorm constants) with the patches kept for the backward."""
    lib = L.lib()
    conv, bn = net.conv1, net.bn1
    N, Cin, H, W = x.shape
    k, st, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
    Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    win = (k * Cin + 7) // 8 * 8                                                   # 24
    Kp = (k * win + 63) // 64 * 64                                                 # 192: whole 64-column tiles for the GEMM kernels
    extra = (win - k * Cin + Cin - 1) // Cin                                       # whole pixels the rounded-up window reaches past
    xp = torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, 0, pad, pad + extra, pad, pad)).to(_tdtype(dt)).contiguous()
    Hp, Wp = H + 2 * pad, W + 2 * pad + extra
    co = conv.weight.shape[0]
    key = (conv.weight.data_ptr(), conv.weight._version, dt, "patches", _GEN[0])
    if getattr(conv, "_ustrun_pack_key", None) != key:
        w = conv.weight.detach().permute(0, 2, 3, 1).reshape(co, k, k * Cin)       # [co][ky][kx*Cin + ci]
        w = torch.nn.functional.pad(w, (0, win - k * Cin)).reshape(co, k * win)    # [co][ky*win + kx*Cin + ci]
        w = torch.nn.functional.pad(w, (0, Kp - k * win)).contiguous()             # zero weights meet the windows' overhang
        buf = torch.zeros(lib.ustrun_pack_conv_elems(co, Kp, 1), dtype=_tdtype(dt), device=x.device)
        L.check(lib.ustrun_pack_conv(w.data_ptr(), co, Kp, 1, buf.data_ptr(), dt, stream_ptr()), "ustrun_pack_conv")
        conv._ustrun_pack, conv._ustrun_pack_key = buf, key
    wsrc = L.Src(xp.data_ptr(), None, None, win, Hp, Wp - (win + Cin - 1) // Cin + 1, Hp * Wp * Cin, Wp * Cin, Cin, 1, 0, 0, 0, 0, 0, 0, 0)
    patches = torch.empty(N, Ho, Wo, Kp, dtype=_tdtype(dt), device=x.device)
    L.check(lib.ustrun_rowwin_patches(C.byref(wsrc), N, Ho, Wo, k, st, Kp, patches.data_ptr(), dt, stream_ptr()), "ustrun_rowwin_patches")
    pa = Act(patches, N, Ho, Wo, Kp)
    src = pa.src()
    y = torch.empty(N, Ho, Wo, co, dtype=_tdtype(dt), device=x.device)
    aff = torch.empty(4, co, device=x.device)
    stat = torch.empty(lib.ustrun_conv_mtiles(N, Ho, Wo, co), 2, co, device=x.device) if train else None
    used = C.c_int(0)
    L.check(lib.ustrun_conv2d_fwd(C.byref(src), 1, conv._ustrun_pack.data_ptr(), None, N, Ho, Wo, co, 1, 1, 1, y.data_ptr(), 0,
                                  stat.data_ptr() if train else None, C.byref(used), dt, stream_ptr()), "ustrun_conv2d_fwd")
    if train:
        L.check(lib.ustrun_bn_finalize(stat.data_ptr(), used.value, co, N * Ho * Wo, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                       bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr(),
                                       float(bn.momentum), float(bn.eps), 1, aff[0].data_ptr(), aff[1].data_ptr(), aff[2].data_ptr(),
                                       aff[3].data_ptr(), stream_ptr()), "ustrun_bn_finalize")
    else:
        L.check(lib.ustrun_bn_eval_affine(co, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                          bn.running_var.data_ptr(), float(bn.eps), aff[0].data_ptr(), aff[1].data_ptr(), stream_ptr()),
                "ustrun_bn_eval_affine")
    a = Act(y, N, Ho, Wo, co, aff=aff, relu=1)
    a.keep = (pa, win, k, Cin)                        # the weight gradient is a TN GEMM over the same patches
    return a


def bottleneck(blk, x, dt, train, tape=None):
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
    o = Act(out, y3.N, y3.H, y3.W, y3.C)
    if tape is not None:
        tape.append(("block", blk, x, y1, y2, y3, yd if blk.downsample is not None else None, o))
    return o


def _check_input(net, x, differentiable=False):
    if not x.is_cuda:
        raise RuntimeError("ResNet / DeepLabV2 run on an MI355X through libustrun.so: there is no CPU fallback for this path")
    if x.dim() != 4 or x.shape[1] != 3 or x.dtype != torch.float32:
        raise RuntimeError(f"expected a float32 input [N,3,H,W], got {x.dtype} {tuple(x.shape)}")
    if not differentiable and torch.is_grad_enabled() and net.training and any(p.requires_grad for p in net.parameters()):
        raise NotImplementedError("the stand-alone ResNet feature path records no autograd graph: call it under torch.no_grad(), "
                                  "or train through DeepLabV2.forward (ustrun.resnet_engine.DeepLabFn)")


def backbone_features(net, x, tape=None):
    """resnet.py:159-171 -> [c1, c2, c3, c4] as finished NHWC activations."""
    _check_input(net, x, differentiable=tape is not None)
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
    if tape is not None:
        tape.append(("stem", net, y0, a))
    feats = []
    for stage in (net.layer1, net.layer2, net.layer3, net.layer4):
        for blk in stage:
            a = bottleneck(blk, a, dt, train, tape)
        feats.append(a)
    return feats


def to_nchw(a):
    return a.t.float().permute(0, 3, 1, 2).contiguous()


def _classifier_gemm(net, dt):
    """The four dilated classifier convolutions as one 1x1 weight [nrates*9*K, 2048, 1, 1] (row (r*9+tap)*K+k) and the sum of
    their biases; cached until a classifier parameter changes."""
    key = tuple((c.weight.data_ptr(), c.weight._version, c.bias._version) for c in net.classifier) + (dt, _GEN[0])
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


def deeplabv2_forward(net, x, tape=None):
    """deeplabv2.py:22-33: logits NCHW float32 [N, nclass, H, W].  The four dilated 3x3 branches (Cout = nclass = 2: a
    64-column MFMA tile would be 97 % idle, 2.1-2.6 ms each at N = 8) run as ONE 1x1 GEMM with nrates*9*nclass columns followed
    by a shifted add (a convolution is linear in its taps)."""
    lib = L.lib()
    N, _, H, W = x.shape
    c4 = backbone_features(net.backbone, x, tape)[-1]
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
    if tape is not None:
        tape.append(("head", net, c4, (N, K, H, W)))
    return out


# ---- backward ---------------------------------------------------------------------------------------------------------------
class _Scratch:
    """Grow-only device scratch shared by the backward's launches.  Reuse is ordered by the stream, so the buffers are keyed
    by (name, device, current stream): a backward on another stream or device gets its own set instead of aliasing."""
    def __init__(self):
        self.buf, self.const = {}, {}

    def get(self, name, nbytes, dev):
        key = (name, dev, torch.cuda.current_stream(dev).cuda_stream)
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            b = self.buf[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        return b

    def noact(self, Cc, dev):
        """(scale, shift) = (0, 1): the BatchNorm-backward kernels' ReLU mask (y*scale+shift > 0) becomes all ones"""
        k = (Cc, dev)
        if k not in self.const:
            self.const[k] = (torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev))
        return self.const[k]


_scratch = _Scratch()


def _packed_dgrad(conv, dt):
    """pack of the input-gradient convolution's weight: w.flip(2, 3).transpose(0, 1) as a conv weight [Cin][Cout][k][k]"""
    key = (conv.weight.data_ptr(), conv.weight._version, dt, _GEN[0])
    if getattr(conv, "_ustrun_dpack_key", None) != key:
        lib = L.lib()
        co, ci, kh, kw = conv.weight.shape
        wd = conv.weight.detach().flip(2, 3).transpose(0, 1).contiguous()
        buf = torch.zeros(lib.ustrun_pack_conv_elems(ci, co, kh * kw), dtype=_tdtype(dt), device=wd.device)
        L.check(lib.ustrun_pack_conv(wd.data_ptr(), ci, co, kh * kw, buf.data_ptr(), dt, stream_ptr()), "ustrun_pack_conv")
        conv._ustrun_dpack, conv._ustrun_dpack_key = buf, key
    return conv._ustrun_dpack


def _bn_backward(y, da, relu, bn, grads, dt):
    """autograd of train-mode BatchNorm (+ReLU): da = gradient of the (activated) output, y = the raw convolution output with its
    forward constants -> dy; fills grads[bn.weight], grads[bn.bias]."""
    lib = L.lib()
    dev = y.t.device
    sc, sh = (y.aff[0], y.aff[1]) if relu else _scratch.noact(y.C, dev)
    npix = y.N * y.H * y.W
    pb = lib.ustrun_bn_bwd_partials_bytes(npix, y.C)
    part = _scratch.get("bn", pb, dev)
    coef = torch.empty(3, y.C, device=dev)
    dg, db = torch.empty(y.C, device=dev), torch.empty(y.C, device=dev)
    L.check(lib.ustrun_bn_bwd_reduce(da.data_ptr(), None, y.t.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.aff[2].data_ptr(),
                                     y.aff[3].data_ptr(), bn.weight.data_ptr(), y.N, y.H, y.W, y.C, dg.data_ptr(), db.data_ptr(), 0,
                                     coef.data_ptr(), part.data_ptr(), pb, dt, stream_ptr()), "ustrun_bn_bwd_reduce")
    dy = torch.empty_like(y.t)
    L.check(lib.ustrun_bn_bwd_apply(da.data_ptr(), None, y.t.data_ptr(), sc.data_ptr(), sh.data_ptr(), coef.data_ptr(), y.N, y.H, y.W,
                                    y.C, dy.data_ptr(), dt, stream_ptr()), "ustrun_bn_bwd_apply")
    grads[bn.weight], grads[bn.bias] = dg, db
    return dy


_SIDE = {}          # device -> the weight gradients' stream
_WGRAD_SIDE_STREAM = os.environ.get("USTRUN_DEEPLAB_WGRAD_STREAM", "1") != "0"       # "0": everything on the caller's stream (rounds 2-5; A/B runs)


def _conv_wgrad(x, dy, y, conv, grads, dt):
    """grads[conv.weight], on a stream of its own (round 6): a layer's weight gradient and its input gradient both only READ dy, and
    nothing waits for dW before the backward returns -- the main chain's HBM-bound passes (BatchNorm apply, the join's GEMM) and small
    launches (finalizes, reduces) leave matrix pipes idle that the weight-gradient kernels use: 38.9-39.3 -> 35.9-36.0 ms per forward +
    backward at n = 16, same box (profiles/r06_ab_deeplab_wgrad_stream.log).  The side stream waits for dy's producer;
    deeplabv2_backward joins it before it returns."""
    if not _WGRAD_SIDE_STREAM:
        return _conv_wgrad_here(x, dy, y, conv, grads, dt)
    dev = dy.device
    main = torch.cuda.current_stream(dev)
    side = _SIDE.get(dev)
    if side is None:
        side = _SIDE[dev] = torch.cuda.Stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        _conv_wgrad_here(x, dy, y, conv, grads, dt)
    for t in (dy, x.t) + ((x.aff[0], x.aff[1]) if x.aff is not None else ()):
        t.record_stream(side)           # (the caching allocator must not hand these to the main stream while the side stream reads them)


def _join_side_stream(dev):
    if _WGRAD_SIDE_STREAM and dev in _SIDE:
        torch.cuda.current_stream(dev).wait_stream(_SIDE[dev])


def _conv_wgrad_here(x, dy, y, conv, grads, dt):
    """grads[conv.weight] from the loader's view of the input Act x and the raw-output gradient dy (shape of y)"""
    lib = L.lib()
    co, ci, k, _ = conv.weight.shape
    s, d = conv.stride[0], conv.dilation[0]
    pb = lib.ustrun_wgrad_partials_bytes(k * k, ci, co, y.N * y.H * y.W)
    part = _scratch.get("wgrad", pb, dy.device)
    dw = torch.empty_like(conv.weight)
    src = x.src()
    if k == 3 and d > 1 and s == 1 and dt != L.F32 and _SPACE_TO_BATCH_WGRAD and ci % 64 == 0 and co % 64 == 0:
        # dilated 3x3: both operands re-laid as d x d sub-grid images (zero-padded to one extent; the activation pass does the
        # re-laying, so only dy costs a pass of its own) -> an ORDINARY 3x3 weight gradient over N d d images on the all-taps kernel
        Hs, Ws = -(-x.H // d), -(-x.W // d)
        n2 = x.N * d * d
        xs = _scratch.get("s2b_x", n2 * Hs * Ws * ci * 2, dy.device)
        ds = _scratch.get("s2b_dy", n2 * Hs * Ws * co * 2, dy.device)
        L.check(lib.ustrun_space_to_batch(C.byref(src), x.N, d, xs.data_ptr(), dt, stream_ptr()), "ustrun_space_to_batch")
        dsrc = L.nhwc_src(dy.data_ptr(), co, y.H, y.W)
        L.check(lib.ustrun_space_to_batch(C.byref(dsrc), y.N, d, ds.data_ptr(), dt, stream_ptr()), "ustrun_space_to_batch")
        pb = lib.ustrun_wgrad_partials_bytes(9, ci, co, n2 * Hs * Ws)
        part = _scratch.get("wgrad", pb, dy.device)
        src = L.nhwc_src(xs.data_ptr(), ci, Hs, Ws)
        L.check(lib.ustrun_conv2d_wgrad(C.byref(src), 1, ds.data_ptr(), n2, Hs, Ws, co, 3, 1, 1, dw.data_ptr(), 0, part.data_ptr(), pb, dt,
                                        stream_ptr()), "ustrun_conv2d_wgrad")
        grads[conv.weight] = dw
        return
    if x.aff is not None and dt != L.F32 and _MATERIALISE_WGRAD_OPERAND:
        # the operand relu(bn(x)) written out once (4 B per element at the HBM rate) instead of being formed per staged item in
        # every one of the Cout / 128 column tiles and k * k taps of the weight gradient (2-18 times over: round 6)
        act = _scratch.get("wgrad_act", x.t.numel() * 2, dy.device)
        L.check(lib.ustrun_act16(C.byref(src), x.N, act.data_ptr(), dt, stream_ptr()), "ustrun_act16")
        src = L.nhwc_src(act.data_ptr(), x.C, x.H, x.W)
    L.check(lib.ustrun_conv2d_wgrad(C.byref(src), 1, dy.data_ptr(), y.N, y.H, y.W, co, k, s, d, dw.data_ptr(), 0, part.data_ptr(), pb, dt,
                                    stream_ptr()), "ustrun_conv2d_wgrad")
    grads[conv.weight] = dw


def _zero_insert(t, H, W):
    """[N,h,w,C] -> [N,H,W,C] with t at the even positions (the adjoint of a stride-2 subsampling; tensor plumbing)"""
    z = torch.zeros(t.shape[0], H, W, t.shape[3], dtype=t.dtype, device=t.device)
    z[:, ::2, ::2][:, :t.shape[1], :t.shape[2]] = t
    return z


def _conv_dgrad(dy, y, x, conv, dt):
    """gradient of the convolution's input (an Act of x's shape) from dy (shape of y): a convolution of dy with the flipped,
    transposed weight; stride 2: 3x3 over the zero-inserted dy, 1x1 at the low resolution then zero-inserted."""
    lib = L.lib()
    co, ci, k, _ = conv.weight.shape
    s, d = conv.stride[0], conv.dilation[0]
    wd = _packed_dgrad(conv, dt)
    if s == 2 and k == 3:
        dy, Hs, Ws = _zero_insert(dy, x.H, x.W), x.H, x.W
    elif s == 2:
        Hs, Ws = y.H, y.W
    else:
        Hs, Ws = x.H, x.W
    src = L.nhwc_src(dy.data_ptr(), co, Hs, Ws)
    dx = torch.empty(x.N, Hs, Ws, ci, dtype=dy.dtype, device=dy.device)
    L.check(lib.ustrun_conv2d_fwd(C.byref(src), 1, wd.data_ptr(), None, x.N, Hs, Ws, ci, k, 1, d, dx.data_ptr(), 0, None, None, dt,
                                  stream_ptr()), "ustrun_conv2d_fwd")
    if s == 2 and k == 1:
        dx = _zero_insert(dx, x.H, x.W)
    return dx


def _join(a, b, ref, dt):
    lib = L.lib()
    g = torch.empty_like(a)
    L.check(lib.ustrun_relu_bwd_add(a.data_ptr(), None if b is None else b.data_ptr(), None if ref is None else ref.data_ptr(), a.numel(),
                                    g.data_ptr(), dt, stream_ptr()), "ustrun_relu_bwd_add")
    return g


def _bn_apply(y, da, coef, dt, relu=False):
    """the apply half of _bn_backward from coefficients already formed (sums fused into the launch that produced da)"""
    lib = L.lib()
    sc, sh = (y.aff[0], y.aff[1]) if relu else _scratch.noact(y.C, y.t.device)
    dy = torch.empty_like(y.t)
    L.check(lib.ustrun_bn_bwd_apply(da.data_ptr(), None, y.t.data_ptr(), sc.data_ptr(), sh.data_ptr(), coef.data_ptr(), y.N, y.H, y.W,
                                    y.C, dy.data_ptr(), dt, stream_ptr()), "ustrun_bn_bwd_apply")
    return dy


def _finalize_fused_sums(stat, rows, y, bn, grads):
    """rows of {sum(g mask), sum(g mask y)} a fused epilogue wrote -> dgamma, dbeta (into grads) and the apply coefficients"""
    lib = L.lib()
    dev = y.t.device
    coef = torch.empty(3, y.C, device=dev)
    dg, db = torch.empty(y.C, device=dev), torch.empty(y.C, device=dev)
    L.check(lib.ustrun_bn_bwd_finalize_stat(stat.data_ptr(), rows, 1, y.C, y.N * y.H * y.W, bn.weight.data_ptr(), y.aff[2].data_ptr(),
                                            y.aff[3].data_ptr(), 0, dg.data_ptr(), db.data_ptr(), 0, coef.data_ptr(), stream_ptr()),
            "ustrun_bn_bwd_finalize_stat")
    grads[bn.weight], grads[bn.bias] = dg, db
    return coef


def _conv3_dgrad_bn2(dy3, y3, y2, conv, bn, dt, grads):
    """da2 = dgrad(conv3)(dy3) and dy2 = BatchNorm + ReLU backward of bn2: with the sums of bn2's backward formed in the 1x1 input
    gradient's epilogue where the library covers the shape (one launch instead of the gradient + a reduce pass over da2 and y2)"""
    lib = L.lib()
    co, ci, k, _ = conv.weight.shape
    if k == 1 and conv.stride[0] == 1 and dt != L.F32:
        da2 = torch.empty_like(y2.t)
        rows, fused = C.c_int(0), C.c_int(0)
        stat = _scratch.get("join_stat", lib.ustrun_conv_mtiles(y2.N, y2.H, y2.W, ci) * 2 * ci * 4, dy3.device)
        L.check(lib.ustrun_conv1x1_dgrad_join(dy3.data_ptr(), _packed_dgrad(conv, dt).data_ptr(), y2.N, y2.H, y2.W, co, ci, None, None,
                                              da2.data_ptr(), y2.t.data_ptr(), y2.aff[0].data_ptr(), y2.aff[1].data_ptr(), stat.data_ptr(),
                                              C.byref(rows), C.byref(fused), dt, stream_ptr()), "ustrun_conv1x1_dgrad_join")
        if fused.value:
            return _bn_apply(y2, da2, _finalize_fused_sums(stat, rows.value, y2, bn, grads), dt, relu=True)
    da2 = _conv_dgrad(dy3, y3, y2, conv, dt)
    return _bn_backward(y2, da2, True, bn, grads, dt)


def _conv2_dgrad_bn1(dy2, y2, y1, conv, bn, dt, grads):
    """da1 = dgrad(conv2)(dy2) and dy1 = BatchNorm + ReLU backward of bn1, the sums formed in the (dilated) 3x3 input gradient's
    epilogue where the library covers the shape (stride 1, 16-bit storage, >= 128 channels)"""
    lib = L.lib()
    co, ci, k, _ = conv.weight.shape
    if k == 3 and conv.stride[0] == 1 and dt != L.F32:
        da1 = torch.empty_like(y1.t)
        rows = C.c_int(0)
        stat = _scratch.get("join_stat", lib.ustrun_conv_mtiles(y1.N, y1.H, y1.W, ci) * 2 * ci * 4, dy2.device)
        L.check(lib.ustrun_conv2d_dgrad_bnsum(dy2.data_ptr(), _packed_dgrad(conv, dt).data_ptr(), y1.N, y1.H, y1.W, co, ci, conv.dilation[0],
                                              da1.data_ptr(), y1.t.data_ptr(), y1.aff[0].data_ptr(), y1.aff[1].data_ptr(), stat.data_ptr(),
                                              C.byref(rows), dt, stream_ptr()), "ustrun_conv2d_dgrad_bnsum")
        if rows.value:
            return _bn_apply(y1, da1, _finalize_fused_sums(stat, rows.value, y1, bn, grads), dt, relu=True)
    da1 = _conv_dgrad(dy2, y2, y1, conv, dt)
    return _bn_backward(y1, da1, True, bn, grads, dt)


def _conv1_dgrad_join(dy1, y1, x, conv, dxb, ref, prev, dt, grads):
    """Gradient of the PREVIOUS block's pre-ReLU sum: (dgrad(conv1)(dy1) + dxb) * (ref > 0), ref = this block's input (the previous
    block's output; None behind the max-pool).  One launch where the library fuses the join into the 1x1 input gradient's epilogue
    (ustrun_conv1x1_dgrad_join: 16-bit storage, round 6), which then also forms the sums of the previous block's bn3 backward:
    -> (G_prev, coef of prev.bn3 or None)."""
    lib = L.lib()
    co, ci, k, _ = conv.weight.shape
    if k == 1 and conv.stride[0] == 1 and dt != L.F32:
        dev = dy1.device
        g = torch.empty_like(x.t)
        y3p, bn3p = (prev[5], prev[1].bn3) if prev is not None else (None, None)
        rows, fused = C.c_int(0), C.c_int(0)
        stat = None
        if y3p is not None:
            stat = _scratch.get("join_stat", lib.ustrun_conv_mtiles(x.N, x.H, x.W, ci) * 2 * ci * 4, dev)
        L.check(lib.ustrun_conv1x1_dgrad_join(dy1.data_ptr(), _packed_dgrad(conv, dt).data_ptr(), x.N, x.H, x.W, co, ci,
                                              None if dxb is None else dxb.data_ptr(), None if ref is None else ref.data_ptr(), g.data_ptr(),
                                              None if y3p is None else y3p.t.data_ptr(), None, None, None if stat is None else stat.data_ptr(),
                                              C.byref(rows), C.byref(fused), dt, stream_ptr()), "ustrun_conv1x1_dgrad_join")
        if fused.value:
            return g, (None if y3p is None else _finalize_fused_sums(stat, rows.value, y3p, bn3p, grads))
    dxa = _conv_dgrad(dy1, y1, x, conv, dt)
    return _join(dxa, dxb, ref, dt), None


def _block_backward(rec, G, dt, grads, prev=None, coef3=None, first=False):
    """G = gradient of the block's pre-ReLU join sum -> (gradient of the PREVIOUS block's pre-ReLU sum -- or of the max-pool output
    for the first block --, the coefficients of the previous block's bn3 backward when the join's launch formed its sums).
    coef3: this block's own bn3 coefficients, when the join that produced G formed them."""
    _, blk, x, y1, y2, y3, yd, _ = rec
    dy3 = _bn_apply(y3, G, coef3, dt) if coef3 is not None else _bn_backward(y3, G, False, blk.bn3, grads, dt)
    _conv_wgrad(y2, dy3, y3, blk.conv3, grads, dt)
    dy2 = _conv3_dgrad_bn2(dy3, y3, y2, blk.conv3, blk.bn2, dt, grads)
    _conv_wgrad(y1, dy2, y2, blk.conv2, grads, dt)
    dy1 = _conv2_dgrad_bn1(dy2, y2, y1, blk.conv2, blk.bn1, dt, grads)
    _conv_wgrad(x, dy1, y1, blk.conv1, grads, dt)
    if yd is None:
        dxb = G
    else:
        dyd = _bn_backward(yd, G, False, blk.downsample[1], grads, dt)
        _conv_wgrad(x, dyd, yd, blk.downsample[0], grads, dt)
        dxb = _conv_dgrad(dyd, yd, x, blk.downsample[0], dt)
    # the previous block's ReLU (none after the max-pool); a previous block WITH a projection shortcut needs G for two BatchNorms:
    # its sums stay with _bn_backward there (4 of ResNet-101's 33 blocks)
    return _conv1_dgrad_join(dy1, y1, x, blk.conv1, dxb, None if first else x.t, None if (first or prev is None or prev[6] is not None) else prev,
                             dt, grads)


def _head_backward(rec, dlogits, dt, grads):
    """deeplabv2.py:26-30 backward: dlogits NCHW f32 -> classifier gradients and the gradient of c4"""
    lib = L.lib()
    _, net, c4, (N, K, H, W) = rec
    dev = dlogits.device
    nr = len(net.classifier)
    dlow = torch.empty(N, c4.H, c4.W, K, device=dev)
    L.check(lib.ustrun_sum_resize_bilinear_bwd(dlogits.data_ptr(), N, c4.H, c4.W, K, H, W, dlow.data_ptr(), stream_ptr()),
            "ustrun_sum_resize_bilinear_bwd")
    db = torch.empty(K, device=dev)
    L.check(lib.ustrun_colsum(dlow.data_ptr(), N * c4.H * c4.W, K, db.data_ptr(), 0, stream_ptr()), "ustrun_colsum")
    ZC = nr * 9 * K
    ZCp = (ZC + 63) // 64 * 64
    dz = torch.empty(N, c4.H, c4.W, ZCp, dtype=_tdtype(dt), device=dev)
    rates = (C.c_int * 4)(*[c.dilation[0] for c in net.classifier])
    L.check(lib.ustrun_aspp_scatter(dlow.data_ptr(), N, c4.H, c4.W, K, nr, rates, ZCp, dz.data_ptr(), dt, stream_ptr()), "ustrun_aspp_scatter")
    Cin = c4.C
    # weight gradients: one TN GEMM c4^T x dz -> [ZCp][Cin], rows (r*9+tap)*K+k
    pb = lib.ustrun_wgrad_partials_bytes(1, Cin, ZCp, N * c4.H * c4.W)
    part = _scratch.get("wgrad", pb, dev)
    dwall = torch.empty(ZCp, Cin, device=dev)
    src = c4.src()
    L.check(lib.ustrun_conv2d_wgrad(C.byref(src), 1, dz.data_ptr(), N, c4.H, c4.W, ZCp, 1, 1, 1, dwall.data_ptr(), 0, part.data_ptr(), pb, dt,
                                    stream_ptr()), "ustrun_conv2d_wgrad")
    dw = dwall[:ZC].reshape(nr, 9, K, Cin).permute(0, 2, 3, 1).reshape(nr, K, Cin, 3, 3)
    for r, c in enumerate(net.classifier):
        grads[c.weight] = dw[r].contiguous()
        grads[c.bias] = db.clone()
    # input gradient: dz x Wall -> [.., Cin]; the transposed, column-padded weight as a 1x1 convolution [Cin][ZCp]
    key = tuple((c.weight.data_ptr(), c.weight._version) for c in net.classifier) + (dt, ZCp, _GEN[0])
    if getattr(net, "_ustrun_clsd_key", None) != key:
        wall = torch.stack([c.weight.detach().reshape(K, Cin, 9).permute(2, 0, 1) for c in net.classifier], 0).reshape(ZC, Cin)
        wd = torch.nn.functional.pad(wall.t(), (0, ZCp - ZC)).contiguous()                          # [Cin][ZCp]
        buf = torch.zeros(lib.ustrun_pack_conv_elems(Cin, ZCp, 1), dtype=_tdtype(dt), device=dev)
        L.check(lib.ustrun_pack_conv(wd.data_ptr(), Cin, ZCp, 1, buf.data_ptr(), dt, stream_ptr()), "ustrun_pack_conv")
        net._ustrun_clsd, net._ustrun_clsd_key = buf, key
    dc4 = torch.empty_like(c4.t)
    zsrc = L.nhwc_src(dz.data_ptr(), ZCp, c4.H, c4.W)
    L.check(lib.ustrun_conv2d_fwd(C.byref(zsrc), 1, net._ustrun_clsd.data_ptr(), None, N, c4.H, c4.W, Cin, 1, 1, 1, dc4.data_ptr(), 0, None,
                                  None, dt, stream_ptr()), "ustrun_conv2d_fwd")
    return dc4


def _stem_backward(rec, dpool, dt, grads):
    """resnet.py:160-163 backward: gradient of the max-pool output -> stem BatchNorm and 7x7 weight gradients"""
    lib = L.lib()
    _, net, y0, p = rec
    dev = dpool.device
    da0 = torch.empty_like(y0.t)
    L.check(lib.ustrun_maxpool3x3s2_bwd(dpool.data_ptr(), y0.t.data_ptr(), y0.aff[0].data_ptr(), y0.aff[1].data_ptr(), y0.N, y0.H, y0.W,
                                        y0.C, da0.data_ptr(), dt, stream_ptr()), "ustrun_maxpool3x3s2_bwd")
    dy0 = _bn_backward(y0, da0, True, net.bn1, grads, dt)
    pa, win, k, Cin = y0.keep
    co, Kp = y0.C, pa.C
    pb = lib.ustrun_wgrad_partials_bytes(1, Kp, co, y0.N * y0.H * y0.W)
    part = _scratch.get("wgrad", pb, dev)
    dwp = torch.empty(co, Kp, device=dev)                                        # [co][ky*win + kx*Cin + ci]
    src = pa.src()
    L.check(lib.ustrun_conv2d_wgrad(C.byref(src), 1, dy0.data_ptr(), y0.N, y0.H, y0.W, co, 1, 1, 1, dwp.data_ptr(), 0, part.data_ptr(), pb, dt,
                                    stream_ptr()), "ustrun_conv2d_wgrad")
    grads[net.conv1.weight] = dwp[:, :k * win].reshape(co, k, win)[:, :, :k * Cin].reshape(co, k, k, Cin).permute(0, 3, 1, 2).contiguous()


def deeplabv2_backward(net, tape, dlogits):
    """Walk the tape of one train-mode deeplabv2_forward in reverse: {parameter: gradient (f32, the parameter's shape)}."""
    dt = _DT[net.backbone.compute_dtype]
    grads = {}
    dlogits = dlogits.contiguous().float()
    rec = tape[-1]
    assert rec[0] == "head"
    dc4 = _head_backward(rec, dlogits, dt, grads)
    G = _join(dc4, None, rec[2].t, dt)                           # through the last block's ReLU
    coef3 = None
    for i in range(len(tape) - 2, 0, -1):
        G, coef3 = _block_backward(tape[i], G, dt, grads, prev=tape[i - 1] if i > 1 else None, coef3=coef3, first=(i == 1))
    _stem_backward(tape[0], G, dt, grads)
    _join_side_stream(dlogits.device)
    return grads


class DeepLabFn(torch.autograd.Function):
    """logits = DeepLabV2(x) with parameter gradients from deeplabv2_backward.  The parameters enter as inputs only so that autograd
    routes their gradients; the engine reads them from the module."""

    @staticmethod
    def forward(ctx, net, x, *params):
        tape = []
        out = deeplabv2_forward(net, x, tape)
        ctx.net, ctx.tape, ctx.params = net, tape, params
        lib = L.lib()
        ctx.debug_flags = lib.ustrun_debug_flags(0)      # (per calling thread; autograd runs backward() on its own thread)
        lib.ustrun_debug_flags(ctx.debug_flags)
        return out

    @staticmethod
    def backward(ctx, dlogits):
        if ctx.tape is None:
            raise RuntimeError("DeepLabFn: the forward's activations were released by an earlier backward")
        restore = L.lib().ustrun_debug_flags(ctx.debug_flags)
        try:
            grads = deeplabv2_backward(ctx.net, ctx.tape, dlogits)
        finally:
            L.lib().ustrun_debug_flags(restore)
        ctx.tape = None
        byid = {id(k): v for k, v in grads.items()}
        named = list(ctx.net.parameters())
        sink = getattr(ctx.net, "_ustrun_grad_sink", None)
        if sink is not None:
            # the trainer's flat gradient buffer (ustrun.trainer.SSLTrainer): written by the first backward of a step, added to
            # by the following ones -- one multi-tensor launch instead of autograd's per-parameter accumulation
            gl = [byid[id(p)] for p in named]
            if getattr(ctx.net, "_ustrun_sink_fresh", True):
                torch._foreach_copy_(list(sink), gl)
            else:
                torch._foreach_add_(list(sink), gl)
            ctx.net._ustrun_sink_fresh = False
            return (None, None) + (None,) * len(named)
        return (None, None) + tuple(byid.get(id(p)) if p.requires_grad else None for p in named)


def deeplabv2_apply(net, x):
    """DeepLabV2.base_forward: differentiable when autograd is recording and the module trains, plain forward otherwise."""
    if x.dim() == 4 and x.shape[1] == 1:
        x = x.expand(-1, 3, -1, -1)              # a grey image (BUSI, prostate) as three equal channels: the backbone's stem is 3-channel
    params = list(net.parameters())
    wants_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    if wants_grad and net.training:
        _check_input(net.backbone, x, differentiable=True)
        return DeepLabFn.apply(net, x, *params)
    if wants_grad:
        # the reference module is differentiable in eval mode too (frozen-BatchNorm fine-tuning); this backward only knows
        # train-mode BatchNorm, and returning logits without a grad_fn would fail silently
        raise NotImplementedError("DeepLabV2 in eval mode records no autograd graph here: call it under torch.no_grad(), or "
                                  "switch the module to train() to differentiate it")
    return deeplabv2_forward(net, x)
