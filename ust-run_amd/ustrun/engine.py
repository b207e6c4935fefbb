"""Whole-network forward/backward through the C ABI (ustrun_unet_forward / _backward).

PyTorch's role here is plumbing: it owns device memory (parameters, workspaces, outputs), the
stream, and the autograd graph node; every FLOP runs in libustrun.so.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

_DT = {"f32": L.F32, "bf16": L.BF16, "f16": L.F16, "f32x3": L.F32X3}


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def conv_bn_list(model):
    """The 18 (conv, bn) pairs in network order + the 4 ConvTranspose modules + the head conv."""
    dcs = [model.inc, model.down1.maxpool_conv[1], model.down2.maxpool_conv[1], model.down3.maxpool_conv[1],
           model.down4.maxpool_conv[1], model.up1.conv, model.up2.conv, model.up3.conv, model.up4.conv]
    pairs = []
    dom = getattr(model, "_ustrun_domain", 0) if getattr(model, "num_domains", 0) else None
    pick = (lambda bn: bn) if dom is None else (lambda bn: bn.bns[dom])       # domain-specific BatchNorm: the call's domain (networks/dsbn.py)
    for dc in dcs:
        s = dc.double_conv
        pairs += [(s[0], pick(s[1])), (s[3], pick(s[4]))]
    # (bilinear=True: Up.up is a parameter-free nn.Upsample -- the plan interpolates, there is no ConvTranspose to hand over)
    ups = [] if getattr(model, "bilinear", False) else [model.up1.up, model.up2.up, model.up3.up, model.up4.up]
    return pairs, ups, model.outc.conv


def _check_param(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise RuntimeError(f"{name}: parameters/buffers must be contiguous float32 HIP tensors")


def _desc(model, N, H, W, train, groups=1, tail=0, lead=0):
    pairs, ups, head = conv_bn_list(model)
    d = L.UNetDesc()
    d.N, d.C, d.H, d.W, d.K = N, model.n_channels, H, W, model.n_classes
    d.groups, d.tail, d.lead = groups, tail, lead
    d.base, d.dtype = model.base_channels, _DT[model.compute_dtype]
    d.bilinear = int(bool(getattr(model, "bilinear", False)))
    d.train, d.update_running = int(train), int(train)
    bn0 = pairs[0][1]
    d.momentum, d.eps = float(bn0.momentum), float(bn0.eps)
    for i, (cv, bn) in enumerate(pairs):
        for t in (cv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var):
            _check_param(t, "UNet")
        d.conv_w[i] = cv.weight.data_ptr()
        d.bn_w[i], d.bn_b[i] = bn.weight.data_ptr(), bn.bias.data_ptr()
        d.bn_rm[i], d.bn_rv[i] = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        d.bn_nbt[i] = bn.num_batches_tracked.data_ptr()
    for j, u in enumerate(ups):
        _check_param(u.weight, "UNet.up")
        d.up_w[j], d.up_b[j] = u.weight.data_ptr(), u.bias.data_ptr()
    _check_param(head.weight, "UNet.outc")
    d.head_w, d.head_b = head.weight.data_ptr(), head.bias.data_ptr()
    return d


def _weights_key(model):
    pairs, ups, _ = conv_bn_list(model)
    return tuple((m.weight.data_ptr(), m.weight._version) for m, _ in pairs) + \
        tuple((u.weight.data_ptr(), u.weight._version) for u in ups)


def invalidate_packed(model):
    """Call after parameters were modified behind autograd's back (e.g. the fused SGD kernel)."""
    model._ustrun_packed_key = None
    if not hasattr(model, "outc"):               # a DeepLabV2 / ResNet: its packs are cached per convolution
        from . import resnet_engine
        resnet_engine.invalidate_packs()


def _ensure_packed(model, d):
    lib = L.lib()
    key = _weights_key(model)
    pk = getattr(model, "_ustrun_packed", None)
    if pk is None or pk.device != model.outc.conv.weight.device:
        nbytes = lib.ustrun_unet_packed_bytes(C.byref(d))
        if nbytes < 0:
            L.check(1, "ustrun_unet_packed_bytes")
        pk = torch.empty(nbytes, dtype=torch.uint8, device=model.outc.conv.weight.device)
        model._ustrun_packed, model._ustrun_packed_key = pk, None
    d.packed = pk.data_ptr()
    if getattr(model, "_ustrun_packed_key", None) != key:
        L.check(lib.ustrun_unet_pack(C.byref(d), stream_ptr()), "ustrun_unet_pack")
        model._ustrun_packed_key = key


def model_params(model):
    """the parameters ustrun_unet_backward writes gradients for, in the order its grads[] array expects: model.parameters() -- of a
    network with domain-specific BatchNorm, the call's domain's members in their place (the other domains' get no gradient)"""
    if not getattr(model, "num_domains", 0):
        return list(model.parameters())
    pairs, ups, head = conv_bn_list(model)
    out = []
    for i, (cv, bn) in enumerate(pairs):
        if i >= 10 and i % 2 == 0:                # a decoder block: its ConvTranspose comes first (Up.up before Up.conv)
            u = ups[(i - 10) // 2]
            out += [u.weight, u.bias]
        out += [cv.weight, bn.weight, bn.bias]
    return out + [head.weight, head.bias]


def _current_debug_flags(lib):
    """both words (ustrun_debug_flags, ustrun_debug_flags2) of the calling thread"""
    f = lib.ustrun_debug_flags(0)
    lib.ustrun_debug_flags(f)
    f2 = lib.ustrun_debug_flags2(0)
    lib.ustrun_debug_flags2(f2)
    return f, f2


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, feature, groups, tail, lead, *params):
        logits, feat, ws, d = _run_forward(model, x, feature, groups, tail, lead)
        ctx.model, ctx.ws, ctx.desc, ctx.x = model, ws, d, x
        # ustrun_debug_flags is per calling thread and autograd runs backward() on a thread of its own: the backward runs under
        # the flags the forward ran under (some of them shape the plan both halves share)
        ctx.debug_flags = _current_debug_flags(L.lib())
        ctx.nparams = len(params)
        if feature:
            ctx.mark_non_differentiable(feat)
            return logits, feat
        return logits

    @staticmethod
    def backward(ctx, dlogits, *unused):
        model, d = ctx.model, ctx.desc
        lib = L.lib()
        restore = lib.ustrun_debug_flags(ctx.debug_flags[0]), lib.ustrun_debug_flags2(ctx.debug_flags[1])
        try:
            return _UNetFn._backward(ctx, lib, model, d, dlogits)
        finally:
            lib.ustrun_debug_flags(restore[0])
            lib.ustrun_debug_flags2(restore[1])

    @staticmethod
    def _backward(ctx, lib, model, d, dlogits):
        dlogits = dlogits.contiguous()
        nbytes = lib.ustrun_unet_bwd_scratch_bytes(C.byref(d))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dlogits.device)
        params = model_params(model)
        sink = getattr(model, "_ustrun_grad_sink", None)
        if sink is not None:
            targets, accumulate = sink, 0 if getattr(model, "_ustrun_sink_fresh", True) else 1
            model._ustrun_sink_fresh = False
        else:
            flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=dlogits.device)
            targets, o = [], 0
            for p in params:
                targets.append(flat[o:o + p.numel()].view_as(p))
                o += p.numel()
            accumulate = 0
        arr = (C.c_void_p * len(targets))(*[t.data_ptr() for t in targets])
        split = getattr(model, "_ustrun_backward_split_hook", None)
        if split is None:
            L.check(lib.ustrun_unet_backward(C.byref(d), ctx.x.data_ptr(), dlogits.data_ptr(), ctx.ws.data_ptr(),
                                             scratch.data_ptr(), arr, accumulate, stream_ptr()), "ustrun_unet_backward")
        else:       # head + decoder, hand the (now final) decoder gradients to the caller, then the encoder -- down4
            mid = getattr(model, "_ustrun_backward_mid_hook", None)        # first when the caller wants its gradients early
            for part in ((1, 3, 4) if mid is not None else (1, 2)):
                L.check(lib.ustrun_unet_backward_part(C.byref(d), ctx.x.data_ptr(), dlogits.data_ptr(), ctx.ws.data_ptr(),
                                                      scratch.data_ptr(), arr, accumulate, part, stream_ptr()),
                        "ustrun_unet_backward_part")
                if part == 1:
                    split()
                elif part == 3:
                    mid()
        ctx.ws = None
        grads = (None,) * ctx.nparams if sink is not None else tuple(targets)
        return (None, None, None, None, None, None) + grads


def _run_forward(model, x, feature, groups=1, tail=0, lead=0):
    lib = L.lib()
    if x.dim() != 4 or x.shape[1] != model.n_channels:
        raise RuntimeError(f"UNet: expected input [N,{model.n_channels},H,W], got {tuple(x.shape)}")
    if x.dtype != torch.float32:
        raise RuntimeError("UNet: input must be float32")
    x = x.contiguous()
    N, _, H, W = x.shape
    if groups < 1 or tail < 0 or tail >= N or (N - tail) % groups or (tail and tail >= (N - tail) // groups):
        raise RuntimeError(f"UNet: batch {N} does not split into {groups} passes" + (f" and a shorter tail of {tail}" if tail else ""))
    if lead < 0 or lead >= groups:
        raise RuntimeError(f"UNet: {lead} leading passes without gradient of {groups} passes")
    d = _desc(model, N, H, W, model.training, groups, tail, lead)
    _ensure_packed(model, d)
    nbytes = lib.ustrun_unet_fwd_workspace_bytes(C.byref(d))
    if nbytes < 0:
        L.check(1, "ustrun_unet_fwd_workspace_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    logits = torch.empty((N, model.n_classes, H, W), dtype=torch.float32, device=x.device)
    feat = torch.empty((N, model.base_channels, H, W), dtype=torch.float32, device=x.device) if feature else None
    L.check(lib.ustrun_unet_forward(C.byref(d), x.data_ptr(), logits.data_ptr(), L.ptr(feat), ws.data_ptr(),
                                    stream_ptr()), "ustrun_unet_forward")
    if tail:        # the tail pass moves BatchNorm statistics only: the head skipped it, the backward covers the passes in front of it
        logits = logits[:N - tail]
        feat = feat[:N - tail] if feat is not None else None
    return logits, feat, ws, d


def unet_forward(model, x, feature=False, groups=1, tail=0, lead=0):
    """groups > 1: x holds `groups` independent forward passes laid end to end along the batch axis; they run as one
    batched call with BatchNorm statistics (and running-buffer updates, in order) kept per pass -- the results are
    those of `groups` separate calls, at the launch count and GPU fill of one.  tail > 0: the last `tail` images of x are
    one more, shorter pass whose output nobody needs (it only moves the BatchNorm running statistics, after the others):
    the returned logits cover the passes in front of it.  lead > 0: the first `lead` passes get no gradient (their rows of the
    logits' gradient are ignored): a no-grad forward that must come first in BatchNorm order rides in the same call."""
    params = model_params(model)
    needs_grad = torch.is_grad_enabled() and model.training and any(p.requires_grad for p in params)
    if x.requires_grad:
        raise NotImplementedError("gradient w.r.t. the network input is not on the hot path (train.py never needs it)")
    if needs_grad:
        return _UNetFn.apply(x, model, feature, groups, tail, lead, *params)
    logits, feat, _, _ = _run_forward(model, x, feature, groups, tail, lead)
    return (logits, feat) if feature else logits
