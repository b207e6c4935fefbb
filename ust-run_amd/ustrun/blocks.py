"""Block-level entry points: DoubleConv / Down / Up / OutConv called on their own (reference
networks/unet_parts.py:8-76), through the operator-level C ABI.

The reference's scripts never call the blocks directly -- only `UNet.forward` does (train.py:643-702) -- and
`UNet.forward` runs the whole network as one fused call (engine.py).  A stand-alone block has to materialise the
activated tensor at its edges, so this module is the convenience / parity surface, not the hot path: it works in
f32 (the exact path), NCHW in and out like the reference modules; layout changes at the block edge (NCHW <-> the
kernels' NHWC) are tensor plumbing, every FLOP runs in libustrun.so.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .engine import stream_ptr

DT = L.F32


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a float32 HIP tensor, got {t.dtype} on {t.device}")
    return t.contiguous()


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def _pack3(w):
    lib = L.lib()
    co, ci = w.shape[:2]
    wf, wd = torch.empty(9 * ci * co, device=w.device), torch.empty(9 * ci * co, device=w.device)
    L.check(lib.ustrun_pack_conv3x3(w.data_ptr(), co, ci, wf.data_ptr(), wd.data_ptr(), DT, stream_ptr()), "pack_conv3x3")
    return wf, wd


def _conv_bn_fwd(srcs, nsrc, wf, N, H, W, bn, train):
    """raw conv output y (NHWC) and the BatchNorm constants aff = [scale, shift, mean, rstd] (train: batch statistics,
    running buffers updated; eval: running statistics)."""
    lib = L.lib()
    Cout = bn.num_features
    y = torch.empty(N, H, W, Cout, device=wf.device)
    aff = torch.empty(4, Cout, device=wf.device)
    if train:
        rows = lib.ustrun_conv_mtiles(N, H, W, Cout)
        stat = torch.empty(rows, 2, Cout, device=wf.device)
        used = C.c_int(0)
        L.check(lib.ustrun_conv3x3_fwd_rows(srcs, nsrc, wf.data_ptr(), N, H, W, Cout, y.data_ptr(), stat.data_ptr(),
                                            C.byref(used), DT, stream_ptr()), "conv3x3_fwd")
        L.check(lib.ustrun_bn_finalize(stat.data_ptr(), used.value, Cout, N * H * W, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                       bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                       bn.num_batches_tracked.data_ptr(), float(bn.momentum), float(bn.eps), 1,
                                       aff[0].data_ptr(), aff[1].data_ptr(), aff[2].data_ptr(), aff[3].data_ptr(),
                                       stream_ptr()), "bn_finalize")
    else:
        L.check(lib.ustrun_conv3x3_fwd(srcs, nsrc, wf.data_ptr(), N, H, W, Cout, y.data_ptr(), None, DT, stream_ptr()),
                "conv3x3_fwd")
        L.check(lib.ustrun_bn_eval_affine(Cout, bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(),
                                          bn.running_var.data_ptr(), float(bn.eps), aff[0].data_ptr(), aff[1].data_ptr(),
                                          stream_ptr()), "bn_eval_affine")
    return y, aff


def _bn_bwd(da, y, aff, bn, N, H, W):
    """dy (grad wrt the raw conv output) from da (grad wrt the activated output); returns dy, dgamma, dbeta"""
    lib = L.lib()
    Cc = bn.num_features
    dg, db = torch.empty(Cc, device=da.device), torch.empty(Cc, device=da.device)
    coef = torch.empty(3, Cc, device=da.device)
    nb = lib.ustrun_bn_bwd_partials_bytes(N * H * W, Cc)
    part = torch.empty(nb // 4, device=da.device)
    L.check(lib.ustrun_bn_bwd_reduce(da.data_ptr(), None, y.data_ptr(), aff[0].data_ptr(), aff[1].data_ptr(), aff[2].data_ptr(),
                                     aff[3].data_ptr(), bn.weight.data_ptr(), N, H, W, Cc, dg.data_ptr(), db.data_ptr(), 0,
                                     coef.data_ptr(), part.data_ptr(), nb, DT, stream_ptr()), "bn_bwd_reduce")
    dy = torch.empty_like(da)
    L.check(lib.ustrun_bn_bwd_apply(da.data_ptr(), None, y.data_ptr(), aff[0].data_ptr(), aff[1].data_ptr(), coef.data_ptr(), N,
                                    H, W, Cc, dy.data_ptr(), DT, stream_ptr()), "bn_bwd_apply")
    return dy, dg, db


def _wgrad3(srcs, nsrc, dy, N, H, W, Cin, Cout):
    lib = L.lib()
    nb = lib.ustrun_wgrad_partials_bytes(9, Cin, Cout, N * H * W)
    part = torch.empty(max(nb, 4) // 4, device=dy.device)
    dw = torch.empty(Cout, Cin, 3, 3, device=dy.device)
    L.check(lib.ustrun_conv3x3_wgrad(srcs, nsrc, dy.data_ptr(), N, H, W, Cout, dw.data_ptr(), 0, part.data_ptr(), nb, DT,
                                     stream_ptr()), "conv3x3_wgrad")
    return dw


def _one(src):
    a = (L.Src * 1)()
    a[0] = src
    return a


class _DoubleConvFn(torch.autograd.Function):
    """DoubleConv on one source (optionally max-pooled first: Down) or on cat[skip, pad(up)] (the tail of Up)."""

    @staticmethod
    def forward(ctx, module, pool, x, up, *params):
        lib = L.lib()
        dc = module.double_conv
        c1, b1, c2, b2 = dc[0], dc[1], dc[3], dc[4]
        train = module.training
        xh = _nhwc(_f32c(x, "DoubleConv input"))
        N, H, W, Cx = xh.shape
        if pool:
            ph = torch.empty(N, H // 2, W // 2, Cx, device=x.device)
            L.check(lib.ustrun_pool_act(C.byref(L.nhwc_src(xh.data_ptr(), Cx, H, W)), N, ph.data_ptr(), DT, stream_ptr()), "pool_act")
            H, W = H // 2, W // 2
        else:
            ph = xh
        if up is not None:                         # Up: channels = [skip (x), ConvTranspose output padded to the skip's extent]
            uh, uw = up.shape[1], up.shape[2]
            srcs = (L.Src * 2)()
            srcs[0] = L.nhwc_src(ph.data_ptr(), Cx, H, W)
            srcs[1] = L.nhwc_src(up.data_ptr(), up.shape[3], uh, uw, off=((H - uh) // 2, (W - uw) // 2))
            nsrc, cin = 2, Cx + up.shape[3]
        else:
            srcs, nsrc, cin = _one(L.nhwc_src(ph.data_ptr(), Cx, H, W)), 1, Cx
        wf1, wd1 = _pack3(c1.weight)
        wf2, wd2 = _pack3(c2.weight)
        y1, aff1 = _conv_bn_fwd(srcs, nsrc, wf1, N, H, W, b1, train)
        s1 = _one(L.nhwc_src(y1.data_ptr(), b1.num_features, H, W, scale=aff1[0].data_ptr(), shift=aff1[1].data_ptr(), relu=1))
        y2, aff2 = _conv_bn_fwd(s1, 1, wf2, N, H, W, b2, train)
        Co = b2.num_features
        out = torch.empty(N, Co, H, W, device=x.device)
        L.check(lib.ustrun_bn_relu_apply(y2.data_ptr(), aff2[0].data_ptr(), aff2[1].data_ptr(), N * H * W, Co, H * W, out.data_ptr(),
                                         1, DT, stream_ptr()), "bn_relu_apply")
        ctx.module, ctx.pool, ctx.train = module, pool, train
        ctx.saved = (xh, ph, up, y1, aff1, y2, aff2, wd1, wd2, cin)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.train:
            raise NotImplementedError("backward through eval-mode BatchNorm is not built (the reference never needs it)")
        lib = L.lib()
        dc = ctx.module.double_conv
        c1, b1, c2, b2 = dc[0], dc[1], dc[3], dc[4]
        xh, ph, up, y1, aff1, y2, aff2, wd1, wd2, cin = ctx.saved
        N, H, W, Cx = ph.shape
        Cm, Co = b1.num_features, b2.num_features
        da2 = _nhwc(_f32c(dout, "grad"))
        dy2, dg2, db2 = _bn_bwd(da2, y2, aff2, b2, N, H, W)
        s1 = _one(L.nhwc_src(y1.data_ptr(), Cm, H, W, scale=aff1[0].data_ptr(), shift=aff1[1].data_ptr(), relu=1))
        dw2 = _wgrad3(s1, 1, dy2, N, H, W, Cm, Co)
        da1 = torch.empty(N, H, W, Cm, device=dout.device)
        L.check(lib.ustrun_conv3x3_dgrad(dy2.data_ptr(), wd2.data_ptr(), N, H, W, Co, Cm, da1.data_ptr(), Cm, None, 0, 0, 0, 0, DT,
                                         stream_ptr()), "conv3x3_dgrad")
        dy1, dg1, db1 = _bn_bwd(da1, y1, aff1, b1, N, H, W)
        if up is not None:
            uh, uw = up.shape[1], up.shape[2]
            srcs = (L.Src * 2)()
            srcs[0] = L.nhwc_src(ph.data_ptr(), Cx, H, W)
            srcs[1] = L.nhwc_src(up.data_ptr(), up.shape[3], uh, uw, off=((H - uh) // 2, (W - uw) // 2))
            dw1 = _wgrad3(srcs, 2, dy1, N, H, W, cin, Cm)
            dph = torch.empty_like(ph)
            du = torch.empty_like(up)
            L.check(lib.ustrun_conv3x3_dgrad(dy1.data_ptr(), wd1.data_ptr(), N, H, W, Cm, cin, dph.data_ptr(), Cx, du.data_ptr(), uh,
                                             uw, (H - uh) // 2, (W - uw) // 2, DT, stream_ptr()), "conv3x3_dgrad")
        else:
            dw1 = _wgrad3(_one(L.nhwc_src(ph.data_ptr(), Cx, H, W)), 1, dy1, N, H, W, cin, Cm)
            dph = torch.empty_like(ph)
            du = None
            L.check(lib.ustrun_conv3x3_dgrad(dy1.data_ptr(), wd1.data_ptr(), N, H, W, Cm, cin, dph.data_ptr(), cin, None, 0, 0, 0, 0,
                                             DT, stream_ptr()), "conv3x3_dgrad")
        if ctx.pool:
            dxh = torch.empty_like(xh)
            L.check(lib.ustrun_maxpool_bwd(dph.data_ptr(), xh.data_ptr(), N, xh.shape[1], xh.shape[2], Cx, dxh.data_ptr(), DT,
                                           stream_ptr()), "maxpool_bwd")
        else:
            dxh = dph
        # (module, pool, x, up, conv1.w, bn1.w, bn1.b, conv2.w, bn2.w, bn2.b)
        return None, None, _nchw(dxh), du, dw1, dg1, db1, dw2, dg2, db2


def _dc_params(m):
    dc = m.double_conv
    return dc[0].weight, dc[1].weight, dc[1].bias, dc[3].weight, dc[4].weight, dc[4].bias


def double_conv(module, x, pool=False):
    return _DoubleConvFn.apply(module, pool, x, None, *_dc_params(module))


class _ConvTFn(torch.autograd.Function):
    """ConvTranspose2d(k=2, s=2) with bias: NCHW in, NHWC out (it only ever feeds the Up block's DoubleConv)."""

    @staticmethod
    def forward(ctx, x1, weight, bias):
        lib = L.lib()
        xh = _nhwc(_f32c(x1, "Up input"))
        N, H, W, Ci = xh.shape
        Co = weight.shape[1]
        wf, wd = torch.empty(4 * Ci * Co, device=x1.device), torch.empty(4 * Ci * Co, device=x1.device)
        L.check(lib.ustrun_pack_convT2x2(weight.data_ptr(), Ci, Co, wf.data_ptr(), wd.data_ptr(), DT, stream_ptr()), "pack_convT2x2")
        u = torch.empty(N, 2 * H, 2 * W, Co, device=x1.device)
        L.check(lib.ustrun_convT2x2_fwd(C.byref(L.nhwc_src(xh.data_ptr(), Ci, H, W)), wf.data_ptr(), bias.data_ptr(), N, H, W, Co,
                                        u.data_ptr(), DT, stream_ptr()), "convT2x2_fwd")
        ctx.saved = (xh, wd, Ci, Co)
        return u

    @staticmethod
    def backward(ctx, du):
        lib = L.lib()
        xh, wd, Ci, Co = ctx.saved
        N, H, W, _ = xh.shape
        du = du.contiguous()
        nb = max(lib.ustrun_wgrad_partials_bytes(4, Ci, Co, N * H * W), 512 * Co * 4)
        part = torch.empty(nb // 4, device=du.device)
        dw, db = torch.empty(Ci, Co, 2, 2, device=du.device), torch.empty(Co, device=du.device)
        L.check(lib.ustrun_convT2x2_wgrad(C.byref(L.nhwc_src(xh.data_ptr(), Ci, H, W)), du.data_ptr(), N, H, W, Co, dw.data_ptr(),
                                          db.data_ptr(), 0, part.data_ptr(), nb, DT, stream_ptr()), "convT2x2_wgrad")
        da = torch.empty_like(xh)
        L.check(lib.ustrun_convT2x2_dgrad(du.data_ptr(), wd.data_ptr(), N, H, W, Co, Ci, da.data_ptr(), DT, stream_ptr()),
                "convT2x2_dgrad")
        return _nchw(da), dw, db


class _BilinearFn(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (unet_parts.py:48-50): NCHW in, NHWC out."""

    @staticmethod
    def forward(ctx, x1):
        lib = L.lib()
        xh = _nhwc(_f32c(x1, "Up input"))
        N, H, W, Ci = xh.shape
        if Ci % 4:
            raise RuntimeError(f"Up(bilinear): {Ci} channels (needs a multiple of 4)")
        u = torch.empty(N, 2 * H, 2 * W, Ci, device=x1.device)
        L.check(lib.ustrun_upsample2x_fwd(xh.data_ptr(), N, H, W, Ci, u.data_ptr(), stream_ptr()), "upsample2x_fwd")
        ctx.shape = (N, H, W, Ci)
        return u

    @staticmethod
    def backward(ctx, du):
        lib = L.lib()
        N, H, W, Ci = ctx.shape
        du = du.contiguous()
        dx = torch.empty(N, H, W, Ci, device=du.device)
        L.check(lib.ustrun_upsample2x_bwd(du.data_ptr(), N, H, W, Ci, dx.data_ptr(), stream_ptr()), "upsample2x_bwd")
        return _nchw(dx)


def up(module, x1, x2):
    if isinstance(module.up, torch.nn.ConvTranspose2d):
        u = _ConvTFn.apply(x1, module.up.weight, module.up.bias)      # NHWC
    else:
        u = _BilinearFn.apply(x1)                                     # NHWC, channels unchanged
    return _DoubleConvFn.apply(module.conv, False, x2, u, *_dc_params(module.conv))


class _OutConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = L.lib()
        xh = _nhwc(_f32c(x, "OutConv input"))
        N, H, W, Cc = xh.shape
        K = weight.shape[0]
        w2 = weight.reshape(K, Cc).contiguous()
        out = torch.empty(N, K, H, W, device=x.device)
        L.check(lib.ustrun_head_fwd(xh.data_ptr(), None, None, N * H * W, H * W, Cc, K, w2.data_ptr(), bias.data_ptr(),
                                    out.data_ptr(), DT, stream_ptr()), "head_fwd")
        ctx.saved = (xh, w2, K)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = L.lib()
        xh, w2, K = ctx.saved
        N, H, W, Cc = xh.shape
        dout = _f32c(dout, "grad")
        da = torch.empty_like(xh)
        dw, db = torch.empty(K, Cc, device=dout.device), torch.empty(K, device=dout.device)
        nb = 1024 * (K * Cc + K) * 4
        part = torch.empty(nb // 4, device=dout.device)
        L.check(lib.ustrun_head_bwd(dout.data_ptr(), xh.data_ptr(), None, None, N * H * W, H * W, Cc, K, w2.data_ptr(), da.data_ptr(),
                                    dw.data_ptr(), db.data_ptr(), 0, part.data_ptr(), nb, DT, stream_ptr()), "head_bwd")
        return _nchw(da), dw.reshape(K, Cc, 1, 1), db


def out_conv(module, x):
    return _OutConvFn.apply(x, module.conv.weight, module.conv.bias)
