"""GPU parity of the individual C-ABI operators against the oracle's primitives (torch CPU f32,
with an f64 evaluation as the conditioning yardstick)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def L():
    from ustrun import _lib
    return _lib


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nhwc(t):          # [N,C,H,W] -> contiguous [N,H,W,C] on the GPU
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def from_nhwc(t):     # GPU [N,H,W,C] -> CPU [N,C,H,W]
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def pack_conv(w):
    l = L()
    co, ci = w.shape[:2]
    wf = torch.empty(9 * ci * co, device="cuda")
    wd = torch.empty(9 * ci * co, device="cuda")
    wg = w.contiguous().cuda()
    l.check(l.lib().ustrun_pack_conv3x3(wg.data_ptr(), co, ci, wf.data_ptr(), wd.data_ptr(), 0, None))
    return wf, wd


SHAPES = [(2, 512, 512, 4, 4), (2, 64, 64, 16, 16), (1, 24, 40, 9, 7), (2, 256, 128, 8, 8), (3, 128, 256, 6, 10),
          (2, 3, 64, 16, 16)]


@pytest.mark.parametrize("n,ci,co,h,w", SHAPES)
def test_conv3x3_fwd_dgrad_wgrad(n, ci, co, h, w):
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(ci * 7 + co)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)
    dy = torch.randn(n, co, h, w, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, 1, 1)
    y_ref.backward(dy)
    y64 = F.conv2d(x.double(), wt.double(), None, 1, 1)

    wf, wd = pack_conv(wt)
    xg = nhwc(x)
    src = l.nhwc_src(xg.data_ptr(), ci, h, w)
    y = torch.empty(n, h, w, co, device="cuda")
    mt = lib.ustrun_conv_mtiles(n, h, w, co)
    stat = torch.zeros(mt, 2, co, device="cuda")
    l.check(lib.ustrun_conv3x3_fwd(C.byref(src), 1, wf.data_ptr(), n, h, w, co, y.data_ptr(), stat.data_ptr(), 0, None))
    yc = from_nhwc(y)
    assert rel(yc, y64) < max(5 * rel(y_ref.detach(), y64), 2e-6)   # a K-long f32 fmaf chain (K up to 4608)
    np.testing.assert_allclose(stat[:, 0].sum(0).cpu().numpy(), y_ref.detach().sum((0, 2, 3)).numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(stat[:, 1].sum(0).cpu().numpy(), y_ref.detach().square().sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-4)

    # input gradient, written whole and split over two destinations
    dyg = nhwc(dy)
    da = torch.empty(n, h, w, ci, device="cuda")
    l.check(lib.ustrun_conv3x3_dgrad(dyg.data_ptr(), wd.data_ptr(), n, h, w, co, ci, da.data_ptr(), ci, None, 0, 0, 0, 0, 0, None))
    assert rel(from_nhwc(da), xr.grad) < 1e-5
    if ci % 8 == 0:
        c0 = ci // 2
        d0 = torch.empty(n, h, w, c0, device="cuda")
        d1 = torch.full((n, h, w, ci - c0), 7.0, device="cuda")
        l.check(lib.ustrun_conv3x3_dgrad(dyg.data_ptr(), wd.data_ptr(), n, h, w, co, ci, d0.data_ptr(), c0, d1.data_ptr(), h, w, 0, 0, 0, None))
        assert rel(from_nhwc(torch.cat([d0, d1], 3)), xr.grad) < 1e-5

    # weight gradient (torch layout), then accumulate a second time
    nb = lib.ustrun_wgrad_partials_bytes(9, ci, co, n * h * w)
    part = torch.empty(nb // 4, device="cuda")
    dw = torch.empty(co, ci, 3, 3, device="cuda")
    l.check(lib.ustrun_conv3x3_wgrad(C.byref(src), 1, dyg.data_ptr(), n, h, w, co, dw.data_ptr(), 0, part.data_ptr(), nb, 0, None))
    assert rel(dw.cpu(), wr.grad) < 1e-5
    l.check(lib.ustrun_conv3x3_wgrad(C.byref(src), 1, dyg.data_ptr(), n, h, w, co, dw.data_ptr(), 1, part.data_ptr(), nb, 0, None))
    assert rel(dw.cpu(), 2 * wr.grad) < 1e-5


def test_conv3x3_loader_affine_relu_pool_concat_pad():
    """BatchNorm affine + ReLU, 2x2 max-pool (negative scales included), concat and F.pad offset on load."""
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(1)
    n, c0, c1, co, h, w = 2, 16, 8, 24, 11, 13
    # pooled single source
    ys = torch.randn(n, c0, 2 * h + 1, 2 * w, generator=g)
    sc, sh = torch.randn(c0, generator=g), 0.3 * torch.randn(c0, generator=g)
    wt = torch.randn(co, c0, 3, 3, generator=g) / 12
    a = F.max_pool2d(torch.relu(ys * sc[None, :, None, None] + sh[None, :, None, None]), 2)
    ref = F.conv2d(a, wt, None, 1, 1)
    wf, _ = pack_conv(wt)
    yg, scg, shg = nhwc(ys), sc.cuda(), sh.cuda()
    src = l.nhwc_src(yg.data_ptr(), c0, 2 * h + 1, 2 * w, scg.data_ptr(), shg.data_ptr(), relu=1, pool=1)
    out = torch.empty(n, h, w, co, device="cuda")
    l.check(lib.ustrun_conv3x3_fwd(C.byref(src), 1, wf.data_ptr(), n, h, w, co, out.data_ptr(), None, 0, None))
    assert rel(from_nhwc(out), ref) < 1e-5
    # concat [skip (affine+relu), up (offset-padded, smaller extent)]
    skip = torch.randn(n, c0, h, w, generator=g)
    up = torch.randn(n, c1, h - 3, w - 2, generator=g)
    wt2 = torch.randn(co, c0 + c1, 3, 3, generator=g) / 14
    upp = F.pad(up, [1, 1, 1, 2])
    a2 = torch.cat([torch.relu(skip * sc[None, :, None, None] + sh[None, :, None, None]), upp], 1)
    ref2 = F.conv2d(a2, wt2, None, 1, 1)
    wf2, _ = pack_conv(wt2)
    sg, ug = nhwc(skip), nhwc(up)
    srcs = (l.Src * 2)(l.nhwc_src(sg.data_ptr(), c0, h, w, scg.data_ptr(), shg.data_ptr(), relu=1),
                       l.nhwc_src(ug.data_ptr(), c1, h - 3, w - 2, off=(1, 1)))
    out2 = torch.empty(n, h, w, co, device="cuda")
    l.check(lib.ustrun_conv3x3_fwd(srcs, 2, wf2.data_ptr(), n, h, w, co, out2.data_ptr(), None, 0, None))
    assert rel(from_nhwc(out2), ref2) < 1e-5


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 64, 32, 5, 7), (1, 512, 256, 4, 4), (2, 16, 8, 8, 8)])
def test_convT2x2_fwd_bwd(n, ci, co, h, w):
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(ci + co)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(ci, co, 2, 2, generator=g) / ci ** 0.5
    b = torch.randn(co, generator=g)
    du = torch.randn(n, co, 2 * h, 2 * w, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    u_ref = F.conv_transpose2d(xr, wr, br, stride=2)
    u_ref.backward(du)
    wf = torch.empty(4 * ci * co, device="cuda")
    wd = torch.empty(4 * ci * co, device="cuda")
    wg = wt.cuda()
    l.check(lib.ustrun_pack_convT2x2(wg.data_ptr(), ci, co, wf.data_ptr(), wd.data_ptr(), 0, None))
    xg, bg = nhwc(x), b.cuda()
    src = l.nhwc_src(xg.data_ptr(), ci, h, w)
    u = torch.empty(n, 2 * h, 2 * w, co, device="cuda")
    l.check(lib.ustrun_convT2x2_fwd(C.byref(src), wf.data_ptr(), bg.data_ptr(), n, h, w, co, u.data_ptr(), 0, None))
    assert rel(from_nhwc(u), u_ref.detach()) < 1e-5
    dug = nhwc(du)
    da = torch.empty(n, h, w, ci, device="cuda")
    l.check(lib.ustrun_convT2x2_dgrad(dug.data_ptr(), wd.data_ptr(), n, h, w, co, ci, da.data_ptr(), 0, None))
    assert rel(from_nhwc(da), xr.grad) < 1e-5
    nb = max(lib.ustrun_wgrad_partials_bytes(4, ci, co, n * h * w), 512 * co * 4)
    part = torch.empty(nb // 4, device="cuda")
    dw, db = torch.empty(ci, co, 2, 2, device="cuda"), torch.empty(co, device="cuda")
    l.check(lib.ustrun_convT2x2_wgrad(C.byref(src), dug.data_ptr(), n, h, w, co, dw.data_ptr(), db.data_ptr(), 0, part.data_ptr(), nb, 0, None))
    assert rel(dw.cpu(), wr.grad) < 1e-5 and rel(db.cpu(), br.grad) < 1e-5


@pytest.mark.parametrize("n,c,h,w,pool", [(2, 512, 4, 4, False), (2, 64, 16, 16, True), (1, 24, 9, 7, True),
                                          (3, 128, 6, 10, False), (2, 1024, 2, 2, False), (2, 512, 4, 4, True)])
def test_bn_relu_pool_backward(n, c, h, w, pool):
    """BatchNorm(train)+ReLU(+MaxPool) backward against torch autograd of the same expression."""
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(c + h)
    y = torch.randn(n, c, h, w, generator=g) * 2 + 0.5
    gamma = 1 + 0.5 * torch.randn(c, generator=g)
    beta = 0.2 * torch.randn(c, generator=g)
    da = torch.randn(n, c, h, w, generator=g)
    dp = torch.randn(n, c, h // 2, w // 2, generator=g)

    def ref(dtype):
        yr = y.detach().clone().to(dtype).requires_grad_(True)
        gr, br = gamma.detach().clone().to(dtype).requires_grad_(True), beta.detach().clone().to(dtype).requires_grad_(True)
        var, mean = torch.var_mean(yr, dim=(0, 2, 3), unbiased=False)
        a = torch.relu((yr - mean[None, :, None, None]) * torch.rsqrt(var + 1e-5)[None, :, None, None] * gr[None, :, None, None] + br[None, :, None, None])
        loss = (a * da.to(dtype)).sum()
        if pool:
            loss = loss + (F.max_pool2d(a, 2) * dp.to(dtype)).sum()
        loss.backward()
        return yr.grad, gr.grad, br.grad, mean.detach(), var.detach()

    dy32, dg32, db32, mean, var = ref(torch.float32)
    dy64, dg64, db64, _, _ = ref(torch.float64)
    rstd = torch.rsqrt(var + 1e-5)
    scale = gamma * rstd
    shift = beta - mean * scale
    yg, dag, dpg = nhwc(y), nhwc(da), nhwc(dp)
    t = [v.cuda() for v in (scale, shift, mean, rstd, gamma)]
    dgam, dbet = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    coef = torch.empty(3 * c, device="cuda")
    nb = lib.ustrun_bn_bwd_partials_bytes(n * h * w, c)
    part = torch.empty(nb // 4, device="cuda")
    dpp = dpg.data_ptr() if pool else None
    l.check(lib.ustrun_bn_bwd_reduce(dag.data_ptr(), dpp, yg.data_ptr(), t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                     t[3].data_ptr(), t[4].data_ptr(), n, h, w, c, dgam.data_ptr(), dbet.data_ptr(), 0,
                                     coef.data_ptr(), part.data_ptr(), nb, 0, None))
    dy = torch.empty(n, h, w, c, device="cuda")
    l.check(lib.ustrun_bn_bwd_apply(dag.data_ptr(), dpp, yg.data_ptr(), t[0].data_ptr(), t[1].data_ptr(), coef.data_ptr(),
                                    n, h, w, c, dy.data_ptr(), 0, None))
    assert rel(dgam.cpu(), dg64) < max(3 * rel(dg32, dg64), 1e-5)
    assert rel(dbet.cpu(), db64) < max(3 * rel(db32, db64), 1e-5)
    assert rel(from_nhwc(dy), dy64) < max(3 * rel(dy32, dy64), 1e-5)


@pytest.mark.parametrize("n,c,k,h,w", [(2, 64, 2, 16, 16), (1, 8, 4, 9, 7), (3, 64, 4, 12, 12)])
def test_head_fwd_bwd(n, c, k, h, w):
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(c + k)
    y = torch.randn(n, c, h, w, generator=g)
    sc, sh = 1 + 0.3 * torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g)
    wt, b = torch.randn(k, c, generator=g) / c ** 0.5, torch.randn(k, generator=g)
    dl = torch.randn(n, k, h, w, generator=g)
    ar = torch.relu(y * sc[None, :, None, None] + sh[None, :, None, None]).requires_grad_(True)
    wr, br = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.conv2d(ar, wr[:, :, None, None], br)
    ref.backward(dl)
    yg, scg, shg, wg, bg, dlg = nhwc(y), sc.cuda(), sh.cuda(), wt.cuda(), b.cuda(), dl.cuda()
    lg = torch.empty(n, k, h, w, device="cuda")
    l.check(lib.ustrun_head_fwd(yg.data_ptr(), scg.data_ptr(), shg.data_ptr(), n * h * w, h * w, c, k, wg.data_ptr(),
                                bg.data_ptr(), lg.data_ptr(), 0, None))
    assert rel(lg.cpu(), ref.detach()) < 1e-5
    da = torch.empty(n, h, w, c, device="cuda")
    dw, db = torch.empty(k, c, device="cuda"), torch.empty(k, device="cuda")
    nb = 1024 * (k * c + k) * 4
    part = torch.empty(nb // 4, device="cuda")
    l.check(lib.ustrun_head_bwd(dlg.data_ptr(), yg.data_ptr(), scg.data_ptr(), shg.data_ptr(), n * h * w, h * w, c, k,
                                wg.data_ptr(), da.data_ptr(), dw.data_ptr(), db.data_ptr(), 0, part.data_ptr(), nb, 0, None))
    assert rel(from_nhwc(da), ar.grad) < 1e-5
    assert rel(dw.cpu(), wr.grad) < 1e-5 and rel(db.cpu(), br.grad) < 1e-5


# ---- dtype = 3 (USTRUN_F32X3): f32 tensors, products as six bf16 MFMAs over three-term operand splits (csrc/x3.hip) -------------
def pack_conv_x3(w):
    l = L()
    co, ci = w.shape[:2]
    n = 9 * ci * co
    wf = torch.zeros(3 * n, device="cuda")          # the f32 pack + its three bf16 planes (2.5 n floats)
    wd = torch.zeros(3 * n, device="cuda")
    wg = w.contiguous().cuda()
    l.check(l.lib().ustrun_pack_conv3x3(wg.data_ptr(), co, ci, wf.data_ptr(), wd.data_ptr(), 3, None))
    return wf, wd


@pytest.mark.parametrize("n,c0,c1,co,h,w,exact", [(2, 64, 0, 64, 16, 16, True), (2, 64, 0, 64, 16, 16, False), (3, 64, 64, 128, 9, 21, False),
                                                  (2, 256, 0, 128, 12, 10, False), (1, 128, 0, 192, 37, 19, True), (4, 512, 0, 512, 8, 8, False)])
def test_conv3x3_f32x3(n, c0, c1, co, h, w, exact):
    """The 3x3 convolution trio under dtype 3 against torch in float64: small integers are exact (x1 = x2 = 0), random data lands
    within f32 summation noise of the exact-f32 matrix-core path -- bounded by 3x torch-f32's own distance from float64, i.e. NOT the
    4e-3 of a single bf16 rounding; BatchNorm + ReLU on load, concat with an offset window, ragged M tiles, statistics rows, the input
    gradient whole and split, the weight gradient with accumulation; the kernels that ran are the x3 ones (flag bit 29 switches them
    off: the results then come from the f32 matrix-core kernels and agree to the same bound)."""
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(c0 + co + h)
    ci = c0 + c1
    rnd = (lambda *s: torch.randint(-3, 4, s, generator=g).float()) if exact else (lambda *s: torch.randn(*s, generator=g))
    y0 = rnd(n, c0, h, w)
    sc = torch.tensor([0.5, 1.0, 2.0, -1.0])[torch.randint(0, 4, (c0,), generator=g)] if exact else 1 + 0.3 * torch.randn(c0, generator=g)
    sh = torch.randint(-1, 2, (c0,), generator=g).float() if exact else 0.2 * torch.randn(c0, generator=g)
    parts = [torch.relu(y0 * sc[None, :, None, None] + sh[None, :, None, None])]
    if c1:
        up = rnd(n, c1, h - 3, w - 2)
        parts.append(F.pad(up, [1, 1, 1, 2]))
    a = torch.cat(parts, 1)
    wt = rnd(co, ci, 3, 3) if exact else torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)
    dy = rnd(n, co, h, w)

    def ref(dt):
        ar, wr = a.to(dt).clone().requires_grad_(True), wt.to(dt).clone().requires_grad_(True)
        out = F.conv2d(ar, wr, None, 1, 1)
        out.backward(dy.to(dt))
        return out.detach(), ar.grad, wr.grad
    y64, da64, dw64 = ref(torch.float64)
    y32, da32, dw32 = ref(torch.float32)
    tol = lambda r32, r64: 1e-7 if exact else max(3 * rel(r32, r64), 2e-6)
    wf, wd = pack_conv_x3(wt)
    y0g, scg, shg = nhwc(y0), sc.cuda(), sh.cuda()
    srcs = (l.Src * 2)()
    srcs[0] = l.nhwc_src(y0g.data_ptr(), c0, h, w, scg.data_ptr(), shg.data_ptr(), relu=1)
    if c1:
        ug = nhwc(up)
        srcs[1] = l.nhwc_src(ug.data_ptr(), c1, h - 3, w - 2, off=(1, 1))
    ns = 2 if c1 else 1
    for flags in (0, 1 << 30, 1 << 29):          # halo-tiled x3 kernel; the generic x3 kernel (what the ConvTranspose pair runs); the f32 matrix cores
        old = lib.ustrun_debug_flags(flags)
        try:
            y = torch.empty(n, h, w, co, device="cuda")
            mt = lib.ustrun_conv_mtiles(n, h, w, co)
            stat = torch.zeros(mt, 2, co, device="cuda")
            l.check(lib.ustrun_conv3x3_fwd(srcs, ns, wf.data_ptr(), n, h, w, co, y.data_ptr(), stat.data_ptr(), 3, None))
            assert rel(from_nhwc(y), y64) < tol(y32, y64), (flags, rel(from_nhwc(y), y64), rel(y32, y64))
            np.testing.assert_allclose(stat[:, 0].sum(0).cpu().numpy(), y64.sum((0, 2, 3)).numpy(), rtol=1e-3, atol=2e-3)
            dyg = nhwc(dy)
            da = torch.empty(n, h, w, ci, device="cuda")
            if c1:        # split over the two sources' gradients (the second through its offset window)
                d0 = torch.empty(n, h, w, c0, device="cuda")
                d1 = torch.empty(n, h - 3, w - 2, c1, device="cuda")
                l.check(lib.ustrun_conv3x3_dgrad(dyg.data_ptr(), wd.data_ptr(), n, h, w, co, ci, d0.data_ptr(), c0, d1.data_ptr(), h - 3, w - 2, 1, 1, 3, None))
                assert rel(from_nhwc(d0), da64[:, :c0]) < tol(da32[:, :c0], da64[:, :c0])
                assert rel(from_nhwc(d1), da64[:, c0:, 1:h - 2, 1:w - 1]) < tol(da32[:, c0:, 1:h - 2, 1:w - 1], da64[:, c0:, 1:h - 2, 1:w - 1])
            else:
                l.check(lib.ustrun_conv3x3_dgrad(dyg.data_ptr(), wd.data_ptr(), n, h, w, co, ci, da.data_ptr(), ci, None, 0, 0, 0, 0, 3, None))
                assert rel(from_nhwc(da), da64) < tol(da32, da64), (flags, rel(from_nhwc(da), da64))
            nb = lib.ustrun_wgrad_partials_bytes(9, ci, co, n * h * w)
            part = torch.empty(nb // 4, device="cuda")
            dw = torch.empty(co, ci, 3, 3, device="cuda")
            l.check(lib.ustrun_conv3x3_wgrad(srcs, ns, dyg.data_ptr(), n, h, w, co, dw.data_ptr(), 0, part.data_ptr(), nb, 3, None))
            assert rel(dw.cpu(), dw64) < tol(dw32, dw64), (flags, rel(dw.cpu(), dw64), rel(dw32, dw64))
            l.check(lib.ustrun_conv3x3_wgrad(srcs, ns, dyg.data_ptr(), n, h, w, co, dw.data_ptr(), 1, part.data_ptr(), nb, 3, None))
            assert rel(dw.cpu(), 2 * dw64) < tol(dw32, dw64)
        finally:
            lib.ustrun_debug_flags(old)


@pytest.mark.parametrize("n,c,h,w", [(2, 3, 16, 32), (1, 1, 19, 37), (3, 3, 40, 33), (4, 3, 136, 96), (8, 3, 256, 256)])
def test_conv_first_weight_gradient_f32x3(n, c, h, w):
    """The first convolution's weight gradient under dtype 3 (f32 dY): the streaming kernel with three-term products against
    torch in float64 -- random data within f32 summation noise (3x torch-f32's own distance), integers exact."""
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(5 * c + w)
    for exact in (True, False):
        x = torch.randint(-3, 4, (n, c, h, w), generator=g).float() if exact else torch.randn(n, c, h, w, generator=g)
        dy = torch.randint(-2, 3, (n, 64, h, w), generator=g).float() if exact else torch.randn(n, 64, h, w, generator=g)
        def ref(dt):
            wr = torch.zeros(64, c, 3, 3, dtype=dt, requires_grad=True)
            F.conv2d(x.to(dt), wr, None, 1, 1).backward(dy.to(dt))
            return wr.grad
        g64, g32 = ref(torch.float64), ref(torch.float32)
        xg, dyg = x.contiguous().cuda(), nhwc(dy)
        src = l.nchw_src(xg.data_ptr(), c, h, w)
        nb = lib.ustrun_wgrad_partials_bytes(9, c, 64, n * h * w)
        part = torch.empty(nb // 4, device="cuda")
        dw = torch.empty(64, c, 3, 3, device="cuda")
        l.check(lib.ustrun_conv3x3_wgrad(C.byref(src), 1, dyg.data_ptr(), n, h, w, 64, dw.data_ptr(), 0, part.data_ptr(), nb, 3, None))
        tol = 1e-7 if exact else max(3 * rel(g32, g64), 2e-6)
        assert rel(dw.cpu(), g64) < tol, (exact, rel(dw.cpu(), g64), rel(g32, g64))


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 128, 64, 5, 7), (3, 256, 128, 12, 10), (1, 64, 64, 3, 50), (2, 256, 128, 32, 32), (2, 1024, 512, 4, 4)])
def test_convT2x2_f32x3(n, ci, co, h, w):
    """The ConvTranspose trio under dtype 3: forward and input gradient on the generic three-term kernel, the weight gradient on
    its own (four parity classes as four accumulators), bias gradient; float64 torch as the reference, f32 summation noise as the bound."""
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(ci + 3 * w)
    y = torch.randn(n, ci, h, w, generator=g)
    sc, sh = 1 + 0.3 * torch.randn(ci, generator=g), 0.2 * torch.randn(ci, generator=g)
    a = torch.relu(y * sc[None, :, None, None] + sh[None, :, None, None])
    wt, b = torch.randn(ci, co, 2, 2, generator=g) / ci ** 0.5, torch.randn(co, generator=g)
    du = torch.randn(n, co, 2 * h, 2 * w, generator=g)

    def ref(dt):
        ar, wr, br = a.to(dt).clone().requires_grad_(True), wt.to(dt).clone().requires_grad_(True), b.to(dt).clone().requires_grad_(True)
        out = F.conv_transpose2d(ar, wr, br, stride=2)
        out.backward(du.to(dt))
        return out.detach(), ar.grad, wr.grad, br.grad
    r64, r32 = ref(torch.float64), ref(torch.float32)
    tol = lambda k: max(3 * rel(r32[k], r64[k]), 2e-6)
    nel = 4 * ci * co
    wf, wd = torch.zeros(3 * nel, device="cuda"), torch.zeros(3 * nel, device="cuda")
    wg = wt.cuda()
    l.check(lib.ustrun_pack_convT2x2(wg.data_ptr(), ci, co, wf.data_ptr(), wd.data_ptr(), 3, None))
    yg, scg, shg, bg, dug = nhwc(y), sc.cuda(), sh.cuda(), b.cuda(), nhwc(du)
    src = l.nhwc_src(yg.data_ptr(), ci, h, w, scg.data_ptr(), shg.data_ptr(), relu=1)
    u = torch.empty(n, 2 * h, 2 * w, co, device="cuda")
    l.check(lib.ustrun_convT2x2_fwd(C.byref(src), wf.data_ptr(), bg.data_ptr(), n, h, w, co, u.data_ptr(), 3, None))
    assert rel(from_nhwc(u), r64[0]) < tol(0)
    da = torch.empty(n, h, w, ci, device="cuda")
    l.check(lib.ustrun_convT2x2_dgrad(dug.data_ptr(), wd.data_ptr(), n, h, w, co, ci, da.data_ptr(), 3, None))
    assert rel(from_nhwc(da), r64[1]) < tol(1)
    nb = max(lib.ustrun_wgrad_partials_bytes(4, ci, co, n * h * w), 512 * co * 4)
    part = torch.full((nb // 4 + 256,), 5.0, device="cuda")
    dw, db = torch.empty(ci, co, 2, 2, device="cuda"), torch.empty(co, device="cuda")
    l.check(lib.ustrun_convT2x2_wgrad(C.byref(src), dug.data_ptr(), n, h, w, co, dw.data_ptr(), db.data_ptr(), 0, part.data_ptr(), nb, 3, None))
    assert rel(dw.cpu(), r64[2]) < tol(2), (rel(dw.cpu(), r64[2]), rel(r32[2], r64[2]))
    assert rel(db.cpu(), r64[3]) < max(tol(3), 1e-5)
    assert bool((part[nb // 4:] == 5.0).all()), "slabs beyond the published partials bound"
    l.check(lib.ustrun_convT2x2_wgrad(C.byref(src), dug.data_ptr(), n, h, w, co, dw.data_ptr(), db.data_ptr(), 1, part.data_ptr(), nb, 3, None))
    assert rel(dw.cpu(), 2 * r64[2]) < tol(2)


# ---- dtype = 1: bf16 tensors in HBM, bf16 MFMA operands, f32 accumulate/statistics -------------
def nhwc16(t):
    return nhwc(t).bfloat16()


def r16(t):          # what an exact f32 result reads back as after bf16 storage
    return t.bfloat16().float()

def pack_conv_bf16(w):
    l = L()
    co, ci = w.shape[:2]
    n = 9 * ((ci + 7) // 8 * 8) * ((co + 7) // 8 * 8)
    wf = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    wd = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    wg = w.contiguous().cuda()
    l.check(l.lib().ustrun_pack_conv3x3(wg.data_ptr(), co, ci, wf.data_ptr(), wd.data_ptr(), 1, None))
    return wf, wd


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 64, 64, 16, 16), (1, 24, 40, 9, 7), (2, 256, 128, 8, 8), (2, 128, 256, 6, 10),
                                         (2, 3, 64, 16, 16), (1, 192, 64, 12, 12)])
@pytest.mark.parametrize("exact", [True, False])
def test_conv3x3_bf16_mfma(n, ci, co, h, w, exact):
    """exact=True: small-integer data is exactly representable in bf16 and sums exactly in f32, so any
    fragment-layout or indexing slip shows as an O(1) error; exact=False: random data, bf16 rounding."""
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(ci * 3 + co)
    if exact:
        x = torch.randint(-3, 4, (n, ci, h, w), generator=g).float()
        wt = torch.randint(-2, 3, (co, ci, 3, 3), generator=g).float()
        dy = torch.randint(-3, 4, (n, co, h, w), generator=g).float()
        tol = 1e-6
    else:
        x = torch.randn(n, ci, h, w, generator=g).bfloat16().float()
        wt = torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)
        dy = torch.randn(n, co, h, w, generator=g).bfloat16().float()
        tol = 1e-2
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, 1, 1)
    y_ref.backward(dy)
    wf, wd = pack_conv_bf16(wt)
    xg, dyg = nhwc16(x), nhwc16(dy)
    src = l.nhwc_src(xg.data_ptr(), ci, h, w)
    y = torch.empty(n, h, w, co, device="cuda", dtype=torch.bfloat16)
    stat = torch.zeros(lib.ustrun_conv_mtiles(n, h, w, co), 2, co, device="cuda")
    l.check(lib.ustrun_conv3x3_fwd(C.byref(src), 1, wf.data_ptr(), n, h, w, co, y.data_ptr(), stat.data_ptr(), 1, None))
    assert rel(from_nhwc(y.float()), r16(y_ref.detach())) < tol
    np.testing.assert_allclose(stat[:, 0].sum(0).cpu().numpy(), from_nhwc(y.float()).sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-2)
    da = torch.empty(n, h, w, ci, device="cuda", dtype=torch.bfloat16)
    l.check(lib.ustrun_conv3x3_dgrad(dyg.data_ptr(), wd.data_ptr(), n, h, w, co, ci, da.data_ptr(), ci, None, 0, 0, 0, 0, 1, None))
    assert rel(from_nhwc(da.float()), r16(xr.grad)) < tol
    nb = lib.ustrun_wgrad_partials_bytes(9, ci, co, n * h * w)
    part = torch.empty(nb // 4, device="cuda")
    dw = torch.empty(co, ci, 3, 3, device="cuda")
    l.check(lib.ustrun_conv3x3_wgrad(C.byref(src), 1, dyg.data_ptr(), n, h, w, co, dw.data_ptr(), 0, part.data_ptr(), nb, 1, None))
    assert rel(dw.cpu(), wr.grad) < tol


@pytest.mark.parametrize("elt", ["bf16", "f16"])
@pytest.mark.parametrize("flags", [0, 1 << 28])
@pytest.mark.parametrize("n,c,h,w", [(2, 3, 16, 32), (1, 1, 19, 37), (2, 4, 9, 70), (3, 3, 40, 33), (4, 3, 136, 96), (16, 1, 72, 100),
                                     (64, 3, 64, 64), (8, 3, 256, 256), (5, 1, 288, 288)])
def test_conv_first_weight_gradient_exact(n, c, h, w, flags, elt):
    """Weight gradient of the first convolution (NCHW f32 network input, C <= 4 -> 64; autograd of nn.Conv2d at unet_parts.py:16 for
    `inc`): the streaming kernel of round 5 (a wave walks a 16-pixel strip a row a step, dY and the lane's im2col row of x fetched
    four steps ahead through range-checked buffer loads, masks applied at use) and, with ustrun_debug_flags bit 28, the tile kernel
    of rounds 1-4.  Small-integer data is exact: image borders (rows above / below, columns left / right), ragged strips (w % 16),
    segments that do not divide the height, im2col rows past 9 C and accumulation all show as O(1) errors; dw sits between sentinels."""
    l = L()
    lib = l.lib()
    t16, code = (torch.bfloat16, 1) if elt == "bf16" else (torch.float16, 2)
    g = torch.Generator().manual_seed(7 * c + w)
    x = torch.randint(-3, 4, (n, c, h, w), generator=g).float()
    dy = torch.randint(-2, 3, (n, 64, h, w), generator=g).float()
    wr = torch.zeros(64, c, 3, 3, requires_grad=True)
    F.conv2d(x, wr, None, 1, 1).backward(dy)
    xg = x.contiguous().cuda()
    dyg = dy.permute(0, 2, 3, 1).contiguous().cuda().to(t16)
    src = l.nchw_src(xg.data_ptr(), c, h, w)
    nb = lib.ustrun_wgrad_partials_bytes(9, c, 64, n * h * w)
    part = torch.full((nb // 4 + 1024,), 5.0, device="cuda")
    Z = 512
    buf = torch.full((Z + 64 * c * 9 + Z,), 9.0, device="cuda")
    dw = buf[Z:Z + 64 * c * 9].view(64, c, 3, 3)
    old = lib.ustrun_debug_flags(flags)
    try:
        l.check(lib.ustrun_conv3x3_wgrad(C.byref(src), 1, dyg.data_ptr(), n, h, w, 64, dw.data_ptr(), 0, part.data_ptr(), nb, code, None))
        assert rel(dw.cpu(), wr.grad) < 1e-6, rel(dw.cpu(), wr.grad)
        l.check(lib.ustrun_conv3x3_wgrad(C.byref(src), 1, dyg.data_ptr(), n, h, w, 64, dw.data_ptr(), 1, part.data_ptr(), nb, code, None))
        assert rel(dw.cpu(), 2 * wr.grad) < 1e-6
    finally:
        lib.ustrun_debug_flags(old)
    assert bool((buf[:Z] == 9.0).all()) and bool((buf[-Z:] == 9.0).all())
    assert bool((part[nb // 4:] == 5.0).all()), "slabs beyond the published partials bound"


@pytest.mark.parametrize("with_stat", [True, False])
@pytest.mark.parametrize("elt", ["bf16", "f16"])
@pytest.mark.parametrize("flags", [0, 16384])
@pytest.mark.parametrize("n,c,h,w", [(2, 3, 16, 32), (1, 1, 19, 37), (2, 4, 9, 70), (3, 3, 40, 33), (4, 3, 136, 96), (16, 1, 72, 100),
                                     (2, 2, 200, 64), (32, 3, 256, 256), (24, 3, 256, 256), (40, 1, 288, 288)])
def test_conv_first_bf16_mfma_exact(n, c, h, w, flags, elt, with_stat):
    """First convolution (NCHW f32 network input, C <= 4 -> 64) on the 16-bit matrix cores as an im2col GEMM, both builds (the
    streaming kernel of round 4: a block walks a 32-pixel strip 8 rows per step -- several steps, a ragged last step, ragged
    strips, segments of 8..64 rows; and, with ustrun_debug_flags bit 14, the tile-per-block kernel of rounds 1-3), both
    element types: small-integer data is exact, so a slip in the k -> (channel, tap) gather, the patch ring or the ragged
    masks shows as an O(1) error; BatchNorm-statistics partials must be the sums of the stored outputs PER IMAGE ROW RANGE
    (rows of one image are contiguous: batched passes split the table by rows); nothing outside the output or beyond the
    reported row count is written."""
    l = L()
    lib = l.lib()
    t16, code = (torch.bfloat16, 1) if elt == "bf16" else (torch.float16, 2)
    g = torch.Generator().manual_seed(11 * c + h)
    x = torch.randint(-3, 4, (n, c, h, w), generator=g).float()
    wt = torch.randint(-2, 3, (64, c, 3, 3), generator=g).float()
    y_ref = F.conv2d(x, wt, None, 1, 1)
    nel = 9 * 8 * 64
    wf, wd = torch.zeros(nel, dtype=t16, device="cuda"), torch.zeros(nel, dtype=t16, device="cuda")
    wg = wt.cuda()
    l.check(lib.ustrun_pack_conv3x3(wg.data_ptr(), 64, c, wf.data_ptr(), wd.data_ptr(), code, None))
    xg = x.contiguous().cuda()
    src = l.nchw_src(xg.data_ptr(), c, h, w)
    Z = 4096
    buf = torch.full((Z + n * h * w * 64 + Z,), 9.0, device="cuda", dtype=t16)
    y = buf[Z:Z + n * h * w * 64].view(n, h, w, 64)
    rows_max = lib.ustrun_conv_mtiles(n, h, w, 64)
    stat = torch.full((rows_max + 16, 2, 64), 7.0, device="cuda")
    rows = C.c_int(0)
    old = lib.ustrun_debug_flags(flags)
    try:
        l.check(lib.ustrun_conv3x3_fwd_rows(C.byref(src), 1, wf.data_ptr(), n, h, w, 64, y.data_ptr(), stat.data_ptr() if with_stat else None,
                                            C.byref(rows), code, None))
    finally:
        lib.ustrun_debug_flags(old)
    yc = from_nhwc(y.float())
    assert rel(yc, y_ref.to(t16).float()) < 1e-6
    assert bool((buf[:Z] == 9.0).all()) and bool((buf[-Z:] == 9.0).all())
    if not with_stat:                       # the eval-mode forward: no statistics (its own build of the kernel); nothing written to the table
        assert bool((stat == 7.0).all())
        return
    assert 0 < rows.value <= rows_max and rows.value % n == 0
    assert bool((stat[rows.value:] == 7.0).all())
    st = stat[:rows.value].view(n, rows.value // n, 2, 64).sum(1).cpu()            # per image
    np.testing.assert_allclose(st[:, 0].numpy(), yc.sum((2, 3)).numpy(), rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(st[:, 1].numpy(), (yc * yc).sum((2, 3)).numpy(), rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 128, 64, 5, 7), (3, 256, 128, 12, 10), (1, 64, 64, 3, 50), (4, 256, 128, 128, 128),
                                         (2, 40, 24, 6, 5)])
def test_convT2x2_bf16_mfma_exact(n, ci, co, h, w):
    """Dedicated GEMM kernels (convT_bf16.hip) for 64-multiple channels -- ragged pixel counts, odd widths,
    the 256-column dgrad tile at the large size -- and the generic fallback for the last shape."""
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-3, 4, (n, ci, h, w), generator=g).float()
    wt = torch.randint(-2, 3, (ci, co, 2, 2), generator=g).float()
    b = torch.randint(-2, 3, (co,), generator=g).float()
    du = torch.randint(-3, 4, (n, co, 2 * h, 2 * w), generator=g).float()
    # producer's BatchNorm affine + ReLU on load: power-of-two scales and integer shifts keep everything exact
    sc = torch.tensor([0.5, 1.0, 2.0])[torch.randint(0, 3, (ci,), generator=g)]
    sh = torch.randint(-1, 2, (ci,), generator=g).float()
    xa = torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None])
    xr, wr, br = xa.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    u_ref = F.conv_transpose2d(xr, wr, br, stride=2)
    u_ref.backward(du)
    nel = 4 * ((ci + 7) // 8 * 8) * ((co + 7) // 8 * 8)
    wf = torch.zeros(nel, dtype=torch.bfloat16, device="cuda")
    wd = torch.zeros(nel, dtype=torch.bfloat16, device="cuda")
    wg = wt.cuda()
    l.check(lib.ustrun_pack_convT2x2(wg.data_ptr(), ci, co, wf.data_ptr(), wd.data_ptr(), 1, None))
    xg, bg, dug = nhwc16(x), b.cuda(), nhwc16(du)
    scg, shg = sc.cuda(), sh.cuda()
    src = l.nhwc_src(xg.data_ptr(), ci, h, w, scale=scg.data_ptr(), shift=shg.data_ptr(), relu=1)
    u = torch.empty(n, 2 * h, 2 * w, co, device="cuda", dtype=torch.bfloat16)
    l.check(lib.ustrun_convT2x2_fwd(C.byref(src), wf.data_ptr(), bg.data_ptr(), n, h, w, co, u.data_ptr(), 1, None))
    assert rel(from_nhwc(u.float()), r16(u_ref.detach())) < 1e-6
    da = torch.empty(n, h, w, ci, device="cuda", dtype=torch.bfloat16)
    l.check(lib.ustrun_convT2x2_dgrad(dug.data_ptr(), wd.data_ptr(), n, h, w, co, ci, da.data_ptr(), 1, None))
    assert rel(from_nhwc(da.float()), r16(xr.grad)) < 1e-6
    nb = max(lib.ustrun_wgrad_partials_bytes(4, ci, co, n * h * w), 512 * co * 4)
    part = torch.empty(nb // 4, device="cuda")
    dw, db = torch.empty(ci, co, 2, 2, device="cuda"), torch.empty(co, device="cuda")
    l.check(lib.ustrun_convT2x2_wgrad(C.byref(src), dug.data_ptr(), n, h, w, co, dw.data_ptr(), db.data_ptr(), 0, part.data_ptr(), nb, 1, None))
    assert rel(dw.cpu(), wr.grad) < 1e-6 and rel(db.cpu(), br.grad) < 1e-6


def test_halo_bf16_pool_concat_pad_and_split_dgrad():
    """The halo-tiled bf16 kernel with a pooled source, with a two-source concat + pad offset, with
    statistics on an extent that is not a multiple of the tile, and with the split input-gradient."""
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(3)
    n, c0, c1, co, h, w = 2, 64, 64, 128, 11, 21
    sc, sh = torch.randint(-2, 3, (c0,), generator=g).float(), torch.randint(-2, 3, (c0,), generator=g).float()
    scg, shg = sc.cuda(), sh.cuda()
    ri = lambda *s: torch.randint(-2, 3, s, generator=g).float()
    # pooled source (integers stay exact through affine/relu/max and bf16)
    ys = ri(n, c0, 2 * h + 1, 2 * w)
    wt = ri(co, c0, 3, 3)
    a = F.max_pool2d(torch.relu(ys * sc[None, :, None, None] + sh[None, :, None, None]), 2)
    ref = F.conv2d(a, wt, None, 1, 1)
    wf, _ = pack_conv_bf16(wt)
    yg = nhwc16(ys)
    src = l.nhwc_src(yg.data_ptr(), c0, 2 * h + 1, 2 * w, scg.data_ptr(), shg.data_ptr(), relu=1, pool=1)
    out = torch.empty(n, h, w, co, device="cuda", dtype=torch.bfloat16)
    rows = lib.ustrun_conv_mtiles(n, h, w, co)
    stat = torch.full((rows, 2, co), 9.0, device="cuda")
    l.check(lib.ustrun_conv3x3_fwd(C.byref(src), 1, wf.data_ptr(), n, h, w, co, out.data_ptr(), stat.data_ptr(), 1, None))
    assert rel(from_nhwc(out.float()), r16(ref)) < 1e-6
    np.testing.assert_allclose(stat[:, 0].sum(0).cpu().numpy(), r16(ref).sum((0, 2, 3)).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(stat[:, 1].sum(0).cpu().numpy(), r16(ref).square().sum((0, 2, 3)).numpy(), rtol=1e-5, atol=1e-3)
    # concat [skip (affine+relu), up (offset-padded, smaller extent)]
    skip, up = ri(n, c0, h, w), ri(n, c1, h - 3, w - 2)
    wt2 = ri(co, c0 + c1, 3, 3)
    a2 = torch.cat([torch.relu(skip * sc[None, :, None, None] + sh[None, :, None, None]), F.pad(up, [1, 1, 1, 2])], 1)
    a2r = a2.clone().requires_grad_(True)
    ref2 = F.conv2d(a2r, wt2, None, 1, 1)
    dy = ri(n, co, h, w)
    ref2.backward(dy)
    wf2, wd2 = pack_conv_bf16(wt2)
    sg, ug = nhwc16(skip), nhwc16(up)
    srcs = (l.Src * 2)(l.nhwc_src(sg.data_ptr(), c0, h, w, scg.data_ptr(), shg.data_ptr(), relu=1),
                       l.nhwc_src(ug.data_ptr(), c1, h - 3, w - 2, off=(1, 1)))
    out2 = torch.empty(n, h, w, co, device="cuda", dtype=torch.bfloat16)
    l.check(lib.ustrun_conv3x3_fwd(srcs, 2, wf2.data_ptr(), n, h, w, co, out2.data_ptr(), None, 1, None))
    assert rel(from_nhwc(out2.float()), r16(ref2.detach())) < 1e-6
    # input gradient split into the skip part and the (offset, smaller) up part
    dyg = nhwc16(dy)
    d0 = torch.empty(n, h, w, c0, device="cuda", dtype=torch.bfloat16)
    d1 = torch.empty(n, h - 3, w - 2, c1, device="cuda", dtype=torch.bfloat16)
    l.check(lib.ustrun_conv3x3_dgrad(dyg.data_ptr(), wd2.data_ptr(), n, h, w, co, c0 + c1, d0.data_ptr(), c0, d1.data_ptr(),
                                     h - 3, w - 2, 1, 1, 1, None))
    assert rel(from_nhwc(d0.float()), r16(a2r.grad[:, :c0])) < 1e-6
    assert rel(from_nhwc(d1.float()), r16(a2r.grad[:, c0:, 1:h - 2, 1:w - 1])) < 1e-6


def test_bf16_storage_bn_head_backward():
    """BatchNorm/ReLU/pool backward and the head on bf16 tensors: same math as the f32 kernels, inputs and
    outputs rounded to bf16 (compared against the f32 kernels run on the rounded inputs)."""
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(8)
    n, c, h, w, k = 2, 64, 12, 16, 2
    y = (torch.randn(n, c, h, w, generator=g) * 2 + 0.5).bfloat16().float()
    da = torch.randn(n, c, h, w, generator=g).bfloat16().float()
    dp = torch.randn(n, c, h // 2, w // 2, generator=g).bfloat16().float()
    var, mean = torch.var_mean(y, dim=(0, 2, 3), unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    gamma, beta = 1 + 0.3 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    t = [v.cuda() for v in (gamma * rstd, beta - mean * gamma * rstd, mean, rstd, gamma)]
    nb = lib.ustrun_bn_bwd_partials_bytes(n * h * w, c)
    part = torch.empty(nb // 4, device="cuda")
    outs = {}
    for dt, cast in ((0, lambda v: v), (1, lambda v: v.bfloat16())):
        yg, dag, dpg = cast(nhwc(y)), cast(nhwc(da)), cast(nhwc(dp))
        dgam, dbet, coef = torch.empty(c, device="cuda"), torch.empty(c, device="cuda"), torch.empty(3 * c, device="cuda")
        l.check(lib.ustrun_bn_bwd_reduce(dag.data_ptr(), dpg.data_ptr(), yg.data_ptr(), t[0].data_ptr(), t[1].data_ptr(),
                                         t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(), n, h, w, c, dgam.data_ptr(),
                                         dbet.data_ptr(), 0, coef.data_ptr(), part.data_ptr(), nb, dt, None))
        dy = torch.empty_like(yg)
        l.check(lib.ustrun_bn_bwd_apply(dag.data_ptr(), dpg.data_ptr(), yg.data_ptr(), t[0].data_ptr(), t[1].data_ptr(),
                                        coef.data_ptr(), n, h, w, c, dy.data_ptr(), dt, None))
        wt, b = torch.randn(k, c, generator=torch.Generator().manual_seed(1)).cuda() / 8, torch.zeros(k).cuda()
        lg = torch.empty(n, k, h, w, device="cuda")
        l.check(lib.ustrun_head_fwd(yg.data_ptr(), t[0].data_ptr(), t[1].data_ptr(), n * h * w, h * w, c, k, wt.data_ptr(),
                                    b.data_ptr(), lg.data_ptr(), dt, None))
        outs[dt] = (dgam.cpu(), dbet.cpu(), dy.float().cpu(), lg.cpu())
    assert rel(outs[1][0], outs[0][0]) < 1e-5 and rel(outs[1][1], outs[0][1]) < 1e-5
    assert rel(outs[1][2], outs[0][2].bfloat16().float()) < 1e-5
    assert rel(outs[1][3], outs[0][3]) < 1e-6


@pytest.mark.parametrize("n,c,h,w,pool,flags", [
    (2, 64, 12, 16, False, 0),        # plain, 8 channels per lane (G8 = 8)
    (3, 8, 9, 7, False, 0),           # plain, one octet (G8 = 1), odd extent
    (2, 1024, 6, 10, False, 0),       # plain, G8 = 128
    (2, 2048, 3, 5, False, 0),        # plain, G8 = 256 (ResNet layer4)
    (2, 96, 8, 8, False, 0),          # 12 octets: not a power of two -> the 4-channel kernels
    (2, 64, 12, 16, False, 4096),     # ... which bit 12 selects for any C
    (2, 64, 12, 16, True, 0),         # pooled windows of an even-sized map (no validity selects)
    (2, 128, 10, 14, True, 0),
    (2, 64, 11, 15, True, 0),         # odd-sized map: the general pooled form
])
def test_bn_backward_bf16_builds(n, c, h, w, pool, flags):
    """Every build of the BatchNorm + ReLU (+ MaxPool routing) backward on bf16 tensors: against torch autograd in float64 through
    `F.batch_norm -> relu (-> max_pool2d)` on the same (bf16-rounded) inputs (reference networks/unet_parts.py:17-18,34), and
    against the f32 kernels; the build each launch ran is asserted (`ustrun_debug_last_bn_variant`)."""
    import torch.nn.functional as F
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(c + h + w)
    y = (torch.randn(n, c, h, w, generator=g) * 2 + 0.5).bfloat16().float()
    da = torch.randn(n, c, h, w, generator=g).bfloat16().float()
    dp = torch.randn(n, c, h // 2, w // 2, generator=g).bfloat16().float()
    var, mean = torch.var_mean(y, dim=(0, 2, 3), unbiased=False)
    rstd = torch.rsqrt(var + 1e-5)
    gamma, beta = 1 + 0.3 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    t = [v.cuda() for v in (gamma * rstd, beta - mean * gamma * rstd, mean, rstd, gamma)]
    nb = lib.ustrun_bn_bwd_partials_bytes(n * h * w, c)
    part = torch.empty(nb // 4, device="cuda")
    # expectation: autograd in float64 (batch statistics of y itself = the mean / rstd handed to the kernels)
    yd, gd, bd = y.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    act = torch.relu(F.batch_norm(yd, None, None, gd, bd, True, 0.0, 1e-5))
    obj = (act * da.double()).sum()
    if pool:
        obj = obj + (F.max_pool2d(act, 2) * dp.double()).sum()
    obj.backward()
    outs = {}
    old = lib.ustrun_debug_flags(flags)
    try:
        for dt, cast in ((0, lambda v: v), (1, lambda v: v.bfloat16())):
            yg, dag, dpg = cast(nhwc(y)), cast(nhwc(da)), cast(nhwc(dp))
            dpp = dpg.data_ptr() if pool else None
            dgam, dbet, coef = torch.empty(c, device="cuda"), torch.empty(c, device="cuda"), torch.empty(3 * c, device="cuda")
            l.check(lib.ustrun_bn_bwd_reduce(dag.data_ptr(), dpp, yg.data_ptr(), t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                             t[3].data_ptr(), t[4].data_ptr(), n, h, w, c, dgam.data_ptr(), dbet.data_ptr(), 0,
                                             coef.data_ptr(), part.data_ptr(), nb, dt, None))
            g8 = c // 8
            x8 = dt == 1 and not pool and not (flags & 4096) and c % 8 == 0 and g8 <= 256 and (g8 & (g8 - 1)) == 0
            code = 0x424E0000 | (2 if dt else 4) << 4 | (4 if pool and h % 2 == 0 and w % 2 == 0 else 0) | (2 if pool else 0) | (1 if x8 else 0)
            assert lib.ustrun_debug_last_bn_variant() == code, hex(lib.ustrun_debug_last_bn_variant())
            buf = torch.full((4096 + yg.numel() + 4096,), 7.0, device="cuda", dtype=yg.dtype)
            dy = buf[4096:4096 + yg.numel()].view_as(yg)
            l.check(lib.ustrun_bn_bwd_apply(dag.data_ptr(), dpp, yg.data_ptr(), t[0].data_ptr(), t[1].data_ptr(), coef.data_ptr(),
                                            n, h, w, c, dy.data_ptr(), dt, None))
            assert lib.ustrun_debug_last_bn_variant() == code | 1 << 8, hex(lib.ustrun_debug_last_bn_variant())
            assert bool((buf[:4096] == 7.0).all()) and bool((buf[-4096:] == 7.0).all())
            outs[dt] = (dgam.cpu(), dbet.cpu(), dy.float().cpu(), coef.cpu())
    finally:
        lib.ustrun_debug_flags(old)
    from_nhwc = lambda v: v.permute(0, 3, 1, 2)
    for dt in (0, 1):                       # both dtypes against autograd: parameter gradients 2e-5, dy to the stored precision
        assert rel(outs[dt][0], gd.grad) < 2e-5 and rel(outs[dt][1], bd.grad) < 2e-5, dt
        assert rel(from_nhwc(outs[dt][2]), yd.grad) < (2e-5 if dt == 0 else 4e-3), dt
    assert rel(outs[1][0], outs[0][0]) < 1e-5 and rel(outs[1][1], outs[0][1]) < 1e-5 and rel(outs[1][3], outs[0][3]) < 1e-5
    # the bf16 pass rounds once; its f32 value may differ from the f32 kernel's by the coefficient error above -> one bf16 ulp
    want = outs[0][2]
    err = (outs[1][2] - want).abs()
    assert bool((err <= want.abs() * 2 ** -7 + 1e-6).all()), float((err / (want.abs() + 1e-6)).max())
    assert rel(outs[1][2], want.bfloat16().float()) < 2e-3


@pytest.mark.parametrize("k", [1, 2, 3, 4])
@pytest.mark.parametrize("dt,tdt", [(1, torch.bfloat16), (2, torch.float16)])
def test_head_fwd_64_channel_kernel_is_bit_identical(k, dt, tdt):
    """ustrun_head_fwd on 16-bit storage with C = 64 and H*W % 256 == 0 runs head_fwd_bf16_g8_kernel (round 4: one scalar divide per
    256-pixel trip, the class sums reduce-scattered over the pixel's eight lanes); ustrun_debug_flags bit 27 keeps the generic
    kernel.  Same pairing of the same additions: the logits must be bit-identical, with and without BatchNorm constants, and match
    torch on the rounded activations."""
    l = L()
    lib = l.lib()
    n, c, h, w = 3, 64, 32, 40          # 1280 pixels per image
    g = torch.Generator().manual_seed(40 + k)
    y = torch.randn(n, h, w, c, generator=g).cuda().to(tdt)
    sc, sh = (torch.rand(c, generator=g) + 0.5).cuda(), (torch.randn(c, generator=g) * 0.3).cuda()
    wt, b = (torch.randn(k, c, generator=g) / 8).cuda(), torch.randn(k, generator=g).cuda()
    for aff in (True, False):
        outs = []
        for flags in (0, 1 << 27):
            old = lib.ustrun_debug_flags(flags)
            try:
                lg = torch.full((n, k, h, w), 7.0, device="cuda")
                l.check(lib.ustrun_head_fwd(y.data_ptr(), sc.data_ptr() if aff else None, sh.data_ptr() if aff else None, n * h * w, h * w, c, k,
                                            wt.data_ptr(), b.data_ptr(), lg.data_ptr(), dt, None))
            finally:
                lib.ustrun_debug_flags(old)
            outs.append(lg)
        assert torch.equal(outs[0], outs[1])
        a = y.float()
        if aff:
            a = torch.relu(a * sc + sh)
        ref = torch.einsum("nhwc,kc->nkhw", a.double(), wt.double()) + b.double()[None, :, None, None]
        assert rel(outs[0].cpu(), ref.cpu()) < 1e-6
