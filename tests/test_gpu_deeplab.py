"""GPU parity of the DeepLabV2-ResNet forward (reference networks/deeplabv2.py:10-33,
networks/backbone/resnet.py:55-176): the new operators against torch-CPU, the network against the CPU oracle
(oracle/deeplab_ref.py) and against outputs captured from the reference itself (G10).  Backward: test_gpu_deeplab_bwd.py."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu


def L():
    from ustrun import _lib
    return _lib


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def nhwc(t, dt):
    t = t.permute(0, 2, 3, 1).contiguous().cuda()
    return t.bfloat16() if dt else t


def from_nhwc(t):
    return t.float().permute(0, 3, 1, 2).contiguous().cpu()


@pytest.mark.parametrize("dt", [0, 1])
@pytest.mark.parametrize("n,ci,co,h,w,k,s,d,bias", [(2, 64, 64, 17, 23, 1, 1, 1, False), (2, 256, 128, 16, 12, 1, 2, 1, False),
                                                    (1, 64, 64, 19, 21, 3, 2, 1, False), (2, 128, 128, 14, 18, 3, 1, 2, False),
                                                    (2, 256, 256, 12, 12, 3, 1, 4, False), (1, 128, 128, 24, 20, 3, 1, 4, False), (1, 512, 2, 16, 20, 3, 1, 6, True),
                                                    (1, 512, 4, 9, 9, 3, 1, 24, True)])
def test_conv2d_general(n, ci, co, h, w, k, s, d, bias, dt):
    """1x1 / 3x3 with stride 2 and dilation 2, 4, 6, 24 (padding = dilation, most taps out of the image at rate 24), bias,
    Cout = 2: exact small integers in both dtypes, plus the BatchNorm-statistics rows."""
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(ci + co + k + d)
    x = torch.randint(-3, 4, (n, ci, h, w), generator=g).float()
    wt = torch.randint(-2, 3, (co, ci, k, k), generator=g).float()
    b = torch.randint(-2, 3, (co,), generator=g).float() if bias else None
    ref = F.conv2d(x, wt, b, s, d * (k // 2), d)
    ho, wo = ref.shape[-2:]
    wf = torch.zeros(lib.ustrun_pack_conv_elems(co, ci, k * k), dtype=torch.bfloat16 if dt else torch.float32, device="cuda")
    wg = wt.cuda()
    l.check(lib.ustrun_pack_conv(wg.data_ptr(), co, ci, k * k, wf.data_ptr(), dt, None))
    xg = nhwc(x, dt)
    src = l.nhwc_src(xg.data_ptr(), ci, h, w)
    y = torch.empty(n, ho, wo, co, device="cuda", dtype=torch.bfloat16 if dt else torch.float32)
    rows = lib.ustrun_conv_mtiles(n, ho, wo, co)           # (the bound the caller must allocate; `used` rows are written)
    stat = torch.zeros(rows, 2, co, device="cuda")
    used = C.c_int(0)
    bg = b.cuda() if bias else None
    l.check(lib.ustrun_conv2d_fwd(C.byref(src), 1, wf.data_ptr(), bg.data_ptr() if bias else None, n, ho, wo, co, k, s, d, y.data_ptr(),
                                  0, stat.data_ptr(), C.byref(used), dt, None))
    want = ref.bfloat16().float() if dt else ref
    assert rel(from_nhwc(y), want) < 1e-6
    if bias:        # f32 output from the bf16 kernel (the classifier maps): no rounding of the result
        y32 = torch.empty(n, ho, wo, co, device="cuda")
        l.check(lib.ustrun_conv2d_fwd(C.byref(src), 1, wf.data_ptr(), bg.data_ptr(), n, ho, wo, co, k, s, d, y32.data_ptr(), 1, None, None,
                                      dt, None))
        assert rel(from_nhwc(y32), ref) < 1e-6
    assert 0 < used.value <= rows
    np.testing.assert_allclose(stat[:, 0].sum(0).cpu().numpy(), want.sum((0, 2, 3)).numpy(), rtol=1e-5, atol=1e-2)   # (of the stored values)


def test_stem_conv7x7_nchw_input_maxpool():
    """The stem: 7x7 / stride 2 / padding 3 on the NCHW f32 network input (49 taps through the generic kernel), BatchNorm
    constants, then MaxPool2d(3, 2, 1) of the activated tensor -- odd extents."""
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(7)
    n, h, w = 2, 37, 45
    x = torch.randint(-3, 4, (n, 3, h, w), generator=g).float()
    wt = torch.randint(-2, 3, (64, 3, 7, 7), generator=g).float()
    ref = F.conv2d(x, wt, None, 2, 3)
    ho, wo = ref.shape[-2:]
    sc = torch.tensor([0.5, 1.0, 2.0, -1.0])[torch.randint(0, 4, (64,), generator=g)]
    sh = torch.randint(-2, 3, (64,), generator=g).float()
    pooled = F.max_pool2d(torch.relu(ref * sc[None, :, None, None] + sh[None, :, None, None]), 3, 2, 1)
    for dt in (0, 1):
        wf = torch.zeros(lib.ustrun_pack_conv_elems(64, 3, 49), dtype=torch.bfloat16 if dt else torch.float32, device="cuda")
        wg, xg = wt.cuda(), x.cuda()
        l.check(lib.ustrun_pack_conv(wg.data_ptr(), 64, 3, 49, wf.data_ptr(), dt, None))
        src = l.nchw_src(xg.data_ptr(), 3, h, w)
        y = torch.empty(n, ho, wo, 64, device="cuda", dtype=torch.bfloat16 if dt else torch.float32)
        l.check(lib.ustrun_conv2d_fwd(C.byref(src), 1, wf.data_ptr(), None, n, ho, wo, 64, 7, 2, 1, y.data_ptr(), 0, None, None, dt, None))
        assert rel(from_nhwc(y), ref.bfloat16().float() if dt else ref) < 1e-6
        scg, shg = sc.cuda(), sh.cuda()
        p = torch.empty(n, (ho + 1) // 2, (wo + 1) // 2, 64, device="cuda", dtype=y.dtype)
        l.check(lib.ustrun_maxpool3x3s2(y.data_ptr(), scg.data_ptr(), shg.data_ptr(), n, ho, wo, 64, p.data_ptr(), dt, None))
        assert tuple(p.shape[1:3]) == tuple(pooled.shape[-2:])
        assert rel(from_nhwc(p), pooled) < 1e-6           # (integers and halves: exact in bf16 too)


@pytest.mark.parametrize("proj", [False, True])
def test_bn_add_relu_and_sum_resize(proj):
    l = L()
    lib = l.lib()
    # bf16 storage of the join: exact on halves / integers
    gi = torch.Generator().manual_seed(12)
    yi, ii = torch.randint(-3, 4, (2, 64, 5, 7), generator=gi).float(), torch.randint(-3, 4, (2, 64, 5, 7), generator=gi).float()
    sci = torch.tensor([0.5, 1.0, 2.0, -1.0])[torch.randint(0, 4, (64,), generator=gi)]
    shi = torch.randint(-2, 3, (64,), generator=gi).float()
    ei = lambda v: v[None, :, None, None]
    refi = torch.relu(yi * ei(sci) + ei(shi) + (ii * ei(sci) + ei(shi) if proj else ii))
    yg16, ig16, scg, shg = nhwc(yi, 1), nhwc(ii, 1), sci.cuda(), shi.cuda()
    o16 = torch.empty_like(yg16)
    l.check(lib.ustrun_bn_add_relu(yg16.data_ptr(), scg.data_ptr(), shg.data_ptr(), ig16.data_ptr(), scg.data_ptr() if proj else None,
                                   shg.data_ptr() if proj else None, 2 * 5 * 7, 64, o16.data_ptr(), 1, None))
    assert rel(from_nhwc(o16), refi) < 1e-6
    g = torch.Generator().manual_seed(11)
    n, c, h, w = 2, 64, 9, 13
    y, idn = torch.randn(n, c, h, w, generator=g), torch.randn(n, c, h, w, generator=g)
    sc, sh, isc, ish = (torch.randn(c, generator=g) for _ in range(4))
    e = lambda v: v[None, :, None, None]
    ref = torch.relu(y * e(sc) + e(sh) + (idn * e(isc) + e(ish) if proj else idn))
    yg, ig = nhwc(y, 0), nhwc(idn, 0)
    t = [v.cuda() for v in (sc, sh, isc, ish)]
    out = torch.empty_like(yg)
    l.check(lib.ustrun_bn_add_relu(yg.data_ptr(), t[0].data_ptr(), t[1].data_ptr(), ig.data_ptr(), t[2].data_ptr() if proj else None,
                                   t[3].data_ptr() if proj else None, n * h * w, c, out.data_ptr(), 0, None))
    assert rel(from_nhwc(out), ref) < 1e-6
    # sum of four maps + bilinear resize, align_corners=True, to a non-multiple extent
    K, H, W = 3, 41, 50
    maps = [torch.randn(n, K, h, w, generator=g) for _ in range(4)]
    want = F.interpolate(sum(maps), size=(H, W), mode="bilinear", align_corners=True)
    mg = [nhwc(m, 0) for m in maps]
    arr = (C.c_void_p * 4)(*[m.data_ptr() for m in mg])
    o = torch.empty(n, K, H, W, device="cuda")
    l.check(lib.ustrun_sum_resize_bilinear(arr, 4, n, h, w, K, H, W, o.data_ptr(), None))
    assert rel(o.cpu(), want) < 1e-6


def _model(arch, k, seed, dtype):
    from networks.deeplabv2 import DeepLabV2
    torch.manual_seed(seed)
    return DeepLabV2(arch, k, pretrained=False, dtype=dtype).cuda()


@pytest.mark.parametrize("name,arch", [("g10_deeplabv2_r50_n2_96x80", "resnet50"), ("g10_deeplabv2_r101_n1_128", "resnet101")])
def test_deeplab_reference_goldens(name, arch):
    """The HIP network (f32) against outputs captured from the reference's own modules: train-mode logits, backbone feature
    norms, running statistics after one call, eval-mode logits.  Bars: logits rel-L2 5e-4 (two f32 evaluations of 53-104
    BatchNorm layers differ by ~1e-4), feature norms 1e-4, running-statistic sums 1e-4."""
    g = load_golden(name)
    n, _, h, w, k = [int(v) for v in g["shape"]]
    m = _model(arch, k, int(g["model_seed"]), "f32").train()
    gen = torch.Generator().manual_seed(int(g["input_seed"]))
    x = (torch.randint(0, 256, (n, 3, h, w), generator=gen).float() / 127.5 - 1).cuda()
    import copy
    m2 = copy.deepcopy(m)
    with torch.no_grad():
        feats = m2.backbone.base_forward(x)
        logits = m(x)
    np.testing.assert_allclose([float(f.double().norm()) for f in feats], g["feat_l2"], rtol=1e-4)
    assert [list(f.shape) for f in feats] == g["feat_shape"].tolist()
    idx = torch.from_numpy(g["sample_idx"])
    flat = logits.flatten().cpu()
    assert rel(flat[idx], torch.from_numpy(g["sample_val"])) < 5e-4
    assert abs(float(flat.double().norm()) - float(g["logit_l2"])) <= 1e-4 * float(g["logit_l2"])
    sd = m.state_dict()
    # (sums over channels of per-channel means cancel: absolute floor 2e-4 on sums of magnitude 1)
    np.testing.assert_allclose([float(v.double().sum()) for kk, v in sd.items() if kk.endswith("running_mean")], g["rm_sums"], rtol=1e-3, atol=2e-4)
    np.testing.assert_allclose([float(v.double().sum()) for kk, v in sd.items() if kk.endswith("running_var")], g["rv_sums"], rtol=1e-3, atol=2e-4)
    assert all(int(v) == 1 for kk, v in sd.items() if kk.endswith("num_batches_tracked"))
    m.eval()
    with torch.no_grad():
        ev = m(x).flatten().cpu()
    assert rel(ev[idx], torch.from_numpy(g["eval_val"])) < 5e-4


def _bf16_emulation(x, sd, arch, train):
    """The oracle with every convolution's operands and (bias-free) outputs rounded to bf16: what bf16 operands + bf16 stored
    pre-BatchNorm tensors cost on THIS net and input, independent of any HIP kernel (the yardstick for the bf16 path)."""
    from oracle import deeplab_ref as D
    real = F.conv2d
    r16 = lambda t: t.bfloat16().float()

    def conv16(inp, w, b=None, *a, **k):
        y = real(r16(inp), r16(w), b, *a, **k)
        return r16(y) if b is None else y
    D.F.conv2d = conv16
    try:
        return D.deeplabv2_forward(x, {k: v.clone() for k, v in sd.items()}, arch, train)
    finally:
        D.F.conv2d = real


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_deeplab_vs_oracle(dtype):
    """resnet50 DeepLabV2 against the CPU oracle on an odd-extent input, train mode then eval mode.  f32: 5e-4 rel-L2
    (measured 5e-5 / 2e-6).  bf16: eval 4e-2 (measured 2.2e-2: operands, raw conv outputs and block outputs stored in bf16
    through 53 convolutions).  TRAIN-mode bf16 on a random-init ResNet at batch 2 is a different matter: the residual stream's
    per-channel mean outgrows its spatial deviation with depth, a bf16 tensor keeps 8 bits of the MEAN, and the following
    BatchNorm rescales what is left of the deviation -- a pure torch-CPU emulation of bf16 operand / output rounding, no HIP
    code involved, already sits 0.43 rel-L2 from the f32 result.  That emulation is the yardstick: the HIP path must stay
    within 1.3x of it (measured 0.44 vs 0.43)."""
    from oracle import deeplab_ref as D
    sd = D.make_state_dict("resnet50", 4, 21)
    m = _model("resnet50", 4, 21, dtype).train()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 72, 104, generator=g)
    with torch.no_grad():
        sdo = {k: v.clone() for k, v in sd.items()}          # (its running statistics move with the train-mode call, like the model's)
        ref = D.deeplabv2_forward(x, sdo, "resnet50", True)
        got = m(x.cuda()).cpu()
        assert got.shape == ref.shape
        e_train = rel(got, ref)
        m.eval()
        ref_e = D.deeplabv2_forward(x, sdo, "resnet50", False)
        e_eval = rel(m(x.cuda()).cpu(), ref_e)
        print(f"deeplab {dtype}: train rel-L2 {e_train:.2e}, eval {e_eval:.2e}")
        if dtype == "f32":
            assert e_train < 5e-4 and e_eval < 5e-4
        else:
            yard = rel(_bf16_emulation(x, sd, "resnet50", True), ref)
            print(f"deeplab bf16: torch-CPU bf16-rounding emulation, train: {yard:.2e}")
            assert e_eval < 4e-2 and e_train < 1.3 * yard + 2e-2


def test_deeplab_refuses_cpu_and_training_graph():
    from networks.deeplabv2 import DeepLabV2
    m = DeepLabV2("resnet50", 2, pretrained=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 64, 64))
    m = m.cuda().train()
    with pytest.raises(NotImplementedError, match="records no autograd graph"):      # the stand-alone feature path; DeepLabV2.forward
        m.backbone(torch.zeros(1, 3, 64, 64, device="cuda"))                         # itself is differentiable (test_gpu_deeplab_bwd)


def test_rowwin_stem_and_aspp_gather_direct():
    """The two reshaped operators on their own, exact integers: the 7x7 / stride-2 stem as 7 row-window segments of the padded
    NHWC input (ustrun_conv_rowwin_fwd) against F.conv2d, and the classifier's shifted add (ustrun_aspp_gather) of a 1x1 GEMM's
    columns against the sum of four dilated 3x3 convolutions with bias."""
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(31)
    n, h, w = 2, 29, 34
    x = torch.randint(-3, 4, (n, 3, h, w), generator=g).float()
    wt = torch.randint(-2, 3, (64, 3, 7, 7), generator=g).float()
    ref = F.conv2d(x, wt, None, 2, 3)
    ho, wo = ref.shape[-2:]
    for dt in (0, 1):
        td = torch.bfloat16 if dt else torch.float32
        xp = F.pad(x.permute(0, 2, 3, 1), (0, 0, 3, 4, 3, 3)).to(td).contiguous().cuda()        # [n, h+6, w+7, 3]
        wr = F.pad(wt.permute(0, 2, 3, 1).reshape(64, 7, 21), (0, 3)).permute(0, 2, 1).contiguous().cuda()   # [co][24][ky]
        wf = torch.zeros(lib.ustrun_pack_conv_elems(64, 24, 7), dtype=td, device="cuda")
        l.check(lib.ustrun_pack_conv(wr.data_ptr(), 64, 24, 7, wf.data_ptr(), dt, None))
        hp, wp = h + 6, w + 7
        src = l.Src(xp.data_ptr(), None, None, 24, hp, wp - 7, hp * wp * 3, wp * 3, 3, 1, 0, 0, 0, 0, 0, 0, 0)
        y = torch.empty(n, ho, wo, 64, device="cuda", dtype=td)
        l.check(lib.ustrun_conv_rowwin_fwd(C.byref(src), wf.data_ptr(), n, ho, wo, 64, 7, 2, y.data_ptr(), None, None, dt, None))
        assert rel(from_nhwc(y), ref.bfloat16().float() if dt else ref) < 1e-6
    # shifted add: z holds, per pixel, the 1x1 products of every (rate, tap, class)
    K, ci, hh, ww = 3, 16, 21, 17
    feat = torch.randint(-3, 4, (n, ci, hh, ww), generator=g).float()
    ws = [torch.randint(-2, 3, (K, ci, 3, 3), generator=g).float() for _ in range(4)]
    bs = [torch.randint(-2, 3, (K,), generator=g).float() for _ in range(4)]
    rates = (6, 12, 18, 24)
    want = sum(F.conv2d(feat, ws[r], bs[r], 1, rates[r], rates[r]) for r in range(4))
    wall = torch.stack([wq.reshape(K, ci, 9).permute(2, 0, 1) for wq in ws], 0).reshape(4 * 9 * K, ci)       # row (r*9+tap)*K+k
    z = torch.einsum("nchw,oc->nhwo", feat, wall).contiguous().cuda()
    bsum = sum(bs).cuda()
    out = torch.empty(n, hh, ww, K, device="cuda")
    rr = (C.c_int * 4)(*rates)
    l.check(lib.ustrun_aspp_gather(z.data_ptr(), n, hh, ww, K, 4, rr, bsum.data_ptr(), out.data_ptr(), None))
    assert rel(from_nhwc(out), want) < 1e-6


@pytest.mark.parametrize("dt", [0, 1])
def test_rowwin_patches_exact(dt):
    """ustrun_rowwin_patches: the stem's seven 24-element row windows per output pixel as one 192-column GEMM row (zero tail), and
    the 1x1 GEMM over them equal to the 7x7 / stride-2 convolution -- exact small integers, odd extents."""
    l = L()
    lib = l.lib()
    g = torch.Generator().manual_seed(77)
    n, h, w = 2, 31, 26
    td = torch.bfloat16 if dt else torch.float32
    x = torch.randint(-3, 4, (n, 3, h, w), generator=g).float()
    wt = torch.randint(-2, 3, (64, 3, 7, 7), generator=g).float()
    ref = F.conv2d(x, wt, None, 2, 3)
    ho, wo = ref.shape[-2:]
    xp = F.pad(x.permute(0, 2, 3, 1), (0, 0, 3, 4, 3, 3)).to(td).contiguous().cuda()        # [n, h+6, w+7, 3]
    hp, wp = h + 6, w + 7
    src = l.Src(xp.data_ptr(), None, None, 24, hp, wp - 7, hp * wp * 3, wp * 3, 3, 1, 0, 0, 0, 0, 0, 0, 0)
    pt = torch.full((n, ho, wo, 192), 9.0, device="cuda", dtype=td)
    l.check(lib.ustrun_rowwin_patches(C.byref(src), n, ho, wo, 7, 2, 192, pt.data_ptr(), dt, None))
    flat = xp.float().cpu().reshape(n, hp, wp * 3)
    want = torch.zeros(n, ho, wo, 192)
    for s in range(7):
        for xx in range(wo):
            want[:, :, xx, s * 24:(s + 1) * 24] = flat[:, s:s + 2 * ho:2, 6 * xx:6 * xx + 24][:, :ho]
    assert torch.equal(pt.float().cpu(), want)
    wk = F.pad(F.pad(wt.permute(0, 2, 3, 1).reshape(64, 7, 21), (0, 3)).reshape(64, 168), (0, 24)).contiguous().cuda()   # [co][ky*24+kx*3+ci]
    wf = torch.zeros(lib.ustrun_pack_conv_elems(64, 192, 1), dtype=td, device="cuda")
    l.check(lib.ustrun_pack_conv(wk.data_ptr(), 64, 192, 1, wf.data_ptr(), dt, None))
    psrc = l.nhwc_src(pt.data_ptr(), 192, ho, wo)
    y = torch.empty(n, ho, wo, 64, device="cuda", dtype=td)
    l.check(lib.ustrun_conv2d_fwd(C.byref(psrc), 1, wf.data_ptr(), None, n, ho, wo, 64, 1, 1, 1, y.data_ptr(), 0, None, None, dt, None))
    assert rel(from_nhwc(y), ref.bfloat16().float() if dt else ref) < 1e-6


def test_deeplab_tta_matches_the_reference_formula():
    """BaseNet.forward(x, tta=True) (base.py:24-45): ten views -- five scales, each plain and mirrored -- softmaxed, un-mirrored,
    resized back and summed, against the same formula evaluated with the CPU oracle as `base_forward` (eval mode, f32)."""
    from oracle import deeplab_ref as D
    sd = D.make_state_dict("resnet50", 3, 41)
    m = _model("resnet50", 3, 41, "f32").eval()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 3, 48, 64, generator=g)
    with torch.no_grad():
        got = m(x.cuda(), tta=True).cpu()
        h, w = x.shape[-2:]
        want = None
        for scale in (0.5, 0.75, 1.0, 1.5, 2.0):
            cur = F.interpolate(x, size=(int(h * scale), int(w * scale)), mode="bilinear", align_corners=True)
            for flip in (False, True):
                out = F.softmax(D.deeplabv2_forward(cur.flip(3) if flip else cur, sd, "resnet50", False), dim=1)
                out = F.interpolate(out.flip(3) if flip else out, (h, w), mode="bilinear", align_corners=True)
                want = out if want is None else want + out
    assert got.shape == want.shape and rel(got, want) < 1e-4
    assert abs(float(got.sum(1).mean()) - 10.0) < 1e-4          # ten probability maps
