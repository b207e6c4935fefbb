"""GPU parity of the loss / pseudo-label / mixing / optimizer kernels (C ABI via ustrun.functional)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import host_ref as H
from oracle import losses_ref as L

pytestmark = pytest.mark.gpu


def t(a):
    return torch.from_numpy(np.asarray(a))


def F():
    from ustrun import functional
    return functional


@pytest.mark.parametrize("K", [2, 4])
def test_dice_loss_module_matches_reference_goldens(K):
    """utils.losses.DiceLossWithMask on the GPU vs values/grads captured from the reference."""
    from utils.losses import DiceLossWithMask
    g = load_golden("g4_losses")
    dl = DiceLossWithMask(K)
    cases = {"sm": dict(target=t(g[f"K{K}.tgt"]).cuda(), softmax=True),
             "sm_mask": dict(target=t(g[f"K{K}.tgt"]).cuda(), mask=t(g[f"K{K}.mask"]).cuda(), softmax=True),
             "sg": dict(target=t(g[f"K{K}.tgt_ml"]).unsqueeze(1).cuda(), sigmoid=True, multi=True),
             "sg_mask": dict(target=t(g[f"K{K}.tgt_ml"]).unsqueeze(1).cuda(), mask=t(g[f"K{K}.mask_ml"]).cuda(), sigmoid=True, multi=True)}
    for tag, kw in cases.items():
        lg = t(g[f"K{K}.logits"]).cuda().requires_grad_(True)
        val = dl(lg, **kw)
        val.backward()
        np.testing.assert_allclose(float(val), float(g[f"K{K}.{tag}.val"]), rtol=2e-5)
        np.testing.assert_allclose(lg.grad.cpu().numpy(), g[f"K{K}.{tag}.grad"], rtol=2e-4, atol=1e-8)


@pytest.mark.parametrize("K", [2, 4])
def test_dice_loss_module_remaining_modes_match_reference_goldens(K):
    """The rest of DiceLossWithMask's signature (losses.py:236-268; no reference script calls these): class weights, sigmoid per
    class (5-D target), softmax + multi (full and class-broadcast target / mask), raw inputs -- value and gradient against what the
    reference itself returned (tools/gen_goldens.py g4b), through ustrun_dice_fwd/_bwd."""
    from utils.losses import DiceLossWithMask
    g = load_golden("g4b_losses_rest")
    dl = DiceLossWithMask(K)
    tgt, mask = t(g[f"K{K}.tgt"]).cuda(), t(g[f"K{K}.mask"]).cuda()
    tml, mml = t(g[f"K{K}.tgt_ml"]).cuda(), t(g[f"K{K}.mask_ml"]).cuda()
    w = [float(v) for v in g[f"K{K}.weight"]]
    cases = {"sm_w": dict(target=tgt, softmax=True, weight=w), "sm_mask_w": dict(target=tgt, mask=mask, softmax=True, weight=w),
             "sg_pc": dict(target=tgt.unsqueeze(1), sigmoid=True), "sg_pc_mask_w": dict(target=tgt.unsqueeze(1), mask=mask, sigmoid=True, weight=w),
             "sm_multi": dict(target=tml, softmax=True, multi=True), "sm_multi_mask": dict(target=tml, mask=mml, softmax=True, multi=True),
             "sm_multi_bcast": dict(target=tgt.float(), mask=mask, softmax=True, multi=True),
             "raw_pc": dict(target=tgt), "raw_pc_mask_w": dict(target=tgt, mask=mask, weight=w),
             "raw_multi": dict(target=tml, multi=True), "raw_multi_mask": dict(target=tml, mask=mml, multi=True)}
    for tag, kw in cases.items():
        lg = t(g[f"K{K}.logits"]).cuda().requires_grad_(True)
        val = dl(lg, **kw)
        (2.0 * val).backward()                                    # (an upstream factor: the device-side gradient scale is used)
        np.testing.assert_allclose(float(val.detach()), float(g[f"K{K}.{tag}.val"]), rtol=2e-5, err_msg=tag)
        np.testing.assert_allclose(lg.grad.cpu().numpy(), 2.0 * g[f"K{K}.{tag}.grad"], rtol=2e-4, atol=1e-8, err_msg=tag)
    with pytest.raises(AssertionError):
        dl(t(g[f"K{K}.logits"]).cuda(), tgt.squeeze(1))          # losses.py:253: one-hot of a 3-D target does not match
    with pytest.raises(AssertionError):
        dl(t(g[f"K{K}.logits"]).cuda(), tgt, softmax=True, sigmoid=True)


@pytest.mark.parametrize("mode,K,N,H", [("softmax", 2, 3, 40), ("softmax", 4, 2, 33), ("sigmoid", 2, 3, 40)])
@pytest.mark.parametrize("masked", [False, True])
def test_seg_loss_vs_oracle(mode, K, N, H, masked):
    g = torch.Generator().manual_seed(K + N + H)
    logits = 3 * torch.randn(N, K, H, H, generator=g)
    if mode == "softmax":
        tgt = torch.randint(0, K, (N, H, H), generator=g)
        mask = (torch.rand(N, 1, H, H, generator=g) > 0.4).float() if masked else None
    else:
        tgt = (torch.rand(N, K, H, H, generator=g) > 0.5).float()
        mask = (torch.rand(N, K, H, H, generator=g) > 0.4).float() if masked else None
    lr = logits.clone().requires_grad_(True)
    ce, dc = L.seg_loss(lr, tgt, mask, mode, K)
    (ce + 0.7 * dc).backward()
    lg = logits.cuda().requires_grad_(True)
    ce_g, dc_g = F().seg_loss(lg, tgt.cuda(), None if mask is None else mask.cuda(), mode)
    (ce_g + 0.7 * dc_g).backward()
    np.testing.assert_allclose(float(ce_g), float(ce), rtol=2e-5)
    np.testing.assert_allclose(float(dc_g), float(dc), rtol=2e-5)
    np.testing.assert_allclose(lg.grad.cpu().numpy(), lr.grad.numpy(), rtol=2e-4, atol=1e-9)


@pytest.mark.parametrize("mode,K", [("softmax", 2), ("softmax", 4), ("sigmoid", 2)])
def test_pseudo_label_bit_exact(mode, K):
    g = torch.Generator().manual_seed(17 + K)
    logits = 4 * torch.randn(4, K, 64, 64, generator=g)
    logits[0, :, :4] = 0.0                                   # exact ties -> first index
    logits[1, 0, :8] = logits[1, K - 1, :8]
    label, mask = L.pseudo_label(logits, 0.95, mode)
    lab_g, mask_g = F().pseudo_label(logits.cuda(), 0.95, mode)
    assert lab_g.dtype == label.dtype and lab_g.shape == label.shape and mask_g.shape == mask.shape
    assert torch.equal(lab_g.cpu(), label)                   # argmax / >=0.5 masks: bit-exact
    # confidence masks are ulp-sensitive exactly at the threshold: exclude |p - th| < 1e-6 and report
    p = torch.softmax(logits, 1).max(1)[0][:, None] if mode == "softmax" else torch.sigmoid(logits)
    near = ((p - 0.95).abs() < 1e-6) | ((p - 0.05).abs() < 1e-6)
    assert int(near.sum()) < 10
    assert torch.equal(mask_g.cpu()[~near], mask[~near])


@pytest.mark.parametrize("mode", ["softmax", "sigmoid"])
def test_mix_targets_and_box_mix_exact(mode):
    g = torch.Generator().manual_seed(23)
    N, K, H = 3, 2, 24
    box = torch.zeros(N, H, H)
    box[0, 3:11, 5:20] = 1
    box[2, :, :7] = 1

    def lab():
        return torch.randint(0, K, (N, H, H), generator=g) if mode == "softmax" else (torch.rand(N, K, H, H, generator=g) > 0.5).float()

    def msk():
        return (torch.rand(N, 1 if mode == "softmax" else K, H, H, generator=g) > 0.3).float()

    args = [lab(), msk(), lab(), msk(), lab(), msk()]
    cut_label, cut_mask = lab(), msk()
    ref = L.mix_targets(mode, *args, box, cut_label, cut_mask)
    got = F().mix_targets(mode, box.cuda(), *[a.cuda() for a in args], cut_label.cuda(), cut_mask.cuda())
    for r, o in zip(ref, got):
        assert r.dtype == o.dtype and torch.equal(o.cpu(), r)
    a, b = torch.randn(N, 3, H, H, generator=g), torch.randn(N, 3, H, H, generator=g)
    ib = box[:, None]
    assert torch.equal(F().box_mix(a.cuda(), b.cuda(), box.cuda()).cpu(), a * (1 - ib) + b * ib)


def test_rect_masks_equal_oracle_cutmix_and_cover_maps():
    """The device-built CutMix maps equal the oracle's host maps under the same RNG streams (train.py:222-251)."""
    import random
    from ustrun import trainer as T
    for seed in (0, 1, 2):
        random.seed(seed); np.random.seed(seed)
        ref = np.stack([H.cutmix_box(64, p=0.7) for _ in range(9)])
        random.seed(seed); np.random.seed(seed)
        rects = [T.cutmix_rect(64, p=0.7) for _ in range(9)]
        got = F().rect_masks(rects, 64, 64, "cuda")
        assert got.shape == (9, 64, 64) and np.array_equal(got.cpu().numpy(), ref)
    region = np.zeros((48, 48), dtype=np.float32)
    region[7, 30] = 1; region[20, 4] = 1; region[33, 11] = 2
    assert np.array_equal(F().rect_masks([T.all_cover_rect(region)], 48, 48, "cuda")[0].cpu().numpy(), H.all_cover_box(region))
    random.seed(5); np.random.seed(5)
    ref = H.all_cover_box(np.zeros((48, 48), dtype=np.float32))          # empty region: falls back to a random box
    random.seed(5); np.random.seed(5)
    got = F().rect_masks([T.all_cover_rect(np.zeros((48, 48), dtype=np.float32))], 48, 48, "cuda")[0]
    assert np.array_equal(got.cpu().numpy(), ref)
    many = np.array([[i % 5, 5 + i % 3, i % 7, 4 + i % 4] for i in range(130)])        # more than one launch's 64
    got = F().rect_masks(many, 8, 8, "cuda").cpu().numpy()
    assert np.array_equal(got, np.stack([T.rect_map(r, 8) for r in many]))
    with pytest.raises(RuntimeError, match="rect_masks"):
        F().rect_masks(np.zeros((0, 4)), 8, 8, "cuda")


def test_assemble_gathers_copies_and_composites_exactly():
    """ustrun_assemble against the torch expressions it replaces (train.py:627,643-647,689-702): rows gathered by address from two
    tensors (cat + fancy index), CutMix composites with the box broadcast over channels, int64 label rows, one row aliased many
    times -- bit-exact."""
    Fn = F()
    g = torch.Generator().manual_seed(3)
    a = torch.randn(5, 3, 24, 24, generator=g).cuda()
    b = torch.randn(4, 3, 24, 24, generator=g).cuda()
    box = (torch.rand(5, 24, 24, generator=g) > 0.6).float().cuda()
    choice = [6, 0, 8, 3, 3]
    rows_ab = Fn.row_ptrs(a) + Fn.row_ptrs(b)
    bx = Fn.row_ptrs(box)
    cut = [rows_ab[c] for c in choice]
    ar = Fn.row_ptrs(a)
    out = Fn.assemble([(p, 0, 0) for p in ar] + [(ar[i], cut[i], bx[i]) for i in range(5)] + [(cut[i], ar[i], bx[i]) for i in range(5)] +
                      [(c, 0, 0) for c in cut], a, 24 * 24)
    mix = torch.cat((a, b), 0)[choice]
    bb = box[:, None]
    ref = torch.cat((a, a * (1 - bb) + mix * bb, mix * (1 - bb) + a * bb, mix), 0)
    assert torch.equal(out, ref)
    lab = torch.randint(0, 4, (6, 40, 24), generator=g).cuda()
    sel = [5, 5, 0, 2]
    lr = Fn.row_ptrs(lab)
    assert torch.equal(Fn.assemble([(lr[i], 0, 0) for i in sel], lab), lab[sel])
    ones = torch.ones(1, 2, 24, 24).cuda()
    m = torch.rand(3, 2, 24, 24, generator=g).cuda()
    rows = [ones.data_ptr()] * 4 + Fn.row_ptrs(m)
    pick = [0, 6, 3, 4]
    assert torch.equal(Fn.assemble([(rows[i], 0, 0) for i in pick], ones), torch.cat((ones.expand(4, -1, -1, -1), m), 0)[pick])
    big = torch.randn(130, 1, 16, 16, generator=g).cuda()           # more rows than one launch's table holds
    perm = torch.randperm(130, generator=g).tolist()
    br = Fn.row_ptrs(big)
    assert torch.equal(Fn.assemble([(br[i], 0, 0) for i in perm], big), big[perm])


@pytest.mark.parametrize("dataset", ["fundus", "prostate", "BUSI", "MNMS"])
def test_decode_labels_equals_the_loop_head(dataset):
    """ustrun_decode_labels == the torch expressions of train.py:590-608 / train_mnms.py:549-556 (oracle/step_ref.decode_labels)."""
    from oracle import step_ref as S
    from ustrun import synthetic
    y = synthetic.labels(dataset, 3, 48, torch.Generator().manual_seed(2))
    got = F().decode_labels(dataset, y.cuda())
    ref = S.decode_labels(dataset, y)
    assert got.dtype == ref.dtype and got.shape == ref.shape
    assert torch.equal(got.cpu(), ref)


def test_region_bbox_equals_the_cover_box_of_the_region():
    """ustrun_region_bbox + fold_bbox == oracle/host_ref.all_cover_box (train.py:242-251) of the region train.py:722-729 builds,
    for float (fundus) and int64 (softmax datasets) planes, an empty region and a single pixel."""
    from ustrun import trainer as T
    Fn = F()
    g = torch.Generator().manual_seed(7)
    for trial in range(6):
        S = 40
        planes = [torch.zeros(S, S) for _ in range(4)]
        for p in planes[:3]:
            if trial == 4:
                continue
            y0, x0 = int(torch.randint(0, S - 8, (1,), generator=g)), int(torch.randint(0, S - 8, (1,), generator=g))
            p[y0:y0 + int(torch.randint(1, 8, (1,), generator=g)), x0:x0 + int(torch.randint(1, 8, (1,), generator=g))] = 1
        if trial == 5:
            planes = [torch.zeros(S, S) for _ in range(4)]
            planes[2][17, 31] = 1
        region = planes[1].clone()                            # the reference's construction (fundus form)
        region[planes[0].long() == 1] = 1
        region[planes[2].long() == 1] = 1
        region[planes[3].long() == 1] = 1
        part = torch.empty(64, 4, dtype=torch.int32, device="cuda")
        Fn.region_bbox_partials([planes[1].cuda(), planes[0].cuda(), planes[2].cuda(), planes[3].cuda()], S, S, part)
        rect = Fn.fold_bbox(part.cpu().numpy())
        if trial == 4:
            assert rect is None
            continue
        assert np.array_equal(T.rect_map(rect, S), H.all_cover_box(region.numpy())), trial
        lab = (planes[0] * 3).long().cuda()                   # class ids: non-zero = set
        part2 = torch.empty(64, 4, dtype=torch.int32, device="cuda")
        Fn.region_bbox_partials([lab, planes[2].long().cuda()], S, S, part2)
        r2 = region * 0
        r2[planes[0] != 0] = 1
        r2[planes[2] != 0] = 1
        assert np.array_equal(T.rect_map(Fn.fold_bbox(part2.cpu().numpy()), S), H.all_cover_box(r2.numpy())), trial


def test_upload_small_round_trip():
    for arr, dt in ((np.arange(16)[::-1].copy(), torch.long), (np.linspace(0, 1, 7, dtype=np.float32), torch.float32),
                    (np.zeros(0, dtype=np.int64), torch.long), (np.arange(256), torch.long)):
        got = F().upload_small(arr, "cuda", dt)
        assert got.dtype == dt and np.array_equal(got.cpu().numpy(), arr)
    big = np.arange(1000)                                   # 8000 bytes: four launches
    assert np.array_equal(F().upload_small(big, "cuda", torch.long).cpu().numpy(), big)
    with pytest.raises(RuntimeError, match="upload_small"):
        F().upload_small(np.zeros(20000), "cuda", torch.float64)


def test_dice_counts_match_numpy_dice():
    from utils import metrics
    g = load_golden("g6_metrics")
    p2, t2 = t(g["l2.pred"]).cuda(), t(g["l2.tgt"]).cuda()
    c = F().dice_counts(p2, t2).cpu().numpy().astype(np.float64)
    d = metrics.dice_from_counts(c[..., 0], c[..., 1], c[..., 2])
    np.testing.assert_allclose(d.T, g["l2.arr"], rtol=1e-12)
    p3, t3 = t(g["l3.pred"]).cuda(), t(g["l3.tgt"]).cuda()
    c = F().dice_counts(p3, t3, by_class=True, n_classes=3).cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(metrics.dice_from_counts(c[..., 0], c[..., 1], c[..., 2]).T, g["l3.arr"], rtol=1e-12)
    a, b = t(g["bin.pred"].astype(np.int64)).cuda(), t(g["bin.tgt"].astype(np.int64)).cuda()
    c = F().dice_counts(a, b).cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(metrics.dice_from_counts(c[..., 0], c[..., 1], c[..., 2])[:, 0], g["bin.each"], rtol=1e-12)


def test_sgd_ema_matches_torch_sgd_trajectory():
    g = load_golden("g8_sgd")
    p = torch.zeros(36, device="cuda")
    p[:35] = t(g["p0"]).flatten().cuda()
    v, tt = torch.zeros_like(p), torch.zeros_like(p)
    tt[:35] = 1.0
    lr = 0.03
    for s in range(3):
        gr = torch.zeros_like(p)
        gr[:35] = t(g[f"g{s}"]).flatten().cuda()
        alpha = H.ema_alpha(s, 0.99)
        t_prev = tt.clone()
        F().sgd_ema(p, gr, v, tt, lr, 0.9, 1e-4, s == 0, alpha)
        np.testing.assert_allclose(p[:35].cpu().numpy(), g[f"p{s + 1}"].flatten(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(tt.cpu().numpy(), (alpha * t_prev + (1 - alpha) * p).cpu().numpy(), rtol=1e-6, atol=1e-7)
        lr = H.poly_lr(0.03, s, 30000)


@pytest.mark.parametrize("C,S,LB", [(3, 32, 0.1), (1, 64, 0.05), (3, 256, 0.01), (1, 384, 0.01), (1, 512, 0.01), (1, 100, 0.04)])
def test_freq_mix_device_matches_numpy_fft(C, S, LB):
    """Device amplitude mix vs the oracle's numpy FFT restatement (pinned to the reference by G7).  Window half-widths b = 3, 3, 2,
    3 (7 x 7 at 384), 5 (11 x 11 at 512: bands of 4 window rows per block, > 64 KB of twiddles in LDS) and 4 (9 x 9, bands of 3)."""
    from ustrun.fftmix import freq_mix_device
    g = torch.Generator().manual_seed(C + S)
    n = 3
    src = torch.randint(0, 256, (n, C, S, S), generator=g).float() / 127.5 - 1
    trg = torch.randint(0, 256, (n, C, S, S), generator=g).float() / 127.5 - 1
    ratios = [0.0, 0.37, 1.0]
    got = freq_mix_device(src.cuda(), trg.cuda(), LB, ratios).cpu().numpy()
    for i in range(n):
        amp = H_amp(((trg[i] + 1) * 127.5).numpy())
        ref = H.freq_mix(((src[i] + 1) * 127.5).numpy().astype(np.float64), amp, L=LB, ratio=ratios[i])
        ref = np.clip(ref, 0, 255).astype(np.float32) / 127.5 - 1
        np.testing.assert_allclose(got[i], ref, rtol=0, atol=2e-5)


def H_amp(x):
    return H.amp_spectrum(x.astype(np.float64))


def test_loss_scale_follows_gradscaler():
    """The device-side loss scale (ustrun_amp_check / ustrun_sgd_ema_scaled / ustrun_amp_update) against the semantics of
    torch.cuda.amp.GradScaler around torch.optim.SGD as the reference uses them (train.py:552,842-851): a finite scaled
    gradient is unscaled and applied; a non-finite one skips optimizer.step (parameters, momentum untouched) while the EMA
    line still runs and the scale backs off; `growth_interval` clean steps in a row double it.  Expectation: torch.optim.SGD on
    the CPU driven by the same found_inf decisions (GradScaler itself refuses to run without a CUDA device in this image's
    CPU build, so its three rules are restated here: step skipped iff inf/nan, scale x0.5 on skip, x2 after `interval`)."""
    from ustrun import functional as F
    n, lr, mu, wd, alpha = 10007, 0.03, 0.9, 1e-4, 0.99
    g0 = torch.Generator().manual_seed(4)
    p0 = torch.randn(n, generator=g0)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([p_ref], lr=lr, momentum=mu, weight_decay=wd)
    t_ref = p0.clone() * 0.5
    p, v, t = p0.clone().cuda(), torch.zeros(n).cuda(), (p0 * 0.5).cuda()
    ls = F.LossScale("cuda", init_scale=1024.0, growth_interval=3)
    scale, tracker = 1024.0, 0
    plan = ["ok", "inf", "ok", "ok", "ok", "nan", "ok"]
    for step, kind in enumerate(plan):
        grad = torch.randn(n, generator=g0)                    # the UNSCALED gradient
        gs = grad * scale                                      # what a backward through scaler.scale(loss) leaves
        if kind == "inf":
            gs[n - 2] = float("inf")                           # (in the scalar tail of the 16-byte loop)
        if kind == "nan":
            gs[17] = float("nan")
        gd = gs.cuda()
        assert float(ls.state[0]) == scale
        ls.step(p, gd, v, t, lr, mu, wd, step == 0, alpha, grad_scale=0.5)     # grad_scale: the 1 / world of two ranks
        if kind == "ok":
            p_ref.grad = grad * 0.5
            opt.step()
            tracker += 1
            if tracker == 3:
                scale, tracker = scale * 2, 0
        else:
            scale, tracker = scale * 0.5, 0
        t_ref = alpha * t_ref + (1 - alpha) * p_ref.detach()
        assert float((p.cpu() - p_ref.detach()).abs().max()) <= 2e-6, (step, kind)
        assert float((t.cpu() - t_ref).abs().max()) <= 2e-6, (step, kind)
        if step >= 1:
            vb = opt.state[p_ref]["momentum_buffer"]
            assert float((v.cpu() - vb).abs().max()) <= 1e-5 * float(vb.abs().max()), (step, kind)
        st = ls.state.cpu()
        assert float(st[0]) == scale and float(st[1]) == scale and int(st[2]) == tracker and float(st[3]) == 0.0
    assert ls.skipped_steps() == (2, len(plan))
    sd = ls.state_dict()
    assert sd["scale"] == scale and sd["_growth_tracker"] == tracker and sd["growth_interval"] == 3
