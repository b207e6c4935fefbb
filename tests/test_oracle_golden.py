"""Pin the oracle (CPU restatement) against vectors captured from the reference itself.

Fixtures: tests/golden/*.npz written by tools/gen_goldens.py (reference imported in place).
"""
import random

import numpy as np
import pytest
import torch

from oracle import host_ref as H
from oracle import losses_ref as L
from oracle import unet_ref as U
from conftest import load_golden

RT, AT = 2e-5, 2e-6


def t(a):
    return torch.from_numpy(np.asarray(a))


def sd_from(g, tag):
    return {k[len(tag):]: t(g[k]).clone() for k in g.files if k.startswith(tag)}


def close(a, b, rtol=RT, atol=AT):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol)


def _remap(sd, kind):
    """Block fixtures use the block's own key names; the oracle functions take a prefix."""
    return {("blk." + k): v for k, v in sd.items()}


BLOCKS = [
    ("g1_doubleconv_3_8", lambda ins, sd, tr: U.double_conv(ins[0], "blk", sd, tr)),
    ("g1_doubleconv_8_8_mid4", lambda ins, sd, tr: U.double_conv(ins[0], "blk", sd, tr)),
    ("g1_down_8_16", lambda ins, sd, tr: U.down(ins[0], "blk", sd, tr)),
    ("g1_down_8_16_odd", lambda ins, sd, tr: U.down(ins[0], "blk", sd, tr)),
    ("g1_up_16_8_convT", lambda ins, sd, tr: U.up(ins[0], ins[1], "blk", sd, tr, False)),
    ("g1_up_16_8_convT_odd", lambda ins, sd, tr: U.up(ins[0], ins[1], "blk", sd, tr, False)),
    ("g1_up_16_8_bilinear", lambda ins, sd, tr: U.up(ins[0], ins[1], "blk", sd, tr, True)),
    ("g1_outconv_8_2", lambda ins, sd, tr: U.out_conv(ins[0], "blk", sd)),
]


@pytest.mark.parametrize("name,fn", BLOCKS, ids=[b[0] for b in BLOCKS])
def test_block_matches_reference(name, fn):
    g = load_golden(name)
    sd = _remap(sd_from(g, "sd0."), name)
    pk = U.param_keys(sd)
    for k in pk:
        sd[k].requires_grad_(True)
    ins = [t(g[f"in{i}"]).clone().requires_grad_(True) for i in range(2) if f"in{i}" in g.files]
    out = fn(ins, sd, True)
    close(out, g["out_train"])
    loss = out.square().mean()
    loss.backward()
    close(loss, g["loss"])
    for i, x in enumerate(ins):
        close(x.grad, g[f"gin{i}"], rtol=1e-4, atol=1e-6)
    for k in pk:
        close(sd[k].grad, g["g." + k[len("blk."):]], rtol=1e-4, atol=2e-6)
    for k, v in sd.items():          # running stats after one train call
        if "running" in k or "num_batches" in k:
            close(v, g["sd1." + k[len("blk."):]])
    with torch.no_grad():
        fn([x.detach() for x in ins], sd, True)
    for k, v in sd.items():
        if "running" in k or "num_batches" in k:
            close(v, g["sd2." + k[len("blk."):]])
    with torch.no_grad():
        close(fn([x.detach() for x in ins], sd, False), g["out_eval"])


def test_state_dict_keys_and_init_match_reference():
    g = load_golden("g2_unet_1_2_n2_32")
    torch.manual_seed(int(g["model_seed"]))
    sd = U.make_state_dict(1, 2)
    assert len(sd) == 118
    pk = U.param_keys(sd)
    assert len(pk) == 64
    sums = np.array([float(sd[k].double().sum()) for k in pk])
    np.testing.assert_allclose(sums, g["weight_sums"], rtol=1e-12, atol=1e-12)
    assert sum(sd[k].numel() for k in pk) == 31037698 - (3 - 1) * 64 * 9 + 0  # UNet(1,2): 1-channel stem


def test_unet_small_spatial_forward_backward():
    g = load_golden("g2_unet_1_2_n2_32")
    torch.manual_seed(int(g["model_seed"]))
    sd = U.clone_sd(U.make_state_dict(1, 2), requires_grad=True)
    x = t(g["x"])
    logits = U.unet_forward(x, sd, train=True)
    close(logits, g["logits"], rtol=1e-4, atol=1e-5)
    loss = logits.square().mean()
    loss.backward()
    close(loss, g["loss"], rtol=1e-5)
    pk = U.param_keys(sd)
    norms = np.array([float(sd[k].grad.double().norm()) for k in pk])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-4, atol=1e-7)
    for j, k in enumerate(pk):
        gf = sd[k].grad.flatten()
        idx = torch.linspace(0, gf.numel() - 1, 16).long()
        np.testing.assert_allclose(gf[idx].numpy(), g["grad_samples"][j], rtol=2e-3, atol=2e-6 + 1e-3 * norms[j] / np.sqrt(gf.numel()))
    rm = np.array([float(v.double().sum()) for k, v in sd.items() if k.endswith("running_mean")])
    rv = np.array([float(v.double().sum()) for k, v in sd.items() if k.endswith("running_var")])
    np.testing.assert_allclose(rm, g["rm_sums"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rv, g["rv_sums"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["g3_unet_3_2_n4_256", "g3_unet_1_2_n2_384", "g3_unet_1_4_n2_288"])
def test_unet_full_size_forward(name):
    g = load_golden(name)
    n, c, h, _, k = [int(v) for v in g["shape"]]
    torch.manual_seed(int(g["model_seed"]))
    sd = U.make_state_dict(c, k)
    gen = torch.Generator().manual_seed(int(g["input_seed"]))
    x = torch.randint(0, 256, (n, c, h, h), generator=gen).float() / 127.5 - 1
    with torch.no_grad():
        logits = U.unet_forward(x, sd, train=True)
    flat = logits.flatten()
    idx = t(g["sample_idx"])
    close(flat[idx], g["sample_val"], rtol=1e-4, atol=1e-5)
    assert abs(float(flat.double().norm()) - float(g["logit_l2"])) <= 1e-5 * float(g["logit_l2"])
    am = logits.argmax(1).numpy().astype(np.uint8)
    if k == 2:
        got = np.packbits(am)
        diff = np.unpackbits(got ^ g["argmax"]).sum()
    else:
        got = am.reshape(-1)[:: max(1, am.size // 65536)]
        diff = int((got != g["argmax"]).sum())
    # argmax can only differ where two logits are within float rounding of each other
    assert diff <= 2, diff


def test_unet_bilinear_against_the_reference():
    """G13 (round 6): the reference's UNet(bilinear=True) itself -- unet_model.py:17-22, unet_parts.py:48-51 -- at full width on a
    2 x 3 x 96 x 136 input (136 -> 17 -> 8: an odd halving, so MaxPool2d drops a column and F.pad places the interpolated map):
    the oracle reproduces its initial weights (RNG order, parameter shapes, key count), train-mode logits, arg-max map, loss,
    gradients and BatchNorm running statistics."""
    g = load_golden("g13_unet_bilinear_3_2_n2_96x136")
    n, c, h, w, k = [int(v) for v in g["shape"]]
    torch.manual_seed(int(g["model_seed"]))
    sd = U.make_state_dict(c, k, bilinear=True)
    assert len(sd) == int(g["n_state_keys"])
    pk = U.param_keys(sd)
    assert len(pk) == 56 == len(g["param_shapes"])
    for kk, shp in zip(pk, g["param_shapes"]):
        assert list(sd[kk].shape) == [int(v) for v in shp[:sd[kk].dim()]], kk
    np.testing.assert_allclose([float(sd[kk].double().sum()) for kk in pk], g["weight_sums"], rtol=1e-6, atol=1e-6)
    gen = torch.Generator().manual_seed(int(g["input_seed"]))
    x = torch.randint(0, 256, (n, c, h, w), generator=gen).float() / 127.5 - 1
    ref = U.clone_sd(sd, requires_grad=True)
    logits = U.unet_forward(x, ref, train=True, bilinear=True)
    loss = logits.square().mean()
    loss.backward()
    flat = logits.detach().flatten()
    close(flat[t(g["sample_idx"])], g["sample_val"], rtol=1e-4, atol=1e-5)
    assert abs(float(flat.double().norm()) - float(g["logit_l2"])) <= 1e-5 * float(g["logit_l2"])
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    diff = np.unpackbits(np.packbits(logits.detach().argmax(1).numpy().astype(np.uint8)) ^ g["argmax"]).sum()
    assert diff <= 2, diff
    norms = np.array([float(ref[kk].grad.double().norm()) for kk in pk])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=2e-3, atol=1e-9)
    rm = np.array([float(v.double().sum()) for kk, v in ref.items() if kk.endswith("running_mean")])
    rv = np.array([float(v.double().sum()) for kk, v in ref.items() if kk.endswith("running_var")])
    np.testing.assert_allclose(rm, g["rm_sums"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rv, g["rv_sums"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("K", [2, 4])
def test_dice_loss_with_mask(K):
    g = load_golden("g4_losses")
    cases = {"sm": dict(target=t(g[f"K{K}.tgt"]), softmax=True),
             "sm_mask": dict(target=t(g[f"K{K}.tgt"]), mask=t(g[f"K{K}.mask"]), softmax=True),
             "sg": dict(target=t(g[f"K{K}.tgt_ml"]).unsqueeze(1), sigmoid=True, multi=True),
             "sg_mask": dict(target=t(g[f"K{K}.tgt_ml"]).unsqueeze(1), mask=t(g[f"K{K}.mask_ml"]), sigmoid=True, multi=True)}
    for tag, kw in cases.items():
        lg = t(g[f"K{K}.logits"]).clone().requires_grad_(True)
        val = L.dice_loss_with_mask(lg, n_classes=K, **kw)
        val.backward()
        close(val, g[f"K{K}.{tag}.val"], rtol=1e-5)
        close(lg.grad, g[f"K{K}.{tag}.grad"], rtol=1e-4, atol=1e-8)


def test_ce_bce_terms_and_q4():
    g = load_golden("g4_losses")
    lg, tg, m = t(g["ce.logits"]), t(g["ce.tgt"]), t(g["ce.mask"])
    close(L.ce_none(lg, tg), g["ce.none"], rtol=1e-5, atol=1e-6)
    close((L.ce_none(lg, tg) * m.squeeze(1)).mean(), g["ce.masked_mean"], rtol=1e-5)
    close(L.bce_logits_none(lg, t(g["bce.tgt"])), g["bce.none"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(g["q4_mask_onehot"][0, 0], np.ones((2, 2), np.float32))      # class 0: all ones
    assert np.array_equal(g["q4_mask_onehot"][0, 1], np.array([[0, 1], [1, 0]], np.float32))


def test_ramps_and_schedules():
    g = load_golden("g5_ramps")
    np.testing.assert_allclose([H.sigmoid_rampup(e, 200) for e in range(201)], g["table"], rtol=1e-14)
    assert H.sigmoid_rampup(3, 0) == float(g["zero_len"][0])
    np.testing.assert_allclose([H.sigmoid_rampup(-5, 200), H.sigmoid_rampup(500, 200)], g["beyond"], rtol=1e-14)
    got = [H.consistency_weight(int(i), 30000) for i in g["iters"]]
    np.testing.assert_allclose(got, g["weight_at_iters"], rtol=1e-14)
    g7 = load_golden("g7_train_helpers")
    np.testing.assert_allclose([1.0 * H.sigmoid_rampup(e, 200.0) for e in (0, 50, 100, 200)], g7["cw"], rtol=1e-14)


def test_numpy_dice_metrics():
    g = load_golden("g6_metrics")
    a, b = g["bin.pred"], g["bin.tgt"]
    np.testing.assert_allclose([H.dice_binary(a[i], b[i]) for i in range(len(a))], g["bin.each"], rtol=1e-14)
    assert H.dice_binary(a[0], b[0]) == 0.0
    af, bf = a.astype(np.float32), b.astype(np.float32)
    np.testing.assert_allclose(H.dice_coeff(af, bf), g["coeff.mean"], rtol=1e-14)
    np.testing.assert_allclose(H.dice_coeff(af, bf, ret_arr=True), g["coeff.arr"], rtol=1e-14)
    np.testing.assert_allclose(H.dice_coeff_2label(g["l2.pred"], g["l2.tgt"]), g["l2.mean"], rtol=1e-14)
    np.testing.assert_allclose(H.dice_coeff_2label(g["l2.pred"], g["l2.tgt"], ret_arr=True), g["l2.arr"], rtol=1e-14)
    np.testing.assert_allclose(H.dice_coeff_3label(g["l3.pred"], g["l3.tgt"]), g["l3.mean"], rtol=1e-14)
    np.testing.assert_allclose(H.dice_coeff_3label(g["l3.pred"], g["l3.tgt"], ret_arr=True), g["l3.arr"], rtol=1e-14)


def test_fft_amplitude_mix():
    g = load_golden("g7_train_helpers")
    np.testing.assert_allclose(H.amp_spectrum(g["fft.trg"]), g["fft.amp_trg"], rtol=1e-12)
    for Lw in (0.01, 0.1):
        for deg in (0.0, 0.5, 1.0):
            random.seed(99)
            out = H.freq_mix(g["fft.src"].copy(), g["fft.amp_trg"], L=Lw, degree=deg)
            np.testing.assert_allclose(out, g[f"fft.out.L{Lw}.d{deg}"], rtol=1e-9, atol=1e-9)


def test_ema_alpha_and_sgd_trajectory():
    g = load_golden("g7_train_helpers")
    for step in (0, 1, 200):
        a = H.ema_alpha(step, 0.99)
        np.testing.assert_allclose(a * g["ema.t.w"] + (1 - a) * g["ema.s.w"], g[f"ema.out{step}.w"], rtol=1e-6, atol=1e-7)
    g8 = load_golden("g8_sgd")
    from oracle.step_ref import RefTrainer
    sd = {"w": t(g8["p0"]).clone()}
    tr = RefTrainer.__new__(RefTrainer)
    tr.student, tr.pkeys, tr.mom = {"w": sd["w"].requires_grad_(True)}, ["w"], {"w": None}
    tr.wd, tr.momentum, tr.lr = 1e-4, 0.9, 0.03
    for s in range(3):
        tr.student["w"].grad = t(g8[f"g{s}"]).clone()
        tr._sgd()
        close(tr.student["w"], g8[f"p{s + 1}"], rtol=1e-6, atol=1e-7)
        tr.lr = H.poly_lr(0.03, s, 30000)


# ---- G10: DeepLabV2-ResNet --------------------------------------------------------------------
import pytest as _pytest


@_pytest.mark.parametrize("name,arch", [("g10_deeplabv2_r50_n2_96x80", "resnet50"), ("g10_deeplabv2_r101_n1_128", "resnet101")])
def test_deeplab_oracle_matches_reference_goldens(name, arch):
    """oracle/deeplab_ref.py against outputs captured from the reference's networks/deeplabv2.py: the mirror's initial weights
    (same seed) are the reference's, train-mode logits / feature norms / running statistics and eval-mode logits agree."""
    from oracle import deeplab_ref as D
    g = load_golden(name)
    n, _, h, w, k = [int(v) for v in g["shape"]]
    sd = D.make_state_dict(arch, k, int(g["model_seed"]))
    pk = [kk for kk, v in sd.items() if v.is_floating_point() and "running" not in kk]
    np.testing.assert_allclose(np.array([float(sd[kk].double().sum()) for kk in pk]), g["weight_sums"], rtol=1e-12, atol=1e-12)
    gen = torch.Generator().manual_seed(int(g["input_seed"]))
    x = torch.randint(0, 256, (n, 3, h, w), generator=gen).float() / 127.5 - 1
    with torch.no_grad():
        sdf = {kk: v.clone() for kk, v in sd.items()}
        feats = D.backbone_features(x, sdf, arch, True)
        np.testing.assert_allclose([float(f.double().norm()) for f in feats], g["feat_l2"], rtol=1e-4)
        assert [list(f.shape) for f in feats] == g["feat_shape"].tolist()
        logits = D.deeplabv2_forward(x, sd, arch, True)
        flat = logits.flatten()
        idx = torch.from_numpy(g["sample_idx"])
        # (two f32 evaluations of 53 BatchNorm layers -- explicit scale/shift here, F.batch_norm there -- differ by ~1e-4 rel)
        rl2 = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
        assert rl2(flat[idx].numpy(), g["sample_val"]) < 5e-4
        assert abs(float(flat.double().norm()) - float(g["logit_l2"])) <= 1e-4 * float(g["logit_l2"])
        np.testing.assert_allclose([float(v.double().sum()) for kk, v in sd.items() if kk.endswith("running_mean")], g["rm_sums"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose([float(v.double().sum()) for kk, v in sd.items() if kk.endswith("running_var")], g["rv_sums"], rtol=1e-4, atol=1e-5)
        ev = D.deeplabv2_forward(x, sd, arch, False).flatten()
        assert rl2(ev[idx].numpy(), g["eval_val"]) < 5e-4


def test_deeplab_oracle_gradients_match_reference_golden():
    """oracle/deeplab_ref.py under torch autograd in float64 against the gradients captured from the reference's DeepLabV2 itself
    in float64 (G10b): every parameter's gradient norm and 64 sampled entries, 1e-7 relative (two float64 evaluations)."""
    from oracle import deeplab_ref as D
    g = load_golden("g10b_deeplabv2_r50_n2_96x80_bwd")
    n, _, h, w, k = [int(v) for v in g["shape"]]
    sd = D.make_state_dict("resnet50", k, int(g["model_seed"]))
    gen = torch.Generator().manual_seed(int(g["input_seed"]))
    x = torch.randint(0, 256, (n, 3, h, w), generator=gen).float() / 127.5 - 1
    R = torch.randn(n, k, h, w, generator=gen)
    sdo = {}
    for kk, v in sd.items():
        v = v.clone().double() if v.is_floating_point() else v.clone()
        sdo[kk] = v.requires_grad_(True) if v.is_floating_point() and "running" not in kk else v
    out = D.deeplabv2_forward(x.double(), sdo, "resnet50", True)
    (out * R.double()).sum().backward()
    assert abs(float(out.detach().norm()) - float(g["logit_l2"])) <= 1e-9 * float(g["logit_l2"])
    names = [str(s) for s in g["names"]]
    assert names == [kk for kk, v in sdo.items() if v.requires_grad]
    for i, kk in enumerate(names):
        flat = sdo[kk].grad.flatten()
        assert abs(float(flat.norm()) - g["grad_l2"][i]) <= 1e-7 * g["grad_l2"][i] + 1e-12, kk
        np.testing.assert_allclose(flat[torch.from_numpy(g["sample_idx"][i])].numpy(), g["sample_val"][i], rtol=1e-6,
                                   atol=1e-7 * g["grad_l2"][i], err_msg=kk)


@_pytest.mark.parametrize("name", ["g3b_unet_1_2_n2_384_bwd", "g3b_unet_1_4_n2_288_bwd"])
def test_unet_full_size_backward_oracle(name):
    """Round 3: the oracle's forward + backward at configs[2] / configs[3]'s real extents against the reference's own
    (gradient norms, 16 samples per tensor, running statistics): pins oracle/unet_ref.py where the GPU test uses it."""
    g = load_golden(name)
    n, c, h, _, k = [int(v) for v in g["shape"]]
    torch.manual_seed(int(g["model_seed"]))
    sd = U.clone_sd(U.make_state_dict(c, k), requires_grad=True)
    gen = torch.Generator().manual_seed(int(g["input_seed"]))
    x = torch.randint(0, 256, (n, c, h, h), generator=gen).float() / 127.5 - 1
    logits = U.unet_forward(x, sd, train=True)
    loss = logits.square().mean()
    loss.backward()
    close(loss, g["loss"], rtol=1e-5)
    pk = U.param_keys(sd)
    norms = np.array([float(sd[kk].grad.double().norm()) for kk in pk])
    np.testing.assert_allclose(norms, g["grad_norms"], rtol=1e-3, atol=1e-9)
    samples = np.stack([sd[kk].grad.flatten()[torch.linspace(0, sd[kk].numel() - 1, 16).long()].numpy() for kk in pk])
    errs = np.linalg.norm(samples - g["grad_samples"], axis=1) / (np.linalg.norm(g["grad_samples"], axis=1) + 1e-30)
    assert float(np.median(errs)) < 5e-3 and float(errs.max()) < 5e-2, (float(np.median(errs)), float(errs.max()))
    rm = np.array([float(v.double().sum()) for kk, v in sd.items() if kk.endswith("running_mean")])
    rv = np.array([float(v.double().sum()) for kk, v in sd.items() if kk.endswith("running_var")])
    np.testing.assert_allclose(rm, g["rm_sums"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rv, g["rv_sums"], rtol=1e-4, atol=1e-5)


def test_deeplab_oracle_matches_the_512_golden():
    """Round 3: configs[4]'s real extent (ResNet-101, one 512 x 512 image) -- the oracle against the reference's train-mode logits,
    feature norms and running statistics (g10_deeplabv2_r101_n1_512)."""
    test_deeplab_oracle_matches_reference_goldens("g10_deeplabv2_r101_n1_512", "resnet101")


@_pytest.mark.parametrize("K", [2, 4])
def test_dice_loss_with_mask_remaining_modes(K):
    """Round 3: the DiceLossWithMask combinations no reference script calls (losses.py:236-268) -- class
    weights, sigmoid per class (5-D target), softmax + multi (full and class-broadcast targets), raw inputs -- values and
    gradients captured from the reference (tools/gen_goldens.py g4b)."""
    g = load_golden("g4b_losses_rest")
    tgt, mask = t(g[f"K{K}.tgt"]), t(g[f"K{K}.mask"])
    tml, mml = t(g[f"K{K}.tgt_ml"]), t(g[f"K{K}.mask_ml"])
    w = [float(v) for v in g[f"K{K}.weight"]]
    cases = {"sm_w": dict(target=tgt, softmax=True, weight=w), "sm_mask_w": dict(target=tgt, mask=mask, softmax=True, weight=w),
             "sg_pc": dict(target=tgt.unsqueeze(1), sigmoid=True), "sg_pc_mask_w": dict(target=tgt.unsqueeze(1), mask=mask, sigmoid=True, weight=w),
             "sm_multi": dict(target=tml, softmax=True, multi=True), "sm_multi_mask": dict(target=tml, mask=mml, softmax=True, multi=True),
             "sm_multi_bcast": dict(target=tgt.float(), mask=mask, softmax=True, multi=True),
             "raw_pc": dict(target=tgt), "raw_pc_mask_w": dict(target=tgt, mask=mask, weight=w),
             "raw_multi": dict(target=tml, multi=True), "raw_multi_mask": dict(target=tml, mask=mml, multi=True)}
    for tag, kw in cases.items():
        lg = t(g[f"K{K}.logits"]).clone().requires_grad_(True)
        val = L.dice_loss_with_mask(lg, n_classes=K, **kw)
        val.backward()
        close(val, g[f"K{K}.{tag}.val"], rtol=1e-5)
        close(lg.grad, g[f"K{K}.{tag}.grad"], rtol=1e-4, atol=1e-8)
