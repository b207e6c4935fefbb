#!/usr/bin/env python3
"""Generate golden vectors from the reference itself (dev-time tool, build container only).

Reads /root/reference IN PLACE via sys.path (nothing is copied), runs the reference's own
modules on seeded inputs with torch-CPU, and writes small .npz fixtures under tests/golden/.
The fixtures are data only: inputs, seeds, expected outputs.  Run:

    PYTHONDONTWRITEBYTECODE=1 python3 -B tools/gen_goldens.py

Fixture groups follow SURVEY.md 8c (G1-G7).
"""
import os
import random
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from networks.unet_model import UNet                      # noqa: E402
from networks.unet_parts import DoubleConv, Down, OutConv, Up   # noqa: E402
from utils import losses, metrics, ramps                  # noqa: E402

META = dict(torch_version=torch.__version__, numpy_version=np.__version__)


def save(name, **arrs):
    arrs = {k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items()}
    arrs["_torch_version"] = np.array(META["torch_version"])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def sd_arrays(mod, prefix="sd."):
    return {prefix + k: v.detach().clone() for k, v in mod.state_dict().items()}


def block_case(name, mod, inputs):
    """Train-mode call x2 (running stats after each), grads of out.square().mean(), then eval call."""
    rec = {}
    rec.update(sd_arrays(mod, "sd0."))
    ins = [t.clone().requires_grad_(True) for t in inputs]
    for i, t in enumerate(inputs):
        rec[f"in{i}"] = t
    mod.train()
    out = mod(*ins)
    rec["out_train"] = out
    loss = out.square().mean()
    loss.backward()
    rec["loss"] = loss
    for i, t in enumerate(ins):
        rec[f"gin{i}"] = t.grad
    for k, p in mod.named_parameters():
        rec["g." + k] = p.grad
    rec.update(sd_arrays(mod, "sd1."))
    with torch.no_grad():
        mod(*inputs)
    rec.update(sd_arrays(mod, "sd2."))
    mod.eval()
    with torch.no_grad():
        rec["out_eval"] = mod(*inputs)
    save(name, **rec)


def g1_blocks():
    torch.manual_seed(11)
    block_case("g1_doubleconv_3_8", DoubleConv(3, 8), [torch.randn(2, 3, 16, 16)])
    torch.manual_seed(12)
    block_case("g1_doubleconv_8_8_mid4", DoubleConv(8, 8, 4), [torch.randn(2, 8, 16, 16)])
    torch.manual_seed(13)
    block_case("g1_down_8_16", Down(8, 16), [torch.randn(2, 8, 16, 16)])
    torch.manual_seed(14)
    block_case("g1_up_16_8_convT", Up(16, 8, False), [torch.randn(2, 16, 8, 8), torch.randn(2, 8, 16, 16)])
    torch.manual_seed(15)
    block_case("g1_up_16_8_bilinear", Up(16, 8, True), [torch.randn(2, 8, 8, 8), torch.randn(2, 8, 16, 16)])
    torch.manual_seed(16)
    block_case("g1_up_16_8_convT_odd", Up(16, 8, False), [torch.randn(2, 16, 8, 8), torch.randn(2, 8, 17, 19)])
    torch.manual_seed(17)
    block_case("g1_outconv_8_2", OutConv(8, 2), [torch.randn(2, 8, 16, 16)])
    torch.manual_seed(18)
    block_case("g1_down_8_16_odd", Down(8, 16), [torch.randn(1, 8, 15, 13)])


def weight_sums(model):
    return np.array([float(p.detach().double().sum()) for p in model.parameters()])


def g2_unet_small_spatial():
    """Full-width UNet(1,2) on N2 x 32x32: full logits, loss, per-parameter grad norms + samples."""
    torch.manual_seed(2024)
    model = UNet(n_channels=1, n_classes=2)
    wsum = weight_sums(model)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(2, 1, 32, 32, generator=g)
    model.train()
    logits = model(x)
    loss = logits.square().mean()
    loss.backward()
    norms, samples = [], []
    for p in model.parameters():
        gflat = p.grad.flatten()
        norms.append(float(gflat.double().norm()))
        idx = torch.linspace(0, gflat.numel() - 1, 16).long()
        samples.append(gflat[idx].numpy())
    bufs = {k: v for k, v in model.state_dict().items() if "running" in k}
    save("g2_unet_1_2_n2_32", model_seed=2024, input_seed=77, x=x, logits=logits, loss=loss,
         weight_sums=wsum, grad_norms=np.array(norms), grad_samples=np.stack(samples),
         rm_sums=np.array([float(v.double().sum()) for k, v in bufs.items() if k.endswith("running_mean")]),
         rv_sums=np.array([float(v.double().sum()) for k, v in bufs.items() if k.endswith("running_var")]))


def g3_unet_full(name, c, k, n, h):
    torch.manual_seed(1337)
    model = UNet(n_channels=c, n_classes=k)
    wsum = weight_sums(model)
    g = torch.Generator().manual_seed(1337)
    x = torch.randint(0, 256, (n, c, h, h), generator=g).float() / 127.5 - 1
    model.train()
    with torch.no_grad():
        logits = model(x)
    flat = logits.flatten()
    idx = torch.randperm(flat.numel(), generator=g)[:4096]
    amax = logits.argmax(1).numpy().astype(np.uint8)
    packed = np.packbits(amax) if k == 2 else amax.reshape(-1)[:: max(1, amax.size // 65536)]
    save(name, model_seed=1337, input_seed=1337, shape=np.array([n, c, h, h, k]), weight_sums=wsum,
         logit_sum=float(flat.double().sum()), logit_abs_sum=float(flat.double().abs().sum()),
         logit_l2=float(flat.double().norm()), sample_idx=idx, sample_val=flat[idx], argmax=packed)


def g3b_unet_backward(name, c, k, n, h):
    """One forward + backward of the reference UNet at a FULL-SIZE shape (same weights and input as G3): per-parameter
    gradient norms + 16 samples per tensor for loss = logits.square().mean(), BN running-statistic sums after the call."""
    torch.manual_seed(1337)
    model = UNet(n_channels=c, n_classes=k)
    g = torch.Generator().manual_seed(1337)
    x = torch.randint(0, 256, (n, c, h, h), generator=g).float() / 127.5 - 1
    model.train()
    logits = model(x)
    loss = logits.square().mean()
    loss.backward()
    norms, samples = [], []
    for p in model.parameters():
        gflat = p.grad.flatten()
        norms.append(float(gflat.double().norm()))
        samples.append(gflat[torch.linspace(0, gflat.numel() - 1, 16).long()].numpy())
    bufs = model.state_dict()
    save(name, model_seed=1337, input_seed=1337, shape=np.array([n, c, h, h, k]), loss=loss.detach(),
         logit_l2=float(logits.detach().double().norm()), grad_norms=np.array(norms), grad_samples=np.stack(samples),
         rm_sums=np.array([float(v.double().sum()) for kk, v in bufs.items() if kk.endswith("running_mean")]),
         rv_sums=np.array([float(v.double().sum()) for kk, v in bufs.items() if kk.endswith("running_var")]))


def g13_unet_bilinear(name, c, k, n, h, w):
    """The reference's UNet(bilinear=True) (unet_model.py:17-22, unet_parts.py:48-51) at full width on a seeded input whose width
    halves to an odd extent (136 -> 17 -> 8: MaxPool2d drops a column, F.pad places the 16-wide interpolated map in the 17-wide skip):
    train-mode logits (checksums, 4096 samples, the arg-max map), then gradient norms + 16 samples per parameter tensor for
    loss = logits.square().mean() and the BatchNorm running-statistic sums -- the state_dict's key list rides along (56 parameters, no
    up.weight): pins the halved channel plan and the interpolation against the reference itself (round 6)."""
    torch.manual_seed(1337)
    model = UNet(n_channels=c, n_classes=k, bilinear=True)
    wsum = weight_sums(model)
    g = torch.Generator().manual_seed(1337)
    x = torch.randint(0, 256, (n, c, h, w), generator=g).float() / 127.5 - 1
    model.train()
    logits = model(x)
    loss = logits.square().mean()
    loss.backward()
    flat = logits.detach().flatten()
    idx = torch.randperm(flat.numel(), generator=g)[:4096]
    amax = logits.detach().argmax(1).numpy().astype(np.uint8)
    norms, samples = [], []
    for p in model.parameters():
        gflat = p.grad.flatten()
        norms.append(float(gflat.double().norm()))
        samples.append(gflat[torch.linspace(0, gflat.numel() - 1, 16).long()].numpy())
    bufs = model.state_dict()
    shapes = np.array([list(v.shape) + [0] * (4 - v.dim()) for v in model.parameters()])
    save(name, model_seed=1337, input_seed=1337, shape=np.array([n, c, h, w, k]), weight_sums=wsum, param_shapes=shapes,
         n_state_keys=len(bufs), logit_sum=float(flat.double().sum()), logit_l2=float(flat.double().norm()), sample_idx=idx,
         sample_val=flat[idx], argmax=np.packbits(amax), loss=loss.detach(), grad_norms=np.array(norms), grad_samples=np.stack(samples),
         rm_sums=np.array([float(v.double().sum()) for kk, v in bufs.items() if kk.endswith("running_mean")]),
         rv_sums=np.array([float(v.double().sum()) for kk, v in bufs.items() if kk.endswith("running_var")]))


def g10_deeplab(name, arch, nclass, n, h, w, seed=1337):
    """The reference's DeepLabV2 (networks/deeplabv2.py) on a seeded input: train-mode logits (samples + checksums), the
    backbone feature norms, running-statistic sums after that call, then eval-mode logits.  The constructor always loads
    ../../checkpoints/pretrained/<arch>.pth (base.py:12, resnet.py:179-181), which does not exist here: torch.load is
    stubbed to return an empty dict for the construction (load_state_dict(strict=False) then keeps the seeded init)."""
    from networks.deeplabv2 import DeepLabV2
    real_load = torch.load
    torch.load = lambda *a, **k: {}
    try:
        torch.manual_seed(seed)
        model = DeepLabV2(arch, nclass)
    finally:
        torch.load = real_load
    wsum = weight_sums(model)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randint(0, 256, (n, 3, h, w), generator=g).float() / 127.5 - 1
    model.train()
    with torch.no_grad():
        feats = model.backbone.base_forward(x)
    # a fresh model for the logits so that the running statistics below are those of exactly ONE train-mode call
    torch.load = lambda *a, **k: {}
    try:
        torch.manual_seed(seed)
        model = DeepLabV2(arch, nclass)
    finally:
        torch.load = real_load
    model.train()
    with torch.no_grad():
        logits = model(x)
    sd = model.state_dict()
    rm = np.array([float(v.double().sum()) for k, v in sd.items() if k.endswith("running_mean")])
    rv = np.array([float(v.double().sum()) for k, v in sd.items() if k.endswith("running_var")])
    model.eval()
    with torch.no_grad():
        logits_eval = model(x)
    flat, flat_e = logits.flatten(), logits_eval.flatten()
    idx = torch.randperm(flat.numel(), generator=g)[:4096]
    save(name, model_seed=seed, input_seed=seed + 1, shape=np.array([n, 3, h, w, nclass]), weight_sums=wsum,
         feat_l2=np.array([float(f.double().norm()) for f in feats]), feat_shape=np.array([list(f.shape) for f in feats]),
         logit_l2=float(flat.double().norm()), logit_sum=float(flat.double().sum()), sample_idx=idx, sample_val=flat[idx],
         eval_l2=float(flat_e.double().norm()), eval_val=flat_e[idx], rm_sums=rm, rv_sums=rv)


def g10b_deeplab_backward(name, arch, nclass, n, h, w, seed=1337):
    """The reference's DeepLabV2 under autograd, in FLOAT64 (the float32 gradient of fifty train-mode BatchNorms at batch 2 is
    only good to ~2e-2, see tests/test_gpu_deeplab_bwd.py): train-mode forward, loss = <logits, R>, every parameter's gradient
    as its L2 norm plus 64 sampled entries."""
    from networks.deeplabv2 import DeepLabV2
    real_load = torch.load
    torch.load = lambda *a, **k: {}
    try:
        torch.manual_seed(seed)
        model = DeepLabV2(arch, nclass)
    finally:
        torch.load = real_load
    wsum = weight_sums(model)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randint(0, 256, (n, 3, h, w), generator=g).float() / 127.5 - 1
    R = torch.randn(n, nclass, h, w, generator=g)
    model = model.double().train()
    out = model(x.double())
    (out * R.double()).sum().backward()
    names, norms, idxs, vals = [], [], [], []
    for k, p in model.named_parameters():
        flat = p.grad.flatten()
        idx = torch.randperm(flat.numel(), generator=g)[:64]
        if idx.numel() < 64:
            idx = torch.cat([idx, idx.new_zeros(64 - idx.numel())])
        names.append(k); norms.append(float(flat.norm())); idxs.append(idx.numpy()); vals.append(flat[idx].numpy())
    save(name, model_seed=seed, input_seed=seed + 1, shape=np.array([n, 3, h, w, nclass]), weight_sums=wsum, names=np.array(names),
         grad_l2=np.array(norms), sample_idx=np.stack(idxs), sample_val=np.stack(vals), logit_l2=float(out.detach().norm()))


def g4_dice():
    rec = {}
    for K in (2, 4):
        torch.manual_seed(40 + K)
        dl = losses.DiceLossWithMask(K)
        logits = torch.randn(2, K, 16, 16)
        tgt = torch.randint(0, K, (2, 1, 16, 16))
        mask = (torch.rand(2, 1, 16, 16) > 0.4).float()
        tgt_ml = (torch.rand(2, K, 16, 16) > 0.5).float()
        mask_ml = (torch.rand(2, K, 16, 16) > 0.4).float()
        rec[f"K{K}.logits"], rec[f"K{K}.tgt"], rec[f"K{K}.mask"] = logits, tgt, mask
        rec[f"K{K}.tgt_ml"], rec[f"K{K}.mask_ml"] = tgt_ml, mask_ml
        for tag, kw in (("sm", dict(target=tgt, softmax=True)),
                        ("sm_mask", dict(target=tgt, mask=mask, softmax=True)),
                        ("sg", dict(target=tgt_ml.unsqueeze(1), sigmoid=True, multi=True)),
                        ("sg_mask", dict(target=tgt_ml.unsqueeze(1), mask=mask_ml, sigmoid=True, multi=True))):
            lg = logits.clone().requires_grad_(True)
            val = dl(lg, **kw)
            val.backward()
            rec[f"K{K}.{tag}.val"], rec[f"K{K}.{tag}.grad"] = val, lg.grad
    # Q4: class-0 channel of the mask one-hot is all ones
    dl = losses.DiceLossWithMask(2)
    rec["q4_mask_onehot"] = dl._one_hot_mask_encoder(torch.tensor([[[[0.0, 1.0], [1.0, 0.0]]]]))
    # CE / BCE terms exactly as the step evaluates them (train.py:516-519,829)
    torch.manual_seed(49)
    lg = torch.randn(2, 4, 16, 16)
    t = torch.randint(0, 4, (2, 16, 16))
    m = (torch.rand(2, 1, 16, 16) > 0.4).float()
    rec["ce.logits"], rec["ce.tgt"], rec["ce.mask"] = lg, t, m
    rec["ce.none"] = torch.nn.CrossEntropyLoss(reduction="none")(lg, t)
    rec["ce.masked_mean"] = (rec["ce.none"] * m.squeeze(1)).mean()
    tb = (torch.rand(2, 4, 16, 16) > 0.5).float()
    rec["bce.tgt"] = tb
    rec["bce.none"] = torch.nn.BCEWithLogitsLoss(reduction="none")(lg, tb)
    save("g4_losses", **rec)


def g4b_dice_rest():
    """The DiceLossWithMask mode combinations no reference script calls (losses.py:236-268): class weights, sigmoid per class
    (the target then arrives 5-D: squeeze(1) must leave [B,1,H,W] for the one-hot encoder), softmax + multi (one global Dice
    over the soft-maxed maps, target broadcast over classes or full), and raw inputs (no activation) in both forms; value and
    gradient with respect to the logits each, masks included (Q4 applies to the per-class forms)."""
    rec = {}
    for K in (2, 4):
        torch.manual_seed(140 + K)
        dl = losses.DiceLossWithMask(K)
        logits = torch.randn(2, K, 16, 16)
        tgt = torch.randint(0, K, (2, 1, 16, 16))
        mask = (torch.rand(2, 1, 16, 16) > 0.4).float()
        tgt_ml = (torch.rand(2, K, 16, 16) > 0.5).float()
        mask_ml = (torch.rand(2, K, 16, 16) > 0.4).float()
        w = [0.5 + 0.25 * i for i in range(K)]
        rec[f"K{K}.logits"], rec[f"K{K}.tgt"], rec[f"K{K}.mask"] = logits, tgt, mask
        rec[f"K{K}.tgt_ml"], rec[f"K{K}.mask_ml"], rec[f"K{K}.weight"] = tgt_ml, mask_ml, np.array(w)
        cases = (("sm_w", dict(target=tgt, softmax=True, weight=w)),
                 ("sm_mask_w", dict(target=tgt, mask=mask, softmax=True, weight=w)),
                 ("sg_pc", dict(target=tgt.unsqueeze(1), sigmoid=True)),
                 ("sg_pc_mask_w", dict(target=tgt.unsqueeze(1), mask=mask, sigmoid=True, weight=w)),
                 ("sm_multi", dict(target=tgt_ml, softmax=True, multi=True)),
                 ("sm_multi_mask", dict(target=tgt_ml, mask=mask_ml, softmax=True, multi=True)),
                 ("sm_multi_bcast", dict(target=tgt.float(), mask=mask, softmax=True, multi=True)),
                 ("raw_pc", dict(target=tgt)),
                 ("raw_pc_mask_w", dict(target=tgt, mask=mask, weight=w)),
                 ("raw_multi", dict(target=tgt_ml, multi=True)),
                 ("raw_multi_mask", dict(target=tgt_ml, mask=mask_ml, multi=True)))
        for tag, kw in cases:
            lg = logits.clone().requires_grad_(True)
            val = dl(lg, **kw)
            val.backward()
            rec[f"K{K}.{tag}.val"], rec[f"K{K}.{tag}.grad"] = val, lg.grad
    save("g4b_losses_rest", **rec)


def g5_ramps():
    save("g5_ramps", table=np.array([ramps.sigmoid_rampup(e, 200) for e in range(0, 201)]),
         zero_len=np.array([ramps.sigmoid_rampup(3, 0)]),
         beyond=np.array([ramps.sigmoid_rampup(-5, 200), ramps.sigmoid_rampup(500, 200)]),
         iters=np.array([0, 149, 150, 29999]),
         weight_at_iters=np.array([1.0 * ramps.sigmoid_rampup(i // (30000 / 200.0), 200.0) for i in (0, 149, 150, 29999)]))


def g6_metrics():
    rng = np.random.RandomState(6)
    rec = {}
    a, b = rng.rand(5, 12, 12) > 0.5, rng.rand(5, 12, 12) > 0.5
    a[0] = False
    b[0] = False                                   # both-empty -> 0.0 (Q18)
    a[1] = False
    rec["bin.pred"], rec["bin.tgt"] = a, b
    rec["bin.each"] = np.array([metrics.dice_coefficient_numpy(a[i], b[i]) for i in range(5)])
    rec["coeff.mean"] = np.array(metrics.dice_coeff(a.astype(np.float32), torch.tensor(b.astype(np.float32))))
    rec["coeff.arr"] = np.array(metrics.dice_coeff(a.astype(np.float32), torch.tensor(b.astype(np.float32)), ret_arr=True))
    p2, t2 = (rng.rand(3, 2, 12, 12) > 0.5).astype(np.float32), (rng.rand(3, 2, 12, 12) > 0.5).astype(np.float32)
    rec["l2.pred"], rec["l2.tgt"] = p2, t2
    rec["l2.mean"] = np.array(metrics.dice_coeff_2label(p2, torch.tensor(t2)))
    rec["l2.arr"] = np.array(metrics.dice_coeff_2label(p2, torch.tensor(t2), ret_arr=True))
    p3, t3 = rng.randint(0, 4, (3, 12, 12)), rng.randint(0, 4, (3, 12, 12))
    rec["l3.pred"], rec["l3.tgt"] = p3, t3
    rec["l3.mean"] = np.array(metrics.dice_coeff_3label(p3, torch.tensor(t3)))
    rec["l3.arr"] = np.array(metrics.dice_coeff_3label(p3, torch.tensor(t3), ret_arr=True))
    save("g6_metrics", **rec)


def g7_train_helpers():
    """Pure helpers of train.py, imported with throw-away stubs for the absent third-party modules."""
    for name in ("torchvision", "torchvision.transforms", "torchvision.utils", "tensorboardX", "medpy",
                 "medpy.metric", "cv2", "tqdm"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = types.ModuleType(name)
                sys.modules[name] = m
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision.utils"].make_grid = lambda *a, **k: None
    sys.modules["torchvision.transforms"].Compose = lambda x: x
    sys.modules["tensorboardX"].SummaryWriter = object
    sys.modules["medpy"].metric = sys.modules["medpy.metric"]
    sys.modules["medpy.metric"].binary = types.ModuleType("medpy.metric.binary")
    for name in ("matplotlib", "matplotlib.pyplot", "PIL", "PIL.Image", "PIL.ImageOps", "PIL.ImageFilter",
                 "PIL.ImageEnhance", "scipy.ndimage", "scipy.ndimage.filters", "scipy.ndimage.interpolation"):
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    argv = sys.argv
    sys.argv = ["train.py", "--dataset", "fundus"]
    try:
        import train as T
    finally:
        sys.argv = argv
    rec = {}
    rng = np.random.RandomState(7)
    src = rng.rand(3, 32, 32) * 255
    trg = rng.rand(3, 32, 32) * 255
    rec["fft.src"], rec["fft.trg"] = src, trg
    rec["fft.amp_trg"] = T.extract_amp_spectrum(trg)
    for L in (0.01, 0.1):
        for deg in (0.0, 0.5, 1.0):
            random.seed(99)
            out = T.source_to_target_freq(src.copy(), rec["fft.amp_trg"], L=L, degree=deg)
            random.seed(99)
            rec[f"fft.out.L{L}.d{deg}"] = out
            rec[f"fft.ratio.L{L}.d{deg}"] = np.array(random.uniform(0, deg))
    # update_ema_variables on two tiny models for steps 0, 1, 200
    torch.manual_seed(70)
    a, b = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3)
    rec["ema.s.w"], rec["ema.s.b"] = a.weight.detach().clone(), a.bias.detach().clone()
    rec["ema.t.w"], rec["ema.t.b"] = b.weight.detach().clone(), b.bias.detach().clone()
    for step in (0, 1, 200):
        bb = torch.nn.Linear(4, 3)
        bb.load_state_dict(b.state_dict())
        T.update_ema_variables(a, bb, 0.99, step)
        rec[f"ema.out{step}.w"], rec[f"ema.out{step}.b"] = bb.weight.detach().clone(), bb.bias.detach().clone()
    T.args.consistency, T.args.consistency_rampup = 1.0, 200.0
    rec["cw"] = np.array([T.get_current_consistency_weight(e) for e in (0, 50, 100, 200)])
    save("g7_train_helpers", **rec)


def g8_sgd():
    """torch.optim.SGD(momentum=.9, wd=1e-4) trajectory, as constructed at train.py:512."""
    torch.manual_seed(80)
    p = torch.nn.Parameter(torch.randn(5, 7))
    opt = torch.optim.SGD([p], lr=0.03, momentum=0.9, weight_decay=0.0001)
    rec = {"p0": p.detach().clone()}
    for s in range(3):
        g = torch.randn(5, 7)
        rec[f"g{s}"] = g
        p.grad = g.clone()
        opt.step()
        rec[f"p{s + 1}"] = p.detach().clone()
        for grp in opt.param_groups:
            grp["lr"] = 0.03 * (1.0 - s / 30000) ** 0.9
    save("g8_sgd", **rec)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "g10":          # DeepLabV2-ResNet (round 2)
        g10_deeplab("g10_deeplabv2_r50_n2_96x80", "resnet50", 2, 2, 96, 80)
        g10_deeplab("g10_deeplabv2_r101_n1_128", "resnet101", 2, 1, 128, 128)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g10b":         # DeepLabV2-ResNet backward (round 2)
        g10b_deeplab_backward("g10b_deeplabv2_r50_n2_96x80_bwd", "resnet50", 2, 2, 96, 80)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "r3":           # round 3: configs[2..4] at their real sizes, the rest of DiceLossWithMask
        g10_deeplab("g10_deeplabv2_r101_n1_512", "resnet101", 2, 1, 512, 512)
        g3b_unet_backward("g3b_unet_1_2_n2_384_bwd", 1, 2, 2, 384)
        g3b_unet_backward("g3b_unet_1_4_n2_288_bwd", 1, 4, 2, 288)
        g4b_dice_rest()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g4b":
        g4b_dice_rest()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g13":          # only the round-6 addition
        g13_unet_bilinear("g13_unet_bilinear_3_2_n2_96x136", 3, 2, 2, 96, 136)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g3b":          # only the round-2 addition
        g3b_unet_backward("g3b_unet_3_2_n4_256_bwd", 3, 2, 4, 256)
        sys.exit(0)
    g1_blocks()
    g2_unet_small_spatial()
    g3_unet_full("g3_unet_3_2_n4_256", 3, 2, 4, 256)
    g3_unet_full("g3_unet_1_2_n2_384", 1, 2, 2, 384)
    g3_unet_full("g3_unet_1_4_n2_288", 1, 4, 2, 288)
    g3b_unet_backward("g3b_unet_3_2_n4_256_bwd", 3, 2, 4, 256)
    g13_unet_bilinear("g13_unet_bilinear_3_2_n2_96x136", 3, 2, 2, 96, 136)
    g10_deeplab("g10_deeplabv2_r50_n2_96x80", "resnet50", 2, 2, 96, 80)
    g10_deeplab("g10_deeplabv2_r101_n1_128", "resnet101", 2, 1, 128, 128)
    g10b_deeplab_backward("g10b_deeplabv2_r50_n2_96x80_bwd", "resnet50", 2, 2, 96, 80)
    g4_dice()
    g5_ramps()
    g6_metrics()
    g7_train_helpers()
    g8_sgd()
