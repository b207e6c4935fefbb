"""Validation on the device: the reference's `test()` (train.py:253-395, train_mnms.py twin, test.py:64-206).

Eval-mode forward through the HIP U-Net (BatchNorm from the running statistics, applied on load by the consuming
kernel: no separate normalisation pass), prediction and the per-sample overlap counts on the device
(ustrun_pseudo_label, ustrun_dice_counts), so one [N, parts, 3] int32 copy per batch reaches the host instead of the
logits and masks the reference moves with .cpu().  Dice and its averaging (per batch, per domain loader, over the
domains) are the reference's.  The medpy metrics it prints beside the Dice (jc / hd95 / asd) are outside this build;
the per-batch loss it computes is never accumulated there and is not computed here.
"""
import logging

import numpy as np
import torch

from utils import metrics

from . import functional as F
from .trainer import DATASETS, decode_labels

PARTS = {"fundus": ["cup", "disc"], "prostate": ["base"], "BUSI": ["base"], "MNMS": ["lv", "myo", "rv"]}


def predict(dataset, logits):
    """Device prediction: fundus -> f32 {0,1} [N,2,H,W] (sigmoid >= .5 per channel); others -> int64 arg-max [N,H,W]
    (first index on ties), train.py:292-299."""
    mode = DATASETS[dataset][3]
    return F.pseudo_label(logits, 0.5, mode)[0]


def batch_dice(dataset, pred, mask):
    """Per-part Dice of one batch, averaged over its samples (utils/metrics.py:149-231 without ret_arr)."""
    if dataset == "MNMS":
        cnt = F.dice_counts(pred, mask, by_class=True, n_classes=3)
    else:
        cnt = F.dice_counts(pred, mask)
    c = cnt.cpu().numpy().astype(np.float64)              # [N, parts, 3]
    d = metrics.dice_from_counts(c[..., 0], c[..., 1], c[..., 2])
    return [float(sum(d[:, p]) / len(d)) for p in range(d.shape[1])]


@torch.no_grad()
def validate(dataset, model, loaders, epoch=0, log=logging.info):
    """loaders: one iterable of (image, raw label) batches per domain (any device; moved to the model's).
    Returns (val_dice[parts], per_domain[domain][parts]); leaves the model in train mode, as the reference does."""
    part = PARTS[dataset]
    dev = next(model.parameters()).device
    model.eval()
    val = [0.0] * len(part)
    per_domain = []
    for i, loader in enumerate(loaders):
        dom, nb = [0.0] * len(part), 0
        for image, label in loader:
            image, label = image.to(dev), label.to(dev)
            mask = decode_labels(dataset, label)
            dice = batch_dice(dataset, predict(dataset, model(image)), mask)
            for p in range(len(part)):
                dom[p] += dice[p]
            nb += 1
        dom = [d / max(nb, 1) for d in dom]
        per_domain.append(dom)
        for p in range(len(part)):
            val[p] += dom[p]
        if log:
            log("domain%d epoch %d :\n\t%s" % (i + 1, epoch, "".join("val_%s_dice: %f, " % (n, dom[k]) for k, n in enumerate(part))))
    model.train()
    val = [v / max(len(loaders), 1) for v in val]
    if log:
        log("epoch %d :\n\t%s" % (epoch, "".join("val_%s_dice: %f, " % (n, val[k]) for k, n in enumerate(part))))
    return val, per_domain
