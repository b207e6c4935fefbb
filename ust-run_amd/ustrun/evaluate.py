"""Validation on the device: the reference's `test()` (train.py:253-395, train_mnms.py twin, test.py:64-206).

Eval-mode forward through the HIP U-Net (BatchNorm from the running statistics, applied on load by the consuming
kernel: no separate normalisation pass), prediction and the per-sample overlap counts on the device
(ustrun_pseudo_label, ustrun_dice_counts), so one [N, parts, 3] int32 copy per batch reaches the host instead of the
logits and masks the reference moves with .cpu(); consecutive loader batches share one forward (eval mode: samples do
not interact).  Dice and its averaging (per batch, per domain loader, over the
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


def sample_dice(dataset, pred, mask):
    """Per-sample, per-part Dice [N, parts] from device overlap counts (utils/metrics.py:114-146 on every sample)."""
    if dataset == "MNMS":
        cnt = F.dice_counts(pred, mask, by_class=True, n_classes=3)
    else:
        cnt = F.dice_counts(pred, mask)
    c = cnt.cpu().numpy().astype(np.float64)              # [N, parts, 3]
    return metrics.dice_from_counts(c[..., 0], c[..., 1], c[..., 2])


def batch_dice(dataset, pred, mask):
    """Per-part Dice of one batch, averaged over its samples (utils/metrics.py:149-231 without ret_arr)."""
    d = sample_dice(dataset, pred, mask)
    return [float(sum(d[:, p]) / len(d)) for p in range(d.shape[1])]


@torch.no_grad()
def validate(dataset, model, loaders, epoch=0, log=logging.info, coalesce=64):
    """loaders: one iterable of (image, raw label) batches per domain (any device; moved to the model's).
    Returns (val_dice[parts], per_domain[domain][parts]); leaves the model in train mode, as the reference does.

    In eval mode the samples of a batch do not interact (BatchNorm uses the running statistics), so up to `coalesce`
    images of consecutive loader batches go through ONE forward -- the reference's `test_bs` 1 would otherwise leave the
    deep layers with 8-64 workgroups -- and the Dice is still averaged per loader batch, then per domain, then over the
    domains, exactly as train.py:318-372 does."""
    part = PARTS[dataset]
    dev = next(model.parameters()).device
    model.eval()
    val = [0.0] * len(part)
    per_domain = []

    def flush(pending, dom):
        if not pending:
            return 0
        image = torch.cat([b[0] for b in pending], 0) if len(pending) > 1 else pending[0][0]
        label = torch.cat([b[1] for b in pending], 0) if len(pending) > 1 else pending[0][1]
        d = sample_dice(dataset, predict(dataset, model(image)), decode_labels(dataset, label))
        o = 0
        for b in pending:                                   # the batch's Dice = mean over ITS samples
            n = len(b[0])
            for p in range(len(part)):
                dom[p] += float(sum(d[o:o + n, p]) / n)
            o += n
        return len(pending)

    for i, loader in enumerate(loaders):
        dom, nb, pending, held = [0.0] * len(part), 0, [], 0
        for image, label in loader:
            image, label = image.to(dev), label.to(dev)
            if pending and (held + len(image) > coalesce or image.shape[1:] != pending[0][0].shape[1:]):
                nb += flush(pending, dom)
                pending, held = [], 0
            pending.append((image, label))
            held += len(image)
        nb += flush(pending, dom)
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
