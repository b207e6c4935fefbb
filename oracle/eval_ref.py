"""CPU restatement of the reference's validation loop (test infrastructure; never imported by the product).

Follows `test()` of train.py:253-395 (twins: train_mnms.py `test`, test.py:64-206): eval-mode forward per batch,
prediction (`sigmoid >= 0.5` per channel for fundus, first-index arg-max of the softmax otherwise), the batch's
per-part Dice from utils/metrics.py:149-231, averaged per domain loader and then over the domains.  The medpy
numbers printed beside it (dc / jc / hd95 / asd) are outside this build; the loss the reference
computes per batch is never accumulated there (`domain_val_loss` stays 0: train.py:267-334) and is not restated.
"""
import numpy as np
import torch

from . import step_ref as S
from . import unet_ref as U


def predict(dataset, logits):
    """train.py:292-299 / test.py:93-104."""
    if dataset == "fundus":
        return torch.sigmoid(logits).ge(0.5)
    return torch.max(torch.softmax(logits, dim=1), dim=1)[1]


def validate(dataset, sd, loaders):
    """loaders: one iterable of (image [N,C,H,W] f32, raw label) per domain -> (val_dice[parts], per_domain[d][parts])."""
    n_part = S.DATASETS[dataset][4]
    val = [0.0] * n_part
    per_domain = []
    for loader in loaders:
        dom = [0.0] * n_part
        nb = 0
        for image, label in loader:
            mask = S.decode_labels(dataset, label)
            with torch.no_grad():
                out = U.unet_forward(image, sd, train=False)
            dice = S.sample_dice(dataset, np.asarray(predict(dataset, out)), mask)
            for i in range(n_part):
                dom[i] += dice[i]
            nb += 1
        dom = [d / nb for d in dom]
        per_domain.append(dom)
        for i in range(n_part):
            val[i] += dom[i]
    return [v / len(loaders) for v in val], per_domain
