"""CPU restatement of the loss terms and pseudo-label algebra (test infrastructure).

All functions take/return torch CPU tensors and are differentiable through torch
autograd where the reference's are.
"""
from __future__ import annotations

import torch

SMOOTH = 1e-10


def dice_loss_with_mask(inputs, target, n_classes, mask=None, weight=None,
                        softmax=False, sigmoid=False, multi=False):
    """reference: utils/losses.py:236-268 (DiceLossWithMask.forward) and helpers :199-234.

    * sigmoid: p = sigmoid(inputs), target loses its dim-1 singleton (:240-241)
    * softmax: p = softmax(inputs, 1) (:242-243)
    * multi: ONE global dice over everything, masked when mask is given (:244-249)
    * otherwise per-class dice on one-hot target, averaged over classes; the mask's
      class-0 channel is all ones because ``mask*0 == 0`` (Q4, :207-213).
    dice = 1 - (2*sum(p*t*m) + 1e-10) / (sum(p*p*m) + sum(t*t*m) + 1e-10).
    """
    assert not (sigmoid and softmax)
    if sigmoid:
        p = torch.sigmoid(inputs)
        target = target.squeeze(1)
    elif softmax:
        p = torch.softmax(inputs, dim=1)
    else:
        p = inputs

    def dice(score, tgt, m=None):
        tgt = tgt.float()
        if m is None:
            inter = (score * tgt).sum()
            ysum = (tgt * tgt).sum()
            zsum = (score * score).sum()
        else:
            m = m.float()
            inter = (score * tgt * m).sum()
            ysum = (tgt * tgt * m).sum()
            zsum = (score * score * m).sum()
        return 1 - (2 * inter + SMOOTH) / (zsum + ysum + SMOOTH)

    if multi:
        return dice(p, target, mask)
    onehot = torch.cat([(target == i).float() for i in range(n_classes)], dim=1)
    assert p.shape == onehot.shape, "predict & target shape do not match"
    w = [1] * n_classes if weight is None else weight
    total = 0.0
    for i in range(n_classes):
        if mask is not None:
            mi = torch.ones_like(mask[:, 0]) if i == 0 else (mask[:, 0] == 1).float()
            total = total + dice(p[:, i], onehot[:, i], mi) * w[i]
        else:
            total = total + dice(p[:, i], onehot[:, i]) * w[i]
    return total / n_classes


def ce_none(logits, target):
    """CrossEntropyLoss(reduction='none').  reference call site: train.py:519."""
    logp = torch.log_softmax(logits, dim=1)
    return -logp.gather(1, target[:, None].long()).squeeze(1)


def bce_logits_none(logits, target):
    """BCEWithLogitsLoss(reduction='none').  reference call site: train.py:516.

    Stable form max(x,0) - x*t + log1p(exp(-|x|)) (what ATen evaluates).
    """
    return torch.clamp(logits, min=0) - logits * target + torch.log1p(torch.exp(-logits.abs()))


def seg_loss(logits, target, mask, mode, n_classes):
    """One `ce + dice` term of the step.  reference: train.py:816-817,829-836.

    mode 'softmax' (prostate/BUSI/MNMS): target [B,H,W] int64, mask [B,1,H,W] or None.
    mode 'sigmoid' (fundus, multi-label): target/mask [B,K,H,W] float.
    Masked CE is (ce*mask).mean() -- divides by ALL elements (Q5).
    """
    if mode == "softmax":
        ce = ce_none(logits, target)
        ce = ce.mean() if mask is None else (ce * mask.squeeze(1)).mean()
        d = dice_loss_with_mask(logits, target[:, None], n_classes, mask=mask, softmax=True)
    else:
        ce = bce_logits_none(logits, target)
        ce = ce.mean() if mask is None else (ce * mask).mean()
        d = dice_loss_with_mask(logits, target[:, None], n_classes, mask=mask, sigmoid=True, multi=True)
    return ce, d


def pseudo_label(logits, threshold, mode):
    """Teacher pseudo-labels and confidence masks.  reference: train.py:648-667.

    softmax: prob,label = max(softmax(logits,1),1); mask = (prob > th)[:,None].float()
    sigmoid: label = (p >= .5).float(); mask = (p >= th).float() + (p <= 1-th).float()
    """
    if mode == "softmax":
        prob = torch.softmax(logits, dim=1)
        conf, label = torch.max(prob, dim=1)
        return label, (conf > threshold).unsqueeze(1).float()
    p = torch.sigmoid(logits)
    return (p >= 0.5).float(), (p >= threshold).float() + (p <= 1 - threshold).float()


def mix_targets(mode, pl, mask, pl_w_ul, mask_w_ul, pl_w_lu, mask_w_lu, box, cut_label, cut_mask):
    """Ensemble mask and CutMix'd targets.  reference: train.py:677-697.

    ``box`` is [B,H,W] in {0,1}.  Returns (pl_w, mask_w, pl_ul, mask_ul, pl_lu, mask_lu).
    """
    img_box = box[:, None]
    label_box = img_box if mode == "sigmoid" else box
    mask_w = mask_w_ul * (1 - img_box) + mask_w_lu * img_box
    pl_w = (pl_w_ul * (1 - label_box) + pl_w_lu * label_box).long()
    if mode == "sigmoid":
        pl_w = pl_w.float()
        ens = (pl_w == pl).float() * mask
    else:
        ens = (pl_w == pl).unsqueeze(1).float() * mask
    mask_w = torch.where(ens == 0, torch.zeros_like(mask_w), mask_w)
    sel = img_box.expand(mask.shape) == 1
    pl_ul = (pl * (1 - label_box) + cut_label * label_box).long()
    mask_ul = torch.where(sel, cut_mask, mask)
    pl_lu = (cut_label * (1 - label_box) + pl * label_box).long()
    mask_lu = torch.where(sel, mask, cut_mask)
    if mode == "sigmoid":
        pl_ul, pl_lu = pl_ul.float(), pl_lu.float()
    return pl_w, mask_w, pl_ul, mask_ul, pl_lu, mask_lu
