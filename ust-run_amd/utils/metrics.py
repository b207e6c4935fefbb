"""Binary Dice metrics (host side) -- same surface as the reference's utils/metrics.py:114-231.

`dice_coefficient_numpy` is (2I+1)/(1.001+S+G), 0.0 when both masks are empty.  The per-sample
overlap counts can also come from the device (ustrun.functional.dice_counts): `dice_from_counts`
applies the same formula to them, so the training step needs one small D2H copy instead of moving
whole masks to the host.
"""
import numpy as np


def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


def dice_from_counts(s, g, i):
    """Dice from |pred|, |gt|, |pred & gt| (scalars or arrays)."""
    s, g, i = np.asarray(s, dtype=np.float64), np.asarray(g, dtype=np.float64), np.asarray(i, dtype=np.float64)
    d = (2.0 * i + 1.0) / (1.001 + s + g)
    return np.where((s == 0) & (g == 0), 0.0, d)


def dice_coefficient_numpy(binary_segmentation, binary_gt_label):
    seg = _np(binary_segmentation).astype(bool)
    gt = _np(binary_gt_label).astype(bool)
    return float(dice_from_counts(seg.sum(), gt.sum(), np.logical_and(seg, gt).sum()))


def dice_coeff(pred, target, ret_arr=False):
    pred, target = _np(pred), _np(target)
    if pred.ndim == 2:
        return dice_coefficient_numpy(pred, target)
    vals = [dice_coefficient_numpy(pred[i], target[i]) for i in range(pred.shape[0])]
    if ret_arr:
        return [np.array(vals)]
    return [sum(vals) / len(vals)]


def dice_coeff_2label(pred, target, ret_arr=False):
    pred, target = _np(pred), _np(target)
    if pred.ndim == 3:
        return dice_coefficient_numpy(pred[0], target[0]), dice_coefficient_numpy(pred[1], target[1])
    cup = [dice_coefficient_numpy(pred[i, 0], target[i, 0]) for i in range(pred.shape[0])]
    disc = [dice_coefficient_numpy(pred[i, 1], target[i, 1]) for i in range(pred.shape[0])]
    if ret_arr:
        return [np.array(cup), np.array(disc)]
    return [sum(cup) / len(cup), sum(disc) / len(disc)]


def dice_coeff_3label(pred, target, ret_arr=False, multi_layer=False):
    pred, target = _np(pred), _np(target)
    if pred.ndim == 2:
        return tuple(dice_coefficient_numpy(pred == c, target == c) for c in (1, 2, 3))
    cols = [[dice_coefficient_numpy(pred[i] == c, target[i] == c) for i in range(pred.shape[0])] for c in (1, 2, 3)]
    if ret_arr:
        return [np.array(c) for c in cols]
    return [sum(c) / len(c) for c in cols]
