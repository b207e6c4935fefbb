"""Host-side scalars and numpy helpers of the step (test infrastructure)."""
from __future__ import annotations

import math
import random

import numpy as np


def sigmoid_rampup(current, rampup_length):
    """exp(-5 (1 - clip(cur,0,T)/T)^2); 1.0 when T == 0.  reference: utils/ramps.py:19-26."""
    if rampup_length == 0:
        return 1.0
    cur = min(max(float(current), 0.0), float(rampup_length))
    phase = 1.0 - cur / rampup_length
    return float(math.exp(-5.0 * phase * phase))


def consistency_weight(iter_num, max_iterations, consistency=1.0, rampup=200.0):
    """reference: train.py:82-84 with the argument built at train.py:819-820."""
    return consistency * sigmoid_rampup(iter_num // (max_iterations / rampup), rampup)


def poly_lr(base_lr, iter_num, max_iterations):
    """reference: train.py:854 (evaluated with the PRE-increment iter_num, Q10)."""
    return base_lr * (1.0 - iter_num / max_iterations) ** 0.9


def ema_alpha(global_step, ema_decay):
    """reference: train.py:91."""
    return min(1 - 1 / (global_step + 1), ema_decay)


def dice_binary(seg, gt):
    """(2I+1)/(1.001+S+G); 0.0 when both masks are empty.  reference: utils/metrics.py:114-146."""
    seg = np.asarray(seg, dtype=bool)
    gt = np.asarray(gt, dtype=bool)
    s, g = float(seg.sum()), float(gt.sum())
    if s == 0 and g == 0:
        return 0.0
    return (2.0 * float(np.logical_and(seg, gt).sum()) + 1.0) / (1.001 + s + g)


def dice_coeff(pred, target, ret_arr=False):
    """reference: utils/metrics.py:149-175."""
    target = np.asarray(target)
    if pred.ndim == 2:
        return dice_binary(pred, target)
    vals = [dice_binary(pred[i], target[i]) for i in range(pred.shape[0])]
    return [np.array(vals)] if ret_arr else [sum(vals) / len(vals)]


def dice_coeff_2label(pred, target, ret_arr=False):
    """reference: utils/metrics.py:177-201 (channel 0 = cup, 1 = disc)."""
    target = np.asarray(target)
    if pred.ndim == 3:
        return dice_binary(pred[0], target[0]), dice_binary(pred[1], target[1])
    cup = [dice_binary(pred[i, 0], target[i, 0]) for i in range(pred.shape[0])]
    disc = [dice_binary(pred[i, 1], target[i, 1]) for i in range(pred.shape[0])]
    if ret_arr:
        return [np.array(cup), np.array(disc)]
    return [sum(cup) / len(cup), sum(disc) / len(disc)]


def dice_coeff_3label(pred, target, ret_arr=False):
    """reference: utils/metrics.py:203-231 (classes 1,2,3 = lv,myo,rv)."""
    target = np.asarray(target)
    if pred.ndim == 2:
        return tuple(dice_binary(pred == c, target == c) for c in (1, 2, 3))
    cols = [[dice_binary(pred[i] == c, target[i] == c) for i in range(pred.shape[0])] for c in (1, 2, 3)]
    if ret_arr:
        return [np.array(c) for c in cols]
    return [sum(c) / len(c) for c in cols]


def cutmix_box(img_size, p=0.5, size_min=0.02, size_max=0.4, ratio_1=0.3, ratio_2=1 / 0.3):
    """Random CutMix rectangle as a numpy {0,1} float32 map.  reference: train.py:222-240.

    RNG streams are the reference's: one ``random.random()`` draw for the skip test, then
    ``np.random`` for area, aspect and the corner, re-drawn until the box fits.
    """
    box = np.zeros((img_size, img_size), dtype=np.float32)
    if random.random() > p:
        return box
    size = np.random.uniform(size_min, size_max) * img_size * img_size
    while True:
        ratio = np.random.uniform(ratio_1, ratio_2)
        w = int(np.sqrt(size / ratio))
        h = int(np.sqrt(size * ratio))
        x = np.random.randint(0, img_size)
        y = np.random.randint(0, img_size)
        if x + w <= img_size and y + h <= img_size:
            break
    box[y:y + h, x:x + w] = 1
    return box


def all_cover_box(region):
    """Bounding box of the nonzero pixels of ``region`` ([H,W] numpy).  reference: train.py:242-251.

    Rows come from the first/last nonzero in scan order, columns from min/max.
    """
    loc = np.argwhere(region != 0)
    if len(loc) == 0:
        return cutmix_box(region.shape[0], p=1.0)
    box = np.zeros(region.shape, dtype=np.float32)
    y1, y2 = loc[0, 0], loc[-1, 0]
    x1, x2 = loc[:, 1].min(), loc[:, 1].max()
    box[y1:y2 + 1, x1:x2 + 1] = 1
    return box


def amp_spectrum(img):
    """|fft2| over the last two axes.  reference: train.py:158-165."""
    return np.abs(np.fft.fft2(img, axes=(-2, -1)))


def freq_mix(src_img, amp_trg, L=0.1, degree=1.0, ratio=None):
    """Swap the low-frequency amplitude window of src with trg's.  reference: train.py:167-207.

    Window half-width b = floor(min(h,w)*L) around the fftshift centre; blend ratio is
    ``random.uniform(0, degree)`` (drawn here unless given); phase of src is kept.
    """
    f = np.fft.fft2(src_img, axes=(-2, -1))
    amp, pha = np.abs(f), np.angle(f)
    a_s = np.fft.fftshift(amp, axes=(-2, -1))
    a_t = np.fft.fftshift(amp_trg, axes=(-2, -1))
    _, h, w = a_s.shape
    b = int(np.floor(min(h, w) * L))
    ch, cw = int(np.floor(h / 2.0)), int(np.floor(w / 2.0))
    if ratio is None:
        ratio = random.uniform(0, degree)
    sl = (slice(None), slice(ch - b, ch + b + 1), slice(cw - b, cw + b + 1))
    a_s[sl] = a_s[sl] * (1 - ratio) + a_t[sl] * ratio
    a_s = np.fft.ifftshift(a_s, axes=(-2, -1))
    return np.real(np.fft.ifft2(a_s * np.exp(1j * pha), axes=(-2, -1)))
