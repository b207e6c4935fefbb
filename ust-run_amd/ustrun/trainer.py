"""One semi-supervised training iteration on MI355X -- the counterpart of the reference's inline
loop body train.py:577-858 (twin train_mnms.py:535-791), minus data loading and logging.

Every tensor op of the iteration runs in libustrun.so (U-Net forward/backward, pseudo-labels,
target mixing, CutMix compositing, losses, SGD+EMA); this module only sequences the calls, draws
the reference's random numbers in the reference's order, and keeps the reference's cross-iteration
state (memory bank, low-quality sample, choice threshold).  Quirks reproduced: Q2-Q7, Q10, Q11, Q13,
Q16, Q17.  Deliberate deviations: batch sizes are honoured (Q1); precision is the
model's compute dtype rather than fp16 autocast (Q12); the memory bank's truncation length is clamped at 0
(it goes negative in the reference once unlabel_bs exceeds queue_len and the bank then grows without bound).
"""
from __future__ import annotations

import math
import random
import time

import os

import numpy as np
import torch

from utils import metrics, ramps

from . import engine
from . import functional as F
from ._lib import BBOX_BLOCKS as L_BBOX

DATASETS = {
    # name: (in_channels, patch, classes, loss mode, parts, max_iterations)   train.py:404-436, train_mnms.py:397-404
    "fundus": (3, 256, 2, "sigmoid", 2, 30000),
    "prostate": (1, 384, 2, "softmax", 1, 60000),
    "BUSI": (1, 256, 2, "softmax", 1, 30000),
    "MNMS": (1, 288, 4, "softmax", 3, 60000),
}


def decode_labels(dataset, y):
    """train.py:590-608, train_mnms.py:549-556."""
    if dataset == "fundus":
        return torch.stack([y.eq(0).float(), y.le(128).float()], dim=1)
    if dataset == "prostate":
        return y.eq(0).long()
    if dataset == "BUSI":
        return y.eq(255).long()
    m = y[..., 0].eq(255).float()
    m[y[..., 1].eq(255)] = 2
    m[y[..., 2].eq(255)] = 3
    return m.long()


def cutmix_rect(img_size, p=0.5, size_min=0.02, size_max=0.4, ratio_1=0.3, ratio_2=1 / 0.3):
    """train.py:222-240 on the host, as the rectangle {y0,y1,x0,x1} of the map's ones (empty = all zeros); same RNG
    streams and draw order as the reference."""
    if random.random() > p:
        return (0, 0, 0, 0)
    size = np.random.uniform(size_min, size_max) * img_size * img_size
    while True:
        ratio = np.random.uniform(ratio_1, ratio_2)
        w = int(np.sqrt(size / ratio))
        h = int(np.sqrt(size * ratio))
        x = np.random.randint(0, img_size)
        y = np.random.randint(0, img_size)
        if x + w <= img_size and y + h <= img_size:
            break
    return (y, y + h, x, x + w)


def rect_map(rect, img_size):
    box = np.zeros((img_size, img_size), dtype=np.float32)
    box[rect[0]:rect[1], rect[2]:rect[3]] = 1
    return box


def cutmix_box(img_size, **kw):
    """The {0,1} map of train.py:222-240 (numpy)."""
    return rect_map(cutmix_rect(img_size, **kw), img_size)


def all_cover_rect(region):
    """train.py:242-251: bounding rectangle of the nonzero pixels (rows from scan order, cols min/max)."""
    loc = np.argwhere(region != 0)
    if len(loc) == 0:
        return cutmix_rect(region.shape[0], p=1.0)
    return (int(loc[0, 0]), int(loc[-1, 0]) + 1, int(loc[:, 1].min()), int(loc[:, 1].max()) + 1)


def all_cover_box(region):
    return rect_map(all_cover_rect(region), region.shape[0])


def freq_mix_host(src_img, trg_img, L, ratio):
    """train.py:158-207 (numpy FFT on the host, as the reference does)."""
    amp_trg = np.abs(np.fft.fft2(trg_img, axes=(-2, -1)))
    f = np.fft.fft2(src_img, axes=(-2, -1))
    amp, pha = np.abs(f), np.angle(f)
    a_s = np.fft.fftshift(amp, axes=(-2, -1))
    a_t = np.fft.fftshift(amp_trg, axes=(-2, -1))
    _, h, w = a_s.shape
    b = int(np.floor(min(h, w) * L))
    ch, cw = int(np.floor(h / 2.0)), int(np.floor(w / 2.0))
    sl = (slice(None), slice(ch - b, ch + b + 1), slice(cw - b, cw + b + 1))
    a_s[sl] = a_s[sl] * (1 - ratio) + a_t[sl] * ratio
    a_s = np.fft.ifftshift(a_s, axes=(-2, -1))
    return np.real(np.fft.ifft2(a_s * np.exp(1j * pha), axes=(-2, -1)))


def flatten_parameters(model):
    """Re-home all parameters in one contiguous f32 buffer (views keep their names/shapes)."""
    params = list(model.parameters())
    total = sum((p.numel() + 3) // 4 * 4 for p in params)
    flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)   # (the 4-element alignment gaps stay 0)
    views, o = [], 0
    for p in params:
        v = flat[o:o + p.numel()].view_as(p)
        v.copy_(p.data)
        p.data = v
        views.append(v)
        o += (p.numel() + 3) // 4 * 4
    return flat, views


def flat_like(flat, params):
    buf = torch.zeros_like(flat)
    views, o = [], 0
    for p in params:
        views.append(buf[o:o + p.numel()].view_as(p))
        o += (p.numel() + 3) // 4 * 4
    return buf, views


class SGDState:
    """The fused SGD's state behind torch.optim.SGD's `state_dict()` / `load_state_dict()` (the object the reference
    hands to util.save_osmancheckpoint / load_osmancheckpoint, train.py:512,544,957): per-parameter `momentum_buffer`
    views of the flat momentum buffer and one param group carrying the current learning rate, so checkpoints
    interchange with a `torch.optim.SGD(model.parameters(), lr, momentum=0.9, weight_decay=1e-4)`."""

    def __init__(self, trainer):
        self.t = trainer

    def _views(self):
        views, o = [], 0
        for p in self.t.model.parameters():
            views.append(self.t.flat_v[o:o + p.numel()].view_as(p))
            o += (p.numel() + 3) // 4 * 4
        return views

    def state_dict(self):
        views = self._views()
        state = {} if self.t.first_step else {i: {"momentum_buffer": v.clone()} for i, v in enumerate(views)}
        group = {"lr": self.t.lr, "momentum": self.t.momentum, "dampening": 0, "weight_decay": self.t.wd, "nesterov": False,
                 "maximize": False, "foreach": None, "differentiable": False, "fused": None, "params": list(range(len(views)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        views = self._views()
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(views):
            raise ValueError("optimizer state_dict: expected one param group over %d parameters" % len(views))
        g = groups[0]
        if g.get("nesterov") or g.get("dampening", 0) != 0:
            raise ValueError("optimizer state_dict: nesterov / dampening are not what the reference trains with")
        self.t.lr, self.t.momentum, self.t.wd = float(g["lr"]), float(g["momentum"]), float(g["weight_decay"])
        state = sd["state"]
        self.t.first_step = len(state) == 0              # torch semantics: the first step sets v = g
        self.t.flat_v.zero_()
        for i, v in enumerate(views):
            buf = state.get(g["params"][i], state.get(str(g["params"][i]), {})).get("momentum_buffer")
            if buf is not None:
                v.copy_(buf.to(v.device))


class SSLTrainer:
    def __init__(self, dataset, model, ema_model, base_lr=0.03, max_iterations=None, threshold=0.95,
                 ema_decay=0.99, consistency=1.0, consistency_rampup=200.0, cutmix_prob=1.0, LB=0.01,
                 increase=1.0005, queue_len=10, num_eval_iter=500, momentum=0.9, weight_decay=1e-4,
                 patch_size=None, grad_allreduce=None, world_size=1, fft="host", batch_passes=True):
        cfg = DATASETS[dataset]
        self.dataset = dataset
        self.n_classes, self.mode, self.n_part = cfg[2], cfg[3], cfg[4]
        self.patch = patch_size or cfg[1]
        self.max_iterations = max_iterations or cfg[5]
        self.model, self.ema_model = model, ema_model
        if getattr(model, "num_domains", 0) or getattr(ema_model, "num_domains", 0):
            # the reference's loop never builds such a network; the step's flat gradient / SGD / EMA buffers are
            # laid out over model.parameters() and the student's passes are batched across domains
            raise ValueError("SSLTrainer drives networks with plain BatchNorm2d; UNet(num_domains > 0) is for forward / backward through "
                             "autograd with a per-call domain_label")
        for p in ema_model.parameters():
            p.detach_()                                  # train.py:501-502
        model.train()
        ema_model.train()                                # teacher stays in train mode (train.py:566, Q11)
        self.flat_p, _ = flatten_parameters(model)
        self.flat_t, _ = flatten_parameters(ema_model)
        params = list(model.parameters())
        self.flat_g, self.grad_views = flat_like(self.flat_p, params)
        # offset of the first decoder parameter (up1.up.weight, index 30 of 64): the tail [dec_off:] is up1..up4 + outc
        self.dec_off = sum((p.numel() + 3) // 4 * 4 for p in params[:30]) if len(params) == 64 else 0
        # offset of down4's first parameter (index 24): [mid_off, dec_off) is final after the first encoder block's backward
        self.mid_off = sum((p.numel() + 3) // 4 * 4 for p in params[:24]) if len(params) == 64 else 0
        self.flat_v = torch.zeros_like(self.flat_p)
        # IEEE-half storage (the reference's autocast type) needs the reference's GradScaler around the backward
        # (train.py:552,842-845): scale on the device, skipped steps and scale updates without a host sync
        self.scaler = F.LossScale(self.flat_p.device) if getattr(model, "compute_dtype", "f32") == "f16" else None
        model._ustrun_grad_sink = self.grad_views        # backward accumulates straight into flat_g
        engine.invalidate_packed(model)
        engine.invalidate_packed(ema_model)
        self.base_lr, self.lr = base_lr, base_lr
        self.momentum, self.wd = momentum, weight_decay
        self.threshold, self.ema_decay = threshold, ema_decay
        self.consistency, self.rampup = consistency, consistency_rampup
        self.cutmix_prob, self.LB, self.increase, self.queue_len = cutmix_prob, LB, increase, queue_len
        self.num_eval_iter = num_eval_iter
        self.grad_allreduce = grad_allreduce             # callable(flat_g): SUM over ranks (RCCL all-reduce)
        self.world_size = world_size
        self.fft = fft
        # run the 3 teacher / 4 student passes as one batched call each (the U-Net engine; a DeepLabV2 runs them one by one)
        self.batch_passes = batch_passes and hasattr(model, "forward_passes")
        self.iter_num = 0
        self.first_step = True
        self.optimizer = SGDState(self)                  # what the reference's checkpoint helpers call `optimizer`
        # memory bank + low-quality sample state (train.py:554-561,576)
        self.simple_ulb = None
        self.cor_pl = self.cor_gt = self.cor_mask = None
        self.cor_hardness = []
        self.choice_th = 0.1
        self.lq_u = self.lq_pl = self.lq_mask = None
        self.last = {}
        self.timeline = None                             # list of (label, perf_counter) when host timing is on
        self._pin, self._ones = {}, None                 # pinned host buffers / a row of ones, kept across steps

    # ---------------------------------------------------------------------------------------
    def _mark(self, label):
        if self.timeline is not None:
            self.timeline.append((label, time.perf_counter()))

    @staticmethod
    def _h2d(arr, dev, dtype):
        """A few host values -> device inside a launch's arguments (ustrun_upload_small): ordered on the stream,
        no host wait.  (A pageable .to(dev) drains the queue; a pinned non-blocking copy on a busy stream measured
        20 ms/step slower on this ROCm.)"""
        return F.upload_small(arr, dev, dtype)

    def _pl(self, logits):
        return F.pseudo_label(logits, self.threshold, self.mode)

    def _sample_dice_async(self, pred, gt):
        """Device overlap counts -> pinned host buffer (non-blocking); pair with an event before reading."""
        if self.dataset == "MNMS":
            cnt = F.dice_counts(pred, gt, by_class=True, n_classes=3)
        else:
            cnt = F.dice_counts(pred, gt)
        host = self._pinned("dice", cnt.shape, cnt.dtype)
        host.copy_(cnt, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host, ev

    @staticmethod
    def _dice_from_host(host):
        c = host.numpy().astype(np.float64)               # [B, parts, 3]
        return metrics.dice_from_counts(c[..., 0], c[..., 1], c[..., 2]).T

    def _sample_dice(self, pred, gt):
        """Per-sample Dice (numpy array [n_part, B]) from device overlap counts: one small D2H copy."""
        if self.dataset == "MNMS":
            cnt = F.dice_counts(pred, gt, by_class=True, n_classes=3)
        else:
            cnt = F.dice_counts(pred, gt)
        c = cnt.cpu().numpy().astype(np.float64)          # [B, parts, 3]
        return metrics.dice_from_counts(c[..., 0], c[..., 1], c[..., 2]).T

    def _freq_mix(self, mix_img, ulb_x_w, n, ratios):
        """ratios: one draw per image, random.uniform(0, iter / max_iterations) (train.py:182), drawn by the caller."""
        if self.fft == "host":
            src = ((mix_img[:n] + 1) * 127.5).cpu().numpy()
            trg = ((ulb_x_w[:n] + 1) * 127.5).cpu().numpy()
            out = [np.clip(freq_mix_host(src[i], trg[i], self.LB, ratios[i]), 0, 255).astype(np.float32) for i in range(n)]
            return (torch.tensor(np.array(out), dtype=torch.float32) / 127.5 - 1).to(mix_img.device)
        from . import fftmix
        return fftmix.freq_mix_device(mix_img[:n], ulb_x_w[:n], self.LB, ratios,
                                      h2d=lambda r: self._h2d(np.asarray(r, dtype=np.float32), mix_img.device, torch.float32))

    # ---------------------------------------------------------------------------------------
    def _pinned(self, name, shape, dtype):
        """A pinned host buffer kept across steps (allocating one per step is a hipHostMalloc per step)."""
        buf = self._pin.get(name)
        if buf is None or tuple(buf.shape) != tuple(shape) or buf.dtype != dtype:
            buf = self._pin[name] = torch.empty(shape, dtype=dtype, pin_memory=True)
        return buf

    def _ones_row(self, shape, dev):
        """One row of ones of a mask's per-sample shape: the labelled samples' cut_mask rows all point at it (train.py:615,620)."""
        t = self._ones
        if t is None or tuple(t.shape[1:]) != tuple(shape) or t.device != dev:
            t = self._ones = torch.ones((1,) + tuple(shape), dtype=torch.float32, device=dev)
        return t

    def step(self, lb_x_w, lb_y, ulb_x_w, ulb_x_s, ulb_y, epoch_start=False):
        ds, mode, K = self.dataset, self.mode, self.n_classes
        model, ema = self.model, self.ema_model
        dev = lb_x_w.device
        B, nlb = len(ulb_x_s), len(lb_x_w)
        HW = self.patch * self.patch
        epoch_num = self.iter_num // self.num_eval_iter
        self._mark("start")
        if epoch_start:
            self.lq_u = self.lq_pl = self.lq_mask = None
        lb_x_w, ulb_x_w, ulb_x_s = (F._f32c(t, "images") for t in (lb_x_w, ulb_x_w, ulb_x_s))
        lb_mask = F.decode_labels(ds, lb_y)
        ulb_mask = F.decode_labels(ds, ulb_y)

        # ---- every host random number of the iteration, in the reference's order; none of them depends on device data:
        # CutMix partners (train.py:612-625), one FFT ratio per image (:182 via :631), the B CutMix boxes (:639), the labelled
        # partner of the low-quality sample (:721).  (The cover box's fallback draw, :242-251, follows once its region is known;
        # nothing draws in between.)
        bank = self.simple_ulb if self.simple_ulb is not None and len(self.simple_ulb) > 0 else None
        if bank is None:
            choice = np.random.randint(0, nlb, B)
        else:
            n_s = min(int(B * 0.5), len(bank))
            c_lb = np.random.randint(0, nlb, B - n_s)
            c_s = np.random.randint(nlb, nlb + len(bank), n_s)
            choice = np.random.permutation(np.concatenate((c_lb, c_s)))
        degree = self.iter_num / self.max_iterations
        ratios = [random.uniform(0, degree) for _ in range(nlb)]
        rects = [cutmix_rect(self.patch, p=self.cutmix_prob) for _ in range(B)]
        new_choice = np.random.randint(0, nlb) if self.lq_u is not None else None

        # ---- the low-quality sample's region (train.py:722-729) is last step's pseudo-label united with this step's picked label:
        # its bounding rectangle is asked for NOW (one small kernel, 1 KB to pinned memory) and read when the student's inputs are
        # put together -- with the teacher's passes queued in between, the host never waits for it with the GPU idle
        bbox_ev = None
        if new_choice is not None:
            if ds == "fundus":
                planes = [self.lq_pl[0, 1], self.lq_pl[0, 0], lb_mask[new_choice, 0], lb_mask[new_choice, 1]]
            else:
                planes = [self.lq_pl[0], lb_mask[new_choice]]
            part = F.region_bbox_partials(planes, self.patch, self.patch,
                                          torch.empty((L_BBOX, 4), dtype=torch.int32, device=dev))
            bbox_host = self._pinned("bbox", part.shape, part.dtype)
            bbox_host.copy_(part, non_blocking=True)
            bbox_ev = torch.cuda.Event()
            bbox_ev.record()

        # ---- teacher inputs and the CutMix partners, gathered by address (train.py:627,643-647): rows of lb_x_w / the memory bank
        def rows_of(first, more):
            return F.row_ptrs(first) + (F.row_ptrs(more) if more is not None else [])
        box = F.rect_masks(rects, self.patch, self.patch, dev)
        bx, uw = F.row_ptrs(box), F.row_ptrs(ulb_x_w)
        img_rows = rows_of(lb_x_w, bank)
        cut = [img_rows[c] for c in choice]
        t_buf = F.assemble([(u, 0, 0) for u in uw] + [(uw[i], cut[i], bx[i]) for i in range(B)] +
                           [(cut[i], uw[i], bx[i]) for i in range(B)] + [(c, 0, 0) for c in cut], ulb_x_w, HW)
        mix_img = t_buf[3 * B:]
        lab_rows = rows_of(lb_mask, self.cor_pl if bank is not None else None)
        cut_label_c = F.assemble([(lab_rows[c], 0, 0) for c in choice], lb_mask)
        mshape = [K if ds == "fundus" else 1, self.patch, self.patch]
        ones = self._ones_row(mshape, dev)
        msk_rows = [ones.data_ptr()] * nlb + (F.row_ptrs(self.cor_mask) if bank is not None else [])
        cut_mask_c = F.assemble([(msk_rows[c], 0, 0) for c in choice], ones)

        # FFT low-frequency amplitude mix (train.py:628-636, Q13)
        move_transx = self._freq_mix(mix_img, ulb_x_w, nlb, ratios)
        self._mark("select+freqmix")

        with torch.no_grad():
            # teacher: three train-mode forwards (train.py:638-667, Q11), batched into one call with BatchNorm per pass
            if self.batch_passes:
                t_out = ema.forward_batched(t_buf[:3 * B], 3).split(B)
            else:
                t_out = [ema(t) for t in t_buf[:3 * B].split(B)]
            pl, mask = self._pl(t_out[0])
            pl_w_ul, mask_w_ul = self._pl(t_out[1])
            pl_w_lu, mask_w_lu = self._pl(t_out[2])
            pl_w, mask_w, pl_ul, mask_ul, pl_lu, mask_lu = F.mix_targets(
                mode, box, pl, mask, pl_w_ul, mask_w_ul, pl_w_lu, mask_w_lu, cut_label_c, cut_mask_c)
        # the student's forward on the weak view (train.py:668: only its arg-max is used, Q3 -- but its BatchNorm updates come
        # first): as the LEADING, gradient-free pass of the student's batched call below, or as a call of its own
        fold_weak = self.batch_passes and nlb == B and B > 1
        stu_pl = None
        if not fold_weak:
            with torch.no_grad():
                stu_pl, _ = self._pl(model(ulb_x_w))
        self._mark("teacher+targets issued")

        # ---- student: four forwards that carry gradient (train.py:699-702) and, behind them, the low-quality sample's
        # (train.py:734-740: result unused, Q2 -- only the BatchNorm running statistics move, after the other four's): ONE batch,
        # put together in one launch
        us, mt, lbr = F.row_ptrs(ulb_x_s), F.row_ptrs(move_transx), F.row_ptrs(lb_x_w)
        rows = ([(p, 0, 0) for p in uw] if fold_weak else []) + [(p, 0, 0) for p in lbr] + \
               [(us[i], mt[i], bx[i]) for i in range(B)] + [(mt[i], us[i], bx[i]) for i in range(B)] + [(p, 0, 0) for p in us]
        ib_lq = None
        if new_choice is not None:
            bbox_ev.synchronize()
            rect = F.fold_bbox(bbox_host.numpy())
            if rect is None:                                    # empty region: a random box (train.py:245-246)
                rect = cutmix_rect(self.patch, p=1.0)
            ib_lq = F.rect_masks([rect], self.patch, self.patch, dev)
            rows.append((self.lq_u.data_ptr(), lbr[new_choice], ib_lq.data_ptr()))
        x_all = F.assemble(rows, lb_x_w, HW)
        n4 = nlb + 3 * B
        lg_all = None
        if fold_weak:
            lg_all = model.forward_batched(x_all, 5, tail=len(x_all) - n4 - B, lead=1)
            lg_weak, lg_lb, lg_ul, lg_lu, lg_s = lg_all.detach().split(B)
            stu_pl, _ = self._pl(lg_weak)
        else:
            lg_lb, lg_ul, lg_lu, lg_s = (model(t) for t in x_all[:n4].split([nlb, B, B, B]))
            if ib_lq is not None:
                with torch.no_grad():
                    model(x_all[n4:])
        # the per-sample Dice behind the hardness ranking: counted and copied to pinned memory here, read at the END of the step
        # (nothing before the next step depends on it)
        dice_host, dice_ev = self._sample_dice_async(stu_pl, pl)
        self._mark("student fwd issued")

        # losses and backward (train.py:816-848; Q5, Q6): loss = sup + w*(ul + lu + w*s)
        w = self.consistency * ramps.sigmoid_rampup(self.iter_num // (self.max_iterations / self.rampup), self.rampup)
        terms = ((lg_lb, lb_mask, None, 1.0), (lg_ul, pl_ul, mask_ul, w), (lg_lu, pl_lu, mask_lu, w), (lg_s, pl_w, mask_w, w * w))
        outs = []
        model._ustrun_sink_fresh = True
        dl_all = torch.empty_like(lg_all) if lg_all is not None else None
        for k, (lg, tgt, msk, coef) in enumerate(terms):
            out = F.seg_loss_fwd(lg.detach(), tgt, msk, mode)
            outs.append(out)
            dl = F.seg_loss_bwd(lg.detach(), tgt, msk, mode, out, gscale=coef,
                                gdev=self.scaler.state if self.scaler is not None else None,       # scaler.scale(loss)
                                out=dl_all[(k + 1) * B:(k + 2) * B] if dl_all is not None else None)     # (rows of the leading pass: ignored)
            if lg_all is None:
                lg.backward(dl)
        overlap = self.grad_allreduce is not None and hasattr(self.grad_allreduce, "start_tail")
        if lg_all is not None:                    # one backward over the four passes
            if overlap:                           # decoder gradients go out while the encoder half still runs
                model._ustrun_backward_split_hook = lambda: self.grad_allreduce.start_tail(self.flat_g, self.dec_off)
                if hasattr(self.grad_allreduce, "start_mid") and 0 < self.mid_off < self.dec_off:
                    model._ustrun_backward_mid_hook = lambda: self.grad_allreduce.start_mid(self.flat_g, self.mid_off)
            try:
                lg_all.backward(dl_all)
            finally:
                model._ustrun_backward_split_hook = None
                model._ustrun_backward_mid_hook = None
        self._mark("backward issued")
        if self.grad_allreduce is not None:
            if overlap:
                self.grad_allreduce.finish(self.flat_g)
            else:
                self.grad_allreduce(self.flat_g)

        # SGD + EMA (train.py:848-851; alpha from the pre-increment iter_num, Q10), poly LR for the NEXT step
        alpha = min(1 - 1 / (self.iter_num + 1), self.ema_decay)
        if self.scaler is not None:               # scaler.step(optimizer); scaler.update() -- and the EMA line, in one pass
            self.scaler.step(self.flat_p, self.flat_g, self.flat_v, self.flat_t, self.lr, self.momentum, self.wd,
                             self.first_step, alpha, grad_scale=1.0 / self.world_size)
        else:
            F.sgd_ema(self.flat_p, self.flat_g, self.flat_v, self.flat_t, self.lr, self.momentum, self.wd,
                      self.first_step, alpha, grad_scale=1.0 / self.world_size)
        self.first_step = False
        engine.invalidate_packed(model)
        engine.invalidate_packed(ema)
        self.lr = self.base_lr * (1.0 - self.iter_num / self.max_iterations) ** 0.9
        self.iter_num += 1
        self._mark("update issued")

        # ---- what the host keeps for the next iteration (train.py:705-718,749-782): hardness ranking, the low-quality sample,
        # the memory bank.  The Dice counts were copied in front of the student's passes; the GPU is far past that point
        dice_ev.synchronize()
        d = self._dice_from_host(dice_host)
        hardness = 1 - d.sum(0) / self.n_part
        if epoch_num == 0:
            hardness[:] = 1                                   # Q7
        lq_idx = int(np.argmax(hardness))
        self.lq_u = ulb_x_w[lq_idx:lq_idx + 1].clone()        # (the caller may reuse its input buffers)
        self.lq_pl = pl[lq_idx:lq_idx + 1]                    # (pl / mask are this step's own outputs: nobody writes them again)
        self.lq_mask = mask[lq_idx:lq_idx + 1]

        # memory bank of easy unlabelled samples (train.py:749-782)
        simple = hardness < self.choice_th
        n_cur = int(simple.sum())
        sel = [int(i) for i in np.nonzero(simple)[0]]

        def merged(cur, old, keep):                # rows `sel` of cur, then the first `keep` rows of the old bank: one launch
            rows = [F.row_ptrs(cur)[i] for i in sel] + (F.row_ptrs(old)[:keep] if old is not None else [])
            if not rows:
                return cur[:0]
            return F.assemble([(r, 0, 0) for r in rows], cur)
        if bank is None:
            self.simple_ulb, self.cor_pl = merged(ulb_x_w, None, 0), merged(pl, None, 0)
            self.cor_gt, self.cor_mask = merged(ulb_mask, None, 0), merged(mask, None, 0)
            self.cor_hardness = hardness[simple].copy()
            if len(self.simple_ulb) > 0:
                self.choice_th = min(self.choice_th, self.cor_hardness.max())
        elif n_cur > 0:
            # train.py:768-771.  The reference's `newlen = max_len - cur_simple_num` goes NEGATIVE once a batch holds more
            # easy samples than the queue is long (possible only with unlabel_bs > queue_len = 10; the reference runs 4),
            # and `bank[:negative]` then lets the bank grow by up to unlabel_bs - queue_len entries per step without bound
            # (29 GB after 1000 steps at B = 16).  Clamped at 0: identical whenever unlabel_bs <= queue_len.
            keep = max(0, self.queue_len - n_cur) if len(bank) + n_cur > self.queue_len else len(bank)
            keep = min(keep, len(bank))
            self.simple_ulb, self.cor_pl = merged(ulb_x_w, self.simple_ulb, keep), merged(pl, self.cor_pl, keep)
            self.cor_gt, self.cor_mask = merged(ulb_mask, self.cor_gt, keep), merged(mask, self.cor_mask, keep)
            self.cor_hardness = np.concatenate((hardness[simple], self.cor_hardness[:keep]))
            self.choice_th = min(self.choice_th, self.cor_hardness.max())
        else:
            self.choice_th = min(self.increase * self.choice_th, 0.1)

        self.last = {"outs": outs, "w": w, "pl": pl, "ulb_mask": ulb_mask, "mask": mask}
        self._mark("end")
        return self.last

    def scalars(self):
        """Loss values of the last step as python floats (one D2H copy; logging only)."""
        o = torch.stack([t[:2] for t in self.last["outs"]]).cpu().numpy().astype(np.float64)
        w = self.last["w"]
        sup, ul, lu, s = (float(o[i, 0] + o[i, 1]) for i in range(4))
        d = self._sample_dice(self.last["pl"], self.last["ulb_mask"]).mean(1)
        return {"loss": sup + w * (ul + lu + w * s), "sup": sup, "ul": ul, "lu": lu, "s": s, "w": w,
                "ulb_dice": [float(v) for v in d], "mask_ratio": float(self.last["mask"].mean())}
