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
        self._side, self._side_busy = None, False        # side stream of the batch-1 low-quality-sample forward
        # where the step issues it: "split" = between the decoder and encoder halves of the backward (it then runs under the
        # encoder's many-block kernels: 29.34-29.42 ms per step against 29.51-29.53 for "start" = right after the student
        # passes, under the head / up4 weight gradients whose one-block-per-CU grids wait for the CUs it takes; its whole cost is
        # ~0.5 ms per step wherever it goes -- same-box runs of tools/ab_env.sh, profiles/r04_ab_side_forward.log)
        self._side_at = os.environ.get("USTRUN_SIDE_AT", "split")

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

    def _side_stream(self, dev):
        if self._side is None:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def _pl(self, logits):
        return F.pseudo_label(logits, self.threshold, self.mode)

    def _sample_dice_async(self, pred, gt):
        """Device overlap counts -> pinned host buffer (non-blocking); pair with an event before reading."""
        if self.dataset == "MNMS":
            cnt = F.dice_counts(pred, gt, by_class=True, n_classes=3)
        else:
            cnt = F.dice_counts(pred, gt)
        host = torch.empty(cnt.shape, dtype=cnt.dtype, pin_memory=True)
        host.copy_(cnt, non_blocking=True)
        return host, None

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

    def _freq_mix(self, mix_img, ulb_x_w, n):
        degree = self.iter_num / self.max_iterations
        ratios = [random.uniform(0, degree) for _ in range(n)]     # one draw per image (train.py:182)
        if self.fft == "host":
            src = ((mix_img[:n] + 1) * 127.5).cpu().numpy()
            trg = ((ulb_x_w[:n] + 1) * 127.5).cpu().numpy()
            out = [np.clip(freq_mix_host(src[i], trg[i], self.LB, ratios[i]), 0, 255).astype(np.float32) for i in range(n)]
            return (torch.tensor(np.array(out), dtype=torch.float32) / 127.5 - 1).to(mix_img.device)
        from . import fftmix
        return fftmix.freq_mix_device(mix_img[:n], ulb_x_w[:n], self.LB, ratios,
                                      h2d=lambda r: self._h2d(np.asarray(r, dtype=np.float32), mix_img.device, torch.float32))

    # ---------------------------------------------------------------------------------------
    def step(self, lb_x_w, lb_y, ulb_x_w, ulb_x_s, ulb_y, epoch_start=False):
        ds, mode, K = self.dataset, self.mode, self.n_classes
        model, ema = self.model, self.ema_model
        dev = lb_x_w.device
        B = len(ulb_x_s)
        epoch_num = self.iter_num // self.num_eval_iter
        self._mark("start")
        if epoch_start:
            self.lq_u = self.lq_pl = self.lq_mask = None
        lb_mask = decode_labels(ds, lb_y)
        ulb_mask = decode_labels(ds, ulb_y)
        mshape = [len(lb_x_w), K if ds == "fundus" else 1, self.patch, self.patch]

        # CutMix partner selection (train.py:612-627)
        ones = torch.ones(mshape, device=dev)
        if self.simple_ulb is None or len(self.simple_ulb) == 0:
            cut_img, cut_label, cut_mask = lb_x_w, lb_mask, ones
            choice = np.random.randint(0, len(lb_x_w), B)
        else:
            cut_img = torch.cat((lb_x_w, self.simple_ulb), 0)
            cut_label = torch.cat((lb_mask, self.cor_pl), 0)
            cut_mask = torch.cat((ones, self.cor_mask), 0)
            n_s = min(int(B * 0.5), len(self.simple_ulb))
            c_lb = np.random.randint(0, len(lb_x_w), B - n_s)
            c_s = np.random.randint(len(lb_x_w), len(lb_x_w) + len(self.simple_ulb), n_s)
            choice = np.random.permutation(np.concatenate((c_lb, c_s)))
        idx = self._h2d(np.asarray(choice), dev, torch.long)
        mix_img = cut_img.index_select(0, idx)
        cut_label_c, cut_mask_c = cut_label.index_select(0, idx), cut_mask.index_select(0, idx)

        # FFT low-frequency amplitude mix (train.py:628-636, Q13)
        move_transx = self._freq_mix(mix_img, ulb_x_w, len(lb_x_w))
        self._mark("select+freqmix")

        with torch.no_grad():
            box = F.rect_masks([cutmix_rect(self.patch, p=self.cutmix_prob) for _ in range(B)], self.patch, self.patch, dev)
            self._mark("boxes")
            # teacher: three train-mode forwards (train.py:638-667, Q11), batched into one call with BatchNorm per pass
            t_in = [ulb_x_w, F.box_mix(ulb_x_w, mix_img, box), F.box_mix(mix_img, ulb_x_w, box)]
            if self.batch_passes:
                t_out = ema.forward_passes(t_in).split(B)
            else:
                t_out = [ema(t) for t in t_in]
            pl, mask = self._pl(t_out[0])
            pl_w_ul, mask_w_ul = self._pl(t_out[1])
            pl_w_lu, mask_w_lu = self._pl(t_out[2])
            # student forward on the weak view: only its pseudo-label is used (Q3)
            stu_pl, _ = self._pl(model(ulb_x_w))
            pl_w, mask_w, pl_ul, mask_ul, pl_lu, mask_lu = F.mix_targets(
                mode, box, pl, mask, pl_w_ul, mask_w_ul, pl_w_lu, mask_w_lu, cut_label_c, cut_mask_c)
            x_s_ul = F.box_mix(ulb_x_s, move_transx, box)
            x_s_lu = F.box_mix(move_transx, ulb_x_s, box)

        # Everything the host decides from device data this step -- the per-sample Dice behind the hardness ranking and
        # the low-quality sample's region -- is computed and copied to pinned host memory NOW, in front of the student's
        # gradient passes in stream order: the host reads it while those passes run instead of draining the queue
        # after them (the GPU sat idle for the 1.5 ms of host work that followed).  The np.random draw keeps its place
        # in the stream of draws (nothing else consumes it in between).
        dice_host, dice_ev = self._sample_dice_async(stu_pl, pl)
        region_host = new_choice = None
        if self.lq_u is not None:
            new_choice = np.random.randint(0, len(lb_x_w))
            if ds == "fundus":
                region = self.lq_pl[0, 1].clone()
                region[self.lq_pl[0, 0].long() == 1] = 1
                region[lb_mask[new_choice, 0].long() == 1] = 1
                region[lb_mask[new_choice, 1].long() == 1] = 1
            else:
                region = self.lq_pl[0].clone()
                region[lb_mask[new_choice].long() > 0] = 1
            region_host = torch.empty(region.shape, dtype=region.dtype, pin_memory=True)
            region_host.copy_(region, non_blocking=True)
        host_ev = torch.cuda.Event()
        host_ev.record()

        self._mark("teacher+targets issued")
        # student: four forwards that carry gradient (train.py:699-702)
        lg_all = None
        if self.batch_passes and len(lb_x_w) == B:
            lg_all = model.forward_passes([lb_x_w, x_s_ul, x_s_lu, ulb_x_s])
            lg_lb, lg_ul, lg_lu, lg_s = lg_all.detach().split(B)
        else:
            lg_lb, lg_ul, lg_lu, lg_s = model(lb_x_w), model(x_s_ul), model(x_s_lu), model(ulb_x_s)
        self._mark("student fwd issued")

        # hardness and the low-quality sample forward (train.py:705-747, Q2, Q7)
        host_ev.synchronize()                     # the copies were queued in front of the student passes
        d = self._dice_from_host(dice_host)
        self._mark("dice on host")
        hardness = 1 - d.sum(0) / self.n_part
        if epoch_num == 0:
            hardness[:] = 1
        lq_idx = int(np.argmax(hardness))
        side_later = None
        if region_host is not None:
            ib_lq = F.rect_masks([all_cover_rect(region_host.numpy())], self.patch, self.patch, dev)
            # result unused (Q2); student BN running stats still move.  A batch-1 forward fills 16-256 workgroups per
            # launch, so it runs on a side stream underneath the losses and the backward that follow (ordered after
            # the student passes issued so far; joined before the optimizer update touches the parameters).
            lb_pick = lb_x_w[new_choice:new_choice + 1]
            lq_prev = self.lq_u

            def issue_side():
                side = self._side_stream(dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                for t_ in (lq_prev, lb_pick, ib_lq):
                    t_.record_stream(side)
                with torch.cuda.stream(side), torch.no_grad():
                    model(F.box_mix(lq_prev, lb_pick, ib_lq))
                self._side_busy = True
            if self._side_at in ("split", "mid") and lg_all is not None:
                side_later = issue_side
            elif self._side_at == "off_for_measurement_only":      # (its whole cost: never in a run whose statistics matter)
                pass
            else:
                issue_side()
        self.lq_u = ulb_x_w[lq_idx:lq_idx + 1].clone()
        self.lq_pl = pl[lq_idx:lq_idx + 1].clone()
        self.lq_mask = mask[lq_idx:lq_idx + 1].clone()

        # memory bank of easy unlabelled samples (train.py:749-782)
        simple = hardness < self.choice_th
        n_cur = int(simple.sum())
        sel = self._h2d(np.nonzero(simple)[0], dev, torch.long)        # indices from the host: no device-side nonzero, no sync

        def pick(t):
            return t.index_select(0, sel)
        if self.simple_ulb is None or len(self.simple_ulb) == 0:
            self.simple_ulb, self.cor_pl = pick(ulb_x_w), pick(pl)
            self.cor_gt, self.cor_mask = pick(ulb_mask), pick(mask)
            self.cor_hardness = hardness[simple].copy()
            if len(self.simple_ulb) > 0:
                self.choice_th = min(self.choice_th, self.cor_hardness.max())
        elif n_cur > 0:
            # train.py:768-771.  The reference's `newlen = max_len - cur_simple_num` goes NEGATIVE once a batch holds more
            # easy samples than the queue is long (possible only with unlabel_bs > queue_len = 10; the reference runs 4),
            # and `bank[:negative]` then lets the bank grow by up to unlabel_bs - queue_len entries per step without bound
            # (29 GB after 1000 steps at B = 16).  Clamped at 0: identical whenever unlabel_bs <= queue_len.
            keep = max(0, self.queue_len - n_cur) if len(self.simple_ulb) + n_cur > self.queue_len else len(self.simple_ulb)
            self.simple_ulb = torch.cat((pick(ulb_x_w), self.simple_ulb[:keep]), 0)
            self.cor_pl = torch.cat((pick(pl), self.cor_pl[:keep]), 0)
            self.cor_gt = torch.cat((pick(ulb_mask), self.cor_gt[:keep]), 0)
            self.cor_mask = torch.cat((pick(mask), self.cor_mask[:keep]), 0)
            self.cor_hardness = np.concatenate((hardness[simple], self.cor_hardness[:keep]))
            self.choice_th = min(self.choice_th, self.cor_hardness.max())
        else:
            self.choice_th = min(self.increase * self.choice_th, 0.1)

        self._mark("lq+bank")
        # losses and backward (train.py:816-848; Q5, Q6): loss = sup + w*(ul + lu + w*s)
        w = self.consistency * ramps.sigmoid_rampup(self.iter_num // (self.max_iterations / self.rampup), self.rampup)
        terms = ((lg_lb, lb_mask, None, 1.0), (lg_ul, pl_ul, mask_ul, w), (lg_lu, pl_lu, mask_lu, w), (lg_s, pl_w, mask_w, w * w))
        outs = []
        model._ustrun_sink_fresh = True
        dls = []
        for lg, tgt, msk, coef in terms:
            out = F.seg_loss_fwd(lg.detach(), tgt, msk, mode)
            outs.append(out)
            dl = F.seg_loss_bwd(lg.detach(), tgt, msk, mode, out, gscale=coef,
                                gdev=self.scaler.state if self.scaler is not None else None)     # scaler.scale(loss)
            if lg_all is None:
                lg.backward(dl)
            else:
                dls.append(dl)
        overlap = self.grad_allreduce is not None and hasattr(self.grad_allreduce, "start_tail")
        if lg_all is not None:                    # one backward over the four passes
            if overlap or side_later is not None:  # decoder gradients go out while the encoder half still runs
                side_mid = side_later is not None and self._side_at == "mid"

                def at_split():
                    if overlap:
                        self.grad_allreduce.start_tail(self.flat_g, self.dec_off)
                    if side_later is not None and not side_mid:
                        side_later()
                model._ustrun_backward_split_hook = at_split
                ar_mid = overlap and hasattr(self.grad_allreduce, "start_mid") and 0 < self.mid_off < self.dec_off
                if ar_mid or side_mid:
                    def at_mid():
                        if ar_mid:
                            self.grad_allreduce.start_mid(self.flat_g, self.mid_off)
                        if side_mid:
                            side_later()
                    model._ustrun_backward_mid_hook = at_mid
            try:
                lg_all.backward(torch.cat(dls, 0))
            finally:
                model._ustrun_backward_split_hook = None
                model._ustrun_backward_mid_hook = None
        self._mark("backward issued")
        if self.grad_allreduce is not None:
            if overlap:
                self.grad_allreduce.finish(self.flat_g)
            else:
                self.grad_allreduce(self.flat_g)

        # The low-quality-sample forward on the side stream reads BatchNorm gamma/beta, the ConvTranspose biases and the
        # head straight from flat_p (only the conv weights are packed copies): join it BEFORE the update rewrites flat_p.
        if self._side_busy:
            torch.cuda.current_stream(dev).wait_stream(self._side)
            self._side_busy = False
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
