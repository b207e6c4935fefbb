"""CPU restatement of one semi-supervised training iteration (test infrastructure).

Follows the reference's inline loop body, train.py:577-858 (twin: train_mnms.py:535-791),
minus data loading and logging, on batches handed in as tensors.  Quirks Q2-Q7, Q10, Q11,
Q13, Q16, Q17 of SURVEY.md 7 are reproduced; Q1 (forced batch 4) and Q12 (fp16 autocast,
a no-op on CPU) are not.
"""
from __future__ import annotations

import numpy as np
import torch

from . import host_ref as H
from . import losses_ref as L
from . import unet_ref as U

DATASETS = {
    # name: (in_channels, patch, classes, loss mode, parts, max_iterations)   train.py:404-436, train_mnms.py:397-404
    "fundus": (3, 256, 2, "sigmoid", 2, 30000),
    "prostate": (1, 384, 2, "softmax", 1, 60000),
    "BUSI": (1, 256, 2, "softmax", 1, 30000),
    "MNMS": (1, 288, 4, "softmax", 3, 60000),
}


def decode_labels(dataset, y):
    """Raw label tensor -> training target.  reference: train.py:590-608, train_mnms.py:549-556."""
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


def sample_dice(dataset, pred, target, ret_arr=False):
    fn = {"fundus": H.dice_coeff_2label, "MNMS": H.dice_coeff_3label}.get(dataset, H.dice_coeff)
    return fn(np.asarray(pred), np.asarray(target), ret_arr=ret_arr)


class RefTrainer:
    """State that the reference keeps in locals of train() across iterations."""

    def __init__(self, dataset, sd, base_lr=0.03, max_iterations=None, threshold=0.95,
                 ema_decay=0.99, consistency=1.0, consistency_rampup=200.0, cutmix_prob=1.0,
                 LB=0.01, increase=1.0005, queue_len=10, num_eval_iter=500, momentum=0.9,
                 weight_decay=1e-4, patch_size=None, forward=None):
        cfg = DATASETS[dataset]
        self.forward = forward or U.unet_forward      # (x, sd, train=...) -> logits; oracle/deeplab_ref for the DeepLabV2 row
        self.dataset = dataset
        self.n_classes, self.mode, self.n_part = cfg[2], cfg[3], cfg[4]
        self.patch = patch_size or cfg[1]
        self.max_iterations = max_iterations or cfg[5]
        self.student = U.clone_sd(sd, requires_grad=True)
        self.teacher = U.clone_sd(sd)            # create_model(ema=True): separate init in the
        self.pkeys = U.param_keys(sd)            # reference; callers pass the sd they want
        self.mom = {k: None for k in self.pkeys}
        self.base_lr, self.lr = base_lr, base_lr
        self.momentum, self.wd = momentum, weight_decay
        self.threshold, self.ema_decay = threshold, ema_decay
        self.consistency, self.rampup = consistency, consistency_rampup
        self.cutmix_prob, self.LB, self.increase, self.queue_len = cutmix_prob, LB, increase, queue_len
        self.num_eval_iter = num_eval_iter
        self.iter_num = 0
        # memory bank + low-quality sample state (train.py:554-561,576)
        self.simple_ulb = None
        self.cor_pl = self.cor_gt = self.cor_mask = None
        self.cor_hardness = []
        self.choice_th = 0.1
        self.lq_u = self.lq_pl = self.lq_mask = None

    def set_teacher(self, sd):
        self.teacher = U.clone_sd(sd)

    # -- pieces -------------------------------------------------------------------
    def _fwd(self, sd, x, train=True):
        return self.forward(x, sd, train=train)

    def _pl(self, logits):
        return L.pseudo_label(logits, self.threshold, self.mode)

    def _sgd(self):
        """torch.optim.SGD(momentum, weight_decay) step.  reference: train.py:512,848."""
        with torch.no_grad():
            for k in self.pkeys:
                p = self.student[k]
                g = p.grad + self.wd * p
                if self.mom[k] is None:
                    self.mom[k] = g.clone()
                else:
                    self.mom[k].mul_(self.momentum).add_(g)
                p.add_(self.mom[k], alpha=-self.lr)
                p.grad = None

    def _ema(self):
        """reference: train.py:87-93 (parameters only, Q11)."""
        a = H.ema_alpha(self.iter_num, self.ema_decay)
        with torch.no_grad():
            for k in self.pkeys:
                self.teacher[k].mul_(a).add_(self.student[k].detach(), alpha=1 - a)

    # -- one iteration ------------------------------------------------------------
    def step(self, lb_x_w, lb_y, ulb_x_w, ulb_x_s, ulb_y, epoch_start=False, defer_update=False):
        """defer_update: stop after loss.backward() with the gradients left in `.grad` (data-parallel tests average them
        over replicas first); `finish_update()` then applies SGD + EMA + LR as the reference does."""
        ds, mode = self.dataset, self.mode
        B = len(ulb_x_s)
        epoch_num = self.iter_num // self.num_eval_iter
        if epoch_start:
            self.lq_u = self.lq_pl = self.lq_mask = None
        lb_mask = decode_labels(ds, lb_y)
        ulb_mask = decode_labels(ds, ulb_y)
        mshape = [len(lb_x_w), self.n_classes if ds == "fundus" else 1, self.patch, self.patch]

        # CutMix partner selection (train.py:612-627)
        if self.simple_ulb is None or len(self.simple_ulb) == 0:
            cut_img, cut_label, cut_mask = lb_x_w, lb_mask, torch.ones(mshape)
            choice = np.random.randint(0, len(lb_x_w), B)
        else:
            cut_img = torch.cat((lb_x_w, self.simple_ulb), 0)
            cut_label = torch.cat((lb_mask, self.cor_pl), 0)
            cut_mask = torch.cat((torch.ones(mshape), self.cor_mask), 0)
            n_s = min(int(B * 0.5), len(self.simple_ulb))
            c_lb = np.random.randint(0, len(lb_x_w), B - n_s)
            c_s = np.random.randint(len(lb_x_w), len(lb_x_w) + len(self.simple_ulb), n_s)
            choice = np.random.permutation(np.concatenate((c_lb, c_s)))
        mix_img = cut_img[choice]

        # FFT low-frequency amplitude mix, per image on the host (train.py:628-636, Q13)
        moved = []
        for i in range(len(lb_x_w)):
            amp_trg = H.amp_spectrum((ulb_x_w[i].numpy() + 1) * 127.5)
            f = H.freq_mix(((mix_img[i] + 1) * 127.5).numpy(), amp_trg, L=self.LB,
                           degree=self.iter_num / self.max_iterations)
            moved.append(np.clip(f, 0, 255).astype(np.float32))
        move_transx = torch.tensor(np.array(moved), dtype=torch.float32) / 127.5 - 1

        # teacher: three train-mode forwards under no_grad (train.py:638-667, Q11)
        with torch.no_grad():
            box = torch.from_numpy(np.stack([H.cutmix_box(self.patch, p=self.cutmix_prob) for _ in range(B)]))
            ib = box[:, None]
            logits_w = self._fwd(self.teacher, ulb_x_w)
            logits_w_ul = self._fwd(self.teacher, ulb_x_w * (1 - ib) + mix_img * ib)
            logits_w_lu = self._fwd(self.teacher, mix_img * (1 - ib) + ulb_x_w * ib)
            pl, mask = self._pl(logits_w)
            pl_w_ul, mask_w_ul = self._pl(logits_w_ul)
            pl_w_lu, mask_w_lu = self._pl(logits_w_lu)
            # student forward on the weak view: only its pseudo-label is used (Q3)
            stu_pl, _ = self._pl(self._fwd(self.student, ulb_x_w))
            pl_w, mask_w, pl_ul, mask_ul, pl_lu, mask_lu = L.mix_targets(
                mode, pl, mask, pl_w_ul, mask_w_ul, pl_w_lu, mask_w_lu, box, cut_label[choice], cut_mask[choice])
            x_s_ul = ulb_x_s * (1 - ib) + move_transx * ib
            x_s_lu = move_transx * (1 - ib) + ulb_x_s * ib

        # student: four forwards that carry gradient (train.py:699-702)
        lg_lb = self._fwd(self.student, lb_x_w)
        lg_ul = self._fwd(self.student, x_s_ul)
        lg_lu = self._fwd(self.student, x_s_lu)
        lg_s = self._fwd(self.student, ulb_x_s)

        # hardness and the low-quality sample forward (train.py:705-747, Q2, Q7)
        d = sample_dice(ds, stu_pl.numpy(), pl.numpy(), ret_arr=True)
        hardness = 1 - sum(d[1:], d[0].copy()) / self.n_part
        if epoch_num == 0:
            hardness[:] = 1
        lq_idx = int(np.argmax(hardness))            # first maximum, as the reference's scan
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
            ib_lq = torch.from_numpy(H.all_cover_box(region.numpy()))[None, None]
            lq_s = self.lq_u * (1 - ib_lq) + lb_x_w[[new_choice]] * ib_lq
            with torch.no_grad():                     # result unused (Q2); BN stats still move
                self._fwd(self.student, lq_s)
        self.lq_u = ulb_x_w[[lq_idx]].clone()
        self.lq_pl = pl[[lq_idx]].clone()
        self.lq_mask = mask[[lq_idx]].clone()

        # memory bank of easy unlabelled samples (train.py:749-782)
        simple = hardness < self.choice_th
        n_cur = int(simple.sum())
        sel = torch.from_numpy(simple)
        if self.simple_ulb is None or len(self.simple_ulb) == 0:
            self.simple_ulb, self.cor_pl = ulb_x_w[sel].clone(), pl[sel].clone()
            self.cor_gt, self.cor_mask = ulb_mask[sel].clone(), mask[sel].clone()
            self.cor_hardness = hardness[simple].copy()
            if len(self.simple_ulb) > 0:
                self.choice_th = min(self.choice_th, self.cor_hardness.max())
        elif n_cur > 0:
            # train.py:768-771.  The reference's `newlen = max_len - cur_simple_num` goes NEGATIVE once a batch holds more
            # easy samples than the queue is long (possible only with unlabel_bs > queue_len = 10; the reference runs 4),
            # and `bank[:negative]` then lets the bank grow by up to unlabel_bs - queue_len entries per step without bound
            # (29 GB after 1000 steps at B = 16).  Clamped at 0: identical whenever unlabel_bs <= queue_len.
            keep = max(0, self.queue_len - n_cur) if len(self.simple_ulb) + n_cur > self.queue_len else len(self.simple_ulb)
            self.simple_ulb = torch.cat((ulb_x_w[sel], self.simple_ulb[:keep]), 0)
            self.cor_pl = torch.cat((pl[sel], self.cor_pl[:keep]), 0)
            self.cor_gt = torch.cat((ulb_mask[sel], self.cor_gt[:keep]), 0)
            self.cor_mask = torch.cat((mask[sel], self.cor_mask[:keep]), 0)
            self.cor_hardness = np.concatenate((hardness[simple], self.cor_hardness[:keep]))
            self.choice_th = min(self.choice_th, self.cor_hardness.max())
        else:
            self.choice_th = min(self.increase * self.choice_th, 0.1)

        # losses (train.py:816-838; Q5, Q6)
        K = self.n_classes
        ce, dc = L.seg_loss(lg_lb, lb_mask, None, mode, K)
        sup = ce + dc
        w = H.consistency_weight(self.iter_num, self.max_iterations, self.consistency, self.rampup)
        ce, dc = L.seg_loss(lg_ul, pl_ul, mask_ul, mode, K)
        l_ul = ce + dc
        ce, dc = L.seg_loss(lg_lu, pl_lu, mask_lu, mode, K)
        l_lu = ce + dc
        ce, dc = L.seg_loss(lg_s, pl_w, mask_w, mode, K)
        l_s = ce + dc
        loss = sup + w * (l_ul + l_lu + w * l_s)

        loss.backward()
        ulb_dice = sample_dice(ds, pl.numpy(), ulb_mask.numpy())
        out = {"loss": float(loss.detach()), "sup": float(sup.detach()), "ul": float(l_ul.detach()), "lu": float(l_lu.detach()),
               "s": float(l_s.detach()), "w": w, "ulb_dice": [float(v) for v in ulb_dice],
               "mask_ratio": float(mask.mean())}
        if not defer_update:
            self.finish_update()
        return out

    def finish_update(self):
        self._sgd()
        self._ema()
        self.lr = H.poly_lr(self.base_lr, self.iter_num, self.max_iterations)   # Q10
        self.iter_num += 1
