"""Python wrappers of the loss / pseudo-label / target-mixing / optimizer entry points."""
from __future__ import annotations

import torch

from . import _lib as L
from .engine import stream_ptr

_MODE = {"softmax": L.LOSS_SOFTMAX, "sigmoid": L.LOSS_SIGMOID}


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a float32 HIP tensor, got {t.dtype} on {t.device}")
    return t.contiguous()


def _shape(logits):
    N, K = logits.shape[:2]
    HW = logits[0, 0].numel()
    return N, K, HW


def _check_targets(logits, target, mask, mode):
    N, K, HW = _shape(logits)
    if mode == "softmax":
        if target.dtype != torch.int64 or target.numel() != N * HW:
            raise RuntimeError(f"softmax target must be int64 [N,H,W], got {target.dtype} {tuple(target.shape)}")
        if mask is not None and mask.numel() != N * HW:
            raise RuntimeError(f"softmax mask must have N*H*W elements, got {tuple(mask.shape)}")
    else:
        if target.dtype != torch.float32 or target.numel() != N * K * HW:
            raise RuntimeError(f"sigmoid target must be float32 [N,K,H,W], got {target.dtype} {tuple(target.shape)}")
        if mask is not None and mask.numel() != N * K * HW:
            raise RuntimeError(f"sigmoid mask must have N*K*H*W elements, got {tuple(mask.shape)}")


def seg_loss_fwd(logits, target, mask, mode):
    """-> out tensor: [0]=ce mean, [1]=dice, [2:]=reduced sums kept for the backward."""
    lib = L.lib()
    logits = _f32c(logits, "logits")
    target = target.contiguous()
    mask = None if mask is None else _f32c(mask, "mask")
    _check_targets(logits, target, mask, mode)
    N, K, HW = _shape(logits)
    out = torch.empty(4 + 3 * K, dtype=torch.float32, device=logits.device)
    nb = lib.ustrun_loss_partials_bytes(N, K, HW)
    part = torch.empty(nb // 4, dtype=torch.float32, device=logits.device)
    L.check(lib.ustrun_seg_loss_fwd(logits.data_ptr(), target.data_ptr(), L.ptr(mask), N, K, HW, _MODE[mode],
                                    out.data_ptr(), part.data_ptr(), nb, stream_ptr()), "ustrun_seg_loss_fwd")
    return out


def seg_loss_bwd(logits, target, mask, mode, sums, gscale=1.0, ce_weight=1.0, dice_weight=1.0, gdev=None, out=None):
    """out: where the gradient goes (a contiguous f32 tensor of the logits' shape, e.g. this term's slice of the batched
    passes' gradient); a new tensor otherwise."""
    lib = L.lib()
    logits = _f32c(logits, "logits")
    target = target.contiguous()
    mask = None if mask is None else _f32c(mask, "mask")
    N, K, HW = _shape(logits)
    if out is None:
        dl = torch.empty_like(logits)
    else:
        if out.shape != logits.shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != logits.device:
            raise RuntimeError("seg_loss_bwd: out must be a contiguous float32 tensor of the logits' shape on their device")
        dl = out
    L.check(lib.ustrun_seg_loss_bwd(logits.data_ptr(), target.data_ptr(), L.ptr(mask), N, K, HW, _MODE[mode],
                                    sums.data_ptr(), L.ptr(gdev), float(gscale), float(ce_weight), float(dice_weight),
                                    dl.data_ptr(), stream_ptr()), "ustrun_seg_loss_bwd")
    return dl


class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, mask, mode, ce_weight, dice_weight):
        out = seg_loss_fwd(logits, target, mask, mode)
        ctx.save_for_backward(logits, target, out) if mask is None else ctx.save_for_backward(logits, target, out, mask)
        ctx.cfg = (mode, ce_weight, dice_weight, mask is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        mode, cw, dw, has_mask = ctx.cfg
        saved = ctx.saved_tensors
        logits, target, out = saved[:3]
        mask = saved[3] if has_mask else None
        g = gout.contiguous()
        dl = seg_loss_bwd(logits, target, mask, mode, out, 1.0, cw, dw, gdev=g)
        return dl, None, None, None, None, None


def seg_loss(logits, target, mask, mode, ce_weight=1.0, dice_weight=1.0):
    """(ce, dice) of one loss term (train.py:816-817,829-836); differentiable w.r.t. logits."""
    out = _SegLossFn.apply(logits, target, mask, mode, float(ce_weight), float(dice_weight))
    return out[0], out[1]


_ACT = {"none": 0, "softmax": 1, "sigmoid": 2}


class _DiceFn(torch.autograd.Function):
    """DiceLossWithMask.forward in any mode combination (utils/losses.py:236-268) through ustrun_dice_fwd/_bwd."""

    @staticmethod
    def forward(ctx, logits, target, mask, act, multi, weight):
        import ctypes as C
        lib = L.lib()
        logits = _f32c(logits, "inputs")
        N, K, HW = _shape(logits)
        if multi:
            target = _f32c(target.float() if target.dtype != torch.float32 else target, "target")
        else:
            target = target.contiguous() if target.dtype == torch.int64 else _f32c(target.float(), "target")
        mask = None if mask is None else _f32c(mask.float() if mask.dtype != torch.float32 else mask, "mask")
        if target.numel() not in ((N * HW,) if not multi else (N * HW, N * K * HW)):
            raise AssertionError("predict & target shape do not match")          # losses.py:253
        if mask is not None and mask.numel() not in ((N * HW,) if not multi else (N * HW, N * K * HW)):
            raise RuntimeError(f"mask of {tuple(mask.shape)} does not match inputs {tuple(logits.shape)}")
        tper = int(target.numel() != N * K * HW or not multi)              # one value per pixel (class index, or broadcast over classes)
        mper = int(mask is not None and (mask.numel() != N * K * HW or not multi))
        w = None if weight is None else (C.c_float * K)(*[float(v) for v in weight])
        out = torch.empty(1 + 3 * K, dtype=torch.float32, device=logits.device)
        nb = lib.ustrun_loss_partials_bytes(N, K, HW)
        part = torch.empty(nb // 4, dtype=torch.float32, device=logits.device)
        cfg = (int(target.dtype == torch.int64), tper, mper, N, K, HW, _ACT[act], int(multi))
        L.check(lib.ustrun_dice_fwd(logits.data_ptr(), target.data_ptr(), cfg[0], cfg[1], L.ptr(mask), cfg[2], N, K, HW, cfg[6], cfg[7], w,
                                    out.data_ptr(), part.data_ptr(), nb, stream_ptr()), "ustrun_dice_fwd")
        ctx.save_for_backward(logits, target, out) if mask is None else ctx.save_for_backward(logits, target, out, mask)
        ctx.cfg, ctx.w = cfg, w
        return out[0]

    @staticmethod
    def backward(ctx, g):
        lib = L.lib()
        saved = ctx.saved_tensors
        logits, target, out = saved[:3]
        mask = saved[3] if len(saved) > 3 else None
        t64, tper, mper, N, K, HW, act, multi = ctx.cfg
        g = g.contiguous().float().reshape(1)
        dl = torch.empty_like(logits)
        L.check(lib.ustrun_dice_bwd(logits.data_ptr(), target.data_ptr(), t64, tper, L.ptr(mask), mper, N, K, HW, act, multi, ctx.w,
                                    out.data_ptr(), g.data_ptr(), 1.0, dl.data_ptr(), stream_ptr()), "ustrun_dice_bwd")
        return dl, None, None, None, None, None


def dice_general(logits, target, mask=None, act="none", multi=False, weight=None):
    """One DiceLossWithMask value, differentiable w.r.t. logits.  per class (multi False): target = class indices with N*H*W
    elements; multi: target / mask with N*K*H*W elements or N*H*W (broadcast over the classes)."""
    return _DiceFn.apply(logits, target, mask, act, bool(multi), weight)


def pseudo_label(logits, threshold, mode):
    """train.py:648-667.  softmax -> (label int64 [N,H,W], mask f32 [N,1,H,W]); sigmoid -> f32 [N,K,H,W] x2."""
    lib = L.lib()
    logits = _f32c(logits, "logits")
    N, K, HW = _shape(logits)
    sp = logits.shape[2:]
    if mode == "softmax":
        label = torch.empty((N,) + tuple(sp), dtype=torch.int64, device=logits.device)
        mask = torch.empty((N, 1) + tuple(sp), dtype=torch.float32, device=logits.device)
    else:
        label = torch.empty_like(logits)
        mask = torch.empty_like(logits)
    L.check(lib.ustrun_pseudo_label(logits.data_ptr(), N, K, HW, float(threshold), _MODE[mode], label.data_ptr(),
                                    mask.data_ptr(), stream_ptr()), "ustrun_pseudo_label")
    return label, mask


def mix_targets(mode, box, pl, mask, pl_w_ul, mask_w_ul, pl_w_lu, mask_w_lu, cut_label, cut_mask):
    """train.py:677-697 -> (pl_w, mask_w, pl_ul, mask_ul, pl_lu, mask_lu)."""
    lib = L.lib()
    ts = [t.contiguous() for t in (pl, mask, pl_w_ul, mask_w_ul, pl_w_lu, mask_w_lu, cut_label, cut_mask)]
    box = _f32c(box, "box")
    N = box.shape[0]
    HW = box[0].numel()
    K = mask.shape[1] if mode == "sigmoid" else 1
    want = torch.float32 if mode == "sigmoid" else torch.int64
    for t in (ts[0], ts[2], ts[4], ts[6]):
        if t.dtype != want or t.numel() != ts[0].numel():
            raise RuntimeError("mix_targets: label tensors must share dtype/shape")
    for t in (ts[1], ts[3], ts[5], ts[7]):
        if t.dtype != torch.float32 or t.numel() != ts[1].numel():
            raise RuntimeError("mix_targets: mask tensors must be float32 of one shape")
    outs = [torch.empty_like(ts[0]), torch.empty_like(ts[1]), torch.empty_like(ts[0]), torch.empty_like(ts[1]),
            torch.empty_like(ts[0]), torch.empty_like(ts[1])]
    L.check(lib.ustrun_mix_targets(_MODE[mode], N, K, HW, box.data_ptr(), *[t.data_ptr() for t in ts],
                                   *[t.data_ptr() for t in outs], stream_ptr()), "ustrun_mix_targets")
    return tuple(outs)


def box_mix(a, b, box):
    """a*(1-box) + b*box with box [N,H,W] broadcast over channels (train.py:644-646,688,691)."""
    lib = L.lib()
    a, b, box = _f32c(a, "a"), _f32c(b, "b"), _f32c(box, "box")
    N, C = a.shape[:2]
    out = torch.empty_like(a)
    L.check(lib.ustrun_box_mix(a.data_ptr(), b.data_ptr(), box.data_ptr(), N, C, a[0, 0].numel(), out.data_ptr(),
                               stream_ptr()), "ustrun_box_mix")
    return out


def row_ptrs(t):
    """Device address of every row t[i] of a contiguous tensor (for `assemble`)."""
    if not t.is_cuda or not t.is_contiguous():
        raise RuntimeError("row_ptrs: expected a contiguous HIP tensor")
    base, step = t.data_ptr(), (t[0].numel() * t.element_size() if len(t) else 0)
    return [base + i * step for i in range(len(t))]


def assemble(rows, like, HW=0, out=None):
    """One launch that builds a batch from rows scattered over the device (ustrun_assemble): `rows` = a list of
    (a, b, box) device ADDRESSES -- b = box = 0: row = the bytes at a; else the f32 CutMix composite a*(1-box) + b*box with
    box [HW] broadcast over the row's channels.  `like`: a tensor whose [0] gives the row's shape and dtype (e.g. one of the
    sources).  Replaces x[index], torch.cat and clone chains around the forwards (train.py:627,643-647,689-702,734)
    without an index tensor or an intermediate concatenation.  The caller keeps the source tensors alive until this returns
    (the launch is then ordered on the stream)."""
    lib = L.lib()
    n = len(rows)
    row_shape, row_bytes = tuple(like.shape[1:]), like[0].numel() * like.element_size()
    if out is None:
        out = torch.empty((n,) + row_shape, dtype=like.dtype, device=like.device)
    elif tuple(out.shape) != (n,) + row_shape or out.dtype != like.dtype or not out.is_contiguous():
        raise RuntimeError("assemble: out does not match the rows")
    for o in range(0, n, L.ASM_MAX):
        part = rows[o:o + L.ASM_MAX]
        tab = (L.AsmRow * len(part))()
        for i, (a, b, bx) in enumerate(part):
            tab[i].a, tab[i].b, tab[i].box = a, b or None, bx or None
        L.check(lib.ustrun_assemble(tab, len(part), row_bytes, int(HW), out.data_ptr() + o * row_bytes, stream_ptr()),
                "ustrun_assemble")
    return out


_LABEL_KIND = {"fundus": 0, "prostate": 1, "BUSI": 2, "MNMS": 3}


def decode_labels(dataset, y):
    """train.py:590-608, train_mnms.py:549-556 in one pass (ustrun_decode_labels): the loader's float label tensor ->
    fundus: float [N,2,H,W] {cup = y == 0, disc = y <= 128}; prostate / BUSI: int64 [N,H,W] foreground map; MNMS: int64
    class map from the three 255-coded channels."""
    lib = L.lib()
    y = _f32c(y, "labels")
    kind = _LABEL_KIND[dataset]
    if kind == 3:
        if y.dim() != 4 or y.shape[-1] != 3:
            raise RuntimeError(f"MNMS labels must be [N,H,W,3], got {tuple(y.shape)}")
        sp = tuple(y.shape[1:3])
    else:
        if y.dim() != 3:
            raise RuntimeError(f"{dataset} labels must be [N,H,W], got {tuple(y.shape)}")
        sp = tuple(y.shape[1:])
    N, HW = y.shape[0], sp[0] * sp[1]
    if kind == 0:
        out = torch.empty((N, 2) + sp, dtype=torch.float32, device=y.device)
    else:
        out = torch.empty((N,) + sp, dtype=torch.int64, device=y.device)
    L.check(lib.ustrun_decode_labels(y.data_ptr(), kind, N, HW, out.data_ptr(), stream_ptr()), "ustrun_decode_labels")
    return out


def region_bbox_partials(planes, H, W, out):
    """Per-block bounding rectangles of the union of the planes' non-zero pixels -> `out` (int32 [BBOX_BLOCKS,4] on the
    device: {min y, max y, min x, max x}, {H,-1,W,-1} where a block saw none); fold them with `fold_bbox` on the host."""
    import ctypes as C
    lib = L.lib()
    bits = 0
    arr = (C.c_void_p * len(planes))()
    for k, t in enumerate(planes):
        if not t.is_cuda or not t.is_contiguous() or t.numel() != H * W or t.dtype not in (torch.float32, torch.int64):
            raise RuntimeError("region_bbox: planes must be contiguous float32 / int64 HIP tensors of H*W elements")
        bits |= int(t.dtype == torch.int64) << k
        arr[k] = t.data_ptr()
    L.check(lib.ustrun_region_bbox(arr, len(planes), bits, H, W, out.data_ptr(), stream_ptr()), "ustrun_region_bbox")
    return out


def fold_bbox(partial):
    """numpy int32 [blocks,4] -> the rectangle {y0, y1, x0, x1} of train.py:242-251 (exclusive ends), or None when no pixel is set."""
    y0, y1 = int(partial[:, 0].min()), int(partial[:, 1].max())
    x0, x1 = int(partial[:, 2].min()), int(partial[:, 3].max())
    if y1 < 0:
        return None
    return (y0, y1 + 1, x0, x1 + 1)


def rect_masks(rects, H, W, device):
    """{0,1} maps [N,H,W] on the device from N host rectangles {y0,y1,x0,x1} (train.py:222-251's maps, built where
    they are used; the corners ride in the launch's arguments, so nothing waits on a copy)."""
    import numpy as np
    lib = L.lib()
    r = np.ascontiguousarray(rects, dtype=np.int32).reshape(-1, 4)
    if len(r) == 0:
        raise RuntimeError("rect_masks: no rectangles")
    out = torch.empty((len(r), H, W), dtype=torch.float32, device=device)
    for o in range(0, len(r), MAX_RECTS):                 # the launch's argument block holds 64 rectangles
        part = np.ascontiguousarray(r[o:o + MAX_RECTS])
        L.check(lib.ustrun_rect_masks(part.ctypes.data, len(part), H, W, out[o:o + len(part)].data_ptr(), stream_ptr()),
                "ustrun_rect_masks")
    return out


UPLOAD_MAX = 2048
MAX_RECTS = 64


def upload_small(arr, device, dtype):
    """A few host values -> a device tensor, stream-ordered, without a copy-engine transfer or a host wait."""
    import numpy as np
    lib = L.lib()
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(dtype).contiguous()
    out = torch.empty(t.shape, dtype=dtype, device=device)
    nb = t.numel() * t.element_size()
    if nb == 0:
        return out
    if nb % 4 or nb > 64 * UPLOAD_MAX:
        raise RuntimeError(f"upload_small: {nb} bytes (needs a multiple of 4, at most {64 * UPLOAD_MAX}: it is meant for a few values)")
    for o in range(0, nb, UPLOAD_MAX):                    # 2 KB per launch's argument block
        n = min(UPLOAD_MAX, nb - o)
        L.check(lib.ustrun_upload_small(out.data_ptr() + o, t.data_ptr() + o, n, stream_ptr()), "ustrun_upload_small")
    return out


def dice_counts(pred, gt, by_class=False, n_classes=1):
    """Per-sample {|pred|, |gt|, |pred&gt|} as int32 [N,K,3] (inputs of utils/metrics.py:114-146)."""
    lib = L.lib()
    pred, gt = pred.contiguous(), gt.contiguous()
    N = pred.shape[0]
    if by_class:
        K, HW = n_classes, pred[0].numel()
    else:
        K = pred.shape[1] if pred.dim() == 4 else 1
        HW = pred[0].numel() // K
    counts = torch.empty((N, K, 3), dtype=torch.int32, device=pred.device)
    L.check(lib.ustrun_dice_counts(pred.data_ptr(), gt.data_ptr(), int(pred.dtype == torch.int64),
                                   int(gt.dtype == torch.int64), N, K, HW, int(by_class), counts.data_ptr(),
                                   stream_ptr()), "ustrun_dice_counts")
    return counts


def sgd_ema(p, g, v, t, lr, momentum, weight_decay, first, alpha, grad_scale=1.0):
    """Fused SGD(momentum, wd) step on flat f32 buffers + EMA teacher update (train.py:512,848,87-93)."""
    lib = L.lib()
    L.check(lib.ustrun_sgd_ema(p.data_ptr(), g.data_ptr(), v.data_ptr(), L.ptr(t), p.numel(), float(lr), float(momentum),
                               float(weight_decay), int(first), float(alpha), float(grad_scale), stream_ptr()),
            "ustrun_sgd_ema")


class LossScale:
    """torch.cuda.amp.GradScaler semantics (train.py:552,842-845) on the device: no step waits for the host.  `state` is the four
    floats {scale, scale, growth_tracker, found_inf} of include/ustrun.h; pass `state` as `gdev` to `seg_loss_bwd` to scale the
    backward, then call `step(...)` where the reference calls scaler.step(optimizer); scaler.update()."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.state = torch.tensor([init_scale, init_scale, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval

    def step(self, p, g, v, t, lr, momentum, weight_decay, first, alpha, grad_scale=1.0):
        lib = L.lib()
        st = stream_ptr()
        L.check(lib.ustrun_amp_check(g.data_ptr(), g.numel(), self.state.data_ptr(), st), "ustrun_amp_check")
        L.check(lib.ustrun_sgd_ema_scaled(p.data_ptr(), g.data_ptr(), v.data_ptr(), L.ptr(t), p.numel(), float(lr), float(momentum),
                                          float(weight_decay), int(first), float(alpha), float(grad_scale), self.state.data_ptr(),
                                          st), "ustrun_sgd_ema_scaled")
        L.check(lib.ustrun_amp_update(self.state.data_ptr(), float(self.growth_factor), float(self.backoff_factor),
                                      int(self.growth_interval), st), "ustrun_amp_update")

    def skipped_steps(self):
        """(host read) -> (steps skipped, steps seen)"""
        s = self.state.cpu()
        return int(s[4]), int(s[5])

    def get_scale(self):
        """(host read: logging / checkpoints only)"""
        return float(self.state[0])

    def state_dict(self):
        s = self.state.cpu()
        return {"scale": float(s[0]), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(s[2])}

    def load_state_dict(self, sd):
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])
        self.state.copy_(torch.tensor([sd["scale"], sd["scale"], float(sd.get("_growth_tracker", 0)), 0.0, 0.0, 0.0, 0.0, 0.0]))
