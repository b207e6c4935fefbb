"""Data parallelism for the training step: one process per GPU, one exchange per step.

The path shards by samples: every rank runs the whole step on its own labelled +
unlabelled mini-batch with its own BatchNorm statistics, CutMix boxes and memory bank; the only
collective is ONE all-reduce (SUM) of the flat gradient buffer (31.04 M f32 = 124 MB) over RCCL/xGMI,
and the 1/world factor is folded into the fused SGD+EMA kernel.  SGD, EMA and the LR schedule are
deterministic functions of the reduced gradient, so student and teacher parameters stay bit-identical
across ranks without further communication (BatchNorm buffers intentionally diverge: "dsbn"-style
per-rank statistics).  The reference has no distributed path (its init_process_group at
utils/util.py:243-247 is dead code); this module is the build's addition.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


_WAIT = [None]
WAIT_TIMEOUT_S = 6 * 3600


def init(backend=None, device=None, wait_timeout_s=WAIT_TIMEOUT_S):
    """Join the process group described by the torchrun environment (RANK/WORLD_SIZE/MASTER_*).  The host-side gloo group
    behind `wait_for_rank0` is created HERE, while every rank is present and no GPU work is pending: `new_group` is itself a
    rendezvous of all ranks, and created lazily at the first epoch end it would have met rank 0 minutes late, in the middle
    of its validation."""
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")   # "nccl" IS RCCL on ROCm
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    if world > 1 and _WAIT[0] is None:
        import datetime
        _WAIT[0] = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=wait_timeout_s))
    return rank, local, world


def wait_for_rank0(timeout_s=WAIT_TIMEOUT_S):
    """Park the other ranks while rank 0 validates and saves at an epoch end.  A barrier on the RCCL group is itself a
    collective under that group's watchdog timeout, so a long validation would trip it just the same; the wait runs on the
    host-side gloo group `init` created, with its own long timeout."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    if _WAIT[0] is None:
        raise RuntimeError("ddp.wait_for_rank0: the process group was not set up by ddp.init (its gloo side group is missing)")
    import datetime
    dist.monitored_barrier(_WAIT[0], timeout=datetime.timedelta(seconds=timeout_s))


class GradReducer:
    """SUM of the flat gradient buffer over ranks, in place.

    Called with the buffer it reduces everything at once (optionally in back-to-back buckets).  The training step
    uses it in two pieces instead: the decoder gradients are the contiguous TAIL of the buffer (parameter order
    inc, down1..4, up1..4, outc) and are final once the head + decoder half of the backward has been issued, so
    `start_tail` launches their all-reduce there (async: RCCL runs it on its own stream behind the backward stream's
    work so far) and it overlaps the encoder half; `start_mid` does the same for down4's gradients once the encoder half
    has produced them; `finish` reduces the remaining head and joins."""

    def __init__(self, world, bucket_elems=0):
        self.world, self.bucket_elems = world, bucket_elems
        self._work, self._off = None, 0
        self._mid, self._mid_lo = None, 0

    def __call__(self, flat):
        b = self.bucket_elems
        if b <= 0 or flat.numel() <= b:
            dist.all_reduce(flat)
            return
        works = [dist.all_reduce(flat[o:o + b], async_op=True) for o in range(0, flat.numel(), b)]
        for w in works:
            w.wait()

    def start_tail(self, flat, off):
        self._off = int(off)
        self._work = dist.all_reduce(flat[self._off:], async_op=True)

    def start_mid(self, flat, lo):
        """After `start_tail`: [lo, tail offset) is final too (down4's gradients, 57 of the encoder's 75 MB, are the first
        the encoder half finishes) -- their all-reduce runs under the high-resolution encoder layers."""
        if self._work is None or not (0 <= lo < self._off):
            raise RuntimeError("start_mid: call start_tail first, with a larger offset")
        self._mid_lo = int(lo)
        self._mid = dist.all_reduce(flat[self._mid_lo:self._off], async_op=True)

    def finish(self, flat):
        if self._work is None:
            self(flat)
            return
        head = self._mid_lo if self._mid is not None else self._off
        if head > 0:
            dist.all_reduce(flat[:head])
        if self._mid is not None:
            self._mid.wait()
            self._mid = None
        self._work.wait()
        self._work = None


def make_grad_allreduce(world, bucket_elems=0):
    """GradReducer for world > 1, None for a single process (no collective)."""
    if world <= 1:
        return None
    return GradReducer(world, bucket_elems)


def rank_seed(base, rank):
    """Per-rank seeds for the synthetic generator / CutMix streams (seed + rank)."""
    return base + rank


def params_identical_across_ranks(flat):
    """Debug check: max |flat - flat_rank0| over ranks == 0."""
    ref = flat.clone()
    dist.broadcast(ref, 0)
    diff = (flat - ref).abs().max()
    dist.all_reduce(diff, op=dist.ReduceOp.MAX)
    return float(diff) == 0.0
