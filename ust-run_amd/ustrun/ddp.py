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


def init(backend=None, device=None):
    """Join the process group described by the torchrun environment (RANK/WORLD_SIZE/MASTER_*)."""
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")   # "nccl" IS RCCL on ROCm
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, local, world


def make_grad_allreduce(world, bucket_elems=0):
    """callable(flat_grad) summing it over ranks in place.  bucket_elems > 0 splits the buffer into
    buckets launched back to back (async) so that the tail of one overlaps the head of the next."""
    if world <= 1:
        return None

    def allreduce(flat):
        if bucket_elems <= 0 or flat.numel() <= bucket_elems:
            dist.all_reduce(flat)
            return
        works = [dist.all_reduce(flat[o:o + bucket_elems], async_op=True) for o in range(0, flat.numel(), bucket_elems)]
        for w in works:
            w.wait()
    return allreduce


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
