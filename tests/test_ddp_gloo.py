"""world_size-2 gloo test of the data-parallel exchange (CPU, no GPU): the flat-gradient all-reduce of
ustrun.ddp + the 1/world scale folded into the SGD update reproduces the average of R independent
replicas' gradients, and keeps parameters bit-identical across ranks."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sgd(p, g, v, lr, mu, wd, first, gscale):
    """same update as ustrun_sgd_ema (loss.hip) / torch.optim.SGD"""
    g = g * gscale + wd * p
    v = g.clone() if first else mu * v + g
    return p - lr * v, v


def _worker(rank, world, port, bucket, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "ust-run_amd")]
    from ustrun import ddp
    r, _, w = ddp.init("gloo")
    assert (r, w) == (rank, world)
    n = 10007
    torch.manual_seed(1337)                                # identical initial parameters on every rank
    p = torch.randn(n)
    v = torch.zeros(n)
    ar = ddp.make_grad_allreduce(world, bucket_elems=bucket)
    for step in range(3):
        g = torch.Generator().manual_seed(ddp.rank_seed(100 * step, rank))
        grad = torch.randn(n, generator=g)                 # this rank's local gradient
        local = grad.clone()
        ar(grad)                                           # SUM over ranks, in place
        # reference: mean of the R replicas' gradients
        tot = sum(torch.randn(n, generator=torch.Generator().manual_seed(ddp.rank_seed(100 * step, k))) for k in range(world))
        assert torch.allclose(grad, tot, rtol=0, atol=1e-6)
        assert not torch.equal(grad, local)
        p, v = _sgd(p, grad, v, 0.03, 0.9, 1e-4, step == 0, 1.0 / world)
        p_ref, _ = None, None
        assert ddp.params_identical_across_ranks(p)        # deterministic update => no drift, no extra traffic
    q.put((rank, float(p.double().sum())))
    dist.destroy_process_group()


def _run(bucket):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get() for _ in range(2))
    assert res[0] == res[1]


def test_flat_gradient_allreduce_world2():
    _run(bucket=0)


def test_bucketed_allreduce_world2():
    _run(bucket=4096)


def _worker_tail(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "ust-run_amd")]
    from ustrun import ddp
    ddp.init("gloo")
    n, off = 9001, 5432
    ar = ddp.make_grad_allreduce(world)
    for step in range(2):
        grad = torch.randn(n, generator=torch.Generator().manual_seed(ddp.rank_seed(100 * step, rank)))
        tot = sum(torch.randn(n, generator=torch.Generator().manual_seed(ddp.rank_seed(100 * step, k))) for k in range(world))
        head_before = grad[:off].clone()
        ar.start_tail(grad, off)            # decoder gradients (contiguous tail) leave first ...
        grad[:off] += 0.0                   # ... while the "encoder half" still produces the head
        assert torch.equal(grad[:off], head_before)
        ar.finish(grad)                     # head reduced, tail joined
        assert torch.allclose(grad, tot, rtol=0, atol=1e-6)
        # three pieces: tail, then the middle block (down4's gradients) once the encoder half has produced it, then the head
        grad = torch.randn(n, generator=torch.Generator().manual_seed(ddp.rank_seed(100 * step, rank)))
        mid = 2100
        ar.start_tail(grad, off)
        ar.start_mid(grad, mid)
        assert torch.equal(grad[:mid], head_before[:mid])
        ar.finish(grad)
        assert torch.allclose(grad, tot, rtol=0, atol=1e-6)
    q.put((rank, float(grad.double().sum())))
    dist.destroy_process_group()


def test_two_piece_reducer_world2():
    """GradReducer.start_tail / finish (the overlap used by the training step) == one all-reduce of the whole buffer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_tail, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get() for _ in range(2))
    assert res[0] == res[1]


def _worker_wait(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "ust-run_amd")]
    from ustrun import ddp
    ddp.init("gloo", wait_timeout_s=60)
    assert ddp._WAIT[0] is not None             # the side group exists before any epoch end
    t0 = time.time()
    if rank == 0:
        time.sleep(3.0)                         # "validation": rank 0 arrives late
    ddp.wait_for_rank0(timeout_s=60)
    waited = time.time() - t0
    # the epoch-end wait must not break the data path: a collective on the main group right after it
    x = torch.full((5,), float(rank + 1))
    dist.all_reduce(x)
    q.put((rank, waited, float(x[0])))
    dist.destroy_process_group()


def test_epoch_end_wait_with_rank0_delayed_world2():
    """`wait_for_rank0` (train.py's epoch-end wait) on the gloo side group that `ddp.init` creates up front: rank 1 parks
    until the late rank 0 arrives, nobody times out, the main group still works afterwards."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_wait, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = {r: (w, s) for r, w, s in (q.get() for _ in range(2))}
    assert res[1][0] >= 2.5                      # rank 1 really waited for rank 0's three seconds
    assert res[0][1] == 3.0 and res[1][1] == 3.0


def test_single_process_needs_no_collective():
    from ustrun import ddp
    assert ddp.make_grad_allreduce(1) is None
