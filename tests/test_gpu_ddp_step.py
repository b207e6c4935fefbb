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


def _worker(rank, world, port, q, deeplab=False):
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
    if deeplab:      # BASELINE.json configs[4]'s model under the same exchange: one all-reduce of its flat gradient buffer
        from networks.deeplabv2 import DeepLabV2
        model = DeepLabV2("resnet50", 2, pretrained=False, dtype="bf16").cuda()
        ema = DeepLabV2("resnet50", 2, pretrained=False, dtype="bf16").cuda()
        tr = SSLTrainer("fundus", model, ema, patch_size=64, base_lr=1e-6, grad_allreduce=ddp.make_grad_allreduce(world),
                        world_size=world, fft="device")
        assert tr.batch_passes is False and tr.dec_off == 0
    else:
        model = UNet(3, 2, base_channels=16, dtype="bf16").cuda()
        ema = UNet(3, 2, base_channels=16, dtype="bf16").cuda()
        tr = SSLTrainer("fundus", model, ema, patch_size=64, grad_allreduce=ddp.make_grad_allreduce(world), world_size=world,
                        fft="device")
        assert tr.dec_off > 0
    random.seed(1212 + rank); np.random.seed(1337 + rank)
    ok = True
    for step in range(2 if deeplab else 3):
        b = [t.cuda() for t in synthetic.batch("fundus", 2, 3, 64, 100 * step + rank)]
        tr.step(*b, epoch_start=(step == 0))
        ok = ok and ddp.params_identical_across_ranks(tr.flat_p) and ddp.params_identical_across_ranks(tr.flat_t)
        ok = ok and bool(torch.isfinite(tr.flat_p).all())
    # the reduced gradient is the SUM over ranks: identical on both
    ok = ok and ddp.params_identical_across_ranks(tr.flat_g)
    q.put((rank, ok, float(tr.flat_p.double().abs().sum())))
    dist.destroy_process_group()


@pytest.mark.parametrize("deeplab", [False, True])
def test_training_step_world2_parameters_stay_identical(deeplab):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, deeplab)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    res = [q.get() for _ in range(2)]
    assert all(r[1] for r in res)
    assert res[0][2] == res[1][2]


# ---- reduced gradient == mean of R independent oracle replicas' gradients (SURVEY.md 8e verification) ----------------
_KW = dict(max_iterations=300, threshold=0.52, patch_size=64, num_eval_iter=2)


def _rank_batches(rank, steps):
    from ustrun import synthetic
    return [synthetic.batch("fundus", 2, 3, 64, 100 * s + rank) for s in range(steps)]


def _worker_oracle(rank, world, port, q, steps):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import random
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "ust-run_amd")]
    import torch.distributed as dist
    from networks.unet_model import UNet
    from oracle import unet_ref as U
    from ustrun import ddp
    from ustrun.trainer import SSLTrainer
    torch.cuda.set_device(0)
    ddp.init("gloo")
    torch.manual_seed(1)
    sd_s, sd_t = U.make_state_dict(3, 2, base=8), U.make_state_dict(3, 2, base=8)
    stu, tea = UNet(3, 2, base_channels=8), UNet(3, 2, base_channels=8)
    stu.load_state_dict({k: v.clone() for k, v in sd_s.items()})
    tea.load_state_dict({k: v.clone() for k, v in sd_t.items()})
    tr = SSLTrainer("fundus", stu.cuda(), tea.cuda(), grad_allreduce=ddp.make_grad_allreduce(world), world_size=world, fft="host", **_KW)
    random.seed(1212 + rank); np.random.seed(1337 + rank)
    grads = []
    for s, b in enumerate(_rank_batches(rank, steps)):
        tr.step(*[t.cuda() for t in b], epoch_start=(s % 2 == 0))
        grads.append([(v / world).cpu().numpy() for v in tr.grad_views])      # flat_g holds the SUM over ranks
    q.put((rank, grads, [p.detach().cpu().numpy() for p in stu.parameters()]))   # (numpy: pickled by value, no fd passing)
    dist.destroy_process_group()


def test_reduced_gradient_is_the_mean_of_independent_oracle_replicas():
    """Two HIP ranks (f32) against two CPU oracle replicas (oracle/step_ref.py) stepped in lockstep on the same per-rank
    batches and RNG streams, with the oracle's gradients averaged over the replicas before its SGD: after every step the
    all-reduced gradient / world equals the oracle mean (whole buffer rel-L2 <= 1e-4), and after the
    trajectory the parameters agree."""
    import random
    import numpy as np
    from oracle import unet_ref as U
    from oracle.step_ref import RefTrainer
    steps, world = 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_oracle, args=(r, world, port, q, steps)) for r in range(world)]
    for p in procs:
        p.start()
    # oracle replicas (this process, CPU) while the ranks run
    torch.manual_seed(1)
    sd_s, sd_t = U.make_state_dict(3, 2, base=8), U.make_state_dict(3, 2, base=8)
    reps, rng = [], []
    for r in range(world):
        t = RefTrainer("fundus", sd_s, **_KW)
        t.set_teacher(sd_t)
        reps.append(t)
        random.seed(1212 + r); np.random.seed(1337 + r)
        rng.append((random.getstate(), np.random.get_state()))
    batches = [_rank_batches(r, steps) for r in range(world)]
    pk = U.param_keys(sd_s)
    ref_grads = []
    for s in range(steps):
        for r, t in enumerate(reps):                       # each replica draws from its own RNG streams
            random.setstate(rng[r][0]); np.random.set_state(rng[r][1])
            t.step(*batches[r][s], epoch_start=(s % 2 == 0), defer_update=True)
            rng[r] = (random.getstate(), np.random.get_state())
        mean = [sum(t.student[k].grad for t in reps) / world for k in pk]
        ref_grads.append([m.clone() for m in mean])
        for t in reps:
            for k, m in zip(pk, mean):
                t.student[k].grad = m.clone()
            t.finish_update()
    res = {}
    for _ in range(world):
        rank, grads, params = q.get(timeout=300)
        res[rank] = ([[torch.from_numpy(a) for a in g] for g in grads], [torch.from_numpy(a) for a in params])
    for p in procs:
        p.join(300)
        assert p.exitcode == 0

    def rel(a, b):
        num = sum(float((x.double() - y.double()).square().sum()) for x, y in zip(a, b))
        den = sum(float(y.double().square().sum()) for y in b)
        return (num / den) ** 0.5
    for s in range(steps):
        for r in range(world):
            e = rel(res[r][0][s], ref_grads[s])
            assert e < 1e-4, (s, r, e)
        assert all(torch.equal(a, b) for a, b in zip(res[0][0][s], res[1][0][s]))      # identical on both ranks
    for r in range(world):
        e = rel(res[r][1], [reps[r].student[k].detach() for k in pk])
        assert e < 1e-4, ("params", r, e)


def test_grad_reducer_on_rccl_single_rank():
    """The data-parallel exchange on the REAL backend (RCCL), as far as one GPU allows: a world-size-1 `nccl` process group, the
    GradReducer's three overlapped all-reduces (start_tail between the backward halves from the autograd thread, start_mid,
    finish) and the 1/world scale in the fused SGD.  With one rank every all-reduce is the identity, so parameters after two
    steps must be bit-identical to a run without any collective -- what is exercised is RCCL's stream ordering against the
    backward stream and the async work handles, which gloo cannot show."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl1, args=(_free_port(), q))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    ok, backend, nranks = q.get(timeout=10)
    assert backend == "nccl" and nranks == 1 and ok


def _worker_rccl1(port, q):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    finals = []
    for use_reducer in (True, False):
        torch.manual_seed(1337)
        model = UNet(3, 2, base_channels=16, dtype="bf16").cuda()
        ema = UNet(3, 2, base_channels=16, dtype="bf16").cuda()
        tr = SSLTrainer("fundus", model, ema, patch_size=64, grad_allreduce=ddp.GradReducer(1) if use_reducer else None, world_size=1,
                        fft="device")
        random.seed(1212); np.random.seed(1337)
        for step in range(2):
            b = [t.cuda() for t in synthetic.batch("fundus", 2, 3, 64, 100 * step)]
            tr.step(*b, epoch_start=(step == 0))
        torch.cuda.synchronize()
        finals.append((tr.flat_p.clone(), tr.flat_t.clone()))
    ok = torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1]) and bool(torch.isfinite(finals[0][0]).all())
    q.put((ok, dist.get_backend(), dist.get_world_size()))
    dist.destroy_process_group()
