"""GPU parity of the whole semi-supervised iteration (ustrun.trainer) against the CPU oracle step."""
import random

import numpy as np
import pytest
import torch

from oracle import unet_ref as U
from oracle.step_ref import RefTrainer

pytestmark = pytest.mark.gpu


def synth(dataset, B, C, H, seed):
    g = torch.Generator().manual_seed(seed)
    img = lambda: torch.randint(0, 256, (B, C, H, H), generator=g).float() / 127.5 - 1
    if dataset == "fundus":
        lab = lambda: torch.tensor([0.0, 128.0, 255.0])[torch.randint(0, 3, (B, H, H), generator=g)]
    elif dataset == "MNMS":
        def lab():
            cls = torch.randint(0, 4, (B, H, H), generator=g)
            return torch.stack([(cls == c).float() * 255 for c in (1, 2, 3)], dim=-1)
    else:
        lab = lambda: torch.tensor([0.0, 255.0])[torch.randint(0, 2, (B, H, H), generator=g)]
    return img(), lab(), img(), img(), lab()


@pytest.mark.parametrize("dataset,C,K", [("prostate", 1, 2), ("fundus", 3, 2), ("MNMS", 1, 4)])
def test_ssl_step_matches_oracle(dataset, C, K):
    from networks.unet_model import UNet
    from ustrun.trainer import SSLTrainer
    B, H, base, steps = 2, 32, 8, 3
    torch.manual_seed(1)
    sd_s = U.make_state_dict(C, K, base=base)
    sd_t = U.make_state_dict(C, K, base=base)
    kw = dict(max_iterations=300, threshold=0.52, patch_size=H, num_eval_iter=2)

    ref = RefTrainer(dataset, sd_s, **kw)
    ref.set_teacher(sd_t)
    stu, tea = UNet(C, K, base_channels=base), UNet(C, K, base_channels=base)
    stu.load_state_dict({k: v.clone() for k, v in sd_s.items()})
    tea.load_state_dict({k: v.clone() for k, v in sd_t.items()})
    trn = SSLTrainer(dataset, stu.cuda(), tea.cuda(), **kw)

    batches = [synth(dataset, B, C, H, 100 + s) for s in range(steps)]
    random.seed(7); np.random.seed(7)
    ref_out = [ref.step(*b, epoch_start=(s % 2 == 0)) for s, b in enumerate(batches)]
    random.seed(7); np.random.seed(7)
    got = []
    for s, b in enumerate(batches):
        trn.step(*[t.cuda() for t in b], epoch_start=(s % 2 == 0))
        got.append(trn.scalars())
    for r, o in zip(ref_out, got):
        for key in ("sup", "ul", "lu", "s", "loss"):
            np.testing.assert_allclose(o[key], r[key], rtol=2e-3, atol=1e-5, err_msg=key)
        assert o["w"] == r["w"]
        np.testing.assert_allclose(o["ulb_dice"], r["ulb_dice"], rtol=1e-3, atol=1e-4)
    assert trn.iter_num == ref.iter_num and abs(trn.lr - ref.lr) < 1e-12
    # parameters (student and EMA teacher) and BN running stats after the trajectory
    for name, sd_ref, m in (("student", ref.student, stu), ("teacher", ref.teacher, tea)):
        msd = m.state_dict()
        num = sum(float((msd[k].cpu().double() - sd_ref[k].detach().double()).square().sum()) for k in sd_ref if sd_ref[k].is_floating_point())
        den = sum(float(sd_ref[k].detach().double().square().sum()) for k in sd_ref if sd_ref[k].is_floating_point())
        assert (num / den) ** 0.5 < 2e-3, (name, (num / den) ** 0.5)
        for k in sd_ref:
            if k.endswith("num_batches_tracked"):
                assert int(msd[k]) == int(sd_ref[k]), k


def test_memory_bank_stays_bounded_when_the_batch_exceeds_the_queue():
    """unlabel_bs > queue_len: the reference's `newlen = max_len - cur_simple_num` would go negative and the bank would
    grow by unlabel_bs - queue_len entries per step; the clamp keeps it at the current batch's easy samples."""
    from networks.unet_model import UNet
    from ustrun.trainer import SSLTrainer
    B, H, base = 12, 32, 8
    torch.manual_seed(3)
    stu, tea = UNet(1, 2, base_channels=base).cuda(), UNet(1, 2, base_channels=base).cuda()
    trn = SSLTrainer("prostate", stu, tea, max_iterations=300, threshold=0.52, patch_size=H, num_eval_iter=1, queue_len=10)
    trn.choice_th = 2.0                      # every sample counts as easy (hardness <= 1)
    random.seed(1); np.random.seed(1)
    for s in range(6):
        trn.step(*[t.cuda() for t in synth("prostate", B, 1, H, 50 + s)], epoch_start=False)
        trn.choice_th = 2.0
        assert trn.simple_ulb is None or len(trn.simple_ulb) <= B, (s, len(trn.simple_ulb))
    assert len(trn.simple_ulb) == B == len(trn.cor_pl) == len(trn.cor_gt) == len(trn.cor_mask) == len(trn.cor_hardness)
