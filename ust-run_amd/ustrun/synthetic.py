"""Seeded synthetic batches with the value domains the reference's data pipeline emits
(SURVEY.md 8d): images on the 256-level grid k/127.5-1 (Normalize_tf, custom_transforms.py:650-684),
labels as float tensors holding uint8 values (ToTensor, :728-753)."""
import torch


# Task difficulty (tools/gen_traj_golden.py, tests/test_gpu_trajectory.py): the default task -- large bright discs -- is
# learnt to Dice 0.997 within 200 steps and then says little about the trajectory.  "medium" (smaller discs, 40 grey
# levels over stronger noise) is at teacher Dice 0.963 after 200 steps with the loss still falling, and is the hardest of
# the tasks tried (tools/calib_task.py, profiles/r02_calib_task.log) on which two f32 runs that differ by 1e-6 in the
# initial weights still land within 3e-4 of each other in EMA-teacher validation Dice; on "hard" (Dice 0.62) such twins
# are 7e-3 apart, so no 1e-3 gate can be read from it.
TASKS = {"default": dict(contrast=100.0, noise=0.6, rmin=0.15, rspan=0.15),
         "medium": dict(contrast=40.0, noise=0.8, rmin=0.10, rspan=0.15),
         "hard": dict(contrast=14.0, noise=1.0, rmin=0.06, rspan=0.10)}


def _discs(B, H, g, radii, rmin=0.15, rspan=0.15):
    """Concentric random discs per sample -> list of boolean maps, outermost first."""
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(H), indexing="ij")
    cy = (0.3 + 0.4 * torch.rand(B, generator=g)) * H
    cx = (0.3 + 0.4 * torch.rand(B, generator=g)) * H
    r0 = (rmin + rspan * torch.rand(B, generator=g)) * H
    d2 = (yy[None] - cy[:, None, None]) ** 2 + (xx[None] - cx[:, None, None]) ** 2
    return [d2 <= (r0 * f)[:, None, None] ** 2 for f in radii]


def images(B, C, H, g, fg=None, contrast=100.0, noise=0.6):
    """Noise on the 256-level grid, 3x3 box low-passed; when a foreground map [B,H,H] in [0,1] is given the
    foreground is brightened so that the segmentation task is learnable (Dice is then a meaningful gate)."""
    x = torch.randint(0, 256, (B, C, H, H), generator=g).float()
    x = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode="replicate"), 3, 1)
    if fg is not None:
        x = noise * x + contrast * fg[:, None]
    return x.round().clamp(0, 255) / 127.5 - 1


def foreground(dataset, y):
    """[B,H,H] map in [0,1] of how 'bright' each pixel's class is."""
    if dataset == "fundus":
        return (y <= 128).float() * 0.5 + (y == 0).float() * 0.5
    if dataset == "prostate":
        return (y == 0).float()
    if dataset == "BUSI":
        return (y == 255).float()
    return (y[..., 0] == 255).float() * 0.33 + (y[..., 1] == 255).float() * 0.66 + (y[..., 2] == 255).float()


def labels(dataset, B, H, g, rmin=0.15, rspan=0.15):
    if dataset == "fundus":                      # 0 = cup, 128 = disc rim, 255 = background
        disc, cup = _discs(B, H, g, (1.0, 0.5), rmin, rspan)
        y = torch.full((B, H, H), 255.0)
        y[disc] = 128.0
        y[cup] = 0.0
        return y
    if dataset == "prostate":                    # foreground = 0 (train.py:600)
        (fg,) = _discs(B, H, g, (1.0,), rmin, rspan)
        return torch.where(fg, torch.tensor(0.0), torch.tensor(255.0))
    if dataset == "BUSI":                        # foreground = 255 (train.py:605)
        (fg,) = _discs(B, H, g, (1.0,), rmin, rspan)
        return torch.where(fg, torch.tensor(255.0), torch.tensor(0.0))
    a, b, c = _discs(B, H, g, (1.0, 0.7, 0.4), rmin, rspan)   # MNMS: channel c == 255 <=> class c+1, disjoint rings
    y = torch.zeros(B, H, H, 3)
    y[..., 0][a & ~b] = 255.0
    y[..., 1][b & ~c] = 255.0
    y[..., 2][c] = 255.0
    return y


def batch(dataset, B, C, H, seed, task="default"):
    """(lb_x_w, lb_y, ulb_x_w, ulb_x_s, ulb_y) on the CPU."""
    t = TASKS[task] if isinstance(task, str) else task
    g = torch.Generator().manual_seed(seed)
    lb_y, ulb_y = labels(dataset, B, H, g, t["rmin"], t["rspan"]), labels(dataset, B, H, g, t["rmin"], t["rspan"])
    f_lb, f_ulb = foreground(dataset, lb_y), foreground(dataset, ulb_y)
    im = lambda f: images(B, C, H, g, f, t["contrast"], t["noise"])
    return im(f_lb), lb_y, im(f_ulb), im(f_ulb), ulb_y


def test_loaders(dataset, domain_num, batches, test_bs, C, H, seed, task="default"):
    """One list of (image, raw label) CPU batches per domain, seeded: stands in for the reference's per-domain test
    DataLoaders (test.py:222-230) when no dataset is on disk."""
    t = TASKS[task] if isinstance(task, str) else task
    out = []
    for d in range(domain_num):
        dom = []
        for b in range(batches):
            g = torch.Generator().manual_seed(seed + 7919 * d + b)
            y = labels(dataset, test_bs, H, g, t["rmin"], t["rspan"])
            dom.append((images(test_bs, C, H, g, foreground(dataset, y), t["contrast"], t["noise"]), y))
        out.append(dom)
    return out
