"""Checkpoint I/O and meters used by the training scripts -- surface of the reference's
utils/util.py:167-183,259-297 (the rest of that file is dead code there).  state_dict keys are
identical to the reference's, so checkpoints interchange in both directions."""
import torch


class AverageMeter(object):
    """Computes and stores the average and current value"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def save_osmancheckpoint(epoch, ema_model, model, optimizer, best_dice, best_iter, stu_best_dice, stu_best_iter, path):
    torch.save({"epoch": epoch, "ema_state_dict": ema_model.state_dict(), "state_dict": model.state_dict(),
                "optimizer_state_dict": optimizer.state_dict(), "best_dice": best_dice, "best_iter": best_iter,
                "stu_best_dice": stu_best_dice, "stu_best_iter": stu_best_iter}, path)


def load_osmancheckpoint(path, ema_model, model, optimizer, from_ddp=False):
    ck = torch.load(path, map_location="cpu")
    ema_model.load_state_dict(ck["ema_state_dict"])
    model.load_state_dict(ck["state_dict"])
    optimizer.load_state_dict(ck["optimizer_state_dict"])
    return (ck["epoch"], ema_model, model, optimizer, ck["best_dice"], ck["best_iter"],
            ck["stu_best_dice"], ck["stu_best_iter"])
