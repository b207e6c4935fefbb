#!/usr/bin/env python3
"""train.py -- MI355X build of the reference's training driver for the U-Net hot path.

Keeps the reference's command line (train.py:38-79: same flag names and defaults); new flags are
additive (--synthetic, --backend_dtype, --data_root, --fft, --log_every).  The step itself is
ustrun.trainer.SSLTrainer (HIP kernels through libustrun.so).  Data loading, augmentation,
the medpy metrics and tensorboard logging are outside this build's scope; the Dice
validation at every epoch end is ustrun.evaluate.validate: batches
come from the seeded synthetic generator unless a loader is plugged in through `make_loaders`.

Single GPU:   python train.py --dataset fundus --save_name run0 --synthetic 1
Data parallel: python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...
"""
import argparse
import logging
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

parser = argparse.ArgumentParser()
parser.add_argument('--dataset', type=str, default='BUSI', choices=['fundus', 'prostate', 'BUSI'])
parser.add_argument("--save_name", type=str, default="debug", help="experiment_name")
parser.add_argument("--overwrite", action='store_true')
parser.add_argument("--model", type=str, default="unet", help="model_name")
parser.add_argument("--max_iterations", type=int, default=60000, help="maximum epoch number to train")
parser.add_argument('--num_eval_iter', type=int, default=500)
parser.add_argument("--deterministic", type=int, default=1, help="whether use deterministic training")
parser.add_argument("--base_lr", type=float, default=0.03, help="segmentation network learning rate")
parser.add_argument("--seed", type=int, default=1337, help="random seed")
parser.add_argument("--gpu", type=str, default='0')
parser.add_argument('--load', action='store_true')
parser.add_argument('--eval', action='store_true')
parser.add_argument('--load_path', type=str, default='../model/lb1_ratio0.2/iter_6000.pth')
parser.add_argument("--threshold", type=float, default=0.95, help="confidence threshold for using pseudo-labels")
parser.add_argument('--amp', type=int, default=1, help='use mixed precision training or not')
parser.add_argument("--label_bs", type=int, default=4, help="labeled_batch_size per gpu")
parser.add_argument("--unlabel_bs", type=int, default=4)
parser.add_argument("--test_bs", type=int, default=1)
parser.add_argument('--domain_num', type=int, default=6)
parser.add_argument('--lb_domain', type=int, default=1)
parser.add_argument('--lb_num', type=int, default=40)
parser.add_argument('--lb_ratio', type=float, default=0)
parser.add_argument("--ema_decay", type=float, default=0.99, help="ema_decay")
parser.add_argument("--consistency_type", type=str, default="mse", help="consistency_type")
parser.add_argument("--consistency", type=float, default=1.0, help="consistency")
parser.add_argument("--consistency_rampup", type=float, default=200.0, help="consistency_rampup")
parser.add_argument('--depth', type=int, default=28)
parser.add_argument('--widen_factor', type=int, default=2)
parser.add_argument('--leaky_slope', type=float, default=0.1)
parser.add_argument('--bn_momentum', type=float, default=0.1)
parser.add_argument('--dropout', type=float, default=0.0)
parser.add_argument('--cutmix_prob', default=1.0, type=float)
parser.add_argument('--LB', default=0.01, type=float)
parser.add_argument('--increase', default=1.0005, type=float)
parser.add_argument('--queue_len', default=10, type=int)
# additive flags of this build
parser.add_argument('--synthetic', type=int, default=1, help='seeded synthetic batches (SURVEY.md 8d)')
parser.add_argument('--amp_dtype', default='fp16', choices=['fp16', 'bf16'],
                    help="storage / matrix-core type under --amp 1: 'fp16' = the reference's torch.cuda.amp autocast + GradScaler "
                         "(train.py:30,551-552,842-845: IEEE half, dynamic loss scale on the device), 'bf16' = bfloat16, no loss scale")
parser.add_argument('--backend_dtype', default='', choices=['', 'f32', 'f32x3', 'bf16', 'f16'],
                    help="explicit override of what --amp / --amp_dtype select ('' = follow them; --amp 0 = f32x3 for the U-Net, f32 for DeepLabV2: the exact paths)")
parser.add_argument('--fft', default='device', choices=['host', 'device'])
parser.add_argument('--data_root', type=str, default='../../data')
parser.add_argument('--log_every', type=int, default=50)
parser.add_argument('--synthetic_pool', type=int, default=8, help='distinct synthetic batches kept resident and cycled (0: a fresh batch every step)')
parser.add_argument('--backbone', default='resnet101', choices=['resnet50', 'resnet101'],
                    help='--model deeplabv2: the dilated ResNet of networks/deeplabv2.py (BASELINE.json configs[4])')
parser.add_argument('--image_size', type=int, default=0, help='patch extent override (0: the dataset default; configs[4] runs BUSI at 512)')


def compute_dtype(args):
    """--amp / --amp_dtype / --backend_dtype -> the library's dtype name.  The reference's default (--amp 1) trains under fp16
    autocast with a GradScaler; --amp 0 is its fp32 path (train.py:551-552,842-847)."""
    if args.backend_dtype:
        return args.backend_dtype
    if args.amp:
        return {"fp16": "f16", "bf16": "bf16"}[args.amp_dtype]
    # --amp 0, the reference's fp32 branch (train.py:846-848): f32 tensors throughout.  For the U-Net the convolutions' products
    # run as three-term bf16 splits on the matrix cores (f32x3: the reference's logits to 3e-6 rel-L2, arg-max flips only at
    # margins < 3e-6 -- the same bars as `f32`, at ~3x its rate); the f32-MFMA path stays reachable as --backend_dtype f32 and is
    # what DeepLabV2 (no x3 kernels for its 1x1 / dilated / strided convolutions) takes
    return "f32x3" if args.model == "unet" else "f32"


def make_loaders(args, C, H, dev=None):
    """-> iterator of (lb_x_w, lb_y, ulb_x_w, ulb_x_s, ulb_y) CPU tensors."""
    if not args.synthetic:
        raise SystemExit("the PIL/scipy dataset pipeline of the reference (dataloaders/) is outside this build's scope; "
                         "run with --synthetic 1 or plug a loader in here")
    from ustrun import synthetic
    from ustrun.ddp import env_world
    rank = env_world()[0]
    step = 0
    if args.synthetic_pool > 0:       # a pool of distinct seeded batches, generated once and cycled: the host generator
        pool = [synthetic.batch(args.dataset, args.label_bs, C, H, args.seed + 100003 * rank + i)   # (0.2 s per batch)
                for i in range(args.synthetic_pool)]                                                # would cap the step rate
        if dev is not None:
            pool = [[t.to(dev) for t in b] for b in pool]
        while True:
            yield pool[step % len(pool)]
            step += 1
    while True:
        yield synthetic.batch(args.dataset, args.label_bs, C, H, args.seed + 100003 * rank + step)
        step += 1


def train(args, snapshot_path):
    from networks.unet_model import UNet
    from ustrun import ddp
    from ustrun.trainer import DATASETS, SSLTrainer
    rank, local, world = ddp.env_world()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ddp.init(device=dev)
    C, H, K, _, _, max_it = DATASETS[args.dataset]
    H = args.image_size or H
    args.max_iterations = max_it                       # per-dataset schedule of the reference (train.py:412,423,434)

    def create_model(ema=False):
        if args.model == 'deeplabv2':
            # the reference defines networks/deeplabv2.py but its create_model never builds it (train.py:496-503; SURVEY.md 8f
            # row 4): same SSL step around it here.  Seeded init unless ../../checkpoints/pretrained/<arch>.pth exists
            # (base.py:12); a 1-channel dataset feeds three equal channels.
            from networks.deeplabv2 import DeepLabV2
            ck = "../../checkpoints/pretrained/%s.pth" % args.backbone
            model = DeepLabV2(args.backbone, K, pretrained=os.path.exists(ck), dtype=compute_dtype(args))
        elif args.model != 'unet':
            raise SystemExit("--model is 'unet' (the reference's path, train.py:496-503) or 'deeplabv2' (networks/deeplabv2.py)")
        else:
            model = UNet(n_channels=C, n_classes=K, dtype=compute_dtype(args))
        if ema:
            for p in model.parameters():
                p.detach_()
        return model.to(dev)

    model, ema_model = create_model(), create_model(ema=True)
    # (the reference trains under fp16 autocast + GradScaler by default, train.py:54,551-552: so does a plain `python train.py`
    # here; --amp 0 or --backend_dtype f32 selects the exact f32 path that was this script's default before round 4)
    logging.info("compute dtype %s (--amp %d, --amp_dtype %s, --backend_dtype '%s'); dynamic loss scale: %s", compute_dtype(args),
                 args.amp, args.amp_dtype, args.backend_dtype, "on (GradScaler semantics, initial scale 65536)"
                 if compute_dtype(args) == "f16" else "off")
    trainer = SSLTrainer(args.dataset, model, ema_model, base_lr=args.base_lr, max_iterations=args.max_iterations,
                         threshold=args.threshold, ema_decay=args.ema_decay, consistency=args.consistency,
                         consistency_rampup=args.consistency_rampup, cutmix_prob=args.cutmix_prob, LB=args.LB,
                         increase=args.increase, queue_len=args.queue_len, num_eval_iter=args.num_eval_iter,
                         grad_allreduce=ddp.make_grad_allreduce(world), world_size=world, fft=args.fft, patch_size=H)
    loader = make_loaders(args, C, H, dev)
    from ustrun import synthetic
    from ustrun.evaluate import validate
    test_loaders = synthetic.test_loaders(args.dataset, min(args.domain_num, 2), 4, args.test_bs, C, H, args.seed + 17)
    best = {"avg": 0.0, "iter": 0, "stu_avg": 0.0, "stu_iter": 0}
    start_epoch = 0
    if args.load:      # train.py:542-548: resume from the run's own checkpoint.pth (--load_path is parsed but unused there too)
        from ustrun import engine
        from utils import util
        path_str = '../model/{}/{}/checkpoint.pth'.format(args.dataset, args.save_name)
        (start_epoch, _, _, _, best["avg"], best["iter"], best["stu_avg"], best["stu_iter"]) = util.load_osmancheckpoint(
            path_str, ema_model, model, trainer.optimizer)
        trainer.iter_num = start_epoch * args.num_eval_iter
        engine.invalidate_packed(model)
        engine.invalidate_packed(ema_model)
        logging.info('Models restored from epoch {}'.format(start_epoch))
    max_epoch = args.max_iterations // args.num_eval_iter
    logging.info("%d iterations per epoch, %d epochs", args.num_eval_iter, max_epoch)
    t0 = time.time()
    for epoch in range(start_epoch, max_epoch):
        for i in range(args.num_eval_iter):
            batch = [t.to(dev, non_blocking=True) for t in next(loader)]
            trainer.step(*batch, epoch_start=(i == 0))
            if rank == 0 and trainer.iter_num % args.log_every == 0:
                s = trainer.scalars()
                ips = (args.label_bs + args.unlabel_bs) * world * args.log_every / (time.time() - t0)
                t0 = time.time()
                logging.info("iteration %d: loss:%.4f sup:%.4f ul:%.4f lu:%.4f s:%.4f cons_w:%.4f mask:%.4f ulb_dice:%s  %.1f img/s",
                             trainer.iter_num, s["loss"], s["sup"], s["ul"], s["lu"], s["s"], s["w"], s["mask_ratio"],
                             ["%.4f" % v for v in s["ulb_dice"]], ips)
        # epoch end (train.py:913-957): validate teacher then student, keep the student with the best mean Dice
        # (`unet_avg_dice_best_model.pth`, the file test.py loads), write the checkpoint
        if rank == 0:
            logging.info('test ema model')
            val_dice, _ = validate(args.dataset, ema_model, test_loaders, epoch + 1)
            if sum(val_dice) / len(val_dice) > best["avg"]:
                best["avg"], best["iter"] = sum(val_dice) / len(val_dice), trainer.iter_num
            logging.info('val_best_avg_dice: %f at %d iter', best["avg"], best["iter"])
            logging.info('test stu model')
            stu_dice, _ = validate(args.dataset, model, test_loaders, epoch + 1)
            if sum(stu_dice) / len(stu_dice) > best["stu_avg"]:
                best["stu_avg"], best["stu_iter"] = sum(stu_dice) / len(stu_dice), trainer.iter_num
                save_best = os.path.join(snapshot_path, "{}_avg_dice_best_model.pth".format(args.model))
                logging.info('save cur best avg model to {}'.format(save_best))
                torch.save(model.state_dict(), save_best)
            logging.info('val_best_avg_dice: %f at %d iter', best["stu_avg"], best["stu_iter"])
            from utils import util
            util.save_osmancheckpoint(epoch + 1, ema_model, model, trainer.optimizer, best["avg"], best["iter"], best["stu_avg"],
                                      best["stu_iter"], os.path.join(snapshot_path, "checkpoint.pth"))
            logging.info('save checkpoint to {}'.format(os.path.join(snapshot_path, "checkpoint.pth")))
        if world > 1:      # the other ranks wait on a host-side gloo group with a long timeout while rank 0 validates: a barrier
            from ustrun.ddp import wait_for_rank0      # on the RCCL group would sit under the same watchdog as the all-reduce
            wait_for_rank0()


if __name__ == "__main__":
    args = parser.parse_args()
    snapshot_path = "../model/" + args.dataset + "/" + args.save_name + "/"
    if "LOCAL_RANK" not in os.environ:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", args.gpu)
    from ustrun.ddp import env_world
    rank = env_world()[0]
    random.seed(args.seed + rank)                      # CutMix p-draws and FFT-mix ratios: a stream per rank, like the boxes
    np.random.seed(args.seed + rank)
    torch.manual_seed(args.seed)                       # same initial weights on every rank
    if rank == 0:
        if os.path.exists(snapshot_path) and not args.overwrite:
            raise Exception('file {} is exist!'.format(snapshot_path))
        os.makedirs(snapshot_path, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format='[%(asctime)s.%(msecs)03d] %(message)s', datefmt='%H:%M:%S',
                        handlers=[logging.StreamHandler(sys.stdout)])
    logging.info(str(args))
    train(args, snapshot_path)
