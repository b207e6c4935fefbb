#!/usr/bin/env python3
"""test.py -- MI355X build of the reference's evaluation driver (test.py:19-250).

Same command line (flag names and defaults of test.py:19-32; `--dataset` also accepts BUSI, which the reference
trains but does not list here); additive flags: --synthetic, --test_batches, --backend_dtype, --seed.  Loads
`../model/<dataset>/<save_name>/unet_avg_dice_best_model.pth` (a plain state_dict with the reference's keys, test.py:241)
and prints the per-domain and mean Dice of ustrun.evaluate.validate.  The dataset classes and the medpy metrics
(jc / hd95 / asd) are outside this build: batches come from the seeded synthetic generator unless a loader is plugged
into `make_loaders`.
"""
import argparse
import logging
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

parser = argparse.ArgumentParser()
parser.add_argument('--dataset', type=str, default='prostate', choices=['fundus', 'prostate', 'MNMS', 'BUSI'])
parser.add_argument("--save_name", type=str, default="debug", help="experiment_name")
parser.add_argument("--overwrite", action='store_true')
parser.add_argument("--model", type=str, default="unet", help="model_name")
parser.add_argument("--gpu", type=str, default='0')
parser.add_argument('--eval', type=bool, default=True)
parser.add_argument("--test_bs", type=int, default=1)
parser.add_argument('--domain_num', type=int, default=6)
parser.add_argument('--lb_domain', type=int, default=1)
parser.add_argument('--save_img', action='store_true')
# additive flags of this build
parser.add_argument('--synthetic', type=int, default=1)
parser.add_argument('--test_batches', type=int, default=8, help='synthetic batches per domain')
parser.add_argument('--backend_dtype', default='f32', choices=['f32', 'f32x3', 'bf16', 'f16'])
parser.add_argument('--seed', type=int, default=1337)
parser.add_argument('--load_path', type=str, default='', help='state_dict file (default: the reference\'s path)')
parser.add_argument('--backbone', default='resnet101', choices=['resnet50', 'resnet101'], help='--model deeplabv2')
parser.add_argument('--image_size', type=int, default=0, help='patch extent override (0: the dataset default)')

DOMAINS = {"fundus": 4, "prostate": 6, "MNMS": 4, "BUSI": 1}     # test.py:209-223


def make_loaders(args, C, H):
    if not args.synthetic:
        raise SystemExit("the dataset classes of the reference (dataloaders/) are outside this build's scope; "
                         "run with --synthetic 1 or plug a loader in here")
    from ustrun import synthetic
    return synthetic.test_loaders(args.dataset, args.domain_num, args.test_batches, args.test_bs, C, H, args.seed)


def main(args):
    from networks.unet_model import UNet
    from ustrun.evaluate import validate
    from ustrun.trainer import DATASETS
    C, H, K = DATASETS[args.dataset][:3]
    H = args.image_size or H
    args.domain_num = min(args.domain_num, DOMAINS[args.dataset])
    if args.model not in ('unet', 'deeplabv2'):
        raise SystemExit("--model is 'unet' (the reference's path) or 'deeplabv2' (train.py --model deeplabv2 of this build)")
    if args.save_img:
        raise SystemExit("--save_img (cv2 contour drawing, util.py:300-360) is outside this build's scope")
    if args.model == 'deeplabv2':
        from networks.deeplabv2 import DeepLabV2
        model = DeepLabV2(args.backbone, K, pretrained=False, dtype=args.backend_dtype).cuda()
    else:
        model = UNet(n_channels=C, n_classes=K, dtype=args.backend_dtype).cuda()
    path = args.load_path or '../model/{}/{}/{}_avg_dice_best_model.pth'.format(args.dataset, args.save_name, args.model)
    model.load_state_dict(torch.load(path, map_location="cuda"))
    return validate(args.dataset, model, make_loaders(args, C, H), epoch=args.lb_domain)


if __name__ == "__main__":
    args = parser.parse_args()
    os.environ.setdefault("HIP_VISIBLE_DEVICES", args.gpu)
    logging.basicConfig(level=logging.INFO, format='[%(asctime)s.%(msecs)03d] %(message)s', datefmt='%H:%M:%S',
                        handlers=[logging.StreamHandler(sys.stdout)])
    logging.info(str(args))
    main(args)
