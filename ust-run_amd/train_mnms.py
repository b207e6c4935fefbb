#!/usr/bin/env python3
"""train_mnms.py -- MI355X build of the reference's M&Ms training driver (train_mnms.py:38-77).

Same flags and defaults as the reference's file (`--dataset MNMS`, `--domain_num 4`, `--lb_num 20`, its `--load_path`
default, no `--lb_ratio`); everything else -- the step, the epoch-end validation, the checkpoints, the additive flags --
is train.py's: the reference's two drivers differ only in the label decoding (train_mnms.py:549-552, three 0/255 planes
-> classes 1..3), the 4-class softmax head and the 288 x 288 patch, all of which ustrun.trainer.DATASETS["MNMS"] carries.

    python train_mnms.py --save_name run0 --synthetic 1
"""
import logging
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import train as T  # noqa: E402

parser = T.parser
for act in parser._actions:
    if act.dest == "dataset":
        act.choices, act.default = ["MNMS"], "MNMS"
parser.set_defaults(dataset="MNMS", domain_num=4, lb_num=20, load_path="../model/lb1_ratio0.2/iter_6000.pth")


if __name__ == "__main__":
    args = parser.parse_args()
    snapshot_path = "../model/" + args.dataset + "/" + args.save_name + "/"
    if "LOCAL_RANK" not in os.environ:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", args.gpu)
    from ustrun.ddp import env_world
    rank = env_world()[0]
    random.seed(args.seed + rank)
    np.random.seed(args.seed + rank)
    torch.manual_seed(args.seed)
    if rank == 0:
        if os.path.exists(snapshot_path) and not args.overwrite:
            raise Exception('file {} is exist!'.format(snapshot_path))
        os.makedirs(snapshot_path, exist_ok=True)
    logging.basicConfig(level=logging.INFO, format='[%(asctime)s.%(msecs)03d] %(message)s', datefmt='%H:%M:%S',
                        handlers=[logging.StreamHandler(sys.stdout)])
    logging.info(str(args))
    T.train(args, snapshot_path)
