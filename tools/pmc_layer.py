#!/usr/bin/env python3
"""GPU box, under rocprofv3 --pmc: a few launches of one conv3x3 layer shape (forward with BatchNorm + ReLU on load, input gradient)
so that the counters of exactly those kernels can be read.   python3 tools/pmc_layer.py [ci co hw n]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_layers import conv_layer, l
ci, co, hw, n = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (512, 512, 32, 64))]
lib = l.lib()
print(conv_layer(lib, n, ci, co, hw, hw, False, False, 3))
