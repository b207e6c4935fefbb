"""GPU box, under rocprofv3 --pmc: a few launches of one ConvTranspose layer (forward / input gradient / weight gradient) under the
debug flags given, so that the counters of exactly those kernels can be read.  The interpreter itself goes behind `--` (no
shebang here on purpose: an `env` hop would be an exec after the profiler's library has initialised the GPU):

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU -d gpurun_out/pmc_convT -- python3 tools/pmc_convT.py [ci co hw n flags]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from bench_layers import convT_layer, l
ci, co, hw, n, flags = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (128, 64, 128, 64, 0))]
lib = l.lib()
lib.ustrun_debug_flags(flags)
print(convT_layer(lib, n, ci, co, hw, hw, 3))
