set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_deeplab_tiles.py tests/test_gpu_deeplab_bwd.py tests/test_gpu_deeplab.py -m gpu -q -x --durations=5 > $O/r6_t5.log 2>&1; echo rc=$? >> $O/r6_t5.log
tail -12 $O/r6_t5.log
python tools/bench_deeplab.py --backward --reps 5 2>&1 | grep -v amdgpu.ids | cut -c1-700
USTRUN_DEBUG_FLAGS2=8 python tools/bench_deeplab.py --backward --reps 5 2>&1 | grep -v amdgpu.ids | cut -c1-300
