set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_deeplab_tiles.py tests/test_gpu_deeplab_bwd.py tests/test_gpu_deeplab.py tests/test_gpu_production_tiles.py -m gpu -q -x --durations=8 > $O/r6_t3.log 2>&1; echo rc=$? >> $O/r6_t3.log
tail -14 $O/r6_t3.log
echo "== deeplab fwd+bwd: all new paths" > $O/r6_dl_ab2.log
python tools/bench_deeplab.py --backward --reps 5 >> $O/r6_dl_ab2.log 2>&1
echo "== dilated wgrad tap by tap" >> $O/r6_dl_ab2.log
USTRUN_DEEPLAB_WGRAD_TAPS=1 python tools/bench_deeplab.py --backward --reps 5 >> $O/r6_dl_ab2.log 2>&1
echo "== fused epilogues off (flags2=8)" >> $O/r6_dl_ab2.log
USTRUN_DEBUG_FLAGS2=8 python tools/bench_deeplab.py --backward --reps 5 >> $O/r6_dl_ab2.log 2>&1
echo "== ssl step" >> $O/r6_dl_ab2.log
python tools/bench_deeplab.py --ssl --reps 3 >> $O/r6_dl_ab2.log 2>&1
grep -E "==|ms|images" $O/r6_dl_ab2.log | cut -c1-420
