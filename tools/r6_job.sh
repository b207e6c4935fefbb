set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -q --durations=8 > $O/r6_full5.log 2>&1; echo rc=$? >> $O/r6_full5.log
tail -14 $O/r6_full5.log
python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | cut -c1-900
