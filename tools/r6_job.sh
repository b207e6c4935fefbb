set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -q --durations=12 > $O/r6_full3.log 2>&1; echo rc=$? >> $O/r6_full3.log
tail -22 $O/r6_full3.log
