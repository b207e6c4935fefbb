set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
USTRUN_TEST_FULL=1 python -m pytest tests -m gpu -q --durations=12 > $O/r6_full4.log 2>&1; echo rc=$? >> $O/r6_full4.log
tail -18 $O/r6_full4.log
