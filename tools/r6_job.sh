set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt_dl; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dl -- python3 $R/tools/bench_deeplab.py --backward --reps 5 > $O/kt_dl.log 2>&1
find $O/kt_dl -name "*kernel_stats.csv" -exec cp {} $O/r6a_deeplab_kernel_stats.csv \;
find $O/kt_dl -name "*.csv" -size +6M -delete
head -30 $O/r6a_deeplab_kernel_stats.csv | cut -c1-200
cd $R
bash tools/pmc_wgrad_tap.sh "l3.conv1" l3c1_plain 1 | cut -c1-200
