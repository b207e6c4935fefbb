# one GPU call: bench.py (full line), rocprofv3 --kernel-trace --stats, the two --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs), tools/pmc_hbm.py
# -> gpurun_out/profile_round/{bench.json,kernel_stats.csv,pmc_hbm.csv,traffic_bf16.json}; copy what is to be judged into profiles/
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profile_round
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --dump-full $O/bench_full.json > $O/bench.json 2> $O/bench.err
wc -c $O/bench.json
tail -c 600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 8 --warmup 2 --no-secondary --no-cpu-baseline > $O/kt.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pf -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-secondary > $O/pf.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pw -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-secondary > $O/pw.log 2>&1
cd $R
python tools/pmc_hbm.py $O/pf $O/pw $O/pmc_hbm.csv $O/traffic_bf16.json 2 > $O/pmc.log
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
# keep the merge-back small
find $O/kt $O/pf $O/pw -name "*.csv" -size +8M -delete
du -sh $O
