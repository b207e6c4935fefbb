set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
tail -3 gpurun_out/profile_round.log | cut -c1-300
bash tools/profile_configs.sh > gpurun_out/profile_configs.log 2>&1
tail -8 gpurun_out/profile_configs.log
