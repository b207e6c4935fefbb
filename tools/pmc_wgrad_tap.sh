# GPU box: SQ counters of the one-tap weight gradient on one DeepLabV2 shape -> gpurun_out/pmc_wgrad_tap_<tag>.txt
#   bash tools/pmc_wgrad_tap.sh "l3.conv1" tag [plain 0|1]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_wgrad_tap_$2
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/bench_wgrad_tap.py --only "$1" --reps 2 --plain ${3:-0} > $O/p$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $O wgrad_tap > $R/gpurun_out/pmc_wgrad_tap_$2.txt 2>&1 || true
cat $R/gpurun_out/pmc_wgrad_tap_$2.txt
