# GPU box: SQ / cache counters of one 1x1 GEMM launch class of DeepLabV2 -> gpurun_out/pmc_conv1x1_<tag>.txt
#   bash tools/pmc_conv1x1.sh "l3.conv1" "fwd plain" tag
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_conv1x1_$3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/bench_conv1x1.py --only "$1" --ops "$2" --reps 2 > $O/p$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $O convT_bf16 > $R/gpurun_out/pmc_conv1x1_$3.txt 2>&1 || true
cat $R/gpurun_out/pmc_conv1x1_$3.txt
