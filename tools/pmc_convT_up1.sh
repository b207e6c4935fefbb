# GPU box: SQ counters of the ConvTranspose kernels on up1.up's shape (1024 -> 512 at 16 x 16, N = 64) -> gpurun_out/pmc_convT_up1.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_convT_up1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/pmc_convT.py ${1:-1024} ${2:-512} ${3:-16} ${4:-64} 0 > $O/p$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $O convT > $R/gpurun_out/pmc_convT_up1.txt 2>&1 || true
cat $R/gpurun_out/pmc_convT_up1.txt | head -80
