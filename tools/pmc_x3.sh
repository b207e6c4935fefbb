# GPU box: SQ counters of the f32x3 kernels on one 3x3 layer (default 512 -> 512 at 32 x 32, N = 64) -> gpurun_out/pmc_x3.txt
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_x3
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY" "SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/pmc_x3.py ${1:-512} ${2:-512} ${3:-32} ${4:-64} 3 > $O/p$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $O ${5:-wgrad_x3} > $R/gpurun_out/pmc_x3.txt 2>&1 || true
cat $R/gpurun_out/pmc_x3.txt
