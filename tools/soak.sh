# three ~150 s training runs of ust-run_amd/train.py on synthetic data with validation + checkpoint every 60 iterations:
# bf16 (--amp_dtype bf16) and the reference's default fp16 + loss scale (--amp 1 --amp_dtype fp16); excerpts -> gpurun_out/soak/
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/soak; mkdir -p $O
cd $R/ust-run_amd
for dt in bf16 fp16; do
  set +e
  timeout -k 10 150 python train.py --dataset fundus --synthetic 1 --amp 1 --amp_dtype $dt --label_bs 16 --unlabel_bs 16 --num_eval_iter 60 --log_every 20 --save_name r4soak_$dt --overwrite > $O/$dt.full.log 2>&1
  echo "rc=$?" >> $O/$dt.full.log
  set -e
  (head -12 $O/$dt.full.log | cut -c1-400; echo ...; grep -c "iteration" $O/$dt.full.log; grep -i "nan\|inf \|skipped\|scale" $O/$dt.full.log | tail -5; tail -14 $O/$dt.full.log | cut -c1-300) > $O/$dt.log
  rm -f $O/$dt.full.log
done
# configs[3]'s shape through train_mnms.py (8 + 8 images of 288^2: the 72 / 36 / 18-pixel levels run the halo kernel's linear tiles)
set +e
timeout -k 10 150 python train_mnms.py --synthetic 1 --amp 1 --amp_dtype bf16 --label_bs 8 --unlabel_bs 8 --num_eval_iter 60 --log_every 20 --save_name r5soak_mnms --overwrite > $O/mnms.full.log 2>&1
echo "rc=$?" >> $O/mnms.full.log
set -e
(head -12 $O/mnms.full.log | cut -c1-400; echo ...; grep -c "iteration" $O/mnms.full.log; grep -i "nan\|inf \|skipped\|scale" $O/mnms.full.log | tail -5; tail -14 $O/mnms.full.log | cut -c1-300) > $O/mnms.log
rm -f $O/mnms.full.log
tail -6 $O/bf16.log; tail -6 $O/fp16.log; tail -6 $O/mnms.log
