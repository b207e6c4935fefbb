# round 6: ~100 s training runs of ust-run_amd/train.py on synthetic data (validation + checkpoint every 60 iterations) through this
# round's changes: fundus 16 + 16 bf16 (the flat plan of the 64 -> 64 kernel at 81 images), --amp 0 (= f32x3 since this round),
# M&Ms 8 + 8 (flat plan at 41 / 32 images of 288^2), DeepLabV2-ResNet50 on BUSI (fused join / BatchNorm sums, space-to-batch weight
# gradients); excerpts -> gpurun_out/soak6/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/soak6; mkdir -p $O
cd $R/ust-run_amd
run() { # name, seconds, script, args...
  name=$1; secs=$2; shift 2
  timeout -k 10 $secs python "$@" > $O/$name.full.log 2>&1
  echo "rc=$?" >> $O/$name.full.log
  (head -8 $O/$name.full.log | cut -c1-300; echo ...; grep -c "iteration" $O/$name.full.log; grep -i "nan\|inf \|skipped\|Traceback\|Error" $O/$name.full.log | tail -5; tail -8 $O/$name.full.log | cut -c1-260) > $O/$name.log
  rm -f $O/$name.full.log
  tail -4 $O/$name.log
}
run fundus_bf16 100 train.py --dataset fundus --synthetic 1 --amp 1 --amp_dtype bf16 --label_bs 16 --unlabel_bs 16 --num_eval_iter 60 --log_every 20 --save_name r6soak_bf16 --overwrite
run fundus_amp0 100 train.py --dataset fundus --synthetic 1 --amp 0 --label_bs 16 --unlabel_bs 16 --num_eval_iter 60 --log_every 20 --save_name r6soak_amp0 --overwrite
run mnms_bf16 100 train_mnms.py --synthetic 1 --amp 1 --amp_dtype bf16 --label_bs 8 --unlabel_bs 8 --num_eval_iter 60 --log_every 20 --save_name r6soak_mnms --overwrite
run deeplab_r50_busi 120 train.py --dataset BUSI --model deeplabv2 --backbone resnet50 --image_size 128 --label_bs 4 --unlabel_bs 4 --num_eval_iter 10 --log_every 5 --backend_dtype bf16 --save_name r6soak_dl --overwrite
