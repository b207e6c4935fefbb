# one GPU call: rocprofv3 --kernel-trace --stats of the other BASELINE.json shapes on one GPU (configs[2]: prostate 384^2 8+8,
# configs[3]: M&Ms 288^2 8+8, configs[4]: DeepLabV2-ResNet101 512^2 forward + backward at 16, and the f32x3 step at configs[1]'s
# shape) -> gpurun_out/profile_configs/*_kernel_stats.csv; copy what is to be judged into profiles/
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profile_configs
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name, program args...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python3 "$@" > $O/$name.log 2>&1
  find $O/$name -name "*kernel_stats.csv" -exec cp {} $O/${name}_kernel_stats.csv \;
  rm -rf $O/$name
}
run prostate_384_b8 $R/bench.py --dataset prostate --label_bs 8 --unlabel_bs 8 --steps 6 --warmup 2 --no-secondary --no-cpu-baseline
run mnms_288_b8 $R/bench.py --dataset MNMS --label_bs 8 --unlabel_bs 8 --steps 6 --warmup 2 --no-secondary --no-cpu-baseline
run deeplab_r101_512_n16_fwdbwd $R/tools/bench_deeplab.py --n 16 --hw 512 --backward --reps 3
run fundus_256_b16_f32x3 $R/bench.py --dtype f32x3 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline
ls -la $O
