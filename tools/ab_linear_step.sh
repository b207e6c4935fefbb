# GPU box: the SSL step of configs[2] (prostate 384^2, 8 + 8) and configs[3] (M&Ms 288^2, 8 + 8) with the halo kernel's linear tiles
# (default) and without (USTRUN_DEBUG_FLAGS2=1), alternating, same box -> gpurun_out/ab_linear_step.log
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
: > $O/ab_linear_step.log
for ds in MNMS prostate; do
  for rep in 1 2; do
    for fl in 1 0; do
      USTRUN_DEBUG_FLAGS2=$fl timeout -k 10 200 python3 $R/bench.py --dataset $ds --label_bs 8 --unlabel_bs 8 --steps 20 --warmup 3 --no-secondary \
        --no-cpu-baseline --dump-layers $O/layers_${ds}_$fl.json 2> /dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=b['roofline']
print('$ds flags2 $fl: %.3f ms/step, %.1f images/s, conv class %.3f, wgrad class %.3f' % (b['ms_per_step'], b['value'], r['frac'], r['wgrad']['frac']))" >> $O/ab_linear_step.log
    done
  done
done
cat $O/ab_linear_step.log
