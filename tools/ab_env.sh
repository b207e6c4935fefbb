# same-box A/B of whole-step time under an environment variable: tools/ab_env.sh VAR "v1 v2" [reps]
set -e
mkdir -p gpurun_out/ab
for rep in $(seq 1 ${3:-2}); do
  for v in $2; do
    env $1=$v timeout -k 10 150 python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-profile > gpurun_out/ab/e${v}_r${rep}.json 2> gpurun_out/ab/e${v}_r${rep}.err
    python -c "
import json,sys; j=json.load(open('gpurun_out/ab/e${v}_r${rep}.json')); print('$1', '$v', 'rep', $rep, j['value'], 'img/s', j['ms_per_step'], 'ms')"
  done
done
