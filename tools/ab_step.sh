# same-box A/B of whole-step time under ustrun_debug_flags values: tools/ab_step.sh "0 4194304 2097152" [reps]
set -e
mkdir -p gpurun_out/ab
for rep in $(seq 1 ${2:-2}); do
  for f in $1; do
    USTRUN_DEBUG_FLAGS=$f timeout -k 10 150 python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-profile > gpurun_out/ab/f${f}_r${rep}.json 2> gpurun_out/ab/f${f}_r${rep}.err
    python -c "
import json,sys; j=json.load(open('gpurun_out/ab/f${f}_r${rep}.json')); print('flags', '$f', 'rep', $rep, j['value'], 'img/s', j['ms_per_step'], 'ms')"
  done
done
