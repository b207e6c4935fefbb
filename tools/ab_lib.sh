# same-box A/B of two builds of the library: tools/ab_lib.sh "libA.so libB.so" [reps]   (paths relative to ust-run_amd/ustrun/)
set -e
mkdir -p gpurun_out/ab
for rep in $(seq 1 ${2:-2}); do
  for v in $1; do
    USTRUN_LIB=$PWD/ust-run_amd/ustrun/$v timeout -k 10 150 python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-profile > gpurun_out/ab/l${v}_r${rep}.json 2> gpurun_out/ab/l${v}_r${rep}.err
    python -c "
import json,sys; j=json.load(open('gpurun_out/ab/l${v}_r${rep}.json')); print('lib', '$v', 'rep', $rep, j['value'], 'img/s', j['ms_per_step'], 'ms')"
  done
done
