# same-box A/B of the whole step between this tree and a second checkout (e.g. `git worktree add --detach .ab_old HEAD` + make
# there): tools/ab_tree.sh .ab_old [reps] [extra bench flags]
set -e
mkdir -p gpurun_out/ab
here=$(pwd)
for rep in $(seq 1 ${2:-2}); do
  for t in "$1" .; do
    name=$(basename $(cd $t && pwd))
    (cd $t && timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-profile $3 > $here/gpurun_out/ab/t_${name}_r${rep}.json 2> $here/gpurun_out/ab/t_${name}_r${rep}.err)
    python -c "
import json; j=json.load(open('gpurun_out/ab/t_${name}_r${rep}.json')); print('tree', '$name', 'rep', $rep, j['value'], 'img/s', j['ms_per_step'], 'ms')"
  done
done
