cd $GRAFT_REPO_ROOT
for v in 16 0 16 0; do echo "== USTRUN_DEBUG_FLAGS2=$v"; USTRUN_DEBUG_FLAGS2=$v python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | cut -c1-200; done
for v in 16 0; do echo "== prostate USTRUN_DEBUG_FLAGS2=$v"; USTRUN_DEBUG_FLAGS2=$v python bench.py --dataset prostate --label_bs 8 --unlabel_bs 8 --steps 30 --warmup 5 --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | cut -c1-200; done
