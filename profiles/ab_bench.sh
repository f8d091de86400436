#!/bin/bash
# A/B timing of two builds of the library on ONE box (box-to-box variation is 5-7 %):
#   bash profiles/ab_bench.sh <libA.so> <libB.so> [n] [rounds]   -> ms_per_step / level-0 SpMV / level-0 relax / Gram-Schmidt per run
A=$1; B=$2; n=${3:-512}; rounds=${4:-2}
for r in $(seq $rounds); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    MI_HYPRE_LIB=$(realpath $lib) python3 bench.py --n $n --steps 4 --warmup 1 --no-cpu --no-general 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d.get('gram_schmidt') or {}
print('$v round $r: ms/solve %.1f  iters %d  spmv_l0 %.3f ms  relax_l0 %.3f ms  gram-schmidt %.1f ms/solve  setup %.1f s' % (d['ms_per_step'], d['iterations_per_solve'], d['roofline']['avg_ms'], d['roofline_relax']['avg_ms'], g.get('ms_per_solve', 0.0), d['setup_s']))"
  done
done
