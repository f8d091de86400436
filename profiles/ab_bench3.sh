#!/bin/bash
# A/B/C timing of library builds on ONE box:  bash profiles/ab_bench3.sh <n> <rounds> <lib> [<lib> ...]
n=$1; rounds=$2; shift 2
for r in $(seq $rounds); do
  for lib in "$@"; do
    MI_HYPRE_LIB=$(realpath $lib) python3 bench.py --grid $n --steps 4 --warmup 1 --no-cpu --no-general --sideline-non-galerkin 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$(basename $lib) round $r: ms/solve %.1f  iters %d  spmv_l0 %.3f ms  relax_l0 %.3f ms' % (d['ms_per_step'], d['iterations_per_solve'], d['roofline']['avg_ms'], d['roofline_relax']['avg_ms']))"
  done
done
