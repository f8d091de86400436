#!/bin/bash
# A/B timing of an environment switch on ONE box:  bash profiles/ab_env.sh <n> <rounds> VAR=a VAR=b ...
n=$1; rounds=$2; shift 2
for r in $(seq $rounds); do
  for kv in "$@"; do
    env $kv python3 bench.py --grid $n --steps 4 --warmup 1 --no-cpu --no-general --sideline-non-galerkin 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$kv round $r: ms/solve %.1f  iters %d  spmv_l0 %.3f ms  relax_l0 %.3f ms  setup %.1f s  res %.6e' % (d['ms_per_step'], d['iterations_per_solve'], d['roofline']['avg_ms'], d['roofline_relax']['avg_ms'], d['setup_s'], d['final_rel_residual']))"
  done
done
