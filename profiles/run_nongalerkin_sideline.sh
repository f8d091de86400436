#!/bin/bash
# Non-Galerkin coarse operators (src/HypreSystem.cpp:161-176) as a side-line of the benchmark configuration:
#   gpurun -- bash profiles/run_nongalerkin_sideline.sh [n]   -> tolerance, levels, operator complexity, iterations, ms/solve
n=${1:-512}
for tol in 0 0.02 0.05 0.1 0.2; do
  extra=""; [ "$tol" != 0 ] && extra="--amg non_galerkin_tol=$tol"
  python3 bench.py --n $n --steps 3 --warmup 1 --no-cpu --no-general $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('non_galerkin_tol %-5s n %d: levels %2d  operator complexity %.3f  iterations %2d  ms/solve %7.1f  GDOF/s %.3f  rel res %.2e  max|x-1| %.1e  setup %.1f s' % ('$tol', $n, d['amg_levels'], d['operator_complexity'], d['iterations_per_solve'], d['ms_per_step'], d['value'], d['final_rel_residual'], d['max_abs_error_vs_ones'], d['setup_s']))" || exit 1
done
