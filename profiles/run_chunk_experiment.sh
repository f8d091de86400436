#!/bin/bash
# what does the hybrid-GS chunk size cost in iterations at the benchmark size?
#   bash profiles/run_chunk_experiment.sh [n]      -> chunk, iterations, ms/solve (chunks != 8 run the generic
# lane-per-chunk kernel gs_hybrid_k: only the iteration count is meaningful for them)
n=${1:-512}
mkdir -p gpurun_out
for ch in 8 16 32; do
  MI_HYPRE_GS_CHUNK=$ch MI_HYPRE_SETUP_TIMING=1 timeout -k 10 400 python3 bench.py --n $n --steps 1 --warmup 1 --no-cpu 2> gpurun_out/chunk_$ch.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chunk $ch n $n: iterations %d  ms/solve %.1f  rel_res %.3e  setup_s %.1f  levels %d  opcx %.3f' % (d['iterations_per_solve'], d['ms_per_step'], d['final_rel_residual'], d['setup_s'], d['amg_levels'], d['operator_complexity']))" || exit 1
done
