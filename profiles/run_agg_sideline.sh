#!/bin/bash
# Side-line of VERDICT r1 item 2: operator complexity and ms per iteration at 256^3 / 512^3 with aggressive
# coarsening on the first level beside the default hierarchy (the default stays the headline).
# usage (GPU box): bash profiles/run_agg_sideline.sh [n]    -> gpurun_out/agg_sideline_<n>.txt
n=${1:-512}
out=gpurun_out/agg_sideline_${n}.txt
: > $out
for amg in "" "--amg agg_num_levels=1" "--amg agg_num_levels=1 --amg agg_pmax_elmts=4" "--amg agg_num_levels=2 --amg agg_pmax_elmts=4"; do
  echo "== bench.py --n $n $amg" >> $out
  MI_HYPRE_SETUP_TIMING=1 python3 bench.py --n $n --steps 3 --warmup 1 --no-cpu $amg >> $out 2>&1 || exit 1
done
python3 - "$out" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print("%-90s iters %3d  ms/solve %8.1f  ms/iter %6.2f  GDOF/s %.3f  levels %2d  opcx %.3f  setup %.1f s" % (
            d["config"]["workload"][-88:], d["iterations_per_solve"], d["ms_per_step"],
            d["ms_per_step"] / d["iterations_per_solve"], d["value"], d["amg_levels"], d["operator_complexity"], d["setup_s"]))
PY
