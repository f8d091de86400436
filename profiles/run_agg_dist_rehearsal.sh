#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): aggressive coarsening (agg_num_levels 1, multipass interpolation) through the
# DISTRIBUTED setup on 2 ranks sharing the GPU at 256^3, next to the single-rank run of the same settings: the
# hierarchies are partition-independent, so the level counts agree and the iteration counts (the hybrid Gauss-Seidel
# smoother treats the other rank's values as old ones, so the residuals agree to a few digits only; the reported operator
# complexity is rank 0's share, equal to the global figure to ~1e-5) -- with the
# internal locality numbering off (it renumbers every rank's rows by itself, rows with halo entries last, which
# changes the PMIS random numbers of the points: with it on, 1 rank and 2 ranks build different, equally good hierarchies).
set -e
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
N=${1:-256}
export MI_HYPRE_LOCALITY_ORDER=${MI_HYPRE_LOCALITY_ORDER:-0}
python3 bench.py --grid $N --steps 2 --warmup 1 --no-cpu --no-general --amg agg_num_levels=1 > gpurun_out/agg_1rank.log 2>&1
MI_BENCH_SHARED_GPU=1 MI_HYPRE_SETUP_TIMING=1 MI_BENCH_VERBOSE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 \
  bench.py --gpus 2 --grid $N --steps 2 --warmup 1 --no-cpu --no-ipc-sideline --amg agg_num_levels=1 > gpurun_out/agg_2ranks.log 2>&1
python3 - <<'PY'
import json
def line(p):
    return json.loads([l for l in open(p) if l.startswith("{")][-1])
a, b = line("gpurun_out/agg_1rank.log"), line("gpurun_out/agg_2ranks.log")
for k in ("iterations_per_solve", "final_rel_residual", "amg_levels", "operator_complexity", "ms_per_step", "setup_s"):
    print(f"{k:24s} 1 rank {a.get(k)}   2 ranks (one GPU) {b.get(k)}")
assert abs(a["iterations_per_solve"] - b["iterations_per_solve"]) <= 1 and a["amg_levels"] == b["amg_levels"]
assert abs(a["operator_complexity"] - b["operator_complexity"]) < 1e-3 * a["operator_complexity"]
assert a["final_rel_residual"] <= 1e-8 and b["final_rel_residual"] <= 1e-8
print("agg dist rehearsal ok")
PY
grep "distributed setup" gpurun_out/agg_2ranks.log | head -40
