#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the upstream sample's boomeramg_settings (/root/reference/etc/hypre_app.yaml:
# coarsen_type 6 Falgout, interp_type 0, relax_type 6, num_sweeps 2, strong_threshold 0.57) on the generator's 7-point
# operator, one rank against the DISTRIBUTED setup on 2 ranks sharing the GPU.  Falgout is a per-rank algorithm
# (Ruge-Stueben in every rank's interior, CLJP on the boundary): the two hierarchies differ at the slab boundary, the
# iteration counts must stay close.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
N=${1:-160}
AMG="--amg coarsen_type=6 --amg interp_type=0 --amg relax_type=6 --amg num_sweeps=2 --amg strong_threshold=0.57"
python3 bench.py --grid $N --steps 2 --warmup 1 --no-cpu --no-general $AMG > gpurun_out/sample_1rank.log 2>&1
MI_BENCH_SHARED_GPU=1 MI_HYPRE_SETUP_TIMING=1 MI_BENCH_VERBOSE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29657 \
  bench.py --gpus 2 --grid $N --steps 2 --warmup 1 --no-cpu --no-ipc-sideline $AMG > gpurun_out/sample_2ranks.log 2>&1
python3 - <<'PY'
import json
def line(p):
    return json.loads([l for l in open(p) if l.startswith("{")][-1])
a, b = line("gpurun_out/sample_1rank.log"), line("gpurun_out/sample_2ranks.log")
for k in ("iterations_per_solve", "final_rel_residual", "amg_levels", "operator_complexity", "ms_per_step", "setup_s"):
    print(f"{k:24s} 1 rank {a.get(k)}   2 ranks (one GPU) {b.get(k)}")
assert abs(a["iterations_per_solve"] - b["iterations_per_solve"]) <= 3
assert a["final_rel_residual"] <= 1e-8 and b["final_rel_residual"] <= 1e-8
print("sample settings rehearsal ok")
PY
grep "mi_hypre BoomerAMG" gpurun_out/sample_2ranks.log | head -5
grep "distributed setup, pmis\|distributed setup, interp: b" gpurun_out/sample_2ranks.log | head
