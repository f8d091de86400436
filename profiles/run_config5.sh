#!/bin/bash
# BASELINE.json config 5 side-line (3-component convection-diffusion stand-in, BiCGSTAB + BoomerAMG):
# one MI355X at 256^3 rows (multivector solve and segregated solves), and a 4-rank shared-GPU REHEARSAL of the weak-
# scaled N > 1 path (gloo callbacks; not a measurement).   -> gpurun_out/config5.txt
out=gpurun_out/config5.txt
: > $out
python3 bench.py --workload convdiff3 --n ${1:-256} --steps 3 --warmup 1 --tol 1e-8 >> $out 2>&1 || exit 1
python3 bench.py --workload convdiff3 --n ${1:-256} --steps 3 --warmup 1 --tol 1e-8 --segregated 1 >> $out 2>&1 || exit 1
MI_BENCH_SHARED_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29641 \
  bench.py --gpus 4 --workload convdiff3 --grid 96 --steps 2 --warmup 1 --tol 1e-8 >> $out 2>&1 || exit 1
python3 - "$out" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        print("%s\n   -> %d GPU(s)%s: %d iterations, %.1f ms per step, %.3f GDOF/s, max |x - x_exact| %.2e, setup %.1f s, %d levels, opcx %.2f" % (
            d["config"]["workload"], d["n_gpus"], " (REHEARSAL)" if d.get("rehearsal") else "", d["iterations_per_solve"],
            d["ms_per_step"], d["value"], d["max_abs_error_vs_exact"], d["setup_s"], d["amg_levels"], d["operator_complexity"]))
PY
