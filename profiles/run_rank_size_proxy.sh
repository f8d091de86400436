#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the per-rank problem of the 8-GPU strong-scaling point as a one-GPU proxy.
# 512^3 on 8 ranks leaves 16.7 M rows = 256^3 per rank, so the 256^3 single-GPU solve bounds the compute side of that
# point (no halo exchange, no all-reduce latency): t(512^3) / (8 t(256^3)) is the scaling the kernels alone allow.
# Kernel trace + gap analysis of the same run say how much of it is launch-bound.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
export TMPDIR=/tmp
python3 bench.py --grid 256 --steps 10 --warmup 3 --no-cpu --no-general --sideline-non-galerkin 0 > gpurun_out/proxy256_bench.log 2>&1
grep '^{' gpurun_out/proxy256_bench.log | tail -1 > gpurun_out/proxy256_line.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_proxy256" -- \
  python3 "$REPO/bench.py" --grid 256 --steps 3 --warmup 1 --no-cpu --no-general --sideline-non-galerkin 0 > "$REPO/gpurun_out/prof_proxy256.log" 2>&1
python3 "$REPO/profiles/gap_analysis.py" "$(ls -t "$REPO"/gpurun_out/prof_proxy256/*/*_kernel_trace.csv | head -1)" > "$REPO/gpurun_out/gaps_256.txt" 2>&1 || true
cp "$(ls -t "$REPO"/gpurun_out/prof_proxy256/*/*_kernel_stats.csv | head -1)" "$REPO/gpurun_out/proxy256_kernel_stats.csv" || true
find "$REPO/gpurun_out" -name "*.csv" -size +30M -delete
