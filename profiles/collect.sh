#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel-trace statistics and, in separate passes, the two HBM
# counters for the default bench workload (512^3).  Raw output lands in gpurun_out/prof_*; the summaries
# under profiles/ are made afterwards by profiles/summarize.py.
#   gpurun --timeout 1100 -- 'bash profiles/collect.sh stats'      (or: pmc)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
export TMPDIR=/tmp
mode=${1:-stats}
cd /tmp
if [ "$mode" = stats ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_stats" -- \
    python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu --no-general > "$REPO/gpurun_out/prof_stats.log" 2>&1
  python3 "$REPO/profiles/gap_analysis.py" "$(ls -t "$REPO"/gpurun_out/prof_stats/*/*_kernel_trace.csv | head -1)" > "$REPO/gpurun_out/gaps_512.txt" 2>&1 || true
elif [ "$mode" = setup ]; then
  # host-vs-device split of the setup (profiles/setup_split.py): kernel + copy + roctx marker traces of one setup and one solve
  MI_HYPRE_SETUP_TIMING=1 MI_BENCH_VERBOSE=1 rocprofv3 --kernel-trace --memory-copy-trace --marker-trace --output-format csv \
    -d "$REPO/gpurun_out/prof_setup" -- \
    python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu --no-general --sideline-non-galerkin 0 > "$REPO/gpurun_out/prof_setup.log" 2>&1
  python3 "$REPO/profiles/setup_split.py" "$REPO/gpurun_out/prof_setup" > "$REPO/gpurun_out/setup_split_512.txt" 2>&1 || true
else
  # counter passes.  Round 3 ran them WITHOUT the collapsed coarse tail: tabulating it means ~50 000 tiny dispatches at
  # Setup, and rocprofv3's counter collection crashed in that loop (segmentation fault).  Round 4: the first
  # pass is tried with the default tail; its outcome is recorded in gpurun_out/pmc_tail_attempt.txt (exit code, the tail of
  # the log) and the passes that are kept run without the tail only if that attempt failed -- the level-0 kernels whose
  # traffic is measured are the same either way.
  set +e
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$REPO/gpurun_out/prof_fetch" -- \
    python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu ${PMC_BENCH_ARGS:---no-general} > "$REPO/gpurun_out/prof_fetch.log" 2>&1
  rc=$?
  { echo "rocprofv3 --pmc FETCH_SIZE with the default collapsed tail: exit code $rc"; grep -v "^[EW]2026" "$REPO/gpurun_out/prof_fetch.log" | tail -15; } > "$REPO/gpurun_out/pmc_tail_attempt.txt"
  set -e
  if [ $rc -ne 0 ]; then
    export MI_HYPRE_DENSE_TAIL_ROWS=0
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$REPO/gpurun_out/prof_fetch" -- \
      python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu ${PMC_BENCH_ARGS:---no-general} > "$REPO/gpurun_out/prof_fetch.log" 2>&1
  fi
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$REPO/gpurun_out/prof_write" -- \
    python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu ${PMC_BENCH_ARGS:---no-general} > "$REPO/gpurun_out/prof_write.log" 2>&1
  # the same two passes for a general operator (value dictionary off: 8-byte values in the level-0 stream)
  export MI_HYPRE_VALUE_DICT=0
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$REPO/gpurun_out/prof_fetch_g" -- \
    python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu --no-general > "$REPO/gpurun_out/prof_fetch_g.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$REPO/gpurun_out/prof_write_g" -- \
    python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu --no-general > "$REPO/gpurun_out/prof_write_g.log" 2>&1
  unset MI_HYPRE_VALUE_DICT
fi
find "$REPO/gpurun_out" -name "*.csv" -size +60M -delete   # keep the merge-back small
ls -la "$REPO"/gpurun_out/prof_*/*/ 2>/dev/null | tail -20
