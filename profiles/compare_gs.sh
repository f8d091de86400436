#!/bin/bash
# kernel-trace of the 256^3 run for a list of env settings:  bash profiles/compare_gs.sh "MI_HYPRE_GS_OLD=1" "MI_HYPRE_GS_OLD=0" ...
cd "${GRAFT_REPO_ROOT:-.}"
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
i=0
for setting in "$@"; do
  i=$((i+1))
  rm -rf "$REPO/gpurun_out/cmp_$i"
  env $setting MI_CMP_DUMMY=1 true
  ( export $setting; rocprofv3 --kernel-trace --output-format csv -d "$REPO/gpurun_out/cmp_$i" -- python3 "$REPO/bench.py" --grid ${MI_CMP_GRID:-256} --steps 2 --warmup 1 --no-cpu > "$REPO/gpurun_out/cmp_$i.log" 2>&1 )
  echo "$i: $setting"
done
