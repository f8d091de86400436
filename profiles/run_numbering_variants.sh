#!/bin/bash
# time and level-0 SpMV HBM traffic of the internal numbering's variants (cluster size, segment length) at 512^3:
#   gpurun -- bash profiles/run_numbering_variants.sh      -> gpurun_out/numbering_variants.txt
cd "${GRAFT_REPO_ROOT:-.}"; REPO=$(pwd); export TMPDIR=/tmp
out=$REPO/gpurun_out/numbering_variants.txt; : > $out
run() {  # label, env...
  label=$1; shift
  env "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu --no-general --sideline-non-galerkin 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label: ms/solve %.1f iters %d spmv_l0 %.3f ms relax_l0 %.3f ms setup %.1f s' % (d['ms_per_step'], d['iterations_per_solve'], d['roofline']['avg_ms'], d['roofline_relax']['avg_ms'], d['setup_s']))" >> $out
}
run "default (cells of 512, segment from the matrix)" X=1
run "cells of 1024" MI_HYPRE_LOCALITY_CLUSTER=1024
run "cells of 256" MI_HYPRE_LOCALITY_CLUSTER=256
run "segments of 2^22 rows" MI_HYPRE_LOCALITY_SEGMENT=22
run "segments of 2^19 rows" MI_HYPRE_LOCALITY_SEGMENT=19
cat $out
