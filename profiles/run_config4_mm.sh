#!/bin/bash
# config 4 stand-in (BASELINE.json): a MatrixMarket system of ~10 M rows through the driver, on the GPU box:
#   gpurun -- bash profiles/run_config4_mm.sh 216 [var]     (var: variable-coefficient diffusion, no value dictionary)
set -e
cd "${GRAFT_REPO_ROOT:-.}"
N=${1:-216}
D=/tmp/mm_$N
mkdir -p $D
gcc -O2 -o /tmp/gen_mm hypre-mini-app_amd/host/tools/gen_mm.c -lm
/tmp/gen_mm $N $D $2
ls -la $D
cat > $D/in.yaml <<YAML
linear_system:
  type: matrix_market
  matrix_file: $D/mat.mm
  rhs_file: $D/rhs.mm
  sln_file: $D/sln.mm

solver_settings:
  method: gmres
  preconditioner: boomeramg
  tolerance: 1.0e-8
  max_iterations: 100
  kspace: 100
  print_level: 0

boomeramg_settings:
  print_level: 1
  coarsen_type: 8
  relax_type: 8
  relax_order: 1
  num_sweeps: 1
  max_levels: 20
  strong_threshold: 0.57
YAML
./hypre-mini-app_amd/hypre_app $D/in.yaml 2>&1 | grep -v "^   level\|^\t" | tail -30
