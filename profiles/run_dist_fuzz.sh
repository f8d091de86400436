#!/bin/bash
# exploration: dist_worker over seeded parameter combinations (not part of the test suite)
cd "${GRAFT_REPO_ROOT:-.}"
export MI_HYPRE_REPLICATED_SETUP=0 MI_HYPRE_HOST_THREADS=2 OMP_NUM_THREADS=1 MI_HYPRE_NATURAL_R_MIN_NNZ=0 HSA_ENABLE_IPC_MODE_LEGACY=0
fail=0
for seed in $(seq ${FUZZ_FROM:-20} ${FUZZ_TO:-59}); do
  np=$(( 2 + seed % 3 )); n=$(( 11 + seed % 4 )); seqs=(0 100 300 -1 150); sq=${seqs[$(( seed % 5 ))]}
  timeout -k 5 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$np --master-addr 127.0.0.1 --master-port $(( 31000 + seed )) \
     tests/dist_worker.py --mode solve --grid $n --stencil 7 --seq $sq --combo $seed $FUZZ_EXTRA > gpurun_out/fuzz_$seed.log 2>&1
  rc=$?
  if grep -q "dist solve ok" gpurun_out/fuzz_$seed.log; then echo "seed $seed np $np n $n seq $sq ok"; rm -f gpurun_out/fuzz_$seed.log; else echo "seed $seed np $np n $n seq $sq FAILED rc $rc"; fail=1; grep "rank[0-9]\]:.*Error\|AssertionError" gpurun_out/fuzz_$seed.log | head -3 | cut -c1-300; fi
done
exit $fail
