"""How much does the ordering of a level cost the SpMV?  Takes level L of the 256^3 hierarchy (C-first ordered,
as the solver keeps it), rebuilds it in natural order and in a blocked C-first order (C then F inside blocks of
B natural rows), and times HYPRE_ParCSRMatrixMatvec on each.   gpurun -- python3 profiles/run_ordering_experiment.py 256 1"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import __graft_entry__ as ge

mi = ge.load_binding()
mi.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lev = int(sys.argv[2]) if len(sys.argv) > 2 else 1
A, b, x, _ = mi.build_laplace_system(n, n, n, 7, 0, 1)
amg = mi.BoomerAMG(print_level=0)
amg.setup(A)
ia, ja, a, shape = amg.level_csr(lev, 0)
perm = amg.level_perm(lev)          # perm[new] = old (natural)
cf = amg.level_cf(lev)              # in the new ordering
M = sp.csr_matrix((a, ja, ia), shape=shape)
N = shape[0]
pos = np.empty(N, dtype=np.int64)
pos[perm] = np.arange(N)            # old -> new
nat = M[pos][:, pos].tocsr()        # natural order: row old = row pos[old] of M
nat.sort_indices()
cf_nat = np.empty(N, dtype=np.int32)
cf_nat[perm] = cf


def timed(Msp, label):
    Msp = Msp.tocsr()
    Msp.sort_indices()
    m = Msp.shape[0]
    IJ = mi.IJMatrix(0, m - 1)
    coo = Msp.tocoo()
    IJ.set_values_coo(coo.row.astype(np.int64), coo.col.astype(np.int64), coo.data.astype(np.float64))
    IJ.assemble()
    xv = mi.IJVector(0, m - 1, np.random.default_rng(0).standard_normal(m))
    yv = mi.IJVector(0, m - 1, np.zeros(m))
    for _ in range(3):
        mi.call("HYPRE_ParCSRMatrixMatvec", 1.0, IJ.par, xv.par, 0.0, yv.par)
    mi.call("HYPRE_MI_StreamSynchronize")
    t = time.perf_counter()
    reps = 20
    for _ in range(reps):
        mi.call("HYPRE_ParCSRMatrixMatvec", 1.0, IJ.par, xv.par, 0.0, yv.par)
    mi.call("HYPRE_MI_StreamSynchronize")
    dt = (time.perf_counter() - t) / reps
    print(f"{label:34s} {dt * 1e3:8.3f} ms   ({(12.0 * Msp.nnz + 20.0 * m) / dt / 1e9:7.0f} GB/s algorithmic)", flush=True)


print(f"level {lev}: {N} rows, {M.nnz} entries, {100.0 * (cf == 1).mean():.0f} % C points")
timed(M, "C-first (solver's ordering)")
timed(nat, "natural")
for B in (512, 4096, 32768):
    # blocked C-first: inside every block of B natural rows C points first, then F points
    blk = np.arange(N) // B
    order = np.lexsort((np.arange(N), (cf_nat != 1).astype(np.int64), blk))  # new -> old
    p2 = np.empty(N, dtype=np.int64)
    p2[order] = np.arange(N)
    Mb = nat[order][:, order]
    timed(Mb, f"blocked C-first, B = {B}")
