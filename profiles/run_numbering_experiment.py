"""Does a 3-D blocked numbering of the unknowns pay?  (round 2 experiment, single GPU)

The x-cache kernels gather, per tile of <= 256 consecutive rows, the tile's unique columns.  With the lexicographic
numbering of the benchmark a tile is a piece of ONE grid line (U/N ~ 5 unique columns per row on level 0); a numbering
by bx x by x bz blocks makes a tile a 3-D brick (U/N ~ 2.3) but shortens the contiguous runs of the gathers.  This
script solves the same 7-pt Laplacian under both numberings of the CALLER's matrix (the hierarchy inherits the
numbering: C points keep their relative order) and prints time per iteration and level-0 kernel times.
   gpurun -- python3 profiles/run_numbering_experiment.py 256 8 4 4"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

mi = ge.load_binding()
mi.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bx, by, bz = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (8, 4, 4)
N = n ** 3


def run(newid, label):
    g = mi.laplace3d(n, n, n, 7, 0, N - 1)
    nnz = g["nnz"]
    rows = np.ctypeslib.as_array(C.cast(g["rows"], C.POINTER(C.c_longlong)), shape=(nnz,))
    cols = np.ctypeslib.as_array(C.cast(g["cols"], C.POINTER(C.c_longlong)), shape=(nnz,))
    vals = np.ctypeslib.as_array(C.cast(g["vals"], C.POINTER(C.c_double)), shape=(nnz,))
    rhs = np.ctypeslib.as_array(C.cast(g["rhs"], C.POINTER(C.c_double)), shape=(N,)).copy()
    A = mi.IJMatrix(0, N - 1)
    if newid is None:
        A.set_values_coo(rows, cols, vals)
        b = mi.IJVector(0, N - 1, rhs)
    else:
        r2, c2 = newid[rows], newid[cols]
        order = np.argsort(r2, kind="stable")  # row-ordered input takes the fast assembly path
        A.set_values_coo(r2[order], c2[order], vals[order])
        rp = np.empty(N)
        rp[newid] = rhs
        b = mi.IJVector(0, N - 1, rp)
        del r2, c2, order
    mi.laplace3d_free(g)
    A.assemble()
    x = mi.IJVector(0, N - 1, np.zeros(N))
    amg = mi.BoomerAMG(print_level=0)
    gm = mi.GMRES(tolerance=1e-8, max_iterations=200, kspace=50, print_level=0)
    gm.set_precond(amg)
    gm.setup(A, b, x)
    x.fill(0.0)
    gm.solve(A, b, x)
    mi.profile_enable(mi.PROF_SPMV_L0, 4096)
    mi.profile_enable(mi.PROF_RELAX_L0, 4096)
    mi.profile_reset()
    ts = []
    for _ in range(3):
        x.fill(0.0)
        mi.call("HYPRE_MI_StreamSynchronize")
        t0 = time.perf_counter()
        gm.solve(A, b, x)
        mi.call("HYPRE_MI_StreamSynchronize")
        ts.append(time.perf_counter() - t0)
    cs, ms, _ = mi.profile_get(mi.PROF_SPMV_L0)
    cr, mr, _ = mi.profile_get(mi.PROF_RELAX_L0)
    err = np.abs(x.get() - 1.0).max()
    print(f"{label:34s} {gm.num_iterations:3d} iterations  {min(ts) * 1e3:8.1f} ms per solve  {min(ts) * 1e3 / gm.num_iterations:6.2f} ms/iteration  "
          f"level-0 SpMV {ms / max(cs, 1):.3f} ms  level-0 relaxation pass {mr / max(cr, 1):.3f} ms  opcx {amg.operator_complexity:.2f}  "
          f"levels {amg.num_levels}  max|x-1| {err:.1e}", flush=True)
    for o in (gm, amg, x, b, A):
        o.destroy()


run(None, "lexicographic numbering")
z, y, x_ = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
nbx, nby = n // bx, n // by
blk = (x_ // bx) + nbx * ((y // by) + nby * (z // bz))
loc = (x_ % bx) + bx * ((y % by) + by * (z % bz))
newid = (blk.astype(np.int64) * (bx * by * bz) + loc).ravel()
del z, y, x_, blk, loc
run(newid, f"{bx}x{by}x{bz} block numbering")
if len(sys.argv) > 5 and sys.argv[5] == "morton":
    # Z-order over (x / 8, y, z) with the 8 points of an x-segment innermost: bricks nested in bricks
    z, y, x_ = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    def spread(v):
        v = v.astype(np.int64)
        out = np.zeros_like(v)
        for bit in range(10):
            out |= ((v >> bit) & 1) << (3 * bit)
        return out
    key = (spread(x_ >> 3) | (spread(y) << 1) | (spread(z) << 2)) * 8 + (x_ & 7)
    order = np.argsort(key.ravel(), kind="stable")
    newid = np.empty(N, dtype=np.int64)
    newid[order] = np.arange(N)
    del z, y, x_, key, order
    run(newid, "Morton order of 8-point x-segments")
