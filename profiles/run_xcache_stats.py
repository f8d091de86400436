"""Unique columns per tile of every level operator (x-cache statistics):  gpurun -- python3 profiles/run_xcache_stats.py 256"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

mi = ge.load_binding()
mi.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A, b, x, _ = mi.build_laplace_system(n, n, n, 7, 0, 1)
amg = mi.BoomerAMG(print_level=0)
amg.setup(A)
print(f"laplace_3d {n}^3: level, rows, nnz, nnz/row, tiles, entries/tile, unique columns/tile, U/nnz, column-list bytes / matrix-stream bytes")
for l in range(amg.num_levels):
    def size(which):
        nr, nc, nnz = C.c_int(), C.c_int(), C.c_longlong()
        mi.call("HYPRE_MI_BoomerAMGGetLevelCSRSize", amg.h, l, which, C.byref(nr), C.byref(nc), C.byref(nnz))
        return nr.value, nnz.value
    rows, nnz = size(0)
    tiles, U = size(7)
    if not tiles or not nnz:
        continue
    print(f"{l:2d} {rows:10d} {nnz:11d} {nnz / rows:6.1f} {tiles:8d} {nnz / tiles:7.0f} {U / tiles:7.0f} {U / nnz:6.3f} {4 * U / (10 * nnz):6.3f}")
