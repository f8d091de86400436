"""CPU experiment: what would a numbering of every LEVEL by its own graph buy?
Level l of the product's hierarchy (which inherits level 0's cells through its C points) is fed back to the library as an
INPUT matrix with the internal locality numbering forced on: level 0 of that second setup is level l renumbered by cells
of its own graph and put C-first by a PMIS splitting of the same operator -- what a per-level numbering would hand the
tile kernels.  Tile statistics (unique columns per entry, 64-byte sectors per unique column) of both, same tile rule."""
import os, sys, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MI_HYPRE_LOCALITY_ORDER"] = "1"
os.environ["MI_HYPRE_LOCALITY_MIN_ROWS"] = "0"
import __graft_entry__ as ge
mi = ge.load_binding(); mi.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
A, rhs = mi.build_laplace_system_host(n, n, n, 7, 0, 1)
amg = mi.BoomerAMG(print_level=0)
mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)

def tiles(M):
    ia = M.indptr; nrows = M.shape[0]; r = 0
    while r < nrows:
        start = ia[r]; e = r
        while e < nrows and e - r < 256:
            e2 = min(nrows, e + 8)
            if ia[e2] - start > 2047: break
            e = e2
        if e == r:
            while e < nrows and e - r < 256 and ia[e + 1] - start <= 2047: e += 1
            if e == r: e = r + 1
        yield r, e; r = e

def stats(M):
    ia = M.indptr; U = S64 = nt = 0
    for r, e in tiles(M):
        cols = np.unique(M.indices[ia[r]:ia[e]]); U += len(cols); nt += 1
        S64 += len(np.unique(cols >> 3))
    return M.nnz / M.shape[0], nt, U / M.nnz, S64 * 8 / U, (4 * U + 64 * S64) / M.nnz

print(f"{n}^3 7-pt: level l as the hierarchy has it (inherited numbering) vs renumbered by cells of its own graph")
print("level  rows     nnz/row   |  inherited: U/nnz  sectors*8/U  list+sector bytes/entry  |  own cells: U/nnz  sectors*8/U  bytes/entry")
for l in range(1, min(4, amg.num_levels - 1)):
    ia, ja, a, shape = amg.level_csr(l, 0)
    M = sp.csr_matrix((a, ja, ia), shape=shape); M.sort_indices()
    s0 = stats(M)
    # second setup with level l as the input
    coo = M.tocoo()
    A2 = mi.IJMatrix.__new__(mi.IJMatrix)  # (host-only: no Initialize, which would ask for the device)
    A2.h = mi.vp()
    A2.ilower, A2.iupper = 0, shape[0] - 1
    mi.call("HYPRE_IJMatrixCreate", 0, mi.c_big(0), mi.c_big(shape[0] - 1), mi.c_big(0), mi.c_big(shape[0] - 1), mi.C.byref(A2.h))
    mi.call("HYPRE_IJMatrixSetObjectType", A2.h, mi.HYPRE_PARCSR)
    A2.par = mi.vp()
    mi.call("HYPRE_IJMatrixGetObject", A2.h, mi.C.byref(A2.par))
    A2.set_values_coo(coo.row.astype(np.int64), coo.col.astype(np.int64), coo.data)
    mi.call("HYPRE_MI_IJMatrixAssembleHostOnly", A2.h)
    amg2 = mi.BoomerAMG(print_level=0, max_levels=2)
    mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg2.h, A2.par)
    applied, order = amg2.input_ordering()
    assert applied
    ia2, ja2, a2, shape2 = amg2.level_csr(0, 0)
    M2 = sp.csr_matrix((a2, ja2, ia2), shape=shape2); M2.sort_indices()
    s1 = stats(M2)
    print(f"L{l}   {shape[0]:8d}  {s0[0]:6.1f}   |  {s0[2]:.3f}  {s0[3]:.2f}  {s0[4]:.2f}  |  {s1[2]:.3f}  {s1[3]:.2f}  {s1[4]:.2f}   (gather bytes {100 * (s1[4] / s0[4] - 1):+.0f} %)")
