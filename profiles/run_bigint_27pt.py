"""The reference generator's 27-point operator (/root/reference/src/laplace_3d_weak_scaling.hpp:558,600) with MORE than
2^31 entries in one rank's block: 64-bit entry offsets of the solve format (DevCSR).   gpurun -- python3 profiles/run_bigint_27pt.py 432
432^3 = 80.6 M rows, 2.17e9 entries; 512^3 = 134 M rows, 3.61e9 entries."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
mi = ge.load_binding(); mi.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 432
t0 = time.time()
A, b, x, rhs = mi.build_laplace_system(n, n, n, 27)
t_build = time.time() - t0
N = n ** 3
nr, nc, nnz = mi.c_int(), mi.c_int(), mi.C.c_longlong()
amg = mi.BoomerAMG(print_level=1)
gm = mi.GMRES(tolerance=1e-8, max_iterations=100, kspace=50, print_level=0)
gm.set_precond(amg)
t0 = time.time(); gm.setup(A, b, x); t_setup = time.time() - t0
mi.call("HYPRE_MI_BoomerAMGGetLevelCSRSize", amg.h, 0, 0, mi.C.byref(nr), mi.C.byref(nc), mi.C.byref(nnz))
print(f"27-pt {n}^3: {N} rows, {nnz.value} entries in one block ({'>' if nnz.value >= 2**31 else '<'} 2^31), build {t_build:.1f} s, setup {t_setup:.1f} s, "
      f"{amg.num_levels} levels, operator complexity {amg.operator_complexity:.3f}", flush=True)
# A * 1 = rhs exactly (integer-valued sums): every entry of the big block is read at its right offset
ones = mi.IJVector(0, N - 1, np.ones(N)); y = mi.IJVector(0, N - 1, np.zeros(N))
mi.call("HYPRE_ParCSRMatrixMatvec", 1.0, A.par, ones.par, 0.0, y.par)
assert np.array_equal(y.get(), rhs), "A*1 != rhs"
hist = []
for rep in range(2):
    x.fill(0.0); t0 = time.time(); rc = gm.solve(A, b, x); t_solve = time.time() - t0
    hist.append(np.array(gm.residual_history()))
    print(f"solve {rep}: rc {rc}, {gm.num_iterations} iterations, rel res {gm.final_rel_res:.3e}, {t_solve*1e3:.0f} ms", flush=True)
assert np.array_equal(hist[0], hist[1])
xs = x.get()
print(f"max |x - 1| = {np.abs(xs - 1).max():.2e}")
r = mi.IJVector(0, N - 1, rhs)
mi.call("HYPRE_ParCSRMatrixMatvec", -1.0, A.par, x.par, 1.0, r.par)
rn = np.linalg.norm(r.get()) / np.linalg.norm(rhs)
print(f"true relative residual {rn:.3e}")
assert rn <= 1e-8 and np.abs(xs - 1).max() < 1e-5
print("bigint 27-pt ok")
