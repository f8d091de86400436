"""ILU(0) setup / application / GMRES(50)+ILU timing on the n^3 7-point Laplacian (gpurun -- python3 profiles/run_ilu_timing.py 256)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import __graft_entry__ as ge

mi = ge.load_binding()
mi.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A, b, x, _ = mi.build_laplace_system(n, n, n, 7, 0, 1)
for tri in (1, 0):
    ilu = mi.ILU(max_iterations=1, tolerance=0.0, trisolve=tri, print_level=1)
    t = time.perf_counter()
    ilu.setup(A)
    mi.call("HYPRE_MI_StreamSynchronize")
    ts = time.perf_counter() - t
    gm = mi.GMRES(tolerance=1e-6, max_iterations=400, kspace=50, print_level=0)
    gm.set_precond(ilu)
    gm.setup(A, b, x)
    x.fill(0.0)
    mi.call("HYPRE_MI_StreamSynchronize")
    t = time.perf_counter()
    gm.solve(A, b, x)
    mi.call("HYPRE_MI_StreamSynchronize")
    dt = time.perf_counter() - t
    print(f"trisolve {tri}: setup {ts:.2f} s, GMRES(50)+ILU(0) tol 1e-6: {gm.num_iterations} iterations in {dt:.2f} s "
          f"({dt / max(1, gm.num_iterations) * 1e3:.1f} ms per iteration), rel res {gm.final_rel_res:.2e}", flush=True)
