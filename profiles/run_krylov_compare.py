import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import __graft_entry__ as ge
mi = ge.load_binding(); mi.init()
n = int(sys.argv[1])
A, b, x, _ = mi.build_laplace_system(n, n, n, 7, 0, 1)
amg = mi.BoomerAMG(print_level=0)
for name, cls, cgs in (("gmres", mi.GMRES, None), ("cogmres cgs=0", mi.COGMRES, 0), ("cogmres cgs=2", mi.COGMRES, 2)):
    gm = cls(tolerance=1e-8, max_iterations=200, kspace=50, print_level=0)
    if cgs is not None:
        mi.call("HYPRE_ParCSRCOGMRESSetCGS", gm.h, cgs)
    gm.set_precond(amg)
    gm.setup(A, b, x)
    for k in range(3):
        x.fill(0.0)
        mi.call("HYPRE_MI_StreamSynchronize")
        t = time.perf_counter()
        gm.solve(A, b, x)
        mi.call("HYPRE_MI_StreamSynchronize")
        dt = time.perf_counter() - t
    print(name, "iters", gm.num_iterations, "rel", gm.final_rel_res, "solve ms %.1f" % (dt * 1e3), flush=True)
