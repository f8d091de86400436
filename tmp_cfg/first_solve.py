import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
mi = ge.load_binding(); mi.init()
n = int(sys.argv[1])
A, b, x, _ = mi.build_laplace_system(n, n, n, 7, 0, 1)
amg = mi.BoomerAMG(print_level=0)
gm = mi.GMRES(tolerance=1e-8, max_iterations=200, kspace=50, print_level=0)
gm.set_precond(amg)
gm.setup(A, b, x)
for k in range(3):
    x.fill(0.0)
    gm.solve(A, b, x)
    h = gm.residual_history()
    print("solve", k, "iters", gm.num_iterations, "rel", gm.final_rel_res, "hist[1]", h[1], "len", len(h), flush=True)
