"""One BoomerAMG setup of the n^3 7-point Laplacian (for kernel traces of the setup alone)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge

mi = ge.load_binding()
mi.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A, b, x, _ = mi.build_laplace_system(n, n, n, 7, 0, 1)
amg = mi.BoomerAMG(print_level=0)
amg.setup(A)
print("levels", amg.num_levels)
