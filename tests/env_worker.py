"""One GMRES + BoomerAMG solve in a process of its own, for the library switches that are read once per process
(tests/test_gpu_amg.py::test_default_on_features_against_their_switches): prints iterations, the residual history and
the solution as hex so that two runs can be compared bit for bit."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    n, stencil = int(sys.argv[1]), int(sys.argv[2])
    mi = ge.load_binding()
    mi.init()
    A, b, x, _ = mi.build_laplace_system(n, n, n, stencil)
    amg = mi.BoomerAMG(print_level=0)
    gm = mi.GMRES(tolerance=1e-10, max_iterations=100, kspace=50, print_level=0)
    gm.set_precond(amg)
    gm.setup(A, b, x)
    assert gm.solve(A, b, x) == 0
    xs = x.get()
    v = mi.C.c_longlong()
    mi.call("HYPRE_MI_GetCounter", b"arena_mapped_bytes", mi.C.byref(v))
    print("RESULT " + json.dumps({"iters": gm.num_iterations, "levels": amg.num_levels,
                                  "hist": [float(h).hex() for h in gm.residual_history()],
                                  "x": xs.tobytes().hex(), "arena_mapped": v.value}))


if __name__ == "__main__":
    main()
