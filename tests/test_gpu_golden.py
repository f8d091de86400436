"""The HIP path against the COMMITTED fixtures of tests/golden/ (no oracle in the loop):
iteration counts, residual histories, hierarchy shape, C/F split and solution of every
single-part case, and the scipy direct solve of the random M-matrix fixture.

Tolerances as in test_gpu_amg.py: same iteration count, residual history to 1e-7
relative per step, final relative residual within 1e-10, solution within rtol 1e-6 /
atol 1e-8 (the reference's closeness rule, /root/reference/src/HypreSystem.cpp:815-818).
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# mirrors CASES of tests/golden/make_golden.py for the single-part fixtures (the multi-part ones are
# replayed rank by rank in test_dist.py)
CASES = {
    "lap7_8": dict(n=8, stencil=7, kdim=50, tol=1e-8, method="gmres"),
    "lap7_16": dict(n=16, stencil=7, kdim=50, tol=1e-8, method="gmres"),
    "lap7_16_k5": dict(n=16, stencil=7, kdim=5, tol=1e-10, method="gmres"),
    "lap27_10": dict(n=10, stencil=27, kdim=50, tol=1e-8, method="gmres"),
    "lap7_12_cogmres": dict(n=12, stencil=7, kdim=50, tol=1e-8, method="cogmres"),
    "lap7_12_pcg": dict(n=12, stencil=7, kdim=50, tol=1e-8, method="pcg"),
    "lap7_12_bicgstab": dict(n=12, stencil=7, kdim=50, tol=1e-8, method="bicgstab"),
    "lap7_10_ilu": dict(n=10, stencil=7, kdim=50, tol=1e-8, method="gmres_ilu"),
    "lap7_12_falgout_sgs": dict(n=12, stencil=7, kdim=50, tol=1e-8, method="gmres",
                                amg=dict(coarsen_type=6, relax_type=6, num_sweeps=2, interp_type=0)),
    "lap7_12_hmis": dict(n=12, stencil=7, kdim=50, tol=1e-8, method="gmres", amg=dict(coarsen_type=10)),
    "lap7_12_cljp": dict(n=12, stencil=7, kdim=50, tol=1e-8, method="gmres", amg=dict(coarsen_type=0)),
    "lap7_12_agg1": dict(n=12, stencil=7, kdim=50, tol=1e-8, method="gmres", amg=dict(agg_num_levels=1)),
    "lap7_12_multipass": dict(n=12, stencil=7, kdim=50, tol=1e-8, method="gmres", amg=dict(interp_type=4)),
    "lap7_12_ilu_smoother": dict(n=12, stencil=7, kdim=50, tol=1e-8, method="gmres",
                                 amg=dict(smooth_type=5, smooth_num_levels=2)),
    "lap7_12_two_stage_gs": dict(n=12, stencil=7, kdim=50, tol=1e-8, method="gmres", amg=dict(relax_type=11, relax_order=0)),
}


def _close(x, ref, rtol=1e-6, atol=1e-8):
    d = np.abs(x - ref)
    return np.all(d < np.maximum(rtol * np.maximum(np.abs(x), np.abs(ref)), atol))


def test_every_single_part_fixture_is_covered():
    names = {os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLD, "lap*.npz"))}
    multi = {n for n in names if "_p2" in n or "_p3" in n}
    assert names - multi == set(CASES)


@pytest.mark.parametrize("name", sorted(CASES))
def test_fixture(mi, name):
    c = CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    n = c["n"]
    chunk = mi.c_int()
    mi.call("HYPRE_MI_GetGSChunk", mi.C.byref(chunk))
    assert chunk.value == 8  # the fixtures were made with 8-row hybrid-GS chunks
    A, b, x, rhs = mi.build_laplace_system(n, n, n, c["stencil"])
    assert np.array_equal(b.get(), g["rhs"])
    amg = mi.BoomerAMG(print_level=0, **c.get("amg", {}))
    method = c["method"]
    if method == "gmres_ilu":
        amg.setup(A)  # hierarchy shape only
        pre = mi.ILU(max_iterations=1, tolerance=0.0)
        s = mi.GMRES(tolerance=c["tol"], max_iterations=200, kspace=c["kdim"], print_level=0)
    else:
        pre = amg
        if method == "gmres":
            s = mi.GMRES(tolerance=c["tol"], max_iterations=100, kspace=c["kdim"], print_level=0)
        elif method == "cogmres":
            s = mi.COGMRES(tolerance=c["tol"], max_iterations=100, kspace=c["kdim"], print_level=0)
            mi.call("HYPRE_ParCSRCOGMRESSetCGS", s.h, 0)
        elif method == "pcg":
            s = mi.PCG(tolerance=c["tol"], max_iterations=100, print_level=0)
        else:
            s = mi.BiCGSTAB(tolerance=c["tol"], max_iterations=100, print_level=0)
    s.set_precond(pre)
    s.setup(A, b, x)
    assert s.solve(A, b, x) == 0
    assert s.num_iterations == int(g["iters"])
    assert abs(s.final_rel_res - float(g["rel_res"])) <= 1e-10
    if method != "bicgstab":  # the oracle logs BiCGSTAB's half steps differently from the library's history
        hist = np.asarray(s.residual_history())
        assert len(hist) == len(g["norms"]) and np.allclose(hist, g["norms"], rtol=1e-7, atol=0.0)
    xs = x.get()
    assert _close(xs, g["x"]) and _close(xs, np.ones_like(xs), rtol=max(1e-6, 100 * c["tol"]))
    # hierarchy
    assert amg.num_levels == len(g["level_rows"])
    for l in range(amg.num_levels):
        ia, ja, a, shape = amg.level_csr(l, 0)
        assert shape[0] == g["level_rows"][l] and len(ja) == g["level_nnz"][l]
    assert np.array_equal(np.asarray(amg.level_cf(0), dtype=np.int8), g["cf0"])


def test_random_mmatrix_fixture(mi):
    g = np.load(os.path.join(GOLD, "random_mmatrix_400.npz"))
    n = len(g["rhs"])
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(g["indptr"]))
    A = mi.IJMatrix(0, n - 1)
    A.set_values_coo(rows, g["indices"].astype(np.int64), g["data"])
    A.assemble()
    b = mi.IJVector(0, n - 1, g["rhs"])
    x = mi.IJVector(0, n - 1, np.zeros(n))
    amg = mi.BoomerAMG(print_level=0)
    gm = mi.GMRES(tolerance=1e-12, max_iterations=200, kspace=50, print_level=0)
    gm.set_precond(amg)
    gm.setup(A, b, x)
    assert gm.solve(A, b, x) == 0
    assert np.abs(x.get() - g["x_direct"]).max() <= 1e-9 * np.abs(g["x_direct"]).max()
