"""Multi-component systems (BASELINE.json config 5): `num_components: 3` with
`segregated_solve: 1` (one solve per component, /root/reference/src/HypreSystem.cpp:681-729) and
`segregated_solve: 0` (ONE solve on a 3-component multivector: HYPRE_IJVectorSetNumComponents /
SetComponent, :567-571, :967) on a seeded non-symmetric convection-diffusion operator, BiCGSTAB + BoomerAMG
and GMRES + BoomerAMG, against the CPU oracle and against the committed fixture.

A multivector solve is the block system diag(A, A, A) with ONE Krylov space (inner products over all
components, matvec and preconditioner per component) -- what HYPRE's multivector kernels compute.

Tolerances: same iteration count, residual history 1e-8 relative per step, final relative residual within
1e-10 (north-star bar), solution within the reference's closeness rule rtol 1e-6 / atol 1e-8 (:815-818)
of the oracle's and of scipy's direct solve."""
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from tests.systems import convection_diffusion_3d, three_component_rhs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _close(x, ref, rtol=1e-6, atol=1e-8):
    d = np.abs(x - ref)
    return np.all(d < np.maximum(rtol * np.maximum(np.abs(x), np.abs(ref)), atol))


def _chunk(mi):
    c = mi.c_int()
    mi.call("HYPRE_MI_GetGSChunk", mi.C.byref(c))
    return c.value


def _problem(mi, oc, n):
    M = convection_diffusion_3d(n)
    B, X = three_component_rhs(M)
    A = mi.matrix_from_scipy(M)
    Ao = oc.Csr.from_scipy(M)
    oamg = oc.Amg(Ao, oc.default_params(gs_chunk=_chunk(mi)))
    return M, B, X, A, Ao, oamg


@pytest.mark.parametrize("method", ["bicgstab", "gmres"])
def test_multivector_solve_matches_oracle(mi, oc, method):
    """segregated_solve: 0 -- one Solve call on the 3-component multivector."""
    n = 12
    M, B, X, A, Ao, oamg = _problem(mi, oc, n)
    N = n ** 3
    b = mi.IJVector(0, N - 1, B, ncomp=3)
    x = mi.IJVector(0, N - 1, np.zeros((3, N)), ncomp=3)
    amg = mi.BoomerAMG(print_level=0)
    if method == "bicgstab":
        s = mi.BiCGSTAB(tolerance=1e-9, max_iterations=60, print_level=0)
    else:
        s = mi.GMRES(tolerance=1e-9, max_iterations=60, kspace=20, print_level=0)
    s.set_precond(amg)
    s.setup(A, b, x)
    assert s.solve(A, b, x) == 0
    blk = oc.Csr.from_scipy(sp.kron(sp.eye(3), M).tocsr())
    if method == "bicgstab":
        xo, info = oc.bicgstab(blk, B.ravel(), tol=1e-9, maxit=60, amg=oamg, ncomp=3)
    else:
        xo, info = oc.gmres(blk, B.ravel(), kdim=20, tol=1e-9, maxit=60, amg=oamg, ncomp=3)
    assert s.num_iterations == info["iters"]
    assert abs(s.final_rel_res - info["rel_res"]) <= 1e-10
    h = s.residual_history()
    assert np.allclose(h[: len(info["norms"])], info["norms"], rtol=1e-8, atol=1e-12 * info["norms"][0])
    got = x.get_all()
    assert _close(got.ravel(), xo)
    assert _close(got, X, rtol=1e-6, atol=1e-7)


def test_segregated_solves_match_oracle(mi, oc):
    """segregated_solve: 1 -- three Solve calls, one hierarchy (the reference rebuilds it per component)."""
    n = 12
    M, B, X, A, Ao, oamg = _problem(mi, oc, n)
    N = n ** 3
    amg = mi.BoomerAMG(print_level=0)
    s = mi.BiCGSTAB(tolerance=1e-9, max_iterations=60, print_level=0)
    s.set_precond(amg)
    for c in range(3):
        b = mi.IJVector(0, N - 1, B[c])
        x = mi.IJVector(0, N - 1, np.zeros(N))
        if c == 0:
            s.setup(A, b, x)
        assert s.solve(A, b, x) == 0
        xo, info = oc.bicgstab(Ao, B[c], tol=1e-9, maxit=60, amg=oamg)
        assert s.num_iterations == info["iters"]
        assert abs(s.final_rel_res - info["rel_res"]) <= 1e-10
        assert _close(x.get(), xo) and _close(x.get(), X[c], rtol=1e-6, atol=1e-7)


def test_multicomponent_fixture(mi):
    """Replay of tests/golden/convdiff3_12.npz (no oracle in the loop)."""
    g = np.load(os.path.join(GOLD, "convdiff3_12.npz"))
    n = int(g["n"])
    N = n ** 3
    M = sp.csr_matrix((g["data"], g["indices"], g["indptr"]), shape=(N, N))
    assert _chunk(mi) == 8
    A = mi.matrix_from_scipy(M)
    B = g["rhs"]
    amg = mi.BoomerAMG(print_level=0)
    s = mi.BiCGSTAB(tolerance=1e-9, max_iterations=60, print_level=0)
    s.set_precond(amg)
    b = mi.IJVector(0, N - 1, B, ncomp=3)
    x = mi.IJVector(0, N - 1, np.zeros((3, N)), ncomp=3)
    s.setup(A, b, x)
    assert s.solve(A, b, x) == 0
    assert s.num_iterations == int(g["iters_multi"])
    assert abs(s.final_rel_res - float(g["rel_res_multi"])) <= 1e-10
    assert _close(x.get_all(), g["x_multi"]) and _close(x.get_all(), g["x_direct"], rtol=1e-6, atol=1e-7)
    for c in range(3):
        bc = mi.IJVector(0, N - 1, B[c])
        xc = mi.IJVector(0, N - 1, np.zeros(N))
        assert s.solve(A, bc, xc) == 0
        assert s.num_iterations == int(g["iters_seg"][c])
        assert abs(s.final_rel_res - float(g["rel_res_seg"][c])) <= 1e-10
        assert _close(xc.get(), g["x_seg"][c])


def test_multivector_amg_solver_and_gmres_without_precond(mi, oc):
    """BoomerAMG as the solver and unpreconditioned GMRES on a multivector (every entry point takes one)."""
    n = 10
    M, B, X, A, Ao, oamg = _problem(mi, oc, n)
    N = n ** 3
    b = mi.IJVector(0, N - 1, B, ncomp=3)
    x = mi.IJVector(0, N - 1, np.zeros((3, N)), ncomp=3)
    amg = mi.BoomerAMG(print_level=0, max_iterations=30, tolerance=1e-9)
    amg.setup(A)
    amg.solve(A, b, x)
    got = x.get_all()
    for c in range(3):
        assert np.linalg.norm(B[c] - M @ got[c]) <= 1e-8 * np.linalg.norm(B)
    x.fill(0.0)
    gm = mi.GMRES(tolerance=1e-6, max_iterations=400, kspace=30, print_level=0)
    gm.setup(A, b, x)
    assert gm.solve(A, b, x) == 0
    blk = oc.Csr.from_scipy(sp.kron(sp.eye(3), M).tocsr())
    xo, info = oc.gmres(blk, B.ravel(), kdim=30, tol=1e-6, maxit=400)
    assert gm.num_iterations == info["iters"]
    assert abs(gm.final_rel_res - info["rel_res"]) <= 1e-9


def test_krylov_operator_differs_from_preconditioner_operator(mi, oc):
    """HYPRE multiplies by the A passed to Solve; the preconditioner may have been built on another operator of
    the same size.  The level-ordering fast path must not swallow that."""
    n = 12
    M = convection_diffusion_3d(n, seed=7)
    Mp = convection_diffusion_3d(n, seed=7, peclet=0.0)  # preconditioner: pure diffusion
    N = n ** 3
    rng = np.random.default_rng(3)
    xs = rng.standard_normal(N)
    rhs = M @ xs
    A, Ap = mi.matrix_from_scipy(M), mi.matrix_from_scipy(Mp)
    amg = mi.BoomerAMG(print_level=0)
    amg.setup(Ap)  # explicit setup on the approximate operator
    gm = mi.GMRES(tolerance=1e-10, max_iterations=80, kspace=40, print_level=0)
    L = mi.lib()
    # solve-only binding: the Krylov setup must not rebuild the hierarchy on A
    mi.call("HYPRE_ParCSRGMRESSetPrecond", gm.h, mi.C.cast(L.HYPRE_BoomerAMGSolve, mi.vp), None, amg.h)
    b = mi.IJVector(0, N - 1, rhs)
    x = mi.IJVector(0, N - 1, np.zeros(N))
    gm.setup(A, b, x)
    assert gm.solve(A, b, x) == 0
    Ao, Apo = oc.Csr.from_scipy(M), oc.Csr.from_scipy(Mp)
    oamg = oc.Amg(Apo, oc.default_params(gs_chunk=_chunk(mi)))
    xo, info = oc.gmres(Ao, rhs, kdim=40, tol=1e-10, maxit=80, amg=oamg)
    assert gm.num_iterations == info["iters"]
    assert abs(gm.final_rel_res - info["rel_res"]) <= 1e-10
    assert _close(x.get(), xo) and _close(x.get(), xs, rtol=1e-6, atol=1e-7)
    assert np.linalg.norm(rhs - M @ x.get()) <= 1e-9 * np.linalg.norm(rhs)
