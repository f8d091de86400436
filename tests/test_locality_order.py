"""The internal locality numbering of the input (BoomerAMG::use_locality_order): with MI_HYPRE_LOCALITY_ORDER=1 the
hierarchy is built on Q A Q^T.  Q is a heuristic of the product (any permutation gives a valid solver); what is
checked: it IS a permutation made of graph clusters, the hierarchy equals the oracle's hierarchy of the SAME
permuted matrix bit for bit (CPU, host setup), and on the GPU the V-cycle and GMRES agree with the oracle on the
permuted system while the caller sees its own numbering."""
import numpy as np
import pytest
import scipy.sparse as sp


def _amg_with_order(mi, n, stencil, host_only, monkeypatch, **kw):
    monkeypatch.setenv("MI_HYPRE_LOCALITY_ORDER", "1")
    if host_only:
        A, rhs = mi.build_laplace_system_host(n, n, n, stencil, 0, 1)
        amg = mi.BoomerAMG(print_level=0, **kw)
        mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)
        return A, rhs, amg
    A, b, x, rhs = mi.build_laplace_system(n, n, n, stencil)
    amg = mi.BoomerAMG(print_level=0, **kw)
    amg.setup(A)
    return (A, b, x), rhs, amg


def _permuted_oracle(oc, n, stencil, order, **okw):
    Ao, bo = oc.Csr.laplace(n, n, n, stencil)
    M = Ao.to_scipy().tocsr()
    Mq = M[order][:, order].tocsr()
    Mq.sort_indices()
    Aq = oc.Csr.from_scipy(Mq)
    return Aq, bo[order], oc.Amg(Aq, oc.default_params(**okw))


@pytest.mark.parametrize("n,stencil,kw", [(14, 7, {}), (10, 27, {}), (12, 7, dict(coarsen_type=6, interp_type=0))])
def test_host_hierarchy_on_the_permuted_matrix_equals_oracle(mi_lib, oc, n, stencil, kw, monkeypatch):
    mi = mi_lib
    A, rhs, amg = _amg_with_order(mi, n, stencil, True, monkeypatch, **kw)
    applied, order = amg.input_ordering()
    N = n ** 3
    assert applied and np.array_equal(np.sort(order), np.arange(N)) and not np.array_equal(order, np.arange(N))
    Aq, bq, oamg = _permuted_oracle(oc, n, stencil, order, **kw)
    assert amg.num_levels == oamg.num_levels and amg.num_levels > 1
    for l in range(amg.num_levels):
        ia, ja, a, shape = amg.level_csr(l, 0)
        oia, oja, oa = oamg.level_A(l).arrays()
        assert np.array_equal(ia, oia) and np.array_equal(ja, oja) and np.array_equal(a, oa)
        if l < amg.num_levels - 1:
            assert np.array_equal(amg.level_cf(l), oamg.level_cf(l))
            pia, pja, pa, _ = amg.level_csr(l, 2)
            qia, qja, qa = oamg.level_P(l).arrays()
            assert np.array_equal(pia, qia) and np.array_equal(pja, qja) and np.array_equal(pa, qa)
    # level-0 rows -> caller rows: the oracle's C-first perm of the permuted matrix, seen through the input order
    assert np.array_equal(amg.level_perm(0), order[oamg.level_perm(0)])
    for l in range(1, amg.num_levels - 1):
        assert np.array_equal(amg.level_perm(l), oamg.level_perm(l))


def test_order_is_off_by_default_on_small_systems(mi_lib, monkeypatch):
    mi = mi_lib
    monkeypatch.delenv("MI_HYPRE_LOCALITY_ORDER", raising=False)
    A, rhs = mi.build_laplace_system_host(10, 10, 10, 7, 0, 1)
    amg = mi.BoomerAMG(print_level=0)
    mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)
    applied, order = amg.input_ordering()
    assert not applied and np.array_equal(order, np.arange(1000))


def test_clusters_are_graph_balls(mi_lib, monkeypatch):
    """Rows of one cluster are consecutive in the new order and connected in the matrix graph; a tile of 256
    consecutive new rows of a long-line grid touches far fewer distinct columns than 256 lexicographic rows."""
    mi = mi_lib
    monkeypatch.setenv("MI_HYPRE_LOCALITY_ORDER", "1")
    nx, ny, nz = 200, 12, 12
    N = nx * ny * nz
    ilo, ihi = 0, N - 1
    g = mi.laplace3d(nx, ny, nz, 7, ilo, ihi)
    A = mi.IJMatrix.__new__(mi.IJMatrix)
    A.h = mi.vp()
    A.ilower, A.iupper = ilo, ihi
    mi.call("HYPRE_IJMatrixCreate", 0, mi.c_big(ilo), mi.c_big(ihi), mi.c_big(ilo), mi.c_big(ihi), mi.C.byref(A.h))
    mi.call("HYPRE_IJMatrixSetObjectType", A.h, mi.HYPRE_PARCSR)
    A.par = mi.vp()
    mi.call("HYPRE_IJMatrixGetObject", A.h, mi.C.byref(A.par))
    A.set_values_ptr(g["nnz"], g["rows"], g["cols"], g["vals"])
    mi.call("HYPRE_MI_IJMatrixAssembleHostOnly", A.h)
    mi.laplace3d_free(g)
    amg = mi.BoomerAMG(print_level=0, max_levels=2)
    mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)
    applied, order = amg.input_ordering()
    assert applied and np.array_equal(np.sort(order), np.arange(N))
    x, y, z = order % nx, (order // nx) % ny, order // (nx * ny)

    def unique_cols_per_row(idx):
        tot = 0
        for t in range(0, N - 255, 256):
            rows = idx[t:t + 256]
            X, Y, Z = rows % nx, (rows // nx) % ny, rows // (nx * ny)
            cols = {int(r) for r in rows}
            for dx, dy, dz in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
                ok = (X + dx >= 0) & (X + dx < nx) & (Y + dy >= 0) & (Y + dy < ny) & (Z + dz >= 0) & (Z + dz < nz)
                cols.update(((X + dx) + nx * ((Y + dy) + ny * (Z + dz)))[ok].tolist())
            tot += len(cols)
        return tot / (N // 256 * 256)

    lex, clustered = unique_cols_per_row(np.arange(N)), unique_cols_per_row(order)
    assert lex > 4.0 and clustered < 0.75 * lex, (lex, clustered)
    # spatial extent of 512 consecutive new rows: a ball, not a line
    ext = max(np.ptp(x[:512]), np.ptp(y[:512]), np.ptp(z[:512]))
    assert ext <= 40, ext


@pytest.mark.gpu
@pytest.mark.parametrize("n,stencil", [(16, 7), (10, 27)])
def test_device_solve_with_locality_order(mi, oc, n, stencil, monkeypatch):
    (A, b, x), rhs, amg = _amg_with_order(mi, n, stencil, False, monkeypatch)
    applied, order = amg.input_ordering()
    assert applied
    chunk = mi.c_int()
    mi.call("HYPRE_MI_GetGSChunk", mi.C.byref(chunk))
    Aq, bq, oamg = _permuted_oracle(oc, n, stencil, order, gs_chunk=chunk.value)
    N = n ** 3
    # one V-cycle on a random vector: the caller's numbering in, the caller's numbering out
    rng = np.random.default_rng(5)
    f = rng.standard_normal(N)
    fi, ui = mi.IJVector(0, N - 1, f), mi.IJVector(0, N - 1, np.zeros(N))
    amg.solve(A, fi, ui)
    ref = oamg.cycle(f[order])
    assert np.abs(ui.get()[order] - ref).max() <= 1e-12 * np.abs(ref).max()
    gm = mi.GMRES(tolerance=1e-9, max_iterations=60, kspace=30, print_level=0)
    gm.set_precond(amg)
    gm.setup(A, b, x)
    assert gm.solve(A, b, x) == 0
    xo, info = oc.gmres(Aq, bq, kdim=30, tol=1e-9, maxit=60, amg=oamg)
    assert gm.num_iterations == info["iters"]
    assert np.allclose(gm.residual_history(), info["norms"], rtol=1e-8, atol=0.0)
    assert abs(gm.final_rel_res - info["rel_res"]) <= 1e-10
    xs = x.get()
    assert np.abs(xs[order] - xo).max() < 1e-9 and np.abs(xs - 1.0).max() < 1e-6


@pytest.mark.gpu
def test_irregular_graph_with_locality_order(mi, oc, monkeypatch):
    """A random sparse M-matrix (no grid structure, isolated rows, rows of very different length): the numbering is a
    permutation, the solve matches the oracle on the permuted system and scipy's direct solve in the caller's order."""
    import scipy.sparse.linalg as spl

    monkeypatch.setenv("MI_HYPRE_LOCALITY_ORDER", "1")
    rng = np.random.default_rng(77)
    n = 4000
    M = sp.random(n, n, density=0.0015, random_state=rng, format="csr")
    M = (M + M.T).tocsr()
    M = (M - sp.diags(M.diagonal())).tolil()
    M[17, :] = 0.0  # isolated rows
    M[:, 17] = 0.0
    M[2048, :] = 0.0
    M[:, 2048] = 0.0
    M = M.tocsr()
    M = (-abs(M) + sp.diags(abs(M).sum(axis=1).A1 + 0.3)).tocsr()
    M.eliminate_zeros()
    M.sort_indices()
    xs = rng.standard_normal(n)
    rhs = M @ xs
    A = mi.matrix_from_scipy(M)
    b = mi.IJVector(0, n - 1, rhs)
    x = mi.IJVector(0, n - 1, np.zeros(n))
    amg = mi.BoomerAMG(print_level=0, strong_threshold=0.25)
    gm = mi.GMRES(tolerance=1e-11, max_iterations=100, kspace=50, print_level=0)
    gm.set_precond(amg)
    gm.setup(A, b, x)
    assert gm.solve(A, b, x) == 0
    applied, order = amg.input_ordering()
    assert applied and np.array_equal(np.sort(order), np.arange(n))
    chunk = mi.c_int()
    mi.call("HYPRE_MI_GetGSChunk", mi.C.byref(chunk))
    Mq = M[order][:, order].tocsr()
    Mq.sort_indices()
    Aq = oc.Csr.from_scipy(Mq)
    oamg = oc.Amg(Aq, oc.default_params(gs_chunk=chunk.value, strong_threshold=0.25))
    xo, info = oc.gmres(Aq, rhs[order], kdim=50, tol=1e-11, maxit=100, amg=oamg)
    assert gm.num_iterations == info["iters"] and abs(gm.final_rel_res - info["rel_res"]) <= 1e-10
    got = x.get()
    assert np.abs(got[order] - xo).max() <= 1e-9 * np.abs(xo).max()
    xd = spl.spsolve(M.tocsc(), rhs)
    assert np.abs(got - xd).max() <= 1e-8 * np.abs(xd).max()


@pytest.mark.gpu
@pytest.mark.parametrize("n,stencil", [(30, 7), (28, 27)])
def test_device_numbering_equals_host_numbering(mi, oc, n, stencil, monkeypatch):
    """Levels the device builds (>= 20 000 rows) get their numbering from the device rounds (sk::locality_labels) and
    Q A Q^T from a device permutation of the assembled matrix: the same Q as the host routine, and the hierarchy on
    it equals the oracle's hierarchy of the permuted matrix bit for bit."""
    A_host, rhs, amg_host = _amg_with_order(mi, n, stencil, True, monkeypatch)
    (A, b, x), rhs, amg = _amg_with_order(mi, n, stencil, False, monkeypatch)
    applied_h, order_h = amg_host.input_ordering()
    applied, order = amg.input_ordering()
    N = n ** 3
    assert applied and applied_h and np.array_equal(order, order_h)
    assert np.array_equal(np.sort(order), np.arange(N)) and not np.array_equal(order, np.arange(N))
    Aq, bq, oamg = _permuted_oracle(oc, n, stencil, order)
    assert amg.num_levels == oamg.num_levels and amg.num_levels > 2
    for l in range(amg.num_levels):
        ia, ja, a, shape = amg.level_csr(l, 0)
        oia, oja, oa = oamg.level_A(l).arrays()
        assert np.array_equal(ia, oia) and np.array_equal(ja, oja) and np.array_equal(a, oa), l
    assert np.array_equal(amg.level_perm(0), order[oamg.level_perm(0)])


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(relax_type=11), dict(relax_type=12), dict(smooth_type=5, smooth_num_levels=1), dict()])
def test_order_dependent_smoothers_with_and_without_the_numbering(mi, kw, monkeypatch):
    """the two-stage Gauss-Seidel smoothers (strictly-lower part of the block), the ILU(0) complex smoother,
    the PMIS random stream and the 8-row Gauss-Seidel chunks are all defined on the library's INTERNAL numbering, so a
    solve can differ between numbering on and off (and from HYPRE on the same input).  The difference is bounded: same
    solution, iteration counts within two of each other."""
    n = 40
    its = {}
    for order in ("0", "1"):
        monkeypatch.setenv("MI_HYPRE_LOCALITY_ORDER", order)
        A, b, x, rhs = mi.build_laplace_system(n, n, n, 7)
        amg = mi.BoomerAMG(print_level=0, **kw)
        gm = mi.GMRES(tolerance=1e-9, max_iterations=100, kspace=50, print_level=0)
        gm.set_precond(amg)
        gm.setup(A, b, x)
        assert gm.solve(A, b, x) == 0
        assert amg.input_ordering()[0] == (order == "1")
        assert np.abs(x.get() - 1.0).max() < 1e-6 and gm.final_rel_res <= 1e-9
        its[order] = gm.num_iterations
    assert abs(its["0"] - its["1"]) <= 2, its
