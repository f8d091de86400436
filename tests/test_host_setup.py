"""CPU: the product's host control plane (IJ assembly + BoomerAMG setup, C++ with
threads) through the C ABI's host-only entry points, against the oracle.  The two
are independent implementations of one specification: equal C/F
splittings, interpolation and Galerkin operators mean both restate it the same way."""
import numpy as np
import pytest
import scipy.sparse as sp


def _host_amg(mi, n, stencil, **kw):
    A, rhs = mi.build_laplace_system_host(n, n, n, stencil, 0, 1)
    amg = mi.BoomerAMG(print_level=0, **kw)
    mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)
    return A, amg


@pytest.mark.parametrize("n,stencil,kw", [(14, 7, {}), (9, 27, {}), (12, 7, dict(interp_type=3)),
                                          (12, 7, dict(interp_type=0)), (12, 7, dict(strong_threshold=0.25)),
                                          (12, 7, dict(max_coarse_size=100)), (12, 7, dict(max_levels=3)),
                                          # coarsening types: HMIS, one-pass RS, Falgout / RS (two passes)
                                          (12, 7, dict(coarsen_type=10)), (9, 27, dict(coarsen_type=10)),
                                          (12, 7, dict(coarsen_type=11)), (13, 7, dict(coarsen_type=6)),
                                          (9, 27, dict(coarsen_type=6)), (12, 7, dict(coarsen_type=1, interp_type=0)),
                                          (12, 7, dict(coarsen_type=3, interp_type=3)),
                                          # the upstream sample's AMG block (etc/hypre_app.yaml:33-42)
                                          (14, 7, dict(coarsen_type=6, interp_type=0, relax_type=6, num_sweeps=2)),
                                          # aggressive coarsening + multipass interpolation
                                          (14, 7, dict(agg_num_levels=1)), (16, 7, dict(agg_num_levels=2)),
                                          (10, 27, dict(agg_num_levels=1)),
                                          (14, 7, dict(agg_num_levels=1, agg_pmax_elmts=4)),
                                          (14, 7, dict(agg_num_levels=1, agg_trunc_factor=0.3)),
                                          (14, 7, dict(agg_num_levels=1, coarsen_type=10)),
                                          (13, 7, dict(agg_num_levels=1, coarsen_type=6, interp_type=0)),
                                          # CLJP (0; 7 = its one-global-stream variant, the same thing here)
                                          (12, 7, dict(coarsen_type=0)), (9, 27, dict(coarsen_type=7, interp_type=0)),
                                          (12, 7, dict(coarsen_type=0, agg_num_levels=1)), (11, 7, dict(coarsen_type=9)),
                                          # multipass interpolation on ordinary splittings (interp_type 4)
                                          (12, 7, dict(interp_type=4)), (9, 27, dict(interp_type=4, coarsen_type=10)),
                                          (12, 7, dict(interp_type=4, trunc_factor=0.2))])
def test_host_setup_equals_oracle(mi_lib, oc, n, stencil, kw):
    mi = mi_lib
    A, amg = _host_amg(mi, n, stencil, **kw)
    Ao, bo = oc.Csr.laplace(n, n, n, stencil)
    oamg = oc.Amg(Ao, oc.default_params(**kw))
    assert amg.num_levels == oamg.num_levels and amg.num_levels > 1
    for l in range(amg.num_levels):
        ia, ja, a, shape = amg.level_csr(l, 0)
        oia, oja, oa = oamg.level_A(l).arrays()
        assert np.array_equal(ia, oia) and np.array_equal(ja, oja) and np.array_equal(a, oa)  # bit-exact
        if l < amg.num_levels - 1:
            assert np.array_equal(amg.level_cf(l), oamg.level_cf(l))
            assert np.array_equal(amg.level_perm(l), oamg.level_perm(l))
            pia, pja, pa, _ = amg.level_csr(l, 2)
            qia, qja, qa = oamg.level_P(l).arrays()
            assert np.array_equal(pia, qia) and np.array_equal(pja, qja) and np.array_equal(pa, qa)
            ria, rja, ra, rshape = amg.level_csr(l, 3)
            R = sp.csr_matrix((ra, rja, ria), shape=rshape)
            P = sp.csr_matrix((pa, pja, pia), shape=(rshape[1], rshape[0]))
            assert (abs(R - P.T)).nnz == 0


def test_ij_assembly_semantics_host(mi_lib):
    """Unsorted input, duplicates, Set-after-Add and Add-after-Set in submission order."""
    mi = mi_lib
    C = mi.C
    n = 5
    A = mi.IJMatrix.__new__(mi.IJMatrix)
    A.h = mi.vp()
    mi.call("HYPRE_IJMatrixCreate", 0, mi.c_big(0), mi.c_big(n - 1), mi.c_big(0), mi.c_big(n - 1), C.byref(A.h))
    A.par = mi.vp()
    mi.call("HYPRE_IJMatrixGetObject", A.h, C.byref(A.par))
    # general SetValues2 form: ncols per row, explicit row_indexes
    ncols = np.array([2, 1, 2], dtype=np.int32)
    rows = np.array([4, 0, 2], dtype=np.int64)
    row_indexes = np.array([3, 0, 1], dtype=np.int32)
    cols = np.array([0, 2, 1, 4, 3], dtype=np.int64)          # row0:(0) row2:(2,1) row4:(4,3)
    vals = np.array([10.0, 22.0, 21.0, 44.0, 43.0])
    mi.call("HYPRE_IJMatrixSetValues2", A.h, 3, ncols, rows, row_indexes, cols, vals)
    one = lambda r, c, v, add: A.set_values_coo(np.array([r], dtype=np.int64), np.array([c], dtype=np.int64),
                                                np.array([v]), add=add)
    one(2, 1, 1.0, True)     # 21 + 1
    one(4, 4, 5.0, False)    # overwrite 44 -> 5
    one(4, 4, 0.5, True)     # 5.5
    one(1, 1, 7.0, True)     # AddTo on an empty slot creates it
    one(3, 3, 1.0, False)
    mi.call("HYPRE_MI_IJMatrixAssembleHostOnly", A.h)
    amg = mi.BoomerAMG(print_level=0, max_levels=1)
    mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)
    ia, ja, a, shape = amg.level_csr(0, 0)
    got = sp.csr_matrix((a, ja, ia), shape=shape).toarray()
    want = np.zeros((n, n))
    want[0, 0], want[1, 1], want[2, 1], want[2, 2], want[3, 3], want[4, 3], want[4, 4] = 10, 7, 22, 22, 1, 43, 5.5
    assert np.array_equal(got, want)
    assert np.all(np.diff(ja[ia[2]:ia[3]]) > 0)
    # a row outside the owned range is an error (the driver never produces one)
    B = mi.IJMatrix.__new__(mi.IJMatrix)
    B.h = mi.vp()
    mi.call("HYPRE_IJMatrixCreate", 0, mi.c_big(10), mi.c_big(19), mi.c_big(10), mi.c_big(19), C.byref(B.h))
    B.set_values_coo(np.array([3], dtype=np.int64), np.array([3], dtype=np.int64), np.array([1.0]))
    with pytest.raises(mi.HypreError):
        mi.call("HYPRE_MI_IJMatrixAssembleHostOnly", B.h)
    mi.call("HYPRE_ClearAllErrors")


def test_empty_and_tiny_systems_host(mi_lib, oc):
    mi = mi_lib
    # 1x1x1 grid: one row, one level
    A, amg = _host_amg(mi, 1, 7)
    assert amg.num_levels == 1
    ia, ja, a, shape = amg.level_csr(0, 0)
    assert shape == (1, 1) and a.tolist() == [6.0]
    # diagonal matrix: every row is isolated (no strong connections) -> no coarsening
    n = 50
    D = mi.IJMatrix.__new__(mi.IJMatrix)
    D.h = mi.vp()
    mi.call("HYPRE_IJMatrixCreate", 0, mi.c_big(0), mi.c_big(n - 1), mi.c_big(0), mi.c_big(n - 1), mi.C.byref(D.h))
    D.par = mi.vp()
    mi.call("HYPRE_IJMatrixGetObject", D.h, mi.C.byref(D.par))
    idx = np.arange(n, dtype=np.int64)
    D.set_values_coo(idx, idx, np.arange(1, n + 1, dtype=float))
    mi.call("HYPRE_MI_IJMatrixAssembleHostOnly", D.h)
    amg = mi.BoomerAMG(print_level=0)
    mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, D.par)
    assert amg.num_levels == 1


def test_unrestated_settings_are_refused_not_substituted(mi_lib):
    """The CGC coarsenings (21, 22) and the two-stage aggressive interpolations are not implemented: Setup says so
    instead of silently building a PMIS / multipass hierarchy."""
    mi = mi_lib
    A, rhs = mi.build_laplace_system_host(6, 6, 6, 7, 0, 1)
    for kw in (dict(coarsen_type=21), dict(coarsen_type=22), dict(agg_num_levels=1, agg_interp_type=1),
               dict(relax_type=16), dict(interp_type=18), dict(interp_type=7), dict(interp_type=14)):
        amg = mi.BoomerAMG(print_level=0, **kw)
        with pytest.raises(mi.HypreError, match="not implemented"):
            mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)
        mi.call("HYPRE_ClearAllErrors")


def test_direct_solve_as_down_or_up_smoother_is_refused_at_setup(mi_lib):
    """relax type 9 exists for the coarsest level only: SetCycleRelaxType(9, 1 or 2) is refused at Setup (ADVICE r2:
    it used to be replaced by relax_type[0] on the other levels, or to throw inside the Krylov loop); as the
    coarsest-level choice (k = 3, what HYPRE_BoomerAMGSetRelaxType selects by itself) it is accepted."""
    mi = mi_lib
    A, rhs = mi.build_laplace_system_host(6, 6, 6, 7, 0, 1)
    for k in (1, 2):
        amg = mi.BoomerAMG(print_level=0)
        mi.call("HYPRE_BoomerAMGSetCycleRelaxType", amg.h, 9, k)
        with pytest.raises(mi.HypreError, match="coarsest level only"):
            mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)
        mi.call("HYPRE_ClearAllErrors")
    amg = mi.BoomerAMG(print_level=0)
    mi.call("HYPRE_BoomerAMGSetCycleRelaxType", amg.h, 9, 3)
    mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)
    assert amg.num_levels >= 2


def test_random_mmatrix_coarsening_types_host(mi_lib, oc):
    """Irregular graph (random M-matrix): every coarsening type and aggressive coarsening against the oracle."""
    mi = mi_lib
    rng = np.random.default_rng(5)
    n = 600
    M = sp.random(n, n, density=0.01, random_state=rng, format="csr")
    M = (M + M.T).tocsr()
    M = (M - sp.diags(M.diagonal())).tocsr()
    M = (-abs(M) + sp.diags(abs(M).sum(axis=1).A1 + 0.1)).tocsr()
    M.sort_indices()
    coo = M.tocoo()
    for kw in (dict(coarsen_type=10), dict(coarsen_type=6), dict(agg_num_levels=1), dict(agg_num_levels=1, coarsen_type=6),
               dict(strong_threshold=0.25, coarsen_type=6, interp_type=0), dict(coarsen_type=0),
               dict(coarsen_type=0, strong_threshold=0.25, interp_type=0), dict(coarsen_type=0, agg_num_levels=1)):
        A = mi.IJMatrix.__new__(mi.IJMatrix)
        A.h = mi.vp()
        mi.call("HYPRE_IJMatrixCreate", 0, mi.c_big(0), mi.c_big(n - 1), mi.c_big(0), mi.c_big(n - 1), mi.C.byref(A.h))
        A.par = mi.vp()
        mi.call("HYPRE_IJMatrixGetObject", A.h, mi.C.byref(A.par))
        A.set_values_coo(coo.row.astype(np.int64), coo.col.astype(np.int64), coo.data)
        mi.call("HYPRE_MI_IJMatrixAssembleHostOnly", A.h)
        amg = mi.BoomerAMG(print_level=0, **kw)
        mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)
        oamg = oc.Amg(oc.Csr.from_scipy(M), oc.default_params(**kw))
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


@pytest.mark.parametrize("n,stencil,ng", [(14, 7, dict(non_galerkin_tol=0.05)),
                                          (10, 27, dict(non_galerkin_tol=0.1)),
                                          (14, 7, dict(non_galerkin_tol=0.0, non_galerkin_level_tols=dict(levels=[1, 2], tolerances=[0.1, 0.3])))])
def test_non_galerkin_coarse_operators_match_oracle_host(mi_lib, oc, n, stencil, ng):
    """Non-Galerkin coarse operators (/root/reference/src/HypreSystem.cpp:161-176: non_galerkin_tol and the level-specific
    tolerances): the documented drop-and-lump rule shared with the oracle -- hierarchies bit-identical; the sparsified
    operators are smaller than the Galerkin ones, keep their row sums and stay symmetric."""
    import scipy.sparse as sp

    mi = mi_lib
    A, rhs = mi.build_laplace_system_host(n, n, n, stencil, 0, 1)
    amg = mi.BoomerAMG(print_level=0, **ng)
    mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)
    gal = mi.BoomerAMG(print_level=0)
    mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", gal.h, A.par)
    Ao, bo = oc.Csr.laplace(n, n, n, stencil)
    tols = [ng["non_galerkin_tol"]] * 8
    for lev, t in zip(*(ng.get("non_galerkin_level_tols", dict(levels=[], tolerances=[])).values())):
        tols[lev] = t
    oamg = oc.Amg(Ao, oc.default_params(non_galerkin_tol=tols))
    assert amg.num_levels == oamg.num_levels and amg.num_levels >= 3
    for l in range(amg.num_levels):
        ia, ja, a, shape = amg.level_csr(l, 0)
        oia, oja, oa = oamg.level_A(l).arrays()
        assert np.array_equal(ia, oia) and np.array_equal(ja, oja) and np.array_equal(a, oa), l
    assert amg.operator_complexity < gal.operator_complexity
    # level 1 of the sparsified hierarchy against the Galerkin product of ITS OWN level 0: same row sums, symmetric
    ia, ja, a, shape = amg.level_csr(1, 0)
    A1 = sp.csr_matrix((a, ja, ia), shape=shape)
    pia, pja, pa, pshape = amg.level_csr(0, 2)
    P = sp.csr_matrix((pa, pja, pia), shape=pshape)
    a0 = amg.level_csr(0, 0)
    A0 = sp.csr_matrix((a0[2], a0[1], a0[0]), shape=a0[3])
    G = (P.T @ A0 @ P).tocsr()
    if tols[0] > 0:
        assert A1.nnz < G.nnz
        assert np.abs(A1.sum(axis=1) - G.sum(axis=1)).max() < 1e-12 and abs(A1 - A1.T).max() < 1e-14
    else:
        assert A1.nnz == G.nnz
