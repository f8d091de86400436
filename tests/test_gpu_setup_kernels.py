"""Device kernels of the AMG setup phase (csrc/setup_kernels.hip) against the oracle and scipy.

The bar is BIT equality: the sparse products accumulate every entry in the oracle's order with
individually rounded multiplies and adds, transposes and permutations only move values.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _rand_csr(rng, n, m, per_row, empty_every=0):
    rows, cols = [], []
    for i in range(n):
        if empty_every and i % empty_every == 0:
            continue
        k = min(m, max(1, int(rng.integers(max(1, per_row // 2), per_row + per_row // 2 + 1))))
        c = rng.choice(m, size=k, replace=False)
        rows += [i] * k
        cols += list(c)
    vals = rng.standard_normal(len(rows))
    M = sp.csr_matrix((vals, (rows, cols)), shape=(n, m))
    M.sort_indices()
    return M


def _oracle_matmul(oc, A, B):
    Ao, Bo = oc.Csr.from_scipy(A), oc.Csr.from_scipy(B)
    C = oc.Csr(oc.lib().ocsr_matmul(Ao.h, Bo.h))
    return C.arrays()


@pytest.mark.parametrize("shape", [
    (3000, 2500, 4000, 5, 4),       # short rows: 8-lane groups
    (2000, 1500, 3000, 9, 10),      # 16-lane groups
    (900, 800, 2000, 16, 20),       # one wave per row
    (300, 400, 3000, 40, 50),       # one block per row, LDS table
    (40, 3000, 20000, 200, 60),     # one block per row, table in global scratch
    (500, 300, 64, 30, 20),         # few columns: bound = ncols
])
def test_spgemm_bit_exact(mi, oc, shape):
    n, k, m, pa, pb = shape
    rng = np.random.default_rng(n + k)
    A = _rand_csr(rng, n, k, pa, empty_every=17)
    B = _rand_csr(rng, k, m, pb, empty_every=13)
    ia, ja, a, shp = mi.csr_device_op(0, A, B)
    oia, oja, oa = _oracle_matmul(oc, A, B)
    assert shp == (n, m)
    assert np.array_equal(ia, oia)
    assert np.array_equal(ja, oja)
    assert np.array_equal(a.view(np.int64), oa.view(np.int64))  # bit for bit


def test_spgemm_mixed_bins_and_cancellation(mi, oc):
    """Rows of very different lengths in one product; values chosen so that sums cancel (order matters)."""
    rng = np.random.default_rng(5)
    blocks = [_rand_csr(rng, 400, 1200, p) for p in (2, 12, 60, 300)]
    A = sp.vstack(blocks).tocsr()
    A.sort_indices()
    B = _rand_csr(rng, 1200, 5000, 25)
    B.data = np.where(rng.random(B.nnz) < 0.5, 1e16, 1.0) * np.sign(B.data)
    ia, ja, a, _ = mi.csr_device_op(0, A, B)
    oia, oja, oa = _oracle_matmul(oc, A, B)
    assert np.array_equal(ia, oia) and np.array_equal(ja, oja)
    assert np.array_equal(a.view(np.int64), oa.view(np.int64))


def test_transpose_and_permute(mi):
    rng = np.random.default_rng(11)
    for n, m, per in ((5000, 1300, 4), (700, 900, 70), (64, 4000, 900)):
        A = _rand_csr(rng, n, m, per, empty_every=9)
        ia, ja, a, shp = mi.csr_device_op(1, A)
        T = A.T.tocsr()
        T.sort_indices()
        assert shp == (m, n)
        assert np.array_equal(ia, T.indptr) and np.array_equal(ja, T.indices)
        assert np.array_equal(a.view(np.int64), T.data.view(np.int64))
        perm = rng.permutation(n).astype(np.int32)
        colpos = rng.permutation(m).astype(np.int32)
        for pp, cp in ((perm, colpos), (None, colpos), (perm, None)):
            ia, ja, a, shp = mi.csr_device_op(2, A, perm=pp, colpos=cp)
            R = A if pp is None else A[pp]
            if cp is not None:
                inv = np.empty(m, dtype=np.int64)
                inv[cp] = np.arange(m)
                R = R[:, inv]  # column c of A lands at colpos[c]
            R = R.tocsr()
            R.sort_indices()
            assert np.array_equal(ia, R.indptr) and np.array_equal(ja, R.indices)
            assert np.array_equal(a.view(np.int64), R.data.view(np.int64))


def _assert_same_hierarchy(amg, oamg):
    assert amg.num_levels == oamg.num_levels
    for l in range(amg.num_levels):
        ia, ja, a, _ = amg.level_csr(l, 0)
        oia, oja, oa = oamg.level_A(l).arrays()
        assert np.array_equal(ia, oia) and np.array_equal(ja, oja), l
        assert np.array_equal(a.view(np.int64), oa.view(np.int64)), l
        if l < amg.num_levels - 1:
            assert np.array_equal(amg.level_cf(l), oamg.level_cf(l)), l
            pia, pja, pa, _ = amg.level_csr(l, 2)
            oia, oja, oa = oamg.level_P(l).arrays()
            assert np.array_equal(pia, oia) and np.array_equal(pja, oja), l
            assert np.array_equal(pa.view(np.int64), oa.view(np.int64)), l
            ria, rja, ra, _ = amg.level_csr(l, 3)
            PT = oamg.level_P(l).to_scipy().T.tocsr()
            PT.sort_indices()
            assert np.array_equal(ria, PT.indptr) and np.array_equal(rja, PT.indices), l
            assert np.array_equal(ra.view(np.int64), PT.data.view(np.int64)), l


@pytest.mark.parametrize("n,stencil,kw", [
    (16, 7, {}),
    (14, 27, {}),
    (14, 7, dict(interp_type=0)),                                  # classical modified
    (12, 27, dict(trunc_factor=0.2, true_pmax_elmts=0)),           # relative truncation only
    (16, 7, dict(strong_threshold=0.25, true_pmax_elmts=6)),
    (12, 7, dict(interp_type=3)),                                  # direct: host routine inside the device setup
    # host coarsening (Ruge-Stueben family, aggressive) between the device's strength graph and its Galerkin product
    (16, 7, dict(coarsen_type=10)),                                # HMIS
    (14, 7, dict(coarsen_type=6, interp_type=0)),                  # Falgout + classical: the upstream sample
    (12, 27, dict(coarsen_type=6)),
    (18, 7, dict(agg_num_levels=1)),                               # A2 aggressive coarsening + multipass
    (16, 7, dict(agg_num_levels=2, agg_pmax_elmts=4)),
    (12, 27, dict(agg_num_levels=1, coarsen_type=10)),
    # non-Galerkin coarse operators (src/HypreSystem.cpp:161-176): the device's drop-and-lump pass after the product
    (16, 7, dict(non_galerkin_tol=0.05)),
    (12, 27, dict(non_galerkin_tol=0.0, non_galerkin_level_tols=dict(levels=[1, 2, 3], tolerances=[0.05, 0.1, 0.2]))),
])
def test_device_setup_hierarchy_is_bit_identical(mi, oc, n, stencil, kw, monkeypatch):
    """Strength, PMIS, interpolation, Galerkin products, transposes and the C-first renumbering of EVERY level on
    the device: the hierarchy equals the oracle's bit for bit."""
    monkeypatch.setenv("MI_HYPRE_DEVICE_SETUP_MIN_ROWS", "0")
    A, b, x, rhs = mi.build_laplace_system(n, n, n, stencil)
    amg = mi.BoomerAMG(print_level=0, **kw)
    amg.setup(A)
    Ao, bo = oc.Csr.laplace(n, n, n, stencil)
    chunk = mi.c_int()
    mi.call("HYPRE_MI_GetGSChunk", mi.C.byref(chunk))
    okw = {("pmax_elmts" if k == "true_pmax_elmts" else k): v for k, v in kw.items()}
    if "non_galerkin_tol" in okw:  # the oracle takes one tolerance per fine level (HYPRE's index)
        tols = [okw["non_galerkin_tol"]] * 12
        lt = okw.pop("non_galerkin_level_tols", None)
        for lev, t in zip(*(lt.values() if lt else ([], []))):
            tols[lev] = t
        okw["non_galerkin_tol"] = tols
    oamg = oc.Amg(Ao, oc.default_params(gs_chunk=chunk.value, **okw))
    assert amg.num_levels > 1
    _assert_same_hierarchy(amg, oamg)


def test_device_setup_random_mmatrix(mi, oc, monkeypatch):
    """Irregular rows (the interpolation kernel's larger group sizes) on a random M-matrix."""
    monkeypatch.setenv("MI_HYPRE_DEVICE_SETUP_MIN_ROWS", "0")
    rng = np.random.default_rng(21)
    n = 3000
    B = _rand_csr(rng, n, n, 14)
    B = (B + B.T).tocsr()
    B.data = -np.abs(B.data)
    B.setdiag(0)
    B.eliminate_zeros()
    d = -np.asarray(B.sum(axis=1)).ravel() * (1.0 + 0.05 * rng.random(n)) + 1e-3
    M = (B + sp.diags(d)).tocsr()
    M.sort_indices()
    A = mi.IJMatrix(0, n - 1)
    coo = M.tocoo()
    A.set_values_coo(coo.row.astype(np.int64), coo.col.astype(np.int64), coo.data.astype(np.float64))
    A.assemble()
    amg = mi.BoomerAMG(print_level=0, strong_threshold=0.25)
    amg.setup(A)
    chunk = mi.c_int()
    mi.call("HYPRE_MI_GetGSChunk", mi.C.byref(chunk))
    oamg = oc.Amg(oc.Csr.from_scipy(M), oc.default_params(gs_chunk=chunk.value, strong_threshold=0.25))
    assert amg.num_levels > 2
    _assert_same_hierarchy(amg, oamg)
