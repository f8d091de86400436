"""Parity of the HIP kernels against the CPU oracle, through the C ABI.

Tolerances: SpMV / BLAS-1 / relaxation results are sums of <= a few hundred
fp64 products whose association differs between the wave/LDS reductions and the
oracle's sequential loops, so they must agree to 1e-13 relative to the magnitude
of the terms (stated per test).
"""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _rand_dd_matrix(n, density, seed, longrow=None):
    """Random diagonally dominant CSR with ragged rows (incl. empty off-diagonals)."""
    rng = np.random.default_rng(seed)
    M = sp.random(n, n, density=density, random_state=rng, format="lil")
    if longrow is not None:
        r, cnt = longrow
        cols = rng.choice(n, size=cnt, replace=False)
        for c in cols:
            M[r, c] = rng.standard_normal()
    M = M.tocsr()
    M.setdiag(0.0)
    M.eliminate_zeros()
    d = np.abs(M).sum(axis=1).A1 + 1.0
    M = (M + sp.diags(d)).tocsr()
    M.sort_indices()
    return M


def _ij_from_scipy(mi, M):
    n = M.shape[0]
    A = mi.IJMatrix(0, n - 1)
    coo = M.tocoo()
    A.set_values_coo(coo.row.astype(np.int64), coo.col.astype(np.int64), coo.data)
    A.assemble()
    return A


@pytest.mark.parametrize("n,density,longrow", [(1, 1.0, None), (7, 0.5, None), (1000, 0.01, None),
                                                (5000, 0.002, (17, 3000)), (20000, 0.0005, None),
                                                # >= 100 entries per row: the 4096-entry tiles run by 512 threads
                                                (3000, 0.05, None), (2500, 0.06, (40, 2400)), (1500, 0.3, None)])
def test_spmv_vs_oracle(mi, oc, n, density, longrow):
    M = _rand_dd_matrix(n, density, 1234 + n, longrow)
    A = _ij_from_scipy(mi, M)
    rng = np.random.default_rng(n)
    xv, yv = rng.standard_normal(n), rng.standard_normal(n)
    x = mi.IJVector(0, n - 1, xv)
    y = mi.IJVector(0, n - 1, yv)
    mi.call("HYPRE_ParCSRMatrixMatvec", -1.5, A.par, x.par, 0.75, y.par)
    got = y.get()
    Ao = oc.Csr.from_scipy(M)
    ref = Ao.matvec(xv, alpha=-1.5, beta=0.75, b=yv)
    scale = (abs(M) @ np.abs(xv)) * 1.5 + np.abs(yv)
    assert np.all(np.abs(got - ref) <= 1e-13 * (scale + 1.0))


def test_spmv_laplace_bitwise(mi, oc):
    """Short rows take the one-lane-per-row reduction whose order equals the oracle's: bit-exact."""
    n = 24
    A, b, x, rhs = mi.build_laplace_system(n, n, n, 7)
    Ao, bo = oc.Csr.laplace(n, n, n, 7)
    assert np.array_equal(rhs, bo)
    rng = np.random.default_rng(5)
    xv = rng.standard_normal(n ** 3)
    xi = mi.IJVector(0, n ** 3 - 1, xv)
    yi = mi.IJVector(0, n ** 3 - 1, np.zeros(n ** 3))
    mi.call("HYPRE_ParCSRMatrixMatvec", 1.0, A.par, xi.par, 0.0, yi.par)
    assert np.array_equal(yi.get(), Ao.matvec(xv))


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 511, 100003])
def test_blas1(mi, n):
    rng = np.random.default_rng(n)
    xv, yv = rng.standard_normal(n), rng.standard_normal(n)
    x = mi.IJVector(0, n - 1, xv)
    y = mi.IJVector(0, n - 1, yv)
    prod = mi.c_dbl()
    mi.call("HYPRE_ParVectorInnerProd", x.par, y.par, mi.C.byref(prod))
    assert abs(prod.value - float(xv @ yv)) <= 1e-13 * float(np.abs(xv) @ np.abs(yv)) + 1e-300
    mi.call("HYPRE_ParVectorAxpy", 0.3, x.par, y.par)
    assert np.allclose(y.get(), yv + 0.3 * xv, rtol=1e-15, atol=1e-15)
    mi.call("HYPRE_ParVectorScale", -2.0, y.par)
    assert np.allclose(y.get(), -2.0 * (yv + 0.3 * xv), rtol=1e-15, atol=1e-15)
    mi.call("HYPRE_ParVectorCopy", x.par, y.par)
    assert np.array_equal(y.get(), xv)
    y.fill(3.25)
    assert np.all(y.get() == 3.25)


def test_ij_set_add_semantics(mi):
    """Set overwrites, AddTo accumulates, duplicates are folded in submission order."""
    n = 6
    A = mi.IJMatrix(0, n - 1)
    rows = np.array([0, 0, 1, 5, 0], dtype=np.int64)
    cols = np.array([0, 3, 1, 5, 0], dtype=np.int64)
    vals = np.array([1.0, 2.0, 3.0, 4.0, 10.0])
    A.set_values_coo(rows, cols, vals)                       # (0,0) set twice: last wins = 10
    A.set_values_coo(np.array([0, 2], dtype=np.int64), np.array([3, 2], dtype=np.int64), np.array([5.0, 7.0]),
                     add=True)                               # (0,3): 2 + 5
    A.assemble()
    dense = np.zeros((n, n))
    for j in range(n):
        e = np.zeros(n)
        e[j] = 1.0
        x = mi.IJVector(0, n - 1, e)
        y = mi.IJVector(0, n - 1, np.zeros(n))
        mi.call("HYPRE_ParCSRMatrixMatvec", 1.0, A.par, x.par, 0.0, y.par)
        dense[:, j] = y.get()
    want = np.zeros((n, n))
    want[0, 0], want[0, 3], want[1, 1], want[5, 5], want[2, 2] = 10.0, 7.0, 3.0, 4.0, 7.0
    assert np.array_equal(dense, want)


def test_vector_get_set_indices(mi):
    n = 50
    v = mi.IJVector(10, 10 + n - 1, np.arange(n, dtype=float))
    idx = np.array([10, 59, 33], dtype=np.int64)
    out = np.zeros(3)
    mi.call("HYPRE_IJVectorGetValues", v.h, 3, idx, out)
    assert np.array_equal(out, [0.0, 49.0, 23.0])
    mi.call("HYPRE_IJVectorAddToValues", v.h, 3, idx, np.array([1.0, 1.0, 1.0]))
    mi.call("HYPRE_IJVectorGetValues", v.h, 3, idx, out)
    assert np.array_equal(out, [1.0, 50.0, 24.0])
    bad = np.array([9], dtype=np.int64)
    with pytest.raises(mi.HypreError):
        mi.call("HYPRE_IJVectorGetValues", v.h, 1, bad, out)
    mi.call("HYPRE_ClearAllErrors")


def test_rccl_transport_single_rank(mi):
    """The RCCL transport cannot be run with two ranks on a one-GPU box (RCCL refuses a
    duplicate device); this drives every RCCL entry point it uses in a world of one."""
    mi.call("HYPRE_MI_CommSelfTestRCCL")


@pytest.mark.parametrize("seed,rounds,max_block", [(1, 3000, 1 << 20), (2, 1500, 1 << 28), (3, 500, 1 << 30)])
def test_device_arena_allocation_storm(mi, seed, rounds, max_block):
    """Round 4: every device allocation of the library comes out of one growable arena (HIP virtual-memory API, best-fit
    free list with coalescing, chunks unmapped from the top at a trim, a grow-ahead thread).  A seeded storm of
    allocations and releases from bytes to gigabytes, every live block filled with its own byte and checked before its
    release, trims in the middle: a block handed out twice, an overlap or a trim of live memory would show as a wrong
    byte or a fault."""
    ok, peak = mi.C.c_longlong(), mi.C.c_longlong()
    mi.call("HYPRE_MI_ArenaSelfTest", seed, rounds, mi.C.c_longlong(max_block), mi.C.byref(ok), mi.C.byref(peak))
    assert ok.value > rounds // 8 and peak.value > 0
    v = mi.C.c_longlong()
    mi.call("HYPRE_MI_GetCounter", b"arena_in_use_bytes", mi.C.byref(v))
    # what the storm allocated is gone again (other tests' objects may still hold memory)
    before = v.value
    mi.call("HYPRE_MI_ArenaSelfTest", seed + 10, 200, mi.C.c_longlong(1 << 16), mi.C.byref(ok), mi.C.byref(peak))
    mi.call("HYPRE_MI_GetCounter", b"arena_in_use_bytes", mi.C.byref(v))
    assert v.value == before



@pytest.mark.parametrize("case", ["short", "long_wide", "spmv_only", "with_giants", "empty_rows", "ragged_end", "tiny"])
def test_device_tile_schedule_equals_the_host_loop(mi, case):
    """The setup cuts every operator into tiles on the device, one thread per super-block of 8192 rows, each tile end found
    by bisection on the row pointers (k::tile_end_bisect); the host routine (small operators, and the definition) walks
    chunk by chunk / row by row (k::tile_end).  Same tiles, same chunk-alignment flag, on seeded row lengths of every
    kind the hierarchy produces: 3-8 entries per row, 100-300 with 4096-entry tiles, SpMV-only operators (up to 1024 rows
    per tile, row granularity), rows longer than a tile, empty rows, a row count that is no multiple of 8 or 8192."""
    rng = np.random.default_rng({"short": 1, "long_wide": 2, "spmv_only": 3, "with_giants": 4, "empty_rows": 5,
                                 "ragged_end": 6, "tiny": 7}[case])
    row_cap, tile = 256, 2048
    if case == "short":
        lens = rng.integers(3, 9, size=200_003)
    elif case == "long_wide":
        lens, row_cap, tile = rng.integers(100, 301, size=40_000), 512, 4096
    elif case == "spmv_only":
        lens, row_cap = rng.integers(1, 5, size=150_001), 1024
    elif case == "with_giants":
        lens = rng.integers(5, 60, size=60_000)
        lens[rng.integers(0, lens.size, size=200)] = rng.integers(2048, 9000, size=200)
    elif case == "empty_rows":
        lens = rng.integers(0, 3, size=90_000) * rng.integers(0, 40, size=90_000)
    elif case == "ragged_end":
        lens = rng.integers(20, 90, size=8192 * 3 + 5)
    else:
        lens = rng.integers(1, 10, size=13)
    ia = np.zeros(lens.size + 1, dtype=np.int64)
    np.cumsum(lens, out=ia[1:])
    nt, bad = mi.c_int(), mi.c_int()
    mi.call("HYPRE_MI_TileScheduleCheck", int(lens.size), ia.ctypes.data_as(mi.C.POINTER(mi.C.c_longlong)), row_cap, tile,
            mi.C.byref(nt), mi.C.byref(bad))
    assert bad.value == 0, f"first differing tile (1-based): {bad.value} of {nt.value}"
    assert nt.value >= max(1, lens.size // 1024)
