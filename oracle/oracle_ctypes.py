"""ctypes view of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product package.  PARITY UNPINNED (see oracle/oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


class AmgParams(C.Structure):
    _fields_ = [
        ("coarsen_type", C.c_int),
        ("interp_type", C.c_int),
        ("strong_threshold", C.c_double),
        ("max_row_sum", C.c_double),
        ("trunc_factor", C.c_double),
        ("pmax_elmts", C.c_int),
        ("max_levels", C.c_int),
        ("max_coarse_size", C.c_int),
        ("min_coarse_size", C.c_int),
        ("relax_type", C.c_int * 3),
        ("num_sweeps", C.c_int * 3),
        ("relax_order", C.c_int),
        ("relax_weight", C.c_double),
        ("outer_weight", C.c_double),
        ("cycle_type", C.c_int),
        ("gs_chunk", C.c_int),
        ("nparts", C.c_int),
        ("part_starts", C.POINTER(C.c_longlong)),
        ("max_iter", C.c_int),
        ("tol", C.c_double),
        ("redundant_rows", C.c_longlong),
        ("agg_num_levels", C.c_int),
        ("agg_interp_type", C.c_int),
        ("agg_pmax_elmts", C.c_int),
        ("agg_trunc_factor", C.c_double),
        ("smooth_type", C.c_int),
        ("smooth_num_levels", C.c_int),
        ("ilu_max_iter", C.c_int),
        ("ilu_tri_solve", C.c_int),
        ("ilu_lower_it", C.c_int),
        ("ilu_upper_it", C.c_int),
        ("non_galerkin_num_tol", C.c_int),
        ("non_galerkin_tol", C.POINTER(C.c_double)),
        ("ilu_level", C.c_int),
    ]


class KrylovResult(C.Structure):
    _fields_ = [("iters", C.c_int), ("converged", C.c_int), ("rel_res", C.c_double), ("true_rel_res", C.c_double)]


class _Csr(C.Structure):
    _fields_ = [
        ("nrows", C.c_int),
        ("ncols", C.c_int),
        ("ia", C.POINTER(C.c_longlong)),
        ("ja", C.POINTER(C.c_int)),
        ("a", C.POINTER(C.c_double)),
    ]


PRECOND_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        P = C.POINTER
        L.ocsr_from_arrays.restype = P(_Csr)
        L.ocsr_from_arrays.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ocsr_free.argtypes = [P(_Csr)]
        L.ocsr_nnz.restype = C.c_longlong
        L.ocsr_nnz.argtypes = [P(_Csr)]
        L.ocsr_matvec.argtypes = [C.c_double, P(_Csr), C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        L.ocsr_transpose.restype = P(_Csr)
        L.ocsr_transpose.argtypes = [P(_Csr)]
        L.ocsr_matmul.restype = P(_Csr)
        L.ocsr_matmul.argtypes = [P(_Csr), P(_Csr)]
        L.oracle_laplace.restype = P(_Csr)
        L.oracle_laplace.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_rand_seed.argtypes = [C.c_int]
        L.oracle_rand.restype = C.c_double
        L.oamg_default_params.argtypes = [P(AmgParams)]
        L.oamg_setup.restype = C.c_void_p
        L.oamg_setup.argtypes = [P(_Csr), P(AmgParams)]
        L.oamg_from_levels.restype = C.c_void_p
        L.oamg_from_levels.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, P(AmgParams)]
        L.oamg_free.argtypes = [C.c_void_p]
        L.oamg_num_levels.argtypes = [C.c_void_p]
        for nm in ("oamg_A", "oamg_P"):
            getattr(L, nm).restype = P(_Csr)
            getattr(L, nm).argtypes = [C.c_void_p, C.c_int]
        L.oamg_cf.restype = P(C.c_int)
        L.oamg_cf.argtypes = [C.c_void_p, C.c_int]
        L.oamg_l1.restype = P(C.c_double)
        L.oamg_l1.argtypes = [C.c_void_p, C.c_int]
        L.oamg_perm.restype = P(C.c_int)
        L.oamg_perm.argtypes = [C.c_void_p, C.c_int]
        L.oamg_part_starts.restype = P(C.c_longlong)
        L.oamg_part_starts.argtypes = [C.c_void_p, C.c_int]
        L.oamg_relax.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oamg_cycle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oamg_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, P(C.c_int), P(C.c_double)]
        L.oamg_precond.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ogmres_solve.argtypes = [P(_Csr), C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int,
                                   C.c_void_p, C.c_void_p, P(KrylovResult), C.c_void_p]
        L.ofgmres_solve.argtypes = L.ogmres_solve.argtypes
        L.ocogmres_solve.argtypes = [P(_Csr), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                     C.c_void_p, C.c_void_p, P(KrylovResult), C.c_void_p]
        L.opcg_solve.argtypes = [P(_Csr), C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                 C.c_void_p, C.c_void_p, P(KrylovResult), C.c_void_p]
        L.obicgstab_solve.argtypes = [P(_Csr), C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                      C.c_void_p, C.c_void_p, P(KrylovResult), C.c_void_p]
        L.oilu_setup.restype = C.c_void_p
        L.oilu_setup.argtypes = [P(_Csr), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.oilu_setup_k.restype = C.c_void_p
        L.oilu_setup_k.argtypes = [P(_Csr), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oilu_free.argtypes = [C.c_void_p]
        L.oilu_factor.restype = P(_Csr)
        L.oilu_factor.argtypes = [C.c_void_p]
        L.oilu_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oilu_solve.restype = C.c_int
        L.oilu_solve.argtypes = [C.c_void_p, P(_Csr), C.c_void_p, C.c_void_p, C.c_int, C.c_double, P(C.c_double)]
        L.oracle_set_threads.argtypes = [C.c_int]
        L.omulti_precond.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Csr:
    """Owning handle on an ocsr; numpy views are copies."""

    def __init__(self, handle, own=True):
        self.h = handle
        self.own = own

    @classmethod
    def from_scipy(cls, M):
        M = M.tocsr()
        M.sort_indices()
        ia = np.ascontiguousarray(M.indptr, dtype=np.int64)
        ja = np.ascontiguousarray(M.indices, dtype=np.int32)
        a = np.ascontiguousarray(M.data, dtype=np.float64)
        return cls(lib().ocsr_from_arrays(M.shape[0], M.shape[1], _ptr(ia), _ptr(ja), _ptr(a)))

    @classmethod
    def laplace(cls, nx, ny, nz, stencil=7):
        rhs = np.zeros(nx * ny * nz)
        h = lib().oracle_laplace(nx, ny, nz, stencil, _ptr(rhs))
        return cls(h), rhs

    @property
    def shape(self):
        return (self.h.contents.nrows, self.h.contents.ncols)

    @property
    def nnz(self):
        return lib().ocsr_nnz(self.h)

    def arrays(self):
        n, nnz = self.shape[0], self.nnz
        ia = np.ctypeslib.as_array(self.h.contents.ia, shape=(n + 1,)).copy()
        ja = np.ctypeslib.as_array(self.h.contents.ja, shape=(max(nnz, 1),))[:nnz].copy()
        a = np.ctypeslib.as_array(self.h.contents.a, shape=(max(nnz, 1),))[:nnz].copy()
        return ia, ja, a

    def to_scipy(self):
        import scipy.sparse as sp

        ia, ja, a = self.arrays()
        return sp.csr_matrix((a, ja, ia), shape=self.shape)

    def matvec(self, x, alpha=1.0, beta=0.0, b=None):
        y = np.empty(self.shape[0])
        x = np.ascontiguousarray(x, dtype=np.float64)
        lib().ocsr_matvec(alpha, self.h, _ptr(x), beta, _ptr(b), _ptr(y))
        return y

    def __del__(self):
        if self.own and self.h:
            lib().ocsr_free(self.h)
            self.h = None


def default_params(**kw):
    p = AmgParams()
    lib().oamg_default_params(C.byref(p))
    keep = []
    for k, v in kw.items():
        if k in ("relax_type", "num_sweeps"):
            if isinstance(v, int):
                v = (v, v, getattr(p, k)[2])
            for i in range(3):
                getattr(p, k)[i] = v[i]
        elif k == "non_galerkin_tol":  # a number (every level) or a list indexed by the fine level (HYPRE's index)
            arr = np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64)
            keep.append(arr)
            p.non_galerkin_tol = arr.ctypes.data_as(C.POINTER(C.c_double))
            p.non_galerkin_num_tol = len(arr)
        elif k == "part_starts":
            arr = np.ascontiguousarray(v, dtype=np.int64)
            keep.append(arr)
            p.part_starts = arr.ctypes.data_as(C.POINTER(C.c_longlong))
            p.nparts = len(arr) - 1
        else:
            setattr(p, k, v)
    p._keep = keep
    return p


class Amg:
    def __init__(self, A, params=None, handle=None, keep=None):
        self.A = A
        self.params = params or default_params()
        self._keep = keep
        self.h = handle if handle is not None else lib().oamg_setup(A.h, C.byref(self.params))

    @classmethod
    def from_levels(cls, As, Ps, cfs, params, part_starts=None):
        """Run the oracle's solve phase on an externally built hierarchy."""
        P_ = C.POINTER(_Csr)
        nlev = len(As)
        Aarr = (P_ * nlev)(*[a.h for a in As])
        Parr = (P_ * nlev)(*([p.h for p in Ps] + [None] * (nlev - len(Ps))))
        cfk = [np.ascontiguousarray(c, dtype=np.int32) for c in cfs]
        cfarr = (C.c_void_p * nlev)(*([c.ctypes.data for c in cfk] + [None] * (nlev - len(cfk))))
        psk = None
        psarr = None
        if part_starts is not None:
            psk = [np.ascontiguousarray(s, dtype=np.int64) for s in part_starts]
            psarr = (C.c_void_p * nlev)(*[s.ctypes.data for s in psk])
        h = lib().oamg_from_levels(nlev, Aarr, Parr, cfarr, psarr, C.byref(params))
        return cls(As[0], params, handle=h, keep=(As, Ps, cfk, psk))

    @property
    def num_levels(self):
        return lib().oamg_num_levels(self.h)

    def level_A(self, l):
        return Csr(lib().oamg_A(self.h, l), own=False)

    def level_P(self, l):
        return Csr(lib().oamg_P(self.h, l), own=False)

    def level_cf(self, l):
        n = self.level_A(l).shape[0]
        return np.ctypeslib.as_array(lib().oamg_cf(self.h, l), shape=(n,)).copy()

    def level_l1(self, l):
        n = self.level_A(l).shape[0]
        return np.ctypeslib.as_array(lib().oamg_l1(self.h, l), shape=(n,)).copy()

    def level_perm(self, l):
        """perm[new] = old row of the level's C-first ordering (identity on the coarsest level)."""
        n = self.level_A(l).shape[0]
        p = lib().oamg_perm(self.h, l)
        if not p:
            return np.arange(n, dtype=np.int32)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def level_part_starts(self, l):
        return np.ctypeslib.as_array(lib().oamg_part_starts(self.h, l), shape=(self.params.nparts + 1,)).copy()

    def relax(self, level, rtype, points, f, u):
        f = np.ascontiguousarray(f, dtype=np.float64)
        u = np.array(u, dtype=np.float64)
        lib().oamg_relax(self.h, level, rtype, points, _ptr(f), _ptr(u))
        return u

    def cycle(self, f, u=None):
        f = np.ascontiguousarray(f, dtype=np.float64)
        u = np.zeros_like(f) if u is None else np.array(u, dtype=np.float64)
        lib().oamg_cycle(self.h, _ptr(f), _ptr(u))
        return u

    def solve(self, b, x=None):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b) if x is None else np.array(x, dtype=np.float64)
        it = C.c_int()
        rr = C.c_double()
        lib().oamg_solve(self.h, _ptr(b), _ptr(x), C.byref(it), C.byref(rr))
        return x, it.value, rr.value

    def __del__(self):
        if getattr(self, "h", None):
            lib().oamg_free(self.h)
            self.h = None


class Ilu:
    """Block-Jacobi ILU(k) of the oracle (part_starts: emulated rank partition; level_of_fill k, 0 = ILU(0))."""

    def __init__(self, A, part_starts=None, tri_solve=1, lower_it=5, upper_it=5, level_of_fill=0):
        self.A = A
        ps = None if part_starts is None else np.ascontiguousarray(part_starts, dtype=np.int64)
        self._ps = ps
        self.h = lib().oilu_setup_k(A.h, 0 if ps is None else len(ps) - 1, _ptr(ps), tri_solve, lower_it, upper_it, level_of_fill)
        self.is_ilu = True

    def factor(self):
        return Csr(lib().oilu_factor(self.h), own=False)

    def apply(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.zeros_like(r)
        lib().oilu_apply(self.h, _ptr(r), _ptr(z))
        return z

    def solve(self, b, x0=None, max_iter=20, tol=1e-7):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64)
        rel = C.c_double()
        it = lib().oilu_solve(self.h, self.A.h, _ptr(b), _ptr(x), max_iter, tol, C.byref(rel))
        return x, dict(iters=it, rel_res=rel.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().oilu_free(self.h)
            self.h = None


class _Multi(C.Structure):
    _fields_ = [("M", C.c_void_p), ("Mctx", C.c_void_p), ("n", C.c_int), ("ncomp", C.c_int)]


def _krylov(fn, A, b, x0, args, amg, maxit, ncomp=1):
    """ncomp > 1: A is the block system kron(I_ncomp, A_1), b / x are component-major multivectors and the
    preconditioner (built on A_1) is applied to every component."""
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64)
    res = KrylovResult()
    norms = np.full(maxit + 2, np.nan)
    if amg is not None and getattr(amg, "is_ilu", False):
        M = C.cast(lib().oilu_precond, C.c_void_p)
    else:
        M = C.cast(lib().oamg_precond, C.c_void_p) if amg is not None else None
    ctx = amg.h if amg is not None else None
    if ncomp > 1 and amg is not None:
        multi = _Multi(M, ctx, len(b) // ncomp, ncomp)
        M = C.cast(lib().omulti_precond, C.c_void_p)
        ctx = C.cast(C.pointer(multi), C.c_void_p)
    fn(A.h, _ptr(b), _ptr(x), *args, M, ctx, C.byref(res), _ptr(norms))
    return x, dict(iters=res.iters, converged=bool(res.converged), rel_res=res.rel_res,
                   true_rel_res=res.true_rel_res, norms=norms[: res.iters + 1].copy())


def gmres(A, b, x0=None, kdim=50, tol=1e-6, atol=0.0, maxit=100, amg=None, ncomp=1):
    return _krylov(lib().ogmres_solve, A, b, x0, (kdim, tol, atol, maxit), amg, maxit, ncomp)


def bicgstab(A, b, x0=None, tol=1e-6, atol=0.0, maxit=100, amg=None, ncomp=1):
    return _krylov(lib().obicgstab_solve, A, b, x0, (tol, atol, maxit), amg, maxit, ncomp)


def fgmres(A, b, x0=None, kdim=50, tol=1e-6, atol=0.0, maxit=100, amg=None, ncomp=1):
    return _krylov(lib().ofgmres_solve, A, b, x0, (kdim, tol, atol, maxit), amg, maxit, ncomp)


def cogmres(A, b, x0=None, kdim=50, cgs=0, tol=1e-6, atol=0.0, maxit=100, amg=None):
    return _krylov(lib().ocogmres_solve, A, b, x0, (kdim, cgs, tol, atol, maxit), amg, maxit)


def pcg(A, b, x0=None, tol=1e-6, atol=0.0, maxit=100, amg=None, ncomp=1):
    return _krylov(lib().opcg_solve, A, b, x0, (tol, atol, maxit), amg, maxit, ncomp)
