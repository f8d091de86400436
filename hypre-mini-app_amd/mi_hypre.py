"""ctypes view of libmi_hypre.so for tests/ and bench.py.

This is plumbing, not the product: the host side of the product is the C++
driver in host/ (HypreSystem + main, mirroring /root/reference/src).  Everything
here goes through the C ABI declared in include/*.h.  There is no fallback: if
the shared library is missing, or no HIP device is present at HYPRE_Init, this
raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI_HYPRE_LIB: another build of the same library (A/B timing of kernel variants on one box); never a fallback
LIB_PATH = os.environ.get("MI_HYPRE_LIB") or os.path.join(_HERE, "libmi_hypre.so")

HYPRE_PARCSR = 5555
HYPRE_MEMORY_DEVICE = 1
HYPRE_EXEC_DEVICE = 1
HYPRE_ERROR_CONV = 256

c_big = C.c_longlong
c_int = C.c_int
c_dbl = C.c_double
vp = C.c_void_p

PROF_SPMV_L0, PROF_RELAX_L0, PROF_DOT, PROF_AXPY = 0, 1, 2, 3
PROF_LEVELS = 16
PROF_LVL_RESID, PROF_LVL_RELAX, PROF_LVL_RESTRICT, PROF_LVL_PROLONG = 4, 4 + 16, 4 + 32, 4 + 48
PROF_LVL_RELAX0 = 4 + 64  # first sweep on a zero guess (runs on the level's zero-guess sub-operator)

ALLREDUCE_FN = C.CFUNCTYPE(None, vp, vp, C.c_size_t, c_int, c_int)
ALLGATHER_FN = C.CFUNCTYPE(None, vp, vp, vp, C.c_size_t)
EXCHANGE_FN = C.CFUNCTYPE(None, vp, c_int, C.POINTER(c_int), C.POINTER(vp), C.POINTER(C.c_size_t), c_int,
                          C.POINTER(c_int), C.POINTER(vp), C.POINTER(C.c_size_t))


class HypreError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the shared library (no compute happens here, so this works without a GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HypreError(
                f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                "(make -C hypre-mini-app_amd). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _lib.HYPRE_MI_LastErrorMessage.restype = C.c_char_p
        _lib.hypre_MAlloc.restype = vp
        _lib.hypre_MAlloc.argtypes = [C.c_size_t, c_int]
        _lib.hypre_Free.argtypes = [vp, c_int]
        _lib.hypre_Memcpy.argtypes = [vp, vp, C.c_size_t, c_int, c_int]
        _lib.HYPRE_MI_Free.argtypes = [vp]
    return _lib


def _conv(a):
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(vp)
    if isinstance(a, float):
        return c_dbl(a)
    if isinstance(a, (int, np.integer)):
        return a
    return a


def call(name, *args, allow=()):
    """Call an ABI entry point; raise on a non-zero HYPRE_Int unless allowed."""
    fn = getattr(lib(), name)
    rc = fn(*[_conv(a) for a in args])
    if rc != 0 and rc not in allow:
        msg = lib().HYPRE_MI_LastErrorMessage()
        raise HypreError(f"{name} returned {rc}: {msg.decode() if msg else ''}")
    return rc


def init():
    call("HYPRE_Init")
    call("HYPRE_SetMemoryLocation", HYPRE_MEMORY_DEVICE)
    call("HYPRE_SetExecutionPolicy", HYPRE_EXEC_DEVICE)


def finalize():
    call("HYPRE_Finalize")


def big(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def dbl(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def row_partition(total_rows, nproc, iproc):
    """init_row_decomposition, /root/reference/src/HypreSystem.cpp:525-544 (inclusive bounds)."""
    per, rem = divmod(total_rows, nproc)
    ilower = per * iproc + min(iproc, rem)
    iupper = per * (iproc + 1) + min(iproc + 1, rem) - 1
    return ilower, iupper


class IJMatrix:
    def __init__(self, ilower, iupper, jlower=None, jupper=None):
        jlower = ilower if jlower is None else jlower
        jupper = iupper if jupper is None else jupper
        self.h = vp()
        self.ilower, self.iupper = ilower, iupper
        call("HYPRE_IJMatrixCreate", 0, c_big(ilower), c_big(iupper), c_big(jlower), c_big(jupper), C.byref(self.h))
        call("HYPRE_IJMatrixSetObjectType", self.h, HYPRE_PARCSR)
        call("HYPRE_IJMatrixInitialize", self.h)
        self.par = vp()
        call("HYPRE_IJMatrixGetObject", self.h, C.byref(self.par))

    def set_values_coo(self, rows, cols, vals, add=False):
        """One entry per 'row', ncols == NULL -- the shape the driver uses (HypreSystem.cpp:942)."""
        n = len(vals) if not isinstance(vals, int) else None
        fn = "HYPRE_IJMatrixAddToValues2" if add else "HYPRE_IJMatrixSetValues2"
        if isinstance(rows, np.ndarray):
            rows, cols, vals = big(rows), big(cols), dbl(vals)
            # HYPRE_Int nrows: split calls that would overflow int32
            step = 1 << 30
            for s in range(0, len(vals), step):
                e = min(len(vals), s + step)
                call(fn, self.h, e - s, None, rows[s:e], None, cols[s:e], vals[s:e])
        else:
            raise TypeError("numpy arrays expected")
        return n

    def set_values_ptr(self, n, rows_ptr, cols_ptr, vals_ptr, add=False):
        fn = "HYPRE_IJMatrixAddToValues2" if add else "HYPRE_IJMatrixSetValues2"
        step = 1 << 30
        for s in range(0, n, step):
            e = min(n, s + step)
            call(fn, self.h, e - s, None, vp(rows_ptr + 8 * s), None, vp(cols_ptr + 8 * s), vp(vals_ptr + 8 * s))

    def assemble(self):
        call("HYPRE_IJMatrixAssemble", self.h)
        call("HYPRE_IJMatrixGetObject", self.h, C.byref(self.par))

    def destroy(self):
        if self.h:
            call("HYPRE_IJMatrixDestroy", self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class IJVector:
    def __init__(self, jlower, jupper, values=None, ncomp=1):
        """ncomp > 1: a multivector as init_system builds it for non-segregated solves
        (/root/reference/src/HypreSystem.cpp:567-571); values then has shape (ncomp, n)."""
        self.h = vp()
        self.jlower, self.jupper = jlower, jupper
        self.n = jupper - jlower + 1
        self.ncomp = ncomp
        call("HYPRE_IJVectorCreate", 0, c_big(jlower), c_big(jupper), C.byref(self.h))
        call("HYPRE_IJVectorSetObjectType", self.h, HYPRE_PARCSR)
        if ncomp != 1:
            call("HYPRE_IJVectorSetNumComponents", self.h, ncomp)
        call("HYPRE_IJVectorInitialize", self.h)
        self.par = vp()
        call("HYPRE_IJVectorGetObject", self.h, C.byref(self.par))
        if values is not None:
            if ncomp == 1:
                self.set(values)
            else:
                for c in range(ncomp):
                    self.set_component(c)
                    self.set(values[c])
        call("HYPRE_IJVectorAssemble", self.h)

    def set_component(self, c):
        """HYPRE_IJVectorSetComponent, /root/reference/src/HypreSystem.cpp:967"""
        call("HYPRE_IJVectorSetComponent", self.h, c)

    def get_all(self):
        out = np.empty((self.ncomp, self.n))
        for c in range(self.ncomp):
            self.set_component(c)
            out[c] = self.get()
        return out

    def set(self, values):
        values = dbl(values)
        idx = np.arange(self.jlower, self.jupper + 1, dtype=np.int64)
        call("HYPRE_IJVectorSetValues", self.h, self.n, idx, values)

    def fill(self, v):
        call("HYPRE_ParVectorSetConstantValues", self.par, float(v))

    def get(self):
        out = np.empty(self.n)
        if self.n:
            call("HYPRE_IJVectorGetValues", self.h, self.n, None, out)
        return out

    def destroy(self):
        if self.h:
            call("HYPRE_IJVectorDestroy", self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# YAML keys of boomeramg_settings with the app's defaults (HypreSystem.cpp:119-159)
APP_AMG_DEFAULTS = dict(print_level=1, debug_flag=1, coarsen_type=8, cycle_type=1, relax_type=8, num_sweeps=1,
                        smooth_num_sweeps=1, tolerance=0.0, max_iterations=1, relax_order=1, max_levels=20,
                        strong_threshold=0.57)


class BoomerAMG:
    """setup_boomeramg_precond, /root/reference/src/HypreSystem.cpp:119-326, same YAML keys."""

    def __init__(self, **node):
        cfg = dict(APP_AMG_DEFAULTS)
        cfg.update(node)
        self.cfg = cfg
        self.h = vp()
        call("HYPRE_BoomerAMGCreate", C.byref(self.h))
        s = self.h
        call("HYPRE_BoomerAMGSetPrintLevel", s, cfg["print_level"])
        call("HYPRE_BoomerAMGSetDebugFlag", s, cfg["debug_flag"])
        call("HYPRE_BoomerAMGSetCoarsenType", s, cfg["coarsen_type"])
        call("HYPRE_BoomerAMGSetCycleType", s, cfg["cycle_type"])
        if all(k in cfg for k in ("down_relax_type", "up_relax_type", "coarse_relax_type")):
            call("HYPRE_BoomerAMGSetCycleRelaxType", s, cfg["down_relax_type"], 1)
            call("HYPRE_BoomerAMGSetCycleRelaxType", s, cfg["up_relax_type"], 2)
            call("HYPRE_BoomerAMGSetCycleRelaxType", s, cfg["coarse_relax_type"], 3)
        else:
            call("HYPRE_BoomerAMGSetRelaxType", s, cfg["relax_type"])
        if all(k in cfg for k in ("num_down_sweeps", "num_up_sweeps", "num_coarse_sweeps")):
            call("HYPRE_BoomerAMGSetCycleNumSweeps", s, cfg["num_down_sweeps"], 1)
            call("HYPRE_BoomerAMGSetCycleNumSweeps", s, cfg["num_up_sweeps"], 2)
            call("HYPRE_BoomerAMGSetCycleNumSweeps", s, cfg["num_coarse_sweeps"], 3)
        else:
            call("HYPRE_BoomerAMGSetNumSweeps", s, cfg["num_sweeps"])
        call("HYPRE_BoomerAMGSetSmoothNumSweeps", s, cfg["smooth_num_sweeps"])
        call("HYPRE_BoomerAMGSetTol", s, float(cfg["tolerance"]))
        call("HYPRE_BoomerAMGSetMaxIter", s, cfg["max_iterations"])
        call("HYPRE_BoomerAMGSetRelaxOrder", s, cfg["relax_order"])
        call("HYPRE_BoomerAMGSetMaxLevels", s, cfg["max_levels"])
        call("HYPRE_BoomerAMGSetStrongThreshold", s, float(cfg["strong_threshold"]))
        for key, fn, conv in (("interp_type", "HYPRE_BoomerAMGSetInterpType", int),
                              ("min_coarse_size", "HYPRE_BoomerAMGSetMinCoarseSize", int),
                              ("max_coarse_size", "HYPRE_BoomerAMGSetMaxCoarseSize", int),
                              ("seq_threshold", "HYPRE_BoomerAMGSetSeqThreshold", int),
                              ("agg_num_levels", "HYPRE_BoomerAMGSetAggNumLevels", int),
                              ("agg_interp_type", "HYPRE_BoomerAMGSetAggInterpType", int),
                              ("agg_pmax_elmts", "HYPRE_BoomerAMGSetAggPMaxElmts", int),
                              ("pmax_elmts", "HYPRE_BoomerAMGSetAggPMaxElmts", int),  # sic, HypreSystem.cpp:210-213
                              ("agg_trunc_factor", "HYPRE_BoomerAMGSetAggTruncFactor", float),
                              ("trunc_factor", "HYPRE_BoomerAMGSetTruncFactor", float),
                              ("keep_transpose", "HYPRE_BoomerAMGSetKeepTranspose", int),
                              ("rap2", "HYPRE_BoomerAMGSetRAP2", int),
                              ("smooth_type", "HYPRE_BoomerAMGSetSmoothType", int),  # HypreSystem.cpp:235-320
                              ("smooth_num_sweeps", "HYPRE_BoomerAMGSetSmoothNumSweeps", int),
                              ("smooth_num_levels", "HYPRE_BoomerAMGSetSmoothNumLevels", int),
                              ("ilu_type", "HYPRE_BoomerAMGSetILUType", int),
                              ("ilu_level", "HYPRE_BoomerAMGSetILULevel", int),
                              ("ilu_max_iter", "HYPRE_BoomerAMGSetILUMaxIter", int),
                              ("ilu_tri_solve", "HYPRE_BoomerAMGSetILUTriSolve", int),
                              ("ilu_lower_jacobi_iters", "HYPRE_BoomerAMGSetILULowerJacobiIters", int),
                              ("ilu_upper_jacobi_iters", "HYPRE_BoomerAMGSetILUUpperJacobiIters", int),
                              ("true_pmax_elmts", "HYPRE_BoomerAMGSetPMaxElmts", int)):
            if key in cfg:
                call(fn, s, conv(cfg[key]))
        # non_galerkin_tol + non_galerkin_level_tols {levels, tolerances}, HypreSystem.cpp:161-176
        if "non_galerkin_tol" in cfg:
            call("HYPRE_BoomerAMGSetNonGalerkinTol", s, float(cfg["non_galerkin_tol"]))
            lt = cfg.get("non_galerkin_level_tols")
            if lt:
                for lev, tol in zip(lt["levels"], lt["tolerances"]):
                    call("HYPRE_BoomerAMGSetLevelNonGalerkinTol", s, float(tol), int(lev))

    def setup(self, A):
        call("HYPRE_BoomerAMGSetup", self.h, A.par, None, None)

    def solve(self, A, b, x):
        call("HYPRE_BoomerAMGSolve", self.h, A.par, b.par, x.par)

    @property
    def num_levels(self):
        n = c_int()
        call("HYPRE_MI_BoomerAMGGetNumLevels", self.h, C.byref(n))
        return n.value

    @property
    def operator_complexity(self):
        v = c_dbl()
        call("HYPRE_MI_BoomerAMGGetOperatorComplexity", self.h, C.byref(v))
        return v.value

    @property
    def setup_seconds(self):
        v = c_dbl()
        call("HYPRE_MI_BoomerAMGGetSetupSeconds", self.h, C.byref(v))
        return v.value

    def level_csr(self, level, which):
        """which: 0 A diag, 1 A offd, 2 P diag, 3 R diag, 4 P offd, 5 R offd -> (ia int64, ja int32, a f64, shape)."""
        nr, nc, nnz = c_int(), c_int(), c_big()
        call("HYPRE_MI_BoomerAMGGetLevelCSRSize", self.h, level, which, C.byref(nr), C.byref(nc), C.byref(nnz))
        ia = np.zeros(nr.value + 1, dtype=np.int64)
        ja = np.zeros(max(nnz.value, 1), dtype=np.int32)
        a = np.zeros(max(nnz.value, 1), dtype=np.float64)
        call("HYPRE_MI_BoomerAMGGetLevelCSR", self.h, level, which, ia, ja, a)
        return ia, ja[: nnz.value], a[: nnz.value], (nr.value, nc.value)

    def level_cf(self, level):
        nr, nc, nnz = c_int(), c_int(), c_big()
        call("HYPRE_MI_BoomerAMGGetLevelCSRSize", self.h, level, 0, C.byref(nr), C.byref(nc), C.byref(nnz))
        cf = np.zeros(nr.value, dtype=np.int32)
        call("HYPRE_MI_BoomerAMGGetLevelCF", self.h, level, cf)
        return cf

    def level_perm(self, level):
        """perm[new local row] = old local row of the level's C-first ordering."""
        nr, nc, nnz = c_int(), c_int(), c_big()
        call("HYPRE_MI_BoomerAMGGetLevelCSRSize", self.h, level, 0, C.byref(nr), C.byref(nc), C.byref(nnz))
        perm = np.zeros(nr.value, dtype=np.int32)
        call("HYPRE_MI_BoomerAMGGetLevelPerm", self.h, level, perm)
        return perm

    def input_ordering(self):
        """(applied, order): the internal locality numbering of the input, order[new] = caller's local row."""
        nr, nc, nnz = c_int(), c_int(), c_big()
        call("HYPRE_MI_BoomerAMGGetLevelCSRSize", self.h, 0, 0, C.byref(nr), C.byref(nc), C.byref(nnz))
        applied = c_int()
        order = np.zeros(nr.value, dtype=np.int32)
        call("HYPRE_MI_BoomerAMGGetInputOrdering", self.h, C.byref(applied), order)
        return bool(applied.value), order

    def level_colmap(self, level):
        nr, nc, nnz = c_int(), c_int(), c_big()
        call("HYPRE_MI_BoomerAMGGetLevelCSRSize", self.h, level, 1, C.byref(nr), C.byref(nc), C.byref(nnz))
        cm = np.zeros(max(nc.value, 1), dtype=np.int64)
        rs = c_big()
        call("HYPRE_MI_BoomerAMGGetLevelColMap", self.h, level, cm, C.byref(rs))
        return cm[: nc.value], rs.value

    def level_offd_colmap(self, level, which):
        """sorted global column ids of an offd block (which: 1 A, 4 P, 5 R)."""
        nr, nc, nnz = c_int(), c_int(), c_big()
        call("HYPRE_MI_BoomerAMGGetLevelCSRSize", self.h, level, which, C.byref(nr), C.byref(nc), C.byref(nnz))
        cm = np.zeros(max(nc.value, 1), dtype=np.int64)
        call("HYPRE_MI_BoomerAMGGetLevelOffdColMap", self.h, level, which, cm)
        return cm[: nc.value]

    def relax_level(self, level, relax_type, points, f, u):
        f = dbl(f)
        u = np.array(u, dtype=np.float64)
        call("HYPRE_MI_BoomerAMGRelaxLevel", self.h, level, relax_type, points, f, u)
        return u

    def destroy(self):
        if self.h:
            call("HYPRE_BoomerAMGDestroy", self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class ILU:
    """HYPRE_ILU: block-Jacobi ILU(0) (type 0, fill 0); usable as a preconditioner (set_precond) or a solver."""

    def __init__(self, max_iterations=1, tolerance=0.0, trisolve=1, lower_jacobi_iters=5, upper_jacobi_iters=5,
                 print_level=0, ilu_type=0, fill=0):
        self.h = vp()
        call("HYPRE_ILUCreate", C.byref(self.h))
        call("HYPRE_ILUSetType", self.h, ilu_type)
        call("HYPRE_ILUSetLevelOfFill", self.h, fill)
        call("HYPRE_ILUSetMaxIter", self.h, max_iterations)
        call("HYPRE_ILUSetTol", self.h, float(tolerance))
        call("HYPRE_ILUSetTriSolve", self.h, trisolve)
        call("HYPRE_ILUSetLowerJacobiIters", self.h, lower_jacobi_iters)
        call("HYPRE_ILUSetUpperJacobiIters", self.h, upper_jacobi_iters)
        call("HYPRE_ILUSetPrintLevel", self.h, print_level)
        self.solve_fn, self.setup_fn = "HYPRE_ILUSolve", "HYPRE_ILUSetup"

    def setup(self, A):
        call("HYPRE_ILUSetup", self.h, A.par, None, None)

    def solve(self, A, b, x):
        return call("HYPRE_ILUSolve", self.h, A.par, b.par, x.par)

    @property
    def num_iterations(self):
        n = c_int()
        call("HYPRE_ILUGetNumIterations", self.h, C.byref(n))
        return n.value

    @property
    def final_rel_res(self):
        v = c_dbl()
        call("HYPRE_ILUGetFinalRelativeResidualNorm", self.h, C.byref(v))
        return v.value

    def destroy(self):
        if self.h:
            call("HYPRE_ILUDestroy", self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:  # noqa: BLE001
            pass


class _Krylov:
    prefix = None

    def __init__(self, tolerance=1e-5, max_iterations=1000, kspace=10, print_level=4):
        """setup_gmres / setup_bicg, /root/reference/src/HypreSystem.cpp:390-404, :423-438 (same defaults)."""
        self.h = vp()
        call(f"{self.prefix}Create", 0, C.byref(self.h))
        call(f"{self.prefix}SetTol", self.h, float(tolerance))
        call(f"{self.prefix}SetMaxIter", self.h, int(max_iterations))
        if self.prefix.endswith("GMRES"):
            call(f"{self.prefix}SetKDim", self.h, int(kspace))
        call(f"{self.prefix}SetPrintLevel", self.h, int(print_level))
        self.precond = None

    def set_precond(self, amg):
        """solverPrecondPtr_(solver_, precondSolvePtr_, precondSetupPtr_, precond_), HypreSystem.cpp:687."""
        L = lib()
        solve_fn = getattr(amg, "solve_fn", "HYPRE_BoomerAMGSolve")
        setup_fn = getattr(amg, "setup_fn", "HYPRE_BoomerAMGSetup")
        call(f"{self.prefix}SetPrecond", self.h, C.cast(getattr(L, solve_fn), vp), C.cast(getattr(L, setup_fn), vp), amg.h)
        self.precond = amg

    def setup(self, A, b, x):
        call(f"{self.prefix}Setup", self.h, A.par, b.par, x.par)

    def solve(self, A, b, x):
        return call(f"{self.prefix}Solve", self.h, A.par, b.par, x.par, allow=(HYPRE_ERROR_CONV,))

    @property
    def num_iterations(self):
        n = c_int()
        call(f"{self.prefix}GetNumIterations", self.h, C.byref(n))
        return n.value

    @property
    def final_rel_res(self):
        v = c_dbl()
        call(f"{self.prefix}GetFinalRelativeResidualNorm", self.h, C.byref(v))
        return v.value

    @property
    def solve_seconds(self):
        v = c_dbl()
        call("HYPRE_MI_KrylovGetSolveSeconds", self.h, C.byref(v))
        return v.value

    def residual_history(self):
        n = c_int()
        buf = np.zeros(4096)
        call("HYPRE_MI_KrylovGetResidualHistory", self.h, buf, len(buf), C.byref(n))
        return buf[: min(n.value, len(buf))].copy()

    def destroy(self):
        if self.h:
            call(f"{self.prefix}Destroy", self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class GMRES(_Krylov):
    prefix = "HYPRE_ParCSRGMRES"


class BiCGSTAB(_Krylov):
    prefix = "HYPRE_ParCSRBiCGSTAB"


class FlexGMRES(_Krylov):
    prefix = "HYPRE_ParCSRFlexGMRES"


class PCG(_Krylov):
    prefix = "HYPRE_ParCSRPCG"


class COGMRES(_Krylov):
    prefix = "HYPRE_ParCSRCOGMRES"


def laplace3d(nx, ny, nz, stencil, ilower, iupper):
    """Synthetic COO triples + rhs for global rows [ilower, iupper] (library-side generator)."""
    nnz = c_big()
    rows, cols, vals, rhs = vp(), vp(), vp(), vp()
    call("HYPRE_MI_Laplace3D", nx, ny, nz, stencil, c_big(ilower), c_big(iupper), C.byref(nnz), C.byref(rows),
         C.byref(cols), C.byref(vals), C.byref(rhs))
    return dict(nnz=nnz.value, rows=rows.value, cols=cols.value, vals=vals.value, rhs=rhs.value,
                nloc=iupper - ilower + 1)


def laplace3d_free(g):
    for k in ("rows", "cols", "vals", "rhs"):
        if g.get(k):
            lib().HYPRE_MI_Free(vp(g[k]))
            g[k] = None


def build_laplace_system(nx, ny, nz, stencil=7, rank=0, size=1):
    """IJ matrix + rhs + zero x for this rank's block rows of the n^3 Laplacian."""
    N = nx * ny * nz
    ilower, iupper = row_partition(N, size, rank)
    A = IJMatrix(ilower, iupper)  # (created before the entries are generated, as the reference's driver does: the library
    g = laplace3d(nx, ny, nz, stencil, ilower, iupper)  # starts mapping device memory as soon as it knows the row count)
    A.set_values_ptr(g["nnz"], g["rows"], g["cols"], g["vals"])
    A.assemble()
    rhs = np.ctypeslib.as_array(C.cast(g["rhs"], C.POINTER(c_dbl)), shape=(g["nloc"],)).copy()
    laplace3d_free(g)
    b = IJVector(ilower, iupper, rhs)
    x = IJVector(ilower, iupper)
    x.fill(0.0)
    return A, b, x, rhs


def profile_enable(pid, capacity=4096):
    call("HYPRE_MI_ProfileEnable", pid, capacity)


def profile_reset():
    call("HYPRE_MI_ProfileReset")


def profile_kernel_name(pid):
    """instantiation (template flags included) of the kernel last launched under the class."""
    buf = C.create_string_buffer(128)
    call("HYPRE_MI_ProfileKernelName", pid, buf, 128)
    return buf.value.decode()


def profile_get(pid):
    n, tot, mn = C.c_longlong(), c_dbl(), c_dbl()
    call("HYPRE_MI_ProfileGet", pid, C.byref(n), C.byref(tot), C.byref(mn))
    return n.value, tot.value, mn.value


# ---------------------------------------------------------------------------
# torch.distributed transport for the callback communicator (tests only: gloo,
# host buffers).  The benchmark path uses the library's own RCCL communicator.
_comm_keep = []


def init_comm_torch(dist, device=None):
    """Bind the library's communicator to a torch.distributed process group through
    host-buffer callbacks (several ranks may share one GPU this way).  With a gloo group
    the buffers travel as CPU tensors; with an nccl group pass device="cuda" and they are
    staged through device tensors (bench.py's loud fallback when the library's own RCCL
    communicator cannot be created)."""
    import torch

    rank, size = dist.get_rank(), dist.get_world_size()
    np_dt = {0: np.float64, 1: np.int64, 2: np.int32, 3: np.uint8}
    ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MIN, 2: dist.ReduceOp.MAX}

    def _view(ptr, nbytes, dtype=np.uint8):
        buf = (C.c_char * nbytes).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype)

    def _to(t):
        return t.to(device) if device else t

    def allreduce(ctx, buf, count, dtype, op):
        dt = np_dt[dtype]
        a = _view(buf, count * np.dtype(dt).itemsize, dt)
        if device:
            t = torch.from_numpy(a.copy()).to(device)
            dist.all_reduce(t, op=ops[op])
            a[:] = t.cpu().numpy()
        else:
            dist.all_reduce(torch.from_numpy(a), op=ops[op])

    def allgather(ctx, send, recv, nbytes):
        s = _to(torch.from_numpy(_view(send, nbytes).copy()))
        out = [_to(torch.empty(nbytes, dtype=torch.uint8)) for _ in range(size)]
        dist.all_gather(out, s)
        r = _view(recv, nbytes * size)
        for i, o in enumerate(out):
            r[i * nbytes:(i + 1) * nbytes] = o.cpu().numpy()

    def exchange(ctx, nsend, speers, sptrs, sbytes, nrecv, rpeers, rptrs, rbytes):
        ops_, rbufs = [], []
        for i in range(nrecv):
            t = _to(torch.empty(rbytes[i], dtype=torch.uint8))
            rbufs.append(t)
            ops_.append(dist.P2POp(dist.irecv, t, rpeers[i]))
        keep = []
        for i in range(nsend):
            t = _to(torch.from_numpy(_view(sptrs[i], sbytes[i]).copy()))
            keep.append(t)
            ops_.append(dist.P2POp(dist.isend, t, speers[i]))
        if ops_:
            for r in dist.batch_isend_irecv(ops_):
                r.wait()
        for i in range(nrecv):
            _view(rptrs[i], rbytes[i])[:] = rbufs[i].cpu().numpy()

    cbs = (ALLREDUCE_FN(allreduce), ALLGATHER_FN(allgather), EXCHANGE_FN(exchange))
    _comm_keep.append(cbs)
    call("HYPRE_MI_CommInitCallbacks", None, cbs[0], cbs[1], cbs[2], rank, size)
    return rank, size


def build_laplace_system_host(nx, ny, nz, stencil, rank, size):
    """IJ matrix of this rank's rows, assembled on the HOST only (no device)."""
    N = nx * ny * nz
    ilower, iupper = row_partition(N, size, rank)
    g = laplace3d(nx, ny, nz, stencil, ilower, iupper)
    A = IJMatrix.__new__(IJMatrix)
    A.h = vp()
    A.ilower, A.iupper = ilower, iupper
    call("HYPRE_IJMatrixCreate", 0, c_big(ilower), c_big(iupper), c_big(ilower), c_big(iupper), C.byref(A.h))
    call("HYPRE_IJMatrixSetObjectType", A.h, HYPRE_PARCSR)
    A.par = vp()
    call("HYPRE_IJMatrixGetObject", A.h, C.byref(A.par))
    A.set_values_ptr(g["nnz"], g["rows"], g["cols"], g["vals"])
    call("HYPRE_MI_IJMatrixAssembleHostOnly", A.h)
    rhs = np.ctypeslib.as_array(C.cast(g["rhs"], C.POINTER(c_dbl)), shape=(g["nloc"],)).copy()
    laplace3d_free(g)
    return A, rhs


def halo_plan(A):
    ns, nr = c_int(), c_int()
    call("HYPRE_MI_ParCSRGetHaloPlan", A.par, C.byref(ns), None, None, None, C.byref(nr), None, None, None)
    sp = np.zeros(max(ns.value, 1), dtype=np.int32)
    ss = np.zeros(ns.value + 1, dtype=np.int32)
    rp = np.zeros(max(nr.value, 1), dtype=np.int32)
    rs = np.zeros(nr.value + 1, dtype=np.int32)
    call("HYPRE_MI_ParCSRGetHaloPlan", A.par, C.byref(ns), sp, ss, None, C.byref(nr), rp, rs)
    sm = np.zeros(max(int(ss[-1]), 1), dtype=np.int32)
    call("HYPRE_MI_ParCSRGetHaloPlan", A.par, C.byref(ns), sp, ss, sm, C.byref(nr), rp, rs)
    return dict(send_peers=sp[: ns.value], send_starts=ss, send_map=sm[: int(ss[-1])], recv_peers=rp[: nr.value],
                recv_starts=rs)


def matrix_from_scipy(M, ilower=0, iupper=None):
    """IJ matrix of rows [ilower, iupper] of a scipy sparse matrix (global column ids)."""
    M = M.tocsr()
    iupper = M.shape[0] - 1 if iupper is None else iupper
    coo = M[ilower:iupper + 1].tocoo()
    A = IJMatrix(ilower, iupper)
    A.set_values_coo(coo.row.astype(np.int64) + ilower, coo.col.astype(np.int64), coo.data)
    A.assemble()
    return A


def csr_device_op(op, A, B=None, perm=None, colpos=None):
    """Setup-phase device kernels on scipy CSR matrices: op 0 A@B, 1 A.T, 2 rows of A in perm order with
    columns mapped through colpos.  Returns (ia int64, ja int32, a f64, shape) exactly as the library stores it."""
    def parts(M):
        return (np.ascontiguousarray(M.indptr, dtype=np.int64), np.ascontiguousarray(M.indices, dtype=np.int32),
                np.ascontiguousarray(M.data, dtype=np.float64))
    aia, aja, aa = parts(A)
    if B is not None:
        bia, bja, ba = parts(B)
        bn, bm = B.shape
    else:
        bia = bja = ba = None
        bn = bm = 0
    pp = None if perm is None else np.ascontiguousarray(perm, dtype=np.int32)
    cp = None if colpos is None else np.ascontiguousarray(colpos, dtype=np.int32)
    nr, nc = c_int(), c_int()
    cia, cja, ca = vp(), vp(), vp()
    call("HYPRE_MI_CSRDeviceOp", op, A.shape[0], A.shape[1], aia, aja, aa, bn, bm, bia, bja, ba, pp, cp,
         C.byref(nr), C.byref(nc), C.byref(cia), C.byref(cja), C.byref(ca))
    ia = np.ctypeslib.as_array(C.cast(cia, C.POINTER(c_big)), shape=(nr.value + 1,)).copy()
    nnz = int(ia[-1])
    ja = np.ctypeslib.as_array(C.cast(cja, C.POINTER(c_int)), shape=(max(nnz, 1),)).copy()[:nnz]
    a = np.ctypeslib.as_array(C.cast(ca, C.POINTER(c_dbl)), shape=(max(nnz, 1),)).copy()[:nnz]
    for ptr in (cia, cja, ca):
        lib().HYPRE_MI_Free(ptr)
    return ia, ja, a, (nr.value, nc.value)
