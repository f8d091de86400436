"""Per-level roofline table of the V-cycle kernels (HIP events on the library stream, one id per level and
kernel class) for the benchmark configuration.   gpurun -- python3 profiles/roofline_table.py 512 > table.txt

Algorithmic bytes as in DESIGN.md section 5 / SURVEY 8(d): SpMV 12 nnz + 20 n; one relaxation sweep (C pass + F
pass) 12 nnz + 48 n + n; restriction 12 nnz(R) + 20 n_c; prolongation 12 nnz(P) + 28 n.  The first sweep of the
down leg starts from a zero guess and runs on the level's zero-guess sub-operator (only the entries that can meet
a non-zero): its row is priced with THAT operator's entries, 12 nnz(Az) + 48 n + n.  The residual that follows
reuses the F pass's product with the C values and reads the F rows without their C columns: priced with nnz(Ar)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
import __graft_entry__ as ge

mi = ge.load_binding()
mi.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A, b, x, _ = mi.build_laplace_system(n, n, n, 7, 0, 1)
amg = mi.BoomerAMG(print_level=0)
gm = mi.GMRES(tolerance=1e-8, max_iterations=200, kspace=50, print_level=0)
gm.set_precond(amg)
gm.setup(A, b, x)
nlev = amg.num_levels


def size(level, which):
    nr, nc, nnz = C.c_int(), C.c_int(), C.c_longlong()
    mi.call("HYPRE_MI_BoomerAMGGetLevelCSRSize", amg.h, level, which, C.byref(nr), C.byref(nc), C.byref(nnz))
    return nr.value, nnz.value


x.fill(0.0)
gm.solve(A, b, x)  # warm-up: basis vectors are allocated here
ids = [mi.PROF_SPMV_L0, mi.PROF_DOT, mi.PROF_AXPY]
for l in range(min(nlev, mi.PROF_LEVELS)):
    ids += [mi.PROF_LVL_RESID + l, mi.PROF_LVL_RELAX + l, mi.PROF_LVL_RESTRICT + l, mi.PROF_LVL_PROLONG + l,
            mi.PROF_LVL_RELAX0 + l]
for i in ids:
    mi.profile_enable(i, 8192)
mi.profile_reset()
x.fill(0.0)
gm.solve(A, b, x)
its = gm.num_iterations
print(f"laplace_3d {n}^3 7-pt, GMRES(50)+BoomerAMG, one solve: {its} iterations, {gm.solve_seconds * 1e3:.1f} ms "
      f"(event timing on: a few percent slower than the benchmark run)")
print(f"{'kernel class':34s} {'launches':>8s} {'mean ms':>9s} {'alg. GB':>9s} {'GB/s':>8s} {'% of 8.0 TB/s':>14s} {'% of 6.29 TB/s':>15s} {'ms/iteration':>13s}")


def row(label, pid, nbytes, per=1):
    cnt, ms, mn = mi.profile_get(pid)
    if not cnt:
        return
    mean = ms / cnt * per
    gbs = nbytes / (mean * 1e-3) / 1e9
    print(f"{label:34s} {cnt:8d} {mean:9.3f} {nbytes / 1e9:9.3f} {gbs:8.0f} {100 * gbs / 8000:13.1f}% {100 * gbs / 6290:14.1f}% {ms / its:13.3f}")


n0, nnz0 = size(0, 0)
row("GMRES matvec (level 0, C-first)", mi.PROF_SPMV_L0, 12.0 * nnz0 + 20.0 * n0)
for l in range(min(nlev, mi.PROF_LEVELS)):
    nl, nnzl = size(l, 0)
    nz, nnzz = size(l, 6)
    if nz:
        row(f"level {l:2d} zero-guess sweep (sub-op.)", mi.PROF_LVL_RELAX0 + l, 12.0 * nnzz + 49.0 * nl, per=2)
    row(f"level {l:2d} relaxation sweep (C+F)", mi.PROF_LVL_RELAX + l, 12.0 * nnzl + 49.0 * nl, per=2)
    if l + 1 < nlev:
        nra, nnzra = size(l, 8)  # residual after a zero-guess sweep: F rows without their C columns
        row(f"level {l:2d} residual SpMV" + (" (sub-op.)" if nra else ""), mi.PROF_LVL_RESID + l,
            12.0 * (nnzra if nra else nnzl) + 20.0 * nl)
        nr, nnzr = size(l, 3)
        row(f"level {l:2d} restriction", mi.PROF_LVL_RESTRICT + l, 12.0 * nnzr + 20.0 * nr)
        npr, nnzp = size(l, 2)
        row(f"level {l:2d} prolongation", mi.PROF_LVL_PROLONG + l, 12.0 * nnzp + 28.0 * npr)
cnt, ms, mn = mi.profile_get(mi.PROF_DOT)
print(f"{'inner products (fused MGS steps)':34s} {cnt:8d} {ms / max(cnt, 1):9.3f} {'':9s} {'':8s} {'':14s} {'':15s} {ms / its:13.3f}")
cnt, ms, mn = mi.profile_get(mi.PROF_AXPY)
print(f"{'axpy':34s} {cnt:8d} {ms / max(cnt, 1):9.3f} {'':9s} {'':8s} {'':14s} {'':15s} {ms / its:13.3f}")
