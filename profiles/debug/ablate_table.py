"""Timing ablations of the tile kernels (kernels.hip, -DMI_ABLATE=<bits>): which stage of a tile bounds it?
   MI_HYPRE_LIB=<ablated build> python3 profiles/debug/ablate_table.py 512
Prints the mean launch time per kernel class and level (HIP events on the library stream).  With any ablation bit set
the RESULTS ARE WRONG by construction (the solve runs its 12 steps on garbage); only the times mean anything."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge

mi = ge.load_binding()
mi.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A, b, x, _ = mi.build_laplace_system(n, n, n, 7, 0, 1)
amg = mi.BoomerAMG(print_level=0)
gm = mi.GMRES(tolerance=1e-30, max_iterations=12, kspace=50, print_level=0)
gm.set_precond(amg)
gm.setup(A, b, x)
nlev = amg.num_levels


def solve():
    x.fill(0.0)
    try:
        gm.solve(A, b, x)
    except mi.HypreError as e:  # not converged / NaN: expected here
        return str(e)[:60]
    return "ok"


solve()
ids = [mi.PROF_SPMV_L0]
L = min(nlev, mi.PROF_LEVELS, 5)
for l in range(L):
    ids += [mi.PROF_LVL_RESID + l, mi.PROF_LVL_RELAX + l, mi.PROF_LVL_RESTRICT + l, mi.PROF_LVL_PROLONG + l,
            mi.PROF_LVL_RELAX0 + l]
for i in ids:
    mi.profile_enable(i, 8192)
mi.profile_reset()
st = solve()


def mean(pid, per=1):
    cnt, ms, mn = mi.profile_get(pid)
    return ms / cnt * per if cnt else float("nan")


print(f"lib {os.path.basename(os.environ.get('MI_HYPRE_LIB', 'libmi_hypre.so'))}: solve -> {st}; {gm.solve_seconds * 1e3:.1f} ms")
print(f"  matvec L0 {mean(mi.PROF_SPMV_L0):.3f}")
for l in range(L):
    print(f"  level {l}: zero-guess sweep {mean(mi.PROF_LVL_RELAX0 + l, 2):.3f}  sweep C+F {mean(mi.PROF_LVL_RELAX + l, 2):.3f}  "
          f"residual {mean(mi.PROF_LVL_RESID + l):.3f}  restrict {mean(mi.PROF_LVL_RESTRICT + l):.3f}  prolong {mean(mi.PROF_LVL_PROLONG + l):.3f}")
