import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as ge
mi = ge.load_binding(); oc = ge.load_oracle(); mi.init()
from test_gpu_amg import _combo
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 130
rng = np.random.default_rng(7000 + seed)
n = int(rng.integers(400, 2600)); per_row = float(rng.choice([5, 12, 30, 80, 150, 250]))
M = sp.random(n, n, density=min(0.5, per_row / n), random_state=rng, format="csr")
M = (M + M.T).tocsr() if rng.random() < 0.5 else M
M = (M - sp.diags(M.diagonal())).tocsr(); M.eliminate_zeros()
M = (-abs(M) + sp.diags(abs(M).sum(axis=1).A1 * float(rng.choice([1.0, 1.02, 1.3])) + 1e-3)).tolil()
if rng.random() < 0.4:
    for i in rng.choice(n, size=n // 12, replace=False):
        M.rows[i] = [int(i)]; M.data[i] = [1.0]
M = M.tocsr(); M.sort_indices()
kw = _combo(500 + seed); print(seed, n, per_row, kw)
A = mi.matrix_from_scipy(M)
amg = mi.BoomerAMG(print_level=0, **kw); amg.setup(A)
c = mi.c_int(); mi.call("HYPRE_MI_GetGSChunk", mi.C.byref(c))
Ao = oc.Csr.from_scipy(M); oamg = oc.Amg(Ao, oc.default_params(gs_chunk=c.value, **kw))
print("levels", amg.num_levels, oamg.num_levels)
for l in range(amg.num_levels):
    ia, ja, a, shape = amg.level_csr(l, 0); oia, oja, oa = oamg.level_A(l).arrays()
    same = np.array_equal(ia, oia) and np.array_equal(ja, oja)
    print("level", l, shape, "A pattern same", same, "max |dA|", np.abs(a - oa).max() if same else None, "nnz/row %.1f" % (len(ja) / max(1, shape[0])))
    if l < amg.num_levels - 1:
        print("   cf same", np.array_equal(amg.level_cf(l), oamg.level_cf(l)))
        pia, pja, pa, _ = amg.level_csr(l, 2); qia, qja, qa = oamg.level_P(l).arrays()
        ps = np.array_equal(pia, qia) and np.array_equal(pja, qja)
        print("   P same pattern", ps, "max|dP|", np.abs(pa - qa).max() if ps else None)
rng2 = np.random.default_rng(1)
f = rng2.standard_normal(n)
fi = mi.IJVector(0, n - 1, f); ui = mi.IJVector(0, n - 1, np.zeros(n))
amg.solve(A, fi, ui); got = ui.get(); ref = oamg.cycle(f)
print("cycle diff", np.abs(got - ref).max() / np.abs(ref).max())
for l in range(amg.num_levels - 1):
    nl = oamg.level_A(l).shape[0]
    fl, u0 = rng2.standard_normal(nl), rng2.standard_normal(nl)
    for rt in sorted({kw.get("relax_type", 8)}):
        for pts in (0, 1, -1):
            g = amg.relax_level(l, rt, pts, fl, u0); r = oamg.relax(l, rt, pts, fl, u0)
            print("   level", l, "relax", rt, pts, "diff", np.abs(g - r).max() / max(1.0, np.abs(r).max()))
