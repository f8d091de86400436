import os, sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0,'/root/repo')
os.environ["MI_HYPRE_LOCALITY_ORDER"]="1"
import __graft_entry__ as ge
mi = ge.load_binding(); mi.lib(); oc = ge.load_oracle()
n = int(sys.argv[1]) if len(sys.argv)>1 else 96
N = n**3
Ao, bo = oc.Csr.laplace(n,n,n,7)
S = Ao.to_scipy().tocsr()

def tiles_U(M):
    """emulate build_row_blocks (<=256 rows, <2048 entries, whole 8-row chunks) and count unique columns per tile"""
    M = M.tocsr(); ia = M.indptr; nrows = M.shape[0]
    U = 0; ntiles = 0; r = 0
    while r < nrows:
        start = ia[r]; e = r
        while e < nrows and e - r < 256:
            e2 = min(nrows, e+8)
            if ia[e2]-start > 2047: break
            e = e2
        if e == r:
            while e < nrows and e-r < 256 and ia[e+1]-start <= 2047: e += 1
            if e == r: e = r+1
        U += len(np.unique(M.indices[ia[r]:ia[e]])); ntiles += 1; r = e
    return U, ntiles

def evaluate(name, order):
    Mq = S[order][:, order].tocsr(); Mq.sort_indices()
    t=time.time()
    amg = oc.Amg(oc.Csr.from_scipy(Mq), oc.default_params())
    out=[]
    for l in range(min(4, amg.num_levels-1)):
        Al = amg.level_A(l).to_scipy().tocsr()
        U, nt = tiles_U(Al)
        out.append(f"L{l}: rows {Al.shape[0]:8d} U/nnz {U/Al.nnz:.3f}")
    print(f"{name:28s} " + "  ".join(out), flush=True)

def voronoi_order(S, cl, seed_hash=True):
    N = S.shape[0]
    idx = np.arange(N, dtype=np.uint64)
    # splitmix64
    z = (idx + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    is_seed = (z % np.uint64(cl)) == 0
    label = np.full(N, -1, dtype=np.int64)
    label[is_seed] = np.arange(int(is_seed.sum()))
    BIG = np.int64(1<<62)
    P = (S != 0).astype(np.int8).tocsr()
    rounds = 0
    while (label < 0).any() and rounds < 64:
        lab = np.where(label >= 0, label, BIG)
        # min over neighbours: use reduceat on gathered labels
        g = lab[P.indices]
        mins = np.minimum.reduceat(g, P.indptr[:-1])
        new = np.where((label < 0) & (mins < BIG), mins, label)
        if np.array_equal(new, label): break
        label = new; rounds += 1
    label[label < 0] = label.max()+1
    order = np.lexsort((np.arange(N), label))
    sizes = np.bincount(label)
    return order, rounds, sizes

evaluate("natural (lexicographic)", np.arange(N))
# product's BFS numbering through the host-only setup
A, rhs = mi.build_laplace_system_host(n,n,n,7,0,1)
amg = mi.BoomerAMG(print_level=0)
mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)
applied, order = amg.input_ordering()
assert applied
evaluate("BFS balls of 512 (product)", np.asarray(order))
for cl in (256, 512, 1024):
    o, r, sz = voronoi_order(S, cl)
    print(f"   voronoi cl={cl}: rounds {r}, clusters {len(sz)}, size mean {sz.mean():.0f} p5 {np.percentile(sz,5):.0f} p95 {np.percentile(sz,95):.0f} max {sz.max()}")
    evaluate(f"voronoi seeds 1/{cl}", o)
