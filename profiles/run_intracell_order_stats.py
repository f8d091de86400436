import os, sys, time, numpy as np, scipy.sparse as sp
sys.path.insert(0,'/root/repo')
import __graft_entry__ as ge
oc = ge.load_oracle()
n = int(sys.argv[1]) if len(sys.argv)>1 else 64
N = n**3
Ao, bo = oc.Csr.laplace(n,n,n,7)
S = Ao.to_scipy().tocsr()

def tile_stats(M):
    M = M.tocsr(); ia = M.indptr; nrows = M.shape[0]
    U = 0; L128=0; L64=0; r = 0
    while r < nrows:
        start = ia[r]; e = r
        while e < nrows and e - r < 256:
            e2 = min(nrows, e+8)
            if ia[e2]-start > 2047: break
            e = e2
        if e == r:
            while e < nrows and e-r < 256 and ia[e+1]-start <= 2047: e += 1
            if e == r: e = r+1
        cols=np.unique(M.indices[ia[r]:ia[e]])
        U += len(cols); L128 += len(np.unique(cols>>4)); L64 += len(np.unique(cols>>3)); r = e
    return U, L128, L64

def evaluate(name, order):
    Mq = S[order][:, order].tocsr(); Mq.sort_indices()
    amg = oc.Amg(oc.Csr.from_scipy(Mq), oc.default_params())
    out=[]
    tot=0
    for l in range(min(4, amg.num_levels-1)):
        Al = amg.level_A(l).to_scipy().tocsr()
        U, L128, L64 = tile_stats(Al)
        # bytes per entry for gathers at 128B-line granularity
        out.append(f"L{l}: U/nnz {U/Al.nnz:.3f} lines128*16/U {L128*16/U:.2f} gatherB/entry {L128*128/Al.nnz:.2f}")
        tot += L128*128
    print(f"{name:34s} " + "  ".join(out) + f"  | total gather GB-equivalent {tot/1e9:.3f}", flush=True)

def splitmix(idx):
    z = (idx.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))

def voronoi_labels(P, seeds_mask, same=None, maxr=64):
    N = P.shape[0]
    label = np.full(N, -1, dtype=np.int64)
    label[seeds_mask] = np.arange(int(seeds_mask.sum()))
    BIG = np.int64(1<<62)
    rows = np.repeat(np.arange(N), np.diff(P.indptr))
    ok = np.ones(P.nnz, dtype=bool) if same is None else (same[rows] == same[P.indices])
    for r in range(maxr):
        lab = np.where(label >= 0, label, BIG)
        g = np.where(ok, lab[P.indices], BIG)
        mins = np.minimum.reduceat(g, P.indptr[:-1])
        new = np.where((label < 0) & (mins < BIG), mins, label)
        if np.array_equal(new, label): break
        label = new
    return label

P = (S != 0).astype(np.int8).tocsr()
h = splitmix(np.arange(N))
cell = voronoi_labels(P, (h % np.uint64(512)) == 0)
cell[cell<0] = cell.max()+1
# rank cells by min row
minrow = np.full(cell.max()+1, N); np.minimum.at(minrow, cell, np.arange(N))
crank = np.empty_like(minrow); crank[np.argsort(minrow, kind='stable')] = np.arange(len(minrow))
key_cell = crank[cell]
order0 = np.lexsort((np.arange(N), key_cell))
evaluate("cells 512, natural inside (product)", order0)
# (a) geometric upper bound: inside a cell order by 2x2x4 bricks (z,y,x >> ) using coordinates
x = np.arange(N) % n; y = (np.arange(N)//n) % n; z = np.arange(N)//(n*n)
for bx,by,bz,name in ((4,2,2,"bricks 4x2x2"),(2,2,2,"bricks 2x2x2"),(8,2,1,"bricks 8x2x1"),(4,4,1,"bricks 4x4x1")):
    brick = (z//bz)*(10**8) + (y//by)*(10**4) + (x//bx)
    order = np.lexsort((np.arange(N), brick, key_cell))
    evaluate(f"cells 512, geometric {name}", order)
# (b) graph sub-cells: voronoi inside each cell with seeds 1/16, ranked by min row
for sub in (8,16,32):
    h2 = splitmix(np.arange(N) + 77777)
    seeds2 = (h2 % np.uint64(sub)) == 0
    # every cell needs a seed: its min row
    seeds2[minrow[minrow<N]] = True
    sc = voronoi_labels(P, seeds2, same=cell, maxr=8)
    sc[sc<0] = sc.max()+1+np.arange((sc<0).sum())
    mr2 = np.full(sc.max()+1, N); np.minimum.at(mr2, sc, np.arange(N))
    r2 = np.empty_like(mr2); r2[np.argsort(mr2, kind='stable')] = np.arange(len(mr2))
    order = np.lexsort((np.arange(N), r2[sc], key_cell))
    evaluate(f"cells 512, graph sub-cells 1/{sub}", order)
