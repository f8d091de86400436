import os, sys, numpy as np, scipy.sparse as sp
sys.path.insert(0,'/root/repo')
os.environ["MI_HYPRE_LOCALITY_ORDER"]="1"
import __graft_entry__ as ge
mi = ge.load_binding(); mi.lib(); oc = ge.load_oracle()
n = int(sys.argv[1]) if len(sys.argv)>1 else 96
A, rhs = mi.build_laplace_system_host(n,n,n,7,0,1)
amg = mi.BoomerAMG(print_level=0)
mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)
def tiles(M):
    ia=M.indptr; nrows=M.shape[0]; r=0
    while r<nrows:
        start=ia[r]; e=r
        while e<nrows and e-r<256:
            e2=min(nrows,e+8)
            if ia[e2]-start>2047: break
            e=e2
        if e==r:
            while e<nrows and e-r<256 and ia[e+1]-start<=2047: e+=1
            if e==r: e=r+1
        yield r,e; r=e
print(f"{n}^3, product host setup (Voronoi numbering, C-first levels): per level, sums over tiles")
for l in range(min(4, amg.num_levels-1)):
    ia,ja,a,shape=amg.level_csr(l,0)
    M=sp.csr_matrix((a,ja,ia),shape=shape)
    nnz=M.nnz; U=0; S64=0; S128=0; runs8=0; span8=0; nt=0; runs_le4=0; U_le4=0; span_le4=0
    for r,e in tiles(M):
        cols=np.unique(M.indices[ia[r]:ia[e]]); u=len(cols); U+=u; nt+=1
        S64+=len(np.unique(cols>>3)); S128+=len(np.unique(cols>>4))
        gaps=np.diff(cols); brk=np.flatnonzero(gaps>8)
        nr=len(brk)+1; runs8+=nr
        starts=np.concatenate(([cols[0]],cols[brk+1])); ends=np.concatenate((cols[brk],[cols[-1]]))
        sp_=int((ends-starts+1).sum()); span8+=sp_
        if nr<=4 and sp_<=2048: runs_le4+=1; U_le4+=u; span_le4+=sp_
    print(f"L{l}: rows {shape[0]:8d} nnz/row {nnz/shape[0]:5.1f} tiles {nt:6d} U/nnz {U/nnz:.3f}  64B-sectors*8/U {S64*8/U:.2f}  128B-lines*16/U {S128*16/U:.2f}  runs(gap<=8)/tile {runs8/nt:.1f} span/U {span8/U:.2f}  tiles with <=4 runs: {100*runs_le4/nt:.0f}% (span/U there {span_le4/max(U_le4,1):.2f})")
