#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

The reference cannot run here (libHYPRE is absent: SURVEY.md 0.2, 8c), so these
are NOT outputs of the reference.  They freeze (a) analytic / direct-solve known
answers (x* = 1 for the generator systems, scipy spsolve for a random system)
and (b) the oracle's own iteration histories and hierarchy shapes at the commit
that produced them, so that a later change to oracle/ that alters its numerics is
noticed.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle_ctypes as oc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    dict(name="lap7_8", n=8, stencil=7, kdim=50, tol=1e-8, nparts=1),
    dict(name="lap7_16", n=16, stencil=7, kdim=50, tol=1e-8, nparts=1),
    dict(name="lap7_16_k5", n=16, stencil=7, kdim=5, tol=1e-10, nparts=1),
    dict(name="lap27_10", n=10, stencil=27, kdim=50, tol=1e-8, nparts=1),
    dict(name="lap7_12_p2", n=12, stencil=7, kdim=50, tol=1e-8, nparts=2),  # no redundant levels (seq_threshold 0)
    # the other solver families (method: cogmres / pcg / bicgstab with AMG; gmres with ILU(0); redundant
    # coarse levels on 3 parts)
    dict(name="lap7_12_cogmres", n=12, stencil=7, kdim=50, tol=1e-8, nparts=1, method="cogmres"),
    dict(name="lap7_12_pcg", n=12, stencil=7, kdim=50, tol=1e-8, nparts=1, method="pcg"),
    dict(name="lap7_12_bicgstab", n=12, stencil=7, kdim=50, tol=1e-8, nparts=1, method="bicgstab"),
    dict(name="lap7_10_ilu", n=10, stencil=7, kdim=50, tol=1e-8, nparts=1, method="gmres_ilu"),
    dict(name="lap7_14_p3_seq", n=14, stencil=7, kdim=50, tol=1e-8, nparts=3, redundant_rows=300),
    # the other BoomerAMG choices (round 2): the upstream sample's block (/root/reference/etc/hypre_app.yaml:33-42),
    # HMIS, CLJP, aggressive coarsening, multipass interpolation, the ILU complex smoother, two-stage Gauss-Seidel
    dict(name="lap7_12_falgout_sgs", n=12, stencil=7, kdim=50, tol=1e-8, nparts=1,
         amg=dict(coarsen_type=6, relax_type=6, num_sweeps=2, interp_type=0)),
    dict(name="lap7_12_hmis", n=12, stencil=7, kdim=50, tol=1e-8, nparts=1, amg=dict(coarsen_type=10)),
    dict(name="lap7_12_cljp", n=12, stencil=7, kdim=50, tol=1e-8, nparts=1, amg=dict(coarsen_type=0)),
    dict(name="lap7_12_agg1", n=12, stencil=7, kdim=50, tol=1e-8, nparts=1, amg=dict(agg_num_levels=1)),
    dict(name="lap7_12_multipass", n=12, stencil=7, kdim=50, tol=1e-8, nparts=1, amg=dict(interp_type=4)),
    dict(name="lap7_12_ilu_smoother", n=12, stencil=7, kdim=50, tol=1e-8, nparts=1,
         amg=dict(smooth_type=5, smooth_num_levels=2)),
    dict(name="lap7_12_two_stage_gs", n=12, stencil=7, kdim=50, tol=1e-8, nparts=1, amg=dict(relax_type=11, relax_order=0)),
]


def run_case(c):
    n = c["n"]
    A, b = oc.Csr.laplace(n, n, n, c["stencil"])
    N = n ** 3
    kw = dict(gs_chunk=8)
    if c["nparts"] > 1:
        per, rem = divmod(N, c["nparts"])
        kw["part_starts"] = [per * r + min(r, rem) for r in range(c["nparts"])] + [N]
    if "redundant_rows" in c:
        kw["redundant_rows"] = c["redundant_rows"]
    kw.update(c.get("amg", {}))
    amg = oc.Amg(A, oc.default_params(**kw))
    method = c.get("method", "gmres")
    if method == "gmres":
        x, info = oc.gmres(A, b, kdim=c["kdim"], tol=c["tol"], maxit=100, amg=amg)
    elif method == "cogmres":
        x, info = oc.cogmres(A, b, kdim=c["kdim"], cgs=0, tol=c["tol"], maxit=100, amg=amg)
    elif method == "pcg":
        x, info = oc.pcg(A, b, tol=c["tol"], maxit=100, amg=amg)
    elif method == "bicgstab":
        x, info = oc.bicgstab(A, b, tol=c["tol"], maxit=100, amg=amg)
    elif method == "gmres_ilu":
        x, info = oc.gmres(A, b, kdim=c["kdim"], tol=c["tol"], maxit=200, amg=oc.Ilu(A))
    else:
        raise ValueError(method)
    sizes = np.array([amg.level_A(l).shape[0] for l in range(amg.num_levels)])
    nnzs = np.array([amg.level_A(l).nnz for l in range(amg.num_levels)])
    return dict(rhs=b, x=x, norms=info["norms"], iters=info["iters"], rel_res=info["rel_res"], level_rows=sizes,
                level_nnz=nnzs, cf0=amg.level_cf(0).astype(np.int8))


def main():
    only = set(sys.argv[1:])  # optional: names of the fixtures to (re)generate
    for c in CASES:
        if only and c["name"] not in only:
            continue
        np.savez_compressed(os.path.join(HERE, c["name"] + ".npz"), **run_case(c))
        print("wrote", c["name"])
    if only:
        return
    # independent direct-solve fixture: random M-matrix, scipy spsolve
    rng = np.random.default_rng(20260101)
    n = 400
    M = sp.random(n, n, density=0.02, random_state=rng, format="csr")
    M = (M - sp.diags(M.diagonal())).tocsr()
    M = (-abs(M) + sp.diags(abs(M).sum(axis=1).A1 + 0.25)).tocsr()
    M.sort_indices()
    xs = rng.standard_normal(n)
    b = M @ xs
    xd = spl.spsolve(M.tocsc(), b)
    np.savez_compressed(os.path.join(HERE, "random_mmatrix_400.npz"), indptr=M.indptr, indices=M.indices, data=M.data,
                        rhs=b, x_direct=xd)
    print("wrote random_mmatrix_400")
    multicomponent_fixture()


def multicomponent_fixture():
    """BASELINE.json config 5 stand-in: 3-component seeded convection-diffusion system, BiCGSTAB + BoomerAMG,
    as ONE multivector solve (segregated_solve 0) and as three segregated solves; scipy direct solutions."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from systems import convection_diffusion_3d, three_component_rhs

    n = 12
    M = convection_diffusion_3d(n)
    B, X = three_component_rhs(M)
    A = oc.Csr.from_scipy(M)
    amg = oc.Amg(A, oc.default_params(gs_chunk=8))
    blk = oc.Csr.from_scipy(sp.kron(sp.eye(3), M).tocsr())
    xm, im = oc.bicgstab(blk, B.ravel(), tol=1e-9, maxit=60, amg=amg, ncomp=3)
    seg = [oc.bicgstab(A, B[c], tol=1e-9, maxit=60, amg=amg) for c in range(3)]
    lu = spl.splu(M.tocsc())
    np.savez_compressed(os.path.join(HERE, "convdiff3_12.npz"), n=n, indptr=M.indptr, indices=M.indices, data=M.data,
                        rhs=B, x_direct=np.stack([lu.solve(B[c]) for c in range(3)]),
                        x_multi=xm.reshape(3, -1), iters_multi=im["iters"], rel_res_multi=im["rel_res"],
                        norms_multi=im["norms"], x_seg=np.stack([s[0] for s in seg]),
                        iters_seg=np.array([s[1]["iters"] for s in seg]),
                        rel_res_seg=np.array([s[1]["rel_res"] for s in seg]))
    print("wrote convdiff3_12: multivector", im["iters"], "iterations; segregated", [s[1]["iters"] for s in seg])


if __name__ == "__main__":
    main()
