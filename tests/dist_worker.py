"""One rank of a multi-process parity run (launched by torch.distributed.run).

  --mode host   CPU only: IJ assembly, halo plan and the whole multi-rank AMG setup
                (global hierarchy, this rank's C-first ordered row slices of A, P and R
                with their halo blocks) through the C ABI with a gloo transport, checked
                against the oracle's emulation of the same row partition.
  --mode solve  GPU: the same plus the device solve (ranks share the visible GPU;
                the transport is still gloo -- RCCL refuses two ranks on one device).
Exit code 0 = every rank's checks passed.
"""
import argparse
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def local_rows_global(amg, level, which_cols_global_n, rank, diag=0, offd=1, col_start=None):
    """This rank's rows of a level operator (A: 0/1, P: 2/4, R: 3/5) as a scipy CSR over GLOBAL columns."""
    ia, ja, a, shape = amg.level_csr(level, diag)
    oia, oja, oa, oshape = amg.level_csr(level, offd)
    cm = amg.level_offd_colmap(level, offd)
    _, row_start = amg.level_colmap(level)
    if col_start is None:
        col_start = row_start
    n = shape[0]
    D = sp.csr_matrix((a, ja.astype(np.int64) + col_start, ia), shape=(n, which_cols_global_n))
    if oshape[1] > 0 and len(oa):
        assert np.all(np.diff(cm) > 0)
        assert np.all((cm < col_start) | (cm >= col_start + shape[1]))  # halo columns are off-rank
        assert len(np.unique(oja)) == len(cm), "a halo column that no entry uses"  # (it would be exchanged for nothing)
        O = sp.csr_matrix((oa, cm[oja], oia), shape=(n, which_cols_global_n))
        return (D + O).tocsr(), row_start
    return D, row_start


def same_matrix(mine, ref, tol):
    diff = abs(mine - ref)
    scale = abs(ref).max() if ref.nnz else 1.0
    return (diff.max() if diff.nnz else 0.0) <= tol * scale and (mine != 0).nnz == (ref != 0).nnz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="host")
    ap.add_argument("--grid", type=int, default=12)
    ap.add_argument("--stencil", type=int, default=7)
    ap.add_argument("--staging", default="host", help="host | cuda (device-tensor staging of the callback transport)")
    ap.add_argument("--seq", type=int, default=-1, help="redundant-level threshold (HYPRE seq_threshold); -1 = library default")
    ap.add_argument("--golden", default="", help="name of a multi-part fixture of tests/golden to replay after the oracle checks")
    ap.add_argument("--locality", type=int, default=0,
                    help="1: force the per-rank internal locality numbering (MI_HYPRE_LOCALITY_ORDER=1); the oracle then "
                         "works on the globally permuted system")
    ap.add_argument("--relax", type=int, default=0, help="relax_type of the down / up sweeps (0 = library default)")
    ap.add_argument("--ng", type=float, default=0.0, help="non_galerkin_tol (0 = Galerkin coarse operators)")
    ap.add_argument("--agg", type=int, default=0, help="agg_num_levels (aggressive coarsening + multipass interpolation)")
    ap.add_argument("--interp", type=int, default=-1, help="interp_type (-1 = library default)")
    ap.add_argument("--coarsen", type=int, default=-1, help="coarsen_type (-1 = library default)")
    ap.add_argument("--aggtrunc", type=float, default=0.0, help="agg_trunc_factor")
    ap.add_argument("--aggpmax", type=int, default=0, help="agg_pmax_elmts")
    ap.add_argument("--random", type=int, default=0,
                    help="instead of the grid operator a seeded random M-matrix with this many rows (a chain plus "
                         "random long-range couplings: every rank is a neighbour of most others, halo points' neighbours "
                         "live on third ranks); --grid is then the seed")
    ap.add_argument("--combo", type=int, default=-1,
                    help="seed of a combination of BoomerAMG choices (tests/test_gpu_amg.py::_combo) applied to both sides")
    ap.add_argument("--smooth", type=int, default=0,
                    help="levels with the ILU complex smoother (smooth_type 5): block-Jacobi ILU(0) per rank")
    ap.add_argument("--transport", default="callbacks",
                    help="callbacks (gloo through host staging) | ipc (halo exchanges by peer stores into IPC-mapped "
                         "mailboxes, HYPRE_MI_CommEnablePeerStoreExchange; reductions stay on the callbacks) | tcp (the "
                         "library's own TCP mesh, MI_HYPRE_TRANSPORT=tcp)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    os.environ["MI_HYPRE_LOCALITY_ORDER"] = "1" if args.locality else "0"
    dist.init_process_group(backend="gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    mi = ge.load_binding()
    oc = ge.load_oracle()
    if args.mode == "solve":
        mi.init()
    if args.transport == "tcp":
        # the library's own host transport (csrc/comm.cpp TcpMesh: what the C++ driver uses when ranks share a GPU):
        # bound from RANK / WORLD_SIZE / MASTER_ADDR like the RCCL one; torch.distributed only launches the ranks
        os.environ["MI_HYPRE_TRANSPORT"] = "tcp"
        os.environ["MI_HYPRE_PORT"] = str(int(os.environ["MASTER_PORT"]) + 100)
        mi.call("HYPRE_MI_CommInitFromEnv")
    else:
        mi.init_comm_torch(dist, device="cuda" if args.staging == "cuda" else None)
    if args.transport == "ipc":
        assert args.mode == "solve"
        mi.call("HYPRE_MI_CommEnablePeerStoreExchange", mi.c_big(1 << 16))  # small slots: large halos travel in parts
        nm = mi.C.create_string_buffer(128)
        mi.call("HYPRE_MI_CommName", nm, 128)
        assert size == 1 or nm.value.decode().startswith("ipc-peer-store"), nm.value
        if rank == 0:
            print("transport:", nm.value.decode())
    n, st = args.grid, args.stencil
    N = args.random if args.random else n ** 3
    starts = [mi.row_partition(N, size, r)[0] for r in range(size)] + [N]

    # ---- oracle emulation of the same partition (every rank computes it; it is small)
    if args.random:
        rng = np.random.default_rng(8800 + args.grid)
        R = sp.random(N, N, density=min(0.5, 5.0 / N), random_state=rng, format="csr")
        R = (R + R.T + sp.diags([np.ones(N - 1), np.ones(N - 1)], [-1, 1])).tocsr()
        R = (R - sp.diags(R.diagonal())).tocsr()
        R.eliminate_zeros()
        Mr = (-abs(R) + sp.diags(np.asarray(abs(R).sum(axis=1)).ravel() * float(rng.choice([1.0, 1.02, 1.2])) + 1e-3)).tocsr()
        Mr.sort_indices()
        Ao, bo = oc.Csr.from_scipy(Mr), np.asarray(Mr @ np.ones(N))
    else:
        Ao, bo = oc.Csr.laplace(n, n, n, st)
    chunk = mi.c_int()
    mi.call("HYPRE_MI_GetGSChunk", mi.C.byref(chunk))
    seq = args.seq if args.seq >= 0 else 200000
    smooth_o = dict(smooth_type=5, smooth_num_levels=args.smooth) if args.smooth else {}
    if args.relax:
        smooth_o["relax_type"] = args.relax
    if args.ng > 0.0:
        smooth_o["non_galerkin_tol"] = args.ng
    if args.agg:
        smooth_o["agg_num_levels"] = args.agg
        if args.aggtrunc:
            smooth_o["agg_trunc_factor"] = args.aggtrunc
        if args.aggpmax:
            smooth_o["agg_pmax_elmts"] = args.aggpmax
    if args.interp >= 0:
        smooth_o["interp_type"] = args.interp
    if args.coarsen >= 0:
        smooth_o["coarsen_type"] = args.coarsen
    if args.combo >= 0:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from test_gpu_amg import _combo

        smooth_o.update(_combo(args.combo))
    oamg = oc.Amg(Ao, oc.default_params(gs_chunk=chunk.value, part_starts=starts, redundant_rows=seq, **smooth_o))

    if args.random and args.mode != "host":
        lo, hi = starts[rank], starts[rank + 1]
        A = mi.IJMatrix(lo, hi - 1)
        coo = Mr[lo:hi].tocoo()
        A.set_values_coo(coo.row.astype(np.int64) + lo, coo.col.astype(np.int64), coo.data)
        A.assemble()
        rhs = bo[lo:hi]
        b = mi.IJVector(lo, hi - 1, rhs)
        x = mi.IJVector(lo, hi - 1, np.zeros(hi - lo))
    elif args.random:
        lo, hi = starts[rank], starts[rank + 1]
        A = mi.IJMatrix.__new__(mi.IJMatrix)  # (host-only: no Initialize, which would ask for the device)
        A.h = mi.vp()
        A.ilower, A.iupper = lo, hi - 1
        mi.call("HYPRE_IJMatrixCreate", 0, mi.c_big(lo), mi.c_big(hi - 1), mi.c_big(lo), mi.c_big(hi - 1), mi.C.byref(A.h))
        mi.call("HYPRE_IJMatrixSetObjectType", A.h, mi.HYPRE_PARCSR)
        A.par = mi.vp()
        mi.call("HYPRE_IJMatrixGetObject", A.h, mi.C.byref(A.par))
        coo = Mr[lo:hi].tocoo()
        A.set_values_coo(coo.row.astype(np.int64) + lo, coo.col.astype(np.int64), coo.data)
        mi.call("HYPRE_MI_IJMatrixAssembleHostOnly", A.h)
        rhs = bo[lo:hi]
    elif args.mode == "host":
        A, rhs = mi.build_laplace_system_host(n, n, n, st, rank, size)
    else:
        A, b, x, rhs = mi.build_laplace_system(n, n, n, st, rank, size)
    assert np.array_equal(rhs, bo[starts[rank]:starts[rank + 1]])

    # ---- halo plan: the ranks whose columns my rows touch (z-slab neighbours for the grid operators), what I send is
    # what my neighbour receives
    plan = mi.halo_plan(A)
    S = Ao.to_scipy().tocsr()
    mine = S[starts[rank]:starts[rank + 1]]
    expect = [r for r in range(size) if r != rank and mine[:, starts[r]:starts[r + 1]].nnz > 0]
    expect_send = [r for r in range(size) if r != rank and S[starts[r]:starts[r + 1], starts[rank]:starts[rank + 1]].nnz > 0]
    if not args.random:
        assert expect == [r for r in (rank - 1, rank + 1) if 0 <= r < size]
    assert list(plan["recv_peers"]) == expect, (rank, plan)
    assert list(plan["send_peers"]) == expect_send, (rank, plan)
    for i, p in enumerate(plan["recv_peers"]):
        # columns of peer p that my rows touch
        sub = S[starts[rank]:starts[rank + 1], starts[p]:starts[p + 1]]
        need = np.unique(sub.indices)
        assert plan["recv_starts"][i + 1] - plan["recv_starts"][i] == len(need)
    for i, p in enumerate(plan["send_peers"]):
        sub = S[starts[p]:starts[p + 1], starts[rank]:starts[rank + 1]]
        need = np.unique(sub.indices)
        got = plan["send_map"][plan["send_starts"][i]:plan["send_starts"][i + 1]]
        assert np.array_equal(np.sort(got), need), (rank, p)

    # ---- hierarchy
    amg = mi.BoomerAMG(print_level=0, **({"seq_threshold": args.seq} if args.seq >= 0 else {}), **smooth_o)
    if args.mode == "host":
        mi.call("HYPRE_MI_BoomerAMGSetupHostOnly", amg.h, A.par)
    else:
        amg.setup(A)
    order = np.arange(starts[rank + 1] - starts[rank])
    Ao_used, bo_used = Ao, bo
    # (the replicated setup -- every coarsening but PMIS -- does not renumber on N > 1 ranks)
    by_replication = size > 1 and (smooth_o.get("coarsen_type", 8) not in (8, 9, 10, 11, 1, 6, 0, 7)
                                   or os.environ.get("MI_HYPRE_REPLICATED_SETUP", "0") not in ("", "0"))
    if args.locality and by_replication:
        assert not amg.input_ordering()[0]
    if args.locality and not by_replication:
        # every rank renumbered its rows (clusters of its diag-block graph, rows with halo entries last): the oracle
        # gets the globally permuted system with the same row partition
        applied, order = amg.input_ordering()
        assert applied and np.array_equal(np.sort(order), np.arange(len(order)))
        parts = [None] * size
        dist.all_gather_object(parts, (order + starts[rank]).astype(np.int64))
        order_g = np.concatenate(parts)
        Mq = Ao.to_scipy().tocsr()[order_g][:, order_g].tocsr()
        Mq.sort_indices()
        Ao_used, bo_used = oc.Csr.from_scipy(Mq), bo[order_g]
        oamg = oc.Amg(Ao_used, oc.default_params(gs_chunk=chunk.value, part_starts=starts, redundant_rows=seq, **smooth_o))
        if size > 1:  # rows with halo entries come last on every rank
            S0 = Ao.to_scipy().tocsr()[starts[rank]:starts[rank + 1]]
            touches = np.asarray((S0[:, :starts[rank]].getnnz(axis=1) + S0[:, starts[rank + 1]:].getnnz(axis=1)) > 0)
            nh = int(touches.sum())
            assert nh == 0 or (np.all(touches[order[-nh:]]) and not np.any(touches[order[:-nh]]))
    assert amg.num_levels == oamg.num_levels, (amg.num_levels, oamg.num_levels)
    # the operator complexity every rank reports is the GLOBAL one (summed over the ranks at the end of Setup)
    ocx = sum(oamg.level_A(l).to_scipy().nnz for l in range(oamg.num_levels)) / oamg.level_A(0).to_scipy().nnz
    assert abs(amg.operator_complexity - ocx) <= 1e-12 * ocx, (rank, amg.operator_complexity, ocx)

    # ---- which setup ran, and what it held per rank (SURVEY 8e "coarse levels inherit the partition")
    def counter(name):
        v = mi.C.c_longlong()
        mi.call("HYPRE_MI_GetCounter", name.encode(), mi.C.byref(v))
        return v.value

    replicated = os.environ.get("MI_HYPRE_REPLICATED_SETUP", "0") not in ("", "0")
    # three-pass Ruge-Stueben (type 3) is built by the replicated setup; PMIS, CLJP and the per-rank types (10 HMIS,
    # 11 / 1 Ruge-Stueben on every rank's own graph, 6 Falgout) with any interpolation, aggressive levels included, by
    # the distributed one -- the per-rank types whatever the switch says
    if smooth_o.get("coarsen_type", 8) not in (8, 9, 10, 11, 1, 6, 0, 7):
        replicated = True
    if smooth_o.get("coarsen_type", 8) in (10, 11, 1, 6):
        replicated = False
    if size > 1 and not replicated:
        assert counter("setup_distributed") >= 1, "the distributed setup did not run"
        # per-rank memory: the largest extended sub-problem is this rank's rows plus two halo rings (for z-slabs of
        # an n^3 grid: at most 2 x 2 planes on each side for the 7/27-point operators and their coarse grids,
        # whose stencils reach a few planes), never the global operator; only levels below the redundancy
        # threshold are gathered
        nloc0 = starts[rank + 1] - starts[rank]
        if not args.random:  # (a random operator's two halo rings are most of the system)
            assert counter("setup_ext_rows_max") <= nloc0 + 8 * n * n, (counter("setup_ext_rows_max"), nloc0, N)
            if N >= 8 * (nloc0 + 8 * n * n) // 4:
                assert counter("setup_ext_rows_max") < N
        gathered = counter("setup_global_rows_gathered")
        assert gathered <= max(seq, 0), (gathered, seq)
        # threshold 0 on a GPU: the large levels of the distributed setup are built on the device (extended index
        # spaces, amg_setup_dist.cpp DevLevel) -- at least level 0, where every rank has rows
        if (args.mode == "solve" and os.environ.get("MI_HYPRE_DEVICE_SETUP_MIN_ROWS", "") == "0"
                and os.environ.get("MI_HYPRE_DIST_DEVICE_SETUP", "1") != "0" and smooth_o.get("interp_type", 6) in (0, 6)
                and smooth_o.get("agg_num_levels", 0) == 0  # (aggressive levels are host passes and come first)
                and smooth_o.get("coarsen_type", 8) in (8, 9)):  # (PMIS is what the device loop runs)
            assert counter("setup_device_levels") >= 1, "no level of the distributed setup was built on the device"
            if rank == 0:
                print("distributed setup: %d level(s) built on the device" % counter("setup_device_levels"))
    elif size > 1:
        assert counter("setup_distributed") == 0

    def coarse_partition(l):
        """Row partition of level l+1 as the product builds it: owner of the C point."""
        ps = oamg.level_part_starts(l)
        ocf = oamg.level_cf(l)
        return np.concatenate([[0], np.cumsum([int((ocf[ps[r]:ps[r + 1]] == 1).sum()) for r in range(size)])])

    n_redundant = 0
    for l in range(amg.num_levels):
        OA = oamg.level_A(l).to_scipy()
        ps = oamg.level_part_starts(l)
        # the oracle keeps a redundant level whole in part 0 (a small level whose C points all belong to rank 0 looks
        # the same but is distributed: the row threshold tells them apart)
        redundant = size > 1 and l >= 1 and ps[1] == ps[-1] and 0 < OA.shape[0] <= seq
        last = l == amg.num_levels - 1
        if redundant:
            # every rank holds the whole level in the single-part C-first ordering
            n_redundant += 1
            ia, ja, a, shape = amg.level_csr(l, 0)
            assert shape == OA.shape and amg.level_csr(l, 1)[3][1] == 0, (l, rank, shape, OA.shape, amg.level_csr(l, 1)[3])
            assert same_matrix(sp.csr_matrix((a, ja, ia), shape=shape), OA, 1e-12), (l, rank)
            if not last:
                assert np.array_equal(amg.level_cf(l), oamg.level_cf(l)), (l, rank)
                assert np.array_equal(amg.level_perm(l), oamg.level_perm(l)), (l, rank)
                pia, pja, pa, pshape = amg.level_csr(l, 2)
                assert same_matrix(sp.csr_matrix((pa, pja, pia), shape=pshape), oamg.level_P(l).to_scipy(), 1e-13), (l, rank)
            continue
        mine, row_start = local_rows_global(amg, l, OA.shape[1], rank)
        assert row_start == ps[rank] and mine.shape[0] == ps[rank + 1] - ps[rank], (l, rank)
        assert same_matrix(mine, OA[ps[rank]:ps[rank + 1]], 1e-12), (l, rank)
        if not last:
            cf = amg.level_cf(l)
            assert np.array_equal(cf, oamg.level_cf(l)[ps[rank]:ps[rank + 1]]), (l, rank)
            perm = amg.level_perm(l)
            operm = oamg.level_perm(l)[ps[rank]:ps[rank + 1]] - ps[rank]
            # level 0: the product's perm leads to the CALLER's rows, through the internal numbering
            assert np.array_equal(perm, order[operm] if l == 0 else operm), (l, rank)
            # interpolation reaches C points of other ranks: P and R = P^T carry halo blocks
            OP = oamg.level_P(l).to_scipy()
            psn = oamg.level_part_starts(l + 1)
            if size > 1 and psn[1] == psn[-1]:
                # the next level is redundant: the product numbers it naturally (rank slices = owners of the
                # C points), the oracle in its single-part C-first ordering
                pos = np.empty(OP.shape[1], dtype=np.int64)
                pos[oamg.level_perm(l + 1)] = np.arange(OP.shape[1])
                OP = OP[:, pos].tocsr()
                psn = coarse_partition(l)
            Pm, _ = local_rows_global(amg, l, OP.shape[1], rank, 2, 4, col_start=psn[rank])
            assert same_matrix(Pm, OP[ps[rank]:ps[rank + 1]], 1e-13), (l, rank)
            Rm, _ = local_rows_global(amg, l, OP.shape[0], rank, 3, 5, col_start=ps[rank])
            assert same_matrix(Rm, OP.T.tocsr()[psn[rank]:psn[rank + 1]], 1e-13), (l, rank)
    if size > 1 and seq > 0 and amg.num_levels > 1 and oamg.level_A(1).shape[0] <= seq:
        assert n_redundant == amg.num_levels - 1
    if seq == 0:
        assert n_redundant == 0

    if args.mode == "solve":
        def counters():
            out = {}
            for name in ("allreduce", "halo_exchange", "allgather"):
                v = mi.C.c_longlong()
                mi.call("HYPRE_MI_GetCounter", name.encode(), mi.C.byref(v))
                out[name] = v.value
            return out

        gm = mi.GMRES(tolerance=1e-8, max_iterations=60, kspace=20, print_level=0)
        gm.set_precond(amg)
        gm.setup(A, b, x)
        c0 = counters()
        rc = gm.solve(A, b, x)
        c1 = counters()
        assert rc == 0
        m = gm.num_iterations
        if size > 1 and m <= 20:
            # collectives of one solve (no restart): ||b||, ||r0||, the final true-residual norm, and per Arnoldi
            # step i the i coefficients of modified Gram-Schmidt + the norm -- each coefficient needs the vector
            # the previous one updated, so MGS cannot batch them (HYPRE's GMRES does the same i + 1 reductions)
            assert c1["allreduce"] - c0["allreduce"] == 3 + m + m * (m + 1) // 2, (c0, c1, m)
            n_dist = amg.num_levels - n_redundant
            cycles = m + 1
            # halo updates per cycle and distributed level: <= 2 sweeps x 2 passes + residual + restriction +
            # prolongation (the first pass of the down leg starts from zero and exchanges nothing); + 1 GMRES matvec
            per_cycle = (c1["halo_exchange"] - c0["halo_exchange"]) / cycles
            if args.combo < 0 and not args.smooth and not args.relax:  # (the counts below are those of a V(1,1) cycle)
                assert per_cycle <= 7 * n_dist + 2, (per_cycle, n_dist)
                assert (c1["allgather"] - c0["allgather"]) <= cycles  # coarsest gather or redundant tail: one per cycle
            # COGMRES (method: cogmres, src/HypreSystem.cpp:372-388): one block all-reduce + the norm per step
            x.fill(0.0)
            cg = mi.COGMRES(tolerance=1e-8, max_iterations=60, kspace=20, print_level=0)
            cg.set_precond(amg)
            cg.setup(A, b, x)
            c2 = counters()
            assert cg.solve(A, b, x) == 0
            c3 = counters()
            mc = cg.num_iterations
            if mc <= 20:
                assert c3["allreduce"] - c2["allreduce"] == 3 + 2 * mc, (c2, c3, mc)
            if rank == 0:
                print(f"collectives per solve: GMRES {c1['allreduce'] - c0['allreduce']} all-reduces ({m} iterations), "
                      f"COGMRES {c3['allreduce'] - c2['allreduce']} ({mc} iterations), {per_cycle:.1f} halo exchanges per cycle "
                      f"on {n_dist} distributed levels")
            x.fill(0.0)
            assert gm.solve(A, b, x) == 0  # back to the GMRES solution for the checks below
        xo, info = oc.gmres(Ao_used, bo_used, kdim=20, tol=1e-8, maxit=60, amg=oamg)
        assert gm.num_iterations == info["iters"], (gm.num_iterations, info["iters"])
        hist = gm.residual_history()
        assert np.allclose(hist, info["norms"], rtol=1e-7), (hist, info["norms"])
        assert abs(gm.final_rel_res - info["rel_res"]) <= 1e-10
        xs = x.get()[order]
        ref = xo[starts[rank]:starts[rank + 1]]
        assert np.all(np.abs(xs - ref) < np.maximum(1e-6 * np.maximum(np.abs(xs), np.abs(ref)), 1e-8))
        # distributed matvec / dot against the serial oracle
        rng = np.random.default_rng(3)
        v = rng.standard_normal(N)
        xv = mi.IJVector(starts[rank], starts[rank + 1] - 1, v[starts[rank]:starts[rank + 1]])
        yv = mi.IJVector(starts[rank], starts[rank + 1] - 1, np.zeros(starts[rank + 1] - starts[rank]))
        mi.call("HYPRE_ParCSRMatrixMatvec", 1.0, A.par, xv.par, 0.0, yv.par)
        assert np.allclose(yv.get(), Ao.matvec(v)[starts[rank]:starts[rank + 1]], rtol=1e-13, atol=1e-13)
        prod = mi.c_dbl()
        mi.call("HYPRE_ParVectorInnerProd", xv.par, xv.par, mi.C.byref(prod))
        assert abs(prod.value - float(v @ v)) <= 1e-12 * float(v @ v)
        if args.golden:
            # the committed fixture of this partition (tests/golden/make_golden.py: GMRES(50), tol 1e-8)
            g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", args.golden + ".npz"))
            assert chunk.value == 8 and len(g["rhs"]) == N
            x.fill(0.0)
            g2 = mi.GMRES(tolerance=1e-8, max_iterations=100, kspace=50, print_level=0)
            g2.set_precond(amg)
            g2.setup(A, b, x)
            assert g2.solve(A, b, x) == 0
            assert g2.num_iterations == int(g["iters"]), (g2.num_iterations, int(g["iters"]))
            assert np.allclose(g2.residual_history(), g["norms"], rtol=1e-7)
            assert abs(g2.final_rel_res - float(g["rel_res"])) <= 1e-10
            xs, ref = x.get(), g["x"][starts[rank]:starts[rank + 1]]
            assert np.all(np.abs(xs - ref) < np.maximum(1e-6 * np.maximum(np.abs(xs), np.abs(ref)), 1e-8))
            assert amg.num_levels == len(g["level_rows"])
            assert np.array_equal(np.asarray(amg.level_cf(0), dtype=np.int8), g["cf0"][starts[rank]:starts[rank + 1]])
            if rank == 0:
                print(f"golden {args.golden} ok")
        # the overlapped choreography is what ran (ranks with neighbours; a pass whose halo-free stretch is less
        # than half of its rows stays in order)
        cnt = {}
        for name in ("matvec_overlapped", "gs_overlapped", "gs_in_order"):
            v = mi.C.c_longlong()
            mi.call("HYPRE_MI_GetCounter", name.encode(), mi.C.byref(v))
            cnt[name] = v.value
        if size > 1 and os.environ.get("MI_HYPRE_OVERLAP_HALO", "1") != "0":
            assert cnt["matvec_overlapped"] > 0, cnt
            no_gs = bool(args.smooth) or args.relax in (11, 12, 7, 18) or args.combo >= 0
            if not no_gs:  # (the ILU complex smoother and the two-stage / Jacobi smoothers run no Gauss-Seidel passes)
                assert cnt["gs_overlapped"] + cnt["gs_in_order"] > 0, cnt
            if size == 2 and n >= 12 and not no_gs and not args.random:  # slabs of >= 6 planes with one neighbour: most rows are halo-free
                assert cnt["gs_overlapped"] > 0, cnt
        if rank == 0:
            print(f"overlap counters rank 0: {cnt}")
        if rank == 0:
            print(f"dist solve ok: {size} ranks, {gm.num_iterations} iterations, rel res {gm.final_rel_res:.3e}, "
                  f"{n_redundant} redundant levels")
    elif rank == 0:
        print(f"dist host setup ok: {size} ranks, {amg.num_levels} levels, {n_redundant} redundant")
    if args.mode == "solve":
        mi.call("HYPRE_MI_CommCheck")  # the peer-store transport: no wait ran into its time limit
    dist.barrier()
    mi.call("HYPRE_MI_CommFinalize")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
