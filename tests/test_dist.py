"""N>1 path: world_size-2/3 runs of tests/dist_worker.py.

CPU (gloo): the host half of the distributed path -- partition, IJ assembly, halo
plan, global hierarchy, this rank's slices of A / P / R with their halo blocks, the
redundant coarse tail -- against the oracle's emulation of the same partition.
GPU: the same plus the device solve with several ranks sharing the GPU.
seq = threshold of the redundant coarse levels (-1 library default: every level >= 1
of these small grids; 0 off: all levels distributed; in between: a switch mid-hierarchy).
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _run(nproc, mode, n, stencil, port, staging="host", seq=-1, devmin=None, golden="", replicated=False, locality=0,
         smooth=0, relax=0, combo=-1, transport="", ng=0.0, agg=0, interp=-1, aggtrunc=0.0, aggpmax=0, coarsen=-1, random=0):
    env = dict(os.environ)
    env["MI_HYPRE_REPLICATED_SETUP"] = "1" if replicated else "0"
    if devmin is not None:  # levels with at least this many rows are built (and sliced) on the device
        env["MI_HYPRE_DEVICE_SETUP_MIN_ROWS"] = str(devmin)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MI_HYPRE_HOST_THREADS"] = "2"
    env["MI_HYPRE_NATURAL_R_MIN_NNZ"] = "0"  # restriction blocks of these small grids also take the row-map path
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER, "--mode", mode, "--grid", str(n),
           "--stencil", str(stencil), "--staging", staging, "--seq", str(seq)]
    if golden:
        cmd += ["--golden", golden]
    if locality:
        cmd += ["--locality", "1"]
    if smooth:
        cmd += ["--smooth", str(smooth)]
    if relax:
        cmd += ["--relax", str(relax)]
    if combo >= 0:
        cmd += ["--combo", str(combo)]
    if transport:
        cmd += ["--transport", transport]
    if ng:
        cmd += ["--ng", str(ng)]
    if agg:
        cmd += ["--agg", str(agg), "--aggtrunc", str(aggtrunc), "--aggpmax", str(aggpmax)]
    if interp >= 0:
        cmd += ["--interp", str(interp)]
    if coarsen >= 0:
        cmd += ["--coarsen", str(coarsen)]
    if random:
        cmd += ["--random", str(random)]
    if nproc >= 5:
        # without the torch.distributed.run launcher: on a GPU box it holds the device open too, and launcher + pytest + 5
        # ranks are 7 processes on a card that admits 6
        return _spawn_direct(cmd[cmd.index(WORKER):], nproc, port, env, 600)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-4000:]
    return p.stdout


def _spawn_direct(script_and_args, nproc, port, env, timeout):
    """The ranks as direct children (RANK / WORLD_SIZE / MASTER_* in the environment, what init_process_group's env://
    reads); the first failure or the time limit ends all of them."""
    import tempfile
    import time

    procs, logs = [], []
    for r in range(nproc):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_WORLD_SIZE=str(nproc),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        f = tempfile.TemporaryFile(mode="w+")
        logs.append(f)
        procs.append(subprocess.Popen([sys.executable] + list(script_and_args), env=e, stdout=f, stderr=subprocess.STDOUT))
    t0 = time.time()
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad or time.time() - t0 > timeout:
            failed = "a rank failed" if bad else "time limit"
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
    out = ""
    for f in logs:
        f.seek(0)
        out += f.read()
        f.close()
    assert failed is None and all(p.returncode == 0 for p in procs), (failed, out[-4000:])
    return out


@pytest.mark.parametrize("nproc,n,stencil,seq", [(2, 12, 7, -1), (3, 10, 27, 0), (2, 14, 7, 300),
                                                  (4, 6, 7, 0),    # coarse levels leave ranks without rows
                                                  (8, 8, 7, 0), (8, 10, 7, -1)])  # the node size of the benchmark
def test_host_setup_world_size_n_gloo(nproc, n, stencil, seq):
    """The DISTRIBUTED setup (amg_setup_dist.cpp: distributed PMIS rounds, interpolation on the two-ring extended
    sub-problem, R*(A*P) evaluated by the owners of the coarse rows): the same partition-independent hierarchy as
    the oracle's, level by level, with per-rank sub-problems of local size (asserted by the worker)."""
    out = _run(nproc, "host", n, stencil, 29611 + nproc + (7 if seq > 0 else 0) + n, seq=seq)
    assert "dist host setup ok" in out


@pytest.mark.parametrize("nproc,n,stencil,seq", [(2, 12, 7, -1), (3, 10, 27, 0), (2, 14, 7, 300), (4, 8, 7, 0)])
def test_host_setup_with_locality_numbering_gloo(nproc, n, stencil, seq):
    """The per-rank internal locality numbering (clusters of the diag-block graph, rows with halo entries last) in
    the distributed setup: hierarchy equal to the oracle's on the globally permuted system, level-0 perm composed."""
    out = _run(nproc, "host", n, stencil, 29951 + nproc + n, seq=seq, locality=1)
    assert "dist host setup ok" in out


@pytest.mark.parametrize("nproc,n,stencil,seq,ng,locality", [(2, 14, 7, 0, 0.05, 0), (3, 12, 27, 100, 0.1, 0), (4, 10, 7, 0, 0.02, 1)])
def test_host_setup_non_galerkin_distributed_gloo(nproc, n, stencil, seq, ng, locality):
    """Non-Galerkin coarse operators (HYPRE_BoomerAMGSetNonGalerkinTol, /root/reference/src/HypreSystem.cpp:161-193) in the
    DISTRIBUTED setup: the row maxima of the halo columns come from their owners, the drop-and-lump is row-local --
    the same sparsified hierarchy as the oracle's, whatever the partition."""
    out = _run(nproc, "host", n, stencil, 30111 + nproc + n, seq=seq, ng=ng, locality=locality)
    assert "dist host setup ok" in out


@pytest.mark.parametrize("nproc,n,stencil,seq,agg,interp,aggtrunc,aggpmax,ng,locality",
                         [(2, 14, 7, 0, 1, -1, 0.0, 0, 0.0, 0), (3, 12, 27, 0, 2, -1, 0.0, 0, 0.0, 0),
                          (4, 12, 7, 100, 1, -1, 0.2, 3, 0.0, 0),   # truncated aggressive rows, redundant tail below
                          (2, 14, 7, 0, 0, 4, 0.0, 0, 0.0, 0),      # multipass as the ordinary interpolation
                          (3, 14, 7, 300, 2, -1, 0.0, 0, 0.0, 1),   # aggressive levels reach into the redundant tail
                          (8, 10, 7, 0, 1, -1, 0.0, 0, 0.0, 0),     # the node size of the benchmark
                          (2, 16, 7, 0, 3, -1, 0.0, 0, 0.05, 0),    # three aggressive levels, non-Galerkin operators
                          (8, 8, 27, 0, 0, 4, 0.0, 0, 0.0, 0),      # ranks without rows on the coarse levels still take
                          (8, 8, 27, 0, 0, 0, 0.0, 0, 0.0, 0),      # part in every exchange (classical interpolation too)
                          (4, 6, 7, 0, 2, -1, 0.0, 0, 0.0, 0)])
def test_host_setup_aggressive_levels_distributed_gloo(nproc, n, stencil, seq, agg, interp, aggtrunc, aggpmax, ng, locality):
    """Aggressive coarsening (agg_num_levels, /root/reference/src/HypreSystem.cpp:215-219) and multipass interpolation
    (agg_interp_type 4 `:220-224`, interp_type 4) in the DISTRIBUTED setup: the first-stage C points get global ids,
    the second-generation graph is built with the strong C neighbours of the halo points, the same distributed PMIS
    runs on it, and multipass proceeds pass by pass with the halo rows of the previous pass -- the oracle's
    partition-independent hierarchy level by level, per-rank sub-problems of local size (asserted by the worker)."""
    out = _run(nproc, "host", n, stencil, 30411 + nproc + n + agg, seq=seq, agg=agg, interp=interp, aggtrunc=aggtrunc,
               aggpmax=aggpmax, ng=ng, locality=locality)
    assert "dist host setup ok" in out


@pytest.mark.parametrize("nproc,n,stencil,seq,coarsen,agg,interp,ng,locality",
                         [(2, 12, 7, 0, 10, 0, -1, 0.0, 0), (2, 12, 7, 0, 11, 0, -1, 0.0, 0), (2, 12, 7, 0, 1, 0, -1, 0.0, 0),
                          (3, 10, 27, 0, 10, 0, -1, 0.0, 0), (4, 12, 7, 100, 10, 1, -1, 0.0, 0),  # HMIS on the second-generation graph
                          (8, 10, 7, 0, 10, 0, -1, 0.0, 0), (3, 12, 7, 0, 11, 0, 0, 0.0, 1),
                          (4, 8, 27, 0, 1, 2, -1, 0.0, 0), (2, 14, 7, 200, 10, 0, -1, 0.05, 1),
                          # Falgout (6: the upstream sample's type, with its classical interpolation) and CLJP (0 / 7)
                          (2, 12, 7, 0, 6, 0, -1, 0.0, 0), (4, 12, 7, 0, 6, 0, 0, 0.0, 0), (8, 10, 7, 0, 6, 0, 0, 0.0, 0),
                          (4, 12, 7, 150, 6, 1, -1, 0.0, 0), (4, 10, 27, 0, 6, 0, -1, 0.05, 1),
                          (3, 10, 27, 0, 0, 0, -1, 0.0, 0), (8, 8, 27, 0, 0, 0, -1, 0.0, 0), (3, 12, 7, 0, 7, 0, -1, 0.0, 1),
                          (2, 14, 7, 0, 0, 1, -1, 0.0, 0)])
def test_host_setup_per_rank_coarsening_types_distributed_gloo(nproc, n, stencil, seq, coarsen, agg, interp, ng, locality):
    """The coarsening types HYPRE defines PER PROCESSOR (the reference passes coarsen_type through,
    /root/reference/src/HypreSystem.cpp:126, etc/hypre_app.yaml:35): 11 / 1 = one / two Ruge-Stueben passes on every
    rank's own graph, 10 = HMIS = the first pass per rank, then PMIS from that state on the global graph (interior C
    points kept as the first independent set, boundary and F points decided again).  Built by the distributed setup
    (local Ruge-Stueben + the distributed PMIS with an initial state); the oracle emulates the partition the same way.
    6 = Falgout = both passes per rank, then CLJP on the boundary points from the interior verdicts; CLJP itself
    (0 / 7) is a global algorithm whose rounds commute: distributed with the same splitting on any partition."""
    out = _run(nproc, "host", n, stencil, 30511 + nproc + n + coarsen, seq=seq, coarsen=coarsen, agg=agg, interp=interp, ng=ng,
               locality=locality)
    assert "dist host setup ok" in out


@pytest.mark.parametrize("nproc,rows,seed,seq,coarsen,agg,interp,ng",
                         [(3, 1200, 3, 0, 10, 0, -1, 0.0), (4, 1500, 4, 0, 6, 0, -1, 0.0), (5, 1800, 5, 0, 0, 0, -1, 0.0),
                          (8, 2400, 6, 0, 10, 1, -1, 0.0), (4, 1600, 7, 100, 6, 0, 0, 0.0), (6, 2000, 8, 0, -1, 0, 4, 0.0),
                          (8, 2500, 9, 0, -1, 2, -1, 0.0), (4, 1400, 11, 0, 1, 0, -1, 0.05)])
def test_host_setup_random_operators_distributed_gloo(nproc, rows, seed, seq, coarsen, agg, interp, ng):
    """The distributed setup on UNSTRUCTURED operators: a seeded random M-matrix (a chain plus random long-range
    couplings) in contiguous row blocks -- every rank is a neighbour of most others, the neighbours of halo points and
    the C points an interpolation row reaches live on third ranks, the second-generation graph's remote columns are
    owned by ranks that are not neighbours in A.  PMIS / HMIS / Falgout / CLJP / per-rank Ruge-Stueben, aggressive
    levels, multipass, classical interpolation, non-Galerkin operators: the oracle's hierarchy level by level."""
    out = _run(nproc, "host", seed, 7, 30711 + nproc + seed, seq=seq, coarsen=coarsen, agg=agg, interp=interp, ng=ng, random=rows)
    assert "dist host setup ok" in out


@pytest.mark.parametrize("nproc,n,stencil,seq,coarsen,agg", [(2, 12, 7, 0, -1, 0), (4, 12, 27, 100, 6, 0), (8, 10, 7, 0, -1, 1)])
def test_host_setup_tcp_transport_gloo_free(nproc, n, stencil, seq, coarsen, agg):
    """The library's own host transport (csrc/comm.cpp TcpMesh, MI_HYPRE_TRANSPORT=tcp, bound by
    HYPRE_MI_CommInitFromEnv like the RCCL one): the distributed setup on 2 / 4 / 8 ranks over it -- all-reduces,
    all-gathers and the variable-size neighbour exchanges of the setup -- against the oracle as over gloo."""
    out = _run(nproc, "host", n, stencil, 30611 + nproc + n, seq=seq, coarsen=coarsen, agg=agg, transport="tcp")
    assert "dist host setup ok" in out


@pytest.mark.parametrize("nproc,n,stencil,seq", [(2, 12, 7, -1), (3, 10, 27, 0), (4, 6, 7, 0)])
def test_host_setup_replicated_path_gloo(nproc, n, stencil, seq):
    """The replicated setup (every rank builds the global hierarchy and keeps its slices) stays the path of the
    three-pass Ruge-Stueben coarsening (type 3) on N > 1; MI_HYPRE_REPLICATED_SETUP=1 forces it for PMIS and CLJP."""
    out = _run(nproc, "host", n, stencil, 29911 + nproc + n, seq=seq, replicated=True)
    assert "dist host setup ok" in out


@pytest.mark.parametrize("nproc,n,seq,combo", [(2, 14, 300, 1), (3, 12, 100, 15), (2, 14, 0, 17), (2, 12, 150, 11),
                                                (2, 13, 200, 18)])  # 18: non-Galerkin coarse operators (replicated setup)
def test_host_setup_parameter_combinations_gloo(nproc, n, seq, combo):
    """Seeded combinations of the BoomerAMG choices on N ranks (host half): Ruge-Stueben / CLJP coarsening (replicated
    setup), PMIS with aggressive levels that reach into the redundant tail and multipass interpolation (distributed
    setup); a coarsest level below the redundancy threshold is held whole by every rank, as in the oracle's emulation."""
    out = _run(nproc, "host", n, 7, 30111 + nproc + n + combo, seq=seq, combo=combo)
    assert "dist host setup ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,stencil,seq", [(2, 16, 7, -1), (4, 12, 7, 0), (3, 10, 27, 0), (3, 14, 7, 300),
                                                  (4, 16, 7, 1000), (4, 6, 7, 0),
                                                  (5, 12, 7, 0), (5, 10, 27, 200)])
# 5 ranks is what this pool allows: a GPU box admits 6 processes on its card at once, and the pytest process (which has
# initialised the device in earlier tests) is one of them (the 5-rank cases start their ranks directly: _spawn_direct).  The benchmark's rank count, 8, therefore runs the DEVICE path
# nowhere before the driver's own 8-GPU node; 8 ranks in host mode: test_host_setup_* above.  With 5 ranks of 12^3 / 10^3
# rows the coarse levels leave several ranks without rows (the case the 8-rank host run caught a bug in, commit b7ff8d1).
def test_device_solve_shared_gpu(nproc, n, stencil, seq):
    out = _run(nproc, "solve", n, stencil, 29651 + nproc + (11 if seq > 0 else 0) + n, seq=seq)
    assert "dist solve ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,stencil,seq", [(2, 16, 7, -1), (3, 14, 7, 300), (4, 12, 27, 0)])
def test_device_setup_and_slicing_shared_gpu(nproc, n, stencil, seq):
    """Every level of the global hierarchy built on the device and sliced there (the production path for
    large problems; the small grids of the other tests stay below the device threshold)."""
    out = _run(nproc, "solve", n, stencil, 29731 + nproc + n, seq=seq, devmin=0, replicated=True)
    assert "dist solve ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,stencil,seq", [(2, 16, 7, -1), (3, 14, 7, 300), (4, 12, 27, 0)])
def test_distributed_setup_device_spgemm_shared_gpu(nproc, n, stencil, seq):
    """The distributed setup with its sub-problem products A*P and R*(A*P) on the device (what large levels do)."""
    out = _run(nproc, "solve", n, stencil, 29831 + nproc + n, seq=seq, devmin=0)
    assert "dist solve ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,stencil,seq,locality", [(2, 20, 7, 0, 0), (3, 16, 27, 0, 0), (4, 18, 7, 500, 1),
                                                           (2, 24, 7, 0, 1), (3, 14, 27, 100, 1), (5, 20, 7, 0, 1),
                                                           (5, 15, 27, 0, 0)])
def test_distributed_setup_device_levels_shared_gpu(nproc, n, stencil, seq, locality):
    """The device-resident levels of the distributed setup (extended index spaces; PMIS rounds, interpolation, both
    Galerkin products, the C-first split into diag / halo blocks all on the device, halo-sized pieces through the
    host): the same hierarchy as the oracle's, level by level, with and without the internal locality numbering,
    down to the level where a rank runs out of rows (the host loop continues from there)."""
    out = _run(nproc, "solve", n, stencil, 30011 + nproc + n, seq=seq, devmin=0, locality=locality)
    assert "dist solve ok" in out
    assert "built on the device" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,stencil,seq,ng,devmin", [(2, 20, 7, 0, 0.05, 0), (3, 14, 27, 0, 0.1, 0), (4, 16, 7, 300, 0.02, None)])
def test_distributed_setup_non_galerkin_shared_gpu(nproc, n, stencil, seq, ng, devmin):
    """Non-Galerkin coarse operators on N > 1 ranks, device-resident levels (threshold 0) and the host loop: hierarchy
    equal to the oracle's level by level, GMRES on it converges like the oracle's."""
    out = _run(nproc, "solve", n, stencil, 30211 + nproc + n, seq=seq, ng=ng, devmin=devmin)
    assert "dist solve ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,stencil,seq,agg,interp,devmin", [(2, 20, 7, 0, 1, -1, 0), (3, 16, 27, 0, 2, -1, None),
                                                                  (4, 16, 7, 500, 1, -1, None), (2, 16, 7, 0, 0, 4, 0)])
def test_device_solve_aggressive_levels_distributed_shared_gpu(nproc, n, stencil, seq, agg, interp, devmin):
    """Aggressive levels / multipass interpolation built by the distributed setup (host loop also with the device
    threshold at 0), then the device solve on N ranks: hierarchy, iterations, residual history, solution vs the oracle."""
    out = _run(nproc, "solve", n, stencil, 30451 + nproc + n + agg, seq=seq, agg=agg, interp=interp, devmin=devmin)
    assert "dist solve ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,stencil,seq,coarsen,agg", [(2, 16, 7, 0, 10, 0), (3, 14, 27, 0, 11, 0), (4, 16, 7, 300, 10, 1),
                                                            (2, 16, 7, 0, 6, 0), (4, 14, 7, 0, 0, 0)])
def test_device_solve_per_rank_coarsening_types_shared_gpu(nproc, n, stencil, seq, coarsen, agg):
    """HMIS / per-rank Ruge-Stueben hierarchies from the distributed setup, then the device solve on N ranks:
    hierarchy, iterations, residual history, solution against the oracle's emulation of the partition."""
    out = _run(nproc, "solve", n, stencil, 30551 + nproc + n + coarsen, seq=seq, coarsen=coarsen, agg=agg)
    assert "dist solve ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,rows,seed,seq,devmin,coarsen,agg", [(3, 3000, 21, 0, 0, -1, 0), (4, 4000, 22, 300, None, -1, 0),
                                                                   (4, 3500, 24, 0, 0, -1, 0), (2, 2500, 23, 0, 0, 6, 0),
                                                                   (3, 3000, 25, 0, None, -1, 1)])
def test_device_solve_random_operators_shared_gpu(nproc, rows, seed, seq, devmin, coarsen, agg):
    """UNSTRUCTURED operators on N ranks (seeded random M-matrices in contiguous row blocks: many neighbours per rank,
    third-rank owners) through the distributed setup -- device-resident levels with the threshold at 0, the host loop
    otherwise -- and the device solve: hierarchy, halo plans, iterations, residual history, solution vs the oracle."""
    out = _run(nproc, "solve", seed, 7, 30751 + nproc + seed, seq=seq, devmin=devmin, coarsen=coarsen, agg=agg, random=rows)
    assert "dist solve ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,stencil,seq", [(2, 16, 7, -1), (3, 12, 27, 0), (4, 16, 7, 1000)])
def test_device_solve_with_locality_numbering_shared_gpu(nproc, n, stencil, seq):
    out = _run(nproc, "solve", n, stencil, 29871 + nproc + n, seq=seq, locality=1)
    assert "dist solve ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,seq,smooth", [(2, 16, -1, 1), (3, 14, 300, 3), (4, 12, 0, 2)])
def test_device_solve_with_ilu_complex_smoother_shared_gpu(nproc, n, seq, smooth):
    """smooth_type 5 on the first `smooth` levels (src/HypreSystem.cpp:235-320): every rank smooths with the ILU(0) of
    its own diag block, also on redundant levels (one block there), as the oracle's emulation does."""
    out = _run(nproc, "solve", n, 7, 29991 + nproc + n, seq=seq, smooth=smooth)
    assert "dist solve ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,seq,relax", [(2, 16, -1, 11), (3, 14, 300, 12), (4, 12, 0, 18)])
def test_device_solve_other_smoothers_shared_gpu(nproc, n, seq, relax):
    """Two-stage Gauss-Seidel (11 / 12: the lower triangle of each rank's diag block) and l1-Jacobi (18) on N ranks."""
    out = _run(nproc, "solve", n, 7, 30031 + nproc + n, seq=seq, relax=relax)
    assert "dist solve ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,seq,combo", [(2, 14, 300, 1), (3, 12, 100, 15), (2, 14, 0, 17), (4, 12, 200, 19),
                                                (3, 13, -1, 0), (2, 12, 150, 11),
                                                (2, 13, 200, 18), (3, 12, 0, 14)])  # 18, 14: non-Galerkin coarse operators
def test_device_solve_parameter_combinations_shared_gpu(nproc, n, seq, combo):
    """Seeded combinations of the BoomerAMG choices (test_gpu_amg.py::_combo: Ruge-Stueben / CLJP coarsening,
    aggressive levels, multipass, complex smoother, W cycles ...) on N ranks: the replicated setup for every
    coarsening but PMIS, aggressive levels and smoothed levels that reach into the redundant tail."""
    out = _run(nproc, "solve", n, 7, 30071 + nproc + n + combo, seq=seq, combo=combo)
    assert "dist solve ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc", [2, 3, 4, 5])
def test_peer_store_exchange_raw_shared_gpu(nproc):
    """The hipIpc peer-store transport by itself (tests/ipc_worker.py): empty / tiny / unaligned / multi-slot messages
    between every pair of ranks, many rounds, slot reuse; ranks share the test GPU (IPC handles work between
    processes on one device, where RCCL refuses to run)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if nproc >= 5:  # (see _run)
        out = _spawn_direct([os.path.join(ROOT, "tests", "ipc_worker.py")], nproc, 30211 + nproc, env, 600)
        assert f"ipc exchange ok: {nproc} ranks" in out
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(30211 + nproc), os.path.join(ROOT, "tests", "ipc_worker.py")]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-4000:]
    assert f"ipc exchange ok: {nproc} ranks" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("scenario,nproc", [("refuse", 2), ("gate", 2), ("gate", 3)])
def test_peer_store_transport_failure_modes_shared_gpu(scenario, nproc):
    """(refuse) mailboxes in ordinary device memory are refused when the ranks report different devices, and
    the communicator keeps its transport; (gate) a bounded wait that expired makes the next Solve fail on every rank
    through the plain HYPRE entry point, with x overwritten by NaN -- never a silently wrong result with return code 0."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(30311 + nproc + (7 if scenario == "gate" else 0)),
           os.path.join(ROOT, "tests", "ipc_worker.py"), scenario]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert p.returncode == 0, p.stdout[-4000:]
    assert f"ipc {'refusal' if scenario == 'refuse' else 'gate'} ok: {nproc} ranks" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,stencil,seq", [(2, 16, 7, -1), (3, 14, 7, 0), (4, 12, 27, 0), (4, 16, 7, 1000),
                                                  (5, 14, 7, 0), (5, 10, 27, -1)])
def test_device_solve_peer_store_transport_shared_gpu(nproc, n, stencil, seq):
    """The whole distributed solve with its halo updates on the peer-store transport (64 KiB slots: the fine-level
    halos travel in several parts): hierarchy, iteration count, residual history and solution against the oracle as in
    the other transports, overlapped choreography included."""
    out = _run(nproc, "solve", n, stencil, 30251 + nproc + n, seq=seq, transport="ipc")
    assert "dist solve ok" in out


@pytest.mark.gpu
def test_device_solve_cuda_staged_transport():
    """bench.py's fallback transport (torch.distributed collectives on device tensors behind the callback
    interface), exercised here over gloo because two nccl ranks cannot share the one GPU of the test box."""
    out = _run(2, "solve", 12, 7, 29699, staging="cuda")
    assert "dist solve ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("name,nproc,n,seq", [("lap7_12_p2", 2, 12, 0), ("lap7_14_p3_seq", 3, 14, 300)])
def test_multi_part_golden_fixtures(name, nproc, n, seq):
    """The committed multi-part fixtures (tests/golden), replayed with one rank per part."""
    out = _run(nproc, "solve", n, 7, 29771 + nproc + n, seq=seq, golden=name)
    assert f"golden {name} ok" in out


@pytest.mark.gpu
def test_rccl_overlap_equals_in_order_on_two_gpus():
    """with at least two GPUs, the RCCL transport's side-stream choreography (MI_HYPRE_OVERLAP_HALO=1) must
    give bitwise the same solve as the in-order one (=0).  The one-GPU test boxes skip this; bench.py exercises the
    same path whenever the driver runs it on a multi-GPU node."""
    import json

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    outs = []
    for overlap in ("1", "0"):
        env = dict(os.environ, MI_HYPRE_OVERLAP_HALO=overlap, HSA_ENABLE_IPC_MODE_LEGACY="0")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "96", "--steps", "1",
                            "--warmup", "0", "--no-cpu"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-3000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
        outs.append(json.loads(line))
    assert outs[0]["iterations_per_solve"] == outs[1]["iterations_per_solve"]
    assert outs[0]["final_rel_residual"] == outs[1]["final_rel_residual"]  # bitwise
    assert outs[0]["config"]["transport"].startswith("rccl")
