#!/usr/bin/env python3
"""Solve-phase benchmark of the GMRES + BoomerAMG path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one HYPRE_ParCSRGMRESSolve call (x0 = 0 -> converged) on the
synthetic laplace_3d problem, exactly the "Solve" timer of the reference driver
(/root/reference/src/HypreSystem.cpp:715-727: barrier-fenced, setup excluded).
The global grid is fixed (strong scaling): rank r owns the contiguous block of
rows of init_row_decomposition (HypreSystem.cpp:525-544), i.e. z-slabs.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md:36


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--grid", dest="n", type=int, default=int(os.environ.get("MI_BENCH_N", "512")), help="grid points per side")
    ap.add_argument("--stencil", type=int, default=7)
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--kdim", type=int, default=50)
    ap.add_argument("--max-iter", type=int, default=200)
    ap.add_argument("--cpu-n", type=int, default=int(os.environ.get("MI_BENCH_CPU_N", "-1")),
                    help="grid side of the bounded CPU-baseline sample (0 = skip; default: 256 = BASELINE.json "
                         "config 2 when the host has the memory, else 128)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-general", action="store_true",
                    help="skip the general-operator leg (second setup with the value dictionary off + 3 solves)")
    ap.add_argument("--workload", choices=("laplace", "convdiff3"), default="laplace",
                    help="laplace = the headline (BASELINE.json configs 2/3: strong scaling of one n^3 grid); convdiff3 = "
                         "side-line for config 5: 3-component non-symmetric convection-diffusion system, BiCGSTAB + "
                         "BoomerAMG, WEAK scaling (n^3 rows per rank, z-slabs stacked)")
    ap.add_argument("--segregated", type=int, default=0,
                    help="convdiff3: 1 = one solve per component (segregated_solve 1), 0 = one multivector solve")
    ap.add_argument("--sideline-non-galerkin", type=float, default=0.05,
                    help="non_galerkin_tol of the side-line leg reported beside the headline at N = 1 (0 = skip)")
    ap.add_argument("--halo-transport", choices=("rccl", "ipc"), default=os.environ.get("MI_BENCH_HALO_TRANSPORT", "rccl"),
                    help="N > 1: how the halo updates travel -- rccl = ncclSend/ncclRecv groups (default), ipc = peer stores into "
                         "hipIpc-mapped mailboxes (HYPRE_MI_CommEnablePeerStoreExchange); reductions are RCCL either way")
    ap.add_argument("--ipc-sideline", action="store_true", default=os.environ.get("MI_BENCH_IPC_SIDELINE") == "1",
                    help="N > 1: AFTER the headline line has been printed, repeat the solves on the peer-store transport and "
                         "print the result as a second line tagged [sideline_peer_store].  Opt-in (ADVICE r3): that transport "
                         "has never run between separate devices, and a fault inside a peer store is not a Python exception")
    ap.add_argument("--no-ipc-sideline", action="store_true", help="(accepted for older command lines; the side-line is opt-in now)")
    ap.add_argument("--cpu-child", type=int, default=0, help=argparse.SUPPRESS)  # internal: the CPU baseline at this grid size, in a child process
    ap.add_argument("--chunk", type=int, default=8, help=argparse.SUPPRESS)
    ap.add_argument("--amg", action="append", default=[], metavar="KEY=VALUE",
                    help="boomeramg_settings override for a side-line (e.g. --amg agg_num_levels=1); the headline "
                         "configuration is the one without overrides")
    return ap.parse_args()


def peer_store_sideline(mi, dist, torch, rank, world, one_solve, barrier, steps, head_iters, head_rel_res, head_s, ndof):
    """The headline's solves again with neighbour exchanges and scalar all-reduces by peer stores (see the call site)."""
    what = ("SIDE-LINE, not the headline: the same solves with halo updates and scalar all-reduces on the peer-store transport "
            "(hipIpc-mapped fine-grained mailboxes, one launch per exchange; HYPRE_MI_CommEnablePeerStoreExchange)")

    def agreed(ok):  # every rank continues, or none
        f = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return int(f.item()) == 1

    def step(fn):
        err = ""
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"[:300]
        return err

    os.environ["MI_HYPRE_IPC_TIMEOUT_MS"] = os.environ.get("MI_BENCH_IPC_PROBE_TIMEOUT_MS", "3000")
    err = step(lambda: mi.call("HYPRE_MI_CommEnablePeerStoreExchange", mi.c_big(0)))
    if not agreed(not err):
        return {"what": what, "ran": False, "why": "the mailboxes could not be set up on every rank" + (": " + err if err else "")}
    nm = C.create_string_buffer(160)
    mi.call("HYPRE_MI_CommName", nm, 160)

    def probe():
        # patterned messages to both neighbours of a ring and back, several sizes and rounds (slot reuse), then scalar
        # all-reduces whose rank-ordered sum every rank can predict
        right, left = (rank + 1) % world, (rank - 1) % world
        peers = sorted({right, left} - {rank})
        for rnd in range(12):
            nbytes = (8, 4096, 65536 + 24, 1 << 20)[rnd % 4]
            sb = {q: torch.full((nbytes,), (17 * rank + 3 * q + rnd) % 251, dtype=torch.uint8, device="cuda") for q in peers}
            rb = {q: torch.full((nbytes,), 255, dtype=torch.uint8, device="cuda") for q in peers}
            torch.cuda.synchronize()
            k = len(peers)
            ip = (C.c_int * k)(*peers)
            sp = (C.c_void_p * k)(*[sb[q].data_ptr() for q in peers])
            rp = (C.c_void_p * k)(*[rb[q].data_ptr() for q in peers])
            nb = (C.c_size_t * k)(*([nbytes] * k))
            mi.call("HYPRE_MI_CommExchangeDevice", k, ip, sp, nb, k, ip, rp, nb)
            for q in peers:
                want = (17 * q + 3 * rank + rnd) % 251
                got = rb[q]
                if not bool((got == want).all().item()):
                    raise RuntimeError(f"probe: wrong bytes from rank {q} in round {rnd} ({nbytes} bytes)")
        for rnd in range(16):
            cnt = 1 + rnd % 8
            t = torch.tensor([float(rank + 1) * (rnd + 1) + 0.5 * j for j in range(cnt)], dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            mi.call("HYPRE_MI_CommAllreduceDevice", C.c_void_p(t.data_ptr()), cnt)
            want = [sum(float(r + 1) * (rnd + 1) + 0.5 * j for r in range(world)) for j in range(cnt)]
            if t.cpu().tolist() != want:
                raise RuntimeError(f"probe: all-reduce {rnd} gave {t.cpu().tolist()} for {want}")
        mi.call("HYPRE_MI_CommCheck")

    err = step(probe)
    if not agreed(not err):
        return {"what": what, "ran": False, "transport": nm.value.decode(),
                "why": "the probe (patterned ring exchanges + scalar all-reduces) failed on some rank" + (": " + err if err else "")}
    state = {"iters": 0, "total": 0}

    def timed():
        # (the library itself ends every Solve with a collective look at the transport's error flag and raises on
        # every rank -- capi.cpp transport_gate; the explicit check is the belt to those braces)
        one_solve()  # warm-up on the new transport
        mi.call("HYPRE_MI_CommCheck")
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            state["iters"] = one_solve()
            state["total"] += state["iters"]
            mi.call("HYPRE_MI_CommCheck")
        barrier()
        state["elapsed"] = time.perf_counter() - t0

    err = step(timed)
    if not agreed(not err):
        return {"what": what, "ran": False, "transport": nm.value.decode(),
                "why": "a solve on the peer-store transport failed on some rank" + (": " + err if err else "")}
    t = torch.tensor([state["elapsed"]], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    return {"what": what, "ran": True, "transport": nm.value.decode(), "ms_per_step": el / steps * 1e3,
            "iterations_per_solve": state["iters"], "value_gdofs": ndof * state["total"] / el / 1e9, "steps": steps,
            "same_iterations_as_headline": state["iters"] == head_iters,
            "time_to_solution_vs_headline": (el / steps) / head_s}


def build_convdiff3(mi, n, rank, world):
    """BASELINE.json config 5 stand-in (no nalu-wind dump exists offline): rows of this rank's z-slab of the
    n x n x (n*world) grid (weak scaling; lexicographic numbering = contiguous block rows, HypreSystem.cpp:525-544) of
    a 7-point convection-diffusion operator -- diffusion 6 / -1, first-order upwind convection with a smooth,
    partition-independent velocity field of cell Peclet number <= 0.6 -- and three right-hand sides b_c = A x_c for
    closed-form x_c.  Diagonally dominant, non-symmetric M-matrix."""
    nzg = n * world
    z0 = n * rank
    zz, yy, xx = np.meshgrid(np.arange(z0, z0 + n), np.arange(n), np.arange(n), indexing="ij")
    gid = (xx + n * (yy + n * zz)).astype(np.int64)

    def vel(ax, x_, y_, z_):
        fx, fy, fz = 2 * np.pi * x_ / n, 2 * np.pi * y_ / n, 2 * np.pi * z_ / nzg
        return 0.6 * (np.sin(fy) * np.cos(fz) if ax == 0 else np.sin(fz) * np.cos(fx) if ax == 1 else np.sin(fx) * np.cos(fy))

    def exact(c, x_, y_, z_):
        if c == 0:
            return np.ones_like(x_, dtype=np.float64)
        if c == 1:
            return np.sin(3.0 * x_ / n) * np.cos(2.0 * y_ / n) + z_ / nzg
        return np.cos(5.0 * (x_ + y_) / n) * np.sin(4.0 * z_ / nzg + 0.3)

    rows, cols, vals = [], [], []
    diag = np.full(gid.shape, 6.0)
    rhs = [np.zeros(gid.shape) for _ in range(3)]
    dims = (n, n, nzg)
    for ax, (dx, dy, dz) in enumerate(((1, 0, 0), (0, 1, 0), (0, 0, 1))):
        v = vel(ax, xx, yy, zz)
        diag += np.abs(v)
        for sgn in (-1, 1):
            X, Y, Z = xx + sgn * dx, yy + sgn * dy, zz + sgn * dz
            inside = (X >= 0) & (X < dims[0]) & (Y >= 0) & (Y < dims[1]) & (Z >= 0) & (Z < dims[2])
            coef = -1.0 - (np.maximum(v, 0.0) if sgn < 0 else np.maximum(-v, 0.0))
            rows.append(gid[inside])
            cols.append((X + n * (Y + n * Z))[inside].astype(np.int64))
            vals.append(coef[inside])
            for c in range(3):
                rhs[c] += np.where(inside, coef * exact(c, X, Y, Z), 0.0)
    rows.append(gid.ravel())
    cols.append(gid.ravel())
    vals.append(diag.ravel())
    xs = [exact(c, xx, yy, zz) for c in range(3)]
    for c in range(3):
        rhs[c] += diag * xs[c]
    ilower, iupper = int(gid.min()), int(gid.max())
    A = mi.IJMatrix(ilower, iupper)
    A.set_values_coo(np.concatenate(rows), np.concatenate(cols), np.concatenate(vals))
    A.assemble()
    B = np.stack([r.ravel() for r in rhs])
    X = np.stack([x_.ravel() for x_ in xs])
    return A, B, X, ilower, iupper


def run_convdiff3(args, mi, dist, rank, world, transport, rehearsal, torch):
    """Side-line for BASELINE.json config 5: one step = the solve of all three components (one multivector solve, or
    three segregated solves on one hierarchy); value = 3 * N_global * iterations / t in GDOF/s."""
    n = args.n
    t0 = time.time()
    A, B, X, ilower, iupper = build_convdiff3(mi, n, rank, world)
    t_build = time.time() - t0
    nloc = iupper - ilower + 1
    ndof = n ** 3 * world
    amg = mi.BoomerAMG(print_level=1 if (rank == 0 and os.environ.get("MI_BENCH_VERBOSE")) else 0)
    bi = mi.BiCGSTAB(tolerance=args.tol, max_iterations=args.max_iter, print_level=0)
    bi.set_precond(amg)
    if args.segregated:
        bs = [mi.IJVector(ilower, iupper, B[c]) for c in range(3)]
        xv = [mi.IJVector(ilower, iupper, np.zeros(nloc)) for c in range(3)]
    else:
        bs = [mi.IJVector(ilower, iupper, B, ncomp=3)]
        xv = [mi.IJVector(ilower, iupper, np.zeros((3, nloc)), ncomp=3)]
    t0 = time.time()
    bi.setup(A, bs[0], xv[0])
    t_setup = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        mi.call("HYPRE_MI_StreamSynchronize")

    def one_step():
        its = 0
        for bvec, xvec in zip(bs, xv):
            xvec.fill(0.0)
            bi.solve(A, bvec, xvec)
            its += bi.num_iterations * (1 if args.segregated else 3)
        return its  # component-iterations

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    comp_iters = 0
    for _ in range(args.steps):
        comp_iters += one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    got = xv[0].get_all() if not args.segregated else np.stack([v.get() for v in xv])
    err = float(np.abs(got - X).max())
    if dist is not None:
        e = torch.tensor([err], dtype=torch.float64, device="cpu")
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        err = float(e.item())
    if rank == 0:
        out = {
            "metric": "BiCGSTAB+AMG solve GDOF/s on a 3-component convection-diffusion system (components * N_global * "
                      "iterations / t_solve), config-5 side-line",
            "value": ndof * comp_iters / elapsed / 1e9, "unit": "GDOF/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"convdiff3: 7-pt upwind convection-diffusion, {n}^3 rows per rank x {world} rank(s) "
                                   f"(N={ndof}), 3 components, BiCGSTAB+BoomerAMG tol {args.tol:g}, "
                                   f"{'segregated solves' if args.segregated else 'one multivector solve'}, x0=0",
                       "row_partition": f"{world} contiguous block-row slab(s)", "transport": transport},
            "iterations_per_solve": bi.num_iterations, "final_rel_residual": bi.final_rel_res,
            "max_abs_error_vs_exact": err, "amg_levels": amg.num_levels, "operator_complexity": amg.operator_complexity,
            "setup_s": t_setup, "build_s": t_build, "roofline": None, "cpu_baseline": None,
        }
        if rehearsal:
            out["rehearsal"] = True
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        mi.call("HYPRE_MI_CommFinalize")
        dist.destroy_process_group()


def physical_cores():
    """Physical cores this process may run on: distinct (package, core) pairs of /proc/cpuinfo among the CPUs of
    the affinity mask (SMT siblings count once)."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    pairs, cpu, pkg = set(), None, None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                cpu = int(line.split(":")[1])
            elif line.startswith("physical id"):
                pkg = int(line.split(":")[1])
            elif line.startswith("core id") and cpu in allowed:
                pairs.add((pkg, int(line.split(":")[1])))
    except (OSError, ValueError):
        pairs = set()
    cores = max(1, min(len(pairs) or len(allowed), len(allowed)))
    # a container's CPU quota (cgroup cpu.max) is what the threads can actually use: more threads than that are
    # throttled, not faster (the one-GPU boxes of this pool grant 16 CPUs of a 2 x 64-core host)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        cores = max(1, min(cores, int(quota + 0.5)))
    return cores


def mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_baseline(args, chunk):
    """The CPU side of the record.  A bounded sample first (256^3 when the host has the memory, else 128^3), in this
    process; then -- only when `--cpu-n` was left to the default, the sample was 256^3, its time projects to under 150 s at
    8x the rows and the host has >= 150 GB available -- the metric's own 512^3 in a CHILD process with a time limit
    (VERDICT r3 item 7), so that an out-of-memory kill or an overrun there costs the child, not the bench line."""
    n = args.cpu_n
    auto = n < 0
    if auto:
        n = 256 if mem_available_gb() >= 48.0 else 128
    res = cpu_baseline_at(args, chunk, n)
    if not (auto and n == 256 and args.n >= 512):
        return res
    projected = 8.0 * res["seconds"]
    if projected > 150.0 or mem_available_gb() < 150.0:
        res["sample"] += (f"; the metric's 512^3 was not attempted on the CPU (projected {projected:.0f} s, "
                          f"{mem_available_gb():.0f} GB of host memory available): 256^3 stands in, scaled by rows")
        return res
    import subprocess

    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-child", "512", "--stencil", str(args.stencil), "--kdim", str(args.kdim),
           "--tol", repr(args.tol), "--max-iter", str(args.max_iter), "--chunk", str(chunk)]
    why = ""
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode == 0 and lines:
            big = json.loads(lines[-1])
            big["sample"] += (f"; the bounded 256^3 sample before it: {res['value']:.4f} GDOF/s, {res['iterations']} iterations")
            return big
        why = f"exit code {p.returncode}: {(p.stderr or '')[-200:]}"
    except subprocess.TimeoutExpired:
        why = "not finished within 300 s"
    except Exception as e:  # noqa: BLE001
        why = f"{type(e).__name__}: {e}"[:200]
    res["sample"] += f"; the metric's 512^3 was attempted in a child process and did not complete ({why}): 256^3 stands in"
    return res


def cpu_baseline_at(args, chunk, n):
    """Oracle (CPU restatement of the HYPRE algorithm; libHYPRE is not available offline)
    timed on this host at n^3, OpenMP over all physical cores."""
    oc = ge.load_oracle()
    cores = physical_cores()
    oc.lib().oracle_set_threads(cores)
    A, b = oc.Csr.laplace(n, n, n, args.stencil)
    t0 = time.time()
    amg = oc.Amg(A, oc.default_params(gs_chunk=chunk))
    t_setup = time.time() - t0
    t0 = time.time()
    x, info = oc.gmres(A, b, kdim=args.kdim, tol=args.tol, maxit=args.max_iter, amg=amg)
    t_solve = time.time() - t0
    ndof = n ** 3
    cpu_model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # the same solve on ONE thread, for the per-core figure (only on the small sample: it would not fit the
    # few-minutes budget at 256^3)
    one_thread = ""
    if n <= 128:
        oc.lib().oracle_set_threads(1)
        t0 = time.time()
        _, info1 = oc.gmres(A, b, kdim=args.kdim, tol=args.tol, maxit=args.max_iter, amg=amg)
        t_solve1 = time.time() - t0
        oc.lib().oracle_set_threads(cores)
        one_thread = f"; on 1 thread: {t_solve1:.2f} s = {ndof * info1['iters'] / t_solve1 / 1e9:.4f} GDOF/s"
    return {
        "value": ndof * info["iters"] / t_solve / 1e9,
        "unit": "GDOF/s",
        "cores": cores,
        "kind": "port",
        "sample": f"laplace_3d {n}^3 {args.stencil}-pt, GMRES({args.kdim})+AMG tol {args.tol:g}: "
                  f"{info['iters']} iterations in {t_solve:.2f} s solve (+{t_setup:.2f} s setup on {min(cores, 32)} threads), "
                  f"rel res {info['rel_res']:.2e}; HYPRE-algorithm CPU restatement (oracle/), OpenMP {cores} threads = all "
                  f"physical cores this container may use (CPU quota / affinity; {os.cpu_count()} logical CPUs visible) on "
                  f"{cpu_model}{one_thread}",
        "iterations": info["iters"],
        "iterations_per_s": info["iters"] / t_solve,
        "seconds": t_setup + t_solve,
    }


def spawn_ranks(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as a CHILD torch.distributed.run (nothing
    in this process has touched the GPU yet -- no exec) and pass its output and exit code through."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    # torch.distributed.run's parser trips over an abbreviated "--n" even behind the script name: spell it out
    cmd += ["--grid" if a == "--n" else ("--grid=" + a[4:] if a.startswith("--n=") else a) for a in sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.cpu_child:
        # child of cpu_baseline(): the oracle alone, no GPU, no torch; one JSON line
        args.cpu_n = args.cpu_child
        print(json.dumps(cpu_baseline_at(args, args.chunk, args.cpu_child)), flush=True)
        return
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: never report an N-GPU request on {world} rank(s)")

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU path)")
    # MI_BENCH_SHARED_GPU=1: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (ranks
    # share devices, gloo transport through host callbacks -- RCCL refuses two ranks on one device).
    # The line it prints carries "rehearsal": true and is not a measurement.
    rehearsal = world > 1 and os.environ.get("MI_BENCH_SHARED_GPU") == "1"
    device_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_

        dist = dist_
        # control plane only (unique-id hand-off, barriers, the max over ranks of the elapsed time): gloo.  The data
        # path -- halo exchanges, all-reduces, all-gathers -- runs on the LIBRARY's own RCCL communicator, the only
        # RCCL communicator of the process (a second, torch-owned one would share the GPUs' channels for nothing)
        dist.init_process_group(backend="gloo")

    mi = ge.load_binding()
    mi.init()
    transport = "self"
    if rehearsal:
        transport = "REHEARSAL: gloo host callbacks, ranks share a GPU"
        mi.init_comm_torch(dist, device=None)
    elif world > 1:
        # ncclUniqueId from rank 0 to everyone, then the library opens its own RCCL communicator
        transport = "rccl (library communicator: ncclSend/ncclRecv halo groups, ncclAllReduce dots); control plane gloo"
        ok = 1
        try:
            idbuf = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                raw = (C.c_ubyte * 128)()
                mi.call("HYPRE_MI_CommGetUniqueId", raw)
                idbuf = torch.tensor(list(raw), dtype=torch.uint8)
            dist.broadcast(idbuf, src=0)
            raw = (C.c_ubyte * 128)(*idbuf.tolist())
            mi.call("HYPRE_MI_CommInitRCCL", raw, rank, world)
        except Exception as e:  # noqa: BLE001
            ok = 0
            print(f"[bench rank {rank}] library RCCL communicator failed: {e}", file=sys.stderr, flush=True)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            # LOUD fallback: the same collectives through torch.distributed (gloo), staged through host callbacks
            # (much slower; reported in the JSON so that the line cannot pass for an RCCL measurement)
            transport = "FALLBACK torch.distributed(gloo) host callbacks -- library RCCL communicator FAILED, see stderr"
            if rank == 0:
                print("[bench] WARNING: falling back to the torch.distributed transport", file=sys.stderr, flush=True)
            mi.call("HYPRE_MI_CommFinalize")
            mi.init_comm_torch(dist, device=None)

    if world > 1 and args.halo_transport == "ipc":
        mi.call("HYPRE_MI_CommEnablePeerStoreExchange", mi.c_big(0))
        transport = "halo updates: peer stores into hipIpc-mapped mailboxes (one launch per exchange); " + transport

    if args.workload == "convdiff3":
        return run_convdiff3(args, mi, dist, rank, world, transport, rehearsal, torch)

    n = args.n
    ndof = n ** 3
    t0 = time.time()
    A, b, x, _ = mi.build_laplace_system(n, n, n, args.stencil, rank, world)
    t_build = time.time() - t0
    amg_kw = {}
    for kv in args.amg:
        k_, v_ = kv.split("=", 1)
        amg_kw[k_] = float(v_) if any(c in v_ for c in ".eE") else int(v_)
    amg = mi.BoomerAMG(print_level=1 if (rank == 0 and os.environ.get("MI_BENCH_VERBOSE")) else 0, **amg_kw)
    gm = mi.GMRES(tolerance=args.tol, max_iterations=args.max_iter, kspace=args.kdim, print_level=0)
    gm.set_precond(amg)
    t0 = time.time()
    gm.setup(A, b, x)  # "Preconditioner setup" timer of the reference (HypreSystem.cpp:685-696)
    t_setup = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        mi.call("HYPRE_MI_StreamSynchronize")

    def one_solve():
        x.fill(0.0)  # x0 = 0 (HypreSystem.cpp:580)
        gm.solve(A, b, x)
        return gm.num_iterations

    for _ in range(args.warmup):
        one_solve()
    cap = 8192
    # HIP events in the timed region only around the dominant kernel (the level-0 SpMV of the GMRES loop, ~20 launches
    # per solve).  An event pair costs the queue ~10 us (profiles/r03_rank_size_proxy_256.txt: the gaps in front of every
    # timed launch); around the ~270 relaxation and Gram-Schmidt launches of a solve that is 0.5 % of a 512^3 solve on
    # one GPU but ~3 % of the same solve on 8 -- those classes are timed in one extra solve AFTER the timed region
    mi.profile_enable(mi.PROF_SPMV_L0, cap)
    comm_names = ("halo_exchange", "allreduce", "allgather", "matvec_overlapped", "gs_overlapped", "gs_in_order")

    def comm_counters():
        v = C.c_longlong()
        out_ = {}
        for nm_ in comm_names:
            mi.call("HYPRE_MI_GetCounter", nm_.encode(), C.byref(v))
            out_[nm_] = v.value
        return out_

    barrier()
    c0 = comm_counters()
    t0 = time.perf_counter()
    iters_total = 0
    for _ in range(args.steps):
        iters_total += one_solve()
    barrier()
    elapsed = time.perf_counter() - t0
    c1 = comm_counters()
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    spmv_n, spmv_ms, spmv_min = mi.profile_get(mi.PROF_SPMV_L0)
    spmv_kernel = mi.profile_kernel_name(mi.PROF_SPMV_L0)
    rel_res = gm.final_rel_res
    iters = gm.num_iterations
    rel_n = rel_ms = rel_min = dot_n = dot_ms = axpy_n = axpy_ms = 0
    relax_kernel = ""
    extra_solves = 0
    if world == 1:
        # the other two classes of the record (level-0 relaxation, Gram-Schmidt), outside the timed region
        mi.profile_enable(mi.PROF_SPMV_L0, 0)
        mi.profile_enable(mi.PROF_RELAX_L0, cap)
        mi.profile_enable(mi.PROF_DOT, 4 * cap)   # fused Gram-Schmidt steps (axpy + inner product) and norms
        mi.profile_enable(mi.PROF_AXPY, cap)
        extra_solves = 1
        one_solve()
        rel_n, rel_ms, rel_min = mi.profile_get(mi.PROF_RELAX_L0)
        dot_n, dot_ms, _ = mi.profile_get(mi.PROF_DOT)
        axpy_n, axpy_ms, _ = mi.profile_get(mi.PROF_AXPY)
        relax_kernel = mi.profile_kernel_name(mi.PROF_RELAX_L0)
        mi.profile_enable(mi.PROF_DOT, 0)
        mi.profile_enable(mi.PROF_AXPY, 0)
        mi.profile_enable(mi.PROF_SPMV_L0, cap)  # (the general-operator leg below reads classes 0 and 1)
    xs = x.get()
    err = float(np.abs(xs - 1.0).max()) if xs.size else 0.0
    nlev = amg.num_levels
    opcx = amg.operator_complexity

    # algorithmic bytes of one level-0 launch on this rank (DESIGN.md "Kernels")
    nr, nc, nnz = C.c_int(), C.c_int(), C.c_longlong()
    mi.call("HYPRE_MI_BoomerAMGGetLevelCSRSize", amg.h, 0, 0, C.byref(nr), C.byref(nc), C.byref(nnz))
    nloc, nnz_loc = nr.value, nnz.value
    spmv_bytes = 12.0 * nnz_loc + 20.0 * nloc
    # one relaxation launch sweeps the C rows or the F rows (C-first ordering: rows [0, n_C) / [n_C, n)): the swept
    # rows' entries once (forward + backward sweep out of one read), u_old in (16 N: the gathers' source, read as a
    # whole vector, and the copy-through), u_new out / f / divisor of the swept rows (16 N_sel), the C/F marker (N).
    # C and F launches alternate (F then C in every up leg), so the per-launch mean is the mean of the two
    mi.call("HYPRE_MI_BoomerAMGGetLevelCSRSize", amg.h, 0, 9, C.byref(nr), C.byref(nc), C.byref(nnz))
    n_c, nnz_c = nr.value, nnz.value
    n_f, nnz_f = nloc - n_c, nnz_loc - nnz_c
    relax_bytes_c = 12.0 * nnz_c + 16.0 * nloc + 16.0 * n_c + nloc
    relax_bytes_f = 12.0 * nnz_f + 16.0 * nloc + 16.0 * n_f + nloc
    relax_bytes = 0.5 * (relax_bytes_c + relax_bytes_f)

    out, roof = None, None
    if rank == 0:
        # HBM traffic per launch comes from a separate rocprofv3 --pmc pass (the counters cannot be read from
        # inside this process): the newest recorded figure for this configuration, labelled as recorded
        def recorded_traffic(field):
            for tname in ("traffic_r04.json", "traffic_r03.json", "traffic_r02.json", "traffic_r01.json"):
                tpath = os.path.join(ROOT, "profiles", tname)
                if os.path.exists(tpath):
                    try:
                        v = json.load(open(tpath)).get(f"{n}^3/{args.stencil}pt/{world}gpu", {}).get(field)
                    except Exception:
                        v = None
                    if v is not None:
                        return v, (f"RECORDED in profiles/{tname} by a separate rocprofv3 --pmc pass "
                                   "(FETCH_SIZE x2 + WRITE_SIZE per launch), not observed by this run")
            return None, None

        traffic, traffic_source = recorded_traffic("spmv_hbm_bytes_per_launch")
        if spmv_n:
            a = spmv_bytes / (spmv_ms / spmv_n * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": f"{spmv_kernel} (level-0 CSR SpMV of the GMRES loop, LDS x-cache variant; template "
                              "flags <epilogue, level-0 tag, one-byte value dictionary, workgroup size>; the loop "
                              "runs in the preconditioner's own ordering of level 0 -- graph-clustered internal numbering, C points "
                              "first -- so a V-cycle needs no gather / scatter)", "achieved": a,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": traffic_source, "launches": spmv_n, "avg_ms": spmv_ms / spmv_n, "min_ms": spmv_min,
                    "algorithmic_bytes_per_launch": spmv_bytes}
            if traffic is not None and traffic < spmv_bytes:
                roof["note"] = ("recorded traffic below the algorithmic byte count: the level-0 operator of this stencil has "
                                "<= 256 distinct values and streams one-byte dictionary indices (DESIGN.md section 5), while "
                                "12 nnz + 20 N prices 8-byte values; MI_HYPRE_VALUE_DICT=0 gives the plain stream")
        roof_relax = None
        if rel_n and world == 1:  # N > 1 cuts a pass into up to three launches (halo overlap): no per-launch figure
            a = relax_bytes / (rel_ms / rel_n * 1e-3) / 1e9
            roof_relax = {"bound": "hbm", "kernel": f"{relax_kernel} (level-0 l1 hybrid GS, one C or F pass over the full operator = the "
                                    "up-leg sweeps; the down leg's sweep starts from a zero guess, runs on the zero-guess "
                                    "sub-operator and is not counted here; HIP events in one extra solve after the timed region)",
                          "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                          "launches": rel_n, "avg_ms": rel_ms / rel_n, "min_ms": rel_min,
                          "algorithmic_bytes_per_launch": relax_bytes,
                          "byte_model": {"formula": "12 nnz_sel + 16 N + 16 N_sel + N per pass; mean of the C pass and the F pass "
                                                    "(they alternate); counts from HYPRE_MI_BoomerAMGGetLevelCSRSize(which=9)",
                                         "n_C": n_c, "nnz_C_rows": nnz_c, "n_F": n_f, "nnz_F_rows": nnz_f,
                                         "bytes_C_pass": relax_bytes_c, "bytes_F_pass": relax_bytes_f}}
        # modified Gram-Schmidt of the GMRES loop (second-largest block of a solve after level 0): the fused
        # axpy + inner-product steps and norms (class "dot"), and the plain axpys of the solution update
        gram = None
        if dot_n and world == 1:
            # step i of a cycle: i fused steps of 32 N bytes (read p_j, read + write p_i, 8 N each; <p_j+1, p_i> rides on
            # the same pass) + the norm; one solve of m steps: m(m+1)/2 fused steps + ~m + 3 norms / dots of 8..16 N
            m_it = iters
            gs_bytes = (m_it * (m_it + 1) / 2.0) * 32.0 * nloc + (m_it + 3) * 16.0 * nloc
            per_solve_ms = dot_ms / extra_solves
            gram = {"what": "modified Gram-Schmidt + norms of one solve (HIP events of the fused axpy+dot / dot launches, in one "
                            "extra solve after the timed region)",
                    "ms_per_solve": per_solve_ms, "launches_per_solve": dot_n / extra_solves,
                    "share_of_solve": per_solve_ms / (elapsed / args.steps * 1e3),
                    "algorithmic_bytes_per_solve": gs_bytes,
                    "achieved": gs_bytes / (per_solve_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gs_bytes / (per_solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "axpy_ms_per_solve": axpy_ms / extra_solves}
        chunk = C.c_int()
        mi.call("HYPRE_MI_GetGSChunk", C.byref(chunk))
        out = {
            "metric": "GMRES+AMG solve GDOF/s (N_global*iterations/t_solve; iterations/sec in iterations_per_s), %d^3 Laplacian" % n,
            "value": ndof * iters_total / elapsed / 1e9,
            "unit": "GDOF/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"laplace_3d {n}^3 {args.stencil}-pt (N={ndof}), GMRES({args.kdim})+BoomerAMG "
                            f"(PMIS, ext+i, C/F l1-hybrid-SGS chunk {chunk.value}, V(1,1)), tol {args.tol:g}, x0=0"
                            + (f"; SIDE-LINE with boomeramg_settings overrides {amg_kw}" if amg_kw else ""),
                "row_partition": f"{world} contiguous block-row slab(s)",
                "transport": transport,
            },
            "iterations_per_solve": iters,
            "iterations_per_s": iters_total / elapsed,
            "final_rel_residual": rel_res,
            "max_abs_error_vs_ones": err,
            "amg_levels": nlev,
            "operator_complexity": opcx,
            "internal_locality_numbering": bool(mi.BoomerAMG.input_ordering(amg)[0]),
            "setup_s": t_setup,
            "build_s": t_build,
            "roofline": roof,
            "roofline_relax": roof_relax,
            "gram_schmidt": gram,
        }
        # what one solve asks of the interconnect, counted by the library on this rank inside the timed region
        # (HYPRE_MI_GetCounter; VERDICT r3 item 6: a multi-GPU run should explain its own efficiency).  Per solve of m
        # iterations: 3 + m + m(m+1)/2 scalar all-reduces (modified Gram-Schmidt: one per coefficient), <= 7 neighbour
        # exchanges per distributed level and cycle + 1 per GMRES matvec, one all-gather per cycle (redundant tail).
        m_it = iters
        out["comm_ops"] = {
            "what": "collective operations of ONE solve on rank 0 (timed region / steps), from the library's counters",
            **{k_: (c1[k_] - c0[k_]) / max(1, args.steps) for k_ in comm_names},
            "host_synchronisations": m_it + 4,  # one per Arnoldi step (Hessenberg column), the norms of b, r0 and the final residual
            "expected_allreduce": (3 + m_it + m_it * (m_it + 1) // 2) if world > 1 else 0,
            "note": "halo_exchange counts neighbour exchanges (matvec + relaxation + transfer operators); matvec_overlapped / "
                    "gs_overlapped went beside the diag-block kernel on the side stream, gs_in_order did not; all zero on one rank.  "
                    "Kernel launches are not counted in-process: a rocprofv3 trace of this workload on one GPU has 1324 launches per "
                    "solve, 440 of them shorter than 25 us (4.7 ms together), and 0.35 ms per solve outside kernels "
                    "(RECORDED: profiles/r04_gaps_512.txt)",
        }
        if rehearsal:
            out["rehearsal"] = True
    # ---- side-line (N = 1): the reference's other Krylov choice on the SAME hierarchy, method: cogmres
    # (/root/reference/src/HypreSystem.cpp:372-388) -- GMRES with block classical Gram-Schmidt: one pass of inner products
    # and one block update per step instead of i dependent fused passes.  NOT the headline (the headline is method: gmres).
    if world == 1 and not args.no_general and roof is not None:
        cg = mi.COGMRES(tolerance=args.tol, max_iterations=args.max_iter, kspace=args.kdim, print_level=0)
        cg.set_precond(amg)
        cg.setup(A, b, x)

        def cg_solve():
            x.fill(0.0)
            cg.solve(A, b, x)
            return cg.num_iterations

        cg_solve()
        barrier()
        t0 = time.perf_counter()
        c_steps = 3
        c_iters = sum(cg_solve() for _ in range(c_steps))
        barrier()
        c_elapsed = time.perf_counter() - t0
        out["sideline_cogmres"] = {
            "what": "SIDE-LINE, not the headline: solver_settings method cogmres on the same hierarchy",
            "ms_per_step": c_elapsed / c_steps * 1e3, "iterations_per_solve": cg.num_iterations,
            "value_gdofs": ndof * c_iters / c_elapsed / 1e9, "final_rel_residual": cg.final_rel_res,
            "max_abs_error_vs_ones": float(np.abs(x.get() - 1.0).max()), "steps": c_steps,
            "time_to_solution_vs_headline": (c_elapsed / c_steps) / (elapsed / args.steps)}
        cg.destroy()
    # ---- general-operator leg (N = 1): the headline operator has two distinct values (6, -1), so its level-0 kernels
    # stream one-byte dictionary indices; a variable-coefficient operator (BASELINE.json configs 4 / 5) cannot.  The
    # same problem is set up again with the dictionary off and the same kernel classes are timed on a few solves:
    # "roofline_general" is what a general matrix of this size and sparsity gets.  Outside the timed region.
    if world == 1 and not args.no_general and roof is not None:
        gm.destroy()
        amg.destroy()
        mi.call("HYPRE_MI_SetValueDictionary", 0)
        amg = mi.BoomerAMG(print_level=0, **amg_kw)
        gm = mi.GMRES(tolerance=args.tol, max_iterations=args.max_iter, kspace=args.kdim, print_level=0)
        gm.set_precond(amg)
        t0 = time.time()
        gm.setup(A, b, x)
        t_setup_g = time.time() - t0
        one_solve()
        mi.profile_reset()
        barrier()
        t0 = time.perf_counter()
        g_steps = 3
        g_iters = sum(one_solve() for _ in range(g_steps))
        barrier()
        g_elapsed = time.perf_counter() - t0
        gs_n, gs_ms, gs_min = mi.profile_get(mi.PROF_SPMV_L0)
        gr_n, gr_ms, gr_min = mi.profile_get(mi.PROF_RELAX_L0)
        g_traffic, g_src = recorded_traffic("spmv_general_hbm_bytes_per_launch")
        a = spmv_bytes / (gs_ms / gs_n * 1e-3) / 1e9
        ar = relax_bytes / (gr_ms / gr_n * 1e-3) / 1e9
        out["roofline_general"] = {
            "what": "the same solve with the value dictionary off (HYPRE_MI_SetValueDictionary(0)): 8-byte values in the "
                    "level-0 matrix stream, as for any variable-coefficient operator; results identical bit for bit",
            "bound": "hbm", "kernel": mi.profile_kernel_name(mi.PROF_SPMV_L0), "achieved": a, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": a / HBM_PEAK_GBS, "traffic": g_traffic, "traffic_source": g_src,
            "launches": gs_n, "avg_ms": gs_ms / gs_n, "min_ms": gs_min, "algorithmic_bytes_per_launch": spmv_bytes,
            "relax": {"kernel": mi.profile_kernel_name(mi.PROF_RELAX_L0), "achieved": ar, "frac": ar / HBM_PEAK_GBS,
                      "launches": gr_n, "avg_ms": gr_ms / gr_n, "algorithmic_bytes_per_launch": relax_bytes},
            "ms_per_step": g_elapsed / g_steps * 1e3, "value_gdofs": ndof * g_iters / g_elapsed / 1e9,
            "iterations_per_solve": gm.num_iterations, "final_rel_residual": gm.final_rel_res, "steps": g_steps,
            "setup_s": t_setup_g}
        mi.call("HYPRE_MI_SetValueDictionary", 1)
    # ---- side-line (N = 1): the reference's own knob against the hierarchy's weight, non_galerkin_tol
    # (/root/reference/src/HypreSystem.cpp:161-176), on the same problem.  NOT the headline: `value` above is the
    # Galerkin hierarchy of the app defaults.  Outside the timed region.
    if world == 1 and not args.no_general and roof is not None and not amg_kw and args.sideline_non_galerkin > 0.0:
        gm.destroy()
        amg.destroy()
        amg = mi.BoomerAMG(print_level=0, non_galerkin_tol=args.sideline_non_galerkin)
        gm = mi.GMRES(tolerance=args.tol, max_iterations=args.max_iter, kspace=args.kdim, print_level=0)
        gm.set_precond(amg)
        t0 = time.time()
        gm.setup(A, b, x)
        t_setup_n = time.time() - t0
        one_solve()
        barrier()
        t0 = time.perf_counter()
        n_steps = 3
        n_iters = sum(one_solve() for _ in range(n_steps))
        barrier()
        n_elapsed = time.perf_counter() - t0
        xs_n = x.get()
        out["sideline_non_galerkin"] = {
            "what": f"SIDE-LINE, not the headline: the same solve with boomeramg_settings non_galerkin_tol = "
                    f"{args.sideline_non_galerkin:g} (coarse operators sparsified after the Galerkin product: entries below "
                    "tol * min(row maxima) lumped onto the diagonal, DESIGN.md section 9)",
            "ms_per_step": n_elapsed / n_steps * 1e3, "iterations_per_solve": gm.num_iterations,
            "value_gdofs": ndof * n_iters / n_elapsed / 1e9, "final_rel_residual": gm.final_rel_res,
            "max_abs_error_vs_ones": float(np.abs(xs_n - 1.0).max()), "operator_complexity": amg.operator_complexity,
            "amg_levels": amg.num_levels, "setup_s": t_setup_n, "steps": n_steps,
            "time_to_solution_vs_headline": (n_elapsed / n_steps) / (elapsed / args.steps)}
    # ---- side-line (N > 1, headline over RCCL / callbacks): the SAME solves with the halo updates and the scalar
    # all-reduces on the peer-store transport (hipIpc mailboxes).  It runs LAST, behind a probe with a
    # short bound on every wait and a collective go / no-go after each step, so that a transport that does not work on
    # this machine costs a few seconds and one field of the JSON line, never the headline above.
    if rank == 0:
        if not args.no_cpu and args.cpu_n != 0 and world == 1:  # rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args, chunk.value)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)  # THE line -- before anything that could take the process down
    if world > 1 and args.halo_transport == "rccl" and args.ipc_sideline and not amg_kw:
        side = peer_store_sideline(mi, dist, torch, rank, world, one_solve, barrier, args.steps, iters, rel_res,
                                   elapsed / args.steps, ndof)
        if rank == 0:
            print("[sideline_peer_store] " + json.dumps(side), flush=True)
    if dist is not None:
        dist.barrier()
        mi.call("HYPRE_MI_CommFinalize")
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
