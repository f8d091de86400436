"""N>1 path: world_size-2/3 runs of tests/dist_worker.py.

CPU (gloo): the host half of the distributed path -- partition, IJ assembly, halo
plan, rank-local coarsening, P-row exchange, Galerkin product -- against the
oracle's emulation of the same partition.
GPU: the same plus the device solve with several ranks sharing the GPU.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _run(nproc, mode, n, stencil, port, staging="host"):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MI_HYPRE_HOST_THREADS"] = "2"
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER, "--mode", mode, "--grid", str(n),
           "--stencil", str(stencil), "--staging", staging]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-4000:]
    return p.stdout


@pytest.mark.parametrize("nproc,n,stencil", [(2, 12, 7), (3, 10, 27)])
def test_host_setup_world_size_n_gloo(nproc, n, stencil):
    out = _run(nproc, "host", n, stencil, 29611 + nproc)
    assert "dist host setup ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,n,stencil", [(2, 16, 7), (4, 12, 7), (3, 10, 27)])
def test_device_solve_shared_gpu(nproc, n, stencil):
    out = _run(nproc, "solve", n, stencil, 29631 + nproc)
    assert "dist solve ok" in out


@pytest.mark.gpu
def test_device_solve_cuda_staged_transport():
    """bench.py's fallback transport (torch.distributed collectives on device tensors behind the callback
    interface), exercised here over gloo because two nccl ranks cannot share the one GPU of the test box."""
    out = _run(2, "solve", 12, 7, 29651, staging="cuda")
    assert "dist solve ok" in out
