"""bench.py contract: one JSON line with the agreed keys at N = 1, and the N > 1 code path (rehearsed with
ranks sharing the one GPU of the test box over gloo callbacks -- RCCL refuses two ranks on one device)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--grid", "64", "--cpu-n", "32"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:]
    j = _json_line(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["unit"] == "GDOF/s"
    assert j["vs_baseline"] is None and j["dtype"] == "f64" and j["data"] == "synthetic" and "workload" in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the kernel name comes from the launch itself, template flags included (value dictionary on for the 6 / -1 stencil)
    assert r["kernel"].startswith("spmv_stream_xc<0, 1, true, 256>")
    rr = j["roofline_relax"]
    bm = rr["byte_model"]
    assert bm["n_C"] + bm["n_F"] == 64 ** 3 and bm["nnz_C_rows"] + bm["nnz_F_rows"] == 7 * 64 ** 3 - 6 * 64 ** 2
    assert abs(rr["algorithmic_bytes_per_launch"] - 0.5 * (bm["bytes_C_pass"] + bm["bytes_F_pass"])) < 1.0
    assert rr["kernel"].startswith("gs_tile_k<true, 256>")
    # general-operator leg: dictionary off, same iterations and residual (same doubles), plain-stream kernels
    g = j["roofline_general"]
    assert g["kernel"].startswith("spmv_stream_xc<0, 1, false, 256>") and g["relax"]["kernel"].startswith("gs_tile_k<false, 256>")
    assert g["iterations_per_solve"] == j["iterations_per_solve"] and g["final_rel_residual"] == j["final_rel_residual"]
    assert abs(g["frac"] - g["achieved"] / g["peak"]) < 1e-12 and g["algorithmic_bytes_per_launch"] == r["algorithmic_bytes_per_launch"]
    # side-lines: the reference's other knobs on the same problem, labelled as such
    for key in ("sideline_cogmres", "sideline_non_galerkin"):
        sl = j[key]
        assert sl["what"].startswith("SIDE-LINE") and sl["final_rel_residual"] <= 1e-8 and sl["max_abs_error_vs_ones"] < 1e-5
        assert abs(sl["iterations_per_solve"] - j["iterations_per_solve"]) <= 4
    assert j["sideline_non_galerkin"]["operator_complexity"] < j["operator_complexity"]
    ms = j["gram_schmidt"]
    assert ms["ms_per_solve"] > 0 and 0 < ms["share_of_solve"] < 1
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    co = j["comm_ops"]  # one rank: nothing travels
    assert co["halo_exchange"] == 0 and co["allreduce"] == 0 and co["allgather"] == 0 and co["expected_allreduce"] == 0
    assert 5 <= j["iterations_per_solve"] <= 30 and j["final_rel_residual"] <= 1e-8


def test_bench_two_ranks_rehearsal():
    env = dict(os.environ)
    env["MI_BENCH_SHARED_GPU"] = "1"
    env["MI_BENCH_IPC_PROBE_TIMEOUT_MS"] = "20000"  # two ranks time-slice one device here
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29791", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--grid", "64", "--no-cpu", "--ipc-sideline"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:]
    j = _json_line(p.stdout)
    assert j["n_gpus"] == 2 and j.get("rehearsal") is True and j["cpu_baseline"] is None
    assert 5 <= j["iterations_per_solve"] <= 30 and j["final_rel_residual"] <= 1e-8
    assert j["max_abs_error_vs_ones"] < 1e-5
    # what one solve asks of the interconnect: the library's counters over the timed region
    co, m = j["comm_ops"], j["iterations_per_solve"]
    assert co["allreduce"] == co["expected_allreduce"] == 3 + m + m * (m + 1) // 2, co
    assert co["halo_exchange"] > m and co["allgather"] >= m and co["matvec_overlapped"] + co["gs_overlapped"] > 0, co
    # the opt-in N > 1 side-line: the same solves on the peer-store transport, behind its probe, printed AFTER the headline
    # as a second, tagged line (bench.py peer_store_sideline)
    tagged = [l for l in p.stdout.splitlines() if l.startswith("[sideline_peer_store] ")]
    assert len(tagged) == 1 and p.stdout.index(tagged[0]) > p.stdout.index('{"metric"'), p.stdout[-3000:]
    side = json.loads(tagged[0][len("[sideline_peer_store] "):])
    assert side["ran"] is True and side["same_iterations_as_headline"] is True, side
    assert "ipc-peer-store" in side["transport"] and side["ms_per_step"] > 0


def test_bench_two_ranks_rehearsal_peer_store_halo():
    """The same rehearsal with the halo updates on the hipIpc peer-store transport (`--halo-transport ipc`; reductions stay
    on the gloo callbacks here, on RCCL in a real multi-GPU run): same iteration count and residual as over the callbacks."""
    env = dict(os.environ)
    env["MI_BENCH_SHARED_GPU"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = []
    for port, extra in ((29793, []), (29795, ["--halo-transport", "ipc"])):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
               "--grid", "64", "--no-cpu"] + extra
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, cwd=ROOT)
        assert p.returncode == 0, p.stdout[-3000:]
        res.append(_json_line(p.stdout))
    assert "hipIpc" in res[1]["config"]["transport"] and "hipIpc" not in res[0]["config"]["transport"]
    assert res[0]["iterations_per_solve"] == res[1]["iterations_per_solve"]
    assert res[0]["final_rel_residual"] == res[1]["final_rel_residual"]  # the transport moves the same bytes: bitwise
