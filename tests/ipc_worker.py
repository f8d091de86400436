"""One rank of the peer-store transport test (launched by torch.distributed.run; the ranks share the visible GPU).

Every rank exchanges device buffers of many sizes with every other rank through HYPRE_MI_CommExchangeDevice on the
IPC-mailbox transport (HYPRE_MI_CommEnablePeerStoreExchange on top of gloo callbacks): empty messages, a few bytes,
sizes that are not multiples of 16, messages of several mailbox slots, unaligned device pointers, many rounds in a row
(slot reuse: message k + 2 waits for the acknowledgement of message k).  Payloads are a function of (sender, receiver,
round, size), so a byte that lands in the wrong place or round is seen."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def payload(src, dst, rnd, nbytes):
    rng = np.random.default_rng(1000003 * src + 1009 * dst + rnd)
    return rng.integers(0, 256, size=nbytes, dtype=np.uint8)


def scenario_refuse(mi, dist, rank, size):
    """Ranks that (claim to) sit on different devices with mailboxes in ordinary device memory: the transport must be
    refused on every rank and the communicator must keep working on what it had."""
    import torch

    os.environ["MI_HYPRE_IPC_BUS_ID"] = f"test-device-{rank}"  # (test hook: the identity a rank publishes)
    os.environ["MI_HYPRE_IPC_FINEGRAINED"] = "0"
    C = mi.C
    try:
        mi.call("HYPRE_MI_CommEnablePeerStoreExchange", mi.c_big(4096))
        raise AssertionError("coarse-grained mailboxes across devices were accepted")
    except mi.HypreError as e:
        assert "refused" in str(e) and "different devices" in str(e), str(e)
    nm = C.create_string_buffer(128)
    mi.call("HYPRE_MI_CommName", nm, 128)
    assert not nm.value.decode().startswith("ipc-peer-store"), nm.value
    t = torch.full((3,), float(rank + 1), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    mi.call("HYPRE_MI_CommAllreduceDevice", C.c_void_p(t.data_ptr()), 3)
    assert np.allclose(t.cpu().numpy(), size * (size + 1) / 2.0)
    # with one device identity for everybody the same ordinary-memory mailboxes are fine (ranks share the GPU)
    os.environ["MI_HYPRE_IPC_BUS_ID"] = "test-device-shared"
    mi.call("HYPRE_MI_CommEnablePeerStoreExchange", mi.c_big(4096))
    mi.call("HYPRE_MI_CommName", nm, 128)
    assert nm.value.decode().startswith("ipc-peer-store") and "coarse-grained" in nm.value.decode(), nm.value
    if rank == 0:
        print(f"ipc refusal ok: {size} ranks")


def scenario_gate(mi, dist, rank, size):
    """A wait that expires must not pass for a result: after rank 0 has waited in vain for a message (error flag latched
    on rank 0 only), the next solve fails on EVERY rank through the plain HYPRE entry point and x is poisoned."""
    import torch

    os.environ["MI_HYPRE_IPC_TIMEOUT_MS"] = "300"
    C = mi.C
    mi.call("HYPRE_MI_CommEnablePeerStoreExchange", mi.c_big(1 << 16))
    n = 12
    A, b, x, _ = mi.build_laplace_system(n, n, n, 7, rank, size)
    amg = mi.BoomerAMG(print_level=0)
    gm = mi.GMRES(tolerance=1e-8, max_iterations=40, kspace=20, print_level=0)
    gm.set_precond(amg)
    gm.setup(A, b, x)
    gm.solve(A, b, x)
    assert gm.final_rel_res < 1e-8 and np.abs(x.get() - 1.0).max() < 1e-5
    dist.barrier()
    if rank == 0:
        r = torch.zeros(64, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        try:
            mi.call("HYPRE_MI_CommExchangeDevice", 0, None, None, None, 1, (C.c_int * 1)(1), (C.c_void_p * 1)(r.data_ptr()),
                    (C.c_size_t * 1)(64))
            raise AssertionError("a message nobody sent arrived")
        except mi.HypreError as e:
            assert "did not arrive" in str(e), str(e)
    dist.barrier()
    x.fill(0.0)
    try:
        gm.solve(A, b, x)
        raise AssertionError(f"rank {rank}: a solve on a transport with an expired wait returned normally")
    except mi.HypreError as e:
        assert "peer-store transport" in str(e), str(e)
    xs = x.get()
    assert xs.size == 0 or np.all(np.isnan(xs)), "the result of the failed solve was not poisoned"
    if rank == 0:
        print(f"ipc gate ok: {size} ranks")


def main():
    import torch
    import torch.distributed as dist

    dist.init_process_group(backend="gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    mi = ge.load_binding()
    mi.init()
    mi.init_comm_torch(dist)
    scenario = sys.argv[1] if len(sys.argv) > 1 else "raw"
    if scenario in ("refuse", "gate"):
        (scenario_refuse if scenario == "refuse" else scenario_gate)(mi, dist, rank, size)
        dist.barrier()
        mi.call("HYPRE_MI_CommFinalize")
        dist.destroy_process_group()
        return
    slot = 4096
    mi.call("HYPRE_MI_CommEnablePeerStoreExchange", mi.c_big(slot))
    C = mi.C
    peers = [r for r in range(size) if r != rank]
    sizes = [0, 1, 8, 24, 100, 4096, 4097, 3 * slot + 5, 65536 + 8]
    rounds = 0
    for rnd in range(3 * len(sizes)):
        nbytes = sizes[rnd % len(sizes)]
        off = rnd % 3  # 0: aligned; 1, 2: device pointers that are not multiples of 16 (byte-wise copy path)
        sbuf = {p: torch.zeros(nbytes + 16, dtype=torch.uint8, device="cuda") for p in peers}
        rbuf = {p: torch.full((nbytes + 16,), 255, dtype=torch.uint8, device="cuda") for p in peers}
        for p in peers:
            sbuf[p][off:off + nbytes] = torch.from_numpy(payload(rank, p, rnd, nbytes)).cuda()
        torch.cuda.synchronize()
        n = len(peers)
        ip = (C.c_int * n)(*peers)
        sp = (C.c_void_p * n)(*[sbuf[p].data_ptr() + off for p in peers])
        rp = (C.c_void_p * n)(*[rbuf[p].data_ptr() + off for p in peers])
        nb = (C.c_size_t * n)(*([nbytes] * n))
        mi.call("HYPRE_MI_CommExchangeDevice", n, ip, sp, nb, n, ip, rp, nb)
        for p in peers:
            got = rbuf[p].cpu().numpy()
            assert np.array_equal(got[off:off + nbytes], payload(p, rank, rnd, nbytes)), (rank, p, rnd, nbytes)
            assert np.all(got[:off] == 255) and np.all(got[off + nbytes:] == 255), (rank, p, rnd, "wrote outside")
        rounds += 1
    # one-sided shapes: a ring (send to the right neighbour only, receive from the left one)
    if size > 2:
        right, left = (rank + 1) % size, (rank - 1) % size
        for rnd in range(100, 104):
            nbytes = 2 * slot + 40
            s = torch.from_numpy(payload(rank, right, rnd, nbytes)).cuda()
            r = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            mi.call("HYPRE_MI_CommExchangeDevice", 1, (C.c_int * 1)(right), (C.c_void_p * 1)(s.data_ptr()), (C.c_size_t * 1)(nbytes),
                    1, (C.c_int * 1)(left), (C.c_void_p * 1)(r.data_ptr()), (C.c_size_t * 1)(nbytes))
            assert np.array_equal(r.cpu().numpy(), payload(left, rank, rnd, nbytes))
            rounds += 1
    # ---- scalar all-reduces (the Krylov loops' inner products): 1..8 doubles by peer stores, summed in rank order
    # (so the bits are those of the loop below on every rank); 9 doubles go through the wrapped communicator
    for rnd in range(40):
        cnt = 1 + rnd % 9
        vals = [np.random.default_rng(7919 * r + rnd).standard_normal(cnt) * 10.0 ** (rnd % 5) for r in range(size)]
        t = torch.from_numpy(vals[rank].copy()).cuda()
        torch.cuda.synchronize()
        mi.call("HYPRE_MI_CommAllreduceDevice", C.c_void_p(t.data_ptr()), cnt)
        got = t.cpu().numpy()
        want = vals[0].copy()
        for r in range(1, size):
            want = want + vals[r]
        if cnt <= 8:
            assert np.array_equal(got, want), (rank, rnd, cnt, got, want)
        else:
            assert np.allclose(got, want, rtol=1e-14, atol=0.0), (rank, rnd, cnt)
    mi.call("HYPRE_MI_CommCheck")
    if rank == 0:
        print(f"ipc exchange ok: {size} ranks, {rounds} rounds")
    dist.barrier()
    mi.call("HYPRE_MI_CommFinalize")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
