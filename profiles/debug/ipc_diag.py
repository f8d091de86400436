"""diagnostic: does the peer-store transport fail because two processes' kernels do not run concurrently on a shared
GPU, or because the flag words are not visible?  Phase A: sends only, host barrier, recvs only (no kernel ever waits
for a kernel that has not run yet).  Phase B: send + recv in one call (kernels of two processes must overlap)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import torch, torch.distributed as dist
dist.init_process_group(backend="gloo")
rank, size = dist.get_rank(), dist.get_world_size()
mi = ge.load_binding(); mi.init(); mi.init_comm_torch(dist)
mi.call("HYPRE_MI_CommEnablePeerStoreExchange", mi.c_big(4096))
C = mi.C
peer = 1 - rank
n = 1000
s = torch.full((n,), rank + 1, dtype=torch.uint8, device="cuda"); r = torch.zeros(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
def ex(ns, nr):
    t0 = time.time()
    try:
        mi.call("HYPRE_MI_CommExchangeDevice", ns, (C.c_int * 1)(peer), (C.c_void_p * 1)(s.data_ptr()), (C.c_size_t * 1)(n),
                nr, (C.c_int * 1)(peer), (C.c_void_p * 1)(r.data_ptr()), (C.c_size_t * 1)(n))
        return "ok %.3f s" % (time.time() - t0)
    except Exception as e:
        mi.call("HYPRE_ClearAllErrors")
        return "FAILED after %.1f s" % (time.time() - t0)
a = ex(1, 0); dist.barrier(); b = ex(0, 1)
print(f"[rank {rank}] phase A (separate): send {a}, recv {b}, data ok {bool((r.cpu().numpy() == peer + 1).all())}", flush=True)
dist.barrier()
r.zero_(); torch.cuda.synchronize()
c = ex(1, 1)
print(f"[rank {rank}] phase B (one call): {c}, data ok {bool((r.cpu().numpy() == peer + 1).all())}", flush=True)
dist.barrier()
mi.call("HYPRE_MI_CommFinalize"); dist.destroy_process_group()
