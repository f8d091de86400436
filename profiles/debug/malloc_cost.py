"""hipMalloc / hipFree cost against size on the box, fresh and repeated, and beside a running kernel (round 4:
the setup trace showed ~30 ms of host time per GB in front of the first kernel that touches a fresh allocation)."""
import ctypes as C, time, threading, sys
hip = C.CDLL("libamdhip64.so")
def malloc(nbytes):
    p = C.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)); t = time.perf_counter() - t0
    assert rc == 0, rc
    return p, t
def free(p):
    t0 = time.perf_counter(); hip.hipFree(p); return time.perf_counter() - t0
hip.hipSetDevice(0)
hip.hipDeviceSynchronize()
GB = 1 << 30
for rep in range(2):
    for sz in (0.25, 1, 4, 16):
        p, tm = malloc(int(sz * GB))
        t0 = time.perf_counter(); hip.hipMemset(p, 0, C.c_size_t(int(sz * GB))); hip.hipDeviceSynchronize(); ts = time.perf_counter() - t0
        t0 = time.perf_counter(); hip.hipMemset(p, 0, C.c_size_t(int(sz * GB))); hip.hipDeviceSynchronize(); ts2 = time.perf_counter() - t0
        tf = free(p)
        print(f"rep {rep} size {sz:5.2f} GiB: hipMalloc {tm*1e3:8.2f} ms ({tm*1e3/sz:6.1f} ms/GiB)  first memset {ts*1e3:7.2f} ms  second {ts2*1e3:7.2f} ms  hipFree {tf*1e3:7.2f} ms", flush=True)
# many blocks held at once: does the cost grow with what is already mapped?
held = []
for i in range(12):
    p, tm = malloc(8 * GB); held.append(p)
    print(f"held {8*(i+1):4d} GiB: hipMalloc(8 GiB) {tm*1e3:8.2f} ms", flush=True)
# beside a running kernel: a long memset chain on a stream, malloc from this thread meanwhile
stream = C.c_void_p(); hip.hipStreamCreate(C.byref(stream))
t0 = time.perf_counter()
for i in range(40): hip.hipMemsetAsync(held[i % 12], 0, C.c_size_t(8 * GB), stream)
tl = time.perf_counter() - t0
p, tm = malloc(16 * GB)
t1 = time.perf_counter(); hip.hipStreamSynchronize(stream); tw = time.perf_counter() - t1
print(f"40 async memsets of 8 GiB launched in {tl*1e3:.1f} ms; hipMalloc(16 GiB) beside them {tm*1e3:.1f} ms; then waited {tw*1e3:.1f} ms for the stream")
def bg():
    global bgt
    q, bgt = malloc(16 * GB)
for i in range(40): hip.hipMemsetAsync(held[i % 12], 0, C.c_size_t(8 * GB), stream)
th = threading.Thread(target=bg); t0 = time.perf_counter(); th.start()
hip.hipStreamSynchronize(stream); tw = time.perf_counter() - t0; th.join()
print(f"second thread: hipMalloc(16 GiB) {bgt*1e3:.1f} ms while the main thread waited {tw*1e3:.1f} ms for 40 memsets")
for p in held: free(p)
