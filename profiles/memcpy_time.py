# pageable vs pinned host copies of 1 GiB (what the setup moves per operator for the greedy tile schedule)
import ctypes, time
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipSetDevice(0)
n = 1 << 30
d = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(n)); hip.hipMemset(d, 1, ctypes.c_size_t(n)); hip.hipDeviceSynchronize()
buf = (ctypes.c_char * n)()
ctypes.memset(buf, 0, n)  # touch
for name, kind in (("D2H pageable", 2), ("H2D pageable", 1)):
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        if kind == 2: hip.hipMemcpy(buf, d, ctypes.c_size_t(n), 2)
        else: hip.hipMemcpy(d, buf, ctypes.c_size_t(n), 1)
        ts.append(time.perf_counter() - t0)
    print(name, "%.0f ms  %.1f GB/s" % (1e3 * min(ts), n / min(ts) / 1e9))
t0 = time.perf_counter()
h = ctypes.c_void_p(); hip.hipHostMalloc(ctypes.byref(h), ctypes.c_size_t(n), 0)
print("hipHostMalloc 1 GiB: %.0f ms" % (1e3 * (time.perf_counter() - t0)))
for name, kind in (("D2H pinned", 2), ("H2D pinned", 1)):
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        if kind == 2: hip.hipMemcpy(h, d, ctypes.c_size_t(n), 2)
        else: hip.hipMemcpy(d, h, ctypes.c_size_t(n), 1)
        ts.append(time.perf_counter() - t0)
    print(name, "%.0f ms  %.1f GB/s" % (1e3 * min(ts), n / min(ts) / 1e9))
