import ctypes, time
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
p = ctypes.c_void_p()
hip.hipSetDevice(0)
hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20)); hip.hipFree(p)
for gb in (0.25, 1, 4, 8, 16):
    n = int(gb * (1 << 30))
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n))
        t1 = time.perf_counter()
        hip.hipMemset(p, 0, ctypes.c_size_t(n)); hip.hipDeviceSynchronize()
        t2 = time.perf_counter()
        hip.hipFree(p)
        t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    print(gb, "GB: malloc %.1f ms  memset %.1f ms  free %.1f ms" % tuple(1e3 * min(x[i] for x in ts) for i in range(3)), "first malloc %.1f ms" % (1e3 * ts[0][0]))
