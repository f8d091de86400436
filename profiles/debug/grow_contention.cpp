// Round 4: does a second thread that maps device memory (hipMemCreate + hipMemMap + hipMemSetAccess, 30 ms per GiB when the
// driver has to clear the memory first) stall the main thread's small device-to-host copies?  Which kind of copy target
// (stack variable, hipHostMalloc'ed slot, hipMemcpyAsync vs. a kernel writing into mapped pinned memory) avoids it?
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void bump(int *p) { p[0] += 1; }
__global__ void bump_to(int *p, volatile int *host_slot) { p[0] += 1; host_slot[0] = p[0]; }
static std::atomic<bool> stop{false};
static std::atomic<long long> mapped_gib{0};
static double grow_time = 0;
void grower(size_t chunk, int pause_us) {
  CK(hipSetDevice(0));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  void *base = nullptr;
  CK(hipMemAddressReserve(&base, (size_t)400 << 30, 0, nullptr, 0));
  size_t mapped = 0;
  std::vector<hipMemGenericAllocationHandle_t> hs;
  while (!stop && mapped < ((size_t)230 << 30)) {
    hipMemGenericAllocationHandle_t h;
    const double t0 = now();
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) break;
    CK(hipMemMap((char *)base + mapped, chunk, 0, h, 0));
    CK(hipMemSetAccess((char *)base + mapped, chunk, &acc, 1));
    grow_time += now() - t0;
    hs.push_back(h);
    mapped += chunk;
    mapped_gib = (long long)(mapped >> 30);
    if (pause_us) std::this_thread::sleep_for(std::chrono::microseconds(pause_us));
  }
  for (size_t i = hs.size(); i-- > 0;) { (void)hipMemUnmap((char *)base + i * chunk, chunk); (void)hipMemRelease(hs[i]); }
  (void)hipMemAddressFree(base, (size_t)400 << 30);
}
static void report(const char *what, std::vector<double> &v) {
  std::sort(v.begin(), v.end());
  double sum = 0; for (double x : v) sum += x;
  printf("  %-46s n %5zu  median %8.1f us  p99 %9.1f us  max %9.1f us  total %7.1f ms\n", what, v.size(), v[v.size() / 2] * 1e6, v[v.size() * 99 / 100] * 1e6, v.back() * 1e6, sum * 1e3);
}
int main(int argc, char **argv) {
  const size_t chunk = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20;
  const int pause_us = argc > 2 ? atoi(argv[2]) : 0;
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int *d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
  int *pin; CK(hipHostMalloc((void **)&pin, 64, hipHostMallocDefault));
  int *pin_dev = nullptr; CK(hipHostGetDevicePointer((void **)&pin_dev, pin, 0));
  // dirty the memory first: allocate and free 200 GiB so that later mappings pay the driver's clearing
  { std::vector<void *> ps; for (int i = 0; i < 25; i++) { void *p; if (hipMalloc(&p, (size_t)8 << 30) != hipSuccess) break; CK(hipMemset(p, 1, (size_t)8 << 30)); ps.push_back(p); } CK(hipDeviceSynchronize()); for (void *p : ps) CK(hipFree(p)); }
  for (int phase = 0; phase < 2; phase++) {
    std::thread th;
    if (phase == 1) { stop = false; th = std::thread(grower, chunk, pause_us); std::this_thread::sleep_for(std::chrono::milliseconds(50)); }
    printf("%s\n", phase == 0 ? "no grower thread:" : "grower thread mapping memory meanwhile:");
    const int N = 300;
    std::vector<double> a, b, c, e;
    for (int i = 0; i < N; i++) {  // (1) D2H into a stack variable
      int v = 0; bump<<<1, 1, 0, s>>>(d);
      const double t0 = now(); CK(hipMemcpyAsync(&v, d, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); a.push_back(now() - t0);
    }
    for (int i = 0; i < N; i++) {  // (2) D2H into hipHostMalloc'ed memory
      bump<<<1, 1, 0, s>>>(d);
      const double t0 = now(); CK(hipMemcpyAsync(pin, d, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); b.push_back(now() - t0);
    }
    for (int i = 0; i < N; i++) {  // (3) the kernel writes the value into mapped pinned memory, the host only synchronises
      const double t0 = now(); bump_to<<<1, 1, 0, s>>>(d, pin_dev); CK(hipStreamSynchronize(s)); c.push_back(now() - t0);
    }
    for (int i = 0; i < N; i++) {  // (4) H2D of 4 bytes from the stack
      int v = i; const double t0 = now(); CK(hipMemcpyAsync(d, &v, 4, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); e.push_back(now() - t0);
    }
    report("kernel + hipMemcpyAsync D2H -> stack variable", a);
    report("kernel + hipMemcpyAsync D2H -> hipHostMalloc", b);
    report("kernel writes mapped pinned slot + sync", c);
    report("hipMemcpyAsync H2D <- stack variable", e);
    if (phase == 1) { stop = true; th.join(); printf("  grower: %lld GiB mapped, %.2f s inside create/map/access, chunk %zu MiB, pause %d us\n", (long long)mapped_gib, grow_time, chunk >> 20, pause_us); }
  }
  return 0;
}
