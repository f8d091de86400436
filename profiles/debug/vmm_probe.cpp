// Probe of the HIP virtual-memory API on the box (round 4: one growable arena instead of hipMalloc/hipFree churn):
// reserve a large address range, create physical chunks, map them back to back, touch them from a kernel, time it.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(double *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0; }
int main() {
  CK(hipSetDevice(0));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity %zu\n", gran);
  const size_t GB = 1ull << 30, VA = 512 * GB, CH = 2 * GB;
  void *base = nullptr;
  double t0 = now();
  CK(hipMemAddressReserve(&base, VA, 0, nullptr, 0));
  printf("reserve %zu GiB of address space: %.2f ms, base %p\n", VA / GB, (now() - t0) * 1e3, base);
  std::vector<hipMemGenericAllocationHandle_t> hs;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  size_t mapped = 0;
  for (int i = 0; i < 100; i++) {
    hipMemGenericAllocationHandle_t h;
    t0 = now();
    hipError_t e = hipMemCreate(&h, CH, &prop, 0);
    if (e != hipSuccess) { printf("hipMemCreate failed at %zu GiB: %s\n", mapped / GB, hipGetErrorString(e)); (void)hipGetLastError(); break; }
    const double tc = now() - t0;
    t0 = now();
    CK(hipMemMap((char *)base + mapped, CH, 0, h, 0));
    const double tm = now() - t0;
    t0 = now();
    CK(hipMemSetAccess((char *)base + mapped, CH, &acc, 1));
    const double ta = now() - t0;
    hs.push_back(h);
    mapped += CH;
    if (i < 4 || i % 8 == 7 || tc + tm + ta > 0.02) printf("chunk %3d (%4zu GiB mapped): create %.2f ms  map %.2f ms  access %.2f ms\n", i, mapped / GB, tc * 1e3, tm * 1e3, ta * 1e3);
  }
  // one kernel over the whole contiguous range
  t0 = now();
  touch<<<4096, 256>>>((double *)base, mapped / 8);
  CK(hipDeviceSynchronize());
  printf("kernel over %zu GiB contiguous: %.2f ms\n", mapped / GB, (now() - t0) * 1e3);
  t0 = now();
  touch<<<4096, 256>>>((double *)base, mapped / 8);
  CK(hipDeviceSynchronize());
  printf("again: %.2f ms\n", (now() - t0) * 1e3);
  // pointer attributes + a plain copy out of the range (what the library's hipMemcpyAsync paths need)
  hipPointerAttribute_t at;
  hipError_t e = hipPointerGetAttributes(&at, (char *)base + 3 * GB + 8);
  printf("hipPointerGetAttributes: %s type %d device %d\n", hipGetErrorString(e), (int)at.type, at.device);
  double hv[4] = {0, 0, 0, 0};
  CK(hipMemcpy(hv, (char *)base + CH - 16, 32, hipMemcpyDeviceToHost));  // straddles two chunks
  printf("copy across a chunk boundary: %g %g %g %g\n", hv[0], hv[1], hv[2], hv[3]);
  // shrink from the top: unmap + release half, then map again
  t0 = now();
  size_t keep = hs.size() / 2;
  for (size_t i = hs.size(); i-- > keep;) {
    CK(hipMemUnmap((char *)base + i * CH, CH));
    CK(hipMemRelease(hs[i]));
  }
  printf("unmap + release %zu chunks: %.2f ms\n", hs.size() - keep, (now() - t0) * 1e3);
  mapped = keep * CH;
  hs.resize(keep);
  for (int i = 0; i < 8; i++) {
    hipMemGenericAllocationHandle_t h;
    t0 = now();
    CK(hipMemCreate(&h, CH, &prop, 0));
    CK(hipMemMap((char *)base + mapped, CH, 0, h, 0));
    CK(hipMemSetAccess((char *)base + mapped, CH, &acc, 1));
    printf("re-grow chunk %d: %.2f ms\n", i, (now() - t0) * 1e3);
    hs.push_back(h);
    mapped += CH;
  }
  for (size_t i = hs.size(); i-- > 0;) { CK(hipMemUnmap((char *)base + i * CH, CH)); CK(hipMemRelease(hs[i])); }
  CK(hipMemAddressFree(base, VA));
  printf("done\n");
  return 0;
}
