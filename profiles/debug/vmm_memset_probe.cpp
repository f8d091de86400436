// Round 4: do hipMemsetAsync / hipMemcpyAsync (device to device, host to device, device to host) work on a range that
// spans several physical chunks mapped back to back (the library's device arena)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
  CK(hipSetDevice(0));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t GB = 1ull << 30, CH = GB, NCH = 6;
  void *basev = nullptr;
  CK(hipMemAddressReserve(&basev, 16 * GB, 0, nullptr, 0));
  char *base = (char *)basev;
  for (size_t i = 0; i < NCH; i++) {
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, CH, &prop, 0));
    CK(hipMemMap(base + i * CH, CH, 0, h, 0));
    CK(hipMemSetAccess(base + i * CH, CH, &acc, 1));
  }
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipMemsetAsync(base, 0x11, NCH * CH, s));
  CK(hipStreamSynchronize(s));
  auto peek = [&](size_t off) { unsigned char v = 0; CK(hipMemcpy(&v, base + off, 1, hipMemcpyDeviceToHost)); return (int)v; };
  printf("whole-range memset 0x11: bytes at 0, 1 GiB-1, 1 GiB, 3.5 GiB, 6 GiB-1: %02x %02x %02x %02x %02x\n", peek(0), peek(GB - 1), peek(GB), peek(3 * GB + GB / 2), peek(6 * GB - 1));
  // a memset that starts inside chunk 0 and ends inside chunk 2
  const size_t o = GB / 2, n = 2 * GB + GB / 4;
  CK(hipMemsetAsync(base + o, 0xAB, n, s));
  CK(hipStreamSynchronize(s));
  printf("memset 0xAB over [0.5, 2.75) GiB: at 0.5-1B %02x | 0.5 %02x | 1-1B %02x | 1 %02x | 1.5 %02x | 2-1B %02x | 2 %02x | 2.75-1B %02x | 2.75 %02x\n", peek(o - 1), peek(o), peek(GB - 1),
         peek(GB), peek(GB + GB / 2), peek(2 * GB - 1), peek(2 * GB), peek(o + n - 1), peek(o + n));
  // D2D copy across boundaries
  CK(hipMemcpyAsync(base + 3 * GB + 100, base + o, n, hipMemcpyDeviceToDevice, s));
  CK(hipStreamSynchronize(s));
  printf("D2D copy of that range to 3 GiB+100: first %02x, at +0.5 GiB %02x, at +1.5 GiB %02x, last %02x, one past %02x\n", peek(3 * GB + 100), peek(3 * GB + 100 + GB / 2), peek(3 * GB + 100 + GB + GB / 2),
         peek(3 * GB + 100 + n - 1), peek(3 * GB + 100 + n));
  // H2D / D2H across a boundary
  std::vector<unsigned char> h(64 << 20, 0x5C), back(64 << 20, 0);
  CK(hipMemcpyAsync(base + GB - (32 << 20), h.data(), h.size(), hipMemcpyHostToDevice, s));
  CK(hipStreamSynchronize(s));
  CK(hipMemcpyAsync(back.data(), base + GB - (32 << 20), back.size(), hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  size_t bad = 0;
  for (size_t i = 0; i < back.size(); i++) bad += back[i] != 0x5C;
  printf("64 MiB H2D + D2H across the 1 GiB boundary: %zu wrong bytes\n", bad);
  // 8-byte-pattern memset (hipMemsetD32Async) across chunks
  CK(hipMemsetD32Async((hipDeviceptr_t)(base + 4 * GB - 4096), 0x01020304, (2 * GB) / 4, s));
  CK(hipStreamSynchronize(s));
  printf("memsetD32 over [4 GiB-4 KiB, 6 GiB-4 KiB): at start %02x, 4 GiB %02x, 5 GiB %02x, end-1 %02x\n", peek(4 * GB - 4096), peek(4 * GB), peek(5 * GB), peek(6 * GB - 4096 - 1));
  return 0;
}
