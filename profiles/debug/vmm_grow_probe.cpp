// Round 4: blocks laid out back to back in a growing VMM range (chunks mapped on demand, as the library's arena does),
// each filled with hipMemsetAsync right after the chunk under its tail was mapped; all verified at the end.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void fillk(unsigned char *p, size_t n, unsigned char v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
int main(int argc, char **argv) {
  const int use_kernel = argc > 1 ? atoi(argv[1]) : 0;
  CK(hipSetDevice(0));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  const size_t GB = 1ull << 30, CH = GB;
  void *basev = nullptr;
  CK(hipMemAddressReserve(&basev, 128 * GB, 0, nullptr, 0));
  unsigned char *base = (unsigned char *)basev;
  size_t mapped = 0;
  auto grow_to = [&](size_t end) {
    while (mapped < end) {
      hipMemGenericAllocationHandle_t h;
      CK(hipMemCreate(&h, CH, &prop, 0));
      CK(hipMemMap(base + mapped, CH, 0, h, 0));
      CK(hipMemSetAccess(base + mapped, CH, &acc, 1));
      mapped += CH;
    }
  };
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  struct B { size_t off, n; unsigned char tag; };
  std::vector<B> bs;
  unsigned long long x = 12345;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  size_t off = 0;
  for (int i = 0; i < 120 && off < 60 * GB; i++) {
    const int bits = 10 + (int)(rnd() % 21);
    size_t n = (size_t)(rnd() & ((1ull << bits) - 1)) + 1;
    n = (n + 255) / 256 * 256;
    grow_to(off + n);
    B b{off, n, (unsigned char)(1 + rnd() % 250)};
    if (use_kernel) fillk<<<1024, 256, 0, s>>>(base + off, n, b.tag);
    else CK(hipMemsetAsync(base + off, b.tag, n, s));
    bs.push_back(b);
    off += n;
  }
  CK(hipStreamSynchronize(s));
  printf("granularity %zu, %zu blocks, %.1f GiB, %zu chunks; fill by %s\n", gran, bs.size(), off / 1073741824.0, mapped / CH, use_kernel ? "own kernel" : "hipMemsetAsync");
  int bad = 0;
  for (const B &b : bs) {
    // probe every chunk boundary inside the block, plus head / tail
    std::vector<size_t> at = {0, b.n - 1, b.n / 2};
    for (size_t c = (b.off / CH + 1) * CH; c < b.off + b.n; c += CH) { at.push_back(c - b.off - 1); at.push_back(c - b.off); at.push_back(c - b.off + (4 << 20)); }
    for (size_t o : at) {
      if (o >= b.n) continue;
      unsigned char v = 0;
      CK(hipMemcpy(&v, base + b.off + o, 1, hipMemcpyDeviceToHost));
      if (v != b.tag) { if (bad < 12) printf("  block at %.3f GiB (%zu bytes, tag %d): byte %zu holds %d (offset in chunk %zu)\n", b.off / 1073741824.0, b.n, b.tag, o, v, (b.off + o) % CH); bad++; }
    }
  }
  printf("%d wrong probes\n", bad);
  return 0;
}
