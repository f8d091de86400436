// Runtime context: device/stream ownership, scratch, device CSR upload,
// event timers, the host thread helper.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>

#include <map>
#include <condition_variable>
#include <mutex>
#include <set>
#include <unordered_map>

#include "kernels.hpp"
#include "setup_kernels.hpp"
#include "mi_internal.hpp"
#include "parcsr.hpp"
#include "profile.hpp"

namespace mi {

static Ctx g_ctx;
Ctx &ctx() { return g_ctx; }

Comm &current_comm() {
  if (!g_ctx.comm) g_ctx.comm = make_self_comm();
  return *g_ctx.comm;
}

double wall_time() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// CPUs this process may use: the cgroup quota (v2 cpu.max, v1 cfs quota) when there is one, else what the OS shows
static int usable_cpus() {
  int n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  long long quota = -1, period = -1;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64] = {0};
    if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
    fclose(f);
  } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
    if (fscanf(g, "%lld", &quota) != 1) quota = -1;
    fclose(g);
    if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
      if (fscanf(h, "%lld", &period) != 1) period = -1;
      fclose(h);
    }
  }
  if (quota > 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
  return n;
}

// MI_HYPRE_HOST_THREADS, or this rank's share of the usable CPUs (LOCAL_WORLD_SIZE ranks per node under torchrun),
// between 2 and 32: the host phases of the (distributed) setup are memory-bound loops that stop scaling there
int host_threads() {
  static int n = 0;
  if (n == 0) {
    const char *e = getenv("MI_HYPRE_HOST_THREADS");
    if (e) {
      n = std::max(1, std::min(64, atoi(e)));
    } else {
      const char *lw = getenv("LOCAL_WORLD_SIZE");
      const int ranks_here = std::max(1, lw ? atoi(lw) : 1);
      n = std::max(2, std::min(32, usable_cpus() / ranks_here));
    }
  }
  return n;
}

void parallel_for(int64_t n, const std::function<void(int64_t, int64_t, int)> &fn, int max_threads) {
  int nt = host_threads();
  if (max_threads > 0 && nt > max_threads) nt = max_threads;
  if (n < 4096 || nt == 1) {
    fn(0, n, 0);
    return;
  }
  if (nt > n) nt = (int)n;
  std::vector<std::thread> th;
  std::vector<std::exception_ptr> errs((size_t)nt);
  for (int t = 0; t < nt; t++) {
    const int64_t b = n * t / nt, e = n * (t + 1) / nt;
    th.emplace_back([&, b, e, t]() {
      try {
        fn(b, e, t);
      } catch (...) {
        errs[(size_t)t] = std::current_exception();
      }
    });
  }
  for (auto &x : th) x.join();
  for (auto &e : errs)
    if (e) std::rethrow_exception(e);
}

void d2h(void *dst, const void *src, size_t bytes, hipStream_t s) {
  if (!bytes) return;
  if (s == nullptr) {
    // a copy on the null stream does not wait for the library's (non-blocking) streams: whoever downloads that way
    // (DVec::to_host, LazyInts::host) means "what the kernels launched so far have produced"
    Ctx &c = g_ctx;
    if (c.inited) {
      MI_HIP(hipStreamSynchronize(c.stream));
      MI_HIP(hipStreamSynchronize(c.comm_stream));
    }
  }
  static std::mutex m;
  static char *stage = nullptr;
  constexpr size_t STAGE = (size_t)4 << 20;
  if (bytes > ((size_t)64 << 20)) {  // large: straight into the caller's memory
    MI_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    return;
  }
  std::lock_guard<std::mutex> g(m);
  if (!stage) MI_HIP(hipHostMalloc((void **)&stage, STAGE, hipHostMallocDefault));
  for (size_t off = 0; off < bytes; off += STAGE) {
    const size_t len = std::min(STAGE, bytes - off);
    MI_HIP(hipMemcpyAsync(stage, (const char *)src + off, len, hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    memcpy((char *)dst + off, stage, len);
  }
}

void zero_on_stream(void *p, size_t bytes) {
  if (!p || !bytes) return;
  ensure_init();
  MI_HIP(hipMemsetAsync(p, 0, bytes, ctx().stream));
}

// The product path has no CPU fallback: without a HIP device every entry point
// fails loudly here.
void ensure_init() {
  Ctx &c = g_ctx;
  if (c.inited) return;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    fail(1, "mi_hypre: no HIP device available (this library has no CPU path; it needs an MI355X/gfx950 GPU)");
  MI_HIP(hipGetDevice(&c.device));
  // MI_HYPRE_SPIN_WAIT=1: host waits (hipStreamSynchronize & co.) spin instead of sleeping on the interrupt.  A setup makes
  // ~400 short host <-> device round trips (counts, totals, the tile schedules); on a host whose wake-ups are slow each
  // of them costs a millisecond instead of 20 us (profiles/r04_setup_split_512.txt: 0.65 s of a 2.83 s setup idle on one
  // box, 0.18 s of 2.66 s on another).  Process-wide and therefore opt-in; refused silently where the runtime says no.
  if (getenv("MI_HYPRE_SPIN_WAIT") && atoi(getenv("MI_HYPRE_SPIN_WAIT")) != 0) {
    if (hipSetDeviceFlags(hipDeviceScheduleSpin) != hipSuccess) (void)hipGetLastError();
  }
  MI_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  MI_HIP(hipStreamCreateWithFlags(&c.comm_stream, hipStreamNonBlocking));
  MI_HIP(hipEventCreateWithFlags(&c.ev_packed, hipEventDisableTiming));
  MI_HIP(hipEventCreateWithFlags(&c.ev_halo, hipEventDisableTiming));
  c.red_partials.alloc((size_t)k::RED_MAX_BLOCKS * k::MASS_NV);
  c.red_out.alloc(256);
  c.red_ticket.alloc(4);
  MI_HIP(hipMemset(c.red_ticket.p, 0, 4 * sizeof(unsigned)));
  // [0, 256): result slots the host reads; [256, 512): posted by kernels and polled (h_post_flag = the last word)
  MI_HIP(hipHostMalloc((void **)&c.h_pinned, 512 * sizeof(double), hipHostMallocCoherent | hipHostMallocMapped));  // (coherent: a kernel's system-scope stores are visible to the polling host while the kernel runs)
  memset(c.h_pinned, 0, 512 * sizeof(double));
  c.h_post_flag = reinterpret_cast<unsigned long long *>(c.h_pinned + 511);
  c.post_seq = 0;
  if (!c.comm) c.comm = make_self_comm();
  // both translation units' device code now (the runtime would load each at its first launch, i.e. inside the first
  // Assemble / Setup: 0.2-0.4 s of a 2.3 s setup in the first process on a machine, whose disk cache is cold)
  k::load_device_code(c.stream);
  sk::load_device_code(c.stream);
  MI_HIP(hipStreamSynchronize(c.stream));
  const char *ch = getenv("MI_HYPRE_GS_CHUNK");
  if (ch) c.gs_chunk = atoi(ch);
  if (c.gs_chunk < 1) c.gs_chunk = 1;
  if (c.gs_chunk > k::GS_MAX_CHUNK) c.gs_chunk = k::GS_MAX_CHUNK;
  const char *vb = getenv("MI_HYPRE_VERBOSE");
  if (vb) c.verbose = atoi(vb);
  c.inited = true;
}

namespace {
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char *e = getenv("MI_HYPRE_ROCTX");
    if (e && atoi(e) == 0) return;
    void *h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) push = nullptr, pop = nullptr;
  }
};
Roctx &roctx() {
  static Roctx r;
  return r;
}
}  // namespace

TraceRange::TraceRange(const char *name) {
  Roctx &r = roctx();
  if (r.push) {
    r.push(name);
    on = true;
  }
}
TraceRange::~TraceRange() {
  if (on) roctx().pop();
}

// ---------------------------------------------------------------- device memory behind every DVec (mi_internal.hpp)
// Round 4: ONE GROWABLE ARENA in a reserved address range (HIP virtual-memory API: hipMemAddressReserve, physical
// chunks of 1 GiB created and mapped back to back at the top as the heap grows), with an ordinary best-fit free list
// and coalescing of neighbouring free ranges inside it.  Why: a rocprofv3 trace of the 512^3 setup
// (profiles/r04_setup_split_512_before.txt) showed the device busy 4.8 s of 16.4 s; 11 s of the rest sat in front
// of the first kernel that touches a FRESH allocation, at a constant 30 ms per GiB -- memory that goes back to the
// driver (hipFree, or the size-class cache of round 3 trimming itself at its cap) is cleared by the kernel driver before
// it is handed out again, at ~35 GB/s, inside hipMalloc (profiles/debug/malloc_cost.py, vmm_probe.cpp: the first
// 64-130 GiB of a fresh box are clean and cost nothing, everything after that 30.2 ms per GiB).  The setup allocates
// and frees ~270 GB in hundreds of blocks of every size, so a cache that serves only requests of a similar size
// (round 3) misses most of them.  The arena never gives memory back while it works: every byte is mapped (and, if
// dirty, cleared) at most once, whatever the sequence of sizes; dev_pool_trim() -- end of Setup, HYPRE_Finalize --
// unmaps whole chunks from the top that are free, keeping a quarter of what is in use (the Krylov basis is
// allocated next).  MI_HYPRE_POOL: 2 (default) arena, falling back to 1 when the virtual-memory API is not
// available; 1 the size-class cache of round 3 (MI_HYPRE_POOL_MAX_GB); 0 plain hipMalloc / hipFree.
// A block is handed to a new owner without synchronisation unless part of it was released since the library's
// streams were last drained (release epochs), in which case they are drained first: what hipFree's implicit
// synchronisation gave, paid only when it is needed.
namespace {
struct DevPool {
  std::mutex m;
  std::multimap<size_t, void *> free_blocks;     // size -> block
  std::unordered_map<void *, size_t> size_of;    // every block the pool handed out or holds
  size_t cached = 0;
  long long hits = 0, misses = 0;
  int enabled = -1;
  size_t max_cached = 0;
  void init() {
    if (enabled >= 0) return;
    const char *e = getenv("MI_HYPRE_POOL");
    enabled = e ? atoi(e) : 2;
    if (enabled < 0 || enabled > 2) enabled = 2;
    const double gb = getenv("MI_HYPRE_POOL_MAX_GB") ? atof(getenv("MI_HYPRE_POOL_MAX_GB")) : 48.0;
    max_cached = (size_t)(gb * 1e9);
  }
};
DevPool &pool() {
  static DevPool *p = new DevPool();  // never destroyed: DVecs of static objects may be released at exit
  return *p;
}
size_t pool_round(size_t bytes) {
  if (bytes < 256) return 256;
  if (bytes <= (1u << 20)) return (bytes + 4095) / 4096 * 4096;
  return (bytes + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);
}

void drain_library_streams() {
  Ctx &c = ctx();
  if (c.inited) {
    MI_HIP(hipStreamSynchronize(c.stream));
    MI_HIP(hipStreamSynchronize(c.comm_stream));
  } else {
    MI_HIP(hipDeviceSynchronize());
  }
}

struct DevArena {
  std::mutex m;
  int state = 0;  // 0 untried, 1 working, -1 unavailable
  char *base = nullptr;
  // `top`: where the next chunk is mapped.  It only ever rises: an address range that was unmapped (trim) is NEVER mapped
  // again -- on this stack (ROCm 7.2, gfx950) kernels that write a range which was unmapped and mapped onto new physical
  // memory lose part of their stores (stale translations: profiles/debug/vmm_remap_probe.cpp, 9 of 40 fill / verify
  // cycles incomplete, with hipMemsetAsync and with a plain kernel, with and without a device synchronisation after the
  // unmap; profiles/debug/vmm_burn_probe.cpp: 0 of 59 when the addresses are not reused).  Address space is cheap (8 TiB
  // reserved), so trimmed ranges are simply abandoned.  `mapped` = bytes mapped now.
  size_t va_size = 0, chunk = 0, top = 0, mapped = 0;
  int device = 0;
  struct Chunk {
    size_t off;
    hipMemGenericAllocationHandle_t h;
  };
  std::vector<Chunk> handles;  // mapped chunks, ascending offsets
  struct Free {
    size_t size;
    unsigned long long epoch;  // newest release that went into this range
  };
  std::map<size_t, Free> free_by_off;                // offset -> range
  std::set<std::pair<size_t, size_t>> free_by_size;  // (size, offset)
  std::unordered_map<size_t, size_t> live;           // offset -> size
  unsigned long long release_epoch = 0, drained_epoch = 0;
  size_t in_use = 0, peak_in_use = 0, peak_mapped = 0;
  long long served = 0, grown = 0, drains = 0;
  double t_grow = 0.0, t_drain = 0.0;  // seconds inside hipMemCreate / Map / SetAccess, and waiting for the streams
  double t_wait = 0.0;                 // seconds an allocation waited for the grow-ahead thread
  // Grow-ahead thread: where the driver has to clear the memory first (30 ms per GiB, see above) a chunk is worth
  // having BEFORE a request needs it.  The thread keeps `headroom()` bytes of mapped, free space in the arena; a request
  // that does not fit wakes it and waits for chunks (the only writer of `mapped` is then the thread).  MI_HYPRE_ARENA_AHEAD=0:
  // no thread, growth inside the request.
  std::thread grower;
  std::condition_variable cv_work, cv_done;
  bool grower_on = false, grower_stop = false, grower_busy = false, oom = false;
  size_t demand = 0;  // bytes a waiting request still needs mapped at the top
  size_t target = 0;  // dev_arena_hint(): map up to here in the background whatever the headroom
  size_t top_free() const {
    if (free_by_off.empty()) return 0;
    auto last = std::prev(free_by_off.end());
    return last->first + last->second.size == top ? last->second.size : 0;
  }
  size_t headroom() const {
    static const double gb = getenv("MI_HYPRE_ARENA_AHEAD_GB") ? atof(getenv("MI_HYPRE_ARENA_AHEAD_GB")) : 16.0;
    const size_t cap = (size_t)(gb * 1073741824.0);
    return std::min(cap, std::max<size_t>(chunk, in_use / 4));
  }

  bool init() {
    if (state) return state > 0;
    state = -1;
    if (hipGetDevice(&device) != hipSuccess) return false;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) {
      (void)hipGetLastError();
      return false;
    }
    size_t want_chunk = (size_t)(getenv("MI_HYPRE_ARENA_CHUNK_MB") ? atoll(getenv("MI_HYPRE_ARENA_CHUNK_MB")) : 1024) << 20;
    if (want_chunk < gran) want_chunk = gran;
    chunk = (want_chunk + gran - 1) / gran * gran;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)288 << 30;
    // address space is free, and trimmed ranges are abandoned, not reused: 8 TiB (then 2 TiB, then 4x the device's memory)
    void *b = nullptr;
    for (size_t want_va : {(size_t)8 << 40, (size_t)2 << 40, 4 * total_b}) {
      va_size = ((want_va + chunk - 1) / chunk) * chunk;
      if (hipMemAddressReserve(&b, va_size, 0, nullptr, 0) == hipSuccess && b) break;
      (void)hipGetLastError();
      b = nullptr;
    }
    if (!b) return false;
    base = (char *)b;
    state = 1;
    if (!(getenv("MI_HYPRE_ARENA_AHEAD") && atoi(getenv("MI_HYPRE_ARENA_AHEAD")) == 0)) {
      grower_on = true;
      grower = std::thread([this] { grow_ahead_loop(); });
      grower.detach();  // lives as long as the process (the arena is never destroyed)
    }
    return true;
  }
  // one chunk, created and mapped WITHOUT the lock held (this is where the driver may take 30 ms per GiB)
  bool map_one(size_t at) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipMemGenericAllocationHandle_t h;
    if (at + chunk > va_size) return false;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    if (hipMemMap(base + at, chunk, 0, h, 0) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipMemRelease(h);
      return false;
    }
    if (hipMemSetAccess(base + at, chunk, &acc, 1) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipMemUnmap(base + at, chunk);
      (void)hipMemRelease(h);
      return false;
    }
    pending_handle = h;
    return true;
  }
  hipMemGenericAllocationHandle_t pending_handle{};
  void grow_ahead_loop() {
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv_work.wait(lk, [this] { return grower_stop || (!paused && !oom && (demand > 0 || mapped < target || (in_use > 0 && mapped - in_use < headroom()))); });
      if (grower_stop) return;
      grower_busy = true;
      const size_t at = top;
      lk.unlock();
      const double t0 = wall_time();
      const bool ok = map_one(at);
      const double dt = wall_time() - t0;
      lk.lock();
      grower_busy = false;
      t_grow += dt;
      if (ok) {
        handles.push_back(Chunk{at, pending_handle});
        top += chunk;
        mapped += chunk;
        put_free(at, chunk, 0);
        peak_mapped = std::max(peak_mapped, mapped);
        grown++;
        demand = demand > chunk ? demand - chunk : 0;
      } else {
        oom = true;  // a request that needs more fails; cleared when memory is released or trimmed
        demand = 0;
      }
      cv_done.notify_all();
    }
  }
  bool paused = false;
  void put_free(size_t off, size_t size, unsigned long long epoch) {
    // coalesce with the neighbours
    auto next = free_by_off.lower_bound(off);
    if (next != free_by_off.begin()) {
      auto prev = std::prev(next);
      if (prev->first + prev->second.size == off) {
        off = prev->first;
        size += prev->second.size;
        epoch = std::max(epoch, prev->second.epoch);
        free_by_size.erase({prev->second.size, prev->first});
        free_by_off.erase(prev);
      }
    }
    if (next != free_by_off.end() && off + size == next->first) {
      size += next->second.size;
      epoch = std::max(epoch, next->second.epoch);
      free_by_size.erase({next->second.size, next->first});
      free_by_off.erase(next);
    }
    free_by_off[off] = Free{size, epoch};
    free_by_size.insert({size, off});
  }
  // map `n` more chunks at the top; false when the device has no memory left for them
  bool grow(size_t n) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t old_top = top;
    const double tg0 = wall_time();
    for (size_t q = 0; q < n; q++) {
      if (top + chunk > va_size) break;
      hipMemGenericAllocationHandle_t h;
      if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) {
        (void)hipGetLastError();
        break;
      }
      if (hipMemMap(base + top, chunk, 0, h, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipMemRelease(h);
        break;
      }
      if (hipMemSetAccess(base + top, chunk, &acc, 1) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipMemUnmap(base + top, chunk);
        (void)hipMemRelease(h);
        break;
      }
      handles.push_back(Chunk{top, h});
      top += chunk;
      mapped += chunk;
    }
    if (top > old_top) put_free(old_top, top - old_top, 0);
    peak_mapped = std::max(peak_mapped, mapped);
    grown++;
    t_grow += wall_time() - tg0;
    return top - old_top == n * chunk;
  }
  void *alloc(size_t bytes, std::unique_lock<std::mutex> &lk) {
    size_t want = bytes < 256 ? 256 : (bytes + 255) / 256 * 256;
    if (want >= ((size_t)1 << 16)) want = (want + 4095) / 4096 * 4096;
    auto it = free_by_size.lower_bound({want, 0});
    if (it == free_by_size.end()) {
      if (grower_on) {
        // the free range at the top (if any) counts towards the request; the thread maps the rest chunk by chunk
        const double tw0 = wall_time();
        while ((it = free_by_size.lower_bound({want, 0})) == free_by_size.end()) {
          if (oom) return nullptr;
          const size_t tf = top_free();
          demand = std::max(demand, want - std::min(want, tf));
          cv_work.notify_one();
          cv_done.wait(lk);
        }
        t_wait += wall_time() - tw0;
      } else {
        const size_t need = want - std::min(want, top_free());
        if (!grow((need + chunk - 1) / chunk)) return nullptr;
        it = free_by_size.lower_bound({want, 0});
        if (it == free_by_size.end()) return nullptr;
      }
    } else {
      served++;
    }
    const size_t off = it->second, size = it->first;
    const Free f = free_by_off[off];
    free_by_size.erase(it);
    free_by_off.erase(off);
    if (size > want) {
      free_by_off[off + want] = Free{size - want, f.epoch};
      free_by_size.insert({size - want, off + want});
    }
    live[off] = want;
    in_use += want;
    peak_in_use = std::max(peak_in_use, in_use);
    if (f.epoch > drained_epoch) {
      // part of this range was released after the streams were last drained: its previous owner's kernels may
      // still be running
      const unsigned long long e = release_epoch;
      const double td0 = wall_time();
      drain_library_streams();
      t_drain += wall_time() - td0;
      drains++;
      drained_epoch = e;
    }
    if (grower_on && !oom && mapped - in_use < headroom()) cv_work.notify_one();  // (free bytes anywhere in the arena, not only at the top)
    return base + off;
  }
  bool owns(const void *p) const { return state > 0 && (const char *)p >= base && (const char *)p < base + va_size; }
  void release(void *p) {
    const size_t off = (size_t)((char *)p - base);
    auto it = live.find(off);
    if (it == live.end()) return;  // (double release: ignored, as hipFree would report and the old pool ignored)
    const size_t size = it->second;
    live.erase(it);
    in_use -= size;
    put_free(off, size, ++release_epoch);
    oom = false;
  }
  // unmap whole free chunks from the top, keeping `keep` bytes of free space there
  void trim(size_t keep, std::unique_lock<std::mutex> &lk) {
    if (state <= 0) return;
    // the grow-ahead thread must not be mapping at the top meanwhile
    paused = true;
    while (grower_busy) cv_done.wait(lk);
    struct Resume {
      DevArena &a;
      ~Resume() { a.paused = false; a.oom = false; }
    } resume{*this};
    target = 0;
    if (free_by_off.empty()) return;
    auto last = std::prev(free_by_off.end());
    if (last->first + last->second.size != top) return;  // (what is free at the top; after a trim the top is "burned" and nothing is)
    const size_t start = last->first + std::min(keep, last->second.size);
    const size_t new_end = (start + chunk - 1) / chunk * chunk;
    if (new_end >= top) return;
    drain_library_streams();
    drained_epoch = release_epoch;
    const Free f = last->second;
    const size_t off = last->first;
    free_by_size.erase({f.size, off});
    free_by_off.erase(last);
    // whole chunks in [new_end, top) go back to the driver; their ADDRESSES are abandoned (`top` stays where it is)
    while (!handles.empty() && handles.back().off >= new_end) {
      (void)hipMemUnmap(base + handles.back().off, chunk);
      (void)hipMemRelease(handles.back().h);
      handles.pop_back();
      mapped -= chunk;
    }
    if (new_end > off) {
      free_by_off[off] = Free{new_end - off, f.epoch};
      free_by_size.insert({new_end - off, off});
    }
  }
};
DevArena &arena() {
  static DevArena *a = new DevArena();  // never destroyed (see pool())
  return *a;
}
}  // namespace

void dev_pool_trim() {
  DevPool &P = pool();
  P.init();
  if (P.enabled == 2) {
    DevArena &A = arena();
    std::unique_lock<std::mutex> g(A.m);
    static const double keep_gb = getenv("MI_HYPRE_ARENA_KEEP_GB") ? atof(getenv("MI_HYPRE_ARENA_KEEP_GB")) : -1.0;
    const size_t keep = keep_gb >= 0.0 ? (size_t)(keep_gb * 1073741824.0) : std::max<size_t>((size_t)2 << 30, A.in_use / 4);
    A.trim(A.in_use == 0 ? 0 : keep, g);
  }
  std::vector<void *> blocks;
  {
    std::lock_guard<std::mutex> g(P.m);
    for (auto &kv : P.free_blocks) {
      blocks.push_back(kv.second);
      P.size_of.erase(kv.second);
    }
    P.free_blocks.clear();
    P.cached = 0;
  }
  for (void *b : blocks) (void)hipFree(b);
}

void dev_pool_stats(long long *cached_bytes, long long *hits, long long *misses) {
  DevPool &P = pool();
  P.init();
  if (P.enabled == 2 && arena().state > 0) {
    DevArena &A = arena();
    std::lock_guard<std::mutex> g(A.m);
    if (cached_bytes) *cached_bytes = (long long)(A.mapped - A.in_use);
    if (hits) *hits = A.served;
    if (misses) *misses = A.grown;
    return;
  }
  std::lock_guard<std::mutex> g(P.m);
  if (cached_bytes) *cached_bytes = (long long)P.cached;
  if (hits) *hits = P.hits;
  if (misses) *misses = P.misses;
}

// Expected growth, from whoever knows it (IJMatrixAssemble and BoomerAMGSetup: a multiple of the operator's bytes): the
// grow-ahead thread maps up to in_use + bytes in the background, capped at half of what the device has free now.
// HYPRE_SetUmpireDevicePoolSize (the reference's `umpire_device_pool_mbs`, /root/reference/src/main.cpp:107-114): the
// initial size of the device pool = that many bytes of the arena mapped in the background from now on
void dev_arena_reserve(size_t bytes_total) {
  DevPool &P = pool();
  P.init();
  if (P.enabled != 2 || bytes_total == 0) return;
  DevArena &A = arena();
  std::unique_lock<std::mutex> g(A.m);
  if (!A.init() || !A.grower_on) return;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  A.target = std::max(A.target, std::min(bytes_total, A.mapped + free_b / 2));
  A.cv_work.notify_one();
}

void dev_arena_hint(size_t bytes_more) {
  DevPool &P = pool();
  P.init();
  if (P.enabled != 2) return;
  DevArena &A = arena();
  std::unique_lock<std::mutex> g(A.m);
  if (!A.init() || !A.grower_on) return;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return;
  }
  const size_t want = A.in_use + bytes_more;
  const size_t cap = A.mapped + free_b / 2;
  A.target = std::max(A.target, std::min(want, cap));
  A.cv_work.notify_one();
}

void dev_arena_times(double *t_grow, double *t_drain, long long *grown, long long *drains, double *t_wait) {
  DevArena &A = arena();
  std::lock_guard<std::mutex> g(A.m);
  if (t_wait) *t_wait = A.t_wait;
  if (t_grow) *t_grow = A.t_grow;
  if (t_drain) *t_drain = A.t_drain;
  if (grown) *grown = A.grown;
  if (drains) *drains = A.drains;
}

void dev_arena_stats(long long *mapped, long long *in_use, long long *peak_mapped, long long *peak_in_use) {
  DevArena &A = arena();
  std::lock_guard<std::mutex> g(A.m);
  if (mapped) *mapped = (long long)A.mapped;
  if (in_use) *in_use = (long long)A.in_use;
  if (peak_mapped) *peak_mapped = (long long)A.peak_mapped;
  if (peak_in_use) *peak_in_use = (long long)A.peak_in_use;
}

void *dev_alloc(size_t bytes) {
  DevPool &P = pool();
  P.init();
  void *p = nullptr;
  if (!P.enabled) {
    MI_HIP(hipMalloc(&p, bytes));
    return p;
  }
  if (P.enabled == 2) {
    DevArena &A = arena();
    std::unique_lock<std::mutex> g(A.m);
    if (A.init()) {
      p = A.alloc(bytes, g);
      if (!p)
        fail(2, "device arena: cannot map " + std::to_string(bytes) + " more bytes (" + std::to_string(A.mapped) +
                    " mapped, " + std::to_string(A.in_use) + " in use): out of device memory");
      // MI_HYPRE_POISON_ALLOC=1 (debugging / tests): every block is handed out full of 0xFF bytes -- NaN as a double, -1 as
      // an integer.  A chunk is zero the first time it is used (the driver clears it) and holds its last owner's data ever
      // after, so code that reads memory it has not written works in a short test and fails in a long run; with the
      // poison it fails at once.  (Synchronous: the block's last owner was drained by A.alloc when that was needed.)
      static const bool poison = getenv("MI_HYPRE_POISON_ALLOC") && atoi(getenv("MI_HYPRE_POISON_ALLOC")) != 0;
      if (poison && bytes) {
        g.unlock();
        drain_library_streams();
        MI_HIP(hipMemset(p, 0xFF, bytes));
        MI_HIP(hipDeviceSynchronize());
      }
      return p;
    }
    P.enabled = 1;  // no virtual-memory API here: the size-class cache
    if (getenv("MI_HYPRE_VERBOSE")) fprintf(stderr, "mi_hypre: HIP virtual-memory API unavailable, using the block cache\n");
  }
  const size_t want = pool_round(bytes);
  {
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.free_blocks.lower_bound(want);
    // a cached block serves a request it does not waste more than 1/8 of (small blocks: any of the same class)
    if (it != P.free_blocks.end() && it->first <= want + std::max<size_t>(want / 8, 4096)) {
      p = it->second;
      P.cached -= it->first;
      P.free_blocks.erase(it);
      P.hits++;
    } else {
      P.misses++;
    }
  }
  if (p) {
    // the previous owner's kernels may still be running: what hipFree's implicit synchronisation took care of
    drain_library_streams();
    return p;
  }
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    dev_pool_trim();  // the cached blocks may be what is missing
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess)
    fail(2, std::string("hipMalloc of ") + std::to_string(want) + " bytes failed: " + hipGetErrorString(e));
  std::lock_guard<std::mutex> g(P.m);
  P.size_of[p] = want;
  return p;
}

void dev_free(void *p) {
  if (!p) return;
  DevPool &P = pool();
  P.init();
  if (!P.enabled) {
    (void)hipFree(p);
    return;
  }
  {
    DevArena &A = arena();
    std::lock_guard<std::mutex> g(A.m);
    if (A.owns(p)) {
      A.release(p);
      return;
    }
  }
  bool trim = false;
  {
    std::lock_guard<std::mutex> g(P.m);
    auto it = P.size_of.find(p);
    if (it == P.size_of.end()) {  // not ours (allocated while the pool was off)
      (void)hipFree(p);
      return;
    }
    P.free_blocks.insert({it->second, p});
    P.cached += it->second;
    trim = P.cached > P.max_cached;
  }
  if (trim) dev_pool_trim();
}

void DevCSR::upload(const HostCSR &h) {
  nrows = h.nrows;
  ncols = h.ncols;
  nnz = h.nnz();
  require_int32_block(nrows, nnz, "solve format (host builder: small operators; large ones go through sk::to_solve_format)");
  ia64.release();
  std::vector<int> ia32((size_t)nrows + 1);
  for (int i = 0; i <= nrows; i++) ia32[(size_t)i] = (int)h.ia[(size_t)i];
  ia.upload(ia32);
  ja.upload(h.ja);
  a.upload(h.a);
  {
    std::vector<int> lens((size_t)nrows);
    for (int i = 0; i < nrows; i++) lens[(size_t)i] = (int)(h.ia[(size_t)i + 1] - h.ia[(size_t)i]);
    if (nrows) {
      const size_t kth = (size_t)((double)(nrows - 1) * 0.95);
      std::nth_element(lens.begin(), lens.begin() + (long)kth, lens.end());
      rowlen_p95 = lens[kth];
    }
  }
  bool aligned = false;
  tile_entries = k::choose_tile_entries(nnz, nrows);
  std::vector<int> blocks = k::build_row_blocks(nrows, h.ia.data(), &aligned, row_cap, tile_entries);
  if (row_cap > (tile_entries == k::SPMV_TILE_WIDE ? k::SPMV_BLOCK_WIDE : k::SPMV_BLOCK)) aligned = false;  // such tiles are not for the tile Gauss-Seidel kernel
  nblocks = (int)blocks.size() - 1;
  rb.upload(blocks);
  rb_host = blocks;
  gs_tiles = false;
  max_tile_rows = 1;
  for (size_t b = 0; b + 1 < blocks.size(); b++) max_tile_rows = std::max(max_tile_rows, blocks[b + 1] - blocks[b]);
  // x cache: worth it when a block's entries share columns (long rows); the fine
  // level's short rows gather coalesced already
  static const int xc_min = getenv("MI_HYPRE_XCACHE_MIN") ? atoi(getenv("MI_HYPRE_XCACHE_MIN")) : 3;
  xcache = nrows > 0 && (double)nnz / (double)nrows >= (double)xc_min;
  if (xcache) {
    std::vector<int> up((size_t)nblocks + 1, 0);
    std::vector<unsigned short> lc((size_t)nnz, 0);
    std::vector<std::vector<int>> uniq((size_t)nblocks);
    parallel_for(nblocks, [&](int64_t b0, int64_t b1, int) {
      std::vector<int> tmp;
      for (int64_t b = b0; b < b1; b++) {
        const int64_t s = h.ia[(size_t)blocks[(size_t)b]], e = h.ia[(size_t)blocks[(size_t)b + 1]];
        if (e - s >= tile_entries) continue;  // single long row: direct gathers
        tmp.assign(h.ja.begin() + s, h.ja.begin() + e);
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        for (int64_t q = s; q < e; q++)
          lc[(size_t)q] = (unsigned short)(std::lower_bound(tmp.begin(), tmp.end(), h.ja[(size_t)q]) - tmp.begin());
        // in-chunk code bits for the tile Gauss-Seidel kernel
        for (int i = blocks[(size_t)b]; i < blocks[(size_t)b + 1]; i++)
          for (int64_t q = h.ia[(size_t)i]; q < h.ia[(size_t)i + 1]; q++) {
            const int j = h.ja[(size_t)q];
            if ((j >> 3) == (i >> 3)) lc[(size_t)q] |= (unsigned short)(k::XC_INCH | ((j & 7) << k::XC_OFF_SHIFT));
          }
        uniq[(size_t)b] = tmp;
      }
    });
    for (int b = 0; b < nblocks; b++) up[(size_t)b + 1] = up[(size_t)b] + (int)uniq[(size_t)b].size();
    std::vector<int> uc((size_t)up[(size_t)nblocks]);
    for (int b = 0; b < nblocks; b++)
      if (!uniq[(size_t)b].empty()) memcpy(uc.data() + up[(size_t)b], uniq[(size_t)b].data(), uniq[(size_t)b].size() * sizeof(int));
    uptr.upload(up);
    ucols.upload(uc);
    lcol.upload(lc);
    gs_tiles = aligned && nrows == ncols;
  }
  {
    DVec<long long> wide;
    std::vector<long long> hia(h.ia.begin(), h.ia.end());
    if (hia.empty()) hia.assign((size_t)nrows + 1, 0);
    wide.upload(hia);
    k::build_tile_desc(*this, wide.p, ctx().stream);
    MI_HIP(hipStreamSynchronize(ctx().stream));
  }
}

void DevOffd::upload(int nrows, const HostCSR &h) {
  next = h.ncols;
  std::vector<int> r, ia32;
  ia32.push_back(0);
  for (int i = 0; h.nnz() > 0 && i < nrows; i++)  // (a block without entries may come without row pointers)
    if (h.ia[(size_t)i + 1] > h.ia[(size_t)i]) {
      r.push_back(i);
      ia32.push_back((int)h.ia[(size_t)i + 1]);
    }
  nrows_c = (int)r.size();
  nnz = h.nnz();
  rows.upload(r);
  ia.upload(ia32);
  ja.upload(h.ja);
  a.upload(h.a);
}

// ---------------------------------------------------------------- KernelTimer
void KernelTimer::enable(int id, size_t capacity) {
  if (id < 0 || id >= k::PROF_COUNT) return;
  while (start[id].size() < capacity) {
    hipEvent_t a, b;
    MI_HIP(hipEventCreate(&a));
    MI_HIP(hipEventCreate(&b));
    start[id].push_back(a);
    stop[id].push_back(b);
  }
  enabled[id] = capacity > 0;  // capacity 0 switches the class off again (no events around its launches)
  used[id] = 0;
  dropped[id] = 0;
}
void KernelTimer::reset() {
  for (int i = 0; i < k::PROF_COUNT; i++) used[i] = 0, dropped[i] = 0;
}
void KernelTimer::collect(int id, long long *count, double *total_ms, double *min_ms) {
  *count = 0;
  *total_ms = 0.0;
  *min_ms = 0.0;
  if (id < 0 || id >= k::PROF_COUNT) return;
  MI_HIP(hipStreamSynchronize(ctx().stream));
  double mn = 1e300;
  for (size_t i = 0; i < used[id]; i++) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, start[id][i], stop[id][i]) == hipSuccess) {
      *total_ms += ms;
      (*count)++;
      if (ms < mn) mn = ms;
    }
  }
  if (*count) *min_ms = mn;
}
KernelTimer::~KernelTimer() {
  for (int i = 0; i < k::PROF_COUNT; i++) {
    for (auto e : start[i]) (void)hipEventDestroy(e);
    for (auto e : stop[i]) (void)hipEventDestroy(e);
  }
}

}  // namespace mi
