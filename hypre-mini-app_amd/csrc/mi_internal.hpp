// Internal definitions shared by the host control plane and the HIP kernels.
// Nothing here is part of the C ABI (see include/*.h for that).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace mi {

using gidx = long long;  // global row/column id (HYPRE_BigInt)

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int code, const std::string &msg) { throw Error(code, msg); }

#define MI_HIP(call)                                                                             \
  do {                                                                                           \
    hipError_t mi_e_ = (call);                                                                   \
    if (mi_e_ != hipSuccess)                                                                     \
      ::mi::fail(1, std::string("HIP error ") + hipGetErrorString(mi_e_) + " in " #call " at " + \
                        __FILE__ + ":" + std::to_string(__LINE__));                              \
  } while (0)

#define MI_REQUIRE(cond, msg)                                                                          \
  do {                                                                                                 \
    if (!(cond)) ::mi::fail(4, std::string(msg) + " (" #cond ") at " + __FILE__ + ":" + std::to_string(__LINE__)); \
  } while (0)

// ---------------------------------------------------------------- device memory
// Behind every DVec (runtime.cpp): one growable arena in a reserved address range (HIP virtual-memory API; physical
// 1 GiB chunks mapped at the top as it grows, best-fit free list with coalescing inside), so that memory the setup
// releases is reused whatever the next request's size and never goes back to the driver while the library works --
// the driver clears returned memory before handing it out again (30 ms per GiB inside hipMalloc: 11 of the 16 s of a
// 512^3 setup in round 3, profiles/r04_setup_split_512_before.txt).  A block released since the library's streams were
// last drained is handed out after draining them (what hipFree's implicit synchronisation gave).  dev_pool_trim() --
// end of Setup, HYPRE_Finalize -- unmaps free chunks from the top beyond a quarter of what is in use.
// MI_HYPRE_POOL: 2 = arena (default), 1 = the size-class block cache of round 3 (MI_HYPRE_POOL_MAX_GB, default 48; also
// the fallback when the virtual-memory API is unavailable), 0 = plain hipMalloc / hipFree.
void *dev_alloc(size_t bytes);
void dev_free(void *p);
void dev_pool_trim();
void dev_pool_stats(long long *cached_bytes, long long *hits, long long *misses);
void dev_arena_hint(size_t bytes_more);
void dev_arena_reserve(size_t bytes_total);
void dev_arena_stats(long long *mapped, long long *in_use, long long *peak_mapped, long long *peak_in_use);
void dev_arena_times(double *t_grow, double *t_drain, long long *grown, long long *drains, double *t_wait = nullptr);

// Device -> host copy that is complete when it returns (it synchronises `s`).  Copies of up to 64 MiB go through a pinned
// staging buffer of the library: a copy into PAGEABLE host memory (a stack variable, a std::vector) has the runtime
// register that memory first, and that registration waits behind any hipMemCreate / hipMemMap in flight on another
// thread -- the arena's grow-ahead thread, which may sit 30 ms per GiB in the driver: measured stalls of 0.1-6 s per
// 4-byte read (profiles/debug/grow_contention.cpp); pinned targets are not affected.
void d2h(void *dst, const void *src, size_t bytes, hipStream_t s);

template <class T>
struct DVec {
  T *p = nullptr;
  size_t n = 0;
  DVec() = default;
  explicit DVec(size_t n_) { alloc(n_); }
  DVec(const DVec &) = delete;
  DVec &operator=(const DVec &) = delete;
  DVec(DVec &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr, o.n = 0; }
  DVec &operator=(DVec &&o) noexcept {
    if (this != &o) {
      release();
      p = o.p, n = o.n;
      o.p = nullptr, o.n = 0;
    }
    return *this;
  }
  ~DVec() { release(); }
  void release() {
    if (p) dev_free((void *)p);
    p = nullptr, n = 0;
  }
  // pad elements are allocated past n (kernels read whole vectors of 2).  Floating-point arrays get them ZEROED: the
  // BLAS-1 kernels run over pairs and include the element past an odd n in their sums.  Integer arrays (row pointers,
  // columns, lists, marks, the setup's temporaries -- four fifths of the ~2400 allocations of a 512^3 setup) only need the
  // pad to exist: every kernel that loads them in pairs masks the element past the end (a launch and a synchronisation
  // per allocation was 0.1 s of a 2.6 s setup).  MI_HYPRE_POISON_ALLOC=1 fills them with 0xFF in the tests.
  void alloc(size_t n_, size_t pad = 2) {
    release();
    n = n_;
    p = (T *)dev_alloc((n + pad) * sizeof(T));
    if (pad && std::is_floating_point<T>::value) {  // finished before any stream can touch the allocation
      MI_HIP(hipMemsetAsync((void *)(p + n), 0, pad * sizeof(T), nullptr));
      MI_HIP(hipStreamSynchronize(nullptr));
    }
  }
  void upload(const T *h, size_t cnt) { MI_HIP(hipMemcpy(p, h, cnt * sizeof(T), hipMemcpyHostToDevice)); }
  void upload(const std::vector<T> &h) {
    alloc(h.size());
    if (!h.empty()) upload(h.data(), h.size());
  }
  void download(T *h, size_t cnt) const { d2h(h, p, cnt * sizeof(T), nullptr); }
  std::vector<T> to_host() const {
    std::vector<T> h(n);
    if (n) download(h.data(), n);
    return h;
  }
};

// A per-row integer array of a level (C/F marks, the C-first permutation, the internal numbering) that is born on the
// device when the device builds the level and whose host copy is made only when somebody asks for it (the inspection
// API of the parity tests, the host routines of small levels, the multi-rank setup): at 512^3 every such array is
// 0.5 GB, and copying them to the host, looping over them there and uploading the results was ~1.5 s of a setup in
// which the device waited (round 4, profiles/r04_setup_split_512_arena.txt).  Reads look like a const vector.
struct LazyInts {
  mutable std::vector<int> h;
  mutable bool host_ok = true;  // h holds the data (an empty array is host_ok with h empty)
  mutable DVec<int> d;
  mutable bool dev_ok = false;
  mutable size_t nd = 0;  // length of the device copy
  size_t size() const { return host_ok ? h.size() : nd; }
  bool empty() const { return size() == 0; }
  const std::vector<int> &host() const {
    if (!host_ok) {
      h.resize(nd);
      if (nd) d2h(h.data(), d.p, nd * sizeof(int), nullptr);
      host_ok = true;
    }
    return h;
  }
  // the host copy for writing: whatever the device holds is dropped afterwards
  std::vector<int> &hostw() {
    (void)host();
    d.release();
    dev_ok = false;
    nd = 0;
    return h;
  }
  int operator[](size_t i) const { return host()[i]; }
  void clear() {
    h.clear();
    h.shrink_to_fit();
    host_ok = true;
    d.release();
    dev_ok = false;
    nd = 0;
  }
  LazyInts &operator=(const std::vector<int> &v) {
    clear();
    h = v;
    return *this;
  }
  LazyInts &operator=(std::vector<int> &&v) {
    clear();
    h = std::move(v);
    return *this;
  }
  // take a device array of n entries: it is the data now
  void adopt_device(DVec<int> &&v, size_t n) {
    h.clear();
    h.shrink_to_fit();
    d = std::move(v);
    nd = n;
    dev_ok = true;
    host_ok = (n == 0);
  }
  bool on_device() const { return dev_ok; }
  // device copy (uploaded when only the host has the data); null for an empty array
  const int *dev() const {
    if (!dev_ok && !h.empty()) {
      d.upload(h);
      nd = h.size();
      dev_ok = true;
    }
    return dev_ok ? d.p : nullptr;
  }
};

// ---------------------------------------------------------------- host CSR
struct HostCSR {
  int nrows = 0, ncols = 0;
  std::vector<int64_t> ia;  // nrows+1
  std::vector<int> ja;      // columns ascending inside a row
  std::vector<double> a;
  int64_t nnz() const { return ia.empty() ? 0 : ia.back(); }
};

// ---------------------------------------------------------------- device CSR
// Row pointers: `ia` holds the LOW 32 bits of every entry offset (4 bytes per row in the solve kernels' streams).
// The tile kernels (spmv_stream_xc, gs_tile_k) take a tile's full 64-bit base from its descriptor and form
// tile-local offsets as (unsigned)ia[row] - (unsigned)base, which is exact because a tile holds < 2^16 entries -- so
// an operator may hold more than 2^31 entries (the reference generator's 27-point operator at 512^3: 3.6e9,
// /root/reference/src/laplace_3d_weak_scaling.hpp:558,600).  Such an operator ("big") also keeps the 64-bit row
// pointers (`ia64`, 8 bytes per row, setup paths only) and must run on the tile kernels; the other kernels
// (plain row-block SpMV, chunk Gauss-Seidel, two-stage sweeps) index with `ia` directly and refuse it.
struct DevCSR {
  int nrows = 0, ncols = 0;
  int64_t nnz = 0;
  DVec<int> ia;
  DVec<long long> ia64;  // only when nnz >= 2^31
  bool big() const { return nnz >= (int64_t)2147483647; }
  DVec<int> ja;
  DVec<double> a;
  // row-block schedule of the LDS-staged SpMV (kernels.hip: spmv_stream)
  DVec<int> rb;
  int nblocks = 0;
  int rowlen_p95 = 0;  // 95th percentile of the row lengths (GS kernel variant choice)
  // x cache of the SpMV (levels with long rows): per row block the sorted unique
  // columns (gathered once into LDS) and 16-bit block-local column ids per entry
  bool xcache = false;
  // Gauss-Seidel on the SpMV tiles (gs_tile_k): the row blocks start and end on multiples of 8 rows, no
  // 8-row chunk exceeds a tile, and every lcol entry carries the in-chunk code in its upper bits
  // (bit 15: the column lies in the row's own chunk of 8 rows, bits 12-14: which of the 8)
  bool gs_tiles = false;
  int tile_entries = 2048;  // entries per tile: k::SPMV_TILE, or k::SPMV_TILE_WIDE for operators with long rows
  int row_cap = 0;  // rows per tile (0 = SPMV_BLOCK; set before upload / to_solve_format; more only for SpMV-only operators)
  bool prefer_gs_tiles = false;  // sweep this operator with the tile kernel whatever its mean row length
  std::vector<int> rb_host;  // host copy of rb (row ranges -> tile ranges)
  int max_tile_rows = 256;   // rows of the largest tile
  DVec<int> uptr, ucols;
  // Round 4: the column lists in 2 instead of 4 bytes per unique column.  A tile's sorted unique columns fall into few
  // aligned blocks of 1024 column ids (17-24 on the levels of a 3-D Laplacian hierarchy, never more than 52 at 128^3:
  // a tile's rows and their neighbours live in a handful of cells of the internal numbering), so an id is a 6-bit
  // selector of one of <= 64 block numbers of the tile (`ubase`, loaded once per wave, read by lane shuffle) and a
  // 10-bit offset (`ucode`).  The lists were 1.7 of the 4.7 bytes a level-0 entry of the benchmark costs, and 1.07 of
  // 11.1 on level 1.  A TILE with more than 64 blocks keeps a 4-byte list (`ucols` then holds only those tiles' lists, one
  // after the other; the tile descriptor's last word says which kind and where).
  DVec<unsigned short> ucode;
  DVec<int> ubase;
  long long n_unique = 0;  // total length of the tiles' column lists
  int ucode_max_blocks = 0;  // most 1024-id blocks any tile's columns fall into
  int ucode_wide_tiles = 0;  // tiles with more than 64 blocks: they keep a 4-byte list (in `ucols`, compact)
  DVec<unsigned short> lcol;
  // value dictionary (k::build_value_dictionary): operators with at most 256 distinct values (constant-coefficient
  // stencils such as the reference's own generator: 26 / -1, or 6 / -1) carry one byte per entry besides `a`; the
  // x-cache SpMV and the tile Gauss-Seidel kernel then stream 3 instead of 10 bytes per entry and look the value up
  // in a 2 KB LDS table -- the same doubles, so results do not change by a bit
  DVec<unsigned char> vidx;
  DVec<double> vlut;  // 256 entries (unused ones zero)
  bool val8 = false;
  // Rows stored in another order than the vectors they produce (SpMV only): stored row r is entry rowmap[r] of y
  // (and of b).  The restriction operators use it: their rows are kept in the order of the FINE level's C points --
  // consecutive rows then gather neighbouring fine entries -- while the coarse level's vectors are in its own
  // C-first order (amg_setup.cpp: setup_device).  Empty = identity.
  DVec<int> rowmap;
  DVec<int> tdesc;  // 8 ints per tile: r0, r1, low word of ia[r0], ia[r1] - ia[r0], uptr[b], #unique columns, high word of ia[r0], first entry of the tile in ubase (k::build_tile_desc)
  bool empty() const { return nrows == 0 || nnz == 0; }
  void upload(const HostCSR &h);
};

// compressed-row CSR for the off-diagonal (halo) block: only rows that own
// halo entries are stored
struct DevOffd {
  int nrows_c = 0;  // rows with halo entries
  int next = 0;     // number of halo columns
  int64_t nnz = 0;
  DVec<int> rows;  // local row id per compressed row
  DVec<int> ia, ja;
  DVec<double> a;
  void upload(int nrows, const HostCSR &h);
};

// ---------------------------------------------------------------- communication
enum class CommDType { F64, I64, I32, U8 };
enum class CommOp { SUM, MIN, MAX };

struct PeerBuf {
  int peer;
  void *ptr;
  size_t bytes;
};

// One rank per GPU.  Device-buffer collectives are enqueued on `stream`; the
// host-buffer ones are blocking and only used by setup and the loaders.
struct Comm {
  int rank = 0, size = 1;
  virtual ~Comm() {}
  virtual const char *name() const = 0;
  virtual void allreduce_dev(void *buf, size_t count, CommDType t, CommOp op, hipStream_t s) = 0;
  virtual void exchange_dev(const std::vector<PeerBuf> &sends, const std::vector<PeerBuf> &recvs, hipStream_t s) = 0;
  virtual void allgather_dev(const void *send, void *recv, size_t bytes_per_rank, hipStream_t s) = 0;
  // blocking host-side helpers; the defaults stage through device buffers and
  // the device collectives above
  virtual void allreduce_host(void *buf, size_t count, CommDType t, CommOp op);
  virtual void allgather_host(const void *send, void *recv, size_t bytes_per_rank);
  // fixed-size neighbour exchange of host buffers (sizes known on both sides)
  virtual void exchange_host_fixed(const std::vector<PeerBuf> &sends, const std::vector<PeerBuf> &recvs);
  // variable-size neighbour exchange of host byte strings; recv sizes are learnt
  // through an all-gather of the size matrix
  void exchange_host(const std::vector<int> &peers_send, const std::vector<std::vector<char>> &send,
                     std::vector<int> &peers_recv, std::vector<std::vector<char>> &recv);
  // every rank's byte string on every rank: out = rank 0's | rank 1's | ..., offs[r] = start of rank r's (size+1 entries)
  void allgatherv_host(const void *mine, size_t bytes, std::vector<size_t> &offs, std::vector<char> &out);
  virtual bool host_transport() const { return false; }
  void barrier();
};

std::unique_ptr<Comm> make_self_comm();
// RCCL communicator from a 128-byte ncclUniqueId (dlopen'ed librccl.so.1)
std::unique_ptr<Comm> make_rccl_comm(const void *unique_id, int rank, int size);
void rccl_get_unique_id(void *out128);

// caller-supplied transport (tests: torch.distributed gloo through host staging)
struct CommCallbacks {
  void *ctx;
  // all buffers are HOST pointers; the library stages device data itself
  void (*allreduce)(void *ctx, void *buf, size_t count, int dtype, int op);
  void (*allgather)(void *ctx, const void *send, void *recv, size_t bytes_per_rank);
  void (*exchange)(void *ctx, int nsend, const int *send_peers, void *const *send_ptrs, const size_t *send_bytes,
                   int nrecv, const int *recv_peers, void *const *recv_ptrs, const size_t *recv_bytes);
};
std::unique_ptr<Comm> make_callback_comm(const CommCallbacks &cb, int rank, int size);
// The neighbour exchange by peer stores into IPC-mapped mailboxes (comm.cpp IpcExchangeComm) on top of another
// communicator, which keeps the reductions, gathers and host collectives; size 1: returns inner unchanged
std::unique_ptr<Comm> make_ipc_exchange_comm(std::unique_ptr<Comm> &inner, size_t slot_bytes);
// peer-store transport: raises an error when a bounded wait expired (no-op and false for other transports)
bool comm_check_transport_error(Comm &c, hipStream_t s);
int comm_transport_verdict(Comm &c, hipStream_t s);
// RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT bootstrap (TCP hand-off of the ncclUniqueId)
std::unique_ptr<Comm> make_comm_from_env();

// ---------------------------------------------------------------- runtime context
struct KernelTimer;  // profile.cpp

struct Ctx {
  bool inited = false;
  int device = 0;
  hipStream_t stream = nullptr;
  // side stream of the halo exchanges that overlap with the diag-block SpMV (ParCSR::matvec), and the two
  // events that order it against `stream`
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_packed = nullptr, ev_halo = nullptr;
  std::unique_ptr<Comm> comm;
  // scratch for two-stage reductions
  DVec<double> red_partials;  // MAX_RED_BLOCKS * MAX_RED_SLOTS
  DVec<double> red_out;       // result slots (device scalars)
  DVec<unsigned> red_ticket;  // ticket counter of the reductions' last-block stage (kernels.hip finish_reduction)
  double *h_pinned = nullptr; // pinned host mirror of result slots
  unsigned long long *h_post_flag = nullptr;  // sequence number a posting kernel stores after its values (k::scale_post_k)
  unsigned long long post_seq = 0;
  int gs_chunk = 8;
  int verbose = 0;
  KernelTimer *timer = nullptr;
  // event counters of the multi-rank choreography (HYPRE_MI_GetCounter; the distributed tests check that the
  // overlapped paths are the ones that ran)
  long long n_matvec_overlapped = 0, n_gs_overlapped = 0, n_gs_in_order = 0;
  // collectives the solve phase issued on N > 1 ranks: scalar / block all-reduces (inner products), neighbour
  // exchange groups (one per halo update), all-gathers (coarsest / redundant levels)
  long long n_allreduce = 0, n_halo_exchange = 0, n_allgather = 0;
};
Ctx &ctx();
void ensure_init();
// zero device memory in order with the library stream (hipMemset on the null stream is not: the library
// stream is non-blocking)
void zero_on_stream(void *p, size_t bytes);
Comm &current_comm();  // never needs a device (self comm by default)

// ---------------------------------------------------------------- tracing
// roctx ranges around the phases of the path (rocprofv3 --marker-trace shows them beside the kernels): setup and
// its steps, the Krylov solve, every preconditioner application.  librocprofiler-sdk-roctx is dlopen'ed on first
// use; without it (or with MI_HYPRE_ROCTX=0) a range costs one branch.
struct TraceRange {
  explicit TraceRange(const char *name);
  ~TraceRange();
  TraceRange(const TraceRange &) = delete;
  TraceRange &operator=(const TraceRange &) = delete;
  bool on = false;
};

// ---------------------------------------------------------------- small helpers
void parallel_for(int64_t n, const std::function<void(int64_t, int64_t, int)> &fn, int max_threads = 0);
int host_threads();
double wall_time();

}  // namespace mi
