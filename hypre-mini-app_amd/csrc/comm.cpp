// One rank per GPU.  Three transports behind one interface:
//   self      size-1 no-op
//   rccl      RCCL over xGMI, loaded with dlopen so that a single-GPU run has no
//             dependency on it and a torch process shares torch's copy
//   callback  caller-supplied host transport (tests: gloo; several ranks may then
//             share one GPU, which RCCL refuses)
//   tcp       MI_HYPRE_TRANSPORT=tcp: the callback transport over a mesh of TCP sockets on one node -- what the C++
//             driver uses when several ranks share a GPU (tests) or librccl is not there; host-staged, not a fast path
// The halo exchange is a neighbour send/recv group (<= 2 peers for slab
// partitions), the dot products are 8-byte all-reduces: both latency-bound, so
// what matters is how few of them the solver issues, not their algorithm.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <rccl/rccl.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <thread>

#include <cstring>

#include "kernels.hpp"
#include "mi_internal.hpp"

namespace mi {

// ------------------------------------------------------------------ host helpers (blocking)
static size_t dtype_size(CommDType t) {
  switch (t) {
    case CommDType::F64: return 8;
    case CommDType::I64: return 8;
    case CommDType::I32: return 4;
    default: return 1;
  }
}

void Comm::allreduce_host(void *buf, size_t count, CommDType t, CommOp op) {
  if (size == 1 || count == 0) return;
  ensure_init();
  hipStream_t s = ctx().stream;
  const size_t bytes = count * dtype_size(t);
  DVec<char> d(bytes);
  MI_HIP(hipMemcpyAsync(d.p, buf, bytes, hipMemcpyHostToDevice, s));
  allreduce_dev(d.p, count, t, op, s);
  MI_HIP(hipMemcpyAsync(buf, d.p, bytes, hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
}

void Comm::allgather_host(const void *send, void *recv, size_t bytes_per_rank) {
  if (size == 1) {
    memcpy(recv, send, bytes_per_rank);
    return;
  }
  ensure_init();
  hipStream_t s = ctx().stream;
  DVec<char> ds(bytes_per_rank), dr(bytes_per_rank * (size_t)size);
  MI_HIP(hipMemcpyAsync(ds.p, send, bytes_per_rank, hipMemcpyHostToDevice, s));
  allgather_dev(ds.p, dr.p, bytes_per_rank, s);
  MI_HIP(hipMemcpyAsync(recv, dr.p, bytes_per_rank * (size_t)size, hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
}

void Comm::barrier() {
  if (size == 1) return;
  long long one = 1;
  allreduce_host(&one, 1, CommDType::I64, CommOp::SUM);
}

void Comm::exchange_host_fixed(const std::vector<PeerBuf> &sends, const std::vector<PeerBuf> &recvs) {
  if (size == 1 || (sends.empty() && recvs.empty())) return;
  ensure_init();
  hipStream_t s = ctx().stream;
  std::vector<DVec<char>> dsend(sends.size()), drecv(recvs.size());
  std::vector<PeerBuf> sb, rb;
  for (size_t i = 0; i < sends.size(); i++) {
    dsend[i].alloc(sends[i].bytes);
    MI_HIP(hipMemcpyAsync(dsend[i].p, sends[i].ptr, sends[i].bytes, hipMemcpyHostToDevice, s));
    sb.push_back({sends[i].peer, dsend[i].p, sends[i].bytes});
  }
  for (size_t i = 0; i < recvs.size(); i++) {
    drecv[i].alloc(recvs[i].bytes);
    rb.push_back({recvs[i].peer, drecv[i].p, recvs[i].bytes});
  }
  exchange_dev(sb, rb, s);
  for (size_t i = 0; i < recvs.size(); i++)
    MI_HIP(hipMemcpyAsync(recvs[i].ptr, drecv[i].p, recvs[i].bytes, hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
}

void Comm::exchange_host(const std::vector<int> &peers_send, const std::vector<std::vector<char>> &send,
                         std::vector<int> &peers_recv, std::vector<std::vector<char>> &recv) {
  peers_recv.clear();
  recv.clear();
  if (size == 1) return;
  std::vector<long long> mine((size_t)size, 0), all((size_t)size * size, 0);
  for (size_t i = 0; i < peers_send.size(); i++) mine[(size_t)peers_send[i]] = (long long)send[i].size();
  allgather_host(mine.data(), all.data(), sizeof(long long) * (size_t)size);
  std::vector<PeerBuf> sb, rb;
  for (size_t i = 0; i < peers_send.size(); i++) {
    if (send[i].empty()) continue;
    sb.push_back({peers_send[i], (void *)send[i].data(), send[i].size()});
  }
  for (int p = 0; p < size; p++) {
    const long long bytes = all[(size_t)p * size + rank];
    if (p == rank || bytes == 0) continue;
    peers_recv.push_back(p);
    recv.emplace_back((size_t)bytes);
  }
  for (size_t i = 0; i < peers_recv.size(); i++) rb.push_back({peers_recv[i], recv[i].data(), recv[i].size()});
  exchange_host_fixed(sb, rb);
}

void Comm::allgatherv_host(const void *mine, size_t bytes, std::vector<size_t> &offs, std::vector<char> &out) {
  std::vector<long long> cnt((size_t)size, 0);
  long long b = (long long)bytes;
  allgather_host(&b, cnt.data(), sizeof(long long));
  offs.assign((size_t)size + 1, 0);
  for (int r = 0; r < size; r++) offs[(size_t)r + 1] = offs[(size_t)r] + (size_t)cnt[(size_t)r];
  out.resize(offs[(size_t)size]);
  if (bytes) memcpy(out.data() + offs[(size_t)rank], mine, bytes);
  if (size == 1) return;
  std::vector<PeerBuf> sb, rb;
  if (host_transport()) {
    for (int r = 0; r < size; r++) {
      if (r == rank) continue;
      if (bytes) sb.push_back({r, const_cast<void *>(mine), bytes});
      if (cnt[(size_t)r]) rb.push_back({r, out.data() + offs[(size_t)r], (size_t)cnt[(size_t)r]});
    }
    exchange_host_fixed(sb, rb);
    return;
  }
  // device transport: one upload of this rank's string, one download of everybody's
  ensure_init();
  hipStream_t s = ctx().stream;
  DVec<char> ds(bytes), dr(out.size());
  if (bytes) MI_HIP(hipMemcpyAsync(ds.p, mine, bytes, hipMemcpyHostToDevice, s));
  for (int r = 0; r < size; r++) {
    if (r == rank) continue;
    if (bytes) sb.push_back({r, ds.p, bytes});
    if (cnt[(size_t)r]) rb.push_back({r, dr.p + offs[(size_t)r], (size_t)cnt[(size_t)r]});
  }
  exchange_dev(sb, rb, s);
  for (int r = 0; r < size; r++)
    if (r != rank && cnt[(size_t)r])
      MI_HIP(hipMemcpyAsync(out.data() + offs[(size_t)r], dr.p + offs[(size_t)r], (size_t)cnt[(size_t)r],
                            hipMemcpyDeviceToHost, s));
  MI_HIP(hipStreamSynchronize(s));
}

// ------------------------------------------------------------------ self
namespace {
struct SelfComm : Comm {
  const char *name() const override { return "self"; }
  void allreduce_dev(void *, size_t, CommDType, CommOp, hipStream_t) override {}
  void exchange_dev(const std::vector<PeerBuf> &, const std::vector<PeerBuf> &, hipStream_t) override {}
  void allgather_dev(const void *send, void *recv, size_t bytes, hipStream_t s) override {
    MI_HIP(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s));
  }
};

// ------------------------------------------------------------------ RCCL (dlopen)
struct RcclApi {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

RcclApi &rccl() {
  static RcclApi api;
  if (api.h) return api;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (api.h) break;
  }
  if (!api.h) fail(1, std::string("mi_hypre: cannot dlopen librccl: ") + dlerror());
#define LOAD(sym)                                                          \
  api.sym = reinterpret_cast<decltype(api.sym)>(dlsym(api.h, "nccl" #sym)); \
  if (!api.sym) fail(1, "mi_hypre: librccl lacks nccl" #sym)
  LOAD(GetUniqueId);
  LOAD(CommInitRank);
  LOAD(CommDestroy);
  LOAD(AllReduce);
  LOAD(AllGather);
  LOAD(Send);
  LOAD(Recv);
  LOAD(GroupStart);
  LOAD(GroupEnd);
  LOAD(GetErrorString);
#undef LOAD
  return api;
}

#define MI_NCCL(call)                                                                                          \
  do {                                                                                                         \
    ncclResult_t r_ = (call);                                                                                  \
    if (r_ != ncclSuccess)                                                                                     \
      fail(1, std::string("RCCL error ") + rccl().GetErrorString(r_) + " in " #call " at " + __FILE__ + ":" + \
                  std::to_string(__LINE__));                                                                   \
  } while (0)

struct RcclComm : Comm {
  ncclComm_t comm = nullptr;
  RcclComm(const void *id128, int rank_, int size_) {
    rank = rank_;
    size = size_;
    ncclUniqueId id;
    static_assert(sizeof(id) == 128, "ncclUniqueId size");
    memcpy(&id, id128, sizeof(id));
    MI_NCCL(rccl().CommInitRank(&comm, size, id, rank));
  }
  ~RcclComm() override {
    if (comm) (void)rccl().CommDestroy(comm);
  }
  const char *name() const override { return "rccl"; }
  static ncclDataType_t dt(CommDType t) {
    switch (t) {
      case CommDType::F64: return ncclFloat64;
      case CommDType::I64: return ncclInt64;
      case CommDType::I32: return ncclInt32;
      default: return ncclUint8;
    }
  }
  static ncclRedOp_t op(CommOp o) { return o == CommOp::SUM ? ncclSum : (o == CommOp::MIN ? ncclMin : ncclMax); }
  void allreduce_dev(void *buf, size_t count, CommDType t, CommOp o, hipStream_t s) override {
    MI_NCCL(rccl().AllReduce(buf, buf, count, dt(t), op(o), comm, s));
  }
  void allgather_dev(const void *send, void *recv, size_t bytes, hipStream_t s) override {
    MI_NCCL(rccl().AllGather(send, recv, bytes, ncclUint8, comm, s));
  }
  void exchange_dev(const std::vector<PeerBuf> &sends, const std::vector<PeerBuf> &recvs, hipStream_t s) override {
    if (sends.empty() && recvs.empty()) return;
    MI_NCCL(rccl().GroupStart());
    for (const auto &b : recvs) MI_NCCL(rccl().Recv(b.ptr, b.bytes, ncclUint8, b.peer, comm, s));
    for (const auto &b : sends) MI_NCCL(rccl().Send(b.ptr, b.bytes, ncclUint8, b.peer, comm, s));
    MI_NCCL(rccl().GroupEnd());
  }
};

// ------------------------------------------------------------------ callback (host-staged)
struct CallbackComm : Comm {
  CommCallbacks cb;
  CallbackComm(const CommCallbacks &c, int rank_, int size_) : cb(c) {
    rank = rank_;
    size = size_;
  }
  const char *name() const override { return "callback"; }
  bool host_transport() const override { return true; }
  // the transport is a host transport: host collectives go straight through
  void allreduce_host(void *buf, size_t count, CommDType t, CommOp o) override {
    if (size == 1 || count == 0) return;
    cb.allreduce(cb.ctx, buf, count, (int)t, (int)o);
  }
  void allgather_host(const void *send, void *recv, size_t bytes) override {
    cb.allgather(cb.ctx, send, recv, bytes);
  }
  void exchange_host_fixed(const std::vector<PeerBuf> &sends, const std::vector<PeerBuf> &recvs) override {
    if (sends.empty() && recvs.empty()) return;
    std::vector<int> sp, rp;
    std::vector<void *> sptr, rptr;
    std::vector<size_t> sby, rby;
    for (auto &b : sends) sp.push_back(b.peer), sptr.push_back(b.ptr), sby.push_back(b.bytes);
    for (auto &b : recvs) rp.push_back(b.peer), rptr.push_back(b.ptr), rby.push_back(b.bytes);
    cb.exchange(cb.ctx, (int)sp.size(), sp.data(), sptr.data(), sby.data(), (int)rp.size(), rp.data(), rptr.data(),
                rby.data());
  }
  void allreduce_dev(void *buf, size_t count, CommDType t, CommOp o, hipStream_t s) override {
    const size_t bytes = count * dtype_size(t);
    std::vector<char> h(bytes);
    MI_HIP(hipMemcpyAsync(h.data(), buf, bytes, hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    cb.allreduce(cb.ctx, h.data(), count, (int)t, (int)o);
    MI_HIP(hipMemcpyAsync(buf, h.data(), bytes, hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));
  }
  void allgather_dev(const void *send, void *recv, size_t bytes, hipStream_t s) override {
    std::vector<char> hs(bytes), hr(bytes * (size_t)size);
    MI_HIP(hipMemcpyAsync(hs.data(), send, bytes, hipMemcpyDeviceToHost, s));
    MI_HIP(hipStreamSynchronize(s));
    cb.allgather(cb.ctx, hs.data(), hr.data(), bytes);
    MI_HIP(hipMemcpyAsync(recv, hr.data(), hr.size(), hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));
  }
  void exchange_dev(const std::vector<PeerBuf> &sends, const std::vector<PeerBuf> &recvs, hipStream_t s) override {
    if (sends.empty() && recvs.empty()) return;
    std::vector<std::vector<char>> hs(sends.size()), hr(recvs.size());
    std::vector<int> sp, rp;
    std::vector<void *> sptr, rptr;
    std::vector<size_t> sby, rby;
    for (size_t i = 0; i < sends.size(); i++) {
      hs[i].resize(sends[i].bytes);
      MI_HIP(hipMemcpyAsync(hs[i].data(), sends[i].ptr, sends[i].bytes, hipMemcpyDeviceToHost, s));
      sp.push_back(sends[i].peer);
      sptr.push_back(hs[i].data());
      sby.push_back(sends[i].bytes);
    }
    for (size_t i = 0; i < recvs.size(); i++) {
      hr[i].resize(recvs[i].bytes);
      rp.push_back(recvs[i].peer);
      rptr.push_back(hr[i].data());
      rby.push_back(recvs[i].bytes);
    }
    MI_HIP(hipStreamSynchronize(s));
    cb.exchange(cb.ctx, (int)sp.size(), sp.data(), sptr.data(), sby.data(), (int)rp.size(), rp.data(), rptr.data(),
                rby.data());
    for (size_t i = 0; i < recvs.size(); i++)
      MI_HIP(hipMemcpyAsync(recvs[i].ptr, hr[i].data(), recvs[i].bytes, hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));
  }
};
// ------------------------------------------------------------------ peer-store exchange over IPC-mapped mailboxes
// The neighbour exchange of the halo updates without ncclSend/ncclRecv: every rank owns a mailbox arena in its HBM
// (per peer: two slots of slot_bytes, a flag word and an ack word per slot), exported with hipIpcGetMemHandle and
// mapped by every other rank (hipIpcOpenMemHandle: peer memory over xGMI, or the same device when ranks share a
// GPU).  One kernel launch per exchange (k::ipc_exchange): the sender stores its packed halo straight into the
// receiver's slot and publishes the message number; the receiver waits for it, copies the slot to its x_ext and
// acknowledges.  Two slots per directed pair: message k + 2 waits for the acknowledgement of message k, which the
// receiver posted long ago -- the sender practically never waits.  Everything else (all-reduce, all-gather, host
// collectives, the bootstrap of the handles) goes through the communicator this one wraps (RCCL, or the caller's
// callbacks).  Waits are bounded (MI_HYPRE_IPC_TIMEOUT_MS, default 20 s): a missing peer becomes an error, not a hang.
struct IpcExchangeComm : Comm {
  std::unique_ptr<Comm> inner;
  size_t slot_bytes = 0;
  char *arena = nullptr;                 // my mailboxes: [flags: size*2][acks: size*2][slots: size*2*slot_bytes]
  std::vector<char *> peer_arena;        // every rank's arena as mapped here (self: my own pointer)
  std::vector<unsigned long long> send_seq, recv_seq;
  DVec<unsigned> tickets;
  DVec<int> error_flag;
  unsigned long long spin_limit = 0;
  long long n_launch = 0;
  std::string label;

  size_t header_bytes() const { return (size_t)size * 4 * sizeof(unsigned long long); }
  // all-reduce area behind the message slots: [2][size] numbers, then [2][size][IPC_AR_MAX] values
  size_t ar_offset() const { return ((header_bytes() + 255) / 256) * 256 + (size_t)size * 2 * slot_bytes; }
  unsigned long long *ar_flags(char *base) const { return reinterpret_cast<unsigned long long *>(base + ar_offset()); }
  double *ar_slots(char *base) const {
    return reinterpret_cast<double *>(base + ar_offset() + ((size_t)2 * size * sizeof(unsigned long long) + 255) / 256 * 256);
  }
  size_t ar_bytes() const {
    return ((size_t)2 * size * sizeof(unsigned long long) + 255) / 256 * 256 + (size_t)2 * size * k::IPC_AR_MAX * sizeof(double);
  }
  unsigned long long ar_seq = 0;
  long long n_allreduce_ipc = 0;
  unsigned long long *flag_word(char *base, int from, int slot) const {
    return reinterpret_cast<unsigned long long *>(base) + (size_t)from * 2 + slot;
  }
  unsigned long long *ack_word(char *base, int from, int slot) const {
    return reinterpret_cast<unsigned long long *>(base) + (size_t)size * 2 + (size_t)from * 2 + slot;
  }
  char *slot_ptr(char *base, int from, int slot) const {
    return base + ((header_bytes() + 255) / 256) * 256 + ((size_t)from * 2 + slot) * slot_bytes;
  }

  // Built on a communicator it does not own yet (`in` stays with the caller until make_ipc_exchange_comm adopts it): a
  // refusal -- thrown on EVERY rank, decided from data all ranks hold alike -- leaves the caller's transport in place.
  IpcExchangeComm(Comm &in, size_t slot) : slot_bytes(slot) {
    ensure_init();
    rank = in.rank;
    size = in.size;
    label = std::string("ipc-peer-store + ") + in.name();
    const char *tm = getenv("MI_HYPRE_IPC_TIMEOUT_MS");
    spin_limit = (unsigned long long)(tm ? atoll(tm) : 20000) * 100000ull;  // wall_clock64 ticks at 100 MHz
    const size_t total = ar_offset() + ar_bytes();
    // Fine-grained device memory: flags and payload are written by OTHER devices while this one's kernels poll them
    // (system-scope atomics; coarse-grained memory may keep stale lines in this device's L2 until a kernel boundary).
    // Ordinary device memory is only valid when ALL ranks share one device (the one-GPU test box): every rank
    // publishes the identity of its device with its handle, and a communicator whose ranks sit on different devices
    // refuses mailboxes that are not fine-grained on every rank (MI_HYPRE_IPC_FINEGRAINED=0, which forces
    // ordinary memory, is subject to the same check).
    struct Card {
      hipIpcMemHandle_t handle;
      char bus[32];
      int status;  // 1 fine-grained arena exported, 2 ordinary arena exported, 0 nothing
    } mine;
    memset(&mine, 0, sizeof(mine));
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetPCIBusId(mine.bus, (int)sizeof(mine.bus), dev) != hipSuccess) {
      (void)hipGetLastError();
      snprintf(mine.bus, sizeof(mine.bus), "device-%d-of-pid-%d", dev, (int)getpid());  // (unknown: never equal to a peer's)
    }
    if (const char *fake = getenv("MI_HYPRE_IPC_BUS_ID")) {  // test hook (tests/ipc_worker.py): the identity this rank publishes
      memset(mine.bus, 0, sizeof(mine.bus));
      strncpy(mine.bus, fake, sizeof(mine.bus) - 1);
    }
    const bool want_fine = !(getenv("MI_HYPRE_IPC_FINEGRAINED") && atoi(getenv("MI_HYPRE_IPC_FINEGRAINED")) == 0);
    if (want_fine && hipExtMallocWithFlags((void **)&arena, total, hipDeviceMallocFinegrained) == hipSuccess) {
      if (hipIpcGetMemHandle(&mine.handle, arena) == hipSuccess) {
        mine.status = 1;
      } else {
        (void)hipGetLastError();
        (void)hipFree(arena);
        arena = nullptr;
      }
    } else {
      (void)hipGetLastError();
      arena = nullptr;
    }
    if (!mine.status) {
      if (hipMalloc((void **)&arena, total) == hipSuccess && hipIpcGetMemHandle(&mine.handle, arena) == hipSuccess) {
        mine.status = 2;
      } else {
        (void)hipGetLastError();
        if (arena) (void)hipFree(arena);
        arena = nullptr;
      }
    }
    if (arena) {
      MI_HIP(hipMemset(arena, 0, total));
      MI_HIP(hipDeviceSynchronize());
    }
    std::vector<Card> all((size_t)size);
    in.allgather_host(&mine, all.data(), sizeof(Card));
    bool any_failed = false, all_fine = true, one_device = true;
    for (int r = 0; r < size; r++) {
      any_failed = any_failed || all[(size_t)r].status == 0;
      all_fine = all_fine && all[(size_t)r].status == 1;
      one_device = one_device && strncmp(all[(size_t)r].bus, all[0].bus, sizeof(mine.bus)) == 0;
    }
    auto refuse = [&](const std::string &why) {
      if (arena) (void)hipFree(arena);
      arena = nullptr;
      fail(1, "peer-store exchange refused (the communicator keeps its transport): " + why);
    };
    if (any_failed) refuse("a rank could not allocate or export its mailbox arena");
    if (!one_device && !all_fine)
      refuse("the ranks are on different devices and not every mailbox arena is fine-grained device memory -- peer "
             "stores into ordinary device memory may stay invisible to the polling device until a kernel boundary");
    label += all_fine ? " (fine-grained mailboxes)" : " (coarse-grained mailboxes, ranks share one device)";
    peer_arena.assign((size_t)size, nullptr);
    int opened = 1;
    for (int r = 0; r < size; r++) {
      if (r == rank) {
        peer_arena[(size_t)r] = arena;
        continue;
      }
      void *p = nullptr;
      if (hipIpcOpenMemHandle(&p, all[(size_t)r].handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        opened = 0;
        break;
      }
      peer_arena[(size_t)r] = (char *)p;
    }
    in.allreduce_host(&opened, 1, CommDType::I32, CommOp::MIN);
    if (!opened) {
      for (int r = 0; r < size; r++)
        if (r != rank && peer_arena[(size_t)r]) (void)hipIpcCloseMemHandle(peer_arena[(size_t)r]);
      peer_arena.clear();
      refuse("a rank could not map a peer's mailbox arena (hipIpcOpenMemHandle)");
    }
    send_seq.assign((size_t)size, 0);
    recv_seq.assign((size_t)size, 0);
    tickets.alloc((size_t)k::IPC_MAX_TRANSFERS * 8);  // eight launches' worth: consecutive launches never share a slot
    MI_HIP(hipMemset(tickets.p, 0, (size_t)k::IPC_MAX_TRANSFERS * 8 * sizeof(unsigned)));
    error_flag.alloc(1);
    MI_HIP(hipMemset(error_flag.p, 0, sizeof(int)));
    MI_HIP(hipDeviceSynchronize());
    in.barrier();  // every arena is mapped everywhere before the first message
  }
  ~IpcExchangeComm() override {
    (void)hipDeviceSynchronize();
    for (int r = 0; r < (int)peer_arena.size(); r++)
      if (r != rank && peer_arena[(size_t)r]) (void)hipIpcCloseMemHandle(peer_arena[(size_t)r]);
    if (arena) (void)hipFree(arena);
  }
  const char *name() const override { return label.c_str(); }
  bool host_transport() const override { return inner->host_transport(); }
  // the inner products of the Krylov loops (1 to 8 doubles, sum): one launch, peer stores + rank-ordered sum
  // (MI_HYPRE_IPC_ALLREDUCE=0: through the wrapped communicator, like every other reduction)
  void allreduce_dev(void *buf, size_t count, CommDType t, CommOp op, hipStream_t s) override {
    static const bool on = !(getenv("MI_HYPRE_IPC_ALLREDUCE") && atoi(getenv("MI_HYPRE_IPC_ALLREDUCE")) == 0);
    if (!on || t != CommDType::F64 || op != CommOp::SUM || count == 0 || count > (size_t)k::IPC_AR_MAX || size > 16) {
      // (a rank with an expired wait contributes NaN here too: see k::ipc_allreduce_k)
      if (t == CommDType::F64 && op == CommOp::SUM && count > 0) k::ipc_poison((double *)buf, (int)count, error_flag.p, s);
      inner->allreduce_dev(buf, count, t, op, s);
      return;
    }
    k::IpcAllreduce a{};
    a.rank = rank;
    a.size = size;
    a.count = (int)count;
    a.seq = ++ar_seq;
    a.buf = (double *)buf;
    a.my_slots = ar_slots(arena);
    a.my_flags = ar_flags(arena);
    for (int r = 0; r < size; r++) {
      a.peer_slots[r] = ar_slots(peer_arena[(size_t)r]);
      a.peer_flags[r] = ar_flags(peer_arena[(size_t)r]);
    }
    k::ipc_allreduce(a, spin_limit, error_flag.p, s);
    n_allreduce_ipc++;
  }
  void allgather_dev(const void *send, void *recv, size_t bytes, hipStream_t s) override {
    inner->allgather_dev(send, recv, bytes, s);
  }
  void allreduce_host(void *buf, size_t count, CommDType t, CommOp op) override { inner->allreduce_host(buf, count, t, op); }
  void allgather_host(const void *send, void *recv, size_t bytes) override { inner->allgather_host(send, recv, bytes); }
  void exchange_host_fixed(const std::vector<PeerBuf> &sends, const std::vector<PeerBuf> &recvs) override {
    inner->exchange_host_fixed(sends, recvs);
  }
  void check_error(hipStream_t s) {
    int e = 0;
    d2h(&e, error_flag.p, sizeof(int), s);
    if (e) fail(1, "peer-store exchange: a neighbour's message did not arrive within the time limit (MI_HYPRE_IPC_TIMEOUT_MS)");
  }
  // messages larger than a slot travel in slot-sized parts, each with its own sequence number (both sides know the
  // sizes, so they agree on the parts).  NOTE: two messages between the same pair in ONE call would interleave their
  // parts differently on the two sides; a halo plan has one message per neighbour and direction (parcsr.cpp)
  void exchange_dev(const std::vector<PeerBuf> &sends, const std::vector<PeerBuf> &recvs, hipStream_t s) override {
    if (sends.empty() && recvs.empty()) return;
    k::IpcBatch B;
    B.n = 0;
    auto flush = [&]() {
      if (B.n == 0) return;
      k::ipc_exchange(B, spin_limit, error_flag.p, s);
      n_launch++;
      B.n = 0;
    };
    auto push = [&](const k::IpcTransfer &t) {
      // one launch holds every part it can; a ticket slot belongs to one transfer of the launch
      if (B.n == k::IPC_MAX_TRANSFERS) flush();
      B.t[B.n] = t;
      B.t[B.n].ticket = tickets.p + (size_t)(n_launch % 8) * k::IPC_MAX_TRANSFERS + B.n;  // (also when two streams carry exchanges)
      B.n++;
    };
    auto blocks_for = [](size_t bytes) { return bytes >= (1u << 20) ? 4 : bytes >= (1u << 18) ? 2 : 1; };
    // Order of the transfers: part q of every send, then part q of every receive, q = 0, 1, ...  A send of part
    // q + 2 waits for the peer's acknowledgement of part q, i.e. for a receive that stands EARLIER in the peer's own
    // list and needs nothing but this rank's send of part q: every wait points to a smaller q, so there is no cycle,
    // however the lists are cut into launches (all workgroups of a launch are resident together: <= 32 x 4).
    auto parts_of = [&](size_t bytes) { return bytes == 0 ? (size_t)1 : (bytes + slot_bytes - 1) / slot_bytes; };
    size_t max_parts = 0;
    for (const auto &b : sends) {
      MI_REQUIRE(b.peer >= 0 && b.peer < size && b.peer != rank, "peer-store exchange: bad peer");
      max_parts = std::max(max_parts, parts_of(b.bytes));
    }
    for (const auto &b : recvs) {
      MI_REQUIRE(b.peer >= 0 && b.peer < size && b.peer != rank, "peer-store exchange: bad peer");
      max_parts = std::max(max_parts, parts_of(b.bytes));
    }
    for (size_t q = 0; q < max_parts; q++) {
      for (const auto &b : sends) {
        if (q >= parts_of(b.bytes)) continue;
        const size_t off = q * slot_bytes, part = std::min(slot_bytes, b.bytes - off);
        const unsigned long long seq = ++send_seq[(size_t)b.peer];
        const int slot = (int)(seq & 1);
        k::IpcTransfer t{};
        t.kind = 0;
        t.nblocks = blocks_for(part);
        t.bytes = part;
        t.src = (const char *)b.ptr + off;
        t.dst = slot_ptr(peer_arena[(size_t)b.peer], rank, slot);
        t.wait_word = ack_word(arena, b.peer, slot);  // the peer acknowledges in MY arena
        t.wait_value = seq >= 2 ? seq - 2 : 0;
        t.post_word = flag_word(peer_arena[(size_t)b.peer], rank, slot);
        t.post_value = seq;
        push(t);
      }
      for (const auto &b : recvs) {
        if (q >= parts_of(b.bytes)) continue;
        const size_t off = q * slot_bytes, part = std::min(slot_bytes, b.bytes - off);
        const unsigned long long seq = ++recv_seq[(size_t)b.peer];
        const int slot = (int)(seq & 1);
        k::IpcTransfer t{};
        t.kind = 1;
        t.nblocks = blocks_for(part);
        t.bytes = part;
        t.src = slot_ptr(arena, b.peer, slot);
        t.dst = (char *)b.ptr + off;
        t.wait_word = flag_word(arena, b.peer, slot);
        t.wait_value = seq;
        t.post_word = ack_word(peer_arena[(size_t)b.peer], rank, slot);
        t.post_value = seq;
        push(t);
      }
    }
    flush();
  }
};
}  // namespace

// On refusal (mi::Error from the constructor, on every rank alike) `inner` is untouched: the caller keeps its transport.
std::unique_ptr<Comm> make_ipc_exchange_comm(std::unique_ptr<Comm> &inner, size_t slot_bytes) {
  if (inner->size == 1) return std::move(inner);
  std::unique_ptr<IpcExchangeComm> c(new IpcExchangeComm(*inner, slot_bytes));
  c->inner = std::move(inner);
  return std::unique_ptr<Comm>(c.release());
}
// Has a bounded wait of the transport expired on ANY rank?  Collective over the wrapped communicator (every rank of a
// peer-store communicator calls it at the same points: the end of every Setup / Solve, capi.cpp), so that all ranks
// see the same verdict and leave together.  0 for transports that cannot fail silently.
int comm_transport_verdict(Comm &c, hipStream_t s) {
  auto *ipc = dynamic_cast<IpcExchangeComm *>(&c);
  if (!ipc) return 0;
  int e = 0;
  d2h(&e, ipc->error_flag.p, sizeof(int), s);
  e = e ? 1 : 0;
  ipc->inner->allreduce_host(&e, 1, CommDType::I32, CommOp::MAX);
  return e;
}
bool comm_check_transport_error(Comm &c, hipStream_t s) {
  if (auto *ipc = dynamic_cast<IpcExchangeComm *>(&c)) {
    ipc->check_error(s);
    return true;
  }
  return false;
}

std::unique_ptr<Comm> make_self_comm() { return std::unique_ptr<Comm>(new SelfComm()); }
std::unique_ptr<Comm> make_rccl_comm(const void *id, int rank, int size) {
  return std::unique_ptr<Comm>(new RcclComm(id, rank, size));
}
void rccl_get_unique_id(void *out128) {
  ncclUniqueId id;
  MI_NCCL(rccl().GetUniqueId(&id));
  memcpy(out128, &id, sizeof(id));
}
std::unique_ptr<Comm> make_callback_comm(const CommCallbacks &cb, int rank, int size) {
  return std::unique_ptr<Comm>(new CallbackComm(cb, rank, size));
}

// ------------------------------------------------------------------ launcher-agnostic bootstrap
// One process per GPU started by any launcher that exports RANK / WORLD_SIZE /
// MASTER_ADDR / MASTER_PORT (torchrun does).  Rank 0 creates the ncclUniqueId and
// hands it to the other ranks over a short-lived TCP connection.
static int env_int(const char *a, const char *b, const char *c, int dflt) {
  for (const char *n : {a, b, c})
    if (n && getenv(n)) return atoi(getenv(n));
  return dflt;
}
static void send_all(int fd, const void *buf, size_t n) {
  const char *p = (const char *)buf;
  while (n) {
    ssize_t w = ::send(fd, p, n, 0);
    if (w <= 0) fail(1, "comm bootstrap: send failed");
    p += w;
    n -= (size_t)w;
  }
}
static void recv_all(int fd, void *buf, size_t n) {
  char *p = (char *)buf;
  while (n) {
    ssize_t r = ::recv(fd, p, n, 0);
    if (r <= 0) fail(1, "comm bootstrap: recv failed");
    p += r;
    n -= (size_t)r;
  }
}
// ------------------------------------------------------------------ TCP mesh (single node, host-staged)
// Every pair of ranks shares one socket; rank r listens on port base + r and connects to the lower ranks.  The three
// callbacks of the host-staged transport run over it: reductions and gathers through rank 0 (contributions added in
// rank order: identical bits on every rank), neighbour exchanges directly between the peers with poll()-driven
// progress (no ordering of sends and receives can deadlock).  A pair's stream is FIFO and both ends issue their
// common operations in the same program order, so messages need no tags; every message carries its length, which
// the receiver checks.
namespace {
struct TcpMesh {
  int rank = 0, size = 1;
  std::vector<int> fd;
  ~TcpMesh() {
    for (int f : fd)
      if (f >= 0) ::close(f);
  }
  void send_msg(int peer, const void *p, size_t n) {
    const uint64_t len = n;
    send_all(fd[(size_t)peer], &len, sizeof(len));
    if (n) send_all(fd[(size_t)peer], p, n);
  }
  void recv_msg(int peer, void *p, size_t n) {
    uint64_t len = 0;
    recv_all(fd[(size_t)peer], &len, sizeof(len));
    if (len != n) fail(1, "tcp transport: message of " + std::to_string(len) + " bytes where " + std::to_string(n) + " were expected");
    if (n) recv_all(fd[(size_t)peer], p, n);
  }
  template <class T>
  static void fold(T *acc, const T *v, size_t count, int op) {
    for (size_t k = 0; k < count; k++)
      acc[k] = op == (int)CommOp::SUM ? acc[k] + v[k] : (op == (int)CommOp::MIN ? std::min(acc[k], v[k]) : std::max(acc[k], v[k]));
  }
  void allreduce(void *buf, size_t count, int dtype, int op) {
    if (size == 1 || count == 0) return;
    const size_t bytes = count * dtype_size((CommDType)dtype);
    if (rank != 0) {
      send_msg(0, buf, bytes);
      recv_msg(0, buf, bytes);
      return;
    }
    std::vector<char> tmp(bytes);
    for (int r = 1; r < size; r++) {
      recv_msg(r, tmp.data(), bytes);
      switch ((CommDType)dtype) {
        case CommDType::F64: fold((double *)buf, (const double *)tmp.data(), count, op); break;
        case CommDType::I64: fold((long long *)buf, (const long long *)tmp.data(), count, op); break;
        case CommDType::I32: fold((int *)buf, (const int *)tmp.data(), count, op); break;
        default: fold((unsigned char *)buf, (const unsigned char *)tmp.data(), count, op); break;
      }
    }
    for (int r = 1; r < size; r++) send_msg(r, buf, bytes);
  }
  void allgather(const void *send, void *recv, size_t bytes) {
    char *out = (char *)recv;
    if (bytes) memcpy(out + (size_t)rank * bytes, send, bytes);
    if (size == 1) return;
    if (rank != 0) {
      send_msg(0, send, bytes);
      recv_msg(0, out, bytes * (size_t)size);
      return;
    }
    for (int r = 1; r < size; r++) recv_msg(r, out + (size_t)r * bytes, bytes);
    for (int r = 1; r < size; r++) send_msg(r, out, bytes * (size_t)size);
  }
  // one direction of one pair: header, then payload
  struct Op {
    int fd;
    uint64_t len;
    char *ptr;
    size_t hdr_done = 0, done = 0;
    bool finished() const { return hdr_done == sizeof(uint64_t) && done == len; }
  };
  void exchange(int nsend, const int *sp, void *const *sptr, const size_t *sby, int nrecv, const int *rp, void *const *rptr,
                const size_t *rby) {
    // per socket the operations run in list order; sockets progress independently
    std::vector<std::vector<Op>> out((size_t)size), in((size_t)size);
    std::vector<size_t> oi((size_t)size, 0), ii((size_t)size, 0);
    for (int k = 0; k < nsend; k++) out[(size_t)sp[k]].push_back({fd[(size_t)sp[k]], (uint64_t)sby[k], (char *)sptr[k]});
    for (int k = 0; k < nrecv; k++) in[(size_t)rp[k]].push_back({fd[(size_t)rp[k]], (uint64_t)rby[k], (char *)rptr[k]});
    std::vector<uint64_t> rhdr((size_t)size, 0);
    for (;;) {
      std::vector<pollfd> pf;
      std::vector<int> who;
      for (int r = 0; r < size; r++) {
        short ev = 0;
        if (oi[(size_t)r] < out[(size_t)r].size()) ev |= POLLOUT;
        if (ii[(size_t)r] < in[(size_t)r].size()) ev |= POLLIN;
        if (ev) {
          pf.push_back({fd[(size_t)r], ev, 0});
          who.push_back(r);
        }
      }
      if (pf.empty()) break;
      // (ranks reach an exchange at different times -- a peer may still be in a long host phase of the setup)
      static const int wait_ms = getenv("MI_HYPRE_TCP_TIMEOUT_MS") ? atoi(getenv("MI_HYPRE_TCP_TIMEOUT_MS")) : 1800000;
      if (::poll(pf.data(), (nfds_t)pf.size(), wait_ms) <= 0) fail(1, "tcp transport: neighbour exchange timed out");
      for (size_t q = 0; q < pf.size(); q++) {
        const int r = who[q];
        if ((pf[q].revents & (POLLERR | POLLHUP | POLLNVAL)) && !(pf[q].revents & POLLIN)) fail(1, "tcp transport: peer closed the connection");
        if ((pf[q].revents & POLLOUT) && oi[(size_t)r] < out[(size_t)r].size()) {
          Op &o = out[(size_t)r][oi[(size_t)r]];
          if (o.hdr_done < sizeof(uint64_t)) {
            const ssize_t w = ::send(o.fd, (const char *)&o.len + o.hdr_done, sizeof(uint64_t) - o.hdr_done, MSG_DONTWAIT | MSG_NOSIGNAL);
            if (w > 0) o.hdr_done += (size_t)w;
          } else if (o.done < o.len) {
            const ssize_t w = ::send(o.fd, o.ptr + o.done, o.len - o.done, MSG_DONTWAIT | MSG_NOSIGNAL);
            if (w > 0) o.done += (size_t)w;
          }
          if (o.finished()) oi[(size_t)r]++;
        }
        if ((pf[q].revents & POLLIN) && ii[(size_t)r] < in[(size_t)r].size()) {
          Op &o = in[(size_t)r][ii[(size_t)r]];
          if (o.hdr_done < sizeof(uint64_t)) {
            const ssize_t g = ::recv(o.fd, (char *)&rhdr[(size_t)r] + o.hdr_done, sizeof(uint64_t) - o.hdr_done, MSG_DONTWAIT);
            if (g == 0) fail(1, "tcp transport: peer closed the connection");
            if (g > 0) o.hdr_done += (size_t)g;
            if (o.hdr_done == sizeof(uint64_t) && rhdr[(size_t)r] != o.len)
              fail(1, "tcp transport: halo message of " + std::to_string(rhdr[(size_t)r]) + " bytes where " + std::to_string(o.len) + " were expected");
          } else if (o.done < o.len) {
            const ssize_t g = ::recv(o.fd, o.ptr + o.done, o.len - o.done, MSG_DONTWAIT);
            if (g == 0) fail(1, "tcp transport: peer closed the connection");
            if (g > 0) o.done += (size_t)g;
          }
          if (o.finished()) ii[(size_t)r]++;
        }
      }
    }
  }
};
void tcp_allreduce(void *c, void *buf, size_t count, int dtype, int op) { ((TcpMesh *)c)->allreduce(buf, count, dtype, op); }
void tcp_allgather(void *c, const void *send, void *recv, size_t bytes) { ((TcpMesh *)c)->allgather(send, recv, bytes); }
void tcp_exchange(void *c, int ns, const int *sp, void *const *sptr, const size_t *sby, int nr, const int *rp, void *const *rptr,
                  const size_t *rby) {
  ((TcpMesh *)c)->exchange(ns, sp, sptr, sby, nr, rp, rptr, rby);
}

std::unique_ptr<Comm> make_tcp_comm(int rank, int size, const char *addr, int base_port) {
  TcpMesh *m = new TcpMesh();  // lives as long as the process (the communicator holds plain callbacks)
  m->rank = rank, m->size = size;
  m->fd.assign((size_t)size, -1);
  int ls = ::socket(AF_INET, SOCK_STREAM, 0);
  if (ls < 0) fail(1, "tcp transport: socket");
  int one = 1;
  setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
  sockaddr_in sa{};
  sa.sin_family = AF_INET;
  sa.sin_addr.s_addr = htonl(INADDR_ANY);
  sa.sin_port = htons((uint16_t)(base_port + rank));
  if (::bind(ls, (sockaddr *)&sa, sizeof(sa)) != 0 || ::listen(ls, size) != 0) {
    ::close(ls);
    fail(1, "tcp transport: cannot listen on port " + std::to_string(base_port + rank));
  }
  for (int q = 0; q < rank; q++) {  // connect to the lower ranks
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(addr, std::to_string(base_port + q).c_str(), &hints, &res) != 0 || !res)
      fail(1, std::string("tcp transport: cannot resolve ") + addr);
    int f = -1;
    for (int attempt = 0; attempt < 600; attempt++) {
      f = ::socket(AF_INET, SOCK_STREAM, 0);
      if (f >= 0 && ::connect(f, res->ai_addr, res->ai_addrlen) == 0) break;
      if (f >= 0) ::close(f);
      f = -1;
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    freeaddrinfo(res);
    if (f < 0) fail(1, "tcp transport: cannot reach rank " + std::to_string(q));
    setsockopt(f, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    const int me = rank;
    send_all(f, &me, sizeof(me));
    m->fd[(size_t)q] = f;
  }
  for (int k = rank + 1; k < size; k++) {  // the higher ranks connect to me, in any order
    int f = ::accept(ls, nullptr, nullptr);
    if (f < 0) fail(1, "tcp transport: accept");
    setsockopt(f, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    int who = -1;
    recv_all(f, &who, sizeof(who));
    if (who <= rank || who >= size || m->fd[(size_t)who] >= 0) fail(1, "tcp transport: unexpected peer");
    m->fd[(size_t)who] = f;
  }
  ::close(ls);
  CommCallbacks cb;
  cb.ctx = m;
  cb.allreduce = tcp_allreduce;
  cb.allgather = tcp_allgather;
  cb.exchange = tcp_exchange;
  return make_callback_comm(cb, rank, size);
}
}  // namespace

std::unique_ptr<Comm> make_comm_from_env() {
  const int size = env_int("WORLD_SIZE", "OMPI_COMM_WORLD_SIZE", "PMI_SIZE", 1);
  const int rank = env_int("RANK", "OMPI_COMM_WORLD_RANK", "PMI_RANK", 0);
  if (size <= 1) return make_self_comm();
  const char *addr = getenv("MASTER_ADDR") ? getenv("MASTER_ADDR") : "127.0.0.1";
  int port = getenv("MI_HYPRE_PORT") ? atoi(getenv("MI_HYPRE_PORT"))
                                     : (getenv("MASTER_PORT") ? atoi(getenv("MASTER_PORT")) + 17 : 29517);
  if (getenv("MI_HYPRE_TRANSPORT") && std::string(getenv("MI_HYPRE_TRANSPORT")) == "tcp") return make_tcp_comm(rank, size, addr, port);
  unsigned char id[128];
  if (rank == 0) {
    rccl_get_unique_id(id);
    int ls = ::socket(AF_INET, SOCK_STREAM, 0);
    if (ls < 0) fail(1, "comm bootstrap: socket");
    int one = 1;
    setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in sa{};
    sa.sin_family = AF_INET;
    sa.sin_addr.s_addr = htonl(INADDR_ANY);
    sa.sin_port = htons((uint16_t)port);
    if (::bind(ls, (sockaddr *)&sa, sizeof(sa)) != 0 || ::listen(ls, size) != 0) {
      ::close(ls);
      fail(1, "comm bootstrap: cannot listen on port " + std::to_string(port));
    }
    for (int k = 1; k < size; k++) {
      int fd = ::accept(ls, nullptr, nullptr);
      if (fd < 0) fail(1, "comm bootstrap: accept");
      send_all(fd, id, sizeof(id));
      ::close(fd);
    }
    ::close(ls);
  } else {
    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    if (getaddrinfo(addr, std::to_string(port).c_str(), &hints, &res) != 0 || !res)
      fail(1, std::string("comm bootstrap: cannot resolve ") + addr);
    int fd = -1;
    for (int attempt = 0; attempt < 600; attempt++) {
      fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0) break;
      if (fd >= 0) ::close(fd);
      fd = -1;
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    freeaddrinfo(res);
    if (fd < 0) fail(1, "comm bootstrap: cannot reach rank 0");
    recv_all(fd, id, sizeof(id));
    ::close(fd);
  }
  return make_rccl_comm(id, rank, size);
}

}  // namespace mi
