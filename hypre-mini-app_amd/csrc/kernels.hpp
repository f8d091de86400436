// Launchers of the hand-written gfx950 kernels (kernels.hip).  All launches are
// asynchronous on the given stream; pointers are device pointers.
#pragma once
#include "mi_internal.hpp"

namespace mi {
namespace k {

constexpr int SPMV_BLOCK = 256;   // threads per workgroup (4 wave64)
constexpr int SPMV_TILE = 2048;   // LDS-staged products per workgroup
// Operators with long rows (coarse Galerkin levels with 100-150 entries per row) get tiles of twice the entries,
// run by twice the threads (same work per thread, 32 / 40 KB of LDS = still 32 waves per CU): an 8-row chunk of such
// rows holds 800-1500 entries, so 2048-entry tiles close after one or two chunks, 2/3 full, and share few columns.
// 512^3: level 3 / 4 / 5 relaxation 1.35 / 0.47 / 0.126 -> 1.23 / 0.42 / 0.103 ms; a level with 77 entries per row
// LOSES 6 % (its 2048-entry tiles are 90 % full already), hence the threshold
constexpr int SPMV_BLOCK_WIDE = 512;
constexpr int SPMV_TILE_WIDE = 4096;
constexpr int RED_MAX_BLOCKS = 2048;
constexpr int MASS_NV = 8;        // vectors per pass of the block inner product / block update
constexpr int GS_BLOCK = 256;     // chunks (lanes) per workgroup
constexpr int GS_MAX_CHUNK = 32;

// kernel classes that can be timed with HIP events (profile.cpp)
// 0..3: the classes bench.py times; then per AMG level l < PROF_LEVELS: the residual SpMV of the cycle, the
// relaxation passes, restriction and prolongation (profiles/roofline_table.py)
constexpr int PROF_LEVELS = 16;
enum ProfId {
  PROF_NONE = -1,
  PROF_SPMV_L0 = 0,
  PROF_RELAX_L0 = 1,
  PROF_DOT = 2,
  PROF_AXPY = 3,
  PROF_LVL_RESID = 4,
  PROF_LVL_RELAX = PROF_LVL_RESID + PROF_LEVELS,
  PROF_LVL_RESTRICT = PROF_LVL_RELAX + PROF_LEVELS,
  PROF_LVL_PROLONG = PROF_LVL_RESTRICT + PROF_LEVELS,
  PROF_LVL_RELAX0 = PROF_LVL_PROLONG + PROF_LEVELS,  // first sweep on a zero guess (runs on the sub-operator)
  PROF_COUNT = PROF_LVL_RELAX0 + PROF_LEVELS
};
inline int prof_level(int base, int level) { return level < PROF_LEVELS ? base + level : PROF_NONE; }

// Row-block (tile) schedule of the SpMV / tile Gauss-Seidel kernels: <= row_cap rows and < tile_entries entries per
// tile, or exactly one longer row; greedy, whole 8-row chunks while they fit.  Round 4: tiles never cross a multiple of
// TILE_SUPER_ROWS rows, so the schedule of every such super-block is independent of the others -- the device builds it
// with one thread per super-block (sk::to_solve_format) instead of the host walking the row pointers of the whole
// operator (1 GB of them per operator at 512^3, copied over PCIe first); the price is one underfull tile per 8192 rows.
// tile_end(): one step of the schedule, shared by the host routine and the device kernel.
constexpr int TILE_SUPER_ROWS = 8192;
// (ia: anything indexable by a row that yields its first entry -- the row pointers themselves, or the device kernel's
// staged copy of a super-block's chunk boundaries)
template <class IA>
__host__ __device__ inline int tile_end(int r, int limit, const IA &ia, int row_cap, int block_rows, int tile_entries,
                                        bool &aligned) {
  const long long start = (long long)ia[r];
  int e = r;
  if ((r & 7) == 0 && row_cap <= block_rows) {  // whole chunks while they fit (SpMV-only operators: any row)
    while (e < limit && e - r < row_cap) {
      const int e2 = (e + 8 < limit) ? e + 8 : limit;
      if ((long long)ia[e2] - start > tile_entries - 1) break;
      e = e2;
    }
  }
  if (e == r) {  // not even one chunk fits (or an unaligned start after such a chunk): row granularity
    aligned = false;
    // keep one slot of slack for the aligned-pair start
    while (e < limit && e - r < row_cap && (long long)ia[e + 1] - start <= tile_entries - 1) e++;
    if (e == r) e = r + 1;  // a single row longer than the tile
  }
  return e;
}
// The same step by bisection (the device schedule, sk::to_solve_format): the entry counts ascend with the row, so "as many
// whole chunks / rows as fit" is the last candidate end whose count fits -- ~10 dependent reads per tile instead of one
// per chunk or per row (SpMV-only operators are scheduled row by row: 8192 dependent reads per super-block, 8 ms per
// launch at 512^3).  tests/test_gpu_kernels.py compares the two on random row lengths (HYPRE_MI_TileScheduleCheck).
template <class IA>
__host__ __device__ inline int tile_end_bisect(int r, int limit, const IA &ia, int row_cap, int block_rows, int tile_entries,
                                               bool &aligned) {
  const long long start = (long long)ia[r];
  const long long room = (long long)tile_entries - 1;
  int e = r;
  if ((r & 7) == 0 && row_cap <= block_rows) {
    // candidate ends e_m = min(r + 8 m, limit), m = 1 .. M (the iterations the loop above can make)
    const int by_cap = (row_cap + 7) / 8, by_rows = (limit - r + 7) / 8;
    const int M = by_cap < by_rows ? by_cap : by_rows;
    int lo = 0, hi = M;  // the answer m lies in [lo, hi]; m = 0: not even one chunk
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      const int em = (r + 8 * mid < limit) ? r + 8 * mid : limit;
      if ((long long)ia[em] - start <= room)
        lo = mid;
      else
        hi = mid - 1;
    }
    if (lo > 0) e = (r + 8 * lo < limit) ? r + 8 * lo : limit;
  }
  if (e == r) {
    aligned = false;
    const int top = ((long long)r + row_cap < (long long)limit) ? r + row_cap : limit;
    int lo = r, hi = top;  // the last e in [r, top] with ia[e] - start <= room (e = r always qualifies)
    while (lo < hi) {
      const int mid = (int)(((long long)lo + hi + 1) >> 1);
      if ((long long)ia[mid] - start <= room)
        lo = mid;
      else
        hi = mid - 1;
    }
    e = lo;
    if (e == r) e = r + 1;
  }
  return e;
}
std::vector<int> build_row_blocks(int nrows, const int64_t *ia, bool *chunk_aligned = nullptr,
                                  int row_cap = SPMV_BLOCK, int tile_entries = SPMV_TILE);
// tile size of an operator with nnz entries in nrows rows (MI_HYPRE_WIDE_TILE_MIN_ROWLEN, default 100 entries per row; 0 = never wide)
int choose_tile_entries(int64_t nnz, int nrows);
constexpr int SPMV_ONLY_ROW_CAP = 4 * SPMV_BLOCK;  // rows per tile of operators no Gauss-Seidel kernel sweeps
constexpr int XC_ID_MASK = 0xFFF;   // block-local column id inside an lcol entry (at most SPMV_TILE_WIDE = 4096 ids)
constexpr int XC_INCH = 0x8000;     // the column lies in the row's own 8-row chunk ...
constexpr int XC_OFF_SHIFT = 12;    // ... at this offset (3 bits)

// per-tile descriptors of the SpMV / tile-GS kernels (row range, entry range, column-list range: 8 ints per
// tile); call after rb, ia and (x cache) uptr are in place
// ia64: the operator's 64-bit row pointers (device; the descriptors carry full 64-bit tile bases)
void build_tile_desc(DevCSR &A, const long long *ia64, hipStream_t s);
// value dictionary of an operator with at most 256 distinct values (called by build_tile_desc)
void build_value_dictionary(DevCSR &A, hipStream_t s);
// MI_HYPRE_VALUE_DICT / HYPRE_MI_SetValueDictionary: applies to operators put into the solve format afterwards
bool value_dictionary_enabled();
void set_value_dictionary(bool on);
// y = alpha*A*x + beta*b   (b may alias y)
// b_lo (optional): rows < b_split take their b entry from b_lo instead of b
void spmv(const DevCSR &A, const double *x, double alpha, double beta, const double *b, double *y, hipStream_t s,
          int prof = PROF_NONE, const double *b_lo = nullptr, int b_split = 0);
// y[rows[k]] += alpha * (B*xext)[k]
void spmv_offd_add(const DevOffd &B, const double *xext, double alpha, double *y, hipStream_t s);
// out[rows[k]] = (B*xext)[k]   (other entries of out untouched)
void spmv_offd_set(const DevOffd &B, const double *xext, double *out, hipStream_t s);

// masked Jacobi family: u_new = u_old + w*(f - offc - A*u_old)/d on selected rows, copy elsewhere
void jacobi(const DevCSR &A, const double *u_old, double *u_new, const double *f, const double *offc, const double *d,
            const signed char *cf, int points, double w, hipStream_t s, int prof = PROF_NONE);
// hybrid Gauss-Seidel family: chunks of `chunk` consecutive rows are swept
// sequentially (forward and/or backward), chunks see each other's pre-sweep
// values.  Only the chunks that intersect [row_begin, row_end) are swept (a C or
// an F pass of a C-first ordered level).  Pre-sweep values are read from u_lo
// for rows < split and from u_hi for rows >= split (the second pass of a C/F
// pair reads the first pass's output without a copy); the swept chunks' rows are
// written to out.  zero_from: the caller guarantees that every pre-sweep value with index >= zero_from is
// zero (first sweep on a zero guess); kernels may then skip those gathers -- same result, less traffic.
constexpr int GS_NO_ZEROS = 0x7fffffff;
// whether gs_hybrid sweeps A tile by tile (A.rb_host boundaries) rather than chunk by chunk: a caller that
// splits one pass into several launches must cut at the unit the kernel writes back
bool gs_uses_tiles(const DevCSR &A, int chunk);
// whether a sweep with zero_from == 0 leaves the pre-sweep vector unread (its zero-fill can then be skipped)
bool gs_ignores_zero_vector(const DevCSR &A, int chunk);
void gs_hybrid(const DevCSR &A, const double *u_lo, const double *u_hi, int split, double *out, const double *f,
               const double *offc, const double *d, const signed char *cf, int points, int chunk, bool fwd, bool bwd,
               double w, int row_begin, int row_end, hipStream_t s, int prof = PROF_NONE, int zero_from = GS_NO_ZEROS,
               double *tout = nullptr, int t_from = 0);
// tout (tile kernel only -- check gs_uses_tiles): every swept row i >= t_from also stores f[i] minus its
// out-of-chunk sum there

// BLAS-1
void dot(const double *x, const double *y, int n, double *out_dev, hipStream_t s);  // local sum, no collective
// fused MGS step: y += scale*(*alpha_dev)*xa, then out_dev = <xd, y> (xd == nullptr: <y, y>), local sum
void axpy_dot(const double *alpha_dev, double scale, const double *xa, double *y, const double *xd, int n,
              double *out_dev, hipStream_t s);
// block Gram-Schmidt (COGMRES): out_dev[j] = <vecs[j], w> for j < m (local sums), and w += scale * sum_j coef_dev[j] vecs[j]
void mass_dot(const double *const *vecs, int m, const double *w, int n, double *out_dev, hipStream_t s);
void mass_axpy(const double *const *vecs, int m, const double *coef_dev, double scale, double *w, int n, hipStream_t s);
// w = coef[0] vecs[0] (init) or w += coef[0] vecs[0]; then w += coef[j] vecs[j], j = 1 .. m-1 in this order (host
// coefficients); MASS_NV vectors per pass
void lin_comb(const double *const *vecs, const double *coef_host, int m, bool init, double *w, int n, hipStream_t s);
void axpy(double alpha, const double *x, double *y, int n, hipStream_t s);
void axpy_dev(const double *alpha_dev, double scale, const double *x, double *y, int n, hipStream_t s);
void scale(double alpha, double *x, int n, hipStream_t s);
// scale_inv_sqrt_dev + the `count` doubles at `slots` and then `seq` posted into host memory (see scale_post_k)
void scale_inv_sqrt_post(const double *sumsq_dev, double *x, int n, const double *slots, int count, double *host_out,
                         unsigned long long *host_flag, unsigned long long seq, hipStream_t s);
void scale_inv_sqrt_dev(const double *sumsq_dev, double *x, int n, hipStream_t s);
void load_device_code(hipStream_t s);
void fill(double *x, int n, double v, hipStream_t s);
void copy(const double *x, double *y, int n, hipStream_t s);
void gather(const double *x, const int *map, double *out, int n, hipStream_t s);
// out[i] = (map[i] < split ? lo : hi)[map[i]]
void gather2(const double *lo, const double *hi, int split, const int *map, double *out, int n, hipStream_t s);
// two-stage Gauss-Seidel pieces (relax types 11 / 12): z = r / d, u += z;  zout = (L zin) / d, u += sign * zout
void two_stage_first(const double *r, const double *d, double *z, double *u, int n, hipStream_t s);
void two_stage_lower(const DevCSR &A, const double *d, const double *zin, double sign, double *zout, double *u,
                     hipStream_t s);
// u = M f, M dense n x m row-major
void dense_matvec(const double *M, const double *f, double *u, int n, int m, hipStream_t s);
// u = sum_j f[j] * Mt[j][:], Mt n x n row-major (row j = column j of the map); fixed summation order
void dense_matvec_t(const double *Mt, const double *f, double *u, int n, hipStream_t s);

// ---- peer-store neighbour exchange (comm.cpp IpcExchangeComm): one launch moves every message of a halo update.
// A SEND copies local data into the mailbox slot the peer holds for this rank (an IPC-mapped pointer: stores over
// xGMI) and then publishes the message's sequence number in the peer's flag word; a RECV waits for its flag word,
// copies its own mailbox slot to the destination and acknowledges in the peer's ack word, which is what the
// sender's NEXT use of that slot waits for.  Every wait is bounded (spin_limit ticks of the 100 MHz wall clock):
// a peer that never arrives sets *error_flag instead of hanging the GPU.
struct IpcTransfer {
  int kind;                      // 0 send, 1 recv
  int nblocks;                   // workgroups that share the copy
  unsigned long long bytes;
  const void *src;               // send: local data; recv: my mailbox slot
  void *dst;                     // send: the peer's mailbox slot; recv: local destination
  unsigned long long *wait_word; // send: my ack word for this slot (peer wrote it); recv: my flag word
  unsigned long long wait_value; // proceed when *wait_word >= wait_value
  unsigned long long *post_word; // send: the peer's flag word; recv: the peer's ack word
  unsigned long long post_value;
  unsigned *ticket;              // per-transfer counter of finished workgroups (self-resetting)
};
constexpr int IPC_MAX_TRANSFERS = 32;
struct IpcBatch {
  IpcTransfer t[IPC_MAX_TRANSFERS];
  int n;
};
void ipc_exchange(const IpcBatch &b, unsigned long long spin_limit, int *error_flag, hipStream_t s);
// Peer-store all-reduce (sum) of up to IPC_AR_MAX doubles: every rank stores its values into every peer's mailbox
// (slot [parity][rank]) and publishes the reduction's number; then it waits for all peers' numbers and adds the
// size contributions IN RANK ORDER -- the same order on every rank, so all ranks get the same bits.  Two parities:
// nobody can start reduction k + 2 before every rank has finished reduction k (it needs their k + 1 values first).
constexpr int IPC_AR_MAX = 8;
struct IpcAllreduce {
  int rank, size, count;
  unsigned long long seq;            // number of this reduction (1, 2, ...)
  double *buf;                       // in: this rank's values; out: the sums
  double *my_slots;                  // my arena: [2][size][IPC_AR_MAX]
  unsigned long long *my_flags;      // my arena: [2][size]
  double *peer_slots[16];            // every rank's slots / flags as mapped here (self: my own)
  unsigned long long *peer_flags[16];
};
void ipc_allreduce(const IpcAllreduce &a, unsigned long long spin_limit, int *error_flag, hipStream_t s);
// buf[0..count) = NaN when the transport's error flag is up (reductions that travel on the wrapped communicator)
void ipc_poison(double *buf, int count, const int *error_flag, hipStream_t s);

// IJ helpers
void scatter_set(double *x, const int *idx, const double *vals, int n, hipStream_t s);
// column-by-column tabulation inside a captured graph (see tab_unit_k / tab_store_k)
void bytes_differ(const unsigned char *p, size_t n, unsigned char tag, unsigned long long *out3, hipStream_t s);
void fill_bytes(unsigned char *p, size_t n, unsigned char tag, hipStream_t s);
void tab_unit(double *e, const int *col, hipStream_t s);
void tab_store(double *Bt, const double *u, int n, int *col, hipStream_t s);
void scatter_add(double *x, const int *idx, const double *vals, int n, hipStream_t s);

}  // namespace k
}  // namespace mi
