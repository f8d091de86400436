// ParCSR matrix / ParVector objects: host structure, device mirror, halo plan.
// Layout follows HYPRE's ParCSR model: a contiguous block-row
// partition, diag block with local int32 columns, offd block with compressed
// columns + sorted col_map_offd of global ids.
#pragma once
#include "mi_internal.hpp"

namespace mi {

struct ParVector {
  gidx start = 0, end = 0;  // global range [start, end)
  int n = 0;
  int ncomp = 1, cur = 0;
  DVec<double> d;  // ncomp * n, component-major
  double *data() { return d.p + (size_t)cur * n; }
  const double *data() const { return d.p + (size_t)cur * n; }
  // multivector view (HYPRE_IJVectorSetNumComponents > 1): all components, component-major.  Solvers work on the
  // whole multivector, as HYPRE's do: the block system diag(A, .., A) with inner products over every component
  double *all() { return d.p; }
  const double *all() const { return d.p; }
  int len() const { return n * ncomp; }
  void init(gidx s, gidx e, int nc);
};

// neighbour exchange plan of one matrix (hypre_ParCSRCommPkg)
struct HaloPlan {
  std::vector<int> recv_peers, recv_starts;  // offsets into x_ext, size peers+1
  std::vector<int> send_peers, send_starts;  // offsets into send_map
  std::vector<int> send_map;                 // local row ids
  DVec<int> d_send_map;
  DVec<double> d_send_buf, d_xext;
  int nsend() const { return (int)send_map.size(); }
};

struct ParCSR {
  gidx row_start = 0, row_end = 0;  // [start, end)
  int nrows = 0;
  std::vector<gidx> row_starts;  // size+1 global row partition (square matrices: also the column partition)
  std::vector<gidx> col_starts;  // size+1 global COLUMN partition of a rectangular operator (P, R); empty = row_starts
  const std::vector<gidx> &col_partition() const { return col_starts.empty() ? row_starts : col_starts; }
  HostCSR diag, offd;
  std::vector<gidx> col_map_offd;
  HaloPlan halo;
  DevCSR d_diag;
  DevOffd d_offd;
  DVec<double> d_offc;  // per-row halo contribution scratch (zero outside halo rows)
  bool on_device = false;
  // unique per assembly (0 = never assembled): tells a re-assembled or re-allocated matrix from the one a
  // preconditioner was set up on
  unsigned long long assembly_stamp = 0;
  // A level built by the device setup keeps its diag block on the device only: the host arrays of `diag`
  // are filled on demand (BoomerAMG::ensure_host); nrows / ncols of `diag` are always valid.
  bool host_diag_stale = false;
  int64_t dev_diag_nnz = 0;
  int64_t diag_nnz() const { return host_diag_stale ? dev_diag_nnz : diag.nnz(); }
  gidx global_rows() const { return row_starts.empty() ? nrows : row_starts.back(); }

  // build the halo plan from col_map_offd (collective, host only)
  void build_halo_plan(Comm &comm);
  // Operators that live on the device and have no halo block (one rank: every level of a 512^3 hierarchy) do not
  // carry the halo block's row pointers on the host -- n + 1 zeros of 8 bytes each per operator, ~0.25 s of page
  // faults per 134 M rows in round 3; whoever brings the diag block to the host calls this first
  void ensure_offd_rows() {
    if (offd.ia.size() != (size_t)nrows + 1 && offd.nnz() == 0) {
      offd.nrows = nrows;
      offd.ia.assign((size_t)nrows + 1, 0);
    }
  }
  // mirror matrix + plan to the device (needs a GPU)
  void to_device();
  // the same without the diag block (already built on the device: d_diag is set)
  void to_device_halo();
  void finalize(Comm &comm) {
    build_halo_plan(comm);
    to_device();
  }
  // x_ext <- halo values of x (pack + neighbour exchange), enqueued on stream
  // (optional composite source: rows < split from x, rows >= split from x_hi)
  void halo_exchange(Comm &comm, const double *x, hipStream_t s, const double *x_hi = nullptr, int split = 0);
  // its two halves: gather the send buffer / neighbour transfer into x_ext (matvec runs the transfer on a side
  // stream, beside the diag-block product)
  void halo_pack(const double *x, hipStream_t s, const double *x_hi, int split);
  void halo_transfer(Comm &comm, hipStream_t s);
  // host-side halo exchange of an arbitrary per-row int array (setup only)
  std::vector<int> halo_exchange_host_int(Comm &comm, const std::vector<int> &local) const;
  // y = alpha*A*x + beta*b
  // diag_op (optional): another operator in place of the diag block (same rows and columns; BoomerAMG's residual
  // after a zero-guess sweep leaves out entries whose product it already has); the halo block is always this one's
  // b_lo / b_split (optional): rows < b_split read their b entry from b_lo (composite right-hand side)
  void matvec(Comm &comm, double alpha, const double *x, double beta, const double *b, double *y, hipStream_t s,
              int prof = -1, const DevCSR *diag_op = nullptr, const double *b_lo = nullptr, int b_split = 0);
  // the same with the halo values already in halo.d_xext (no exchange)
  void matvec_ext_ready(double alpha, const double *x, double beta, const double *b, double *y, hipStream_t s);
  // offc[halo rows] = A_offd * x_ext(x)   (used by the smoothers)
  const double *offd_contrib(Comm &comm, const double *x, hipStream_t s, const double *x_hi = nullptr, int split = 0);
};

// global dot product: local two-stage reduction + one all-reduce; result stays on
// the device in out_dev (the caller decides when to read it)
void par_dot(Comm &comm, const double *x, const double *y, int n, double *out_dev, hipStream_t s);
// blocking variant returning the value
double par_dot_host(Comm &comm, const double *x, const double *y, int n, hipStream_t s);

// IJ assembly: COO triples (global ids) -> ParCSR
struct IJEntryBatch {
  std::vector<gidx> rows, cols;
  std::vector<double> vals;
  bool add = false;
};
void assemble_parcsr(Comm &comm, gidx ilower, gidx iupper, gidx jlower, gidx jupper,
                     std::vector<IJEntryBatch> &batches, ParCSR &out);

bool is_device_pointer(const void *p);

// The solve format keeps 32-bit local row ids per rank; a rank's DIAGONAL block may hold any number of entries (64-bit
// tile bases, mi_internal.hpp DevCSR -- e.g. the reference's 27-point operator at 512^3, 3.6e9 entries on one rank,
// /root/reference/src/laplace_3d_weak_scaling.hpp:558,600); the halo block and a single row keep 32-bit offsets.
// Anything beyond that is refused with HYPRE_ERROR_ARG and a message, never wrapped around.
constexpr int64_t MAX_BLOCK_ENTRIES = 2147483000LL;
void require_int32_block(int64_t nrows, int64_t nnz, const char *what);

}  // namespace mi
