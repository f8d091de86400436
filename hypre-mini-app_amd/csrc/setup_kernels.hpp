// Device kernels of the AMG setup phase (hypre_BoomerAMGSetup side of
// src/HypreSystem.cpp:692): sparse products of the Galerkin operator, transposes
// and the C-first renumbering.  Every kernel reproduces the host/oracle
// arithmetic bit for bit (same accumulation order, no fused multiply-add): this
// file is compiled with -ffp-contract=off.
#pragma once
#include "mi_internal.hpp"

namespace mi {
namespace sk {

// device CSR of the setup phase: 64-bit row pointers, columns ascending inside a row
struct DCsr {
  int nrows = 0, ncols = 0;
  int64_t nnz = 0;
  DVec<long long> ia;
  DVec<int> ja;
  DVec<double> a;
  void upload(const HostCSR &h, hipStream_t s);
  void download(HostCSR &h, hipStream_t s) const;
  void release() {
    ia.release();
    ja.release();
    a.release();
    nrows = ncols = 0;
    nnz = 0;
  }
};

// strength-of-connection graph of a single-rank operator (no halo block): row i keeps column j != i iff
// a_ij < theta * min_k a_ik (a_ii >= 0; mirrored for a_ii < 0); rows with |sum_j a_ij| > max_row_sum |a_ii|
// keep nothing.  S has no values (S.a stays empty).
void strength(const DCsr &A, double theta, double max_row_sum, DCsr &S, hipStream_t s);

// PMIS on the graph S: measure = |S^T row| + Park-Miller(seed) drawn in row order (hypre_Rand; element i is
// computed directly as seed * 16807^(i+1) mod 2^31-1).  cf: +1 C, -1 F, -3 F without strong connections.
void pmis(const DCsr &S, int seed, DVec<int> &cf, hipStream_t s);

// Interpolation (interp_type 6 extended+i or 0 classical modified) with truncation to pmax entries /
// trunc_factor, rows of A and S in ascending column order.  cf is updated like the host code does (-3 -> -1).
// Returns false -- and builds nothing -- when a row's interpolatory set may exceed the kernels' LDS capacity
// (the caller then runs the host routine for this level).  nc = number of C points.
bool interp(const DCsr &A, const DCsr &S, DVec<int> &cf, int interp_type, double trunc_factor, int pmax, DCsr &P,
            int &nc, hipStream_t s);

// C = A * B.  Rows of B must have ascending columns.  Entry (i, j) is the sum of
// a_ik * b_kj taken in the stored order of A's row i (first product assigned,
// the others added one by one) -- exactly host_spgemm (amg_setup.cpp) and the
// oracle's ocsr_matmul; output columns ascending.
void spgemm(const DCsr &A, const DCsr &B, DCsr &C, hipStream_t s);

// Non-Galerkin sparsification of a square operator (amg_setup.cpp sparsify_non_galerkin, the oracle's function of the
// same name): with m_i = max_{j != i} |a_ij|, off-diagonal entries with |a_ij| < tol * min(m_i, m_j) are dropped and
// added to the row's diagonal in stored order.  In place (A is replaced).
// N > 1 (a rank's rows in an extended column space, diagonal of row i in column i + row0): `maxima` holds m for EVERY
// column -- the own rows' from non_galerkin_row_maxima, the remote ones fetched from their owners by the caller.
void sparsify_non_galerkin(DCsr &A, double tol, hipStream_t s, int row0 = 0, const double *maxima = nullptr);
// m[row0 + i] = max_{j != row0 + i} |a_ij| for the rows of A
void non_galerkin_row_maxima(const DCsr &A, int row0, double *m, hipStream_t s);

// T = A^T with ascending columns in every row (entries of one output row keep
// the order of A's rows)
void transpose(const DCsr &A, DCsr &T, hipStream_t s);

// B = rows of A taken in `perm` order (perm[new] = old; null = identity) with
// columns mapped through colpos (null = identity) and re-sorted ascending
void permute(const DCsr &A, const int *perm, const int *colpos, DCsr &B, hipStream_t s);

// B = the nout rows rows[0..nout) of A (any subset, any order) with columns mapped through colpos (null =
// identity) and re-sorted ascending -- this rank's slice of a global level in the replicated multi-rank setup
void extract_rows(const DCsr &A, const int *rows, int nout, const int *colpos, DCsr &B, hipStream_t s);

// Solve-phase format of a setup-phase matrix, built on the device: 32-bit row pointers, the row-block
// schedule and x cache of the SpMV (DevCSR::upload builds the same from host arrays).  src's column and
// value arrays are MOVED into dst; only the row pointers travel to the host (for the greedy block schedule).
void to_solve_format(DCsr &src, DevCSR &dst, hipStream_t s);
// its tile schedule alone (k::build_row_blocks on the device, one thread per super-block, k::tile_end_bisect): rb on the
// device, blocks on the host, aligned = all tiles start on a multiple of 8 rows and were cut at whole chunks
void tile_schedule_device(int n, const long long *ia, int row_cap, int tile_entries, DVec<int> &rb, std::vector<int> &blocks,
                          bool &aligned, hipStream_t s);
// the host buffer to_solve_format keeps between calls goes back to the system (end of a setup)
void release_host_scratch();
// launches an empty kernel of this translation unit: its device code is loaded now (HYPRE_Init) instead of at the first setup
void load_device_code(hipStream_t s);
// The part of a C-first ordered square block (C points = indices < nc) that a FIRST relaxation sweep on a
// zero guess can touch: every row keeps the entries inside its own chunk of `chunk` rows, F rows also their C
// columns (written by the C pass that precedes the F pass).  Everything else multiplies zeros.  Columns stay
// ascending; Z goes through to_solve_format like any operator.
// mode 1: the operator of the residual that follows that sweep -- F rows from the first chunk boundary >= nc on
// lose their C columns (the F pass hands over f - A_FC u_C for them), every other row stays whole.
void zero_guess_operator(const DevCSR &A, int nc, int chunk, DCsr &Z, hipStream_t s, int mode = 0);
// setup-phase copy (64-bit row pointers) of an operator that lives in the solve format (device-to-device)
void from_solve_format(const DevCSR &src, DCsr &dst, hipStream_t s);
// Rounds of the internal locality numbering (amg_setup.cpp locality_order: graph Voronoi cells) on the device:
// labels start as the seeds' ranks (seeds ascending), -2 for excluded rows, -1 elsewhere; in every round an
// unlabelled row takes the smallest label among its neighbours of the same segment (row >> segshift) labelled in the
// previous round.  Stops when a round
// changes nothing or after max_rounds; label_host gets the result (-1 = never reached).  Returns the rounds run.
int locality_labels(const DCsr &A, const int *seeds_host, int nseeds, const unsigned char *exclude_host, int segshift,
                    int max_rounds, std::vector<int> &label_host, hipStream_t s);
// The whole internal numbering on the device (one rank, no excluded rows): order[new] = old, exactly hs::locality_order's
// result.  false (nothing usable in `order`) when more than a handful of cells outgrow the in-LDS sort.
bool locality_order_device(const DCsr &A, int segshift, int cluster, int max_rounds, DVec<int> &order, int &nseeds,
                           int &rounds, hipStream_t s);
// out[i] = (signed char)in[i]
void ints_to_i8(const int *in, long long n, signed char *out, hipStream_t s);
// pos[order[q]] = q
void invert_permutation(const int *order, int n, int *pos, hipStream_t s);
// back to host arrays (lazy host copies for the inspection API)
void solve_format_to_host(const DevCSR &src, HostCSR &h, hipStream_t s);

// diagonal, l1 norm of the hybrid-GS chunks (option 4, C/F aware) and full l1 norm per row (level_norms in
// amg_setup.cpp) of a single-rank operator; cf may be null
// halo (optional, N > 1): the level's halo block with full-length row pointers and the C/F type of its columns --
// its entries follow the diag block's in every sum, like hs::level_norms
void level_norms(const DCsr &A, const int *cf, int chunk, double *diag, double *l1gs, double *l1jac, hipStream_t s,
                 const DCsr *halo = nullptr, const int *cf_ext = nullptr);

// ---- ILU(0) of a single-rank block (HYPRE_ILU type 0, fill 0), level-scheduled
// position of the diagonal entry of every row (-1: none)
void ilu_diag_positions(const DCsr &A, long long *dpos, hipStream_t s);
// in-place IKJ factorisation of the rows rows[0..nrows) -- one level set: every row they depend on is final
void ilu_factor_level(DCsr &LU, const long long *dpos, const int *rows, int nrows, hipStream_t s);
// forward / backward substitution for one level set: y[i] = b[i] - sum_{k<i} l_ik y_k ;
// x[i] = (y[i] - sum_{j>i} u_ij x_j) / u_ii
void ilu_lower_level(const DCsr &LU, const long long *dpos, const int *rows, int nrows, const double *b, double *y,
                     hipStream_t s);
void ilu_upper_level(const DCsr &LU, const long long *dpos, const int *rows, int nrows, const double *y, double *x,
                     hipStream_t s);
// Jacobi sweeps on the triangular factors (HYPRE's iterative triangular solve):
// out = b - L_strict in   /   out = D^-1 (b - U_strict in) ; in == nullptr: out = b resp. D^-1 b
void ilu_lower_jacobi(const DCsr &LU, const long long *dpos, const double *b, const double *in, double *out, hipStream_t s);
void ilu_upper_jacobi(const DCsr &LU, const long long *dpos, const double *b, const double *in, double *out, hipStream_t s);

// ---- distributed setup on the device (amg_setup_dist.cpp: BoomerAMG::build_distributed_device)
// Column map between two extended index spaces [remote below | own range | remote above] (ascending global ids):
// column c of the source becomes below[c] (c < nb_old), own_tab[c - nb_old] or own_new0 + (c - nb_old) (own range,
// n_own columns), above[c - nb_old - n_own] (the rest); a missing table, keep_own == false or a negative table value
// drops the entry.  Tables are device arrays.
struct ExtColMap {
  int nb_old = 0, n_own = 0;
  bool keep_own = true;
  int own_new0 = 0;
  const int *own_tab = nullptr;
  const int *below = nullptr;
  const int *above = nullptr;
};
// B = rows rows[0..nout) of A (rows == null: row0, row0 + 1, ...) with the columns mapped (entries keep their stored
// order; sort = true re-sorts every row by the new column)
void select_rows(const DCsr &A, const int *rows, int row0, int nout, const ExtColMap &m, int new_ncols, bool sort,
                 DCsr &B, hipStream_t s);
// the same change of column space in place, for maps that drop nothing and keep the order (checked)
void remap_columns(DCsr &A, const ExtColMap &m, int new_ncols, hipStream_t s);
// C = the parts' rows one block after the other (same column space)
void vconcat(const DCsr *const *parts, int nparts, DCsr &C, hipStream_t s);
// C = [A | B]: row i = A's entries, then B's with columns shifted by A.ncols
void hstack(const DCsr &A, const DCsr &B, DCsr &C, hipStream_t s);
// PMIS on the rows [row0, row0 + n) of an extended strength graph S; cnt / measure / cf / tmp span S.ncols entries,
// the caller exchanges their remote parts between the steps (hs distributed PMIS, amg_setup_dist.cpp)
void pmis_dist_counts(const DCsr &S, int row0, int n, int *cnt, hipStream_t s);  // cnt[col] += 1 per strong entry
int pmis_dist_init(const DCsr &S, int row0, int n, long long gid0, int seed, const int *cnt, double *measure, int *cf,
                   int *counter, hipStream_t s);  // returns the undecided own rows
void pmis_dist_compare(const DCsr &S, int row0, int n, int ne, const int *cf, const double *measure, signed char *tmp,
                       hipStream_t s);  // tmp = 1 everywhere, then 0 for the losers of this round's comparisons
void pmis_dist_select(int row0, int n, int *cf, const signed char *tmp, hipStream_t s);
int pmis_dist_fpoints(const DCsr &S, int row0, int n, int *cf, int *counter, hipStream_t s);  // returns undecided
// dst[k] = src[idx[k] + shift] for elements of 1, 4 or 8 bytes; dst[idx[k] + shift] += v[k]; dst[idx[k] + shift] = 0 where v[k] == 0
void gather_elems(const void *src, const int *idx, int shift, int n, int elem_bytes, void *dst, hipStream_t s);
void scatter_add_int(int *dst, const int *idx, int shift, const int *v, int n, hipStream_t s);
void scatter_zero_flags(signed char *dst, const int *idx, int shift, const signed char *v, int n, hipStream_t s);
// rank[i] = number of C points before entry i of cf[0..n) (rank[n] = their count, returned)
long long count_c_points(const int *cf, int n, DVec<long long> &rank, hipStream_t s);
// cg[i] = first + rank[i] for C points, -1 otherwise
void fill_coarse_ids(const int *cf, const long long *rank, int n, long long first, long long *cg, hipStream_t s);
// C-first order of n rows: pos[i] = new position of row i (C points first, both groups in their old order), perm = its inverse
void cfirst_order(const int *cf, const long long *crank, int n, int nc, int *pos, int *perm, hipStream_t s);
// rows of P with at least one column outside [c0, c1), ascending, on the host
int rows_with_columns_outside(const DCsr &P, int c0, int c1, std::vector<int> &rows_host, hipStream_t s);
// used[c] = 1 for every column that occurs in A
void mark_used_columns(const DCsr &A, DVec<unsigned char> &used, hipStream_t s);
void add_to_ints(int *v, int n, int add, hipStream_t s);

}  // namespace sk
}  // namespace mi
