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
void sparsify_non_galerkin(DCsr &A, double tol, hipStream_t s);

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
// pos[order[q]] = q
void invert_permutation(const int *order, int n, int *pos, hipStream_t s);
// back to host arrays (lazy host copies for the inspection API)
void solve_format_to_host(const DevCSR &src, HostCSR &h, hipStream_t s);

// diagonal, l1 norm of the hybrid-GS chunks (option 4, C/F aware) and full l1 norm per row (level_norms in
// amg_setup.cpp) of a single-rank operator; cf may be null
void level_norms(const DCsr &A, const int *cf, int chunk, double *diag, double *l1gs, double *l1jac, hipStream_t s);

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

}  // namespace sk
}  // namespace mi
