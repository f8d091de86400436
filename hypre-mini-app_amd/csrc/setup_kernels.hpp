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

// T = A^T with ascending columns in every row (entries of one output row keep
// the order of A's rows)
void transpose(const DCsr &A, DCsr &T, hipStream_t s);

// B = rows of A taken in `perm` order (perm[new] = old; null = identity) with
// columns mapped through colpos (null = identity) and re-sorted ascending
void permute(const DCsr &A, const int *perm, const int *colpos, DCsr &B, hipStream_t s);

}  // namespace sk
}  // namespace mi
