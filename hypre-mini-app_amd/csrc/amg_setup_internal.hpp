// Host setup algorithms of amg_setup.cpp that the distributed setup (amg_setup_dist.cpp) runs on its
// per-rank extended sub-problems.  Not part of any ABI.
#pragma once
#include "amg.hpp"

namespace mi {
namespace hs {

constexpr int C_PT = 1, F_PT = -1, SF_PT = -3;

// strength pattern of the diag block (columns ascending, a subsequence of the matrix row)
struct Strength {
  std::vector<int64_t> ia;
  std::vector<int> ja;
};

void strength(const ParCSR &A, double theta, double max_row_sum, Strength &S);
// interpolation rows of the single-rank operator A (diag block).  want_rows (optional): rows with a zero flag
// are left empty and the special-F markers of cf are kept (the caller owns them)
void build_interp(const ParCSR &A, const Strength &S, std::vector<int> &cf, int interp_type, double trunc_factor,
                  int pmax, HostCSR &P, int &nc_out, const std::vector<char> *want_rows = nullptr);

// classical Ruge-Stueben coarsening of a graph (first pass; second_pass: strong F-F pairs must share a C point)
void ruge_stueben(int n, const Strength &S, bool second_pass, std::vector<int> &cf);
// truncation of one interpolation row in place (trunc_factor, pmax; rescaled to the row sum, stored order kept): new length
int truncate_row(int len, int *cols, double *vals, double trunc_factor, int pmax, std::vector<char> &keep);

// non-Galerkin sparsification of a square single-rank operator, in place (amg_setup.cpp)
void sparsify_non_galerkin(HostCSR &A, double tol);

// locality numbering of a block's rows (amg_setup.cpp): order[new] = old; excluded rows (optional) come last
void locality_order(const HostCSR &D, std::vector<int> &order, const std::vector<char> *exclude);
// its pieces, shared with the device path: the seeds (ascending), and the stable sort of the final labels
// (seed rank; nseeds = never reached; LOCALITY_EXCLUDED)
constexpr int LOCALITY_EXCLUDED = -2;
constexpr int LOCALITY_MAX_ROUNDS = 64;
int locality_segment_shift(const HostCSR &D);
std::vector<int> locality_seeds(int n, int segshift, const std::vector<char> *exclude);
void locality_sort(std::vector<int> &label, int nseeds, std::vector<int> &order);  // label: seed ranks, -1, EXCLUDED (overwritten)

}  // namespace hs
}  // namespace mi
