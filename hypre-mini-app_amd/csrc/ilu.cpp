// Block-Jacobi ILU(k) (HYPRE_ILU type 0, level of fill k >= 0): the preconditioner / solver behind
// `preconditioner: ilu` and `method: ilu` of the driver (src/HypreSystem.cpp:328-370, :457-497).
// Every rank factorises its own diagonal block in place; rows are grouped into level sets (row i is one
// level above the deepest row it depends on), which order both the factorisation and the substitutions:
// one kernel launch per level set, every row of a set independent of the others.
#include <algorithm>
#include <cmath>

#include "kernels.hpp"
#include "solvers.hpp"

namespace mi {

namespace {
// Symbolic ILU(k) of a rank's diagonal block (ilu.c hypre_ILUSetupILUKSymbolic; oracle/oracle.c ilu_symbolic; Saad,
// Iterative Methods, alg. 10.5): row by row, level 0 on A's pattern; every kept lower entry (i,k), in ascending k,
// lets every entry (k,j), j > k, of row k propose  lev(i,j) = lev(i,k) + lev(k,j) + 1,  merged into the row's sorted
// list when that is <= fill.  Out: the pattern with A's values on A's positions and zeros on the fill.
void ilu_symbolic(const HostCSR &B, int fill, HostCSR &F) {
  const int n = B.nrows;
  F.nrows = n;
  F.ncols = B.ncols;
  F.ia.assign((size_t)n + 1, 0);
  F.ja.clear();
  F.a.clear();
  F.ja.reserve((size_t)B.nnz() * (size_t)(fill + 1));
  F.a.reserve((size_t)B.nnz() * (size_t)(fill + 1));
  std::vector<int> lv;  // level of every stored entry
  lv.reserve((size_t)B.nnz() * (size_t)(fill + 1));
  std::vector<int64_t> dpos((size_t)n, -1);
  std::vector<int> next((size_t)n + 1, n), lev((size_t)n, 0);
  std::vector<double> val((size_t)n, 0.0);
  for (int i = 0; i < n; i++) {
    int last = n;
    next[(size_t)n] = n;
    for (int64_t k = B.ia[(size_t)i]; k < B.ia[(size_t)i + 1]; k++) {  // columns ascend
      const int j = B.ja[(size_t)k];
      next[(size_t)last] = j;
      next[(size_t)j] = n;
      last = j;
      lev[(size_t)j] = 0;
      val[(size_t)j] = B.a[(size_t)k];
    }
    for (int k = next[(size_t)n]; k < i; k = next[(size_t)k]) {  // (k == n ends the list: n > i)
      if (dpos[(size_t)k] < 0) continue;
      int at = k;  // the columns row k proposes ascend
      for (int64_t q = dpos[(size_t)k] + 1; q < F.ia[(size_t)k + 1]; q++) {
        const int j = F.ja[(size_t)q];
        const int nl = lev[(size_t)k] + lv[(size_t)q] + 1;
        if (nl > fill) continue;
        while (next[(size_t)at] != n && next[(size_t)at] < j) at = next[(size_t)at];
        if (next[(size_t)at] == j) {
          if (nl < lev[(size_t)j]) lev[(size_t)j] = nl;
        } else {
          next[(size_t)j] = next[(size_t)at];
          next[(size_t)at] = j;
          lev[(size_t)j] = nl;
          val[(size_t)j] = 0.0;
        }
      }
    }
    for (int j = next[(size_t)n]; j != n; j = next[(size_t)j]) {
      if (j == i) dpos[(size_t)i] = (int64_t)F.ja.size();
      F.ja.push_back(j);
      lv.push_back(lev[(size_t)j]);
      F.a.push_back(val[(size_t)j]);
    }
    F.ia[(size_t)i + 1] = (int64_t)F.ja.size();
  }
}
}  // namespace

void IluSolver::setup(ParCSR &A) {
  ensure_init();
  hipStream_t s = ctx().stream;
  if (ilu_type != 0 || level_of_fill < 0)
    fail(1, "HYPRE_ILU: this variant is not implemented -- only type 0 (block-Jacobi ILU(k)) is (got type " +
                std::to_string(ilu_type) + ", fill " + std::to_string(level_of_fill) + ")");
  MI_REQUIRE(!A.host_diag_stale, "HYPRE_ILUSetup: the matrix has no host arrays");
  HostCSR filled;
  if (level_of_fill > 0) ilu_symbolic(A.diag, level_of_fill, filled);  // the factors live on the ILU(k) pattern
  const HostCSR &D = level_of_fill > 0 ? filled : A.diag;
  n = D.nrows;
  // level sets of the lower and of the upper factor
  std::vector<int> ll((size_t)n, 0), lu((size_t)n, 0);
  int nl = 0, nu = 0;
  for (int i = 0; i < n; i++) {
    int lv = 0;
    for (int64_t k = D.ia[(size_t)i]; k < D.ia[(size_t)i + 1] && D.ja[(size_t)k] < i; k++)
      lv = std::max(lv, ll[(size_t)D.ja[(size_t)k]] + 1);
    ll[(size_t)i] = lv;
    nl = std::max(nl, lv + 1);
  }
  for (int i = n - 1; i >= 0; i--) {
    int lv = 0;
    for (int64_t k = D.ia[(size_t)i + 1] - 1; k >= D.ia[(size_t)i] && D.ja[(size_t)k] > i; k--)
      lv = std::max(lv, lu[(size_t)D.ja[(size_t)k]] + 1);
    lu[(size_t)i] = lv;
    nu = std::max(nu, lv + 1);
  }
  auto bucket = [&](const std::vector<int> &lev, int nlev, std::vector<int> &ptr, std::vector<int> &order) {
    ptr.assign((size_t)nlev + 1, 0);
    for (int i = 0; i < n; i++) ptr[(size_t)lev[(size_t)i] + 1]++;
    for (int l = 0; l < nlev; l++) ptr[(size_t)l + 1] += ptr[(size_t)l];
    order.resize((size_t)n);
    std::vector<int> cur(ptr.begin(), ptr.end() - 1);
    for (int i = 0; i < n; i++) order[(size_t)cur[(size_t)lev[(size_t)i]]++] = i;
  };
  std::vector<int> ol, ou;
  bucket(ll, nl, lptr, ol);
  bucket(lu, nu, uptr, ou);
  order_l.upload(ol);
  order_u.upload(ou);
  LU.upload(D, s);
  dpos.alloc((size_t)n);
  sk::ilu_diag_positions(LU, dpos.p, s);
  for (int l = 0; l < nl; l++)
    sk::ilu_factor_level(LU, dpos.p, order_l.p + lptr[(size_t)l], lptr[(size_t)l + 1] - lptr[(size_t)l], s);
  MI_HIP(hipGetLastError());
  y.alloc((size_t)n);
  t.alloc((size_t)n);
  r.alloc((size_t)n);
  z.alloc((size_t)n);
  zero_on_stream(y.p, (size_t)n * sizeof(double));
  zero_on_stream(t.p, (size_t)n * sizeof(double));
  MI_HIP(hipStreamSynchronize(s));
  is_setup = true;
  if (print_level > 0 && current_comm().rank == 0)
    printf("mi_hypre ILU(%d): %d rows, %lld entries, %d lower / %d upper level sets, %s triangular solves\n", level_of_fill, n,
           (long long)LU.nnz, nl, nu, tri_solve ? "exact" : "Jacobi");
}

void IluSolver::apply(const double *rhs, double *out) {
  hipStream_t s = ctx().stream;
  if (tri_solve) {
    const int nl = (int)lptr.size() - 1, nu = (int)uptr.size() - 1;
    for (int l = 0; l < nl; l++)
      sk::ilu_lower_level(LU, dpos.p, order_l.p + lptr[(size_t)l], lptr[(size_t)l + 1] - lptr[(size_t)l], rhs, y.p, s);
    for (int l = 0; l < nu; l++)
      sk::ilu_upper_level(LU, dpos.p, order_u.p + uptr[(size_t)l], uptr[(size_t)l + 1] - uptr[(size_t)l], y.p, out, s);
  } else {
    // y <- rhs - L_strict y (lower_it times from y = rhs); out <- D^-1 (y - U_strict out) (upper_it times from D^-1 y)
    double *a = y.p, *b2 = t.p;
    sk::ilu_lower_jacobi(LU, dpos.p, rhs, nullptr, a, s);
    for (int it = 0; it < lower_it; it++) {
      sk::ilu_lower_jacobi(LU, dpos.p, rhs, a, b2, s);
      std::swap(a, b2);
    }
    // a holds y; use b2 and out alternately, finishing in out
    double *zi = (upper_it % 2 == 0) ? out : b2, *zo = (upper_it % 2 == 0) ? b2 : out;
    sk::ilu_upper_jacobi(LU, dpos.p, a, nullptr, zi, s);
    for (int it = 0; it < upper_it; it++) {
      sk::ilu_upper_jacobi(LU, dpos.p, a, zi, zo, s);
      std::swap(zi, zo);
    }
    if (zi != out) k::copy(zi, out, n, s);
  }
  MI_HIP(hipGetLastError());
}

int IluSolver::solve(ParCSR &A, ParVector &b, ParVector &x) {
  if (!is_setup) setup(A);
  MI_REQUIRE(b.n == n && x.n == n, "HYPRE_ILUSolve: vector size does not match the matrix");
  MI_REQUIRE(b.ncomp == x.ncomp, "HYPRE_ILUSolve: b and x differ in their number of components");
  Comm &comm = current_comm();
  hipStream_t s = ctx().stream;
  const int nc = b.ncomp;  // a multivector is treated component by component
  // preconditioner use (one application, no tolerance) on a zero guess: x = M^-1 b
  if (max_iter == 1 && tol <= 0.0 && zero_guess_hint()) {
    zero_guess_hint() = false;
    for (int c = 0; c < nc; c++) apply(b.all() + (size_t)c * (size_t)n, x.all() + (size_t)c * (size_t)n);
    num_iterations = 1;
    return 0;
  }
  zero_guess_hint() = false;
  const double bn = (tol > 0.0) ? std::sqrt(par_dot_host(comm, b.all(), b.all(), b.len(), s)) : 0.0;
  int it = 0;
  double rel = 0.0;
  // with a tolerance the residuals of all components are needed BEFORE the update (one norm over the multivector):
  // they are kept (rall: nc * n, allocated on first use) and reused by the update instead of being recomputed
  if (tol > 0.0 && nc > 1 && rall.n != (size_t)nc * (size_t)n) rall.alloc((size_t)nc * (size_t)n);
  while (it < max_iter) {
    if (tol > 0.0) {
      double rr = 0.0;
      for (int c = 0; c < nc; c++) {
        const size_t o = (size_t)c * (size_t)n;
        double *rc = (nc > 1) ? rall.p + o : r.p;
        A.matvec(comm, -1.0, x.all() + o, 1.0, b.all() + o, rc, s);
        rr += par_dot_host(comm, rc, rc, n, s);
      }
      const double rn = std::sqrt(rr);
      rel = (bn > 0.0) ? rn / bn : rn;
      if (rel <= tol) break;
    }
    for (int c = 0; c < nc; c++) {
      const size_t o = (size_t)c * (size_t)n;
      double *rc = (tol > 0.0 && nc > 1) ? rall.p + o : r.p;
      if (!(tol > 0.0)) A.matvec(comm, -1.0, x.all() + o, 1.0, b.all() + o, rc, s);
      apply(rc, z.p);
      k::axpy(1.0, z.p, x.all() + o, n, s);
    }
    it++;
  }
  num_iterations = it;
  final_rel_res = rel;
  return 0;
}

}  // namespace mi
