// BoomerAMG solve phase on the device: relaxation dispatch, V/W cycle
// (hypre_BoomerAMGCycle, par_cycle.c).  This is
// precondSolvePtr_ = HYPRE_BoomerAMGSolve of src/HypreSystem.cpp:324, called
// once per Arnoldi step from the GMRES loop.
//
// Every level works in its own C-first ordering (amg_setup.cpp,
// apply_cf_ordering): a C pass sweeps rows [0, nc), an F pass rows [nc, n), so
// the two passes of a sweep together stream the level's matrix once.
#include <cmath>

#include "amg.hpp"
#include "kernels.hpp"

namespace mi {

bool &zero_guess_hint() {
  static bool hint = false;
  return hint;
}

// one relaxation call, u updated in place (level ordering)
void BoomerAMG::relax(int level, int type, int points, const double *f, double *u, bool u_is_zero) {
  AmgLevel &Lv = L[(size_t)level];
  ParCSR &A = *Lv.A;
  Comm &comm = current_comm();
  hipStream_t s = ctx().stream;
  const int prof = (level == 0) ? k::PROF_RELAX_L0 : k::PROF_NONE;
  if (type == 9) {
    if (Lv.dense) {
      if (comm.size == 1) {
        k::dense_matvec(Lv.Cinv.p, f, u, Lv.n, Lv.n, s);
      } else {
        k::copy(f, Lv.fslot.p, Lv.n, s);
        comm.allgather_dev(Lv.fslot.p, Lv.fgather.p, (size_t)Lv.slot * sizeof(double), s);
        k::dense_matvec(Lv.Cinv.p, Lv.fgather.p, u, Lv.n, comm.size * Lv.slot, s);
      }
      return;
    }
    type = p.relax_type[0];  // coarsest level too large for a dense solve
  }
  const double w = p.relax_weight * p.outer_weight;
  const bool has_cf = !Lv.cf.empty();
  const signed char *cf = has_cf ? Lv.d_cf.p : nullptr;
  if (!has_cf) points = 0;
  // a zero vector has a zero halo: no exchange, no halo contribution
  const double *offc = u_is_zero ? nullptr : A.offd_contrib(comm, u, s);
  if (type == 0 || type == 7 || type == 18) {
    const double *d = (type == 18) ? Lv.d_l1jac.p : Lv.d_diag.p;
    k::jacobi(A.d_diag, u, Lv.snap.p, f, offc, d, cf, points, w, s, prof);
    k::copy(Lv.snap.p, u, Lv.n, s);
    return;
  }
  const bool l1 = (type == 8 || type == 13 || type == 14);
  const bool fwd = (type == 3 || type == 6 || type == 8 || type == 13);
  const bool bwd = (type == 4 || type == 6 || type == 8 || type == 14);
  if (!fwd && !bwd) fail(4, "BoomerAMG: relax type " + std::to_string(type) + " is not supported");
  const int row_begin = (points == -1) ? Lv.nc : 0;
  const int row_end = (points == 1) ? Lv.nc : Lv.n;
  k::gs_hybrid(A.d_diag, u, Lv.snap.p, f, offc, l1 ? Lv.d_l1gs.p : Lv.d_diag.p, cf, points, chunk(), fwd, bwd, w,
               row_begin, row_end, s, prof);
}

// which: 0 down, 1 up, 2 coarsest.  relax_order 1: C then F going down, F then
// C going up, all points on the coarsest level (hypre_BoomerAMGRelaxIF)
void BoomerAMG::relax_sweeps(int level, int which, const double *f, double *u, bool u_is_zero) {
  const int type = p.relax_type[which];
  const bool has_cf = !L[(size_t)level].cf.empty();
  for (int sw = 0; sw < p.num_sweeps[which]; sw++) {
    const bool zero = u_is_zero && sw == 0;
    if (which == 2 || p.relax_order != 1 || !has_cf) {
      relax(level, type, 0, f, u, zero);
    } else if (which == 0) {
      relax(level, type, 1, f, u, zero);
      relax(level, type, -1, f, u);
    } else {
      relax(level, type, -1, f, u);
      relax(level, type, 1, f, u);
    }
  }
}

void BoomerAMG::cycle(int level, const double *f, double *u, bool u_is_zero) {
  const int nlev = (int)L.size();
  if (level == nlev - 1) {
    relax_sweeps(level, 2, f, u, u_is_zero);
    return;
  }
  AmgLevel &Lv = L[(size_t)level];
  AmgLevel &Ln = L[(size_t)level + 1];
  Comm &comm = current_comm();
  hipStream_t s = ctx().stream;
  relax_sweeps(level, 0, f, u, u_is_zero && p.num_sweeps[0] > 0);
  // r = f - A u ; f_c = P^T r ; u_c = 0
  Lv.A->matvec(comm, -1.0, u, 1.0, f, Lv.tmp.p, s);
  k::spmv(Lv.dR, Lv.tmp.p, 1.0, 0.0, nullptr, Ln.f.p, s);
  k::fill(Ln.u.p, Ln.n, 0.0, s);
  const int ncyc = (p.cycle_type == 2 && level + 1 < nlev - 1) ? 2 : 1;
  for (int c = 0; c < ncyc; c++) cycle(level + 1, Ln.f.p, Ln.u.p, c == 0);
  // u += P e
  k::spmv(Lv.dP, Ln.u.p, 1.0, 1.0, u, u, s);
  relax_sweeps(level, 1, f, u);
}

void BoomerAMG::solve(ParCSR &A, ParVector &b, ParVector &x) {
  if (!is_setup) setup(A);
  MI_REQUIRE(x.ncomp == 1 && b.ncomp == 1, "BoomerAMGSolve: multi-component vectors are not supported");
  MI_REQUIRE(x.n == L[0].n && b.n == L[0].n, "BoomerAMGSolve: vector size does not match the matrix");
  Comm &comm = current_comm();
  hipStream_t s = ctx().stream;
  AmgLevel &L0 = L[0];
  const bool permuted = !L0.perm.empty();
  int it = 0;
  double rel = 0.0, bn = 0.0;
  if (p.tol > 0.0) bn = std::sqrt(par_dot_host(comm, b.data(), b.data(), b.n, s));
  if (permuted) k::gather(b.data(), L0.d_perm.p, L0.f.p, L0.n, s);  // caller order -> C-first order
  // a Krylov solver that has just zeroed x says so (krylov.cpp): the first cycle
  // then needs neither x nor its halo
  const bool zero_first = zero_guess_hint();
  zero_guess_hint() = false;
  while (it < p.max_iter) {
    const bool zero = zero_first && it == 0;
    if (permuted) {
      if (zero)
        k::fill(L0.u.p, L0.n, 0.0, s);
      else
        k::gather(x.data(), L0.d_perm.p, L0.u.p, L0.n, s);
      cycle(0, L0.f.p, L0.u.p, zero);
      k::scatter_set(x.data(), L0.d_perm.p, L0.u.p, L0.n, s);
    } else {
      cycle(0, b.data(), x.data(), zero);
    }
    it++;
    if (p.tol > 0.0) {
      A.matvec(comm, -1.0, x.data(), 1.0, b.data(), L0.tmp.p, s);
      const double rn = std::sqrt(par_dot_host(comm, L0.tmp.p, L0.tmp.p, b.n, s));
      rel = (bn > 0.0) ? rn / bn : rn;
      if (p.print_level > 1 && comm.rank == 0) printf("    BoomerAMG cycle %3d   ||r||/||b|| = %e\n", it, rel);
      if (rel <= p.tol) break;
    }
  }
  num_iterations = it;
  final_rel_res = rel;
}

}  // namespace mi
