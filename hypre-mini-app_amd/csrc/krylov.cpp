// Krylov control loops on the host, vectors and reductions on the device.
//   GMRES   : hypre_GMRESSolve (krylov/gmres.c), SURVEY A.1 -- the Solve call timed
//             at src/HypreSystem.cpp:715-727 (solverSolvePtr_, bound at :403)
//   BiCGSTAB: hypre_BiCGSTABSolve (krylov/bicgstab.c), SURVEY A.6 (bound at :423-438)
// Modified Gram-Schmidt keeps its coefficients on the device: every <p_j,p_i>
// lands in a device slot that the following axpy reads, so one Arnoldi step
// costs a single host synchronisation (the Hessenberg column copy).
#include <cmath>
#include <cstring>

#include "HYPRE_parcsr_ls.h"
#include "kernels.hpp"
#include "solvers.hpp"

extern "C" const char *HYPRE_MI_LastErrorMessage(void);

namespace mi {

// the preconditioner's Setup (HYPRE_BoomerAMGSetup, HYPRE_ILUSetup ...) through the function pointer the caller
// registered; its failure is this Setup's failure (HYPRE reports it through the global error flag)
void KrylovSolver::run_precond_setup(ParCSR &A, ParVector &b, ParVector &x) {
  if (!precond_setup) return;
  const int rc = precond_setup(precond_data, &A, &b, &x);
  if (rc) fail(rc, std::string("preconditioner setup failed: ") + HYPRE_MI_LastErrorMessage());
}


void KrylovSolver::apply_precond(ParCSR &A, ParVector &rhs, ParVector &out) {
  hipStream_t s = ctx().stream;
  if (precond_solve) {
    k::fill(out.all(), out.len(), 0.0, s);
    zero_guess_hint() = true;
    precond_solve(precond_data, &A, &rhs, &out);
    zero_guess_hint() = false;
  } else {
    k::copy(rhs.all(), out.all(), out.len(), s);
  }
}

// y_c = alpha A x_c + beta b_c for every component c of a multivector (component-major storage, nloc rows each).
// HYPRE's ParCSRMatrixMatvec does the same for hypre_ParVectorNumVectors > 1.
void KrylovSolver::matvec_all(ParCSR &A, double alpha, const double *x, double beta, const double *b, double *y, int nloc,
                              int ncomp, int prof) {
  Comm &comm = current_comm();
  hipStream_t s = ctx().stream;
  for (int c = 0; c < ncomp; c++) {
    const size_t o = (size_t)c * (size_t)nloc;
    A.matvec(comm, alpha, x + o, beta, b ? b + o : nullptr, y + o, s, prof);
  }
}

// The BoomerAMG preconditioner works in its own (C-first) ordering of level 0 and gathers / scatters its
// argument on every call.  When GMRES is bound to it in the usual way (solverPrecondPtr_ with
// HYPRE_BoomerAMGSolve, one cycle, zero tolerance -- src/HypreSystem.cpp:687, :154-155) the Krylov loop runs
// in that ordering instead: b and x are permuted once per solve, the matvec uses the level-0 copy of the
// operator, and a preconditioner application is one cycle on the Krylov vector itself -- no gather, no
// scatter, no copy.  Inner products only change their summation order.  MI_HYPRE_GMRES_PERMUTED=0 disables it.
BoomerAMG *KrylovSolver::amg_in_level_order(ParCSR &A, int n) const {
  static const bool enabled = !(getenv("MI_HYPRE_GMRES_PERMUTED") && atoi(getenv("MI_HYPRE_GMRES_PERMUTED")) == 0);
  if (!enabled || !precond_data) return nullptr;
  if (precond_solve != reinterpret_cast<ParSolverFcn>(&HYPRE_BoomerAMGSolve)) return nullptr;
  SolverBase *sb = static_cast<SolverBase *>(precond_data);
  if (sb->kind != SolverBase::K_AMG) return nullptr;
  BoomerAMG &amg = static_cast<AmgSolver *>(sb)->amg;
  if (!amg.is_setup || amg.p.max_iter != 1 || amg.p.tol != 0.0 || amg.L.empty()) return nullptr;
  AmgLevel &L0 = amg.L[0];
  if (L0.perm.empty() || L0.A == &A || L0.n != n || !L0.A->on_device) return nullptr;
  // the level-0 operator is a renumbered copy of the matrix BoomerAMGSetup saw: only when the Krylov solver is
  // handed that very matrix may its matvec run on the copy (HYPRE multiplies by the A passed to Solve, which may
  // legally differ from the operator the preconditioner was built on)
  if (amg.source_matrix != &A || amg.source_stamp != A.assembly_stamp) return nullptr;
  return &amg;
}

BoomerAMG *KrylovSolver::enter_level_order(ParCSR &A_in, ParVector &b_in, ParVector &x_in, ParCSR *&A, ParVector *&b,
                                           ParVector *&x) {
  const int n = b_in.n, nc = b_in.ncomp;
  BoomerAMG *amg = amg_in_level_order(A_in, n);
  A = &A_in;
  b = &b_in;
  x = &x_in;
  if (!amg) return nullptr;
  hipStream_t s = ctx().stream;
  if (bp.n != n || bp.ncomp != nc) {
    bp.init(b_in.start, b_in.end, nc);
    xp.init(b_in.start, b_in.end, nc);
  }
  for (int c = 0; c < nc; c++) {
    const size_t o = (size_t)c * (size_t)n;
    k::gather(b_in.all() + o, amg->L[0].d_perm.p, bp.all() + o, n, s);
    k::gather(x_in.all() + o, amg->L[0].d_perm.p, xp.all() + o, n, s);
  }
  A = amg->L[0].A;
  b = &bp;
  x = &xp;
  return amg;
}

void KrylovSolver::leave_level_order(BoomerAMG *amg, ParVector &x_in) {
  if (!amg) return;
  for (int c = 0; c < x_in.ncomp; c++) {
    const size_t o = (size_t)c * (size_t)x_in.n;
    k::scatter_set(x_in.all() + o, amg->L[0].d_perm.p, xp.all() + o, x_in.n, ctx().stream);
  }
}

const double *KrylovSolver::precond_in_order(BoomerAMG *amg, ParCSR &A, ParVector &rhs, ParVector &out, bool need_copy) {
  if (!amg) {
    apply_precond(A, rhs, out);
    return out.all();
  }
  hipStream_t s = ctx().stream;
  AmgLevel &L0 = amg->L[0];
  const int n = L0.n, nc = rhs.ncomp;
  double *own_f = L0.f.p;
  for (int c = 0; c < nc; c++) {  // one cycle per component of a multivector
    const size_t o = (size_t)c * (size_t)n;
    L0.f.p = rhs.all() + o;  // read-only inside the cycle
    if (!amg->zero_cycle_ignores_u(0)) k::fill(L0.u.p, n, 0.0, s);
    try {
      amg->cycle(0, true);
    } catch (...) {
      L0.f.p = own_f;
      throw;
    }
    L0.f.p = own_f;
    if (nc == 1 && !need_copy) return L0.u.p;
    k::copy(L0.u.p, out.all() + o, n, s);
  }
  return out.all();
}

void GmresSolver::setup(ParCSR &A, ParVector &b, ParVector &x) {
  ensure_init();
  MI_REQUIRE(b.ncomp == x.ncomp, "GMRES: b and x differ in their number of components");
  // Krylov basis vectors are created on demand in solve(): GMRES(50) rarely
  // fills its basis behind an AMG preconditioner
  r.init(b.start, b.end, b.ncomp);
  w.init(b.start, b.end, b.ncomp);
  // the basis survives a repeated Setup on vectors of the same shape (the reference calls Setup before every
  // Solve, src/HypreSystem.cpp:692 inside the loop at :681); another shape starts afresh
  if (!p.empty() && (p[0]->n != b.n || p[0]->ncomp != b.ncomp || p[0]->start != b.start)) p.clear();
  if (!z.empty() && (z[0]->n != b.n || z[0]->ncomp != b.ncomp || z[0]->start != b.start)) z.clear();
  run_precond_setup(A, b, x);
}

int GmresSolver::solve(ParCSR &A_in, ParVector &b_in, ParVector &x_in) {
  TraceRange trace_solve(flexible ? "mi_hypre FlexGMRESSolve" : (ortho ? "mi_hypre COGMRESSolve" : "mi_hypre GMRESSolve"));
  ensure_init();
  Ctx &c = ctx();
  Comm &comm = current_comm();
  hipStream_t s = c.stream;
  const double t_start = wall_time();
  const int nloc = b_in.n, nc = b_in.ncomp;
  const int n = nloc * nc;  // length of the (multi)vector: every BLAS-1 call below runs over all components
  MI_REQUIRE(x_in.ncomp == nc && x_in.n == nloc, "GMRES: b and x differ in size");
  const int kd = k_dim < 1 ? 1 : k_dim;
  MI_REQUIRE(kd + 2 <= (ortho > 1 ? 120 : 250), "GMRES: k_dim too large for the device scalar slots");
  MI_REQUIRE(ortho >= 0 && ortho <= 2, "GMRES: orthogonalisation must be 0 (MGS), 1 or 2 (classical passes)");
  if (r.n != nloc || r.ncomp != nc) setup(A_in, b_in, x_in);
  const double epsmac = 1.e-16;
  auto basis = [&](int i) -> ParVector & {
    while ((int)p.size() <= i) {
      std::unique_ptr<ParVector> v(new ParVector());
      v->init(b_in.start, b_in.end, nc);
      p.push_back(std::move(v));
    }
    return *p[(size_t)i];
  };
  auto zvec = [&](int i) -> ParVector & {
    while ((int)z.size() <= i) {
      std::unique_ptr<ParVector> v(new ParVector());
      v->init(b_in.start, b_in.end, nc);
      z.push_back(std::move(v));
    }
    return *z[(size_t)i];
  };
  std::vector<double> cs((size_t)kd + 1, 0.0), sn((size_t)kd + 1, 0.0), rs((size_t)kd + 1, 0.0);
  std::vector<std::vector<double>> hh((size_t)kd + 1, std::vector<double>((size_t)kd, 0.0));
  double *slots = c.red_out.p;

  // level ordering of the AMG preconditioner (see amg_in_level_order)
  ParCSR *Ap;
  ParVector *bq, *xq;
  BoomerAMG *amg = enter_level_order(A_in, b_in, x_in, Ap, bq, xq);
  ParCSR &A = *Ap;
  ParVector &b = *bq;
  ParVector &x = *xq;
  auto precond = [&](ParVector &rhs, ParVector &out, bool need_copy) -> const double * {
    return precond_in_order(amg, A, rhs, out, need_copy);
  };

  auto mv = [&](double alpha, const double *xv, double beta, const double *bv, double *yv) {
    matvec_all(A, alpha, xv, beta, bv, yv, nloc, nc, k::PROF_SPMV_L0);
  };
  // p_i /= ||p_i|| and the step's Hessenberg column (the `count` leading device slots) into c.h_pinned: the scaling kernel
  // posts the slots and a sequence number into pinned host memory before it starts on the vector, and the host polls the
  // number -- Givens rotations, the convergence test and the launches of the next cycle happen while that kernel still
  // runs, and no copy kernel or stream synchronisation stands between two steps (MI_HYPRE_GMRES_POLL=0: the round-3 way)
  static const bool poll = !(getenv("MI_HYPRE_GMRES_POLL") && atoi(getenv("MI_HYPRE_GMRES_POLL")) == 0);
  auto fetch_column = [&](const double *norm_slot, double *vec, int count) {
    if (!poll || count > 255) {
      k::scale_inv_sqrt_dev(norm_slot, vec, n, s);
      MI_HIP(hipMemcpyAsync(c.h_pinned, slots, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, s));
      MI_HIP(hipStreamSynchronize(s));
      return;
    }
    const unsigned long long seq = ++c.post_seq;
    k::scale_inv_sqrt_post(norm_slot, vec, n, slots, count, c.h_pinned + 256, c.h_post_flag, seq, s);
    const double t0 = wall_time();
    unsigned spins = 0;
    while (__atomic_load_n(c.h_post_flag, __ATOMIC_ACQUIRE) != seq) {
      if ((++spins & 0xffffu) == 0 && wall_time() - t0 > 30.0) {  // something is wrong with the queue: let the runtime say what
        MI_HIP(hipStreamSynchronize(s));
        if (__atomic_load_n(c.h_post_flag, __ATOMIC_ACQUIRE) != seq) fail(1, "GMRES: the Hessenberg column was not posted");
        break;
      }
    }
    memcpy(c.h_pinned, c.h_pinned + 256, (size_t)count * sizeof(double));
  };
  ParVector &p0 = basis(0);
  mv(-1.0, x.all(), 1.0, b.all(), p0.data());
  const double b_norm = std::sqrt(par_dot_host(comm, b.all(), b.all(), n, s));
  double r_norm = std::sqrt(par_dot_host(comm, p0.data(), p0.data(), n, s));
  const double r_norm_0 = r_norm;
  const double den = (b_norm > 0.0) ? b_norm : r_norm;
  const double eps = std::max(atol, tol * den);
  int iter = 0;
  converged = false;
  norms.clear();
  norms.push_back(r_norm);
  const bool chatty = print_level > 1 && comm.rank == 0;
  // gmres.c's IEEE check: a NaN in b, x0 or the operator ends the solve at once (HYPRE_ERROR_GENERIC); the same test on
  // every later residual estimate ends it as soon as a NaN appears (a preconditioner that broke down, a transport
  // whose bounded wait expired -- k::ipc_allreduce_k hands out NaN then)
  bool nan_seen = (b_norm != b_norm) || (r_norm != r_norm);
  if (nan_seen && comm.rank == 0 && print_level > 0)
    printf("ERROR detected by mi_hypre GMRES: NaN in the right-hand side, the initial guess or the operator\n");
  if (chatty) {
    printf("=============================================\n\n");
    printf("Iters     resid.norm     conv.rate  rel.res.norm\n");
    printf("-----    ------------    ---------- ------------\n");
  }

  while (iter < max_iter && !nan_seen) {
    rs[0] = r_norm;
    if (r_norm == 0.0) {
      converged = true;
      break;
    }
    if (r_norm <= eps && iter >= min_iter) {
      mv(-1.0, x.all(), 1.0, b.all(), r.data());
      r_norm = std::sqrt(par_dot_host(comm, r.data(), r.data(), n, s));
      if (r_norm <= eps) {
        converged = true;
        break;
      }
      if (chatty) printf("false convergence 1\n");
    }
    k::scale(1.0 / r_norm, basis(0).data(), n, s);
    int i = 0;
    while (i < kd && iter < max_iter) {
      i++;
      iter++;
      ParVector &pi = basis(i);
      ParVector &pim1 = basis(i - 1);
      ParVector &dir = flexible ? zvec(i - 1) : r;  // M^-1 p_{i-1}
      const double *mp;
      {
        TraceRange tr("precond (BoomerAMG cycle)");
        mp = precond(pim1, dir, flexible);
      }
      mv(1.0, mp, 0.0, nullptr, pi.data());
      TraceRange trace_ortho("Gram-Schmidt + Givens");
      if (ortho == 0) {
        // modified Gram-Schmidt with the axpy of step j-1 fused into the dot of step j
        // (and the last axpy into the norm): h_j = <p_j, w>, w -= h_j p_j, one pass each
        par_dot(comm, basis(0).data(), pi.data(), n, slots, s);
        for (int j = 1; j <= i; j++) {
          k::axpy_dot(slots + j - 1, -1.0, basis(j - 1).data(), pi.data(), j < i ? basis(j).data() : nullptr, n,
                      slots + j, s);
          if (comm.size > 1) comm.allreduce_dev(slots + j, 1, CommDType::F64, CommOp::SUM, s), c.n_allreduce++;
        }
        fetch_column(slots + i, pi.data(), i + 1);
        for (int j = 0; j < i; j++) hh[(size_t)j][(size_t)i - 1] = c.h_pinned[j];
      } else {
        // classical Gram-Schmidt: per pass ONE block of inner products (one all-reduce of i values) and one
        // block update; coefficients of the passes add up.  Slots: pass q at [128 q, 128 q + i), norm^2 at [i].
        std::vector<const double *> vecs((size_t)i);
        for (int j = 0; j < i; j++) vecs[(size_t)j] = basis(j).data();
        for (int q = 0; q < ortho; q++) {
          double *cs_dev = slots + 128 * q;
          k::mass_dot(vecs.data(), i, pi.data(), n, cs_dev, s);
          if (comm.size > 1) comm.allreduce_dev(cs_dev, (size_t)i, CommDType::F64, CommOp::SUM, s), c.n_allreduce++;
          k::mass_axpy(vecs.data(), i, cs_dev, -1.0, pi.data(), n, s);
        }
        double *nrm = (ortho == 1) ? slots + i : slots + 128 + i;  // adjacent to the last pass: one copy below
        par_dot(comm, pi.data(), pi.data(), n, nrm, s);
        fetch_column(nrm, pi.data(), 128 * (ortho - 1) + i + 1);
        for (int j = 0; j < i; j++) {
          double hsum = c.h_pinned[j];
          if (ortho > 1) hsum += c.h_pinned[128 + j];
          hh[(size_t)j][(size_t)i - 1] = hsum;
        }
        c.h_pinned[i] = c.h_pinned[128 * (ortho - 1) + i];
      }
      const double t = std::sqrt(c.h_pinned[i] > 0.0 ? c.h_pinned[i] : 0.0);
      hh[(size_t)i][(size_t)i - 1] = t;
      for (int j = 1; j < i; j++) {
        const double tt = hh[(size_t)j - 1][(size_t)i - 1];
        hh[(size_t)j - 1][(size_t)i - 1] = sn[(size_t)j - 1] * hh[(size_t)j][(size_t)i - 1] + cs[(size_t)j - 1] * tt;
        hh[(size_t)j][(size_t)i - 1] = -sn[(size_t)j - 1] * tt + cs[(size_t)j - 1] * hh[(size_t)j][(size_t)i - 1];
      }
      double gamma = std::sqrt(hh[(size_t)i - 1][(size_t)i - 1] * hh[(size_t)i - 1][(size_t)i - 1] +
                               hh[(size_t)i][(size_t)i - 1] * hh[(size_t)i][(size_t)i - 1]);
      if (gamma == 0.0) gamma = epsmac;
      cs[(size_t)i - 1] = hh[(size_t)i - 1][(size_t)i - 1] / gamma;
      sn[(size_t)i - 1] = hh[(size_t)i][(size_t)i - 1] / gamma;
      rs[(size_t)i] = -hh[(size_t)i][(size_t)i - 1] * rs[(size_t)i - 1];
      rs[(size_t)i] /= gamma;
      rs[(size_t)i - 1] = cs[(size_t)i - 1] * rs[(size_t)i - 1];
      hh[(size_t)i - 1][(size_t)i - 1] =
          sn[(size_t)i - 1] * hh[(size_t)i][(size_t)i - 1] + cs[(size_t)i - 1] * hh[(size_t)i - 1][(size_t)i - 1];
      const double prev = r_norm;
      r_norm = std::fabs(rs[(size_t)i]);
      norms.push_back(r_norm);
      if (chatty)
        printf("% 5d    %e    %f   %e\n", iter, r_norm, prev > 0 ? r_norm / prev : 0.0,
               b_norm > 0 ? r_norm / b_norm : r_norm);
      if (r_norm != r_norm) {
        nan_seen = true;
        break;
      }
      if (r_norm <= eps && iter >= min_iter) break;
    }
    if (nan_seen) break;  // no update of x from a Hessenberg matrix with NaNs
    // back substitution
    std::vector<double> y(rs.begin(), rs.begin() + i);
    y[(size_t)i - 1] = y[(size_t)i - 1] / hh[(size_t)i - 1][(size_t)i - 1];
    for (int kk = i - 2; kk >= 0; kk--) {
      double t = 0.0;
      for (int j = kk + 1; j < i; j++) t -= hh[(size_t)kk][(size_t)j] * y[(size_t)j];
      t += y[(size_t)kk];
      y[(size_t)kk] = t / hh[(size_t)kk][(size_t)kk];
    }
    if (flexible) {
      // x += sum_j y_j z_j
      for (int j = i - 1; j >= 0; j--) k::axpy(y[(size_t)j], zvec(j).data(), x.all(), n, s);
    } else {
      // w = sum_j y_j p_j ; x += M^-1 w   (y_{i-1} p_{i-1} first, then the others in descending j, as gmres.c's
      // copy / scale / axpy chain does -- in one or a few passes over the basis instead of i)
      {
        std::vector<const double *> vp((size_t)i);
        std::vector<double> cf((size_t)i);
        for (int j = i - 1, q = 0; j >= 0; j--, q++) {
          vp[(size_t)q] = basis(j).data();
          cf[(size_t)q] = y[(size_t)j];
        }
        k::lin_comb(vp.data(), cf.data(), i, true, w.data(), n, s);
      }
      const double *mw = precond(w, r, false);
      k::axpy(1.0, mw, x.all(), n, s);
    }
    if (r_norm <= eps && iter >= min_iter) {
      mv(-1.0, x.all(), 1.0, b.all(), r.data());
      r_norm = std::sqrt(par_dot_host(comm, r.data(), r.data(), n, s));
      if (r_norm <= eps) {
        converged = true;
        break;
      }
      if (chatty) printf("false convergence 2\n");
      k::copy(r.data(), basis(0).data(), n, s);
      i = 0;
    }
    if (flexible) {
      // flexgmres.c restarts from the explicitly recomputed residual
      if (i) {
        mv(-1.0, x.all(), 1.0, b.all(), basis(0).data());
        r_norm = std::sqrt(par_dot_host(comm, basis(0).data(), basis(0).data(), n, s));
      }
      continue;
    }
    // residual vector for the restart, rebuilt from the Givens data
    for (int j = i; j > 0; j--) {
      rs[(size_t)j - 1] = -sn[(size_t)j - 1] * rs[(size_t)j];
      rs[(size_t)j] = cs[(size_t)j - 1] * rs[(size_t)j];
    }
    if (i) {
      // p_i = rs_i p_i + rs_{i-1} p_{i-1} + ... + rs_1 p_1 ; p_0 = rs_0 p_0 + p_i  (gmres.c's scale / axpy chain, the
      // same products in the same order, in a few passes over the basis)
      std::vector<const double *> vp;
      std::vector<double> cf;
      for (int j = i; j > 0; j--) {
        vp.push_back(basis(j).data());
        cf.push_back(rs[(size_t)j]);
      }
      k::lin_comb(vp.data(), cf.data(), (int)vp.size(), true, basis(i).data(), n, s);
      const double *v0[2] = {basis(0).data(), basis(i).data()};
      const double c0[2] = {rs[0], 1.0};
      k::lin_comb(v0, c0, 2, true, basis(0).data(), n, s);
    }
  }
  leave_level_order(amg, x_in);
  MI_HIP(hipStreamSynchronize(s));
  num_iterations = iter;
  rel_residual_norm = (b_norm > 0.0) ? r_norm / b_norm : r_norm;
  solve_seconds = wall_time() - t_start;
  if (chatty) {
    printf("\n\nFinal L2 norm of residual: %e\n\n", r_norm);
    (void)r_norm_0;
  }
  if (nan_seen) return 1;                               // HYPRE_ERROR_GENERIC, as gmres.c
  return (iter >= max_iter && r_norm > eps) ? 256 : 0;  // HYPRE_ERROR_CONV
}

void PcgSolver::setup(ParCSR &A, ParVector &b, ParVector &x) {
  ensure_init();
  MI_REQUIRE(b.ncomp == x.ncomp, "PCG: b and x differ in their number of components");
  for (ParVector *v : {&r, &pv, &sv}) v->init(b.start, b.end, b.ncomp);
  run_precond_setup(A, b, x);
}

// hypre_PCGSolve (krylov/pcg.c), default options: two_norm 0 (the convergence
// measure is <C r, r> / <C b, b> against tol^2), no residual recomputation
int PcgSolver::solve(ParCSR &A_in, ParVector &b_in, ParVector &x_in) {
  TraceRange trace_solve("mi_hypre PCGSolve");
  ensure_init();
  Comm &comm = current_comm();
  hipStream_t s = ctx().stream;
  const double t_start = wall_time();
  const int nloc = b_in.n, nc = b_in.ncomp;
  const int n = nloc * nc;
  MI_REQUIRE(x_in.ncomp == nc && x_in.n == nloc, "PCG: b and x differ in size");
  if (r.n != nloc || r.ncomp != nc) setup(A_in, b_in, x_in);
  ParCSR *Ap;
  ParVector *bq, *xq;
  BoomerAMG *amg = enter_level_order(A_in, b_in, x_in, Ap, bq, xq);
  ParCSR &A = *Ap;
  ParVector &b = *bq;
  ParVector &x = *xq;
  auto mv = [&](double alpha, const double *xv, double beta, const double *bv, double *yv) {
    matvec_all(A, alpha, xv, beta, bv, yv, nloc, nc, k::PROF_SPMV_L0);
  };
  double bi_prod;
  if (two_norm) {
    bi_prod = par_dot_host(comm, b.all(), b.all(), n, s);
  } else {
    precond_in_order(amg, A, b, pv, true);
    bi_prod = par_dot_host(comm, pv.data(), b.all(), n, s);
  }
  double eps = tol * tol;
  norms.clear();
  converged = false;
  num_iterations = 0;
  if (!(bi_prod > 0.0)) {  // zero right-hand side: x = 0 (pcg.c)
    k::fill(x.all(), n, 0.0, s);
    leave_level_order(amg, x_in);
    MI_HIP(hipStreamSynchronize(s));
    rel_residual_norm = 0.0;
    converged = true;
    solve_seconds = wall_time() - t_start;
    return 0;
  }
  if (atol > 0.0) eps = std::max(eps, atol * atol / bi_prod);
  mv(-1.0, x.all(), 1.0, b.all(), r.data());
  precond_in_order(amg, A, r, pv, true);
  double gamma = par_dot_host(comm, r.data(), pv.data(), n, s);
  double i_prod = two_norm ? par_dot_host(comm, r.data(), r.data(), n, s) : gamma;
  norms.push_back(std::sqrt(std::fabs(i_prod) / bi_prod));
  int i = 0;
  const bool chatty = print_level > 1 && comm.rank == 0;
  bool pcg_nan = (bi_prod != bi_prod) || (i_prod != i_prod);
  while (i + 1 <= max_iter && !pcg_nan) {
    i++;
    mv(1.0, pv.data(), 0.0, nullptr, sv.data());
    const double sdotp = par_dot_host(comm, sv.data(), pv.data(), n, s);
    if (sdotp == 0.0) break;
    const double alpha = gamma / sdotp;
    const double gamma_old = gamma;
    k::axpy(alpha, pv.data(), x.all(), n, s);
    k::axpy(-alpha, sv.data(), r.data(), n, s);
    precond_in_order(amg, A, r, sv, true);
    gamma = par_dot_host(comm, r.data(), sv.data(), n, s);
    i_prod = two_norm ? par_dot_host(comm, r.data(), r.data(), n, s) : gamma;
    norms.push_back(std::sqrt(std::fabs(i_prod) / bi_prod));
    if (chatty) printf("% 5d    %e\n", i, norms.back());
    if (i_prod != i_prod || sdotp != sdotp) {  // NaN (pcg.c's IEEE check, on every step: see GmresSolver::solve)
      pcg_nan = true;
      break;
    }
    if (i_prod / bi_prod < eps && i >= min_iter) {
      converged = true;
      break;
    }
    const double beta = gamma / gamma_old;
    k::scale(beta, pv.data(), n, s);
    k::axpy(1.0, sv.data(), pv.data(), n, s);
  }
  leave_level_order(amg, x_in);
  MI_HIP(hipStreamSynchronize(s));
  num_iterations = i;
  rel_residual_norm = std::sqrt(std::fabs(i_prod) / bi_prod);
  solve_seconds = wall_time() - t_start;
  if (pcg_nan) return 1;
  return (!converged && i >= max_iter) ? 256 : 0;
}

void BicgstabSolver::setup(ParCSR &A, ParVector &b, ParVector &x) {
  ensure_init();
  MI_REQUIRE(b.ncomp == x.ncomp, "BiCGSTAB: b and x differ in their number of components");
  for (ParVector *v : {&r0, &r, &pv, &v, &q, &sv, &t}) v->init(b.start, b.end, b.ncomp);
  run_precond_setup(A, b, x);
}

int BicgstabSolver::solve(ParCSR &A_in, ParVector &b_in, ParVector &x_in) {
  TraceRange trace_solve("mi_hypre BiCGSTABSolve");
  ensure_init();
  Ctx &c = ctx();
  Comm &comm = current_comm();
  hipStream_t s = c.stream;
  const double t_start = wall_time();
  const int nloc = b_in.n, nc = b_in.ncomp;
  const int n = nloc * nc;
  MI_REQUIRE(x_in.ncomp == nc && x_in.n == nloc, "BiCGSTAB: b and x differ in size");
  if (r.n != nloc || r.ncomp != nc) setup(A_in, b_in, x_in);
  ParCSR *Ap;
  ParVector *bq, *xq;
  BoomerAMG *amg = enter_level_order(A_in, b_in, x_in, Ap, bq, xq);
  ParCSR &A = *Ap;
  ParVector &b = *bq;
  ParVector &x = *xq;
  auto mv = [&](double alpha, const double *xv, double beta, const double *bv, double *yv) {
    matvec_all(A, alpha, xv, beta, bv, yv, nloc, nc, k::PROF_SPMV_L0);
  };
  const double epsmac = 1.e-128;
  mv(-1.0, x.all(), 1.0, b.all(), r0.data());
  k::copy(r0.data(), r.data(), n, s);
  k::copy(r0.data(), pv.data(), n, s);
  const double b_norm = std::sqrt(par_dot_host(comm, b.all(), b.all(), n, s));
  double rho = par_dot_host(comm, r0.data(), r0.data(), n, s);
  double r_norm = std::sqrt(rho);
  const double den = (b_norm > 0.0) ? b_norm : r_norm;
  const double eps = std::max(atol, tol * den);
  int iter = 0;
  converged = (r_norm == 0.0);
  norms.clear();
  norms.push_back(r_norm);
  const bool chatty = print_level > 1 && comm.rank == 0;
  auto true_res_ok = [&]() {
    mv(-1.0, x.all(), 1.0, b.all(), t.data());
    const double tn = std::sqrt(par_dot_host(comm, t.data(), t.data(), n, s));
    if (tn <= eps) {
      r_norm = tn;
      return true;
    }
    return false;
  };
  bool bicg_nan = (b_norm != b_norm) || (r_norm != r_norm);
  while (!converged && iter < max_iter && !bicg_nan) {
    iter++;
    precond_in_order(amg, A, pv, v, true);
    mv(1.0, v.data(), 0.0, nullptr, q.data());
    const double temp = par_dot_host(comm, r0.data(), q.data(), n, s);
    if (std::fabs(temp) < epsmac) break;
    const double alpha = rho / temp;
    k::axpy(alpha, v.data(), x.all(), n, s);
    k::axpy(-alpha, q.data(), r.data(), n, s);
    r_norm = std::sqrt(par_dot_host(comm, r.data(), r.data(), n, s));
    if (r_norm <= eps && iter >= min_iter && true_res_ok()) {
      norms.push_back(r_norm);
      converged = true;
      break;
    }
    precond_in_order(amg, A, r, v, true);
    mv(1.0, v.data(), 0.0, nullptr, sv.data());
    const double ss = par_dot_host(comm, sv.data(), sv.data(), n, s);
    const double gamma = (ss != 0.0) ? par_dot_host(comm, r.data(), sv.data(), n, s) / ss : 0.0;
    k::axpy(gamma, v.data(), x.all(), n, s);
    k::axpy(-gamma, sv.data(), r.data(), n, s);
    r_norm = std::sqrt(par_dot_host(comm, r.data(), r.data(), n, s));
    norms.push_back(r_norm);
    if (chatty) printf("% 5d    %e    %e\n", iter, r_norm, b_norm > 0 ? r_norm / b_norm : r_norm);
    if (r_norm != r_norm) {  // NaN (bicgstab.c's IEEE check, on every step: see GmresSolver::solve)
      bicg_nan = true;
      break;
    }
    if (r_norm <= eps && iter >= min_iter && true_res_ok()) {
      converged = true;
      break;
    }
    if (std::fabs(rho) < epsmac) break;
    double beta = 1.0 / rho;
    rho = par_dot_host(comm, r0.data(), r.data(), n, s);
    beta *= rho;
    k::axpy(-gamma, q.data(), pv.data(), n, s);
    if (std::fabs(gamma) < epsmac) break;
    k::scale(beta * alpha / gamma, pv.data(), n, s);
    k::axpy(1.0, r.data(), pv.data(), n, s);
  }
  leave_level_order(amg, x_in);
  MI_HIP(hipStreamSynchronize(s));
  num_iterations = iter;
  rel_residual_norm = (b_norm > 0.0) ? r_norm / b_norm : r_norm;
  solve_seconds = wall_time() - t_start;
  if (bicg_nan) return 1;
  return (!converged && iter >= max_iter) ? 256 : 0;
}

}  // namespace mi
