// Solver objects behind the opaque HYPRE_Solver handle.
#pragma once
#include "amg.hpp"

namespace mi {

struct SolverBase {
  enum Kind { K_GMRES, K_BICGSTAB, K_PCG, K_AMG, K_ILU } kind;
  explicit SolverBase(Kind k) : kind(k) {}
  virtual ~SolverBase() {}
};

// HYPRE_PtrToParSolverFcn: int f(HYPRE_Solver, HYPRE_ParCSRMatrix, HYPRE_ParVector, HYPRE_ParVector)
typedef int (*ParSolverFcn)(void *, void *, void *, void *);

struct AmgSolver : SolverBase {
  BoomerAMG amg;
  AmgSolver() : SolverBase(K_AMG) {}
};

struct KrylovSolver : SolverBase {
  double tol = 1e-6, atol = 0.0;
  int max_iter = 1000, min_iter = 0, k_dim = 5, print_level = 0, logging = 0;
  ParSolverFcn precond_solve = nullptr, precond_setup = nullptr;
  void run_precond_setup(ParCSR &A, ParVector &b, ParVector &x);
  void *precond_data = nullptr;
  int num_iterations = 0;
  double rel_residual_norm = 0.0;
  bool converged = false;
  std::vector<double> norms;  // residual history (iteration 0 = initial)
  double solve_seconds = 0.0;
  explicit KrylovSolver(Kind k) : SolverBase(k) {}
  void apply_precond(ParCSR &A, ParVector &rhs, ParVector &out);
  // matvec on every component of a multivector (krylov.cpp)
  void matvec_all(ParCSR &A, double alpha, const double *x, double beta, const double *b, double *y, int nloc, int ncomp,
                  int prof);
  // the BoomerAMG behind precond_solve when the Krylov loop may run in its level-0 ordering (krylov.cpp)
  BoomerAMG *amg_in_level_order(ParCSR &A, int n) const;
  // the loop's view of the system: the caller's objects, or (fast path) the level-0 operator of the AMG with
  // b and x permuted into bp / xp; leave_level_order scatters xp back into the caller's x
  ParVector bp, xp;
  BoomerAMG *enter_level_order(ParCSR &A_in, ParVector &b_in, ParVector &x_in, ParCSR *&A, ParVector *&b, ParVector *&x);
  void leave_level_order(BoomerAMG *amg, ParVector &x_in);
  // out = M^-1 rhs; returns where the result lives (the AMG's own level vector on the fast path unless a copy
  // into `out` is asked for)
  const double *precond_in_order(BoomerAMG *amg, ParCSR &A, ParVector &rhs, ParVector &out, bool need_copy);
};

struct GmresSolver : KrylovSolver {
  std::vector<std::unique_ptr<ParVector>> p;
  // FlexGMRES (krylov/flexgmres.c): the preconditioned directions z_j = M^-1 p_j are kept and the
  // update is x += sum y_j z_j (no extra preconditioner call); the restart residual is recomputed
  bool flexible = false;
  // orthogonalisation: 0 modified Gram-Schmidt (GMRES, FlexGMRES); 1 / 2 = classical Gram-Schmidt with one /
  // two passes, every pass one block of inner products (ONE all-reduce) and one block update (COGMRES,
  // krylov/cogmres.c; src/HypreSystem.cpp:372-388)
  int ortho = 0;
  std::vector<std::unique_ptr<ParVector>> z;
  ParVector r, w;
  GmresSolver() : KrylovSolver(K_GMRES) {}
  void setup(ParCSR &A, ParVector &b, ParVector &x);
  int solve(ParCSR &A, ParVector &b, ParVector &x);
};

struct BicgstabSolver : KrylovSolver {
  ParVector r0, r, pv, v, q, sv, t;
  BicgstabSolver() : KrylovSolver(K_BICGSTAB) {}
  void setup(ParCSR &A, ParVector &b, ParVector &x);
  int solve(ParCSR &A, ParVector &b, ParVector &x);
};

// preconditioned conjugate gradients (krylov/pcg.c; src/HypreSystem.cpp:440-455)
struct PcgSolver : KrylovSolver {
  ParVector r, pv, sv;
  int two_norm = 0;
  PcgSolver() : KrylovSolver(K_PCG) {}
  void setup(ParCSR &A, ParVector &b, ParVector &x);
  int solve(ParCSR &A, ParVector &b, ParVector &x);
};

// HYPRE_ILU, type 0 (block Jacobi) with level of fill 0: ILU(0) of this rank's diagonal block, factorised and
// applied on the device by level sets (src/HypreSystem.cpp:328-370 as preconditioner, :457-497 as solver).
// tri_solve 1: exact substitutions (one launch per level set); 0: lower/upper Jacobi sweeps (HYPRE's GPU option)
struct IluSolver : SolverBase {
  int ilu_type = 0, level_of_fill = 0, max_iter = 20, print_level = 0, tri_solve = 1, lower_it = 5, upper_it = 5;
  double tol = 1e-7;
  bool is_setup = false;
  int n = 0;
  sk::DCsr LU;
  DVec<long long> dpos;
  DVec<int> order_l, order_u;            // rows sorted by level set
  std::vector<int> lptr, uptr;           // level set boundaries in order_l / order_u
  DVec<double> y, t, r, z;
  DVec<double> rall;  // residuals of all components of a multivector solve with a tolerance (on first use)
  int num_iterations = 0;
  double final_rel_res = 0.0;
  IluSolver() : SolverBase(K_ILU) {}
  void setup(ParCSR &A);
  void apply(const double *rhs, double *out);           // out = U^-1 L^-1 rhs
  int solve(ParCSR &A, ParVector &b, ParVector &x);     // x += M^-1 (b - A x), max_iter times or to tol
};

}  // namespace mi
