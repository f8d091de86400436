// Solver objects behind the opaque HYPRE_Solver handle.
#pragma once
#include "amg.hpp"

namespace mi {

struct SolverBase {
  enum Kind { K_GMRES, K_BICGSTAB, K_PCG, K_AMG, K_STUB } kind;
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
  void *precond_data = nullptr;
  int num_iterations = 0;
  double rel_residual_norm = 0.0;
  bool converged = false;
  std::vector<double> norms;  // residual history (iteration 0 = initial)
  double solve_seconds = 0.0;
  explicit KrylovSolver(Kind k) : SolverBase(k) {}
  void apply_precond(ParCSR &A, ParVector &rhs, ParVector &out);
  // the BoomerAMG behind precond_solve when the Krylov loop may run in its level-0 ordering (krylov.cpp)
  BoomerAMG *amg_in_level_order(ParCSR &A, int n) const;
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
  ParVector bp, xp;  // b and x in the preconditioner's level-0 ordering (fast path)
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

// placeholder for solver families outside the north-star path (ILU): every call reports HYPRE_ERROR_GENERIC
struct StubSolver : SolverBase {
  std::string family;
  explicit StubSolver(const char *f) : SolverBase(K_STUB), family(f) {}
};

}  // namespace mi
