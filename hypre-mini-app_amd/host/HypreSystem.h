// nalu::HypreSystem -- the mini-app's driver object, rewritten against the
// MI355X-native HYPRE-shaped C ABI (include/*.h).  Same public methods, YAML
// keys, timers and call order as /root/reference/src/HypreSystem.h:66-91; the
// arithmetic lives behind the function-pointer table, exactly as there
// (/root/reference/src/HypreSystem.h:265-277).
#ifndef HYPRESYSTEM_H
#define HYPRESYSTEM_H

#include <string>
#include <utility>
#include <vector>

#include "HYPRE.h"
#include "HYPRE_IJ_mv.h"
#include "HYPRE_parcsr_ls.h"
#include "_hypre_parcsr_ls.h"
#include "krylov.h"
#ifdef MI_HOST_WITH_LIBHYPRE
#include "libhypre_backend.h"  // opt-in: real MPI + real libHYPRE (make app-libhypre HYPRE_DIR=...)
#else
#include "mpi_shim.h"
#endif
#include "yaml_lite.hpp"

namespace nalu {

template <typename T>
T get_optional(const YAML::Node &node, const std::string &key, T default_value) {
  YAML::Node n = node[key];
  return n ? n.as<T>() : default_value;
}

class HypreSystem {
 public:
  HypreSystem(MPI_Comm, YAML::Node &);
  HypreSystem() = delete;
  HypreSystem(const HypreSystem &) = delete;

  void load();
  void setup_precon_and_solver();
  void solve();
  //! Output the matrix, rhs and solution vectors (IJ text dialect)
  void output_linear_system();
  //! Check the solution against the reference solution provided by the user
  void check_solution();
  void summarize_timers();
  void retrieve_timers(std::vector<std::string> &names, std::vector<std::vector<double>> &data);
  void destroy_system();
  //! Device memory in use
  void checkMemory();

  // results the reference never asks HYPRE for
  int num_iterations(int solve = 0) const { return solve < (int)iterations_.size() ? iterations_[solve] : 0; }
  double final_rel_residual(int solve = 0) const { return solve < (int)relres_.size() ? relres_[solve] : 0.0; }
  bool all_close() const { return allClose_; }
  HYPRE_BigInt total_rows() const { return totalRows_; }

 private:
  // loaders / generators -> COO triples in rows_/cols_/vals_
  void load_matrix_market();
  void load_hypre_format();
  void build_stencil(int default_stencil, bool per_rank_dims);
  void determine_ij_system_sizes(const std::string &, int);
  void determine_mm_system_sizes(const std::string &);
  void init_row_decomposition();
  void build_ij_matrix(const std::string &, int);
  void build_ij_vector(std::vector<std::string> &, int, std::vector<HYPRE_IJVector> &);
  void build_mm_matrix(const std::string &);
  void build_mm_vector(std::vector<std::string> &, std::vector<HYPRE_IJVector> &);
  void read_vector_files(const YAML::Node &linsys, std::vector<std::string> &rhs, std::vector<std::string> &sln);
  void hypre_matrix_set_values();
  void hypre_vector_set_values(std::vector<HYPRE_IJVector> &vec, int component);
  void init_system();
  void assemble_system();

  void setup_boomeramg_precond();
  void setup_boomeramg_solver();
  void setup_gmres();
  void setup_cogmres();
  void setup_fgmres();
  void setup_bicg();
  void setup_cg();
  void setup_ilu_precond();
  void setup_ilu();

  void push_timer(const std::string &name, double seconds) { timers_.emplace_back(name, seconds); }

  MPI_Comm comm_;
  YAML::Node &inpfile_;

  std::vector<HYPRE_BigInt> rows_, cols_;
  std::vector<double> vals_;
  std::vector<HYPRE_BigInt> vector_indices_;
  std::vector<double> vector_values_;

  HYPRE_BigInt totalRows_{0}, numRows_{0}, iLower_{0}, iUpper_{0};

  std::vector<std::pair<std::string, double>> timers_;

  HYPRE_IJMatrix mat_ = NULL;
  HYPRE_ParCSRMatrix parMat_ = NULL;
  std::vector<HYPRE_IJVector> rhs_, sln_, slnRef_;
  std::vector<HYPRE_ParVector> parRhs_, parSln_, parSlnRef_;
  HYPRE_Solver solver_ = NULL;
  HYPRE_Solver precond_ = NULL;

  HYPRE_Int numComps_{1}, numSolves_{1}, numVectors_{1};

  HYPRE_Int (*solverDestroyPtr_)(HYPRE_Solver) = nullptr;
  HYPRE_Int (*solverSetupPtr_)(HYPRE_Solver, HYPRE_ParCSRMatrix, HYPRE_ParVector, HYPRE_ParVector) = nullptr;
  HYPRE_Int (*solverSolvePtr_)(HYPRE_Solver, HYPRE_ParCSRMatrix, HYPRE_ParVector, HYPRE_ParVector) = nullptr;
  HYPRE_Int (*solverPrecondPtr_)(HYPRE_Solver, HYPRE_PtrToParSolverFcn, HYPRE_PtrToParSolverFcn,
                                 HYPRE_Solver) = nullptr;
  HYPRE_Int (*solverItersPtr_)(HYPRE_Solver, HYPRE_Int *) = nullptr;
  HYPRE_Int (*solverResPtr_)(HYPRE_Solver, HYPRE_Real *) = nullptr;
  HYPRE_Int (*precondDestroyPtr_)(HYPRE_Solver) = nullptr;
  HYPRE_Int (*precondSetupPtr_)(HYPRE_Solver, HYPRE_ParCSRMatrix, HYPRE_ParVector, HYPRE_ParVector) = nullptr;
  HYPRE_Int (*precondSolvePtr_)(HYPRE_Solver, HYPRE_ParCSRMatrix, HYPRE_ParVector, HYPRE_ParVector) = nullptr;

  int M_{0}, N_{0};
  long long nnz_{0};
  int nx_{0}, ny_{0}, nz_{0};
  int iproc_{0}, nproc_{0};

  bool segregatedSolve_{true}, solveComplete_{false}, checkSolution_{false}, outputSystem_{false},
      outputSolution_{false}, usePrecond_{true}, writeAmgMatrices_{false}, complexNumbers_{false}, allClose_{true}, syntheticOnes_{false};
  double atol_{1.e-8}, rtol_{1.e-6};
  std::vector<int> iterations_;
  std::vector<double> relres_;
};

}  // namespace nalu

#endif /* HYPRESYSTEM_H */
