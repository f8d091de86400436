// BoomerAMG-shaped preconditioner/solver object: parameters, hierarchy, V-cycle.
#pragma once
#include "parcsr.hpp"
#include "setup_kernels.hpp"

namespace mi {

struct AmgParams {
  int print_level = 0;
  int debug_flag = 0;
  int coarsen_type = 10;          // library default HMIS; the app sets 8 (HypreSystem.cpp:126)
  int interp_type = 6;            // extended+i
  double strong_threshold = 0.25; // the app sets 0.57 (HypreSystem.cpp:159)
  double max_row_sum = 0.9;
  double trunc_factor = 0.0;
  int pmax_elmts = 4;
  int max_levels = 25;
  int max_coarse_size = 9;
  int min_coarse_size = 0;
  int relax_type[3] = {13, 14, 9};  // down, up, coarsest
  int num_sweeps[3] = {1, 1, 1};
  int relax_order = 0;
  double relax_weight = 1.0;
  double outer_weight = 1.0;
  int cycle_type = 1;
  int max_iter = 20;
  double tol = 1e-7;
  int gs_chunk = 0;  // 0 => runtime default (ctx().gs_chunk)
  int agg_num_levels = 0, agg_interp_type = 4, agg_pmax_elmts = 0, keep_transpose = 0, rap2 = 0;
  double agg_trunc_factor = 0.0;
  int smooth_num_sweeps = 1;
  // complex smoother on levels < smooth_num_levels (src/HypreSystem.cpp:235-320): 5 = ILU is implemented (block-Jacobi
  // ILU(0), the IluSolver behind HYPRE_ILU); library defaults as in HYPRE
  int smooth_type = 6, smooth_num_levels = 0;
  int ilu_type = 0, ilu_level = 0, ilu_max_iter = 1, ilu_tri_solve = 1, ilu_lower_it = 5, ilu_upper_it = 5;
  // non-Galerkin coarse operators (src/HypreSystem.cpp:161-176): drop tolerance for the coarse operator built FROM
  // level l (HYPRE's index): level_tol[l] when set (>= 0), else the global one; 0 = Galerkin
  double non_galerkin_tol = 0.0;
  std::vector<double> non_galerkin_level_tol;
  double non_galerkin_tol_for(int level) const {
    if (level < (int)non_galerkin_level_tol.size() && non_galerkin_level_tol[(size_t)level] >= 0.0)
      return non_galerkin_level_tol[(size_t)level];
    return non_galerkin_tol;
  }
  bool non_galerkin() const {
    if (non_galerkin_tol > 0.0) return true;
    for (double t : non_galerkin_level_tol)
      if (t > 0.0) return true;
    return false;
  }
  // N > 1 ranks: levels >= 1 with at most this many global rows are kept whole on every rank and cycled
  // redundantly, without halo exchanges (HYPRE_BoomerAMGSetSeqThreshold); -1 = MI_HYPRE_REDUNDANT_ROWS or 200000
  long long redundant_rows = -1;
};

struct IluSolver;

struct AmgLevel {
  ParCSR *A = nullptr;
  std::shared_ptr<IluSolver> smoother;  // complex smoother of this level (smooth_type 5 on levels < smooth_num_levels)
  std::unique_ptr<ParCSR> A_own;
  int n = 0;
  HostCSR P, R;  // interpolation / restriction while the hierarchy is being built (moved into Pm / Rm)
  // transfer operators of the finished hierarchy: rectangular ParCSR (rows: this
  // level / next level, columns: next level / this level), diag + halo blocks
  std::unique_ptr<ParCSR> Pm, Rm;
  // device copies (natural ordering) kept between the Galerkin product and the C-first renumbering
  sk::DCsr sA, sP, sR;  // sR: only in the global hierarchy of the replicated setup (sliced on the device)
  // C-first ordered operators of a level built on the device, until setup_device moves them into the
  // solve-phase format (A->d_diag, Pm->d_diag, Rm->d_diag)
  sk::DCsr oA, oP, oR;
  // sub-operator of A for the first sweep on a zero guess (sk::zero_guess_operator): the down leg of every
  // cycle starts from u = 0, where most of A multiplies zeros
  DevCSR Az;
  bool has_Az = false;
  int Az_chunk = 0;  // the hybrid-GS chunk size Az was cut for (its "in-chunk" entries)
  // operator of the residual that follows a zero-guess C-then-F sweep (sk::zero_guess_operator mode 1): the F
  // pass leaves tvec = f - A_FC u_C for the F rows from t_from on (f itself before), so those rows need only
  // their F columns; t_valid: tvec belongs to the current u
  DevCSR Ar;
  bool has_Ar = false, t_valid = false;
  int t_from = 0;
  DVec<double> tvec;
  // N > 1: rows with halo entries (ascending) and, per swept row range and operator, the largest halo-free
  // stretch, cut at the units the GS kernels write back -- a pass sweeps it while the halo is still travelling
  struct InteriorRange {
    int row_begin, row_end;
    const void *op;
    int ib, ie;
    bool ok;
  };
  std::vector<int> halo_rows;
  bool halo_rows_ready = false;
  std::vector<InteriorRange> interior_cache;
  LazyInts cf;  // +1 C, -1 F (empty on the coarsest level)
  bool has_cf = false;  // the level has a C/F splitting -- a GLOBAL fact (cf itself is empty on a rank without rows)
  DVec<signed char> d_cf;
  // C-first ordering of this level: perm[new] = old local row;
  // rows [0, nc) are the C points, [nc, n) the F points.  Empty = identity.
  LazyInts perm;
  DVec<int> d_perm;
  int nc = 0;
  std::vector<double> diag, l1gs, l1jac;
  DVec<double> d_diag, d_l1gs, d_l1jac;
  DVec<double> u, f, tmp, snap;
  DVec<double> ts_work;  // second work vector of the two-stage Gauss-Seidel (relax types 11 / 12), on first use
  // coarsest level dense solve (relax type 9)
  std::vector<double> Cinv_host;
  DVec<double> Cinv;    // n_local x (size*slot) padded inverse rows
  DVec<double> fgather; // size*slot
  DVec<double> fslot;   // slot
  int slot = 0;
  bool dense = false;
};

struct BoomerAMG {
  AmgParams p;
  std::vector<AmgLevel> L;
  bool is_setup = false, host_ready = false;
  // levels with at least this many rows run their sparse products / transposes / renumbering on the
  // device; -1 = everything on host threads (HYPRE_MI_BoomerAMGSetupHostOnly)
  long long device_min_rows = -1;
  bool keep_natural_R = false;  // the replicated multi-rank setup slices A, P and R = P^T in natural ordering:
                                // device-built levels keep them on the device (sA, sP, sR), host-built ones on the host
  double t_setup_start = 0.0;
  double t_phase[6] = {0, 0, 0, 0, 0, 0};  // strength, pmis, interp, galerkin, ordering, host total
  int num_iterations = 0;
  double final_rel_res = 0.0;
  double setup_seconds = 0.0;
  double global_opcx = -1.0;  // N > 1: sum over ranks, taken at the end of the (collective) Setup
  // Internal locality numbering (single rank; amg_setup.cpp locality_order): the hierarchy is built on Q A Q^T,
  // where Q groups rows into graph-compact clusters so that a tile of consecutive rows touches few distinct columns.
  // input_order[new] = caller's row; empty = identity.  Level 0's perm is composed with it, so every solve path
  // gathers / scatters caller vectors as before.
  LazyInts input_order;
  std::unique_ptr<ParCSR> Aq_own;
  sk::DCsr pending_sA0;  // Q A Q^T built on the device by setup_host, handed to level 0 by build_natural
  bool use_locality_order(const ParCSR &A) const;
  // the matrix HYPRE_BoomerAMGSetup was called with (and its assembly stamp): a Krylov solver may run on the
  // hierarchy's own level-0 copy only when it is handed that very matrix (krylov.cpp amg_in_level_order)
  const ParCSR *source_matrix = nullptr;
  unsigned long long source_stamp = 0;
  int chunk() const;
  // the communicator this hierarchy works on: the process communicator, or a private single-rank one
  // (global hierarchy of the replicated setup, redundant coarse tail)
  Comm *forced_comm = nullptr;
  std::unique_ptr<Comm> own_comm;
  Comm &my_comm() const { return forced_comm ? *forced_comm : current_comm(); }
  void use_private_self_comm();

  // ---- redundant coarse tail (N > 1): the last entry of L is a stub that only carries this rank's slice of
  // the first redundant level (natural ordering); the level itself and everything below it live in `tail`, a
  // single-rank hierarchy every rank holds and cycles for itself
  std::unique_ptr<BoomerAMG> tail;
  std::unique_ptr<ParCSR> tail_A;   // the first redundant level, global, natural ordering (tail's fine level)
  long long stop_rows = 0;          // build_natural: stop coarsening at the first level >= 1 this small
  bool stopped_by_rows = false;
  int tail_slot = 0;                // max rows of the stub over the ranks (all-gather slot)
  gidx tail_start = 0;              // this rank's first row of the redundant level
  DVec<double> tail_fslot, tail_fgather, tail_f, tail_e;
  std::vector<int> tail_map_host;
  DVec<int> tail_map, tail_pcol;    // natural id -> all-gather position; halo columns of the last P
  void tail_cycle(bool zero_guess);
  // ---- collapsed coarse tail (one rank, or the redundant tail hierarchy of N > 1): a cycle that starts from a zero
  // guess is a LINEAR map f -> u, so for the first level with at most MI_HYPRE_DENSE_TAIL_ROWS (default 1024) rows the
  // map of the whole sub-cycle (that level and everything below it) is tabulated at Setup by cycling the unit
  // vectors, and the cycle multiplies by it: one launch instead of ~9 per level (at 512^3: levels 8-11, 36 of the
  // ~75 latency-bound launches of a cycle).  Same operator, other rounding (1e-16 relative).
  // what the tabulated maps depend on besides the operators: the cycle's parameters at tabulation time.  A setter that
  // runs AFTER Setup (relax types, sweeps, weights, cycle type, HYPRE_MI_SetGSChunk, HYPRE_MI_SetZeroGuessMode) changes
  // the cycle of the fine levels at once; cycle(0, ..) compares this signature and tabulates again when it differs
  // (the frozen maps used to keep the old smoother on the coarse levels silently)
  std::vector<double> collapsed_signature;
  std::vector<double> cycle_signature() const;
  int collapsed_level = -1, collapsed_n = 0;
  DVec<double> collapsed_Bt;  // column j of the map = row j here (n x n)
  // second stage: the level above (at most MI_HYPRE_DENSE_TAIL_ROWS2 = 4608 rows), tabulated THROUGH the first map
  int collapsed_level2 = -1, collapsed_n2 = 0;
  DVec<double> collapsed_Bt2;
  void build_collapsed_tail();
  void tabulate_cycle(int level, DVec<double> &Bt);
  void apply_global(const double *f, double *e, bool zero_guess);  // one cycle, caller (natural) ordering
  long long effective_redundant_rows() const;
  // levels of the whole hierarchy (the stub counts once, as the tail's fine level) and the owner of one
  int total_levels() const { return (int)L.size() + (tail ? (int)tail->L.size() - 1 : 0); }
  BoomerAMG &owner_of(int level, int &local) {
    const int own = tail ? (int)L.size() - 1 : (int)L.size();
    if (level < own || !tail) {
      local = level;
      return *this;
    }
    local = level - own;
    return *tail;
  }
  static long long default_device_min_rows();

  void setup(ParCSR &A) {
    device_min_rows = default_device_min_rows();
    setup_host(A);
    setup_device();
    source_matrix = &A;
    source_stamp = A.assembly_stamp;
  }
  // hierarchy construction: host only (threads + host collectives)
  void setup_host(ParCSR &A);
  void build_natural(ParCSR &A);      // strength / PMIS / interpolation / Galerkin loop, natural ordering
  void build_replicated(ParCSR &A);   // N > 1: global hierarchy on every rank, then this rank's row slices
  // N > 1: the same hierarchy built with O(N_global / P) per rank (amg_setup_dist.cpp); PMIS without aggressive
  // levels (the Ruge-Stueben family is sequential on the global graph: replicated path)
  void build_distributed(ParCSR &A);
  bool can_build_distributed() const;
  void make_local_transfer_operators();
  void finish_host();                 // l1 norms, coarsest-level dense inverse
  // device mirror of the hierarchy (needs a GPU)
  void setup_device();
  // HYPRE_BoomerAMGSolve: x is the initial guess; up to max_iter cycles
  void solve(ParCSR &A, ParVector &b, ParVector &x);

  // pieces (exposed for the parity tests).  Each level works on its own vectors
  // Lv.u / Lv.f in the level's C-first ordering.
  // u_is_zero: the caller guarantees Lv.u == 0 on entry (its halo is then known to be zero)
  void relax(int level, int type, int points, const double *f, bool u_is_zero = false);
  void relax_pair(int level, int type, int first, const double *f, bool u_is_zero = false);
  void relax_sweeps(int level, int which, const double *f, bool u_is_zero = false);
  void cycle(int level, bool u_is_zero = false);
  // cycle(level, true) never reads Lv.u before it has overwritten every row (one rank, Gauss-Seidel down sweep on
  // the zero-skipping kernels, or the dense coarsest solve): the caller need not zero-fill u first -- at 512^3 the
  // fills of a cycle are 1.6 GB of writes, 0.8 % of the solve, and one launch per level
  bool zero_cycle_ignores_u(int level);
  // renumber every level C-first (host, collective); called at the end of setup_host
  void apply_cf_ordering();
  double operator_complexity() const;
  void local_entry_counts(double &tot, double &base) const;  // this rank's share (redundant levels: 1 / size of them)
  // fill the host arrays of a level's A / P / R from the device (inspection API, coarse solve)
  void ensure_host(int level);
};

// MI_HYPRE_GS_ZERO_SKIP: 0 = sweeps on a zero guess read like any other sweep, 1 = they skip the gathers of
// known zeros, 2 = they also run on the zero-guess sub-operator (AmgLevel::Az), 3 (default) = and the residual
// that follows reuses the F pass's product with the C values (AmgLevel::Ar)
int zero_skip_mode();
void set_zero_skip_mode(int mode);  // applies to hierarchies set up afterwards (mode 2) / to every later sweep (0, 1)

// set by a Krylov solver right before it calls the preconditioner with x == 0,
// consumed (and cleared) by BoomerAMG::solve
bool &zero_guess_hint();

// counters of the distributed setup (HYPRE_MI_GetCounter): largest per-rank extended sub-problem (rows), global rows
// gathered on every rank (the redundant tail only), number of distributed setups; -1 = unknown name
long long dist_setup_counter(const char *name);
void dist_setup_counters_reset();

// host algorithms (amg_setup.cpp), exposed for tests
void host_transpose(const HostCSR &A, HostCSR &T);
void host_spgemm(const HostCSR &A, const HostCSR &B, HostCSR &C);

}  // namespace mi
