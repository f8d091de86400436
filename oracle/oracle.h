/*
 * oracle.h -- CPU restatement of the GMRES + BoomerAMG solve path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
 * executed by the product library (hypre-mini-app_amd/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and
 * there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in LLNL/hypre
 * (find_package(HYPRE 2.20.0), /root/reference/CMakeLists.txt:63-67), which is
 * not vendored and not installed here, and the reference ships no tests or
 * golden vectors.  This file restates the published
 * HYPRE algorithms (krylov/gmres.c, krylov/bicgstab.c, parcsr_ls/par_cycle.c,
 * par_relax.c, par_strength.c, par_coarsen.c, par_lr_interp.c, par_rap.c) as
 * recalled in SURVEY.md Appendix A and anchors on the reference's own call
 * sites (/root/reference/src/HypreSystem.cpp:119-326, :390-404, :673-737) and on
 * the only known answer the reference holds: its generator's b = A*1, x* = 1
 * (/root/reference/src/laplace_3d_weak_scaling.hpp:321,558,600).
 */
#ifndef ORACLE_H
#define ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef long long obig;

/* CSR, 64-bit row pointer, 32-bit local column, columns sorted ascending. */
typedef struct ocsr {
  int nrows, ncols;
  obig *ia;
  int *ja;
  double *a;
} ocsr;

ocsr *ocsr_new(int nrows, int ncols, obig nnz);
void ocsr_free(ocsr *A);
ocsr *ocsr_from_arrays(int nrows, int ncols, const obig *ia, const int *ja, const double *a);
obig ocsr_nnz(const ocsr *A);
void ocsr_copy_out(const ocsr *A, obig *ia, int *ja, double *a);
void ocsr_matvec(double alpha, const ocsr *A, const double *x, double beta, const double *b, double *y);
ocsr *ocsr_transpose(const ocsr *A);
ocsr *ocsr_matmul(const ocsr *A, const ocsr *B);

/* n^3-type Laplacians, lexicographic numbering row = x + nx*(y + ny*z);
 * stencil 7: diag 6 / off -1; stencil 27: diag 26 / off -1
 * (laplace_3d_weak_scaling.hpp:558,600); rhs = row sum (b = A*1, :321). */
ocsr *oracle_laplace(int nx, int ny, int nz, int stencil, double *rhs);

/* Park-Miller minimal standard generator = hypre_SeedRand / hypre_Rand. */
void oracle_rand_seed(int seed);
double oracle_rand(void);

typedef struct oamg_params {
  int coarsen_type;        /* 8 PMIS (HypreSystem.cpp:126); 10 HMIS, 11 one-pass RS; 6 Falgout, 1 RS, 3 RS3 (two-pass RS);
                            * 0 CLJP, 7 CLJP with one global random stream (the same thing here) */
  int interp_type;         /* 6 ext+i (library default), 3 direct, 0 classical modified, 4 multipass */
  double strong_threshold; /* 0.57 (HypreSystem.cpp:159) */
  double max_row_sum;      /* 0.9 library default */
  double trunc_factor;     /* 0 */
  int pmax_elmts;          /* 4 library default */
  int max_levels;          /* 20 (HypreSystem.cpp:157) */
  int max_coarse_size;     /* 9 */
  int min_coarse_size;     /* 0 */
  int relax_type[3];       /* down, up, coarsest (k = 1,2,3 of SetCycleRelaxType) */
  int num_sweeps[3];
  int relax_order;         /* 1 = C/F (HypreSystem.cpp:156) */
  double relax_weight;     /* 1 */
  double outer_weight;     /* 1 */
  int cycle_type;          /* 1 = V, 2 = W */
  int gs_chunk;            /* rows per hybrid-GS "thread" chunk */
  int nparts;              /* emulated rank count */
  const obig *part_starts; /* nparts+1 row starts of level 0 (NULL => 1 part) */
  int max_iter;            /* 1 as preconditioner (HypreSystem.cpp:155) */
  double tol;              /* 0 as preconditioner (HypreSystem.cpp:154) */
  obig redundant_rows;     /* nparts > 1: levels >= 1 with at most this many rows are solved redundantly by every
                            * rank (one part: HYPRE's seq_threshold idea); 0 = every level distributed */
  /* aggressive coarsening on levels < agg_num_levels (src/HypreSystem.cpp:215-229): coarsen twice (second time on
   * the second-generation strength graph), multipass interpolation (agg_interp_type 4, the only one restated) */
  int agg_num_levels;      /* 0 */
  int agg_interp_type;     /* 4 multipass */
  int agg_pmax_elmts;      /* 0 = no limit (the YAML key `pmax_elmts` lands here, HypreSystem.cpp:210-213) */
  double agg_trunc_factor; /* 0 */
  /* complex smoother on levels < smooth_num_levels (src/HypreSystem.cpp:235-320): only smooth_type 5 = block-Jacobi
   * ILU(0) is restated; a smoothing step is ilu_max_iter times  u += (LU)^-1 (f - A u)  (par_cycle.c: HYPRE_ILUSolve
   * on the level's vectors), once per sweep, on the down and the up leg, regardless of relax_order */
  int smooth_type;         /* 6 library default (Schwarz: not restated) */
  int smooth_num_levels;   /* 0 */
  int ilu_max_iter;        /* 1 */
  int ilu_tri_solve;       /* 1 exact triangular solves, 0 Jacobi iterations */
  int ilu_lower_it, ilu_upper_it; /* 5, 5 */
  /* non-Galerkin coarse operators (src/HypreSystem.cpp:161-176: HYPRE_BoomerAMGSetNonGalerkinTol /
   * SetLevelNonGalerkinTol): entry l = drop tolerance applied to the coarse operator BUILT FROM level l (HYPRE's
   * index); levels beyond the array take the last entry; NULL / 0 = Galerkin.  See sparsify_non_galerkin */
  int non_galerkin_num_tol;
  const double *non_galerkin_tol;
  int ilu_level;           /* 0: level of fill k of the ILU(k) complex smoother (ilu_type 0; src/HypreSystem.cpp:258-262) */
} oamg_params;

void oamg_default_params(oamg_params *p);

typedef struct oamg oamg;
oamg *oamg_setup(const ocsr *A, const oamg_params *p);
void oamg_free(oamg *h);
int oamg_num_levels(const oamg *h);
const ocsr *oamg_A(const oamg *h, int level);
const ocsr *oamg_P(const oamg *h, int level);
const int *oamg_cf(const oamg *h, int level);
const double *oamg_l1(const oamg *h, int level);
const obig *oamg_part_starts(const oamg *h, int level);
/* C-first ordering of a level: perm[new] = old row (NULL when the level has no C/F split) */
const int *oamg_perm(const oamg *h, int level);
/* replace the hierarchy operators by externally supplied ones (used to run the
 * oracle's solve phase on the product's hierarchy) */
oamg *oamg_from_levels(int nlev, const ocsr *const *A, const ocsr *const *P, const int *const *cf,
                       const obig *const *part_starts, const oamg_params *p);

/* one relaxation call on a level; points: 0 all, 1 C, -1 F */
void oamg_relax(const oamg *h, int level, int type, int points, const double *f, double *u);
/* one cycle, u is the initial guess and is overwritten */
void oamg_cycle(const oamg *h, const double *f, double *u);
/* HYPRE_BoomerAMGSolve semantics: up to max_iter cycles, stop on ||r||/||b|| <= tol if tol > 0 */
int oamg_solve(const oamg *h, const double *b, double *x, int *iters, double *relres);

typedef void (*oprecond_fn)(void *ctx, const double *r, double *z); /* z = M^-1 r, z arrives zeroed */
void oamg_precond(void *ctx, const double *r, double *z);

/* a preconditioner applied to every component of a multivector (component-major, n rows each) */
typedef struct omulti {
  oprecond_fn M;
  void *Mctx;
  int n, ncomp;
} omulti;
void omulti_precond(void *ctx, const double *r, double *z);

/* Block-Jacobi ILU(0) (HYPRE_ILU, ilu_type 0, level of fill 0; src/HypreSystem.cpp:328-370, :457-497):
 * every part factorises its own diagonal block in place (IKJ order, entries outside the block dropped).
 * tri_solve 1: exact forward/backward substitution; tri_solve 0: lower_it / upper_it Jacobi iterations per
 * triangular factor (HYPRE's GPU option).  HYPRE's ILU source is not available here: parity unpinned. */
typedef struct oilu oilu;
oilu *oilu_setup(const ocsr *A, int nparts, const obig *part_starts, int tri_solve, int lower_it, int upper_it);
/* the same with level of fill k (HYPRE_ILUSetLevelOfFill, src/HypreSystem.cpp:345-349; ILU(k), Saad: an entry of the
 * factors is kept when its level  lev(i,j) = min over k of lev(i,k) + lev(k,j) + 1  (0 on A's pattern) is <= k) */
oilu *oilu_setup_k(const ocsr *A, int nparts, const obig *part_starts, int tri_solve, int lower_it, int upper_it, int level_of_fill);
void oilu_free(oilu *h);
const ocsr *oilu_factor(const oilu *h);                 /* L (unit, strictly lower part) and U in A's pattern */
void oilu_apply(const oilu *h, const double *r, double *z); /* z = U^-1 L^-1 r */
void oilu_precond(void *ctx, const double *r, double *z);
/* Richardson iteration x += M^-1 (b - A x) (HYPRE_ILUSolve as a solver), returns the iteration count */
int oilu_solve(const oilu *h, const ocsr *A, const double *b, double *x, int max_iter, double tol, double *rel_res);

typedef struct okrylov_result {
  int iters;
  int converged;
  double rel_res;  /* r_norm / b_norm as the solver tracked it */
  double true_rel_res;
} okrylov_result;

/* right-preconditioned restarted GMRES(k), MGS */
void ogmres_solve(const ocsr *A, const double *b, double *x, int kdim, double tol, double atol, int maxit,
                  oprecond_fn M, void *Mctx, okrylov_result *res, double *norms /* maxit+1 or NULL */);
/* FlexGMRES: GMRES that keeps z_j = M^-1 p_j, restart residual recomputed (krylov/flexgmres.c) */
void ofgmres_solve(const ocsr *A, const double *b, double *x, int kdim, double tol, double atol, int maxit,
                   oprecond_fn M, void *Mctx, okrylov_result *res, double *norms);
void ocogmres_solve(const ocsr *A, const double *b, double *x, int kdim, int cgs, double tol, double atol, int maxit,
                    oprecond_fn M, void *Mctx, okrylov_result *res, double *norms);
/* preconditioned conjugate gradients, HYPRE default options (krylov/pcg.c) */
void opcg_solve(const ocsr *A, const double *b, double *x, double tol, double atol, int maxit, oprecond_fn M,
                void *Mctx, okrylov_result *res, double *norms);
/* right-preconditioned BiCGSTAB */
void obicgstab_solve(const ocsr *A, const double *b, double *x, double tol, double atol, int maxit, oprecond_fn M,
                     void *Mctx, okrylov_result *res, double *norms);

/* threads for the timed CPU baseline (rows split evenly); 1 = scalar port */
void oracle_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
