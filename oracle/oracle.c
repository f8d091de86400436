/*
 * oracle.c -- CPU restatement of the GMRES + BoomerAMG path.  TEST INFRASTRUCTURE
 * ONLY, PARITY UNPINNED: see oracle.h for the scope statement.
 *
 * Every function cites the reference call site it serves (paths relative to
 * /root/reference) and the HYPRE source file whose published algorithm it
 * restates (libHYPRE itself is absent from this image).
 */
#include "oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include <time.h>
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static int g_threads = 1;
void oracle_set_threads(int n) {
  g_threads = n < 1 ? 1 : n;
#ifdef _OPENMP
  omp_set_num_threads(g_threads);
#endif
}

/* threads of the setup phase: rows are independent in strength / interpolation / the Galerkin products, every row
 * is computed by one thread in the sequential order, so the hierarchy does not depend on the thread count; the
 * per-thread marker arrays cap the count */
static int setup_threads(void) { return g_threads > 32 ? 32 : g_threads; }
#ifdef _OPENMP
#define OTHREAD() omp_get_thread_num()
#else
#define OTHREAD() 0
#endif

#define C_PT 1
#define F_PT (-1)
#define SF_PT (-3)

static void *xmalloc(size_t n) {
  void *p = malloc(n ? n : 1);
  if (!p) {
    fprintf(stderr, "oracle: out of memory (%zu bytes)\n", n);
    abort();
  }
  return p;
}
static void *xcalloc(size_t n, size_t s) {
  void *p = calloc(n ? n : 1, s);
  if (!p) {
    fprintf(stderr, "oracle: out of memory\n");
    abort();
  }
  return p;
}

/* ------------------------------------------------------------------ CSR -- */

ocsr *ocsr_new(int nrows, int ncols, obig nnz) {
  ocsr *A = (ocsr *)xmalloc(sizeof(ocsr));
  A->nrows = nrows;
  A->ncols = ncols;
  A->ia = (obig *)xcalloc((size_t)nrows + 1, sizeof(obig));
  A->ja = (int *)xmalloc(sizeof(int) * (size_t)nnz);
  A->a = (double *)xmalloc(sizeof(double) * (size_t)nnz);
  return A;
}
void ocsr_free(ocsr *A) {
  if (!A) return;
  free(A->ia);
  free(A->ja);
  free(A->a);
  free(A);
}
ocsr *ocsr_from_arrays(int nrows, int ncols, const obig *ia, const int *ja, const double *a) {
  ocsr *A = ocsr_new(nrows, ncols, ia[nrows]);
  memcpy(A->ia, ia, sizeof(obig) * ((size_t)nrows + 1));
  memcpy(A->ja, ja, sizeof(int) * (size_t)ia[nrows]);
  memcpy(A->a, a, sizeof(double) * (size_t)ia[nrows]);
  return A;
}
obig ocsr_nnz(const ocsr *A) { return A->ia[A->nrows]; }
void ocsr_copy_out(const ocsr *A, obig *ia, int *ja, double *a) {
  memcpy(ia, A->ia, sizeof(obig) * ((size_t)A->nrows + 1));
  memcpy(ja, A->ja, sizeof(int) * (size_t)A->ia[A->nrows]);
  memcpy(a, A->a, sizeof(double) * (size_t)A->ia[A->nrows]);
}

/* y = alpha*A*x + beta*b  (hypre_ParCSRMatrixMatvecOutOfPlace) */
void ocsr_matvec(double alpha, const ocsr *A, const double *x, double beta, const double *b, double *y) {
  const int n = A->nrows;
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < n; i++) {
    double s = 0.0;
    for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) s += A->a[k] * x[A->ja[k]];
    y[i] = (beta == 0.0) ? alpha * s : alpha * s + beta * b[i];
  }
}

ocsr *ocsr_transpose(const ocsr *A) {
  const obig nnz = ocsr_nnz(A);
  ocsr *T = ocsr_new(A->ncols, A->nrows, nnz);
  for (obig k = 0; k < nnz; k++) T->ia[A->ja[k] + 1]++;
  for (int i = 0; i < A->ncols; i++) T->ia[i + 1] += T->ia[i];
  obig *pos = (obig *)xmalloc(sizeof(obig) * ((size_t)A->ncols + 1));
  memcpy(pos, T->ia, sizeof(obig) * ((size_t)A->ncols + 1));
  for (int i = 0; i < A->nrows; i++)
    for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) {
      obig q = pos[A->ja[k]]++;
      T->ja[q] = i;
      T->a[q] = A->a[k];
    }
  free(pos);
  return T;
}

static int cmp_int(const void *a, const void *b) {
  int x = *(const int *)a, y = *(const int *)b;
  return (x > y) - (x < y);
}

/* C = A*B, Gustavson row by row: for k in row i of A (stored order), for j in
 * row k of B (stored order) acc[j] += a_ik*b_kj; output columns ascending.
 * (Galerkin product of par_rap.c is R*(A*P) evaluated with this routine.) */
ocsr *ocsr_matmul(const ocsr *A, const ocsr *B) {
  const int n = A->nrows, m = B->ncols;
  const int nthr = setup_threads();
  obig *cia = (obig *)xcalloc((size_t)n + 1, sizeof(obig));
#pragma omp parallel num_threads(nthr) if (nthr > 1)
  {
    int *mark = (int *)xmalloc(sizeof(int) * (size_t)m);
    for (int j = 0; j < m; j++) mark[j] = -1;
#pragma omp for schedule(static)
    for (int i = 0; i < n; i++) {
      obig cnt = 0;
      for (obig ka = A->ia[i]; ka < A->ia[i + 1]; ka++) {
        int k = A->ja[ka];
        for (obig kb = B->ia[k]; kb < B->ia[k + 1]; kb++) {
          int j = B->ja[kb];
          if (mark[j] != i) {
            mark[j] = i;
            cnt++;
          }
        }
      }
      cia[i + 1] = cnt;
    }
    free(mark);
  }
  for (int i = 0; i < n; i++) cia[i + 1] += cia[i];
  ocsr *C = ocsr_new(n, m, cia[n]);
  memcpy(C->ia, cia, sizeof(obig) * ((size_t)n + 1));
  free(cia);
#pragma omp parallel num_threads(nthr) if (nthr > 1)
  {
    int *mark = (int *)xmalloc(sizeof(int) * (size_t)m);
    double *acc = (double *)xcalloc((size_t)m, sizeof(double));
    for (int j = 0; j < m; j++) mark[j] = -1;
#pragma omp for schedule(static)
    for (int i = 0; i < n; i++) {
      obig q = C->ia[i];
      for (obig ka = A->ia[i]; ka < A->ia[i + 1]; ka++) {
        int k = A->ja[ka];
        double av = A->a[ka];
        for (obig kb = B->ia[k]; kb < B->ia[k + 1]; kb++) {
          int j = B->ja[kb];
          if (mark[j] != i) {
            mark[j] = i;
            C->ja[q++] = j;
            acc[j] = av * B->a[kb];
          } else
            acc[j] += av * B->a[kb];
        }
      }
      qsort(C->ja + C->ia[i], (size_t)(C->ia[i + 1] - C->ia[i]), sizeof(int), cmp_int);
      for (obig k = C->ia[i]; k < C->ia[i + 1]; k++) C->a[k] = acc[C->ja[k]];
    }
    free(acc);
    free(mark);
  }
  return C;
}

/* ------------------------------------------------------------ generator -- */

/* src/laplace_3d_weak_scaling.hpp:216-322 (stencil offsets dz,dy,dx in k order,
 * out-of-domain neighbours dropped), :558/:600 (values), :321 (rhs = row sum);
 * 7-point variant per BASELINE.json / SURVEY 8(d). */
ocsr *oracle_laplace(int nx, int ny, int nz, int stencil, double *rhs) {
  const obig n = (obig)nx * ny * nz;
  const double dv = (stencil == 27) ? 26.0 : 6.0;
  obig nnz = 0;
  for (int z = 0; z < nz; z++)
    for (int y = 0; y < ny; y++)
      for (int x = 0; x < nx; x++)
        for (int dz = -1; dz <= 1; dz++)
          for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
              if (stencil == 7 && (abs(dx) + abs(dy) + abs(dz) > 1)) continue;
              int X = x + dx, Y = y + dy, Z = z + dz;
              if (X < 0 || X >= nx || Y < 0 || Y >= ny || Z < 0 || Z >= nz) continue;
              nnz++;
            }
  ocsr *A = ocsr_new((int)n, (int)n, nnz);
  obig q = 0;
  for (int z = 0; z < nz; z++)
    for (int y = 0; y < ny; y++)
      for (int x = 0; x < nx; x++) {
        obig row = x + (obig)nx * (y + (obig)ny * z);
        double sum = 0.0;
        for (int dz = -1; dz <= 1; dz++)
          for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
              if (stencil == 7 && (abs(dx) + abs(dy) + abs(dz) > 1)) continue;
              int X = x + dx, Y = y + dy, Z = z + dz;
              if (X < 0 || X >= nx || Y < 0 || Y >= ny || Z < 0 || Z >= nz) continue;
              obig col = X + (obig)nx * (Y + (obig)ny * Z);
              double v = (col == row) ? dv : -1.0;
              A->ja[q] = (int)col;
              A->a[q] = v;
              sum += v;
              q++;
            }
        A->ia[row + 1] = q;
        if (rhs) rhs[row] = sum;
      }
  return A;
}

/* ------------------------------------------------------------------ RNG -- */
/* hypre_SeedRand / hypre_Rand (utilities/random.c): Park-Miller, a = 16807,
 * m = 2^31-1, q = 127773, r = 2836.  PMIS seeds it with 2747 (one global stream). */
static int g_seed = 13579;
void oracle_rand_seed(int seed) {
  if (seed == 0) seed = 13579;
  g_seed = seed;
}
double oracle_rand(void) {
  const int a = 16807, m = 2147483647, q = 127773, r = 2836;
  int lo = g_seed % q, hi = g_seed / q;
  int t = a * lo - r * hi;
  g_seed = (t > 0) ? t : t + m;
  return (double)g_seed / m;
}

/* ------------------------------------------------------------ AMG setup -- */

typedef struct olevel {
  ocsr *A, *P, *R;
  int own_A, own_P;
  int *cf;       /* C_PT / F_PT per row (NULL on coarsest) */
  double *diag;  /* a_ii */
  double *l1gs;  /* l1 norm option 4, chunk/part/CF aware (relax 8,13,14) */
  double *l1jac; /* full row l1 norm (relax 18) */
  obig *part_starts;
  int *perm; /* new -> old row of the C-first ordering (NULL = identity) */
  double *u, *f, *tmp, *old; /* work vectors (u,f unused on level 0) */
  double *Cinv;              /* dense inverse of the coarsest operator (relax 9) */
  struct oilu *smoother;     /* complex smoother of this level (smooth_type 5, levels < smooth_num_levels) */
} olevel;

struct oamg {
  oamg_params p;
  int nlev;
  olevel *L;
};

void oamg_default_params(oamg_params *p) {
  memset(p, 0, sizeof(*p));
  p->coarsen_type = 8;
  p->interp_type = 6;
  p->strong_threshold = 0.57;
  p->max_row_sum = 0.9;
  p->trunc_factor = 0.0;
  p->pmax_elmts = 4;
  p->max_levels = 20;
  p->max_coarse_size = 9;
  p->min_coarse_size = 0;
  p->relax_type[0] = p->relax_type[1] = 8;
  p->relax_type[2] = 9;
  p->num_sweeps[0] = p->num_sweeps[1] = p->num_sweeps[2] = 1;
  p->relax_order = 1;
  p->relax_weight = 1.0;
  p->outer_weight = 1.0;
  p->cycle_type = 1;
  p->gs_chunk = 8;
  p->nparts = 1;
  p->part_starts = NULL;
  p->max_iter = 1;
  p->tol = 0.0;
  p->redundant_rows = 0;
  p->agg_num_levels = 0;
  p->agg_interp_type = 4;
  p->agg_pmax_elmts = 0;
  p->agg_trunc_factor = 0.0;
  p->smooth_type = 6;
  p->smooth_num_levels = 0;
  p->ilu_max_iter = 1;
  p->ilu_tri_solve = 1;
  p->ilu_lower_it = p->ilu_upper_it = 5;
  p->non_galerkin_num_tol = 0;
  p->non_galerkin_tol = NULL;
  p->ilu_level = 0;
}

static int *part_of_rows(int n, int nparts, const obig *ps) {
  int *po = (int *)xmalloc(sizeof(int) * (size_t)n);
  for (int p = 0; p < nparts; p++)
    for (obig i = ps[p]; i < ps[p + 1]; i++) po[i] = p;
  return po;
}

/* Strength of connection (par_strength.c): for a_ii > 0 entry j is
 * strong iff a_ij < theta * min_k a_ik; rows with |row sum| > max_row_sum*|a_ii|
 * have no strong connections.  Returns CSR pattern S (off-diagonal only). */
static void strength(const ocsr *A, double theta, double max_row_sum, obig **Sia_out, int **Sja_out) {
  const int n = A->nrows;
  obig *Sia = (obig *)xcalloc((size_t)n + 1, sizeof(obig));
  /* pass 0 counts the strong entries of every row, pass 1 writes them (rows are independent) */
  int *Sja = NULL;
  for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) {
      for (int i = 0; i < n; i++) Sia[i + 1] += Sia[i];
      Sja = (int *)xmalloc(sizeof(int) * (size_t)(Sia[n] ? Sia[n] : 1));
    }
#pragma omp parallel for schedule(static) num_threads(setup_threads()) if (setup_threads() > 1)
    for (int i = 0; i < n; i++) {
      double diag = 0.0, row_sum = 0.0, scale = 0.0;
      for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) {
        row_sum += A->a[k];
        if (A->ja[k] == i) diag = A->a[k];
      }
      for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) {
        if (A->ja[k] == i) continue;
        if (diag < 0) {
          if (A->a[k] > scale) scale = A->a[k];
        } else {
          if (A->a[k] < scale) scale = A->a[k];
        }
      }
      int all_weak = (fabs(row_sum) > fabs(diag) * max_row_sum) && (max_row_sum < 1.0);
      obig q = pass ? Sia[i] : 0;
      if (!all_weak)
        for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) {
          if (A->ja[k] == i) continue;
          int strong = (diag < 0) ? (A->a[k] > theta * scale) : (A->a[k] < theta * scale);
          if (strong) {
            if (pass) Sja[q] = A->ja[k];
            q++;
          }
        }
      if (!pass) Sia[i + 1] = q;
    }
  }
  *Sia_out = Sia;
  *Sja_out = Sja;
}

/* PMIS (par_coarsen.c hypre_BoomerAMGCoarsenPMIS; coarsen_type 8,
 * HypreSystem.cpp:126).  The graph is the GLOBAL strength graph whatever the row
 * partition; measure = |S^T row| + rand, one Park-Miller stream seeded 2747 and
 * drawn in global row order (HYPRE's "seq_rand" mode), so the splitting does
 * not depend on the number of ranks. */
static void pmis(int n, const obig *Sia, const int *Sja, const int *part_of, int nparts, const obig *ps, int *cf) {
  double *measure = (double *)xcalloc((size_t)n, sizeof(double));
  (void)nparts;
  (void)ps;
  for (int i = 0; i < n; i++)
    for (obig k = Sia[i]; k < Sia[i + 1]; k++)
      if (part_of[Sja[k]] == part_of[i]) measure[Sja[k]] += 1.0;
  oracle_rand_seed(2747);
  for (int i = 0; i < n; i++) measure[i] += oracle_rand();
  int *graph = (int *)xmalloc(sizeof(int) * (size_t)n);
  int *tmp = (int *)xmalloc(sizeof(int) * (size_t)n);
  int ng = 0;
  for (int i = 0; i < n; i++) {
    int nloc = 0;
    for (obig k = Sia[i]; k < Sia[i + 1]; k++)
      if (part_of[Sja[k]] == part_of[i]) nloc++;
    if (nloc == 0) {
      cf[i] = SF_PT;
      measure[i] = 0.0;
    } else if (measure[i] < 1.0) {
      cf[i] = F_PT;
      measure[i] = 0.0;
    } else {
      cf[i] = 0;
      graph[ng++] = i;
    }
  }
  while (ng > 0) {
    for (int g = 0; g < ng; g++) tmp[graph[g]] = 1;
    for (int g = 0; g < ng; g++) {
      int i = graph[g];
      for (obig k = Sia[i]; k < Sia[i + 1]; k++) {
        int j = Sja[k];
        if (part_of[j] != part_of[i] || cf[j] != 0) continue; /* decided points have left the graph */
        if (measure[i] > measure[j])
          tmp[j] = 0;
        else if (measure[j] > measure[i])
          tmp[i] = 0;
      }
    }
    for (int g = 0; g < ng; g++) {
      int i = graph[g];
      if (tmp[i] == 1) {
        cf[i] = C_PT;
        measure[i] = 0.0;
      }
    }
    for (int g = 0; g < ng; g++) {
      int i = graph[g];
      if (cf[i] != 0) continue;
      for (obig k = Sia[i]; k < Sia[i + 1]; k++) {
        int j = Sja[k];
        if (part_of[j] == part_of[i] && cf[j] == C_PT) {
          cf[i] = F_PT;
          measure[i] = 0.0;
          break;
        }
      }
    }
    int m = 0;
    for (int g = 0; g < ng; g++)
      if (cf[graph[g]] == 0) graph[m++] = graph[g];
    ng = m;
  }
  free(measure);
  free(graph);
  free(tmp);
}


/* Transposed strength pattern: row i of S^T lists, ascending, the points that strongly depend on i. */
static void strength_transpose(int n, const obig *Sia, const int *Sja, obig **Tia_out, int **Tja_out) {
  obig *Tia = (obig *)xcalloc((size_t)n + 1, sizeof(obig));
  int *Tja = (int *)xmalloc(sizeof(int) * (size_t)Sia[n]);
  for (obig k = 0; k < Sia[n]; k++) Tia[Sja[k] + 1]++;
  for (int i = 0; i < n; i++) Tia[i + 1] += Tia[i];
  obig *pos = (obig *)xmalloc(sizeof(obig) * ((size_t)n + 1));
  memcpy(pos, Tia, sizeof(obig) * ((size_t)n + 1));
  for (int i = 0; i < n; i++)
    for (obig k = Sia[i]; k < Sia[i + 1]; k++) Tja[pos[Sja[k]]++] = i;
  free(pos);
  *Tia_out = Tia;
  *Tja_out = Tja;
}

/* Bucket lists of the Ruge-Stueben first pass (hypre_enter_on_lists / hypre_remove_point, amg_linklist.c):
 * one FIFO list per integer measure; the next C point is the HEAD of the list with the largest measure. */
typedef struct rs_lists {
  int *head, *tail, *prev, *next, top, nb;
} rs_lists;
static void rs_enter(rs_lists *q, int m, int i) {
  q->prev[i] = q->tail[m];
  q->next[i] = -1;
  if (q->tail[m] >= 0)
    q->next[q->tail[m]] = i;
  else
    q->head[m] = i;
  q->tail[m] = i;
  if (m > q->top) q->top = m;
}
static void rs_remove(rs_lists *q, int m, int i) {
  if (q->prev[i] >= 0)
    q->next[q->prev[i]] = q->next[i];
  else
    q->head[m] = q->next[i];
  if (q->next[i] >= 0)
    q->prev[q->next[i]] = q->prev[i];
  else
    q->tail[m] = q->prev[i];
}

/* Classical Ruge-Stueben coarsening (par_coarsen.c hypre_BoomerAMGCoarsenRuge) on the GLOBAL strength graph.
 * First pass: measure_i = |S^T_i|; repeatedly the head of the highest list becomes C, every undecided point
 * that depends on it becomes F and the points THOSE depend on gain one, the points the new C point depends on
 * lose one (falling to zero makes them F).  Points that influence nobody are F from the start; rows without
 * strong connections are special F points.
 * Second pass (second_pass != 0; coarsen types 1, 3, 6): every strong F-F pair must share a C point -- for an
 * F point i the first strong F neighbour j without a common C point tentatively becomes C, a second one makes
 * i itself C and j F again.
 * coarsen_type 10 (HMIS) = first pass, then PMIS on what is left undecided, 6 (Falgout) = both passes, then
 * CLJP on what is left: on ONE part the first pass decides every point and nothing is left -- as on one HYPRE rank.
 * On several parts 10 / 11 / 1 run this routine per part (coarsen_by_type_parts below, HYPRE's per-processor
 * definition); 6 / 3 keep seeing the whole graph. */
static void ruge_stueben(int n, const obig *Sia, const int *Sja, int second_pass, int *cf) {
  obig *Tia;
  int *Tja;
  strength_transpose(n, Sia, Sja, &Tia, &Tja);
  int *measure = (int *)xmalloc(sizeof(int) * (size_t)n);
  int maxm = 0;
  for (int i = 0; i < n; i++) {
    measure[i] = (int)(Tia[i + 1] - Tia[i]);
    if (measure[i] > maxm) maxm = measure[i];
  }
  rs_lists q;
  q.nb = 2 * maxm + 2; /* a measure grows by at most one per point that depends on its owner */
  q.head = (int *)xmalloc(sizeof(int) * (size_t)q.nb);
  q.tail = (int *)xmalloc(sizeof(int) * (size_t)q.nb);
  q.prev = (int *)xmalloc(sizeof(int) * (size_t)n);
  q.next = (int *)xmalloc(sizeof(int) * (size_t)n);
  for (int m = 0; m < q.nb; m++) q.head[m] = q.tail[m] = -1;
  q.top = 0;
  int num_left = 0;
  for (int i = 0; i < n; i++) {
    if (Sia[i + 1] == Sia[i]) {
      cf[i] = SF_PT;
      measure[i] = 0;
    } else {
      cf[i] = 0;
      num_left++;
    }
  }
  for (int j = 0; j < n; j++) {
    if (cf[j] != 0) continue;
    if (measure[j] > 0) {
      rs_enter(&q, measure[j], j);
    } else {
      cf[j] = F_PT;
      num_left--;
      for (obig k = Sia[j]; k < Sia[j + 1]; k++) {
        const int nb = Sja[k];
        if (cf[nb] != 0) continue; /* decided (or special) */
        if (nb < j) {
          if (measure[nb] > 0) rs_remove(&q, measure[nb], nb);
          measure[nb]++;
          rs_enter(&q, measure[nb], nb);
        } else
          measure[nb]++;
      }
    }
  }
  while (num_left > 0) {
    while (q.top > 0 && q.head[q.top] < 0) q.top--;
    const int c = q.head[q.top];
    if (c < 0) break; /* cannot happen: every undecided point sits on a list */
    cf[c] = C_PT;
    rs_remove(&q, measure[c], c);
    measure[c] = 0;
    num_left--;
    for (obig j = Tia[c]; j < Tia[c + 1]; j++) {
      const int nb = Tja[j];
      if (cf[nb] != 0) continue;
      cf[nb] = F_PT;
      rs_remove(&q, measure[nb], nb);
      num_left--;
      for (obig k = Sia[nb]; k < Sia[nb + 1]; k++) {
        const int n2 = Sja[k];
        if (cf[n2] != 0) continue;
        rs_remove(&q, measure[n2], n2);
        measure[n2]++;
        rs_enter(&q, measure[n2], n2);
      }
    }
    for (obig j = Sia[c]; j < Sia[c + 1]; j++) {
      const int nb = Sja[j];
      if (cf[nb] != 0) continue;
      rs_remove(&q, measure[nb], nb);
      measure[nb]--;
      if (measure[nb] > 0)
        rs_enter(&q, measure[nb], nb);
      else {
        cf[nb] = F_PT;
        num_left--;
        for (obig k = Sia[nb]; k < Sia[nb + 1]; k++) {
          const int n2 = Sja[k];
          if (cf[n2] != 0) continue;
          rs_remove(&q, measure[n2], n2);
          measure[n2]++;
          rs_enter(&q, measure[n2], n2);
        }
      }
    }
  }
  if (second_pass) {
    int *mark = (int *)xmalloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) mark[i] = -1;
    for (int i = 0; i < n; i++) {
      if (cf[i] != F_PT) continue;
      int tentative = -1;
      for (;;) {
        for (obig k = Sia[i]; k < Sia[i + 1]; k++)
          if (cf[Sja[k]] == C_PT) mark[Sja[k]] = i;
        int lonely = -1; /* first strong F neighbour that shares no C point with i */
        for (obig k = Sia[i]; k < Sia[i + 1] && lonely < 0; k++) {
          const int j = Sja[k];
          if (cf[j] != F_PT) continue;
          int shared = 0;
          for (obig kk = Sia[j]; kk < Sia[j + 1]; kk++)
            if (mark[Sja[kk]] == i && cf[Sja[kk]] == C_PT) {
              shared = 1;
              break;
            }
          if (!shared) lonely = j;
        }
        if (lonely < 0) break;
        if (tentative < 0) {
          tentative = lonely;
          cf[lonely] = C_PT;
        } else {
          cf[i] = C_PT;
          cf[tentative] = F_PT;
          break;
        }
      }
    }
    free(mark);
  }
  free(Tia);
  free(Tja);
  free(measure);
  free(q.head);
  free(q.tail);
  free(q.prev);
  free(q.next);
}

/* Coarsening by type (HYPRE_BoomerAMGSetCoarsenType, src/HypreSystem.cpp:125-126; the sample input asks for 6,
 * etc/hypre_app.yaml:35).  Returns 0, or -1 for a type that is not restated (0 CLJP, 7, 9, 21, 22). */
/* CLJP coarsening (par_coarsen.c hypre_BoomerAMGCoarsen; Cleary, Luby, Jones, Plassmann): coarsen_type 0, and 7 =
 * its variant with ONE global random stream, which is what this specification always draws from (as PMIS does).
 *   w(i) = |S^T_i| + Park-Miller(2747) element i;  points nobody depends on (w < 1) are F from the start
 *   repeat: D = undecided points whose w beats every undecided neighbour's (neighbours through S in either
 *           direction, the comparisons run along the rows of S as in hypre_BoomerAMGIndepSet) -> new C points
 *     H1: a C point does not interpolate: the edges c -> j leave the graph, w(j)-- for undecided j
 *     H2: for an undecided i, the edges to its C points leave the graph; an undecided j in S_i that shares one of
 *         those C points (j depends on it too) is no longer needed by i: edge i -> j leaves, w(j)--
 *     undecided points with w < 1 become F
 * Every F point with a non-empty row ends with a C point in it: an edge i -> j that is still there keeps w(j) >= 1,
 * so j stays undecided until it is picked. */
static void cljp(int n, const obig *Sia, const int *Sja, int *cf) {
  const obig nnz = Sia[n];
  char *gone = (char *)xcalloc((size_t)(nnz ? nnz : 1), 1);
  double *measure = (double *)xcalloc((size_t)n, sizeof(double));
  for (obig k = 0; k < nnz; k++) measure[Sja[k]] += 1.0;
  oracle_rand_seed(2747);
  for (int i = 0; i < n; i++) measure[i] += oracle_rand();
  int *graph = (int *)xmalloc(sizeof(int) * (size_t)n);
  int *tmp = (int *)xcalloc((size_t)n, sizeof(int));
  int *common = (int *)xcalloc((size_t)n, sizeof(int)); /* common[c] == i + 1: C point c is in the row of i */
  int ng = 0;
  for (int i = 0; i < n; i++) {
    if (measure[i] < 1.0)
      cf[i] = (Sia[i + 1] == Sia[i]) ? SF_PT : F_PT;
    else {
      cf[i] = 0;
      graph[ng++] = i;
    }
  }
  while (ng > 0) {
    for (int g = 0; g < ng; g++) tmp[graph[g]] = 1;
    for (int g = 0; g < ng; g++) {
      const int i = graph[g];
      for (obig k = Sia[i]; k < Sia[i + 1]; k++) {
        const int j = Sja[k];
        if (cf[j] != 0) continue;
        if (measure[i] > measure[j])
          tmp[j] = 0;
        else if (measure[j] > measure[i])
          tmp[i] = 0;
      }
    }
    for (int g = 0; g < ng; g++)
      if (tmp[graph[g]] == 1) cf[graph[g]] = C_PT;
    for (int g = 0; g < ng; g++) { /* H1 */
      const int i = graph[g];
      if (cf[i] != C_PT) continue;
      for (obig k = Sia[i]; k < Sia[i + 1]; k++) {
        if (gone[k]) continue;
        gone[k] = 1;
        if (cf[Sja[k]] == 0) measure[Sja[k]] -= 1.0;
      }
    }
    for (int g = 0; g < ng; g++) { /* H2 */
      const int i = graph[g];
      if (cf[i] != 0) continue;
      for (obig k = Sia[i]; k < Sia[i + 1]; k++)
        if (cf[Sja[k]] == C_PT) {
          gone[k] = 1;
          common[Sja[k]] = i + 1;
        }
      for (obig k = Sia[i]; k < Sia[i + 1]; k++) {
        const int j = Sja[k];
        if (gone[k] || cf[j] != 0) continue;
        for (obig kk = Sia[j]; kk < Sia[j + 1]; kk++)
          if (common[Sja[kk]] == i + 1) {
            gone[k] = 1;
            measure[j] -= 1.0;
            break;
          }
      }
    }
    int m = 0;
    for (int g = 0; g < ng; g++) {
      const int i = graph[g];
      if (cf[i] == C_PT) continue;
      if (measure[i] < 1.0)
        cf[i] = F_PT;
      else
        graph[m++] = i;
    }
    ng = m;
  }
  free(gone);
  free(measure);
  free(graph);
  free(tmp);
  free(common);
}

static void pmis(int n, const obig *Sia, const int *Sja, const int *part_of, int nparts, const obig *ps, int *cf);
static int coarsen_by_type(int type, int n, const obig *Sia, const int *Sja, int *cf) {
  if (type == 0 || type == 7) {
    cljp(n, Sia, Sja, cf);
    return 0;
  }
  if (type == 8 || type == 9) { /* 9 = PMIS with one global random stream: what 8 is here anyway */
    int *one_part = (int *)xcalloc((size_t)n, sizeof(int));
    pmis(n, Sia, Sja, one_part, 1, NULL, cf);
    free(one_part);
    return 0;
  }
  if (type == 10 || type == 11) {
    ruge_stueben(n, Sia, Sja, 0, cf);
    return 0;
  }
  if (type == 6 || type == 1 || type == 3) {
    ruge_stueben(n, Sia, Sja, 1, cf);
    return 0;
  }
  return -1;
}

/* PMIS started from a given splitting (par_coarsen.c hypre_BoomerAMGCoarsenPMIS with CF_init = 1; the second half
 * of HMIS): points marked C stay C unless they are BOUNDARY points -- rows with a strong connection into another
 * part -- which, like every F point, become undecided again; the kept C points act as the first independent set
 * (undecided points that strongly depend on one of them become F before the first selection), then the ordinary
 * PMIS rounds run on what is left, on the GLOBAL graph with the one global random stream. */
static void pmis_from(int n, const obig *Sia, const int *Sja, const int *part_of, int *cf) {
  double *measure = (double *)xcalloc((size_t)n, sizeof(double));
  for (int i = 0; i < n; i++)
    for (obig k = Sia[i]; k < Sia[i + 1]; k++) measure[Sja[k]] += 1.0;
  oracle_rand_seed(2747);
  for (int i = 0; i < n; i++) measure[i] += oracle_rand();
  int *graph = (int *)xmalloc(sizeof(int) * (size_t)n);
  int *tmp = (int *)xmalloc(sizeof(int) * (size_t)n);
  int ng = 0;
  for (int i = 0; i < n; i++) {
    int boundary = 0;
    for (obig k = Sia[i]; k < Sia[i + 1]; k++)
      if (part_of[Sja[k]] != part_of[i]) boundary = 1;
    if (Sia[i + 1] == Sia[i]) {
      cf[i] = SF_PT;
      measure[i] = 0.0;
    } else if (cf[i] == C_PT && !boundary) {
      measure[i] = 0.0; /* kept */
    } else if (measure[i] < 1.0) {
      cf[i] = F_PT;
      measure[i] = 0.0;
    } else
      cf[i] = 0;
  }
  for (int i = 0; i < n; i++) { /* the kept C points are the first independent set */
    if (cf[i] != 0) continue;
    int dep = 0;
    for (obig k = Sia[i]; k < Sia[i + 1] && !dep; k++) dep = cf[Sja[k]] == C_PT;
    if (dep) tmp[i] = 2;
    else tmp[i] = 3;
  }
  for (int i = 0; i < n; i++) {
    if (cf[i] != 0) continue;
    if (tmp[i] == 2) {
      cf[i] = F_PT;
      measure[i] = 0.0;
    } else
      graph[ng++] = i;
  }
  while (ng > 0) {
    for (int g = 0; g < ng; g++) tmp[graph[g]] = 1;
    for (int g = 0; g < ng; g++) {
      int i = graph[g];
      for (obig k = Sia[i]; k < Sia[i + 1]; k++) {
        int j = Sja[k];
        if (cf[j] != 0) continue;
        if (measure[i] > measure[j])
          tmp[j] = 0;
        else if (measure[j] > measure[i])
          tmp[i] = 0;
      }
    }
    for (int g = 0; g < ng; g++) {
      int i = graph[g];
      if (tmp[i] == 1) {
        cf[i] = C_PT;
        measure[i] = 0.0;
      }
    }
    for (int g = 0; g < ng; g++) {
      int i = graph[g];
      if (cf[i] != 0) continue;
      for (obig k = Sia[i]; k < Sia[i + 1]; k++)
        if (cf[Sja[k]] == C_PT) {
          tmp[i] = 2;
          break;
        }
    }
    int m = 0;
    for (int g = 0; g < ng; g++) {
      int i = graph[g];
      if (cf[i] != 0) continue;
      if (tmp[i] == 2) {
        cf[i] = F_PT;
        measure[i] = 0.0;
      } else
        graph[m++] = i;
    }
    ng = m;
  }
  free(measure);
  free(graph);
  free(tmp);
}

/* CLJP started from a given splitting (the second half of Falgout coarsening, coarsen_type 6, on more than one part:
 * "Ruge-Stueben in the interior of every processor, then CLJP on the boundary with the interior C points as the first
 * independent set").  HYPRE's own initialisation of hypre_BoomerAMGCoarsen with CF_init = 1 is not available here
 * (parity unpinned like everything HYPRE-side); this is the rule both implementations share: INTERIOR points -- rows
 * without a strong connection into another part -- keep the verdict of the per-part Ruge-Stueben passes, BOUNDARY
 * points are decided by CLJP.  Decided points no longer vote: before the first selection the edges out of every
 * interior row leave the graph (one decrement at each undecided end -- heuristic H1 applied to the kept points), and
 * heuristic H2 runs once over the undecided rows with the kept C points.  Then the rounds of cljp above. */
static void cljp_from(int n, const obig *Sia, const int *Sja, const int *part_of, int *cf) {
  const obig nnz = Sia[n];
  char *gone = (char *)xcalloc((size_t)(nnz ? nnz : 1), 1);
  double *measure = (double *)xcalloc((size_t)n, sizeof(double));
  for (obig k = 0; k < nnz; k++) measure[Sja[k]] += 1.0;
  oracle_rand_seed(2747);
  for (int i = 0; i < n; i++) measure[i] += oracle_rand();
  int *graph = (int *)xmalloc(sizeof(int) * (size_t)n);
  int *tmp = (int *)xcalloc((size_t)n, sizeof(int));
  int *common = (int *)xcalloc((size_t)n, sizeof(int));
  char *interior = (char *)xcalloc((size_t)(n ? n : 1), 1);
  int ng = 0;
  for (int i = 0; i < n; i++) {
    int boundary = 0;
    for (obig k = Sia[i]; k < Sia[i + 1]; k++)
      if (part_of[Sja[k]] != part_of[i]) boundary = 1;
    if (!boundary) {
      interior[i] = 1; /* cf[i] stays what the Ruge-Stueben passes made it */
    } else if (measure[i] < 1.0)
      cf[i] = F_PT; /* (a boundary row is not empty) */
    else {
      cf[i] = 0;
      graph[ng++] = i;
    }
  }
  for (int i = 0; i < n; i++) { /* decided points no longer vote */
    if (!interior[i]) continue;
    for (obig k = Sia[i]; k < Sia[i + 1]; k++) {
      gone[k] = 1;
      if (cf[Sja[k]] == 0) measure[Sja[k]] -= 1.0;
    }
  }
  for (int round = 0;; round++) {
    if (round > 0) {
      if (ng == 0) break;
      for (int g = 0; g < ng; g++) tmp[graph[g]] = 1;
      for (int g = 0; g < ng; g++) {
        const int i = graph[g];
        for (obig k = Sia[i]; k < Sia[i + 1]; k++) {
          const int j = Sja[k];
          if (cf[j] != 0) continue;
          if (measure[i] > measure[j])
            tmp[j] = 0;
          else if (measure[j] > measure[i])
            tmp[i] = 0;
        }
      }
      for (int g = 0; g < ng; g++)
        if (tmp[graph[g]] == 1) cf[graph[g]] = C_PT;
      for (int g = 0; g < ng; g++) { /* H1 */
        const int i = graph[g];
        if (cf[i] != C_PT) continue;
        for (obig k = Sia[i]; k < Sia[i + 1]; k++) {
          if (gone[k]) continue;
          gone[k] = 1;
          if (cf[Sja[k]] == 0) measure[Sja[k]] -= 1.0;
        }
      }
    }
    for (int g = 0; g < ng; g++) { /* H2 (round 0: with the kept C points) */
      const int i = graph[g];
      if (cf[i] != 0) continue;
      for (obig k = Sia[i]; k < Sia[i + 1]; k++)
        if (cf[Sja[k]] == C_PT) {
          gone[k] = 1;
          common[Sja[k]] = i + 1;
        }
      for (obig k = Sia[i]; k < Sia[i + 1]; k++) {
        const int j = Sja[k];
        if (gone[k] || cf[j] != 0) continue;
        for (obig kk = Sia[j]; kk < Sia[j + 1]; kk++)
          if (common[Sja[kk]] == i + 1) {
            gone[k] = 1;
            measure[j] -= 1.0;
            break;
          }
      }
    }
    int m = 0;
    for (int g = 0; g < ng; g++) {
      const int i = graph[g];
      if (cf[i] == C_PT) continue;
      if (measure[i] < 1.0)
        cf[i] = F_PT;
      else
        graph[m++] = i;
    }
    ng = m;
  }
  free(gone);
  free(measure);
  free(graph);
  free(tmp);
  free(common);
  free(interior);
}

/* The coarsening types HYPRE defines PER PROCESSOR, on more than one part (par_coarsen.c hypre_BoomerAMGCoarsenRuge /
 * hypre_BoomerAMGCoarsenHMIS): 11 = first Ruge-Stueben pass and 1 = both passes on every part's own graph (the strong
 * connections inside the part; "no boundary treatment"), 10 = HMIS = the first pass per part, then PMIS from that
 * state on the global graph (pmis_from), 6 = Falgout = both passes per part, then CLJP on the boundary points from
 * that state (cljp_from).  On one part these are the global routines above.  Type 3 (a third pass on the boundary)
 * keeps its global form here; PMIS and CLJP are global algorithms anyway. */
static int coarsen_by_type_parts(int type, int n, const obig *Sia, const int *Sja, const int *part_of, int nparts,
                                 const obig *ps, int *cf) {
  int used = 0;
  for (int q = 0; q < nparts; q++) used += ps[q + 1] > ps[q];
  if (used <= 1 || !(type == 10 || type == 11 || type == 1 || type == 6)) return coarsen_by_type(type, n, Sia, Sja, cf);
  for (int q = 0; q < nparts; q++) {
    const int lo = (int)ps[q], m = (int)(ps[q + 1] - ps[q]);
    if (m == 0) continue;
    obig *ia = (obig *)xcalloc((size_t)m + 1, sizeof(obig));
    for (int i = 0; i < m; i++) {
      obig c = 0;
      for (obig k = Sia[lo + i]; k < Sia[lo + i + 1]; k++) c += (Sja[k] >= lo && Sja[k] < lo + m);
      ia[i + 1] = ia[i] + c;
    }
    int *ja = (int *)xmalloc(sizeof(int) * (size_t)(ia[m] ? ia[m] : 1));
    obig w = 0;
    for (int i = 0; i < m; i++)
      for (obig k = Sia[lo + i]; k < Sia[lo + i + 1]; k++)
        if (Sja[k] >= lo && Sja[k] < lo + m) ja[w++] = Sja[k] - lo;
    ruge_stueben(m, ia, ja, type == 1 || type == 6, cf + lo);
    free(ia);
    free(ja);
  }
  if (type == 10) pmis_from(n, Sia, Sja, part_of, cf);
  if (type == 6) cljp_from(n, Sia, Sja, part_of, cf);
  return 0;
}

/* Second-generation strength graph of aggressive coarsening (par_strength.c hypre_BoomerAMGCreate2ndS,
 * num_paths 1 = "A2"): a graph on the C points of the first coarsening -- C point i depends on C point j != i
 * iff j is in S_i or in S_k for some k in S_i (a strong path of length at most two).  Rows and columns are
 * coarse indices (running count of C points), columns ascending. */
static void second_strength(int n, const obig *Sia, const int *Sja, const int *cf, obig **S2ia_out, int **S2ja_out,
                            int *nc_out) {
  int *f2c = (int *)xmalloc(sizeof(int) * (size_t)n);
  int nc = 0;
  for (int i = 0; i < n; i++) f2c[i] = (cf[i] == C_PT) ? nc++ : -1;
  int *mark = (int *)xmalloc(sizeof(int) * (size_t)(nc ? nc : 1));
  for (int q = 0; q < nc; q++) mark[q] = -1;
  obig *ia = (obig *)xcalloc((size_t)nc + 1, sizeof(obig));
  obig cap = (obig)nc * 16 + 16, w = 0;
  int *ja = (int *)xmalloc(sizeof(int) * (size_t)cap);
  for (int i = 0; i < n; i++) {
    if (cf[i] != C_PT) continue;
    const int ci = f2c[i];
    const obig row0 = w;
#define S2_ADD(j)                                                \
  do {                                                           \
    const int cj_ = f2c[j];                                      \
    if (cj_ >= 0 && cj_ != ci && mark[cj_] != ci) {              \
      mark[cj_] = ci;                                            \
      if (w >= cap) {                                            \
        cap *= 2;                                                \
        ja = (int *)realloc(ja, sizeof(int) * (size_t)cap);      \
      }                                                          \
      ja[w++] = cj_;                                             \
    }                                                            \
  } while (0)
    for (obig k = Sia[i]; k < Sia[i + 1]; k++) {
      const int k1 = Sja[k];
      S2_ADD(k1);
      for (obig kk = Sia[k1]; kk < Sia[k1 + 1]; kk++) S2_ADD(Sja[kk]);
    }
#undef S2_ADD
    qsort(ja + row0, (size_t)(w - row0), sizeof(int), cmp_int);
    ia[ci + 1] = w;
  }
  free(mark);
  free(f2c);
  *S2ia_out = ia;
  *S2ja_out = ja;
  *nc_out = nc;
}

/* Aggressive coarsening of one level (par_amg_setup.c, level < agg_num_levels; src/HypreSystem.cpp:215-219):
 * coarsen with S, coarsen the resulting C points again with the second-generation graph, and keep as C only
 * what survives both (hypre_BoomerAMGCorrectCFMarker); a first-stage C point the second stage rejects takes the
 * second stage's verdict (F, or special F when it has no second-generation connection). */
static int coarsen_aggressive(int type, int n, const obig *Sia, const int *Sja, const int *part_of, int nparts,
                              const obig *ps, int *cf) {
  if (coarsen_by_type_parts(type, n, Sia, Sja, part_of, nparts, ps, cf)) return -1;
  obig *S2ia;
  int *S2ja, nc;
  second_strength(n, Sia, Sja, cf, &S2ia, &S2ja, &nc);
  int *cf2 = (int *)xmalloc(sizeof(int) * (size_t)(nc ? nc : 1));
  /* the first-stage C points keep their owners: partition of the second-generation graph */
  obig *ps2 = (obig *)xcalloc((size_t)nparts + 1, sizeof(obig));
  int *part2 = (int *)xmalloc(sizeof(int) * (size_t)(nc ? nc : 1));
  {
    int q2 = 0;
    for (int i = 0; i < n; i++)
      if (cf[i] == C_PT) {
        part2[q2++] = part_of[i];
        ps2[part_of[i] + 1]++;
      }
    for (int q = 0; q < nparts; q++) ps2[q + 1] += ps2[q];
  }
  coarsen_by_type_parts(type, nc, S2ia, S2ja, part2, nparts, ps2, cf2);
  free(ps2);
  free(part2);
  int q = 0;
  for (int i = 0; i < n; i++)
    if (cf[i] == C_PT) {
      if (cf2[q] != C_PT) cf[i] = cf2[q];
      q++;
    }
  free(cf2);
  free(S2ia);
  free(S2ja);
  return 0;
}

/* Truncation (par_interp.c hypre_BoomerAMGInterpTruncation): drop entries below
 * trunc_factor*max|p|, keep the pmax largest by (|p| desc, position asc), and
 * rescale the row to its original sum.  Kept entries stay in stored order. */
static int truncate_row(int len, int *cols, double *vals, double trunc_factor, int pmax, int *keep_scratch) {
  if (len == 0) return 0;
  double row_sum = 0.0, maxabs = 0.0;
  for (int k = 0; k < len; k++) {
    row_sum += vals[k];
    if (fabs(vals[k]) > maxabs) maxabs = fabs(vals[k]);
  }
  int *keep = keep_scratch;
  for (int k = 0; k < len; k++) keep[k] = (trunc_factor > 0.0) ? (fabs(vals[k]) >= trunc_factor * maxabs) : 1;
  int nk = 0;
  for (int k = 0; k < len; k++) nk += keep[k];
  if (pmax > 0)
    while (nk > pmax) { /* drop the smallest |p|; ties: the later position goes first */
      int worst = -1;
      for (int k = 0; k < len; k++)
        if (keep[k] && (worst < 0 || fabs(vals[k]) <= fabs(vals[worst]))) worst = k;
      keep[worst] = 0;
      nk--;
    }
  double kept_sum = 0.0;
  for (int k = 0; k < len; k++)
    if (keep[k]) kept_sum += vals[k];
  double scale = (kept_sum != 0.0) ? row_sum / kept_sum : 1.0;
  int m = 0;
  for (int k = 0; k < len; k++)
    if (keep[k]) {
      cols[m] = cols[k];
      vals[m] = vals[k] * scale;
      m++;
    }
  return m;
}

/* Interpolation.  interp_type 6: extended+i (par_lr_interp.c
 * hypre_BoomerAMGBuildExtPIInterp, De Sterck/Falgout/Nolting/Yang 2008);
 * 3: direct (par_interp.c hypre_BoomerAMGBuildDirInterp);
 * 0: classical modified (par_interp.c hypre_BoomerAMGBuildInterp).
 * (Called with a single partition: interpolation is a global algorithm.)
 * Columns of P are coarse indices (fine_to_coarse = running count of C points). */
static ocsr *build_interp(const ocsr *A, const obig *Sia, const int *Sja, int *cf, const int *part_of, int interp_type,
                          double trunc_factor, int pmax, int *ncoarse_out) {
  const int n = A->nrows;
  int *f2c = (int *)xmalloc(sizeof(int) * (size_t)n);
  int nc = 0;
  for (int i = 0; i < n; i++) f2c[i] = (cf[i] == C_PT) ? nc++ : -1;
  *ncoarse_out = nc;
  obig *pia = (obig *)xcalloc((size_t)n + 1, sizeof(obig));
  /* rows are independent: each thread owns a contiguous range of rows (static schedule), its own marker array and
   * its own output buffer; the buffers are concatenated in thread order afterwards */
  const int nthr = setup_threads();
  int **tpj = (int **)xcalloc((size_t)nthr, sizeof(int *));
  double **tpa = (double **)xcalloc((size_t)nthr, sizeof(double *));
  obig *tq = (obig *)xcalloc((size_t)nthr, sizeof(obig));
  int *tfirst = (int *)xmalloc(sizeof(int) * (size_t)nthr);
  for (int t = 0; t < nthr; t++) tfirst[t] = -1;
#pragma omp parallel num_threads(nthr) if (nthr > 1)
  {
  const int tid = OTHREAD();
  int *Pmark = (int *)xmalloc(sizeof(int) * (size_t)n); /* position of fine col in current row, or marker */
  for (int i = 0; i < n; i++) Pmark[i] = -1;
  /* growing buffers */
  obig cap = (obig)n * 4 / nthr + 64, q = 0;
  int *pj = (int *)xmalloc(sizeof(int) * (size_t)cap);
  double *pa = (double *)xmalloc(sizeof(double) * (size_t)cap);
  int rowcap = 64;
  int *rc = (int *)xmalloc(sizeof(int) * (size_t)rowcap);
  double *rv = (double *)xmalloc(sizeof(double) * (size_t)rowcap);
  int *keep = (int *)xmalloc(sizeof(int) * (size_t)rowcap);
  int strong_f_marker = -2;
#define GROW_ROW()                                            \
  do {                                                        \
    if (len >= rowcap) {                                      \
      rowcap *= 2;                                            \
      rc = (int *)realloc(rc, sizeof(int) * (size_t)rowcap);  \
      rv = (double *)realloc(rv, sizeof(double) * (size_t)rowcap); \
      keep = (int *)realloc(keep, sizeof(int) * (size_t)rowcap);   \
    }                                                         \
  } while (0)
#pragma omp for schedule(static)
  for (int i = 0; i < n; i++) {
    int len = 0;
    if (tfirst[tid] < 0) tfirst[tid] = i;
    if (cf[i] == C_PT) {
      rc[0] = i;
      rv[0] = 1.0;
      len = 1;
    } else if (cf[i] != SF_PT) {
      const int mypart = part_of[i];
      strong_f_marker--;
      /* ---- interpolatory set: rc[0..len) are FINE indices for now */
      for (obig k = Sia[i]; k < Sia[i + 1]; k++) {
        int i1 = Sja[k];
        if (part_of[i1] != mypart) continue;
        if (cf[i1] == C_PT) {
          if (Pmark[i1] < 0) {
            GROW_ROW();
            Pmark[i1] = len;
            rc[len] = i1;
            rv[len] = 0.0;
            len++;
          }
        } else if (cf[i1] != SF_PT && interp_type == 6) {
          Pmark[i1] = strong_f_marker;
          for (obig kk = Sia[i1]; kk < Sia[i1 + 1]; kk++) {
            int k1 = Sja[kk];
            if (part_of[k1] != mypart || cf[k1] != C_PT) continue;
            if (Pmark[k1] < 0) {
              GROW_ROW();
              Pmark[k1] = len;
              rc[len] = k1;
              rv[len] = 0.0;
              len++;
            }
          }
        } else if (cf[i1] != SF_PT && interp_type == 0) {
          Pmark[i1] = strong_f_marker;
        }
      }
#define IN_SET(j) (Pmark[j] >= 0)
      double diagonal = 0.0;
      for (obig k = A->ia[i]; k < A->ia[i + 1]; k++)
        if (A->ja[k] == i) diagonal = A->a[k];
      if (interp_type == 3) {
        /* direct interpolation */
        double sum_N_pos = 0, sum_N_neg = 0, sum_P_pos = 0, sum_P_neg = 0;
        for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) {
          int j = A->ja[k];
          if (j == i) continue;
          double v = A->a[k];
          if (v > 0)
            sum_N_pos += v;
          else
            sum_N_neg += v;
          if (IN_SET(j)) {
            rv[Pmark[j]] += v;
            if (v > 0)
              sum_P_pos += v;
            else
              sum_P_neg += v;
          }
        }
        double alfa = 1.0, beta = 1.0;
        if (sum_P_neg != 0) alfa = sum_N_neg / sum_P_neg / diagonal;
        if (sum_P_pos != 0) beta = sum_N_pos / sum_P_pos / diagonal;
        if (sum_P_pos == 0) {
          /* no positive C connections: lump positive entries into the diagonal */
          double d2 = diagonal + sum_N_pos;
          if (sum_P_neg != 0) alfa = sum_N_neg / sum_P_neg / d2;
          beta = 0.0;
        }
        for (int k = 0; k < len; k++) rv[k] = (rv[k] > 0) ? -beta * rv[k] : -alfa * rv[k];
      } else {
        for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) {
          int i1 = A->ja[k];
          if (i1 == i) continue;
          double aik = A->a[k];
          if (IN_SET(i1)) {
            rv[Pmark[i1]] += aik;
          } else if (Pmark[i1] == strong_f_marker) {
            /* strong F neighbour: distribute a_ik over the set through row i1 */
            double dk = 0.0;
            for (obig kk = A->ia[i1]; kk < A->ia[i1 + 1]; kk++)
              if (A->ja[kk] == i1) dk = A->a[kk];
            double sgn = (dk < 0) ? -1.0 : 1.0;
            double sum = 0.0;
            for (obig kk = A->ia[i1]; kk < A->ia[i1 + 1]; kk++) {
              int i2 = A->ja[kk];
              if (i2 == i1) continue;
              if ((IN_SET(i2) || (interp_type == 6 && i2 == i)) && sgn * A->a[kk] < 0) sum += A->a[kk];
            }
            if (sum != 0.0) {
              double distribute = aik / sum;
              for (obig kk = A->ia[i1]; kk < A->ia[i1 + 1]; kk++) {
                int i2 = A->ja[kk];
                if (i2 == i1) continue;
                if (sgn * A->a[kk] < 0) {
                  if (IN_SET(i2))
                    rv[Pmark[i2]] += distribute * A->a[kk];
                  else if (interp_type == 6 && i2 == i)
                    diagonal += distribute * A->a[kk];
                }
              }
            } else
              diagonal += aik;
          } else {
            /* weak neighbour (or any off-partition neighbour) */
            diagonal += aik;
          }
        }
        if (diagonal != 0.0)
          for (int k = 0; k < len; k++) rv[k] /= -diagonal;
      }
#undef IN_SET
      for (int k = 0; k < len; k++) Pmark[rc[k]] = -1;
      len = truncate_row(len, rc, rv, trunc_factor, pmax, keep);
    }
    if (q + len > cap) {
      cap = (q + len) * 2;
      pj = (int *)realloc(pj, sizeof(int) * (size_t)cap);
      pa = (double *)realloc(pa, sizeof(double) * (size_t)cap);
    }
    /* sort the row by coarse column, insertion sort (rows are short) */
    for (int k = 0; k < len; k++) rc[k] = f2c[rc[k]];
    for (int a = 1; a < len; a++) {
      int c = rc[a];
      double v = rv[a];
      int b = a - 1;
      while (b >= 0 && rc[b] > c) {
        rc[b + 1] = rc[b];
        rv[b + 1] = rv[b];
        b--;
      }
      rc[b + 1] = c;
      rv[b + 1] = v;
    }
    for (int k = 0; k < len; k++) {
      pj[q] = rc[k];
      pa[q] = rv[k];
      q++;
    }
    pia[i + 1] = len;
  }
#undef GROW_ROW
  tpj[tid] = pj;
  tpa[tid] = pa;
  tq[tid] = q;
  free(Pmark);
  free(rc);
  free(rv);
  free(keep);
  } /* parallel */
  for (int i = 0; i < n; i++) pia[i + 1] += pia[i];
  ocsr *P = (ocsr *)xmalloc(sizeof(ocsr));
  P->nrows = n;
  P->ncols = nc;
  P->ia = pia;
  P->ja = (int *)xmalloc(sizeof(int) * (size_t)(pia[n] ? pia[n] : 1));
  P->a = (double *)xmalloc(sizeof(double) * (size_t)(pia[n] ? pia[n] : 1));
  for (int t = 0; t < nthr; t++) {
    if (tfirst[t] >= 0 && tq[t] > 0) {
      memcpy(P->ja + pia[tfirst[t]], tpj[t], sizeof(int) * (size_t)tq[t]);
      memcpy(P->a + pia[tfirst[t]], tpa[t], sizeof(double) * (size_t)tq[t]);
    }
    free(tpj[t]);
    free(tpa[t]);
  }
  free(tpj);
  free(tpa);
  free(tq);
  free(tfirst);
  /* special F points act as plain F points from here on (par_amg_setup.c) */
  for (int i = 0; i < n; i++)
    if (cf[i] == SF_PT) cf[i] = F_PT;
  free(f2c);
  return P;
}


/* Multipass interpolation (par_multi_interp.c hypre_BoomerAMGBuildMultipass; Stueben 1999, Yang 2010) -- the
 * interpolation of aggressive-coarsening levels, agg_interp_type 4 (library default; src/HypreSystem.cpp:220-224).
 * Pass 1: F points with a strong C neighbour interpolate directly from their strong C neighbours,
 *   w_ij = -(sum_N / sum_C) a_ij / a_ii,  sum_N = all off-diagonal entries of row i, sum_C = those to C_i^s.
 * Pass k > 1: F points not reached yet with a strong neighbour reached in pass k-1 interpolate THROUGH those
 * neighbours j (their rows w_j. are final):  w_i. = -(sum_N / sum_J) / a_ii * sum_j a_ij w_j. ,  sum_J = sum of
 * the a_ij used.  Points never reached (and special F points) keep an empty row.  Rows are accumulated in
 * discovery order (A's row, then the neighbour's P row), truncated (agg_trunc_factor, agg_P_max_elmts) after
 * the last pass, then sorted by coarse column. */
static ocsr *build_multipass(const ocsr *A, const obig *Sia, const int *Sja, int *cf, double trunc_factor, int pmax,
                             int *ncoarse_out) {
  const int n = A->nrows;
  int *f2c = (int *)xmalloc(sizeof(int) * (size_t)n);
  int nc = 0;
  for (int i = 0; i < n; i++) f2c[i] = (cf[i] == C_PT) ? nc++ : -1;
  *ncoarse_out = nc;
  int *assigned = (int *)xmalloc(sizeof(int) * (size_t)n); /* pass number; 0 = C, -1 = not reached */
  int *rlen = (int *)xcalloc((size_t)n, sizeof(int));
  int **rcol = (int **)xcalloc((size_t)n, sizeof(int *)); /* per row: coarse columns / weights, discovery order */
  double **rval = (double **)xcalloc((size_t)n, sizeof(double *));
  int remaining = 0;
  for (int i = 0; i < n; i++) {
    if (cf[i] == C_PT) {
      assigned[i] = 0;
      rcol[i] = (int *)xmalloc(sizeof(int));
      rval[i] = (double *)xmalloc(sizeof(double));
      rcol[i][0] = f2c[i];
      rval[i][0] = 1.0;
      rlen[i] = 1;
    } else {
      assigned[i] = -1;
      if (cf[i] != SF_PT) remaining++;
    }
  }
  int *smark = (int *)xmalloc(sizeof(int) * (size_t)n);   /* strong neighbour of the previous pass, of row i */
  int *cpos = (int *)xmalloc(sizeof(int) * (size_t)(nc ? nc : 1)); /* coarse column -> position in the row */
  for (int i = 0; i < n; i++) smark[i] = -1;
  for (int q = 0; q < nc; q++) cpos[q] = -1;
  int cap = 64;
  int *tc = (int *)xmalloc(sizeof(int) * (size_t)cap);
  double *tv = (double *)xmalloc(sizeof(double) * (size_t)cap);
  int *list = (int *)xmalloc(sizeof(int) * (size_t)n);
  for (int pass = 1; remaining > 0; pass++) {
    int nl = 0;
    for (int i = 0; i < n; i++) {
      if (assigned[i] != -1 || cf[i] == SF_PT) continue;
      for (obig k = Sia[i]; k < Sia[i + 1]; k++)
        if (assigned[Sja[k]] == pass - 1) {
          list[nl++] = i;
          break;
        }
    }
    if (nl == 0) break; /* the rest cannot be reached along strong connections */
    for (int t = 0; t < nl; t++) {
      const int i = list[t];
      for (obig k = Sia[i]; k < Sia[i + 1]; k++)
        if (assigned[Sja[k]] == pass - 1) smark[Sja[k]] = i;
      double diagonal = 0.0, sum_N = 0.0, sum_J = 0.0;
      int len = 0;
      for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) {
        const int j = A->ja[k];
        if (j == i) {
          diagonal = A->a[k];
          continue;
        }
        sum_N += A->a[k];
        if (smark[j] != i) continue;
        sum_J += A->a[k];
        for (int q = 0; q < rlen[j]; q++) {
          const int c = rcol[j][q];
          if (cpos[c] < 0) {
            if (len >= cap) {
              cap *= 2;
              tc = (int *)realloc(tc, sizeof(int) * (size_t)cap);
              tv = (double *)realloc(tv, sizeof(double) * (size_t)cap);
            }
            cpos[c] = len;
            tc[len] = c;
            tv[len] = 0.0;
            len++;
          }
          tv[cpos[c]] += A->a[k] * rval[j][q];
        }
      }
      const double alfa = (sum_J * diagonal != 0.0) ? -sum_N / (sum_J * diagonal) : 0.0;
      rcol[i] = (int *)xmalloc(sizeof(int) * (size_t)(len ? len : 1));
      rval[i] = (double *)xmalloc(sizeof(double) * (size_t)(len ? len : 1));
      for (int q = 0; q < len; q++) {
        rcol[i][q] = tc[q];
        rval[i][q] = tv[q] * alfa;
        cpos[tc[q]] = -1;
      }
      rlen[i] = len;
    }
    /* the pass is complete before its points count as reached */
    for (int t = 0; t < nl; t++) assigned[list[t]] = pass;
    remaining -= nl;
  }
  /* truncation + sort, row by row */
  obig *pia = (obig *)xcalloc((size_t)n + 1, sizeof(obig));
  obig tot = 0;
  int maxlen = 1;
  for (int i = 0; i < n; i++) {
    tot += rlen[i];
    if (rlen[i] > maxlen) maxlen = rlen[i];
  }
  int *pj = (int *)xmalloc(sizeof(int) * (size_t)(tot ? tot : 1));
  double *pa = (double *)xmalloc(sizeof(double) * (size_t)(tot ? tot : 1));
  int *keep = (int *)xmalloc(sizeof(int) * (size_t)maxlen);
  obig w = 0;
  for (int i = 0; i < n; i++) {
    int len = rlen[i];
    if (cf[i] != C_PT && len > 0) len = truncate_row(len, rcol[i], rval[i], trunc_factor, pmax, keep);
    for (int a = 1; a < len; a++) { /* insertion sort by coarse column */
      const int c = rcol[i][a];
      const double v = rval[i][a];
      int b = a - 1;
      while (b >= 0 && rcol[i][b] > c) {
        rcol[i][b + 1] = rcol[i][b];
        rval[i][b + 1] = rval[i][b];
        b--;
      }
      rcol[i][b + 1] = c;
      rval[i][b + 1] = v;
    }
    for (int q = 0; q < len; q++) {
      pj[w] = rcol[i][q];
      pa[w] = rval[i][q];
      w++;
    }
    pia[i + 1] = w;
    free(rcol[i]);
    free(rval[i]);
  }
  ocsr *P = (ocsr *)xmalloc(sizeof(ocsr));
  P->nrows = n;
  P->ncols = nc;
  P->ia = pia;
  P->ja = pj;
  P->a = pa;
  for (int i = 0; i < n; i++)
    if (cf[i] == SF_PT) cf[i] = F_PT;
  free(f2c);
  free(assigned);
  free(rlen);
  free(rcol);
  free(rval);
  free(smark);
  free(cpos);
  free(tc);
  free(tv);
  free(list);
  free(keep);
  return P;
}

/* Non-Galerkin coarse operator (HYPRE_BoomerAMGSetNonGalerkinTol, src/HypreSystem.cpp:161-176).  HYPRE's
 * par_nongalerkn.c (Falgout / Schroder 2014) is NOT restated line by line -- it is not in the reference tree and
 * the recollection of its lumping onto strong neighbours is not reliable enough; this is the documented simplified
 * form both implementations share: with m_i = max_{j != i} |a_ij|, an off-diagonal entry is dropped iff
 * |a_ij| < tol * min(m_i, m_j) -- small against BOTH rows, so a symmetric operator stays symmetric -- and every
 * dropped entry is added to its row's diagonal in stored order (row sums, i.e. the action on constants, are kept).
 * Kept entries stay in stored order. */
static ocsr *sparsify_non_galerkin(const ocsr *A, double tol) {
  const int n = A->nrows;
  double *m = (double *)xcalloc((size_t)(n ? n : 1), sizeof(double));
  for (int i = 0; i < n; i++) {
    double mx = 0.0;
    for (obig k = A->ia[i]; k < A->ia[i + 1]; k++)
      if (A->ja[k] != i && fabs(A->a[k]) > mx) mx = fabs(A->a[k]);
    m[i] = mx;
  }
  obig *ia = (obig *)xcalloc((size_t)n + 1, sizeof(obig));
  for (int i = 0; i < n; i++) {
    obig c = 0;
    for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) {
      const int j = A->ja[k];
      const double lim = tol * (m[i] < m[j] ? m[i] : m[j]);
      if (j == i || !(fabs(A->a[k]) < lim)) c++;
    }
    ia[i + 1] = ia[i] + c;
  }
  ocsr *B = ocsr_new(n, A->ncols, ia[n]);
  memcpy(B->ia, ia, sizeof(obig) * ((size_t)n + 1));
  free(ia);
  for (int i = 0; i < n; i++) {
    obig w = B->ia[i], dpos = -1;
    double lump = 0.0;
    int first = 1;
    for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) {
      const int j = A->ja[k];
      const double lim = tol * (m[i] < m[j] ? m[i] : m[j]);
      if (j == i || !(fabs(A->a[k]) < lim)) {
        if (j == i) dpos = w;
        B->ja[w] = j;
        B->a[w++] = A->a[k];
      } else {
        lump = first ? A->a[k] : lump + A->a[k];
        first = 0;
      }
    }
    if (!first && dpos >= 0) B->a[dpos] = B->a[dpos] + lump;
  }
  free(m);
  return B;
}

/* l1 norms.  l1gs: hypre_ParCSRComputeL1NormsThreads option 4 (par_relax_more.c),
 * "threads" = hybrid-GS chunks: |a_ii| + 0.5*sum |a_ij| over entries outside the
 * row's chunk (incl. other partitions) whose C/F type equals the row's (all of
 * them when cf == NULL), truncated to a_ii when <= 4/3 a_ii (Remark 6.2 of
 * Baker/Falgout/Kolev/Yang 2011); negative diagonal flips the sign.
 * l1jac: full row l1 norm (option 1), used by relax 18. */
static void level_norms(olevel *L, int chunk, int nparts) {
  const ocsr *A = L->A;
  const int n = A->nrows;
  L->diag = (double *)xcalloc((size_t)n, sizeof(double));
  L->l1gs = (double *)xcalloc((size_t)n, sizeof(double));
  L->l1jac = (double *)xcalloc((size_t)n, sizeof(double));
  for (int p = 0; p < nparts; p++)
    for (obig cs = L->part_starts[p]; cs < L->part_starts[p + 1]; cs += chunk) {
      obig ce = cs + chunk;
      if (ce > L->part_starts[p + 1]) ce = L->part_starts[p + 1];
      for (obig i = cs; i < ce; i++) {
        double d = 0.0, l1 = 0.0, full = 0.0;
        for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) {
          int j = A->ja[k];
          double av = fabs(A->a[k]);
          full += av;
          if (j == i) {
            d = A->a[k];
            l1 += av;
          } else if (j < cs || j >= ce) {
            if (!L->cf || L->cf[j] == L->cf[i]) l1 += 0.5 * av;
          }
        }
        if (l1 <= 4.0 / 3.0 * fabs(d)) l1 = fabs(d);
        if (d < 0) {
          l1 = -l1;
          full = -full;
        }
        L->diag[i] = d;
        L->l1gs[i] = l1;
        L->l1jac[i] = full;
      }
    }
}

static void dense_inverse(const ocsr *A, double *inv) {
  /* Gauss-Jordan with partial pivoting; relax type 9 (par_relax.c / par_gauss_elim) */
  const int n = A->nrows;
  double *M = (double *)xcalloc((size_t)n * n, sizeof(double));
  for (int i = 0; i < n; i++)
    for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) M[(size_t)i * n + A->ja[k]] = A->a[k];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) inv[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int c = 0; c < n; c++) {
    int piv = c;
    for (int r = c + 1; r < n; r++)
      if (fabs(M[(size_t)r * n + c]) > fabs(M[(size_t)piv * n + c])) piv = r;
    if (piv != c)
      for (int j = 0; j < n; j++) {
        double t = M[(size_t)c * n + j];
        M[(size_t)c * n + j] = M[(size_t)piv * n + j];
        M[(size_t)piv * n + j] = t;
        t = inv[(size_t)c * n + j];
        inv[(size_t)c * n + j] = inv[(size_t)piv * n + j];
        inv[(size_t)piv * n + j] = t;
      }
    double d = M[(size_t)c * n + c];
    if (d == 0.0) continue;
    double id = 1.0 / d;
    for (int j = 0; j < n; j++) {
      M[(size_t)c * n + j] *= id;
      inv[(size_t)c * n + j] *= id;
    }
    for (int r = 0; r < n; r++) {
      if (r == c) continue;
      double f = M[(size_t)r * n + c];
      if (f == 0.0) continue;
      for (int j = 0; j < n; j++) {
        M[(size_t)r * n + j] -= f * M[(size_t)c * n + j];
        inv[(size_t)r * n + j] -= f * inv[(size_t)c * n + j];
      }
    }
  }
  free(M);
}

#define ORACLE_MAX_DENSE 4096

static void finish_levels(oamg *h) {
  for (int l = 0; l < h->nlev; l++) {
    olevel *L = &h->L[l];
    const int n = L->A->nrows;
    level_norms(L, h->p.gs_chunk, h->p.nparts);
    L->u = (double *)xcalloc((size_t)n, sizeof(double));
    L->f = (double *)xcalloc((size_t)n, sizeof(double));
    L->tmp = (double *)xcalloc((size_t)n, sizeof(double));
    L->old = (double *)xcalloc((size_t)n, sizeof(double));
    if (L->P) L->R = ocsr_transpose(L->P);
    /* par_amg_setup.c: HYPRE_ILUCreate/Setup per level j < smooth_num_levels (not the coarsest) */
    if (h->p.smooth_type == 5 && l < h->p.smooth_num_levels && l < h->nlev - 1)
      L->smoother = oilu_setup_k(L->A, h->p.nparts, L->part_starts, h->p.ilu_tri_solve, h->p.ilu_lower_it, h->p.ilu_upper_it,
                                 h->p.ilu_level);
  }
  olevel *Lc = &h->L[h->nlev - 1];
  if (h->p.relax_type[2] == 9 && Lc->A->nrows <= ORACLE_MAX_DENSE) {
    Lc->Cinv = (double *)xmalloc(sizeof(double) * (size_t)Lc->A->nrows * Lc->A->nrows);
    dense_inverse(Lc->A, Lc->Cinv);
  }
}


/* C-first ordering.  After the hierarchy is built in natural order, every level
 * with a C/F splitting is renumbered inside each partition: C points first (in
 * their original order, so that C point number q IS coarse unknown q of that
 * partition), F points after.  The smoother's chunks of 8 rows are then all-C or
 * all-F (bar one mixed chunk per partition), a C or F pass touches one contiguous
 * row range, and the two passes of a sweep together read the matrix once.
 * DESIGN.md section 3; the product applies the same renumbering. */
static ocsr *permute_csr(const ocsr *A, const int *rowpos, const int *rowperm, const int *colpos) {
  /* B[rowpos[i], colpos[j]] = A[i,j]; rowperm = inverse of rowpos; NULL maps = identity */
  const int n = A->nrows;
  ocsr *B = ocsr_new(n, A->ncols, ocsr_nnz(A));
  for (int q = 0; q < n; q++) {
    const int i = rowperm ? rowperm[q] : q;
    B->ia[q + 1] = B->ia[q] + (A->ia[i + 1] - A->ia[i]);
  }
#pragma omp parallel for schedule(static) num_threads(setup_threads()) if (setup_threads() > 1)
  for (int q = 0; q < n; q++) {
    const int i = rowperm ? rowperm[q] : q;
    const obig len = A->ia[i + 1] - A->ia[i];
    obig w = B->ia[q];
    for (obig k = A->ia[i]; k < A->ia[i + 1]; k++, w++) {
      B->ja[w] = colpos ? colpos[A->ja[k]] : A->ja[k];
      B->a[w] = A->a[k];
    }
    /* insertion sort of the row by new column (rows are short) */
    for (obig a = B->ia[q] + 1; a < B->ia[q] + len; a++) {
      int c = B->ja[a];
      double v = B->a[a];
      obig b = a - 1;
      while (b >= B->ia[q] && B->ja[b] > c) {
        B->ja[b + 1] = B->ja[b];
        B->a[b + 1] = B->a[b];
        b--;
      }
      B->ja[b + 1] = c;
      B->a[b + 1] = v;
    }
  }
  (void)rowpos;
  return B;
}

static void apply_cf_ordering(oamg *h) {
  const int nparts = h->p.nparts;
  int **pos = (int **)xcalloc((size_t)h->nlev, sizeof(int *));
  for (int l = 0; l < h->nlev; l++) {
    olevel *L = &h->L[l];
    if (!L->cf) continue;
    const int n = L->A->nrows;
    pos[l] = (int *)xmalloc(sizeof(int) * (size_t)n);
    L->perm = (int *)xmalloc(sizeof(int) * (size_t)n);
    for (int p = 0; p < nparts; p++) {
      obig q = L->part_starts[p];
      for (obig i = L->part_starts[p]; i < L->part_starts[p + 1]; i++)
        if (L->cf[i] == C_PT) pos[l][i] = (int)q++;
      for (obig i = L->part_starts[p]; i < L->part_starts[p + 1]; i++)
        if (L->cf[i] != C_PT) pos[l][i] = (int)q++;
    }
    for (int i = 0; i < n; i++) L->perm[pos[l][i]] = i;
  }
  for (int l = 0; l < h->nlev; l++) {
    olevel *L = &h->L[l];
    if (pos[l]) {
      ocsr *A2 = permute_csr(L->A, pos[l], L->perm, pos[l]);
      if (L->own_A) ocsr_free(L->A);
      L->A = A2;
      L->own_A = 1;
      const int n = A2->nrows;
      int *cf2 = (int *)xmalloc(sizeof(int) * (size_t)n);
      for (int q = 0; q < n; q++) cf2[q] = L->cf[L->perm[q]];
      free(L->cf);
      L->cf = cf2;
    }
    if (L->P && (pos[l] || (l + 1 < h->nlev && pos[l + 1]))) {
      ocsr *P2 = permute_csr(L->P, pos[l], L->perm, (l + 1 < h->nlev) ? pos[l + 1] : NULL);
      if (L->own_P) ocsr_free(L->P);
      L->P = P2;
      L->own_P = 1;
    }
  }
  for (int l = 0; l < h->nlev; l++) free(pos[l]);
  free(pos);
}

/* hypre_BoomerAMGSetup (par_amg_setup.c), reached from solverSetupPtr_
 * (src/HypreSystem.cpp:692) through HYPRE_ParCSRGMRESSetup -> precond setup. */
oamg *oamg_setup(const ocsr *A0, const oamg_params *p) {
  oamg *h = (oamg *)xcalloc(1, sizeof(oamg));
  h->p = *p;
  if (h->p.gs_chunk < 1) h->p.gs_chunk = 1;
  const int nparts = p->nparts > 0 ? p->nparts : 1;
  h->p.nparts = nparts;
  h->L = (olevel *)xcalloc((size_t)(p->max_levels > 0 ? p->max_levels : 1), sizeof(olevel));
  h->L[0].A = (ocsr *)A0;
  h->L[0].own_A = 0;
  h->L[0].part_starts = (obig *)xmalloc(sizeof(obig) * ((size_t)nparts + 1));
  if (p->part_starts)
    memcpy(h->L[0].part_starts, p->part_starts, sizeof(obig) * ((size_t)nparts + 1));
  else {
    h->L[0].part_starts[0] = 0;
    h->L[0].part_starts[1] = A0->nrows;
  }
  h->p.part_starts = NULL;
  double tph[6] = {0, 0, 0, 0, 0, 0};
  int l = 0;
  while (l < p->max_levels - 1 && h->L[l].A->nrows > p->max_coarse_size) {
    olevel *L = &h->L[l];
    const ocsr *A = L->A;
    const int n = A->nrows;
    obig *Sia;
    int *Sja;
    double tt = now_s();
    strength(A, p->strong_threshold, p->max_row_sum, &Sia, &Sja);
    tph[0] += now_s() - tt;
    tt = now_s();
    int *part_of = part_of_rows(n, nparts, L->part_starts);
    int *cf = (int *)xmalloc(sizeof(int) * (size_t)n);
    /* coarsening and interpolation are GLOBAL algorithms (independent of the row partition) -- but for the coarsening
     * types HYPRE defines per processor (10, 11, 1: coarsen_by_type_parts) */
    int *one_part = (int *)xcalloc((size_t)n, sizeof(int));
    const int aggressive = l < p->agg_num_levels;
    const int bad = aggressive ? coarsen_aggressive(p->coarsen_type, n, Sia, Sja, part_of, nparts, L->part_starts, cf)
                               : coarsen_by_type_parts(p->coarsen_type, n, Sia, Sja, part_of, nparts, L->part_starts, cf);
    if (bad || (aggressive && p->agg_interp_type != 4)) {
      fprintf(stderr, "oracle: coarsen_type %d / agg_interp_type %d is not restated\n", p->coarsen_type,
              p->agg_interp_type);
      abort();
    }
    tph[1] += now_s() - tt;
    tt = now_s();
    int nc = 0;
    for (int i = 0; i < n; i++) nc += (cf[i] == C_PT);
    if (nc == 0 || nc == n || nc < p->min_coarse_size) {
      free(Sia);
      free(Sja);
      free(part_of);
      free(one_part);
      free(cf);
      break;
    }
    int nc2;
    ocsr *P = aggressive ? build_multipass(A, Sia, Sja, cf, p->agg_trunc_factor, p->agg_pmax_elmts, &nc2)
              : p->interp_type == 4 /* multipass on an ordinary splitting */
                  ? build_multipass(A, Sia, Sja, cf, p->trunc_factor, p->pmax_elmts, &nc2)
                  : build_interp(A, Sia, Sja, cf, one_part, p->interp_type, p->trunc_factor, p->pmax_elmts, &nc2);
    free(one_part);
    L->cf = cf;
    L->P = P;
    L->own_P = 1;
    tph[2] += now_s() - tt;
    tt = now_s();
    ocsr *R = ocsr_transpose(P);
    ocsr *AP = ocsr_matmul(A, P);
    ocsr *Ac = ocsr_matmul(R, AP);
    if (p->non_galerkin_num_tol > 0 && p->non_galerkin_tol) {
      const int q = l < p->non_galerkin_num_tol ? l : p->non_galerkin_num_tol - 1;
      if (p->non_galerkin_tol[q] > 0.0) {
        ocsr *As = sparsify_non_galerkin(Ac, p->non_galerkin_tol[q]);
        ocsr_free(Ac);
        Ac = As;
      }
    }
    tph[3] += now_s() - tt;
    ocsr_free(AP);
    ocsr_free(R);
    olevel *Ln = &h->L[l + 1];
    Ln->A = Ac;
    Ln->own_A = 1;
    Ln->part_starts = (obig *)xcalloc((size_t)nparts + 1, sizeof(obig));
    for (int i = 0; i < n; i++)
      if (cf[i] == C_PT) Ln->part_starts[part_of[i] + 1]++;
    for (int q = 0; q < nparts; q++) Ln->part_starts[q + 1] += Ln->part_starts[q];
    if (p->redundant_rows > 0 && Ln->part_starts[nparts] <= p->redundant_rows) {
      /* small level: every rank holds all of it (one part; the others are empty) */
      const obig all = Ln->part_starts[nparts];
      for (int q = 1; q <= nparts; q++) Ln->part_starts[q] = all;
    }
    free(Sia);
    free(Sja);
    free(part_of);
    l++;
  }
  h->nlev = l + 1;
  double tt = now_s();
  apply_cf_ordering(h);
  tph[4] += now_s() - tt;
  tt = now_s();
  finish_levels(h);
  tph[5] += now_s() - tt;
  if (getenv("ORACLE_TIMING"))
    fprintf(stderr, "oracle setup: strength %.2f  coarsen %.2f  interp %.2f  galerkin %.2f  ordering %.2f  finish %.2f s\n",
            tph[0], tph[1], tph[2], tph[3], tph[4], tph[5]);
  return h;
}

oamg *oamg_from_levels(int nlev, const ocsr *const *A, const ocsr *const *P, const int *const *cf,
                       const obig *const *part_starts, const oamg_params *p) {
  oamg *h = (oamg *)xcalloc(1, sizeof(oamg));
  h->p = *p;
  if (h->p.gs_chunk < 1) h->p.gs_chunk = 1;
  const int nparts = p->nparts > 0 ? p->nparts : 1;
  h->p.nparts = nparts;
  h->p.part_starts = NULL;
  h->nlev = nlev;
  h->L = (olevel *)xcalloc((size_t)nlev, sizeof(olevel));
  for (int l = 0; l < nlev; l++) {
    olevel *L = &h->L[l];
    L->A = (ocsr *)A[l];
    L->part_starts = (obig *)xmalloc(sizeof(obig) * ((size_t)nparts + 1));
    if (part_starts && part_starts[l])
      memcpy(L->part_starts, part_starts[l], sizeof(obig) * ((size_t)nparts + 1));
    else {
      L->part_starts[0] = 0;
      L->part_starts[1] = A[l]->nrows;
    }
    if (l < nlev - 1) {
      L->P = (ocsr *)P[l];
      L->cf = (int *)xmalloc(sizeof(int) * (size_t)A[l]->nrows);
      memcpy(L->cf, cf[l], sizeof(int) * (size_t)A[l]->nrows);
    }
  }
  finish_levels(h);
  return h;
}

void oamg_free(oamg *h) {
  if (!h) return;
  for (int l = 0; l < h->nlev; l++) {
    olevel *L = &h->L[l];
    if (L->own_A) ocsr_free(L->A);
    if (L->own_P) ocsr_free(L->P);
    ocsr_free(L->R);
    free(L->cf);
    free(L->diag);
    free(L->l1gs);
    free(L->l1jac);
    free(L->part_starts);
    free(L->perm);
    oilu_free(L->smoother);
    free(L->u);
    free(L->f);
    free(L->tmp);
    free(L->old);
    free(L->Cinv);
  }
  free(h->L);
  free(h);
}
int oamg_num_levels(const oamg *h) { return h->nlev; }
const ocsr *oamg_A(const oamg *h, int l) { return h->L[l].A; }
const ocsr *oamg_P(const oamg *h, int l) { return h->L[l].P; }
const int *oamg_cf(const oamg *h, int l) { return h->L[l].cf; }
const double *oamg_l1(const oamg *h, int l) { return h->L[l].l1gs; }
const obig *oamg_part_starts(const oamg *h, int l) { return h->L[l].part_starts; }
const int *oamg_perm(const oamg *h, int l) { return h->L[l].perm; }

/* ------------------------------------------------------------ AMG solve -- */

/* hypre_BoomerAMGRelax (par_relax.c), SURVEY A.4.  Hybrid types use the chunk
 * partition as HYPRE uses its OpenMP threads: current values inside the chunk,
 * the pre-sweep snapshot outside it (other chunks, other partitions). */
void oamg_relax(const oamg *h, int level, int type, int points, const double *f, double *u) {
  const olevel *L = &h->L[level];
  const ocsr *A = L->A;
  const int n = A->nrows;
  const double w = h->p.relax_weight * h->p.outer_weight;
  const int *cf = L->cf;
  if (type == 9) {
    if (L->Cinv) {
      for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int j = 0; j < n; j++) s += L->Cinv[(size_t)i * n + j] * f[j];
        u[i] = s;
      }
      return;
    }
    type = h->p.relax_type[0]; /* coarsest level too large for a dense solve */
  }
  double *old = L->old;
  memcpy(old, u, sizeof(double) * (size_t)n);
  if (type == 0 || type == 7 || type == 18) {
    const double *d = (type == 18) ? L->l1jac : L->diag;
#pragma omp parallel for schedule(static) if (g_threads > 1)
    for (int i = 0; i < n; i++) {
      if (points != 0 && cf && cf[i] != points) continue;
      if (d[i] == 0.0) continue;
      double res = f[i];
      for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) res -= A->a[k] * old[A->ja[k]];
      u[i] = old[i] + w * res / d[i];
    }
    return;
  }
  if (type == 11 || type == 12) {
    /* two-stage Gauss-Seidel (par_relax.c hypre_BoomerAMGRelax11/12TwoStageGaussSeidel): the forward solve with
     * D + L of the rank's diag block, replaced by 1 (type 11) or 2 (type 12) further terms of its Neumann series
     *     r = w (f - A u);  z_0 = D^-1 r;  z_k = D^-1 L z_(k-1);  u += z_0 - z_1 (+ z_2)
     * L = strictly lower part inside the row's partition.  The routine does not look at the C/F marker: every
     * point takes part whatever `points` says (with relax_order 1 a sweep therefore applies it twice). */
    const int inner = (type == 11) ? 1 : 2;
    double *z = old, *zn = L->tmp;
    const int *part_of = part_of_rows(n, h->p.nparts, L->part_starts);
    for (int i = 0; i < n; i++) {
      double res = f[i];
      for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) res -= A->a[k] * u[A->ja[k]];
      zn[i] = (L->diag[i] != 0.0) ? w * res / L->diag[i] : 0.0;
    }
    for (int i = 0; i < n; i++) u[i] += zn[i];
    double sign = -1.0;
    for (int it = 0; it < inner; it++) {
      double *t = z;
      z = zn;
      zn = t;
      for (int i = 0; i < n; i++) {
        double sum = 0.0;
        const obig lo = L->part_starts[part_of[i]];
        for (obig k = A->ia[i]; k < A->ia[i + 1]; k++)
          if (A->ja[k] < i && A->ja[k] >= lo) sum += A->a[k] * z[A->ja[k]];
        zn[i] = (L->diag[i] != 0.0) ? sum / L->diag[i] : 0.0;
        u[i] += sign * zn[i];
      }
      sign = -sign;
    }
    free((void *)part_of);
    return;
  }
  const int l1 = (type == 8 || type == 13 || type == 14);
  const int fwd = (type == 3 || type == 6 || type == 8 || type == 13);
  const int bwd = (type == 4 || type == 6 || type == 8 || type == 14);
  if (!fwd && !bwd) {
    fprintf(stderr, "oracle: relax type %d not restated\n", type);
    abort();
  }
  const double *dd = l1 ? L->l1gs : L->diag;
  const int chunk = h->p.gs_chunk;
  for (int p = 0; p < h->p.nparts; p++) {
    const obig ps = L->part_starts[p], pe = L->part_starts[p + 1];
    const obig nch = (pe - ps + chunk - 1) / chunk;
#pragma omp parallel for schedule(static) if (g_threads > 1)
    for (obig c = 0; c < nch; c++) {
      const obig cs = ps + c * chunk;
      const obig ce = (cs + chunk < pe) ? cs + chunk : pe;
      for (int dir = 0; dir < 2; dir++) {
        if ((dir == 0 && !fwd) || (dir == 1 && !bwd)) continue;
        for (obig t = 0; t < ce - cs; t++) {
          const obig i = (dir == 0) ? cs + t : ce - 1 - t;
          if (points != 0 && cf && cf[i] != points) continue;
          if (dd[i] == 0.0) continue;
          double res = f[i];
          for (obig k = A->ia[i]; k < A->ia[i + 1]; k++) {
            const int j = A->ja[k];
            res -= A->a[k] * ((j >= cs && j < ce) ? u[j] : old[j]);
          }
          /* res holds f - (A u)_i including the diagonal term, so both the l1
           * form (u += res/l1) and the plain form (u = (f - sum_{j!=i})/a_ii
           * == u + res/a_ii) are the same statement with their own divisor */
          u[i] += w * res / dd[i];
        }
      }
    }
  }
}

static void relax_sweeps(const oamg *h, int l, int which, const double *f, double *u) {
  /* which: 0 down, 1 up, 2 coarsest; par_cycle.c / hypre_BoomerAMGRelaxIF:
   * relax_order 1 => C then F going down, F then C going up, all on coarsest */
  const int type = h->p.relax_type[which];
  const olevel *L = &h->L[l];
  if (L->smoother && which != 2) {
    /* par_cycle.c: smooth_num_levels > level && smooth_type == 5: HYPRE_ILUSolve(smoother[level], A, F, U) per sweep */
    const int n = L->A->nrows;
    for (int s = 0; s < h->p.num_sweeps[which]; s++)
      for (int it = 0; it < h->p.ilu_max_iter; it++) {
        ocsr_matvec(-1.0, L->A, u, 1.0, f, L->tmp);
        memset(L->old, 0, sizeof(double) * (size_t)n);
        oilu_apply(L->smoother, L->tmp, L->old);
        for (int i = 0; i < n; i++) u[i] += L->old[i];
      }
    return;
  }
  for (int s = 0; s < h->p.num_sweeps[which]; s++) {
    if (which == 2 || h->p.relax_order != 1 || !h->L[l].cf) {
      oamg_relax(h, l, type, 0, f, u);
    } else if (which == 0) {
      oamg_relax(h, l, type, C_PT, f, u);
      oamg_relax(h, l, type, F_PT, f, u);
    } else {
      oamg_relax(h, l, type, F_PT, f, u);
      oamg_relax(h, l, type, C_PT, f, u);
    }
  }
}

static void cycle_level(const oamg *h, int l, const double *f, double *u) {
  /* hypre_BoomerAMGCycle (par_cycle.c), SURVEY A.3 */
  const olevel *L = &h->L[l];
  if (l == h->nlev - 1) {
    relax_sweeps(h, l, 2, f, u);
    return;
  }
  const olevel *Ln = &h->L[l + 1];
  relax_sweeps(h, l, 0, f, u);
  ocsr_matvec(-1.0, L->A, u, 1.0, f, L->tmp);      /* r = f - A u */
  ocsr_matvec(1.0, L->R, L->tmp, 0.0, NULL, Ln->f); /* f_c = P^T r */
  const int ncyc = (h->p.cycle_type == 2 && l + 1 < h->nlev - 1) ? 2 : 1;
  memset(Ln->u, 0, sizeof(double) * (size_t)Ln->A->nrows);
  for (int c = 0; c < ncyc; c++) cycle_level(h, l + 1, Ln->f, Ln->u);
  ocsr_matvec(1.0, L->P, Ln->u, 1.0, u, u); /* u += P e */
  relax_sweeps(h, l, 1, f, u);
}

/* f and u are in the caller's (natural) row order; level 0 works in its C-first order */
void oamg_cycle(const oamg *h, const double *f, double *u) {
  const olevel *L0 = &h->L[0];
  if (!L0->perm) {
    cycle_level(h, 0, f, u);
    return;
  }
  const int n = L0->A->nrows;
  for (int q = 0; q < n; q++) {
    L0->f[q] = f[L0->perm[q]];
    L0->u[q] = u[L0->perm[q]];
  }
  cycle_level(h, 0, L0->f, L0->u);
  for (int q = 0; q < n; q++) u[L0->perm[q]] = L0->u[q];
}

static double vnorm(const double *x, int n) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static) if (g_threads > 1)
  for (int i = 0; i < n; i++) s += x[i] * x[i];
  return sqrt(s);
}
static double vdot(const double *x, const double *y, int n) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static) if (g_threads > 1)
  for (int i = 0; i < n; i++) s += x[i] * y[i];
  return s;
}
static void vaxpy(double a, const double *x, double *y, int n) {
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < n; i++) y[i] += a * x[i];
}
static void vscale(double a, double *x, int n) {
#pragma omp parallel for schedule(static) if (g_threads > 1)
  for (int i = 0; i < n; i++) x[i] *= a;
}

/* HYPRE_BoomerAMGSolve (par_amg_solve.c): as a preconditioner max_iter 1, tol 0
 * (src/HypreSystem.cpp:154-155) => exactly one cycle, no norms. */
int oamg_solve(const oamg *h, const double *b, double *x, int *iters, double *relres) {
  const int n = h->L[0].A->nrows;
  int it = 0;
  double rel = 0.0;
  double bn = 0.0;
  double *r = NULL;
  if (h->p.tol > 0.0) {
    r = (double *)xmalloc(sizeof(double) * (size_t)n);
    bn = vnorm(b, n);
  }
  while (it < h->p.max_iter) {
    oamg_cycle(h, b, x);
    it++;
    if (h->p.tol > 0.0) {
      ocsr_matvec(-1.0, h->L[0].A, x, 1.0, b, r);
      double rn = vnorm(r, n);
      rel = (bn > 0) ? rn / bn : rn;
      if (rel <= h->p.tol) break;
    }
  }
  free(r);
  if (iters) *iters = it;
  if (relres) *relres = rel;
  return 0;
}

void oamg_precond(void *ctx, const double *r, double *z) {
  const oamg *h = (const oamg *)ctx;
  oamg_solve(h, r, z, NULL, NULL);
}

/* Multivectors (HYPRE_IJVectorSetNumComponents > 1; non-segregated solves of
 * src/HypreSystem.cpp:1035-1036, one Solve call at :723 on all components): the Krylov solver sees the block
 * system diag(A, .., A) -- build it with scipy.sparse.kron(I, A) -- and the preconditioner is applied
 * component by component (component-major storage). */
void omulti_precond(void *ctx, const double *r, double *z) {
  const omulti *m = (const omulti *)ctx;
  for (int c = 0; c < m->ncomp; c++) m->M(m->Mctx, r + (size_t)c * m->n, z + (size_t)c * m->n);
}

/* ---------------------------------------------------------------- GMRES -- */


/* ------------------------------------------------------------------ ILU(0) -- */
struct oilu {
  int n;
  ocsr *LU;
  obig *dpos; /* position of the diagonal entry of every row */
  int tri_solve, lower_it, upper_it;
};

/* Symbolic ILU(k) of the block-diagonal part of A (ilu.c hypre_ILUSetupILUKSymbolic; Saad, Iterative Methods, alg.
 * 10.5): row by row, lev = 0 on A's pattern; for every kept lower entry (i,k) in ascending k, every entry (k,j), j > k,
 * of row k's upper part proposes lev(i,j) = lev(i,k) + lev(k,j) + 1 and is merged into row i when that is <= fill.
 * Returns the pattern with A's values on A's positions and zeros on the fill (columns ascending). */
static ocsr *ilu_symbolic(const ocsr *B, int fill) {
  const int n = B->nrows;
  obig cap = B->ia[n] * (fill + 1) + 16, w = 0;
  obig *ia = (obig *)xcalloc((size_t)n + 1, sizeof(obig));
  int *ja = (int *)xmalloc(sizeof(int) * (size_t)cap);
  int *lv = (int *)xmalloc(sizeof(int) * (size_t)cap);
  double *va = (double *)xmalloc(sizeof(double) * (size_t)cap);
  obig *dpos = (obig *)xmalloc(sizeof(obig) * (size_t)(n ? n : 1));
  int *next = (int *)xmalloc(sizeof(int) * ((size_t)n + 1)); /* sorted linked list of the row's columns, head = n */
  int *lev = (int *)xmalloc(sizeof(int) * (size_t)(n ? n : 1));
  double *val = (double *)xcalloc((size_t)(n ? n : 1), sizeof(double));
  for (int i = 0; i < n; i++) {
    int head = n, last = n, cnt = 0;
    next[n] = n;
    for (obig k = B->ia[i]; k < B->ia[i + 1]; k++) { /* columns ascend */
      const int j = B->ja[k];
      next[last] = j;
      next[j] = n;
      last = j;
      lev[j] = 0;
      val[j] = B->a[k];
      cnt++;
    }
    head = next[n];
    for (int k = head; k < i && k != n; k = next[k]) {
      if (lev[k] > fill) continue; /* (never: only kept entries are in the list) */
      if (dpos[k] < 0) continue;
      int at = k; /* insertion cursor: the columns proposed by row k ascend */
      for (obig q = dpos[k] + 1; q < ia[k + 1]; q++) {
        const int j = ja[q];
        const int nl = lev[k] + lv[q] + 1;
        if (nl > fill) continue;
        while (next[at] != n && next[at] < j) at = next[at];
        if (next[at] == j) {
          if (nl < lev[j]) lev[j] = nl;
        } else {
          next[j] = next[at];
          next[at] = j;
          lev[j] = nl;
          val[j] = 0.0;
          cnt++;
        }
      }
    }
    if (w + cnt > cap) {
      cap = (w + cnt) * 2;
      ja = (int *)realloc(ja, sizeof(int) * (size_t)cap);
      lv = (int *)realloc(lv, sizeof(int) * (size_t)cap);
      va = (double *)realloc(va, sizeof(double) * (size_t)cap);
    }
    dpos[i] = -1;
    for (int j = next[n]; j != n; j = next[j]) {
      if (j == i) dpos[i] = w;
      ja[w] = j;
      lv[w] = lev[j];
      va[w] = val[j];
      w++;
    }
    ia[i + 1] = w;
  }
  ocsr *F = ocsr_new(n, n, w);
  memcpy(F->ia, ia, sizeof(obig) * ((size_t)n + 1));
  memcpy(F->ja, ja, sizeof(int) * (size_t)w);
  memcpy(F->a, va, sizeof(double) * (size_t)w);
  free(ia);
  free(ja);
  free(lv);
  free(va);
  free(dpos);
  free(next);
  free(lev);
  free(val);
  return F;
}

oilu *oilu_setup(const ocsr *A, int nparts, const obig *part_starts, int tri_solve, int lower_it, int upper_it) {
  return oilu_setup_k(A, nparts, part_starts, tri_solve, lower_it, upper_it, 0);
}

oilu *oilu_setup_k(const ocsr *A, int nparts, const obig *part_starts, int tri_solve, int lower_it, int upper_it, int level_of_fill) {
  const int n = A->nrows;
  oilu *h = (oilu *)xcalloc(1, sizeof(oilu));
  h->n = n;
  h->tri_solve = tri_solve;
  h->lower_it = lower_it;
  h->upper_it = upper_it;
  obig one_part[2] = {0, n};
  if (nparts < 1 || !part_starts) {
    nparts = 1;
    part_starts = one_part;
  }
  /* the block-diagonal part of A */
  obig *cia = (obig *)xcalloc((size_t)n + 1, sizeof(obig));
  for (int p = 0; p < nparts; p++)
    for (obig i = part_starts[p]; i < part_starts[p + 1]; i++) {
      obig c = 0;
      for (obig k = A->ia[i]; k < A->ia[i + 1]; k++)
        if (A->ja[k] >= part_starts[p] && A->ja[k] < part_starts[p + 1]) c++;
      cia[i + 1] = c;
    }
  for (int i = 0; i < n; i++) cia[i + 1] += cia[i];
  ocsr *LU = ocsr_new(n, n, cia[n]);
  memcpy(LU->ia, cia, sizeof(obig) * ((size_t)n + 1));
  free(cia);
  h->dpos = (obig *)xcalloc((size_t)n, sizeof(obig));
  for (int p = 0; p < nparts; p++)
    for (obig i = part_starts[p]; i < part_starts[p + 1]; i++) {
      obig q = LU->ia[i];
      h->dpos[i] = -1;
      for (obig k = A->ia[i]; k < A->ia[i + 1]; k++)
        if (A->ja[k] >= part_starts[p] && A->ja[k] < part_starts[p + 1]) {
          LU->ja[q] = A->ja[k];
          LU->a[q] = A->a[k];
          if (A->ja[k] == i) h->dpos[i] = q;
          q++;
        }
    }
  if (level_of_fill > 0) { /* the factors live on the ILU(k) pattern: zeros on the fill positions */
    ocsr *F = ilu_symbolic(LU, level_of_fill);
    ocsr_free(LU);
    LU = F;
    for (int i = 0; i < n; i++) {
      h->dpos[i] = -1;
      for (obig k = LU->ia[i]; k < LU->ia[i + 1]; k++)
        if (LU->ja[k] == i) h->dpos[i] = k;
    }
  }
  /* IKJ: for k < i in row i (ascending): l_ik = a_ik / u_kk ; a_ij -= l_ik u_kj for j > k in both patterns */
  for (int i = 0; i < n; i++) {
    for (obig kk = LU->ia[i]; kk < LU->ia[i + 1]; kk++) {
      const int k = LU->ja[kk];
      if (k >= i) break;
      if (h->dpos[k] < 0) continue;
      const double l = LU->a[kk] / LU->a[h->dpos[k]];
      LU->a[kk] = l;
      obig pk = h->dpos[k] + 1; /* entries of row k right of its diagonal */
      const obig ek = LU->ia[k + 1];
      for (obig jj = kk + 1; jj < LU->ia[i + 1]; jj++) {
        const int j = LU->ja[jj];
        while (pk < ek && LU->ja[pk] < j) pk++;
        if (pk < ek && LU->ja[pk] == j) LU->a[jj] -= l * LU->a[pk];
      }
    }
  }
  h->LU = LU;
  return h;
}

void oilu_free(oilu *h) {
  if (!h) return;
  ocsr_free(h->LU);
  free(h->dpos);
  free(h);
}

const ocsr *oilu_factor(const oilu *h) { return h->LU; }

void oilu_apply(const oilu *h, const double *r, double *z) {
  const int n = h->n;
  const ocsr *LU = h->LU;
  double *y = (double *)xmalloc(sizeof(double) * (size_t)n);
  if (h->tri_solve) {
    for (int i = 0; i < n; i++) {
      double s = r[i];
      for (obig k = LU->ia[i]; k < LU->ia[i + 1] && LU->ja[k] < i; k++) s -= LU->a[k] * y[LU->ja[k]];
      y[i] = s;
    }
    for (int i = n - 1; i >= 0; i--) {
      double s = y[i];
      const obig d = h->dpos[i];
      for (obig k = d + 1; k < LU->ia[i + 1]; k++) s -= LU->a[k] * z[LU->ja[k]];
      z[i] = (d >= 0) ? s / LU->a[d] : s;
    }
  } else {
    /* Jacobi iterations on the triangular systems: y <- r - L_strict y ;  z <- D^-1 (y - U_strict z) */
    double *t = (double *)xmalloc(sizeof(double) * (size_t)n);
    memcpy(y, r, sizeof(double) * (size_t)n);
    for (int it = 0; it < h->lower_it; it++) {
      for (int i = 0; i < n; i++) {
        double s = r[i];
        for (obig k = LU->ia[i]; k < LU->ia[i + 1] && LU->ja[k] < i; k++) s -= LU->a[k] * y[LU->ja[k]];
        t[i] = s;
      }
      memcpy(y, t, sizeof(double) * (size_t)n);
    }
    for (int i = 0; i < n; i++) z[i] = (h->dpos[i] >= 0) ? y[i] / LU->a[h->dpos[i]] : y[i];
    for (int it = 0; it < h->upper_it; it++) {
      for (int i = 0; i < n; i++) {
        double s = y[i];
        const obig d = h->dpos[i];
        for (obig k = d + 1; k < LU->ia[i + 1]; k++) s -= LU->a[k] * z[LU->ja[k]];
        t[i] = (d >= 0) ? s / LU->a[d] : s;
      }
      memcpy(z, t, sizeof(double) * (size_t)n);
    }
    free(t);
  }
  free(y);
}

void oilu_precond(void *ctx, const double *r, double *z) { oilu_apply((const oilu *)ctx, r, z); }

int oilu_solve(const oilu *h, const ocsr *A, const double *b, double *x, int max_iter, double tol, double *rel_res) {
  const int n = h->n;
  double *r = (double *)xmalloc(sizeof(double) * (size_t)n);
  double *z = (double *)xmalloc(sizeof(double) * (size_t)n);
  const double bn = vnorm(b, n);
  int it = 0;
  double rel = 0.0;
  while (it < max_iter) {
    ocsr_matvec(-1.0, A, x, 1.0, b, r);
    if (tol > 0.0) {
      const double rn = vnorm(r, n);
      rel = (bn > 0.0) ? rn / bn : rn;
      if (rel <= tol) break;
    }
    oilu_apply(h, r, z);
    vaxpy(1.0, z, x, n);
    it++;
  }
  if (rel_res) {
    ocsr_matvec(-1.0, A, x, 1.0, b, r);
    const double rn = vnorm(r, n);
    *rel_res = (bn > 0.0) ? rn / bn : rn;
  }
  free(r);
  free(z);
  return it;
}

/* hypre_GMRESSolve (krylov/gmres.c); called through solverSolvePtr_
 * at src/HypreSystem.cpp:723 with tol/max_iter/k_dim from :393-397, x0 = 0 (:580). */
static void gmres_core(const ocsr *A, const double *b, double *x, int kdim, double tol, double atol, int maxit,
                       oprecond_fn M, void *Mctx, okrylov_result *res, double *norms, int flexible, int ortho) {
  const int n = A->nrows;
  const double epsmac = 1.e-16;
  double **p = (double **)xmalloc(sizeof(double *) * ((size_t)kdim + 1));
  for (int i = 0; i <= kdim; i++) p[i] = (double *)xcalloc((size_t)n, sizeof(double));
  double *r = (double *)xcalloc((size_t)n, sizeof(double));
  double *w = (double *)xcalloc((size_t)n, sizeof(double));
  double **z = NULL; /* FlexGMRES keeps z_j = M^-1 p_j (krylov/flexgmres.c) */
  if (flexible) {
    z = (double **)xmalloc(sizeof(double *) * (size_t)kdim);
    for (int i = 0; i < kdim; i++) z[i] = (double *)xcalloc((size_t)n, sizeof(double));
  }
  double *c = (double *)xcalloc((size_t)kdim + 1, sizeof(double));
  double *s = (double *)xcalloc((size_t)kdim + 1, sizeof(double));
  double *rs = (double *)xcalloc((size_t)kdim + 1, sizeof(double));
  double **hh = (double **)xmalloc(sizeof(double *) * ((size_t)kdim + 1));
  for (int i = 0; i <= kdim; i++) hh[i] = (double *)xcalloc((size_t)kdim, sizeof(double));

  ocsr_matvec(-1.0, A, x, 1.0, b, p[0]);
  const double b_norm = vnorm(b, n);
  double r_norm = vnorm(p[0], n);
  const double den = (b_norm > 0.0) ? b_norm : r_norm;
  const double eps = fmax(atol, tol * den);
  int iter = 0, converged = 0;
  if (norms) norms[0] = r_norm;

  while (iter < maxit) {
    rs[0] = r_norm;
    if (r_norm == 0.0) {
      converged = 1;
      break;
    }
    if (r_norm <= eps) {
      ocsr_matvec(-1.0, A, x, 1.0, b, r);
      r_norm = vnorm(r, n);
      if (r_norm <= eps) {
        converged = 1;
        break;
      }
      /* false convergence 1: carry on from the true residual norm */
    }
    vscale(1.0 / r_norm, p[0], n);
    int i = 0;
    while (i < kdim && iter < maxit) {
      i++;
      iter++;
      double *dir = flexible ? z[i - 1] : r;
      memset(dir, 0, sizeof(double) * (size_t)n);
      if (M)
        M(Mctx, p[i - 1], dir);
      else
        memcpy(dir, p[i - 1], sizeof(double) * (size_t)n);
      ocsr_matvec(1.0, A, dir, 0.0, NULL, p[i]);
      if (ortho == 0) { /* modified Gram-Schmidt (krylov/gmres.c) */
        for (int j = 0; j < i; j++) {
          hh[j][i - 1] = vdot(p[j], p[i], n);
          vaxpy(-hh[j][i - 1], p[j], p[i], n);
        }
      } else { /* classical Gram-Schmidt, ortho passes: all inner products off the same vector, then the update */
        double *hv = (double *)xcalloc((size_t)i, sizeof(double));
        for (int j = 0; j < i; j++) hh[j][i - 1] = 0.0;
        for (int pass = 0; pass < ortho; pass++) {
          for (int j = 0; j < i; j++) hv[j] = vdot(p[j], p[i], n);
          for (int j = 0; j < i; j++) vaxpy(-hv[j], p[j], p[i], n);
          for (int j = 0; j < i; j++) hh[j][i - 1] = (pass == 0) ? hv[j] : hh[j][i - 1] + hv[j];
        }
        free(hv);
      }
      double t = vnorm(p[i], n);
      hh[i][i - 1] = t;
      if (t != 0.0) vscale(1.0 / t, p[i], n);
      for (int j = 1; j < i; j++) {
        double tt = hh[j - 1][i - 1];
        hh[j - 1][i - 1] = s[j - 1] * hh[j][i - 1] + c[j - 1] * tt;
        hh[j][i - 1] = -s[j - 1] * tt + c[j - 1] * hh[j][i - 1];
      }
      double gamma = sqrt(hh[i - 1][i - 1] * hh[i - 1][i - 1] + hh[i][i - 1] * hh[i][i - 1]);
      if (gamma == 0.0) gamma = epsmac;
      c[i - 1] = hh[i - 1][i - 1] / gamma;
      s[i - 1] = hh[i][i - 1] / gamma;
      rs[i] = -hh[i][i - 1] * rs[i - 1];
      rs[i] /= gamma;
      rs[i - 1] = c[i - 1] * rs[i - 1];
      hh[i - 1][i - 1] = s[i - 1] * hh[i][i - 1] + c[i - 1] * hh[i - 1][i - 1];
      r_norm = fabs(rs[i]);
      if (norms) norms[iter] = r_norm;
      if (r_norm <= eps) break;
    }
    /* back substitution */
    double *y = (double *)xcalloc((size_t)i + 1, sizeof(double));
    memcpy(y, rs, sizeof(double) * (size_t)i);
    y[i - 1] = y[i - 1] / hh[i - 1][i - 1];
    for (int k = i - 2; k >= 0; k--) {
      double t = 0.0;
      for (int j = k + 1; j < i; j++) t -= hh[k][j] * y[j];
      t += y[k];
      y[k] = t / hh[k][k];
    }
    if (flexible) {
      for (int j = i - 1; j >= 0; j--) vaxpy(y[j], z[j], x, n);
      free(y);
    } else {
      /* w = sum y_j p_j ; x += M^-1 w */
      memcpy(w, p[i - 1], sizeof(double) * (size_t)n);
      vscale(y[i - 1], w, n);
      for (int j = i - 2; j >= 0; j--) vaxpy(y[j], p[j], w, n);
      free(y);
      memset(r, 0, sizeof(double) * (size_t)n);
      if (M)
        M(Mctx, w, r);
      else
        memcpy(r, w, sizeof(double) * (size_t)n);
      vaxpy(1.0, r, x, n);
    }
    if (r_norm <= eps) {
      ocsr_matvec(-1.0, A, x, 1.0, b, r);
      r_norm = vnorm(r, n);
      if (r_norm <= eps) {
        converged = 1;
        break;
      }
      /* false convergence 2: restart from the true residual */
      memcpy(p[0], r, sizeof(double) * (size_t)n);
      i = 0;
    }
    if (flexible) { /* restart from the explicitly recomputed residual */
      if (i) {
        ocsr_matvec(-1.0, A, x, 1.0, b, p[0]);
        r_norm = vnorm(p[0], n);
      }
      continue;
    }
    /* residual vector for the restart, rebuilt from the Givens data */
    for (int j = i; j > 0; j--) {
      rs[j - 1] = -s[j - 1] * rs[j];
      rs[j] = c[j - 1] * rs[j];
    }
    if (i) vaxpy(rs[i] - 1.0, p[i], p[i], n);
    for (int j = i - 1; j > 0; j--) vaxpy(rs[j], p[j], p[i], n);
    if (i) {
      vaxpy(rs[0] - 1.0, p[0], p[0], n);
      vaxpy(1.0, p[i], p[0], n);
    }
  }
  if (res) {
    res->iters = iter;
    res->converged = converged;
    res->rel_res = (b_norm > 0.0) ? r_norm / b_norm : r_norm;
    ocsr_matvec(-1.0, A, x, 1.0, b, r);
    double tn = vnorm(r, n);
    res->true_rel_res = (b_norm > 0.0) ? tn / b_norm : tn;
  }
  for (int i = 0; i <= kdim; i++) {
    free(p[i]);
    free(hh[i]);
  }
  if (z) {
    for (int i = 0; i < kdim; i++) free(z[i]);
    free(z);
  }
  free(p);
  free(hh);
  free(r);
  free(w);
  free(c);
  free(s);
  free(rs);
}

void ogmres_solve(const ocsr *A, const double *b, double *x, int kdim, double tol, double atol, int maxit,
                  oprecond_fn M, void *Mctx, okrylov_result *res, double *norms) {
  gmres_core(A, b, x, kdim, tol, atol, maxit, M, Mctx, res, norms, 0, 0);
}

/* hypre_FlexGMRESSolve (krylov/flexgmres.c); bound at src/HypreSystem.cpp:406-421 */
void ofgmres_solve(const ocsr *A, const double *b, double *x, int kdim, double tol, double atol, int maxit,
                   oprecond_fn M, void *Mctx, okrylov_result *res, double *norms) {
  gmres_core(A, b, x, kdim, tol, atol, maxit, M, Mctx, res, norms, 1, 0);
}

/* hypre_COGMRESSolve (krylov/cogmres.c), bound at src/HypreSystem.cpp:372-388: the GMRES skeleton with
 * CLASSICAL Gram-Schmidt -- one block of inner products and one block update per pass, cgs <= 1: one pass,
 * cgs >= 2: two passes (coefficients added).  Restated from the published low-synchronisation GMRES
 * algorithm; like everything HYPRE-side: parity unpinned. */
void ocogmres_solve(const ocsr *A, const double *b, double *x, int kdim, int cgs, double tol, double atol, int maxit,
                    oprecond_fn M, void *Mctx, okrylov_result *res, double *norms) {
  gmres_core(A, b, x, kdim, tol, atol, maxit, M, Mctx, res, norms, 0, cgs >= 2 ? 2 : 1);
}

/* hypre_PCGSolve (krylov/pcg.c), default options (two_norm 0: the measure is <C r,r>/<C b,b>
 * against tol^2); bound at src/HypreSystem.cpp:440-455.  norms[i] = sqrt(i_prod/bi_prod). */
void opcg_solve(const ocsr *A, const double *b, double *x, double tol, double atol, int maxit, oprecond_fn M,
                void *Mctx, okrylov_result *res, double *norms) {
  const int n = A->nrows;
  double *r = (double *)xcalloc((size_t)n, sizeof(double));
  double *p = (double *)xcalloc((size_t)n, sizeof(double));
  double *s = (double *)xcalloc((size_t)n, sizeof(double));
#define OPRECOND(in, out)                          \
  do {                                             \
    memset(out, 0, sizeof(double) * (size_t)n);    \
    if (M)                                         \
      M(Mctx, in, out);                            \
    else                                           \
      memcpy(out, in, sizeof(double) * (size_t)n); \
  } while (0)
  OPRECOND(b, p);
  const double bi_prod = vdot(p, b, n);
  double eps = tol * tol;
  int i = 0, converged = 0;
  double i_prod = 0.0;
  if (!(bi_prod > 0.0)) {
    memset(x, 0, sizeof(double) * (size_t)n);
    converged = 1;
  } else {
    if (atol > 0.0 && atol * atol / bi_prod > eps) eps = atol * atol / bi_prod;
    ocsr_matvec(-1.0, A, x, 1.0, b, r);
    OPRECOND(r, p);
    double gamma = vdot(r, p, n);
    i_prod = gamma;
    if (norms) norms[0] = sqrt(fabs(i_prod) / bi_prod);
    while (i + 1 <= maxit) {
      i++;
      ocsr_matvec(1.0, A, p, 0.0, NULL, s);
      const double sdotp = vdot(s, p, n);
      if (sdotp == 0.0) break;
      const double alpha = gamma / sdotp;
      const double gamma_old = gamma;
      vaxpy(alpha, p, x, n);
      vaxpy(-alpha, s, r, n);
      OPRECOND(r, s);
      gamma = vdot(r, s, n);
      i_prod = gamma;
      if (norms) norms[i] = sqrt(fabs(i_prod) / bi_prod);
      if (i_prod / bi_prod < eps) {
        converged = 1;
        break;
      }
      const double beta = gamma / gamma_old;
      vscale(beta, p, n);
      vaxpy(1.0, s, p, n);
    }
  }
#undef OPRECOND
  if (res) {
    res->iters = i;
    res->converged = converged;
    res->rel_res = (bi_prod > 0.0) ? sqrt(fabs(i_prod) / bi_prod) : 0.0;
    ocsr_matvec(-1.0, A, x, 1.0, b, r);
    const double bn = vnorm(b, n);
    res->true_rel_res = (bn > 0.0) ? vnorm(r, n) / bn : vnorm(r, n);
  }
  free(r);
  free(p);
  free(s);
}

/* ------------------------------------------------------------- BiCGSTAB -- */

/* hypre_BiCGSTABSolve (krylov/bicgstab.c); bound at
 * src/HypreSystem.cpp:423-438. */
void obicgstab_solve(const ocsr *A, const double *b, double *x, double tol, double atol, int maxit, oprecond_fn M,
                     void *Mctx, okrylov_result *res, double *norms) {
  const int n = A->nrows;
  const double epsmac = 1.e-128;
  double *r0 = (double *)xcalloc((size_t)n, sizeof(double));
  double *r = (double *)xcalloc((size_t)n, sizeof(double));
  double *p = (double *)xcalloc((size_t)n, sizeof(double));
  double *v = (double *)xcalloc((size_t)n, sizeof(double));
  double *q = (double *)xcalloc((size_t)n, sizeof(double));
  double *s = (double *)xcalloc((size_t)n, sizeof(double));
  double *t = (double *)xcalloc((size_t)n, sizeof(double));
  ocsr_matvec(-1.0, A, x, 1.0, b, r0);
  memcpy(r, r0, sizeof(double) * (size_t)n);
  memcpy(p, r0, sizeof(double) * (size_t)n);
  const double b_norm = vnorm(b, n);
  double rho = vdot(r0, r0, n);
  double r_norm = sqrt(rho);
  const double den = (b_norm > 0.0) ? b_norm : r_norm;
  const double eps = fmax(atol, tol * den);
  int iter = 0, converged = 0;
  if (norms) norms[0] = r_norm;
  if (r_norm == 0.0) converged = 1;
  while (!converged && iter < maxit) {
    iter++;
    memset(v, 0, sizeof(double) * (size_t)n);
    if (M)
      M(Mctx, p, v);
    else
      memcpy(v, p, sizeof(double) * (size_t)n);
    ocsr_matvec(1.0, A, v, 0.0, NULL, q);
    double temp = vdot(r0, q, n);
    if (fabs(temp) < epsmac) break;
    double alpha = rho / temp;
    vaxpy(alpha, v, x, n);
    vaxpy(-alpha, q, r, n);
    r_norm = vnorm(r, n);
    if (r_norm <= eps) {
      ocsr_matvec(-1.0, A, x, 1.0, b, t);
      double tn = vnorm(t, n);
      if (tn <= eps) {
        r_norm = tn;
        if (norms) norms[iter] = r_norm;
        converged = 1;
        break;
      }
    }
    memset(v, 0, sizeof(double) * (size_t)n);
    if (M)
      M(Mctx, r, v);
    else
      memcpy(v, r, sizeof(double) * (size_t)n);
    ocsr_matvec(1.0, A, v, 0.0, NULL, s);
    double ss = vdot(s, s, n);
    double gamma = (ss != 0.0) ? vdot(r, s, n) / ss : 0.0;
    vaxpy(gamma, v, x, n);
    vaxpy(-gamma, s, r, n);
    r_norm = vnorm(r, n);
    if (norms) norms[iter] = r_norm;
    if (r_norm <= eps) {
      ocsr_matvec(-1.0, A, x, 1.0, b, t);
      double tn = vnorm(t, n);
      if (tn <= eps) {
        r_norm = tn;
        converged = 1;
        break;
      }
    }
    if (fabs(rho) < epsmac) break;
    double beta = 1.0 / rho;
    rho = vdot(r0, r, n);
    beta *= rho;
    vaxpy(-gamma, q, p, n);
    if (fabs(gamma) < epsmac) break;
    vscale(beta * alpha / gamma, p, n);
    vaxpy(1.0, r, p, n);
  }
  if (res) {
    res->iters = iter;
    res->converged = converged;
    res->rel_res = (b_norm > 0.0) ? r_norm / b_norm : r_norm;
    ocsr_matvec(-1.0, A, x, 1.0, b, t);
    double tn = vnorm(t, n);
    res->true_rel_res = (b_norm > 0.0) ? tn / b_norm : tn;
  }
  free(r0);
  free(r);
  free(p);
  free(v);
  free(q);
  free(s);
  free(t);
}
