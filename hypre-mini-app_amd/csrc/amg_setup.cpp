// BoomerAMG setup (HYPRE_BoomerAMGSetup, reached from src/HypreSystem.cpp:692
// through HYPRE_ParCSRGMRESSetup).  Host control plane, threaded over rows;
// outside the solve-phase metric.  Algorithms per SURVEY A.5:
// strength (par_strength.c), PMIS (par_coarsen.c), extended+i / direct /
// classical-modified interpolation with truncation (par_lr_interp.c,
// par_interp.c), Galerkin R*(A*P) (par_rap.c), l1 norms (par_relax_more.c).
//
// Multi-rank variant: coarsening and interpolation are rank-local (connections
// to halo columns are treated as weak), so P has no off-rank columns and the
// Galerkin product needs exactly one exchange: the P rows of the halo columns.
#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstring>

#include "amg.hpp"
#include "amg_setup_internal.hpp"
#include "kernels.hpp"
#include "solvers.hpp"

namespace mi {

namespace hs {  // host setup algorithms, shared with amg_setup_dist.cpp (amg_setup_internal.hpp)

constexpr int MAX_DENSE = 4096;

// hypre_SeedRand / hypre_Rand (Park-Miller minimal standard)
struct ParkMiller {
  int seed;
  explicit ParkMiller(int s) : seed(s ? s : 13579) {}
  double next() {
    const int a = 16807, m = 2147483647, q = 127773, r = 2836;
    const int lo = seed % q, hi = seed / q;
    const int t = a * lo - r * hi;
    seed = (t > 0) ? t : t + m;
    return (double)seed / m;
  }
};

// strong iff a_ij < theta*min_k a_ik (a_ii > 0; mirrored for a_ii < 0); the row
// scale and row sum run over diag AND offd entries; only diag-block entries are
// kept because halo connections do not take part in rank-local coarsening
void strength(const ParCSR &A, double theta, double max_row_sum, Strength &S) {
  const HostCSR &D = A.diag, &O = A.offd;
  const int n = D.nrows;
  std::vector<int> cnt((size_t)n, 0);
  std::vector<double> thr((size_t)n, 0.0);
  std::vector<signed char> mode((size_t)n, 0);  // 0 none, 1 diag>=0, -1 diag<0
  parallel_for(n, [&](int64_t b, int64_t e, int) {
    for (int64_t i = b; i < e; i++) {
      double diag = 0.0, row_sum = 0.0;
      for (int64_t k = D.ia[(size_t)i]; k < D.ia[(size_t)i + 1]; k++) {
        row_sum += D.a[(size_t)k];
        if (D.ja[(size_t)k] == i) diag = D.a[(size_t)k];
      }
      for (int64_t k = O.ia[(size_t)i]; k < O.ia[(size_t)i + 1]; k++) row_sum += O.a[(size_t)k];
      double scale = 0.0;
      auto upd = [&](double v) {
        if (diag < 0) {
          if (v > scale) scale = v;
        } else {
          if (v < scale) scale = v;
        }
      };
      for (int64_t k = D.ia[(size_t)i]; k < D.ia[(size_t)i + 1]; k++)
        if (D.ja[(size_t)k] != i) upd(D.a[(size_t)k]);
      for (int64_t k = O.ia[(size_t)i]; k < O.ia[(size_t)i + 1]; k++) upd(O.a[(size_t)k]);
      const bool all_weak = (std::fabs(row_sum) > std::fabs(diag) * max_row_sum) && (max_row_sum < 1.0);
      if (all_weak) continue;
      mode[(size_t)i] = (diag < 0) ? -1 : 1;
      thr[(size_t)i] = theta * scale;
      int c = 0;
      for (int64_t k = D.ia[(size_t)i]; k < D.ia[(size_t)i + 1]; k++) {
        if (D.ja[(size_t)k] == i) continue;
        const double v = D.a[(size_t)k];
        if ((diag < 0) ? (v > thr[(size_t)i]) : (v < thr[(size_t)i])) c++;
      }
      cnt[(size_t)i] = c;
    }
  });
  S.ia.assign((size_t)n + 1, 0);
  for (int i = 0; i < n; i++) S.ia[(size_t)i + 1] = S.ia[(size_t)i] + cnt[(size_t)i];
  S.ja.resize((size_t)S.ia[(size_t)n]);
  parallel_for(n, [&](int64_t b, int64_t e, int) {
    for (int64_t i = b; i < e; i++) {
      if (!mode[(size_t)i]) continue;
      int64_t q = S.ia[(size_t)i];
      for (int64_t k = D.ia[(size_t)i]; k < D.ia[(size_t)i + 1]; k++) {
        if (D.ja[(size_t)k] == i) continue;
        const double v = D.a[(size_t)k];
        if ((mode[(size_t)i] < 0) ? (v > thr[(size_t)i]) : (v < thr[(size_t)i])) S.ja[(size_t)q++] = D.ja[(size_t)k];
      }
    }
  });
}

// PMIS, rank-local graph; measure = |S^T row| + Park-Miller(2747 + rank)
void pmis(int n, const Strength &S, int rank, std::vector<int> &cf) {
  std::vector<double> measure((size_t)n, 0.0);
  for (int64_t k = 0; k < (int64_t)S.ja.size(); k++) measure[(size_t)S.ja[(size_t)k]] += 1.0;
  ParkMiller rng(2747 + rank);
  for (int i = 0; i < n; i++) measure[(size_t)i] += rng.next();
  cf.assign((size_t)n, 0);
  std::vector<int> graph;
  graph.reserve((size_t)n);
  for (int i = 0; i < n; i++) {
    if (S.ia[(size_t)i + 1] == S.ia[(size_t)i]) {
      cf[(size_t)i] = SF_PT;
      measure[(size_t)i] = 0.0;
    } else if (measure[(size_t)i] < 1.0) {
      cf[(size_t)i] = F_PT;
      measure[(size_t)i] = 0.0;
    } else
      graph.push_back(i);
  }
  std::vector<signed char> tmp((size_t)n, 0);
  while (!graph.empty()) {
    const int64_t ng = (int64_t)graph.size();
    parallel_for(ng, [&](int64_t b, int64_t e, int) {
      for (int64_t g = b; g < e; g++) tmp[(size_t)graph[(size_t)g]] = 1;
    });
    parallel_for(ng, [&](int64_t b, int64_t e, int) {
      for (int64_t g = b; g < e; g++) {
        const int i = graph[(size_t)g];
        for (int64_t k = S.ia[(size_t)i]; k < S.ia[(size_t)i + 1]; k++) {
          const int j = S.ja[(size_t)k];
          if (cf[(size_t)j] != 0) continue;
          if (measure[(size_t)i] > measure[(size_t)j])
            __atomic_store_n(&tmp[(size_t)j], (signed char)0, __ATOMIC_RELAXED);
          else if (measure[(size_t)j] > measure[(size_t)i])
            __atomic_store_n(&tmp[(size_t)i], (signed char)0, __ATOMIC_RELAXED);
        }
      }
    });
    parallel_for(ng, [&](int64_t b, int64_t e, int) {
      for (int64_t g = b; g < e; g++) {
        const int i = graph[(size_t)g];
        if (tmp[(size_t)i] == 1) cf[(size_t)i] = C_PT;
      }
    });
    // new F points: undecided rows that depend on a C point (C points of this
    // round included, exactly as the serial loop sees them)
    std::vector<signed char> becomes_f((size_t)ng, 0);
    parallel_for(ng, [&](int64_t b, int64_t e, int) {
      for (int64_t g = b; g < e; g++) {
        const int i = graph[(size_t)g];
        if (cf[(size_t)i] != 0) continue;
        for (int64_t k = S.ia[(size_t)i]; k < S.ia[(size_t)i + 1]; k++)
          if (cf[(size_t)S.ja[(size_t)k]] == C_PT) {
            becomes_f[(size_t)g] = 1;
            break;
          }
      }
    });
    std::vector<int> next;
    next.reserve(graph.size());
    for (int64_t g = 0; g < ng; g++) {
      const int i = graph[(size_t)g];
      if (becomes_f[(size_t)g]) cf[(size_t)i] = F_PT;
      if (cf[(size_t)i] == 0)
        next.push_back(i);
      else
        measure[(size_t)i] = 0.0;
    }
    graph.swap(next);
  }
}

// keep entries >= trunc_factor*max|p|, then the pmax largest by (|p| desc,
// position asc), rescale to the original row sum; stored order is kept
int truncate_row(int len, int *cols, double *vals, double trunc_factor, int pmax, std::vector<char> &keep) {
  if (len == 0) return 0;
  double row_sum = 0.0, maxabs = 0.0;
  for (int k = 0; k < len; k++) {
    row_sum += vals[k];
    maxabs = std::max(maxabs, std::fabs(vals[k]));
  }
  keep.assign((size_t)len, 1);
  if (trunc_factor > 0.0)
    for (int k = 0; k < len; k++) keep[(size_t)k] = std::fabs(vals[k]) >= trunc_factor * maxabs;
  int nk = 0;
  for (int k = 0; k < len; k++) nk += keep[(size_t)k];
  if (pmax > 0)
    while (nk > pmax) {
      int worst = -1;
      for (int k = 0; k < len; k++)
        if (keep[(size_t)k] && (worst < 0 || std::fabs(vals[k]) <= std::fabs(vals[worst]))) worst = k;
      keep[(size_t)worst] = 0;
      nk--;
    }
  double kept = 0.0;
  for (int k = 0; k < len; k++)
    if (keep[(size_t)k]) kept += vals[k];
  const double scale = (kept != 0.0) ? row_sum / kept : 1.0;
  int m = 0;
  for (int k = 0; k < len; k++)
    if (keep[(size_t)k]) {
      cols[m] = cols[k];
      vals[m] = vals[k] * scale;
      m++;
    }
  return m;
}

// interpolation of the rank-local block; halo entries are lumped into the
// diagonal like weak connections
void build_interp(const ParCSR &A, const Strength &S, std::vector<int> &cf, int interp_type, double trunc_factor,
                  int pmax, HostCSR &P, int &nc_out, const std::vector<char> *want_rows) {
  const HostCSR &D = A.diag, &O = A.offd;
  const int n = D.nrows;
  std::vector<int> f2c((size_t)n, -1);
  int nc = 0;
  for (int i = 0; i < n; i++)
    if (cf[(size_t)i] == C_PT) f2c[(size_t)i] = nc++;
  nc_out = nc;
  const int nt = host_threads();
  std::vector<std::vector<int>> tj((size_t)nt);
  std::vector<std::vector<double>> ta((size_t)nt);
  std::vector<int> rowlen((size_t)n, 0);
  std::vector<int64_t> tbeg((size_t)nt + 1, 0);
  std::vector<char> used((size_t)nt, 0);
  parallel_for(n, [&](int64_t b, int64_t e, int t) {
    used[(size_t)t] = 1;
    tbeg[(size_t)t] = b;
    std::vector<int> &oj = tj[(size_t)t];
    std::vector<double> &oa = ta[(size_t)t];
    std::vector<int> rc, sf;  // interpolatory set (fine ids, discovery order), strong F neighbours
    std::vector<double> rv;
    std::vector<char> keep;
    auto find = [](const std::vector<int> &v, int x) {
      for (size_t q = 0; q < v.size(); q++)
        if (v[q] == x) return (int)q;
      return -1;
    };
    for (int64_t i = b; i < e; i++) {
      rc.clear();
      rv.clear();
      sf.clear();
      if (want_rows && !(*want_rows)[(size_t)i]) {
        // a row of the extended sub-problem that belongs to another rank: left empty
      } else if (cf[(size_t)i] == C_PT) {
        rc.push_back((int)i);
        rv.push_back(1.0);
      } else if (cf[(size_t)i] != SF_PT) {
        for (int64_t k = S.ia[(size_t)i]; k < S.ia[(size_t)i + 1]; k++) {
          const int i1 = S.ja[(size_t)k];
          if (cf[(size_t)i1] == C_PT) {
            if (find(rc, i1) < 0) rc.push_back(i1);
          } else if (cf[(size_t)i1] != SF_PT && interp_type != 3) {
            sf.push_back(i1);
            if (interp_type == 6)
              for (int64_t kk = S.ia[(size_t)i1]; kk < S.ia[(size_t)i1 + 1]; kk++) {
                const int k1 = S.ja[(size_t)kk];
                if (cf[(size_t)k1] == C_PT && find(rc, k1) < 0) rc.push_back(k1);
              }
          }
        }
        rv.assign(rc.size(), 0.0);
        double diagonal = 0.0;
        for (int64_t k = D.ia[(size_t)i]; k < D.ia[(size_t)i + 1]; k++)
          if (D.ja[(size_t)k] == i) diagonal = D.a[(size_t)k];
        if (interp_type == 3) {
          double sNp = 0, sNn = 0, sPp = 0, sPn = 0;
          for (int64_t k = D.ia[(size_t)i]; k < D.ia[(size_t)i + 1]; k++) {
            const int j = D.ja[(size_t)k];
            if (j == i) continue;
            const double v = D.a[(size_t)k];
            (v > 0 ? sNp : sNn) += v;
            const int q = find(rc, j);
            if (q >= 0) {
              rv[(size_t)q] += v;
              (v > 0 ? sPp : sPn) += v;
            }
          }
          for (int64_t k = O.ia[(size_t)i]; k < O.ia[(size_t)i + 1]; k++) (O.a[(size_t)k] > 0 ? sNp : sNn) += O.a[(size_t)k];
          double alfa = 1.0, beta = 1.0;
          if (sPn != 0) alfa = sNn / sPn / diagonal;
          if (sPp != 0) beta = sNp / sPp / diagonal;
          if (sPp == 0) {
            const double d2 = diagonal + sNp;
            if (sPn != 0) alfa = sNn / sPn / d2;
            beta = 0.0;
          }
          for (size_t q = 0; q < rv.size(); q++) rv[q] = (rv[q] > 0) ? -beta * rv[q] : -alfa * rv[q];
        } else {
          for (int64_t k = D.ia[(size_t)i]; k < D.ia[(size_t)i + 1]; k++) {
            const int i1 = D.ja[(size_t)k];
            if (i1 == i) continue;
            const double aik = D.a[(size_t)k];
            const int q = find(rc, i1);
            if (q >= 0) {
              rv[(size_t)q] += aik;
            } else if (find(sf, i1) >= 0) {
              double dk = 0.0;
              for (int64_t kk = D.ia[(size_t)i1]; kk < D.ia[(size_t)i1 + 1]; kk++)
                if (D.ja[(size_t)kk] == i1) dk = D.a[(size_t)kk];
              const double sgn = (dk < 0) ? -1.0 : 1.0;
              double sum = 0.0;
              for (int64_t kk = D.ia[(size_t)i1]; kk < D.ia[(size_t)i1 + 1]; kk++) {
                const int i2 = D.ja[(size_t)kk];
                if (i2 == i1) continue;
                if ((find(rc, i2) >= 0 || (interp_type == 6 && i2 == i)) && sgn * D.a[(size_t)kk] < 0)
                  sum += D.a[(size_t)kk];
              }
              if (sum != 0.0) {
                const double distribute = aik / sum;
                for (int64_t kk = D.ia[(size_t)i1]; kk < D.ia[(size_t)i1 + 1]; kk++) {
                  const int i2 = D.ja[(size_t)kk];
                  if (i2 == i1) continue;
                  if (sgn * D.a[(size_t)kk] < 0) {
                    const int q2 = find(rc, i2);
                    if (q2 >= 0)
                      rv[(size_t)q2] += distribute * D.a[(size_t)kk];
                    else if (interp_type == 6 && i2 == i)
                      diagonal += distribute * D.a[(size_t)kk];
                  }
                }
              } else
                diagonal += aik;
            } else
              diagonal += aik;
          }
          for (int64_t k = O.ia[(size_t)i]; k < O.ia[(size_t)i + 1]; k++) diagonal += O.a[(size_t)k];
          if (diagonal != 0.0)
            for (size_t q = 0; q < rv.size(); q++) rv[q] /= -diagonal;
        }
        const int m = truncate_row((int)rc.size(), rc.data(), rv.data(), trunc_factor, pmax, keep);
        rc.resize((size_t)m);
        rv.resize((size_t)m);
      }
      // coarse column ids, ascending
      const int len = (int)rc.size();
      for (int q = 0; q < len; q++) rc[(size_t)q] = f2c[(size_t)rc[(size_t)q]];
      for (int a = 1; a < len; a++) {
        const int c = rc[(size_t)a];
        const double v = rv[(size_t)a];
        int bb = a - 1;
        while (bb >= 0 && rc[(size_t)bb] > c) {
          rc[(size_t)bb + 1] = rc[(size_t)bb];
          rv[(size_t)bb + 1] = rv[(size_t)bb];
          bb--;
        }
        rc[(size_t)bb + 1] = c;
        rv[(size_t)bb + 1] = v;
      }
      rowlen[(size_t)i] = len;
      oj.insert(oj.end(), rc.begin(), rc.end());
      oa.insert(oa.end(), rv.begin(), rv.end());
    }
  });
  P.nrows = n;
  P.ncols = nc;
  P.ia.assign((size_t)n + 1, 0);
  for (int i = 0; i < n; i++) P.ia[(size_t)i + 1] = P.ia[(size_t)i] + rowlen[(size_t)i];
  P.ja.resize((size_t)P.nnz());
  P.a.resize((size_t)P.nnz());
  for (int t = 0; t < nt; t++) {
    if (!used[(size_t)t] || tj[(size_t)t].empty()) continue;
    const int64_t off = P.ia[(size_t)tbeg[(size_t)t]];
    memcpy(P.ja.data() + off, tj[(size_t)t].data(), tj[(size_t)t].size() * sizeof(int));
    memcpy(P.a.data() + off, ta[(size_t)t].data(), ta[(size_t)t].size() * sizeof(double));
  }
  if (!want_rows)
    for (int i = 0; i < n; i++)
      if (cf[(size_t)i] == SF_PT) cf[(size_t)i] = F_PT;
}

// ---- coarsening types beyond PMIS, and aggressive coarsening (mirrors oracle/oracle.c statement by statement)

// S^T: row i lists, ascending, the points that strongly depend on i
void strength_transpose(int n, const Strength &S, Strength &T) {
  T.ia.assign((size_t)n + 1, 0);
  T.ja.resize(S.ja.size());
  for (size_t k = 0; k < S.ja.size(); k++) T.ia[(size_t)S.ja[k] + 1]++;
  for (int i = 0; i < n; i++) T.ia[(size_t)i + 1] += T.ia[(size_t)i];
  std::vector<int64_t> pos(T.ia.begin(), T.ia.end() - 1);
  for (int i = 0; i < n; i++)
    for (int64_t k = S.ia[(size_t)i]; k < S.ia[(size_t)i + 1]; k++) T.ja[(size_t)pos[(size_t)S.ja[(size_t)k]]++] = i;
}

// bucket lists of the Ruge-Stueben first pass (hypre_enter_on_lists / hypre_remove_point): one FIFO list per
// integer measure, the next C point is the head of the highest non-empty list
struct RsLists {
  std::vector<int> head, tail, prev, next;
  int top = 0;
  RsLists(int nb, int n) : head((size_t)nb, -1), tail((size_t)nb, -1), prev((size_t)n, -1), next((size_t)n, -1) {}
  void enter(int m, int i) {
    prev[(size_t)i] = tail[(size_t)m];
    next[(size_t)i] = -1;
    if (tail[(size_t)m] >= 0)
      next[(size_t)tail[(size_t)m]] = i;
    else
      head[(size_t)m] = i;
    tail[(size_t)m] = i;
    if (m > top) top = m;
  }
  void remove(int m, int i) {
    if (prev[(size_t)i] >= 0)
      next[(size_t)prev[(size_t)i]] = next[(size_t)i];
    else
      head[(size_t)m] = next[(size_t)i];
    if (next[(size_t)i] >= 0)
      prev[(size_t)next[(size_t)i]] = prev[(size_t)i];
    else
      tail[(size_t)m] = prev[(size_t)i];
  }
};

// classical Ruge-Stueben coarsening on the (global) strength graph: hypre_BoomerAMGCoarsenRuge.  First pass by
// measure |S^T_i| with bucket lists; second pass (types 1, 3, 6): strong F-F pairs must share a C point.
// Sequential by definition -- HYPRE's device build offers PMIS only; these types exist so that inputs which ask
// for them (the upstream sample: coarsen_type 6, etc/hypre_app.yaml:35) get what they ask for.
void ruge_stueben(int n, const Strength &S, bool second_pass, std::vector<int> &cf) {
  Strength T;
  strength_transpose(n, S, T);
  std::vector<int> measure((size_t)n);
  int maxm = 0;
  for (int i = 0; i < n; i++) {
    measure[(size_t)i] = (int)(T.ia[(size_t)i + 1] - T.ia[(size_t)i]);
    maxm = std::max(maxm, measure[(size_t)i]);
  }
  RsLists q(2 * maxm + 2, n);
  cf.assign((size_t)n, 0);
  int64_t num_left = 0;
  for (int i = 0; i < n; i++) {
    if (S.ia[(size_t)i + 1] == S.ia[(size_t)i]) {
      cf[(size_t)i] = SF_PT;
      measure[(size_t)i] = 0;
    } else
      num_left++;
  }
  auto bump = [&](int p2) {  // an undecided point gains one
    q.remove(measure[(size_t)p2], p2);
    measure[(size_t)p2]++;
    q.enter(measure[(size_t)p2], p2);
  };
  for (int j = 0; j < n; j++) {
    if (cf[(size_t)j] != 0) continue;
    if (measure[(size_t)j] > 0) {
      q.enter(measure[(size_t)j], j);
    } else {
      cf[(size_t)j] = F_PT;
      num_left--;
      for (int64_t k = S.ia[(size_t)j]; k < S.ia[(size_t)j + 1]; k++) {
        const int nb = S.ja[(size_t)k];
        if (cf[(size_t)nb] != 0) continue;
        if (nb < j) {
          if (measure[(size_t)nb] > 0) q.remove(measure[(size_t)nb], nb);
          measure[(size_t)nb]++;
          q.enter(measure[(size_t)nb], nb);
        } else
          measure[(size_t)nb]++;
      }
    }
  }
  while (num_left > 0) {
    while (q.top > 0 && q.head[(size_t)q.top] < 0) q.top--;
    const int c = q.head[(size_t)q.top];
    if (c < 0) break;
    cf[(size_t)c] = C_PT;
    q.remove(measure[(size_t)c], c);
    measure[(size_t)c] = 0;
    num_left--;
    for (int64_t j = T.ia[(size_t)c]; j < T.ia[(size_t)c + 1]; j++) {
      const int nb = T.ja[(size_t)j];
      if (cf[(size_t)nb] != 0) continue;
      cf[(size_t)nb] = F_PT;
      q.remove(measure[(size_t)nb], nb);
      num_left--;
      for (int64_t k = S.ia[(size_t)nb]; k < S.ia[(size_t)nb + 1]; k++)
        if (cf[(size_t)S.ja[(size_t)k]] == 0) bump(S.ja[(size_t)k]);
    }
    for (int64_t j = S.ia[(size_t)c]; j < S.ia[(size_t)c + 1]; j++) {
      const int nb = S.ja[(size_t)j];
      if (cf[(size_t)nb] != 0) continue;
      q.remove(measure[(size_t)nb], nb);
      measure[(size_t)nb]--;
      if (measure[(size_t)nb] > 0)
        q.enter(measure[(size_t)nb], nb);
      else {
        cf[(size_t)nb] = F_PT;
        num_left--;
        for (int64_t k = S.ia[(size_t)nb]; k < S.ia[(size_t)nb + 1]; k++)
          if (cf[(size_t)S.ja[(size_t)k]] == 0) bump(S.ja[(size_t)k]);
      }
    }
  }
  if (!second_pass) return;
  std::vector<int> mark((size_t)n, -1);
  for (int i = 0; i < n; i++) {
    if (cf[(size_t)i] != F_PT) continue;
    int tentative = -1;
    for (;;) {
      for (int64_t k = S.ia[(size_t)i]; k < S.ia[(size_t)i + 1]; k++)
        if (cf[(size_t)S.ja[(size_t)k]] == C_PT) mark[(size_t)S.ja[(size_t)k]] = i;
      int lonely = -1;  // first strong F neighbour that shares no C point with i
      for (int64_t k = S.ia[(size_t)i]; k < S.ia[(size_t)i + 1] && lonely < 0; k++) {
        const int j = S.ja[(size_t)k];
        if (cf[(size_t)j] != F_PT) continue;
        bool shared = false;
        for (int64_t kk = S.ia[(size_t)j]; kk < S.ia[(size_t)j + 1]; kk++) {
          const int c = S.ja[(size_t)kk];
          if (mark[(size_t)c] == i && cf[(size_t)c] == C_PT) {
            shared = true;
            break;
          }
        }
        if (!shared) lonely = j;
      }
      if (lonely < 0) break;
      if (tentative < 0) {
        tentative = lonely;
        cf[(size_t)lonely] = C_PT;
      } else {
        cf[(size_t)i] = C_PT;
        cf[(size_t)tentative] = F_PT;
        break;
      }
    }
  }
}

bool coarsen_type_restated(int type) {
  return type == 8 || type == 9 || type == 10 || type == 11 || type == 6 || type == 1 || type == 3 || type == 0 || type == 7;
}

// CLJP (par_coarsen.c hypre_BoomerAMGCoarsen; oracle/oracle.c cljp for the statement of the algorithm):
// w = |S^T row| + the global Park-Miller stream; rounds of { independent set of the undecided points -> C;
// H1: edges out of a new C point leave, w-- at their undecided ends; H2: an undecided row loses its edges to C
// points, and the edges to undecided points that share one of its C points, w-- there; w < 1 -> F }.
// Within a round every step reads what the previous step left (counts and flags commute): host threads.
void cljp(int n, const Strength &S, std::vector<int> &cf) {
  const int64_t nnz = (int64_t)S.ja.size();
  std::vector<char> gone((size_t)nnz, 0);
  std::vector<int> dec((size_t)n, 0);  // pending decrements of a round (integers: order does not matter)
  std::vector<double> measure((size_t)n, 0.0);
  for (int64_t k = 0; k < nnz; k++) measure[(size_t)S.ja[(size_t)k]] += 1.0;
  ParkMiller rng(2747);
  for (int i = 0; i < n; i++) measure[(size_t)i] += rng.next();
  cf.assign((size_t)n, 0);
  std::vector<int> graph;
  graph.reserve((size_t)n);
  for (int i = 0; i < n; i++) {
    if (measure[(size_t)i] < 1.0)
      cf[(size_t)i] = (S.ia[(size_t)i + 1] == S.ia[(size_t)i]) ? SF_PT : F_PT;
    else
      graph.push_back(i);
  }
  std::vector<signed char> tmp((size_t)n, 0);
  const int nt = host_threads();
  std::vector<std::vector<int>> common((size_t)nt);  // per thread: common[c] == i + 1: C point c is in the row of i
  while (!graph.empty()) {
    const int64_t ng = (int64_t)graph.size();
    parallel_for(ng, [&](int64_t b, int64_t e, int) {
      for (int64_t g = b; g < e; g++) tmp[(size_t)graph[(size_t)g]] = 1;
    });
    parallel_for(ng, [&](int64_t b, int64_t e, int) {
      for (int64_t g = b; g < e; g++) {
        const int i = graph[(size_t)g];
        for (int64_t k = S.ia[(size_t)i]; k < S.ia[(size_t)i + 1]; k++) {
          const int j = S.ja[(size_t)k];
          if (cf[(size_t)j] != 0) continue;
          if (measure[(size_t)i] > measure[(size_t)j])
            __atomic_store_n(&tmp[(size_t)j], (signed char)0, __ATOMIC_RELAXED);
          else if (measure[(size_t)j] > measure[(size_t)i])
            __atomic_store_n(&tmp[(size_t)i], (signed char)0, __ATOMIC_RELAXED);
        }
      }
    });
    parallel_for(ng, [&](int64_t b, int64_t e, int) {
      for (int64_t g = b; g < e; g++)
        if (tmp[(size_t)graph[(size_t)g]] == 1) cf[(size_t)graph[(size_t)g]] = C_PT;
    });
    // H1 (rows of the new C points) and H2 (undecided rows): disjoint rows, decrements collected in `dec`
    parallel_for(ng, [&](int64_t b, int64_t e, int t) {
      std::vector<int> &cm = common[(size_t)t];
      if (cm.size() != (size_t)n) cm.assign((size_t)n, 0);
      for (int64_t g = b; g < e; g++) {
        const int i = graph[(size_t)g];
        if (cf[(size_t)i] == C_PT) {
          for (int64_t k = S.ia[(size_t)i]; k < S.ia[(size_t)i + 1]; k++) {
            if (gone[(size_t)k]) continue;
            gone[(size_t)k] = 1;
            if (cf[(size_t)S.ja[(size_t)k]] == 0) __atomic_fetch_add(&dec[(size_t)S.ja[(size_t)k]], 1, __ATOMIC_RELAXED);
          }
          continue;
        }
        for (int64_t k = S.ia[(size_t)i]; k < S.ia[(size_t)i + 1]; k++)
          if (cf[(size_t)S.ja[(size_t)k]] == C_PT) {
            gone[(size_t)k] = 1;
            cm[(size_t)S.ja[(size_t)k]] = i + 1;
          }
        for (int64_t k = S.ia[(size_t)i]; k < S.ia[(size_t)i + 1]; k++) {
          const int j = S.ja[(size_t)k];
          if (gone[(size_t)k] || cf[(size_t)j] != 0) continue;
          for (int64_t kk = S.ia[(size_t)j]; kk < S.ia[(size_t)j + 1]; kk++)
            if (cm[(size_t)S.ja[(size_t)kk]] == i + 1) {
              gone[(size_t)k] = 1;
              __atomic_fetch_add(&dec[(size_t)j], 1, __ATOMIC_RELAXED);
              break;
            }
        }
      }
    });
    std::vector<int> next;
    next.reserve(graph.size());
    for (int64_t g = 0; g < ng; g++) {
      const int i = graph[(size_t)g];
      if (cf[(size_t)i] == C_PT) continue;
      for (; dec[(size_t)i] > 0; dec[(size_t)i]--) measure[(size_t)i] -= 1.0;  // one at a time, as the oracle does
      if (measure[(size_t)i] < 1.0)
        cf[(size_t)i] = F_PT;
      else
        next.push_back(i);
    }
    graph.swap(next);
  }
}

// HYPRE_BoomerAMGSetCoarsenType (src/HypreSystem.cpp:125-126): 8 PMIS; 10 HMIS / 11 = one-pass Ruge-Stueben;
// 6 Falgout / 1 / 3 = two-pass Ruge-Stueben.  HMIS and Falgout finish with PMIS / CLJP on what the Ruge-Stueben
// pass leaves undecided; coarsening sees the whole graph here, where nothing is left --
// as on a single HYPRE rank.
void coarsen_by_type(int type, int n, const Strength &S, std::vector<int> &cf) {
  if (type == 8 || type == 9)  // 9 = PMIS with one global random stream: what 8 is here anyway
    pmis(n, S, 0, cf);
  else if (type == 0 || type == 7)  // 7 = CLJP with one global random stream: the only kind this library draws
    cljp(n, S, cf);
  else if (type == 10 || type == 11)
    ruge_stueben(n, S, false, cf);
  else if (type == 6 || type == 1 || type == 3)
    ruge_stueben(n, S, true, cf);
  else
    fail(4, "BoomerAMG: coarsen_type " + std::to_string(type) + " is not implemented (8, 10, 11, 6, 1, 3, 0, 7 are)");
}

// hypre_BoomerAMGCreate2ndS, num_paths 1: graph on the C points of the first coarsening; C point i depends on
// C point j != i iff j is in S_i or in S_k for some k in S_i.  Coarse indices, columns ascending.
void second_strength(int n, const Strength &S, const std::vector<int> &cf, Strength &S2, int &nc_out) {
  std::vector<int> f2c((size_t)n, -1);
  int nc = 0;
  for (int i = 0; i < n; i++)
    if (cf[(size_t)i] == C_PT) f2c[(size_t)i] = nc++;
  nc_out = nc;
  std::vector<int> crow((size_t)nc);
  for (int i = 0; i < n; i++)
    if (f2c[(size_t)i] >= 0) crow[(size_t)f2c[(size_t)i]] = i;
  const int nt = host_threads();
  std::vector<std::vector<int>> tj((size_t)nt);
  std::vector<int> rowlen((size_t)nc, 0);
  std::vector<int64_t> tbeg((size_t)nt, 0);
  std::vector<char> used((size_t)nt, 0);
  parallel_for(nc, [&](int64_t b, int64_t e, int t) {
    used[(size_t)t] = 1;
    tbeg[(size_t)t] = b;
    std::vector<int> &out = tj[(size_t)t];
    std::vector<int> row;
    for (int64_t ci = b; ci < e; ci++) {
      const int i = crow[(size_t)ci];
      row.clear();
      auto add = [&](int j) {
        const int cj = f2c[(size_t)j];
        if (cj >= 0 && cj != ci) row.push_back(cj);
      };
      for (int64_t k = S.ia[(size_t)i]; k < S.ia[(size_t)i + 1]; k++) {
        const int k1 = S.ja[(size_t)k];
        add(k1);
        for (int64_t kk = S.ia[(size_t)k1]; kk < S.ia[(size_t)k1 + 1]; kk++) add(S.ja[(size_t)kk]);
      }
      std::sort(row.begin(), row.end());
      row.erase(std::unique(row.begin(), row.end()), row.end());
      rowlen[(size_t)ci] = (int)row.size();
      out.insert(out.end(), row.begin(), row.end());
    }
  });
  S2.ia.assign((size_t)nc + 1, 0);
  for (int q = 0; q < nc; q++) S2.ia[(size_t)q + 1] = S2.ia[(size_t)q] + rowlen[(size_t)q];
  S2.ja.resize((size_t)S2.ia[(size_t)nc]);
  for (int t = 0; t < nt; t++)
    if (used[(size_t)t] && !tj[(size_t)t].empty())
      memcpy(S2.ja.data() + S2.ia[(size_t)tbeg[(size_t)t]], tj[(size_t)t].data(), tj[(size_t)t].size() * sizeof(int));
}

// aggressive coarsening of one level (par_amg_setup.c, level < agg_num_levels; src/HypreSystem.cpp:215-219):
// coarsen with S, coarsen the C points again with the second-generation graph, keep what survives both
void coarsen_aggressive(int type, int n, const Strength &S, std::vector<int> &cf) {
  coarsen_by_type(type, n, S, cf);
  Strength S2;
  int nc = 0;
  second_strength(n, S, cf, S2, nc);
  std::vector<int> cf2;
  coarsen_by_type(type, nc, S2, cf2);
  int q = 0;
  for (int i = 0; i < n; i++)
    if (cf[(size_t)i] == C_PT) {
      if (cf2[(size_t)q] != C_PT) cf[(size_t)i] = cf2[(size_t)q];
      q++;
    }
}

// Multipass interpolation (hypre_BoomerAMGBuildMultipass; agg_interp_type 4, src/HypreSystem.cpp:220-224): see
// oracle/oracle.c build_multipass for the formulas.  Pass by pass; the rows of one pass are independent (threads),
// every row is accumulated by one thread in the oracle's order.
void build_multipass(const ParCSR &A, const Strength &S, std::vector<int> &cf, double trunc_factor, int pmax,
                     HostCSR &P, int &nc_out) {
  const HostCSR &D = A.diag;
  const int n = D.nrows;
  std::vector<int> f2c((size_t)n, -1);
  int nc = 0;
  for (int i = 0; i < n; i++)
    if (cf[(size_t)i] == C_PT) f2c[(size_t)i] = nc++;
  nc_out = nc;
  std::vector<int> assigned((size_t)n, -1), rlen((size_t)n, 0);
  std::vector<int64_t> rstart((size_t)n, 0);
  std::vector<int> pool_c;  // rows in the order they were built: coarse columns / weights in discovery order
  std::vector<double> pool_v;
  pool_c.reserve((size_t)n);
  pool_v.reserve((size_t)n);
  int64_t remaining = 0;
  for (int i = 0; i < n; i++) {
    if (cf[(size_t)i] == C_PT) {
      assigned[(size_t)i] = 0;
      rstart[(size_t)i] = (int64_t)pool_c.size();
      rlen[(size_t)i] = 1;
      pool_c.push_back(f2c[(size_t)i]);
      pool_v.push_back(1.0);
    } else if (cf[(size_t)i] != SF_PT)
      remaining++;
  }
  const int nt = host_threads();
  std::vector<std::vector<int>> cpos((size_t)nt);
  std::vector<int> list;
  for (int pass = 1; remaining > 0; pass++) {
    list.clear();
    for (int i = 0; i < n; i++) {
      if (assigned[(size_t)i] != -1 || cf[(size_t)i] == SF_PT) continue;
      for (int64_t k = S.ia[(size_t)i]; k < S.ia[(size_t)i + 1]; k++)
        if (assigned[(size_t)S.ja[(size_t)k]] == pass - 1) {
          list.push_back(i);
          break;
        }
    }
    const int64_t nl = (int64_t)list.size();
    if (nl == 0) break;
    std::vector<std::vector<int>> tc((size_t)nt);
    std::vector<std::vector<double>> tv((size_t)nt);
    std::vector<int> newlen((size_t)nl, 0);
    std::vector<int64_t> tbeg((size_t)nt, 0);
    std::vector<char> used((size_t)nt, 0);
    parallel_for(nl, [&](int64_t b, int64_t e, int t) {
      used[(size_t)t] = 1;
      tbeg[(size_t)t] = b;
      std::vector<int> &pos = cpos[(size_t)t];
      if (pos.size() != (size_t)nc) pos.assign((size_t)nc, -1);
      std::vector<int> &oc = tc[(size_t)t];
      std::vector<double> &ov = tv[(size_t)t];
      for (int64_t q = b; q < e; q++) {
        const int i = list[(size_t)q];
        const size_t base = oc.size();
        double diagonal = 0.0, sum_N = 0.0, sum_J = 0.0;
        int64_t ks = S.ia[(size_t)i];
        const int64_t kse = S.ia[(size_t)i + 1];
        for (int64_t k = D.ia[(size_t)i]; k < D.ia[(size_t)i + 1]; k++) {
          const int j = D.ja[(size_t)k];
          if (j == i) {
            diagonal = D.a[(size_t)k];
            continue;
          }
          sum_N += D.a[(size_t)k];
          // S_i is a subsequence of A's row: two-pointer membership test
          while (ks < kse && S.ja[(size_t)ks] < j) ks++;
          const bool strong = ks < kse && S.ja[(size_t)ks] == j;
          if (!strong || assigned[(size_t)j] != pass - 1) continue;
          sum_J += D.a[(size_t)k];
          const int64_t r0 = rstart[(size_t)j];
          for (int w = 0; w < rlen[(size_t)j]; w++) {
            const int c = pool_c[(size_t)(r0 + w)];
            if (pos[(size_t)c] < 0) {
              pos[(size_t)c] = (int)(oc.size() - base);
              oc.push_back(c);
              ov.push_back(0.0);
            }
            ov[base + (size_t)pos[(size_t)c]] += D.a[(size_t)k] * pool_v[(size_t)(r0 + w)];
          }
        }
        for (int64_t k = A.offd.ia[(size_t)i]; k < A.offd.ia[(size_t)i + 1]; k++) sum_N += A.offd.a[(size_t)k];
        const double alfa = (sum_J * diagonal != 0.0) ? -sum_N / (sum_J * diagonal) : 0.0;
        const int len = (int)(oc.size() - base);
        for (int w = 0; w < len; w++) {
          ov[base + (size_t)w] *= alfa;
          pos[(size_t)oc[base + (size_t)w]] = -1;
        }
        newlen[(size_t)q] = len;
      }
    });
    // append the pass's rows to the pool (list order) -- only now do its points count as reached
    int64_t at = (int64_t)pool_c.size();
    for (int64_t q = 0; q < nl; q++) {
      rstart[(size_t)list[(size_t)q]] = at;
      rlen[(size_t)list[(size_t)q]] = newlen[(size_t)q];
      at += newlen[(size_t)q];
    }
    pool_c.resize((size_t)at);
    pool_v.resize((size_t)at);
    for (int t = 0; t < nt; t++)
      if (used[(size_t)t] && !tc[(size_t)t].empty()) {
        const int64_t off = rstart[(size_t)list[(size_t)tbeg[(size_t)t]]];
        memcpy(pool_c.data() + off, tc[(size_t)t].data(), tc[(size_t)t].size() * sizeof(int));
        memcpy(pool_v.data() + off, tv[(size_t)t].data(), tv[(size_t)t].size() * sizeof(double));
      }
    for (int64_t q = 0; q < nl; q++) assigned[(size_t)list[(size_t)q]] = pass;
    remaining -= nl;
  }
  // truncation and sort row by row (in place in the pool), then compaction into P
  std::vector<int> flen((size_t)n, 0);
  parallel_for(n, [&](int64_t b, int64_t e, int) {
    std::vector<char> keep;
    for (int64_t i = b; i < e; i++) {
      int len = rlen[(size_t)i];
      int *rc = pool_c.data() + rstart[(size_t)i];
      double *rv = pool_v.data() + rstart[(size_t)i];
      if (cf[(size_t)i] != C_PT && len > 0) len = truncate_row(len, rc, rv, trunc_factor, pmax, keep);
      for (int a = 1; a < len; a++) {
        const int c = rc[a];
        const double v = rv[a];
        int bb = a - 1;
        while (bb >= 0 && rc[bb] > c) {
          rc[bb + 1] = rc[bb];
          rv[bb + 1] = rv[bb];
          bb--;
        }
        rc[bb + 1] = c;
        rv[bb + 1] = v;
      }
      flen[(size_t)i] = len;
    }
  });
  P.nrows = n;
  P.ncols = nc;
  P.ia.assign((size_t)n + 1, 0);
  for (int i = 0; i < n; i++) P.ia[(size_t)i + 1] = P.ia[(size_t)i] + flen[(size_t)i];
  P.ja.resize((size_t)P.nnz());
  P.a.resize((size_t)P.nnz());
  parallel_for(n, [&](int64_t b, int64_t e, int) {
    for (int64_t i = b; i < e; i++) {
      if (!flen[(size_t)i]) continue;
      memcpy(P.ja.data() + P.ia[(size_t)i], pool_c.data() + rstart[(size_t)i], (size_t)flen[(size_t)i] * sizeof(int));
      memcpy(P.a.data() + P.ia[(size_t)i], pool_v.data() + rstart[(size_t)i], (size_t)flen[(size_t)i] * sizeof(double));
    }
  });
  for (int i = 0; i < n; i++)
    if (cf[(size_t)i] == SF_PT) cf[(size_t)i] = F_PT;
}

void dense_inverse(int n, std::vector<double> &M, std::vector<double> &inv) {
  inv.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) inv[(size_t)i * n + i] = 1.0;
  for (int c = 0; c < n; c++) {
    int piv = c;
    for (int r = c + 1; r < n; r++)
      if (std::fabs(M[(size_t)r * n + c]) > std::fabs(M[(size_t)piv * n + c])) piv = r;
    if (piv != c)
      for (int j = 0; j < n; j++) {
        std::swap(M[(size_t)c * n + j], M[(size_t)piv * n + j]);
        std::swap(inv[(size_t)c * n + j], inv[(size_t)piv * n + j]);
      }
    const double d = M[(size_t)c * n + c];
    if (d == 0.0) continue;
    const double id = 1.0 / d;
    for (int j = 0; j < n; j++) {
      M[(size_t)c * n + j] *= id;
      inv[(size_t)c * n + j] *= id;
    }
    for (int r = 0; r < n; r++) {
      if (r == c) continue;
      const double f = M[(size_t)r * n + c];
      if (f == 0.0) continue;
      for (int j = 0; j < n; j++) {
        M[(size_t)r * n + j] -= f * M[(size_t)c * n + j];
        inv[(size_t)r * n + j] -= f * inv[(size_t)c * n + j];
      }
    }
  }
}

// l1 norms; chunk = hybrid-GS chunk ("thread") size; cf_ext = C/F type of the halo columns
void level_norms(const ParCSR &A, const std::vector<int> &cf, const std::vector<int> &cf_ext, int chunk,
                 std::vector<double> &diag, std::vector<double> &l1gs, std::vector<double> &l1jac) {
  const HostCSR &D = A.diag, &O = A.offd;
  const int n = D.nrows;
  diag.assign((size_t)n, 0.0);
  l1gs.assign((size_t)n, 0.0);
  l1jac.assign((size_t)n, 0.0);
  const bool has_cf = !cf.empty();
  parallel_for(n, [&](int64_t b, int64_t e, int) {
    for (int64_t i = b; i < e; i++) {
      const int64_t cs = (i / chunk) * chunk, ce = cs + chunk;
      double d = 0.0, l1 = 0.0, full = 0.0;
      for (int64_t k = D.ia[(size_t)i]; k < D.ia[(size_t)i + 1]; k++) {
        const int j = D.ja[(size_t)k];
        const double av = std::fabs(D.a[(size_t)k]);
        full += av;
        if (j == i) {
          d = D.a[(size_t)k];
          l1 += av;
        } else if (j < cs || j >= ce) {
          if (!has_cf || cf[(size_t)j] == cf[(size_t)i]) l1 += 0.5 * av;
        }
      }
      for (int64_t k = O.ia[(size_t)i]; k < O.ia[(size_t)i + 1]; k++) {
        const double av = std::fabs(O.a[(size_t)k]);
        full += av;
        if (!has_cf || cf_ext[(size_t)O.ja[(size_t)k]] == cf[(size_t)i]) l1 += 0.5 * av;
      }
      if (l1 <= 4.0 / 3.0 * std::fabs(d)) l1 = std::fabs(d);
      if (d < 0) {
        l1 = -l1;
        full = -full;
      }
      diag[(size_t)i] = d;
      l1gs[(size_t)i] = l1;
      l1jac[(size_t)i] = full;
    }
  });
}

}  // namespace hs
using namespace hs;

void host_transpose(const HostCSR &A, HostCSR &T) {
  T.nrows = A.ncols;
  T.ncols = A.nrows;
  T.ia.assign((size_t)A.ncols + 1, 0);
  const int64_t nnz = A.nnz();
  for (int64_t k = 0; k < nnz; k++) T.ia[(size_t)A.ja[(size_t)k] + 1]++;
  for (int i = 0; i < A.ncols; i++) T.ia[(size_t)i + 1] += T.ia[(size_t)i];
  T.ja.resize((size_t)nnz);
  T.a.resize((size_t)nnz);
  std::vector<int64_t> pos(T.ia.begin(), T.ia.end() - 1);
  for (int i = 0; i < A.nrows; i++)
    for (int64_t k = A.ia[(size_t)i]; k < A.ia[(size_t)i + 1]; k++) {
      const int64_t q = pos[(size_t)A.ja[(size_t)k]]++;
      T.ja[(size_t)q] = i;
      T.a[(size_t)q] = A.a[(size_t)k];
    }
}

// Gustavson row products with a dense accumulator per thread; the accumulation
// order inside a row is (k ascending in A's row) x (stored order of B's row k),
// output columns ascending
void host_spgemm(const HostCSR &A, const HostCSR &B, HostCSR &C) {
  const int n = A.nrows, m = B.ncols;
  C.nrows = n;
  C.ncols = m;
  C.ia.assign((size_t)n + 1, 0);
  std::vector<int> cnt((size_t)n, 0);
  parallel_for(n, [&](int64_t b, int64_t e, int) {
    std::vector<int> mark((size_t)m, -1);
    for (int64_t i = b; i < e; i++) {
      int c = 0;
      for (int64_t ka = A.ia[(size_t)i]; ka < A.ia[(size_t)i + 1]; ka++) {
        const int kr = A.ja[(size_t)ka];
        for (int64_t kb = B.ia[(size_t)kr]; kb < B.ia[(size_t)kr + 1]; kb++) {
          const int j = B.ja[(size_t)kb];
          if (mark[(size_t)j] != i) {
            mark[(size_t)j] = (int)i;
            c++;
          }
        }
      }
      cnt[(size_t)i] = c;
    }
  }, 16);
  for (int i = 0; i < n; i++) C.ia[(size_t)i + 1] = C.ia[(size_t)i] + cnt[(size_t)i];
  C.ja.resize((size_t)C.nnz());
  C.a.resize((size_t)C.nnz());
  parallel_for(n, [&](int64_t b, int64_t e, int) {
    std::vector<int> mark((size_t)m, -1);
    std::vector<double> acc((size_t)m, 0.0);
    for (int64_t i = b; i < e; i++) {
      int64_t q = C.ia[(size_t)i];
      for (int64_t ka = A.ia[(size_t)i]; ka < A.ia[(size_t)i + 1]; ka++) {
        const int kr = A.ja[(size_t)ka];
        const double av = A.a[(size_t)ka];
        for (int64_t kb = B.ia[(size_t)kr]; kb < B.ia[(size_t)kr + 1]; kb++) {
          const int j = B.ja[(size_t)kb];
          if (mark[(size_t)j] != i) {
            mark[(size_t)j] = (int)i;
            C.ja[(size_t)q++] = j;
            acc[(size_t)j] = av * B.a[(size_t)kb];
          } else
            acc[(size_t)j] += av * B.a[(size_t)kb];
        }
      }
      std::sort(C.ja.begin() + C.ia[(size_t)i], C.ja.begin() + C.ia[(size_t)i + 1]);
      for (int64_t k = C.ia[(size_t)i]; k < C.ia[(size_t)i + 1]; k++) C.a[(size_t)k] = acc[(size_t)C.ja[(size_t)k]];
    }
  }, 16);
}

long long BoomerAMG::default_device_min_rows() {
  const char *e = getenv("MI_HYPRE_DEVICE_SETUP_MIN_ROWS");
  return e ? atoll(e) : 20000;
}

void BoomerAMG::use_private_self_comm() {
  own_comm = make_self_comm();
  forced_comm = own_comm.get();
}

long long BoomerAMG::effective_redundant_rows() const {
  if (p.redundant_rows >= 0) return p.redundant_rows;
  const char *e = getenv("MI_HYPRE_REDUNDANT_ROWS");
  return e ? atoll(e) : 200000;
}

void BoomerAMG::ensure_host(int level) {
  if (level < 0 || level >= (int)L.size()) return;
  AmgLevel &Lv = L[(size_t)level];
  auto fetch = [&](ParCSR *M, sk::DCsr &dev) {
    if (!M || !M->host_diag_stale) return;
    hipStream_t s = ctx().stream;
    const int nr = M->diag.nrows, ncl = M->diag.ncols;
    if (dev.nrows == M->nrows && dev.ia.p)
      dev.download(M->diag, s);
    else
      sk::solve_format_to_host(M->d_diag, M->diag, s);
    M->diag.nrows = nr;
    M->diag.ncols = ncl;
    M->host_diag_stale = false;
    M->ensure_offd_rows();
  };
  fetch(Lv.A, Lv.oA);
  fetch(Lv.Pm.get(), Lv.oP);
  fetch(Lv.Rm.get(), Lv.oR);
}

int BoomerAMG::chunk() const { return p.gs_chunk > 0 ? p.gs_chunk : ctx().gs_chunk; }

// Sum of the levels' entries over the entries of level 0.  On N > 1 ranks the GLOBAL figure (what HYPRE prints), summed
// over the ranks once at the end of Setup, where every rank is present -- the getter itself is not collective.
double BoomerAMG::operator_complexity() const {
  if (L.empty()) return 0.0;
  if (global_opcx >= 0.0) return global_opcx;
  double num = 0.0, den = 0.0;
  local_entry_counts(num, den);
  return den > 0 ? num / den : 0.0;
}

void BoomerAMG::local_entry_counts(double &tot_out, double &base_out) const {
  double tot = 0.0;
  const size_t own = tail ? L.size() - 1 : L.size();  // the stub's operator is the tail's fine level
  for (size_t l = 0; l < own; l++) tot += (double)(L[l].A->diag_nnz() + L[l].A->offd.nnz());
  if (tail) {
    double t = 0.0;
    for (const auto &l : tail->L) t += (double)(l.A->diag_nnz() + l.A->offd.nnz());
    tot += t / (double)std::max(1, my_comm().size);  // this rank's share of the redundant levels
  }
  tot_out = tot;
  base_out = (double)(L[0].A->diag_nnz() + L[0].A->offd.nnz());
}

// C-first ordering.  The hierarchy above is built in natural
// order; here every level with a C/F splitting is renumbered inside its rank:
// C points first (original order kept, so C point q IS coarse unknown q of the
// rank), F points after.  Chunks of 8 rows are then all-C or all-F (bar one
// mixed chunk per rank), a C or F relaxation pass touches one contiguous row
// range, and the two passes of a sweep stream the level's matrix once instead of
// twice.  Halo columns are renumbered by their owners (one halo exchange of the
// new positions), so the level gets a fresh col_map_offd and halo plan.  Level 0
// becomes a renumbered COPY of the caller's matrix, which stays untouched.
void BoomerAMG::apply_cf_ordering() {
  Comm &comm = my_comm();
  const size_t nlev = L.size();
  const bool timing = getenv("MI_HYPRE_SETUP_TIMING") != nullptr && comm.rank == 0;
  double tcf0 = wall_time();
  std::unique_ptr<TraceRange> trace_pos(new TraceRange("C-first: positions"));
  // old -> new local row, per level: on the device for the levels whose C/F marks live there (LazyInts: the levels
  // the device built), on the host for the others; hpos() / dpos() hand out the other side's copy when somebody needs it
  std::vector<std::vector<int>> pos(nlev);
  std::vector<DVec<int>> posd(nlev);
  std::vector<char> has_pos(nlev, 0);
  auto hpos = [&](size_t l) -> const std::vector<int> & {
    if (pos[l].empty() && posd[l].p && posd[l].n) pos[l] = posd[l].to_host();
    return pos[l];
  };
  auto dpos = [&](size_t l) -> const int * {
    if (!posd[l].p && !pos[l].empty()) posd[l].upload(pos[l]);
    return posd[l].p;
  };
  for (size_t l = 0; l < nlev; l++) {
    AmgLevel &Lv = L[l];
    if (Lv.cf.empty()) continue;
    const int n = Lv.A->nrows;
    has_pos[l] = 1;
    if (Lv.cf.on_device()) {
      hipStream_t s = ctx().stream;
      DVec<long long> crank;
      const long long nc_tot = sk::count_c_points(Lv.cf.dev(), n, crank, s);
      DVec<int> dperm((size_t)n);
      posd[l].alloc((size_t)n);
      sk::cfirst_order(Lv.cf.dev(), crank.p, n, (int)nc_tot, posd[l].p, dperm.p, s);
      Lv.nc = (int)nc_tot;
      Lv.perm.adopt_device(std::move(dperm), (size_t)n);
      continue;
    }
    const std::vector<int> &cfh = Lv.cf.host();
    pos[l].resize((size_t)n);
    std::vector<int> &permh = Lv.perm.hostw();
    permh.resize((size_t)n);
    // C points first, F points after, original order kept in both: the static partition of parallel_for is the
    // same in both passes (same n), so per-thread C counts give every thread its offsets
    const int nt = host_threads();
    std::vector<int64_t> cb((size_t)nt + 1, 0), ce((size_t)nt + 1, 0), ncnt((size_t)nt + 1, 0);
    parallel_for(n, [&](int64_t b, int64_t e, int t) {
      int64_t c = 0;
      for (int64_t i = b; i < e; i++) c += (cfh[(size_t)i] == C_PT);
      cb[(size_t)t] = b, ce[(size_t)t] = e, ncnt[(size_t)t] = c;
    });
    int64_t nc_tot = 0;
    for (int t = 0; t < nt; t++) nc_tot += ncnt[(size_t)t];
    Lv.nc = (int)nc_tot;
    std::vector<int64_t> coff((size_t)nt + 1, 0), foff((size_t)nt + 1, 0);
    {
      int64_t c = 0, f = nc_tot;
      for (int t = 0; t < nt; t++) {
        coff[(size_t)t] = c, foff[(size_t)t] = f;
        c += ncnt[(size_t)t];
        f += (ce[(size_t)t] - cb[(size_t)t]) - ncnt[(size_t)t];
      }
    }
    parallel_for(n, [&](int64_t b, int64_t e, int t) {
      int64_t c = coff[(size_t)t], f = foff[(size_t)t];
      for (int64_t i = b; i < e; i++) {
        const int q = (int)((cfh[(size_t)i] == C_PT) ? c++ : f++);
        pos[l][(size_t)i] = q;
        permh[(size_t)q] = (int)i;
      }
    });
  }
  trace_pos.reset();
  if (timing) printf("   C-first ordering: positions %.2f s\n", wall_time() - tcf0);
  auto sort_rows = [](HostCSR &M) {
    parallel_for(M.nrows, [&](int64_t b, int64_t e, int) {
      std::vector<std::pair<int, double>> row;
      for (int64_t i = b; i < e; i++) {
        const int64_t s = M.ia[(size_t)i], len = M.ia[(size_t)i + 1] - s;
        bool sorted = true;
        for (int64_t k = 1; k < len; k++)
          if (M.ja[(size_t)(s + k)] < M.ja[(size_t)(s + k - 1)]) {
            sorted = false;
            break;
          }
        if (sorted) continue;
        row.resize((size_t)len);
        for (int64_t k = 0; k < len; k++) row[(size_t)k] = {M.ja[(size_t)(s + k)], M.a[(size_t)(s + k)]};
        std::sort(row.begin(), row.end(), [](const std::pair<int, double> &x, const std::pair<int, double> &y) {
          return x.first < y.first;
        });
        for (int64_t k = 0; k < len; k++) {
          M.ja[(size_t)(s + k)] = row[(size_t)k].first;
          M.a[(size_t)(s + k)] = row[(size_t)k].second;
        }
      }
    });
  };
  // B = rows of M taken in perm order, columns mapped through colpos (may be null)
  auto permute = [&](const HostCSR &M, const std::vector<int> &perm, const int *colpos, HostCSR &B) {
    const int n = M.nrows;
    B.nrows = n;
    B.ncols = M.ncols;
    B.ia.assign((size_t)n + 1, 0);
    for (int q = 0; q < n; q++) {
      const int i = perm.empty() ? q : perm[(size_t)q];
      B.ia[(size_t)q + 1] = B.ia[(size_t)q] + (M.ia[(size_t)i + 1] - M.ia[(size_t)i]);
    }
    B.ja.resize((size_t)B.nnz());
    B.a.resize((size_t)B.nnz());
    parallel_for(n, [&](int64_t b, int64_t e, int) {
      for (int64_t q = b; q < e; q++) {
        const int i = perm.empty() ? (int)q : perm[(size_t)q];
        int64_t w = B.ia[(size_t)q];
        for (int64_t k = M.ia[(size_t)i]; k < M.ia[(size_t)i + 1]; k++, w++) {
          B.ja[(size_t)w] = colpos ? colpos[M.ja[(size_t)k]] : M.ja[(size_t)k];
          B.a[(size_t)w] = M.a[(size_t)k];
        }
      }
    });
    sort_rows(B);
  };
  for (size_t l = 0; l < nlev; l++) {
    AmgLevel &Lv = L[l];
    const double tl0 = wall_time();
    struct LevelTimer {
      bool on;
      size_t l;
      double t0;
      ~LevelTimer() {
        if (on) printf("   C-first ordering: level %zu %.2f s\n", l, wall_time() - t0);
      }
    } level_timer{timing && l < 4, l, tl0};
    TraceRange trace_lv(("C-first: level " + std::to_string(l)).c_str());
    if (!has_pos[l] && Lv.A->host_diag_stale && Lv.sA.nrows == Lv.A->nrows) {
      // a level without C/F splitting (the coarsest) keeps its ordering: fetch it before sA goes away
      const int nr = Lv.A->diag.nrows, ncl = Lv.A->diag.ncols;
      Lv.sA.download(Lv.A->diag, ctx().stream);
      Lv.A->diag.nrows = nr;
      Lv.A->diag.ncols = ncl;
      Lv.A->host_diag_stale = false;
      Lv.A->ensure_offd_rows();
    }
    if (has_pos[l]) {
      ParCSR &A = *Lv.A;
      std::unique_ptr<ParCSR> An(new ParCSR());
      An->nrows = A.nrows;
      An->row_start = A.row_start;
      An->row_end = A.row_end;
      An->row_starts = A.row_starts;
      if (Lv.sA.nnz > 0 && Lv.sA.nrows == A.nrows) {
        hipStream_t s = ctx().stream;
        sk::permute(Lv.sA, Lv.perm.dev(), dpos(l), Lv.oA, s);
        An->diag.nrows = An->diag.ncols = A.nrows;
        An->host_diag_stale = true;
        An->dev_diag_nnz = Lv.oA.nnz;
      } else {
        if (A.host_diag_stale) fail(1, "C-first ordering: a level has neither host nor device arrays");
        permute(A.diag, Lv.perm.host(), hpos(l).data(), An->diag);
      }
      Lv.sA.release();
      // halo columns: new global id = owner's start + owner's new position
      const size_t next = A.col_map_offd.size();
      std::vector<int> ext_pos;
      if (comm.size > 1 || next > 0) ext_pos = A.halo_exchange_host_int(comm, hpos(l));
      std::vector<gidx> newgid(next);
      for (size_t k = 0; k < next; k++) {
        const gidx g = A.col_map_offd[k];
        const size_t owner =
            (size_t)(std::upper_bound(A.row_starts.begin(), A.row_starts.end(), g) - A.row_starts.begin()) - 1;
        newgid[k] = A.row_starts[owner] + ext_pos[k];
      }
      std::vector<gidx> cm(newgid);
      std::sort(cm.begin(), cm.end());
      std::vector<int> colpos(next);
      for (size_t k = 0; k < next; k++)
        colpos[k] = (int)(std::lower_bound(cm.begin(), cm.end(), newgid[k]) - cm.begin());
      if (A.offd.nnz() == 0) {  // no halo block (one rank): nothing to permute -- not even the row pointers
        An->offd.nrows = A.nrows;
        if (!An->host_diag_stale) An->offd.ia.assign((size_t)A.nrows + 1, 0);  // (device-resident: ParCSR::ensure_offd_rows)
      } else {
        permute(A.offd, Lv.perm.host(), next ? colpos.data() : nullptr, An->offd);
      }
      An->offd.ncols = (int)next;
      An->col_map_offd = cm;
      An->build_halo_plan(comm);
      Lv.A_own = std::move(An);
      Lv.A = Lv.A_own.get();
      if (Lv.cf.on_device() && Lv.perm.on_device()) {
        const size_t n = Lv.cf.size();
        DVec<int> cf2(n);
        sk::gather_elems(Lv.cf.dev(), Lv.perm.dev(), 0, (int)n, 4, cf2.p, ctx().stream);
        MI_HIP(hipStreamSynchronize(ctx().stream));
        Lv.cf.adopt_device(std::move(cf2), n);
      } else {
        const std::vector<int> &cfh = Lv.cf.host(), &permh = Lv.perm.host();
        std::vector<int> cf2(cfh.size());
        parallel_for((int64_t)cf2.size(), [&](int64_t b, int64_t e, int) {
          for (int64_t q = b; q < e; q++) cf2[(size_t)q] = cfh[(size_t)permh[(size_t)q]];
        });
        Lv.cf = std::move(cf2);
      }
    }
    if (Lv.P.nrows > 0 && (has_pos[l] || (l + 1 < nlev && has_pos[l + 1]))) {
      const bool map_cols = l + 1 < nlev && has_pos[l + 1];
      if (Lv.sP.nnz > 0 && Lv.sP.nrows == Lv.P.nrows) {
        hipStream_t s = ctx().stream;
        sk::permute(Lv.sP, Lv.perm.empty() ? nullptr : Lv.perm.dev(), map_cols ? dpos(l + 1) : nullptr, Lv.oP, s);
        Lv.sP.release();
        sk::transpose(Lv.oP, Lv.oR, s);
        const int pr = Lv.P.nrows, pc = Lv.P.ncols;
        Lv.P = HostCSR();
        Lv.R = HostCSR();
        Lv.P.nrows = Lv.R.ncols = pr;
        Lv.P.ncols = Lv.R.nrows = pc;
      } else {
        HostCSR P2;
        permute(Lv.P, Lv.perm.host(), map_cols ? hpos(l + 1).data() : nullptr, P2);
        Lv.P = std::move(P2);
        host_transpose(Lv.P, Lv.R);
      }
    } else if (Lv.P.nrows > 0 && !Lv.P.ia.empty() && Lv.R.nrows == 0) {
      host_transpose(Lv.P, Lv.R);  // no renumbering on either side: R was not built yet on the device path
    }
    Lv.sA.release();
    Lv.sP.release();
  }
}

// Non-Galerkin coarse operator (HYPRE_BoomerAMGSetNonGalerkinTol, src/HypreSystem.cpp:161-176): the simplified,
// documented form shared with the oracle (oracle.c sparsify_non_galerkin; HYPRE's par_nongalerkn.c is not restated):
// with m_i = max_{j != i} |a_ij|, an off-diagonal entry is dropped iff |a_ij| < tol * min(m_i, m_j) -- small against
// both rows, so a symmetric operator stays symmetric -- and added to its row's diagonal in stored order (row sums
// are kept).  Kept entries stay in stored order.
namespace hs {
void sparsify_non_galerkin(HostCSR &A, double tol) {
  const int n = A.nrows;
  if (n == 0 || !(tol > 0.0)) return;
  std::vector<double> m((size_t)n, 0.0);
  parallel_for(n, [&](int64_t b, int64_t e, int) {
    for (int64_t i = b; i < e; i++) {
      double mx = 0.0;
      for (int64_t k = A.ia[(size_t)i]; k < A.ia[(size_t)i + 1]; k++)
        if (A.ja[(size_t)k] != i && std::fabs(A.a[(size_t)k]) > mx) mx = std::fabs(A.a[(size_t)k]);
      m[(size_t)i] = mx;
    }
  });
  HostCSR B;
  B.nrows = n;
  B.ncols = A.ncols;
  B.ia.assign((size_t)n + 1, 0);
  auto keeps = [&](int64_t i, int64_t k) {
    const int j = A.ja[(size_t)k];
    const double lim = tol * std::min(m[(size_t)i], m[(size_t)j]);
    return j == i || !(std::fabs(A.a[(size_t)k]) < lim);
  };
  parallel_for(n, [&](int64_t b, int64_t e, int) {
    for (int64_t i = b; i < e; i++) {
      int64_t c = 0;
      for (int64_t k = A.ia[(size_t)i]; k < A.ia[(size_t)i + 1]; k++) c += keeps(i, k) ? 1 : 0;
      B.ia[(size_t)i + 1] = c;
    }
  });
  for (int i = 0; i < n; i++) B.ia[(size_t)i + 1] += B.ia[(size_t)i];
  B.ja.resize((size_t)B.nnz());
  B.a.resize((size_t)B.nnz());
  parallel_for(n, [&](int64_t b, int64_t e, int) {
    for (int64_t i = b; i < e; i++) {
      int64_t w = B.ia[(size_t)i], dpos = -1;
      double lump = 0.0;
      bool first = true;
      for (int64_t k = A.ia[(size_t)i]; k < A.ia[(size_t)i + 1]; k++) {
        if (keeps(i, k)) {
          if (A.ja[(size_t)k] == i) dpos = w;
          B.ja[(size_t)w] = A.ja[(size_t)k];
          B.a[(size_t)w++] = A.a[(size_t)k];
        } else {
          lump = first ? A.a[(size_t)k] : lump + A.a[(size_t)k];
          first = false;
        }
      }
      if (!first && dpos >= 0) B.a[(size_t)dpos] = B.a[(size_t)dpos] + lump;
    }
  });
  A = std::move(B);
}
}  // namespace hs

// ---- internal locality numbering ------------------------------------------------------------------------------
// The x-cache kernels gather, per tile of <= 256 consecutive rows, the tile's unique columns; with a lexicographic
// numbering of a 3-D problem a tile is a piece of ONE grid line and gathers ~5 distinct columns per row, a
// brick-shaped tile ~2.3 (profiles/run_numbering_experiment.py: numbering the same Laplacian by 8x8x8 bricks makes
// the solve 10 % faster per iteration at 256^3).  The geometry is not known here, so compact clusters are grown on
// the matrix graph.  Round 3: graph Voronoi cells instead of sequential breadth-first balls -- rows are cut into
// segments of 2^k consecutive rows (k from the matrix: about four times the typical largest column offset, i.e. four
// z-planes of a lexicographic grid; locality_segment_shift); every row whose hashed index is 0 modulo the cluster size
// (512) is a seed; in rounds, every still unlabelled row takes the smallest label among its neighbours OF ITS SEGMENT
// labelled in the PREVIOUS round (so the result depends neither on threads nor on scheduling), until nothing changes
// (at most 64 rounds; what is farther than that from every seed forms one last cluster in natural order).  The cells
// are then ranked by their smallest row -- a sweep through each segment, so that cells sharing columns stay a few
// cells apart (L2 reuse between neighbouring tiles: a first version ranked the cells by their seeds' indices, which
// scatters neighbouring cells over the whole slab and cost 3.5 GB more HBM traffic per level-0 SpMV at 512^3) -- and
// the rows sorted by cell rank, natural order inside a cell; the coarse levels inherit the order, C points keeping
// their relative order.  Tile statistics equal those of the round-2 balls; the rounds are plain data-parallel
// passes: 512^3 takes tens of milliseconds on the device (sk::locality_labels) where the sequential balls took ~2 s
// of host threads, and the permuted operator is built on the device from the copy HYPRE_IJMatrixAssemble left there.
namespace hs {
const int LOCALITY_CLUSTER = getenv("MI_HYPRE_LOCALITY_CLUSTER") ? std::max(8, atoi(getenv("MI_HYPRE_LOCALITY_CLUSTER"))) : 512;

// Segment length (a power of two, as a shift): about four times the typical largest column offset of a row -- for a
// lexicographic n^3 grid four z-planes.  Cells are confined to segments and ranked in sweep order inside them (below).
int locality_segment_shift(const HostCSR &D) {
  const int n = D.nrows;
  if (getenv("MI_HYPRE_LOCALITY_SEGMENT")) return std::max(8, std::min(30, atoi(getenv("MI_HYPRE_LOCALITY_SEGMENT"))));
  std::vector<int64_t> off;
  const int samples = std::min(n, 4096);
  for (int q = 0; q < samples; q++) {
    const int i = (int)((int64_t)q * n / samples);
    int64_t mx = 0;
    for (int64_t k = D.ia[(size_t)i]; k < D.ia[(size_t)i + 1]; k++)
      mx = std::max<int64_t>(mx, std::llabs((int64_t)D.ja[(size_t)k] - i));
    off.push_back(mx);
  }
  if (off.empty()) return 20;
  std::nth_element(off.begin(), off.begin() + (long)off.size() / 2, off.end());
  const int64_t want = 4 * std::max<int64_t>(1, off[off.size() / 2]);
  int sh = 14;
  while (sh < 22 && ((int64_t)1 << sh) < want) sh++;
  return sh;
}

// seeds of the clustering in ascending order (exclude: rows that stay out): rows whose hashed index is 0 modulo the
// cluster size; a segment without any gets its first row that takes part
std::vector<int> locality_seeds(int n, int segshift, const std::vector<char> *exclude) {
  const unsigned long long cl = (unsigned long long)LOCALITY_CLUSTER;
  const int nseg = (int)((((int64_t)n - 1) >> segshift) + 1);
  std::vector<std::vector<int>> per_seg((size_t)std::max(nseg, 0));
  parallel_for(nseg, [&](int64_t b, int64_t e, int) {
    for (int64_t sg = b; sg < e; sg++) {
      std::vector<int> &mine = per_seg[(size_t)sg];
      const int64_t r0 = sg << segshift, r1 = std::min<int64_t>(n, (sg + 1) << segshift);
      int first = -1;
      for (int64_t i = r0; i < r1; i++) {
        if (exclude && (*exclude)[(size_t)i]) continue;
        if (first < 0) first = (int)i;
        unsigned long long z = (unsigned long long)i + 0x9E3779B97F4A7C15ULL;  // splitmix64
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        if (z % cl == 0) mine.push_back((int)i);
      }
      if (mine.empty() && first >= 0) mine.push_back(first);
    }
  });
  std::vector<int> seeds;
  for (auto &v : per_seg) seeds.insert(seeds.end(), v.begin(), v.end());
  return seeds;
}

// final labels (cell = seed rank, -1 = never reached, LOCALITY_EXCLUDED) -> order[new] = old.  Cells are ranked by their
// SMALLEST ROW: scanning the rows in the caller's order, a cell takes the next rank when its first row appears.  With
// cells confined to a segment that is a sweep through the segment -- on a lexicographic grid along x, then y, through
// a slab of four planes -- so that cells which share columns are a few cells apart in the new order (what one L2 still
// holds), as the round-2 breadth-first balls were.  Then a stable counting sort of the rows by cell rank; unreached
// rows form one last cluster, excluded rows come last, both in natural order.
void locality_sort(std::vector<int> &label, int nseeds, std::vector<int> &order) {
  const int n = (int)label.size();
  std::vector<int> minrow((size_t)nseeds, n);
  for (int i = n - 1; i >= 0; i--)
    if (label[(size_t)i] >= 0) minrow[(size_t)label[(size_t)i]] = i;
  std::vector<int> cells((size_t)nseeds);
  for (int c = 0; c < nseeds; c++) cells[(size_t)c] = c;
  std::sort(cells.begin(), cells.end(), [&](int a, int b) {
    return minrow[(size_t)a] != minrow[(size_t)b] ? minrow[(size_t)a] < minrow[(size_t)b] : a < b;
  });
  std::vector<int> rank((size_t)nseeds);
  for (int q = 0; q < nseeds; q++) rank[(size_t)cells[(size_t)q]] = q;
  parallel_for(n, [&](int64_t b, int64_t e, int) {
    for (int64_t i = b; i < e; i++) {
      int &l = label[(size_t)i];
      if (l >= 0)
        l = rank[(size_t)l];
      else if (l == -1)
        l = nseeds;
    }
  });
  const int nlab = nseeds + 2;  // cells, the unreached rest, the excluded rows
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), (int64_t)n / 65536 + 1));
  std::vector<std::vector<int64_t>> hist((size_t)nt, std::vector<int64_t>((size_t)nlab, 0));
  const int64_t per = ((int64_t)n + nt - 1) / nt;
  auto lab = [&](int i) { return label[(size_t)i] == LOCALITY_EXCLUDED ? nseeds + 1 : label[(size_t)i]; };
  {
    std::vector<std::thread> th;
    auto count = [&](int t) {
      std::vector<int64_t> &h = hist[(size_t)t];
      for (int64_t i = t * per; i < std::min<int64_t>(n, (t + 1) * per); i++) h[(size_t)lab((int)i)]++;
    };
    for (int t = 1; t < nt; t++) th.emplace_back(count, t);
    count(0);
    for (auto &x : th) x.join();
  }
  int64_t run = 0;
  for (int c = 0; c < nlab; c++)
    for (int t = 0; t < nt; t++) {
      const int64_t h = hist[(size_t)t][(size_t)c];
      hist[(size_t)t][(size_t)c] = run;
      run += h;
    }
  order.resize((size_t)n);
  {
    std::vector<std::thread> th;
    auto place = [&](int t) {
      std::vector<int64_t> &h = hist[(size_t)t];
      for (int64_t i = t * per; i < std::min<int64_t>(n, (t + 1) * per); i++) order[(size_t)h[(size_t)lab((int)i)]++] = (int)i;
    };
    for (int t = 1; t < nt; t++) th.emplace_back(place, t);
    place(0);
    for (auto &x : th) x.join();
  }
}

// exclude (optional): rows that stay out of the clustering and come LAST, in natural order (rows with halo
// entries on N > 1 ranks: the halo-free rows then form one stretch that is swept while the halo travels)
void locality_order(const HostCSR &D, std::vector<int> &order, const std::vector<char> *exclude) {
  const int n = D.nrows;
  const int segshift = locality_segment_shift(D);
  const std::vector<int> seeds = locality_seeds(n, segshift, exclude);
  const int nseeds = (int)seeds.size();
  std::vector<int> label((size_t)n, -1);
  if (exclude)
    parallel_for(n, [&](int64_t b, int64_t e, int) {
      for (int64_t i = b; i < e; i++)
        if ((*exclude)[(size_t)i]) label[(size_t)i] = LOCALITY_EXCLUDED;
    });
  for (int k = 0; k < nseeds; k++) label[(size_t)seeds[(size_t)k]] = k;
  std::vector<int> active;
  active.reserve((size_t)n);
  for (int i = 0; i < n; i++)
    if (label[(size_t)i] == -1) active.push_back(i);
  std::vector<int> next;
  for (int round = 0; round < LOCALITY_MAX_ROUNDS && !active.empty(); round++) {
    next.assign(active.size(), -1);
    parallel_for((int64_t)active.size(), [&](int64_t b, int64_t e, int) {
      for (int64_t q = b; q < e; q++) {
        const int i = active[(size_t)q];
        const int sg = i >> segshift;
        int m = -1;
        for (int64_t k = D.ia[(size_t)i]; k < D.ia[(size_t)i + 1]; k++) {
          const int j = D.ja[(size_t)k];
          if ((j >> segshift) != sg) continue;  // cells do not cross segments
          const int lj = label[(size_t)j];
          if (lj >= 0 && (m < 0 || lj < m)) m = lj;
        }
        next[(size_t)q] = m;
      }
    });
    size_t w = 0;
    for (size_t q = 0; q < active.size(); q++) {
      if (next[q] >= 0)
        label[(size_t)active[q]] = next[q];
      else
        active[w++] = active[q];
    }
    if (w == active.size()) break;  // nothing changed
    active.resize(w);
  }
  locality_sort(label, nseeds, order);
}

// B = Q A Q^T: rows of A in `order` (new -> old), columns renumbered, rows re-sorted
void permute_symmetric(const HostCSR &A, const std::vector<int> &order, HostCSR &B) {
  const int n = A.nrows;
  std::vector<int> newid((size_t)n);
  parallel_for(n, [&](int64_t b, int64_t e, int) {
    for (int64_t q = b; q < e; q++) newid[(size_t)order[(size_t)q]] = (int)q;
  });
  B.nrows = n;
  B.ncols = A.ncols;
  B.ia.assign((size_t)n + 1, 0);
  for (int q = 0; q < n; q++) {
    const int i = order[(size_t)q];
    B.ia[(size_t)q + 1] = B.ia[(size_t)q] + (A.ia[(size_t)i + 1] - A.ia[(size_t)i]);
  }
  B.ja.resize((size_t)B.nnz());
  B.a.resize((size_t)B.nnz());
  parallel_for(n, [&](int64_t b, int64_t e, int) {
    std::vector<std::pair<int, double>> row;
    for (int64_t q = b; q < e; q++) {
      const int i = order[(size_t)q];
      row.clear();
      for (int64_t k = A.ia[(size_t)i]; k < A.ia[(size_t)i + 1]; k++) row.push_back({newid[(size_t)A.ja[(size_t)k]], A.a[(size_t)k]});
      std::sort(row.begin(), row.end(), [](const std::pair<int, double> &x, const std::pair<int, double> &y) { return x.first < y.first; });
      int64_t w = B.ia[(size_t)q];
      for (auto &en : row) {
        B.ja[(size_t)w] = en.first;
        B.a[(size_t)w++] = en.second;
      }
    }
  });
}
}  // namespace hs

// one rank: large systems (or MI_HYPRE_LOCALITY_ORDER=1); N > 1 ranks (distributed setup): the same rule on the mean
// rows per rank, so that every rank decides alike
bool BoomerAMG::use_locality_order(const ParCSR &A) const {
  if (forced_comm || A.host_diag_stale) return false;
  const int size = my_comm().size;
  if (size == 1 && !A.col_map_offd.empty()) return false;
  const char *e = getenv("MI_HYPRE_LOCALITY_ORDER");
  if (e && atoi(e) == 0) return false;
  const long long rows = (long long)(A.global_rows() / std::max(1, size));
  if (e && atoi(e) > 0) return rows > 1;
  static const long long min_rows = getenv("MI_HYPRE_LOCALITY_MIN_ROWS") ? atoll(getenv("MI_HYPRE_LOCALITY_MIN_ROWS")) : 1000000;
  return rows >= min_rows;
}

void BoomerAMG::setup_host(ParCSR &A0) {
  TraceRange trace_setup("mi_hypre BoomerAMGSetup (hierarchy)");
  Comm &comm = my_comm();
  t_setup_start = wall_time();
  for (double &t : t_phase) t = 0.0;
  is_setup = false;
  host_ready = false;
  MI_REQUIRE(!A0.row_starts.empty(), "BoomerAMGSetup: matrix is not assembled");
  if (!coarsen_type_restated(p.coarsen_type))
    fail(4, "BoomerAMGSetup: coarsen_type " + std::to_string(p.coarsen_type) +
                " is not implemented (8 PMIS, 10 HMIS, 11, 6 Falgout, 1, 3, 0 CLJP, 7 are); refusing to substitute another one");
  if (p.agg_num_levels > 0 && p.agg_interp_type != 4)
    fail(4, "BoomerAMGSetup: agg_interp_type " + std::to_string(p.agg_interp_type) +
                " is not implemented (4 = multipass is); refusing to substitute another one");
  for (int k = 0; k < 3; k++) {
    const int t = p.relax_type[k];
    const bool known = t == 0 || t == 7 || t == 18 || t == 3 || t == 4 || t == 6 || t == 8 || t == 13 || t == 14 ||
                       t == 11 || t == 12 || t == 9;
    if (!known)
      fail(4, "BoomerAMGSetup: relax_type " + std::to_string(t) +
                  " is not implemented (0, 3, 4, 6, 7, 8, 9, 11, 12, 13, 14, 18 are); refusing to smooth with something else");
    // 9 = direct solve: the coarsest level only (BoomerAMG::relax would otherwise have to smooth the other
    // levels with something else, or fail inside the Krylov loop)
    if (t == 9 && k < 2)
      fail(4, std::string("BoomerAMGSetup: relax_type 9 (Gaussian elimination) is implemented for the coarsest level only, not as the ") +
                  (k == 0 ? "down" : "up") + " smoother; refusing to smooth with something else");
  }
  if (p.interp_type != 0 && p.interp_type != 3 && p.interp_type != 4 && p.interp_type != 6)
    fail(4, "BoomerAMGSetup: interp_type " + std::to_string(p.interp_type) +
                " is not implemented (0 classical modified, 3 direct, 4 multipass, 6 extended+i are); refusing to substitute another one");
  if (p.smooth_num_levels > 0 && p.smooth_type != 5)
    fail(4, "BoomerAMGSetup: smooth_type " + std::to_string(p.smooth_type) + " on " + std::to_string(p.smooth_num_levels) +
                " level(s) is not implemented (5 = ILU is); refusing to smooth with something else");
  if (p.smooth_num_levels > 0 && (p.ilu_type != 0 || p.ilu_level < 0))
    fail(4, "BoomerAMGSetup: ILU smoother type " + std::to_string(p.ilu_type) + " / level of fill " +
                std::to_string(p.ilu_level) + " is not implemented (block-Jacobi ILU(k) = type 0, level k >= 0 is)");
  if (device_min_rows >= 0) dev_arena_hint((size_t)13 * 12 * (size_t)(A0.diag_nnz() + A0.offd.nnz()));
  if (comm.size > 1) input_order.clear();
  if (comm.size > 1 && can_build_distributed()) {
    build_distributed(A0);
  } else if (comm.size > 1) {
    build_replicated(A0);
  } else {
    tail.reset();
    tail_A.reset();
    input_order.clear();
    Aq_own.reset();
    ParCSR *Ain = &A0;
    pending_sA0.release();
    if (use_locality_order(A0)) {
      TraceRange trace_q("setup: internal numbering");
      const double tq0 = wall_time();
      Aq_own.reset(new ParCSR());
      ParCSR &Q = *Aq_own;
      const int n0 = A0.nrows;
      // on the device when the level will be built there anyway: the clustering rounds are data-parallel passes,
      // and Q A Q^T is permuted from the copy HYPRE_IJMatrixAssemble left in HBM -- neither the host's copy of the
      // permuted operator (0.94 G entries at 512^3) nor its upload exist any more
      const bool on_dev = device_min_rows >= 0 && n0 >= device_min_rows && A0.on_device && A0.d_diag.nrows == n0 &&
                          A0.d_diag.nnz == A0.diag.nnz() && !A0.d_diag.rowmap.p;
      if (on_dev) {
        hipStream_t s = ctx().stream;
        sk::DCsr raw;
        std::unique_ptr<TraceRange> tr(new TraceRange("numbering: plain CSR copy + segment length"));
        sk::from_solve_format(A0.d_diag, raw, s);
        const int segshift = locality_segment_shift(A0.diag);
        tr.reset(), tr.reset(new TraceRange("numbering: seeds, label rounds, order (device)"));
        // seeds, rounds, cell ranks and the rows' order all on the device (round 4; the labels used to travel to the host
        // for a counting sort there, the order back: ~0.6 s of host time at 512^3)
        DVec<int> dorder;
        int nseeds = 0, rounds = 0;
        if (!sk::locality_order_device(raw, segshift, LOCALITY_CLUSTER, LOCALITY_MAX_ROUNDS, dorder, nseeds, rounds, s)) {
          const std::vector<int> seeds = locality_seeds(n0, segshift, nullptr);
          std::vector<int> label, order_h;
          rounds = sk::locality_labels(raw, seeds.data(), (int)seeds.size(), nullptr, segshift, LOCALITY_MAX_ROUNDS, label, s);
          nseeds = (int)seeds.size();
          locality_sort(label, nseeds, order_h);
          dorder.upload(order_h);
        }
        tr.reset(), tr.reset(new TraceRange("numbering: Q A Q^T"));
        DVec<int> dpos((size_t)n0);
        sk::invert_permutation(dorder.p, n0, dpos.p, s);
        sk::permute(raw, dorder.p, dpos.p, pending_sA0, s);
        MI_HIP(hipStreamSynchronize(s));
        input_order.adopt_device(std::move(dorder), (size_t)n0);
        tr.reset();
        Q.diag.nrows = Q.diag.ncols = n0;
        Q.host_diag_stale = true;
        Q.dev_diag_nnz = pending_sA0.nnz;
        if (getenv("MI_HYPRE_SETUP_TIMING"))
          printf("   locality numbering on the device: segments of 2^%d rows, %d seeds, %d rounds\n", segshift, nseeds, rounds);
      } else {
        std::vector<int> order_h;
        locality_order(A0.diag, order_h, nullptr);
        permute_symmetric(A0.diag, order_h, Q.diag);
        input_order = std::move(order_h);
      }
      Q.nrows = A0.nrows;
      Q.row_start = A0.row_start;
      Q.row_end = A0.row_end;
      Q.row_starts = A0.row_starts;
      Q.offd.nrows = A0.nrows;
      Q.offd.ncols = 0;
      if (!Q.host_diag_stale) Q.offd.ia.assign((size_t)A0.nrows + 1, 0);  // (device-resident: ParCSR::ensure_offd_rows)
      Q.build_halo_plan(comm);
      Ain = Aq_own.get();
      if (getenv("MI_HYPRE_SETUP_TIMING")) printf("   locality numbering of the input: %.2f s\n", wall_time() - tq0);
    }
    build_natural(*Ain);
    const double tp0 = wall_time();
    {
      TraceRange tr("setup: C-first ordering");
      apply_cf_ordering();
    }
    TraceRange trace_w("setup: transfer operator wrappers + level-0 permutation");
    make_local_transfer_operators();
    if (!input_order.empty()) {
      // level 0 rows -> caller rows
      AmgLevel &L0 = L[0];
      if (L0.perm.empty()) {
        L0.perm = input_order.host();
      } else if (L0.perm.on_device() && input_order.on_device()) {
        const size_t n = L0.perm.size();
        DVec<int> composed(n);
        sk::gather_elems(input_order.dev(), L0.perm.dev(), 0, (int)n, 4, composed.p, ctx().stream);
        MI_HIP(hipStreamSynchronize(ctx().stream));
        L0.perm.adopt_device(std::move(composed), n);
      } else {
        const std::vector<int> &oh = input_order.host();
        std::vector<int> &ph = L0.perm.hostw();
        parallel_for((int64_t)ph.size(), [&](int64_t b, int64_t e, int) {
          for (int64_t q = b; q < e; q++) ph[(size_t)q] = oh[(size_t)ph[(size_t)q]];
        });
      }
      if (L0.A != Aq_own.get()) Aq_own.reset();  // level 0 is its C-first copy: the intermediate matrix can go
    }
    t_phase[4] += wall_time() - tp0;
  }
  {
    TraceRange tr("setup: norms + coarsest inverse");
    finish_host();
  }
  global_opcx = -1.0;
  if (comm.size > 1 && !L.empty()) {  // the operator complexity HYPRE would print: summed over the ranks, here where all are present
    double v[2] = {0.0, 0.0};
    local_entry_counts(v[0], v[1]);
    comm.allreduce_host(v, 2, CommDType::F64, CommOp::SUM);
    global_opcx = v[1] > 0 ? v[0] / v[1] : 0.0;
  }
  t_phase[5] = wall_time() - t_setup_start;
  host_ready = true;
  if (getenv("MI_HYPRE_SETUP_TIMING") && comm.rank == 0)
    // (wall time per phase of the hierarchy build; on levels the device builds this is kernel time -- the split into
    // device-busy and device-idle time comes from a trace: profiles/setup_split.py)
    printf("mi_hypre hierarchy build (wall per phase): strength %.2f  pmis %.2f  interp %.2f  galerkin %.2f  ordering/slicing %.2f  total %.2f s\n",
           t_phase[0], t_phase[1], t_phase[2], t_phase[3], t_phase[4], t_phase[5]);
}

// the coarsening loop in natural ordering (single communicator rank, or the
// global operator in the replicated multi-rank setup)
void BoomerAMG::build_natural(ParCSR &A0) {
  Comm &comm = my_comm();
  MI_REQUIRE(comm.size == 1 && A0.col_map_offd.empty(), "build_natural expects a single-rank operator");
  L.clear();
  L.reserve((size_t)std::max(1, p.max_levels));
  L.emplace_back();
  L[0].A = &A0;
  if (pending_sA0.nrows == A0.nrows && pending_sA0.nrows > 0 && A0.host_diag_stale) L[0].sA = std::move(pending_sA0);
  pending_sA0.release();

  int l = 0;
  stopped_by_rows = false;
  while (l < p.max_levels - 1 && L[(size_t)l].A->global_rows() > p.max_coarse_size) {
    if (stop_rows > 0 && l >= 1 && L[(size_t)l].A->global_rows() <= stop_rows) {
      stopped_by_rows = true;  // the caller continues from here with a redundant hierarchy
      break;
    }
    ParCSR &A = *L[(size_t)l].A;
    const int n = A.nrows;
    AmgLevel &Lv = L[(size_t)l];
    const bool on_device = device_min_rows >= 0 && n >= device_min_rows;
    // aggressive coarsening (level < agg_num_levels) and the Ruge-Stueben family are host algorithms: on a level
    // the device builds, the strength graph still comes from the device and the Galerkin product stays there
    const bool aggressive = l < p.agg_num_levels;
    const bool host_coarsen = aggressive || (p.coarsen_type != 8 && p.coarsen_type != 9);
    Strength S;
    std::vector<int> cf;
    sk::DCsr dS;
    DVec<int> dcf;
    const std::string lname = "level " + std::to_string(l) + ": ";
    std::unique_ptr<TraceRange> trace_level(new TraceRange((lname + "strength + coarsening").c_str()));
    double tp0 = wall_time();
    auto strength_to_host = [&](hipStream_t s) {
      S.ia.resize((size_t)n + 1);
      S.ja.resize((size_t)dS.nnz);
      d2h(S.ia.data(), dS.ia.p, ((size_t)n + 1) * sizeof(int64_t), s);
      if (dS.nnz) d2h(S.ja.data(), dS.ja.p, (size_t)dS.nnz * sizeof(int), s);
      MI_HIP(hipStreamSynchronize(s));
    };
    if (on_device) {
      // strength graph and PMIS splitting on the device (integer/compare work, identical results)
      hipStream_t s = ctx().stream;
      if (Lv.sA.nrows != n) Lv.sA.upload(A.diag, s);
      sk::strength(Lv.sA, p.strong_threshold, p.max_row_sum, dS, s);
      t_phase[0] += wall_time() - tp0;
      tp0 = wall_time();
      if (host_coarsen) {
        strength_to_host(s);
        if (aggressive)
          coarsen_aggressive(p.coarsen_type, n, S, cf);
        else
          coarsen_by_type(p.coarsen_type, n, S, cf);
        dcf.upload(cf);
      } else {
        sk::pmis(dS, 2747, dcf, s);  // (the marks stay on the device: LazyInts)
      }
      t_phase[1] += wall_time() - tp0;
    } else {
      strength(A, p.strong_threshold, p.max_row_sum, S);
      t_phase[0] += wall_time() - tp0;
      tp0 = wall_time();
      if (aggressive)
        coarsen_aggressive(p.coarsen_type, n, S, cf);
      else
        coarsen_by_type(p.coarsen_type, n, S, cf);
      t_phase[1] += wall_time() - tp0;
    }
    long long nc_loc = 0;
    if (on_device && cf.empty() && dcf.p) {
      DVec<long long> crank;
      nc_loc = sk::count_c_points(dcf.p, n, crank, ctx().stream);
    } else {
      std::atomic<long long> acc{0};
      parallel_for(n, [&](int64_t b, int64_t e, int) {
        long long c = 0;
        for (int64_t i = b; i < e; i++) c += (cf[(size_t)i] == C_PT);
        acc += c;
      });
      nc_loc = acc.load();
    }
    long long nc_glob = nc_loc;
    comm.allreduce_host(&nc_glob, 1, CommDType::I64, CommOp::SUM);
    if (nc_glob == 0 || nc_glob == A.global_rows() || nc_glob < p.min_coarse_size) break;

    int nc = 0;
    tp0 = wall_time();
    trace_level.reset(), trace_level.reset(new TraceRange((lname + "interpolation").c_str()));
    bool p_on_device = false;
    if (on_device) {
      hipStream_t s = ctx().stream;
      if (!aggressive && (p.interp_type == 6 || p.interp_type == 0))
        p_on_device = sk::interp(Lv.sA, dS, dcf, p.interp_type, p.trunc_factor, p.pmax_elmts, Lv.sP, nc, s);
      if (p_on_device) {
        // stays on the device; the dimensions are all the host needs
        Lv.P = HostCSR();
        Lv.P.nrows = n;
        Lv.P.ncols = nc;
      } else {
        // multipass or direct interpolation, or a row whose interpolatory set outgrows the kernels' tables:
        // host routine
        if (A.host_diag_stale) {
          Lv.sA.download(A.diag, s);
          A.host_diag_stale = false;
          A.ensure_offd_rows();
        }
        if (!host_coarsen) strength_to_host(s);
        if (cf.empty() && dcf.p) {
          cf.resize((size_t)n);
          d2h(cf.data(), dcf.p, (size_t)n * sizeof(int), nullptr);
        }
        dcf.release();
      }
      dS.release();
    }
    if (!p_on_device) {
      if (aggressive)
        build_multipass(A, S, cf, p.agg_trunc_factor, p.agg_pmax_elmts, Lv.P, nc);
      else if (p.interp_type == 4)  // multipass interpolation on an ordinary splitting (its first pass is all there is
        build_multipass(A, S, cf, p.trunc_factor, p.pmax_elmts, Lv.P, nc);  // unless F points lack a strong C point)
      else
        build_interp(A, S, cf, p.interp_type, p.trunc_factor, p.pmax_elmts, Lv.P, nc);
    }
    if (p_on_device)
      Lv.cf.adopt_device(std::move(dcf), (size_t)n);  // (interpolation has updated the marks there: -3 -> -1)
    else
      Lv.cf = cf;
    Lv.has_cf = true;
    if (!on_device) host_transpose(Lv.P, Lv.R);
    t_phase[2] += wall_time() - tp0;
    tp0 = wall_time();
    trace_level.reset(), trace_level.reset(new TraceRange((lname + "Galerkin product").c_str()));

    // Galerkin product on the (single-rank) operator: A_c = R (A P)
    std::unique_ptr<ParCSR> An(new ParCSR());
    if (on_device) {
      // same arithmetic, same order (setup_kernels.hip); the natural-order device copies stay for the
      // C-first renumbering
      hipStream_t s = ctx().stream;
      if (!p_on_device) Lv.sP.upload(Lv.P, s);
      sk::DCsr dR_local, dAP;
      sk::DCsr &dR = keep_natural_R ? Lv.sR : dR_local;  // the replicated setup slices R on the device later
      sk::transpose(Lv.sP, dR, s);
      sk::spgemm(Lv.sA, Lv.sP, dAP, s);
      L.emplace_back();  // the coarse operator is born on the device (L was reserved: references stay valid)
      sk::DCsr &dAc = L[(size_t)l + 1].sA;
      sk::spgemm(dR, dAP, dAc, s);
      if (p.non_galerkin_tol_for(l) > 0.0) sk::sparsify_non_galerkin(dAc, p.non_galerkin_tol_for(l), s);
      if (nc < device_min_rows) {
        dAc.download(An->diag, s);  // the next level is built by the host routines
      } else {
        An->diag.nrows = An->diag.ncols = nc;
        An->host_diag_stale = true;
        An->dev_diag_nnz = dAc.nnz;
      }
      if (getenv("MI_HYPRE_SETUP_TIMING"))
        printf("   level %d: n %d  device Galerkin %.2f s (nnz A %lld, P %lld, AP %lld, A_c %lld)\n", l, n, wall_time() - tp0,
               (long long)Lv.sA.nnz, (long long)Lv.sP.nnz, (long long)dAP.nnz, (long long)dAc.nnz);
    } else {
      HostCSR AP;
      host_spgemm(A.diag, Lv.P, AP);
      host_spgemm(Lv.R, AP, An->diag);
      if (p.non_galerkin_tol_for(l) > 0.0) sparsify_non_galerkin(An->diag, p.non_galerkin_tol_for(l));
      L.emplace_back();
    }
    An->nrows = nc;
    An->row_start = 0;
    An->row_end = nc;
    An->row_starts = {0, (gidx)nc};
    An->offd.nrows = nc;
    An->offd.ncols = 0;
    if (!An->host_diag_stale) An->offd.ia.assign((size_t)nc + 1, 0);  // (device-resident: ParCSR::ensure_offd_rows)
    An->build_halo_plan(comm);
    t_phase[3] += wall_time() - tp0;
    L[(size_t)l + 1].A_own = std::move(An);
    L[(size_t)l + 1].A = L[(size_t)l + 1].A_own.get();
    trace_level.reset();
    l++;
  }

}

// l1 norms and the coarsest-level dense inverse, on the finished (C-first ordered, per-rank) levels
void BoomerAMG::finish_host() {
  Comm &comm = my_comm();
  // per-level norms (host), on the C-first ordered operators
  const int ch = chunk();
  for (size_t li = 0; li < L.size(); li++) {
    AmgLevel &Lv = L[li];
    ParCSR &A = *Lv.A;
    Lv.n = A.nrows;
    if (A.host_diag_stale && Lv.oA.nrows == A.nrows) {
      // the level lives on the device: norms straight into the solve-phase vectors
      hipStream_t s = ctx().stream;
      DVec<int> dcf_ext;
      Lv.d_diag.alloc((size_t)Lv.n);
      Lv.d_l1gs.alloc((size_t)Lv.n);
      Lv.d_l1jac.alloc((size_t)Lv.n);
      // N > 1 (levels of the distributed setup that were built on the device): the halo block's entries count too
      sk::DCsr dO;
      if (comm.size > 1) {
        std::vector<int> cf_ext;
        if (Lv.has_cf) cf_ext = A.halo_exchange_host_int(comm, Lv.cf.host());  // collective: gated by the global flag
        if (A.offd.nnz() > 0) {
          HostCSR O = A.offd;
          O.nrows = Lv.n;
          O.ncols = (int)A.col_map_offd.size();
          dO.upload(O, s);
          dcf_ext.upload(cf_ext);
        }
      }
      sk::level_norms(Lv.oA, Lv.cf.empty() ? nullptr : Lv.cf.dev(), ch, Lv.d_diag.p, Lv.d_l1gs.p, Lv.d_l1jac.p, s,
                      dO.nnz > 0 ? &dO : nullptr, dO.nnz > 0 ? dcf_ext.p : nullptr);
      MI_HIP(hipStreamSynchronize(s));
      continue;
    }
    std::vector<int> cf_ext;
    if (Lv.has_cf) cf_ext = A.halo_exchange_host_int(comm, Lv.cf.host());  // collective: gated by the global flag
    level_norms(A, Lv.cf.host(), cf_ext, ch, Lv.diag, Lv.l1gs, Lv.l1jac);
  }
  ensure_host((int)L.size() - 1);

  // coarsest level: dense inverse (relax type 9), every rank holds its own rows
  AmgLevel &Lc = L.back();
  const gidx ng = Lc.A->global_rows();
  if (!tail && p.relax_type[2] == 9 && ng <= MAX_DENSE && ng > 0) {
    ParCSR &A = *Lc.A;
    const int n = A.nrows;
    int maxloc = n;
    comm.allreduce_host(&maxloc, 1, CommDType::I32, CommOp::MAX);
    // gather all rows as (global row, global col, value) triples
    std::vector<char> mine;
    auto put = [&](gidx r, gidx c, double v) {
      const size_t off = mine.size();
      mine.resize(off + 2 * sizeof(gidx) + sizeof(double));
      memcpy(mine.data() + off, &r, sizeof(gidx));
      memcpy(mine.data() + off + sizeof(gidx), &c, sizeof(gidx));
      memcpy(mine.data() + off + 2 * sizeof(gidx), &v, sizeof(double));
    };
    for (int i = 0; i < n; i++) {
      for (int64_t k = A.diag.ia[(size_t)i]; k < A.diag.ia[(size_t)i + 1]; k++)
        put(A.row_start + i, A.row_start + A.diag.ja[(size_t)k], A.diag.a[(size_t)k]);
      for (int64_t k = A.offd.ia[(size_t)i]; k < A.offd.ia[(size_t)i + 1]; k++)
        put(A.row_start + i, A.col_map_offd[(size_t)A.offd.ja[(size_t)k]], A.offd.a[(size_t)k]);
    }
    std::vector<double> M((size_t)ng * ng, 0.0);
    auto absorb = [&](const std::vector<char> &buf) {
      const size_t rec = 2 * sizeof(gidx) + sizeof(double);
      for (size_t off = 0; off + rec <= buf.size(); off += rec) {
        gidx r, c;
        double v;
        memcpy(&r, buf.data() + off, sizeof(gidx));
        memcpy(&c, buf.data() + off + sizeof(gidx), sizeof(gidx));
        memcpy(&v, buf.data() + off + 2 * sizeof(gidx), sizeof(double));
        M[(size_t)r * ng + (size_t)c] = v;
      }
    };
    absorb(mine);
    if (comm.size > 1) {
      std::vector<int> peers;
      std::vector<std::vector<char>> send;
      for (int r = 0; r < comm.size; r++)
        if (r != comm.rank) {
          peers.push_back(r);
          send.push_back(mine);
        }
      std::vector<int> from;
      std::vector<std::vector<char>> got;
      comm.exchange_host(peers, send, from, got);
      for (auto &b : got) absorb(b);
    }
    std::vector<double> inv;
    dense_inverse((int)ng, M, inv);
    Lc.slot = maxloc;
    const size_t width = (size_t)comm.size * maxloc;
    std::vector<double> Mp((size_t)n * width, 0.0);
    for (int i = 0; i < n; i++)
      for (int r = 0; r < comm.size; r++) {
        const gidx rs = A.row_starts[(size_t)r], re = A.row_starts[(size_t)r + 1];
        for (gidx g = rs; g < re; g++)
          Mp[(size_t)i * width + (size_t)r * maxloc + (size_t)(g - rs)] = inv[(size_t)(A.row_start + i) * ng + (size_t)g];
      }
    Lc.Cinv_host.swap(Mp);
    Lc.dense = true;
  }
}

namespace {
// wrap a local CSR block as a (possibly rectangular) ParCSR without halo columns
std::unique_ptr<ParCSR> wrap_local(HostCSR &&M, const std::vector<gidx> &row_starts, const std::vector<gidx> &col_starts,
                                   int rank, bool device_resident = false) {
  std::unique_ptr<ParCSR> Q(new ParCSR());
  Q->nrows = M.nrows;
  Q->row_starts = row_starts;
  Q->col_starts = col_starts;
  Q->row_start = row_starts[(size_t)rank];
  Q->row_end = row_starts[(size_t)rank + 1];
  Q->offd.nrows = M.nrows;
  Q->offd.ncols = 0;
  if (!device_resident) Q->offd.ia.assign((size_t)M.nrows + 1, 0);  // (else: ParCSR::ensure_offd_rows)
  Q->diag = std::move(M);
  return Q;
}

// rows (given by their OLD global ids, in NEW order) of the global operator G, columns renumbered through
// colpos (global old -> global new) and split at this rank's column range into diag / halo blocks
std::unique_ptr<ParCSR> slice_rows(const HostCSR &G, const int *rows_old, int nloc, const int *colpos,
                                   const std::vector<gidx> &row_starts, const std::vector<gidx> &col_starts,
                                   int rank) {
  std::unique_ptr<ParCSR> Q(new ParCSR());
  const gidx c0 = col_starts[(size_t)rank], c1 = col_starts[(size_t)rank + 1];
  Q->nrows = nloc;
  Q->row_starts = row_starts;
  Q->col_starts = col_starts;
  Q->row_start = row_starts[(size_t)rank];
  Q->row_end = row_starts[(size_t)rank + 1];
  HostCSR &D = Q->diag, &O = Q->offd;
  D.nrows = O.nrows = nloc;
  D.ncols = (int)(c1 - c0);
  D.ia.assign((size_t)nloc + 1, 0);
  O.ia.assign((size_t)nloc + 1, 0);
  for (int q = 0; q < nloc; q++) {
    const int i = rows_old[q];
    int nd = 0;
    for (int64_t k = G.ia[(size_t)i]; k < G.ia[(size_t)i + 1]; k++) {
      const gidx c = colpos ? colpos[G.ja[(size_t)k]] : G.ja[(size_t)k];
      nd += (c >= c0 && c < c1);
    }
    D.ia[(size_t)q + 1] = D.ia[(size_t)q] + nd;
    O.ia[(size_t)q + 1] = O.ia[(size_t)q] + (G.ia[(size_t)i + 1] - G.ia[(size_t)i] - nd);
  }
  D.ja.resize((size_t)D.nnz());
  D.a.resize((size_t)D.nnz());
  O.ja.resize((size_t)O.nnz());
  O.a.resize((size_t)O.nnz());
  std::vector<gidx> ogid((size_t)O.nnz());
  parallel_for(nloc, [&](int64_t b, int64_t e, int) {
    std::vector<std::pair<gidx, double>> row;
    for (int64_t q = b; q < e; q++) {
      const int i = rows_old[q];
      row.clear();
      for (int64_t k = G.ia[(size_t)i]; k < G.ia[(size_t)i + 1]; k++)
        row.push_back({colpos ? (gidx)colpos[G.ja[(size_t)k]] : (gidx)G.ja[(size_t)k], G.a[(size_t)k]});
      std::sort(row.begin(), row.end(),
                [](const std::pair<gidx, double> &x, const std::pair<gidx, double> &y) { return x.first < y.first; });
      int64_t pd = D.ia[(size_t)q], po = O.ia[(size_t)q];
      for (auto &en : row) {
        if (en.first >= c0 && en.first < c1) {
          D.ja[(size_t)pd] = (int)(en.first - c0);
          D.a[(size_t)pd++] = en.second;
        } else {
          ogid[(size_t)po] = en.first;
          O.a[(size_t)po++] = en.second;
        }
      }
    }
  });
  Q->col_map_offd = ogid;
  std::sort(Q->col_map_offd.begin(), Q->col_map_offd.end());
  Q->col_map_offd.erase(std::unique(Q->col_map_offd.begin(), Q->col_map_offd.end()), Q->col_map_offd.end());
  for (size_t k = 0; k < ogid.size(); k++)
    O.ja[k] = (int)(std::lower_bound(Q->col_map_offd.begin(), Q->col_map_offd.end(), ogid[k]) - Q->col_map_offd.begin());
  O.ncols = (int)Q->col_map_offd.size();
  return Q;
}
}  // namespace

// single rank: the transfer operators have no halo block
void BoomerAMG::make_local_transfer_operators() {
  Comm &comm = my_comm();
  for (size_t l = 0; l + 1 < L.size(); l++) {
    AmgLevel &Lv = L[l];
    if (Lv.P.nrows == 0 && Lv.cf.empty()) continue;
    Lv.Pm = wrap_local(std::move(Lv.P), Lv.A->row_starts, L[l + 1].A->row_starts, comm.rank, Lv.oP.nrows > 0);
    Lv.Rm = wrap_local(std::move(Lv.R), L[l + 1].A->row_starts, Lv.A->row_starts, comm.rank, Lv.oP.nrows > 0);
    if (Lv.oP.nrows > 0) {  // built on the device: host arrays on demand
      Lv.Pm->host_diag_stale = Lv.Rm->host_diag_stale = true;
      Lv.Pm->dev_diag_nnz = Lv.oP.nnz;
      Lv.Rm->dev_diag_nnz = Lv.oR.nnz;
    }
    Lv.Pm->build_halo_plan(comm);
    Lv.Rm->build_halo_plan(comm);
    Lv.P = HostCSR();
    Lv.R = HostCSR();
  }
}

// More than one rank: coarsening, interpolation and the Galerkin products are
// GLOBAL algorithms, so the hierarchy -- and with it the
// iteration count -- does not depend on the number of ranks.  This first
// implementation obtains that by replication: every rank gathers the global
// operator, runs the single-rank setup on it and keeps its own rows of every
// level (in the per-rank C-first ordering) as distributed ParCSR blocks.  The
// solve phase is fully distributed; the setup is O(N_global) per rank and is
// the piece to distribute next.
void BoomerAMG::build_replicated(ParCSR &A0) {
  Comm &comm = my_comm();
  const int rank = comm.rank, size = comm.size;
  // ---- 1. gather the global operator (CSR over global columns)
  double tp0 = wall_time();
  const gidx N = A0.global_rows();
  MI_REQUIRE(N < (gidx)2147483000, "replicated setup: global row count exceeds int32");
  std::vector<char> mine;
  {
    const int n = A0.nrows;
    std::vector<int> len((size_t)n);
    size_t tot = 0;
    for (int i = 0; i < n; i++) {
      len[(size_t)i] = (int)((A0.diag.ia[(size_t)i + 1] - A0.diag.ia[(size_t)i]) +
                             (A0.offd.ia[(size_t)i + 1] - A0.offd.ia[(size_t)i]));
      tot += (size_t)len[(size_t)i];
    }
    // packed as [row lengths | columns | values]: the value section need not be 8-byte aligned, so the arrays are
    // filled on their own and copied in
    std::vector<int> cj(tot);
    std::vector<double> cv(tot);
    size_t q = 0;
    for (int i = 0; i < n; i++) {
      for (int64_t k = A0.diag.ia[(size_t)i]; k < A0.diag.ia[(size_t)i + 1]; k++, q++) {
        cj[q] = (int)(A0.row_start + A0.diag.ja[(size_t)k]);
        cv[q] = A0.diag.a[(size_t)k];
      }
      for (int64_t k = A0.offd.ia[(size_t)i]; k < A0.offd.ia[(size_t)i + 1]; k++, q++) {
        cj[q] = (int)A0.col_map_offd[(size_t)A0.offd.ja[(size_t)k]];
        cv[q] = A0.offd.a[(size_t)k];
      }
    }
    mine.resize(sizeof(int) * (size_t)n + tot * (sizeof(int) + sizeof(double)));
    char *w = mine.data();
    if (n) memcpy(w, len.data(), sizeof(int) * (size_t)n);
    if (tot) {
      memcpy(w + sizeof(int) * (size_t)n, cj.data(), tot * sizeof(int));
      memcpy(w + sizeof(int) * (size_t)n + tot * sizeof(int), cv.data(), tot * sizeof(double));
    }
  }
  std::vector<size_t> offs;
  std::vector<char> everyone;
  comm.allgatherv_host(mine.data(), mine.size(), offs, everyone);
  std::vector<char>().swap(mine);
  std::unique_ptr<ParCSR> Ag(new ParCSR());
  {
    HostCSR &G = Ag->diag;
    G.nrows = G.ncols = (int)N;
    G.ia.assign((size_t)N + 1, 0);
    std::vector<const char *> blob((size_t)size, nullptr);
    for (int r = 0; r < size; r++)
      if (offs[(size_t)r + 1] > offs[(size_t)r]) blob[(size_t)r] = everyone.data() + offs[(size_t)r];
    for (int r = 0; r < size; r++) {
      const gidx rs = A0.row_starts[(size_t)r], re = A0.row_starts[(size_t)r + 1];
      if (re > rs) MI_REQUIRE(blob[(size_t)r] != nullptr, "replicated setup: a rank's rows did not arrive");
      const int *len = reinterpret_cast<const int *>(blob[(size_t)r]);
      for (gidx g = rs; g < re; g++) G.ia[(size_t)g + 1] = G.ia[(size_t)g] + len[(size_t)(g - rs)];
    }
    G.ja.resize((size_t)G.nnz());
    G.a.resize((size_t)G.nnz());
    for (int r = 0; r < size; r++) {
      const gidx rs = A0.row_starts[(size_t)r], re = A0.row_starts[(size_t)r + 1];
      if (re == rs) continue;
      const size_t nr = (size_t)(re - rs);
      const size_t tot = (size_t)(G.ia[(size_t)re] - G.ia[(size_t)rs]);
      const char *base = blob[(size_t)r];
      memcpy(G.ja.data() + G.ia[(size_t)rs], base + sizeof(int) * nr, tot * sizeof(int));
      memcpy(G.a.data() + G.ia[(size_t)rs], base + sizeof(int) * nr + tot * sizeof(int), tot * sizeof(double));
    }
    // rows arrive as [diag | halo]: sort every row by global column
    parallel_for((int64_t)N, [&](int64_t b, int64_t e, int) {
      std::vector<std::pair<int, double>> row;
      for (int64_t i = b; i < e; i++) {
        const int64_t s0 = G.ia[(size_t)i], len = G.ia[(size_t)i + 1] - s0;
        bool sorted = true;
        for (int64_t k = 1; k < len; k++)
          if (G.ja[(size_t)(s0 + k)] < G.ja[(size_t)(s0 + k - 1)]) {
            sorted = false;
            break;
          }
        if (sorted) continue;
        row.resize((size_t)len);
        for (int64_t k = 0; k < len; k++) row[(size_t)k] = {G.ja[(size_t)(s0 + k)], G.a[(size_t)(s0 + k)]};
        std::sort(row.begin(), row.end(),
                  [](const std::pair<int, double> &x, const std::pair<int, double> &y) { return x.first < y.first; });
        for (int64_t k = 0; k < len; k++) {
          G.ja[(size_t)(s0 + k)] = row[(size_t)k].first;
          G.a[(size_t)(s0 + k)] = row[(size_t)k].second;
        }
      }
    });
    Ag->nrows = (int)N;
    Ag->row_start = 0;
    Ag->row_end = N;
    Ag->row_starts = {0, N};
    Ag->offd.nrows = (int)N;
    Ag->offd.ncols = 0;
    Ag->offd.ia.assign((size_t)N + 1, 0);
  }
  std::vector<char>().swap(everyone);
  const double t_gather = wall_time() - tp0;

  // ---- 2. the global hierarchy, natural ordering, through the single-rank code path
  BoomerAMG g;
  g.p = p;
  g.p.print_level = 0;
  g.device_min_rows = device_min_rows;
  g.keep_natural_R = true;
  g.stop_rows = effective_redundant_rows();
  g.use_private_self_comm();
  g.build_natural(*Ag);
  // a level below the threshold is redundant also when the coarsening ended on it (it is then the coarsest
  // level, and its smoother -- if it is not the dense solve -- must not stop at rank boundaries either)
  if (!g.stopped_by_rows && g.stop_rows > 0 && g.L.size() >= 2 && g.L.back().A->global_rows() <= g.stop_rows)
    g.stopped_by_rows = true;
  const bool has_tail = g.stopped_by_rows && g.L.size() >= 2;
  if (g.L[0].sA.ia.p && g.L[0].sA.nrows == (int)N) {  // level 0 lives on the device: drop the host copy
    HostCSR &G0 = Ag->diag;
    std::vector<int64_t>().swap(G0.ia);
    std::vector<int>().swap(G0.ja);
    std::vector<double>().swap(G0.a);
    Ag->host_diag_stale = true;
  }
  for (int q = 0; q < 4; q++) t_phase[q] = g.t_phase[q];
  tp0 = wall_time();
  const size_t nlev = g.L.size();

  // ---- 3. row partition of every level (coarse ownership = owner of the C point) and
  //         the per-rank C-first ordering: pos (old -> new), perm (new -> old), both global
  std::vector<std::vector<gidx>> starts(nlev);
  starts[0] = A0.row_starts;
  std::vector<std::vector<int>> pos(nlev), perm(nlev);
  for (size_t l = 0; l < nlev; l++) {
    const AmgLevel &G = g.L[l];
    const int n = G.A->nrows;
    if (l + 1 < nlev) {
      starts[l + 1].assign((size_t)size + 1, 0);
      for (int r = 0; r < size; r++) {
        gidx c = 0;
        for (gidx i = starts[l][(size_t)r]; i < starts[l][(size_t)r + 1]; i++) c += (G.cf[(size_t)i] == C_PT);
        starts[l + 1][(size_t)r + 1] = starts[l + 1][(size_t)r] + c;
      }
    }
    pos[l].resize((size_t)n);
    perm[l].resize((size_t)n);
    if (G.cf.empty()) {
      for (int i = 0; i < n; i++) pos[l][(size_t)i] = perm[l][(size_t)i] = i;
    } else {
      for (int r = 0; r < size; r++) {
        gidx q = starts[l][(size_t)r];
        for (gidx i = starts[l][(size_t)r]; i < starts[l][(size_t)r + 1]; i++)
          if (G.cf[(size_t)i] == C_PT) pos[l][(size_t)i] = (int)q++;
        for (gidx i = starts[l][(size_t)r]; i < starts[l][(size_t)r + 1]; i++)
          if (G.cf[(size_t)i] != C_PT) pos[l][(size_t)i] = (int)q++;
      }
      for (int i = 0; i < n; i++) perm[l][(size_t)pos[l][(size_t)i]] = i;
    }
  }

  // ---- 4. this rank's slices
  L.clear();
  L.resize(nlev);
  for (size_t l = 0; l < nlev; l++) {
    AmgLevel &G = g.L[l];
    AmgLevel &Lv = L[l];
    const gidx r0 = starts[l][(size_t)rank], r1 = starts[l][(size_t)rank + 1];
    const int nloc = (int)(r1 - r0);
    const int *rows_old = perm[l].data() + r0;
    hipStream_t s = ctx().stream;
    // a level the device built is sliced there (this rank's rows, columns renumbered, re-sorted) and only the
    // slice comes to the host; host-built levels are sliced from their host arrays
    auto slice = [&](sk::DCsr &dev, const HostCSR &host, bool host_ok, const int *rows, int nrows_loc,
                     const std::vector<int> &colpos, const std::vector<gidx> &rstarts,
                     const std::vector<gidx> &cstarts) -> std::unique_ptr<ParCSR> {
      if (dev.ia.p && dev.nnz > 0) {
        DVec<int> d_rows, d_pos;
        d_rows.upload(std::vector<int>(rows, rows + nrows_loc));
        d_pos.upload(colpos);
        sk::DCsr mine;
        sk::extract_rows(dev, d_rows.p, nrows_loc, d_pos.p, mine, s);
        HostCSR local;
        mine.download(local, s);
        std::vector<int> iota((size_t)nrows_loc);
        for (int q = 0; q < nrows_loc; q++) iota[(size_t)q] = q;
        return slice_rows(local, iota.data(), nrows_loc, nullptr, rstarts, cstarts, rank);
      }
      MI_REQUIRE(host_ok, "replicated setup: a level has neither device nor host arrays");
      return slice_rows(host, rows, nrows_loc, colpos.data(), rstarts, cstarts, rank);
    };
    Lv.A_own = slice(G.sA, G.A->diag, !G.A->host_diag_stale, rows_old, nloc, pos[l], starts[l], starts[l]);
    Lv.A = Lv.A_own.get();
    Lv.A->build_halo_plan(comm);
    Lv.has_cf = !G.cf.empty();
    if (Lv.has_cf) {
      const std::vector<int> &gcf = G.cf.host();
      std::vector<int> &lcf = Lv.cf.hostw(), &lperm = Lv.perm.hostw();
      lcf.resize((size_t)nloc);
      lperm.resize((size_t)nloc);
      Lv.nc = 0;
      for (int q = 0; q < nloc; q++) {
        lcf[(size_t)q] = gcf[(size_t)rows_old[q]];
        lperm[(size_t)q] = (int)(rows_old[q] - r0);
        Lv.nc += (lcf[(size_t)q] == C_PT);
      }
      // P: my fine rows x coarse columns; R = P^T: my coarse rows x fine columns
      Lv.Pm = slice(G.sP, G.P, !G.P.ia.empty(), rows_old, nloc, pos[l + 1], starts[l], starts[l + 1]);
      // with a redundant tail the coarse correction is whole on every rank: no exchange for the last P
      if (!(has_tail && l + 2 == nlev)) Lv.Pm->build_halo_plan(comm);
      const gidx c0 = starts[l + 1][(size_t)rank], c1 = starts[l + 1][(size_t)rank + 1];
      Lv.Rm = slice(G.sR, G.R, !G.R.ia.empty(), perm[l + 1].data() + c0, (int)(c1 - c0), pos[l], starts[l + 1], starts[l]);
      Lv.Rm->build_halo_plan(comm);
    }
    G.sP.release();
    G.sR.release();
    if (!(has_tail && l + 1 == nlev)) G.sA.release();
    // the global level is not needed any more (the tail's fine level is)
    if (has_tail && l + 1 == nlev && G.A_own)
      tail_A = std::move(G.A_own);
    else if (G.A_own)
      G.A_own.reset();
    G.P = HostCSR();
    G.R = HostCSR();
  }
  tail.reset();
  if (has_tail) {
    MI_REQUIRE(tail_A != nullptr, "replicated setup: the redundant level was not kept");
    // levels nlev-1 .. : one single-rank hierarchy per rank, built by the ordinary pipeline on the
    // (global, natural-order) operator of the first redundant level
    if (tail_A->host_diag_stale) {
      const int nr = tail_A->diag.nrows, ncl = tail_A->diag.ncols;
      g.L.back().sA.download(tail_A->diag, ctx().stream);
      tail_A->diag.nrows = nr;
      tail_A->diag.ncols = ncl;
      tail_A->host_diag_stale = false;
      tail_A->ensure_offd_rows();
    }
    g.L.back().sA.release();
    tail.reset(new BoomerAMG());
    tail->p = p;
    tail->p.print_level = 0;
    tail->p.max_levels = std::max(1, p.max_levels - (int)(nlev - 1));
    tail->p.smooth_num_levels = std::max(0, p.smooth_num_levels - (int)(nlev - 1));
    tail->p.agg_num_levels = std::max(0, p.agg_num_levels - (int)(nlev - 1));
    {  // the tail counts its levels from 0: level-specific non-Galerkin tolerances move with it
      std::vector<double> shifted;
      for (size_t q = nlev - 1; q < p.non_galerkin_level_tol.size(); q++) shifted.push_back(p.non_galerkin_level_tol[q]);
      tail->p.non_galerkin_level_tol = shifted;
    }
    tail->device_min_rows = device_min_rows;
    tail->use_private_self_comm();
    tail->setup_host(*tail_A);
    tail_start = starts[nlev - 1][(size_t)rank];
    int slot = 0;
    for (int r = 0; r < size; r++)
      slot = std::max(slot, (int)(starts[nlev - 1][(size_t)r + 1] - starts[nlev - 1][(size_t)r]));
    tail_slot = slot;
    std::vector<int> map((size_t)tail_A->nrows);
    for (int r = 0; r < size; r++)
      for (gidx i = starts[nlev - 1][(size_t)r]; i < starts[nlev - 1][(size_t)r + 1]; i++)
        map[(size_t)i] = r * slot + (int)(i - starts[nlev - 1][(size_t)r]);
    tail_map_host.swap(map);
  } else {
    tail_A.reset();
  }
  Ag.reset();
  t_phase[4] = wall_time() - tp0;
  if (p.print_level > 0 && rank == 0)
    printf("mi_hypre BoomerAMG: replicated setup on %d ranks (global gather %.2f s, slicing %.2f s)\n", size, t_gather,
           t_phase[4]);
}

void BoomerAMG::setup_device() {
  TraceRange trace_setup("mi_hypre BoomerAMGSetup (solve-phase format)");
  MI_REQUIRE(host_ready, "BoomerAMG: setup_device before setup_host");
  ensure_init();
  Comm &comm = my_comm();
  const int ch = chunk();
  const bool timing = getenv("MI_HYPRE_SETUP_TIMING") != nullptr && comm.rank == 0;
  for (size_t li = 0; li < L.size(); li++) {
    AmgLevel &Lv = L[li];
    hipStream_t s = ctx().stream;
    const double tdev0 = wall_time();
    struct LevelTimer {
      bool on;
      size_t l;
      double t0;
      ~LevelTimer() {
        if (on) {
          (void)hipDeviceSynchronize();
          printf("   solve-phase format: level %zu %.2f s\n", l, wall_time() - t0);
        }
      }
    } level_timer{timing && li < 4, li, tdev0};
    const std::string fname = "format: level " + std::to_string(li) + " ";
    std::unique_ptr<TraceRange> trace_f(new TraceRange((fname + "A").c_str()));
    auto place = [&](ParCSR &M, sk::DCsr &dev) {
      if (dev.nrows == M.nrows && M.host_diag_stale) {  // built on the device: no host round trip
        sk::to_solve_format(dev, M.d_diag, s);
        M.to_device_halo();
      } else {
        M.to_device();
      }
    };
    if (li > 0 || !Lv.A->on_device) place(*Lv.A, Lv.oA);
    // transfer operators are SpMV-only and short-rowed: more rows per tile (MI_HYPRE_SPMV_ROW_CAP, in rows)
    static const int spmv_cap = getenv("MI_HYPRE_SPMV_ROW_CAP") ? atoi(getenv("MI_HYPRE_SPMV_ROW_CAP")) : k::SPMV_ONLY_ROW_CAP;
    if (Lv.Pm) Lv.Pm->d_diag.row_cap = spmv_cap;
    if (Lv.Rm) Lv.Rm->d_diag.row_cap = spmv_cap;
    trace_f.reset(), trace_f.reset(new TraceRange((fname + "P").c_str()));
    if (Lv.Pm) place(*Lv.Pm, Lv.oP);
    trace_f.reset(), trace_f.reset(new TraceRange((fname + "R").c_str()));
    // Restriction: the coarse vectors live in the coarse level's C-first order, in which neighbouring rows of R are
    // NOT neighbouring coarse points of the fine level (first the coarse level's C points, then its F points): a
    // tile of R then gathers a thinned-out stretch of the fine vector, and every line of it is fetched by several
    // tiles (4x the vector's bytes at 512^3).  R is therefore stored with its rows in the order the coarse points
    // have on the FINE level (= the coarse level's ordering before its own C-first step) and writes through a row map.
    static const bool natural_r = !(getenv("MI_HYPRE_NATURAL_R") && atoi(getenv("MI_HYPRE_NATURAL_R")) == 0);
    const bool r_on_device = Lv.Rm && Lv.oR.nrows == Lv.Rm->nrows && Lv.Rm->host_diag_stale;  // built there
    static const long long nat_min = getenv("MI_HYPRE_NATURAL_R_MIN_NNZ") ? atoll(getenv("MI_HYPRE_NATURAL_R_MIN_NNZ")) : 4000000;
    const bool r_large_host = Lv.Rm && !Lv.Rm->host_diag_stale && Lv.Rm->diag.nnz() >= nat_min;  // e.g. distributed setup
    if (Lv.Rm && natural_r && li + 1 < L.size() && (r_on_device || r_large_host) &&
        L[li + 1].perm.size() == (size_t)Lv.Rm->nrows && Lv.Rm->nrows > 0) {
      // L[li + 1].perm: stored position -> position before the C-first step; its inverse orders the rows of R
      DVec<int> dpos(L[li + 1].perm.size());
      sk::invert_permutation(L[li + 1].perm.dev(), (int)L[li + 1].perm.size(), dpos.p, s);
      sk::DCsr nat;
      if (r_on_device) {
        sk::permute(Lv.oR, dpos.p, nullptr, nat, s);
        Lv.oR.release();
      } else {
        sk::DCsr raw;
        raw.upload(Lv.Rm->diag, s);
        sk::permute(raw, dpos.p, nullptr, nat, s);
      }
      sk::to_solve_format(nat, Lv.Rm->d_diag, s);
      Lv.Rm->d_diag.rowmap = std::move(dpos);
      Lv.Rm->to_device_halo();
    } else if (Lv.Rm) {
      Lv.Rm->d_diag.rowmap.release();
      place(*Lv.Rm, Lv.oR);
    }
    // the down leg starts every level from u = 0: its sweep runs on the entries that can see non-zeros
    trace_f.reset(), trace_f.reset(new TraceRange((fname + "zero-guess sub-operators").c_str()));
    Lv.has_Az = false;
    Lv.Az = DevCSR();
    const int t0 = p.relax_type[0];
    const bool down_gs = t0 == 3 || t0 == 4 || t0 == 6 || t0 == 8 || t0 == 13 || t0 == 14;  // hybrid GS family
    if (zero_skip_mode() > 1 && down_gs && Lv.has_cf && !Lv.cf.empty() && Lv.n > 0 && Lv.nc > 0 &&
        Lv.A->d_diag.nrows == Lv.n && Lv.A->d_diag.ncols == Lv.n) {
      sk::DCsr Z;
      sk::zero_guess_operator(Lv.A->d_diag, Lv.nc, ch, Z, s);
      sk::to_solve_format(Z, Lv.Az, s);
      Lv.has_Az = true;
      Lv.Az_chunk = ch;
      Lv.has_Ar = false;
      Lv.Ar = DevCSR();
      Lv.t_valid = false;
      if (zero_skip_mode() > 2 && ch == 8 && li + 1 < L.size()) {  // a residual follows the down sweep
        sk::DCsr R;
        sk::zero_guess_operator(Lv.A->d_diag, Lv.nc, ch, R, s, 1);
        Lv.Ar.row_cap = spmv_cap;
        sk::to_solve_format(R, Lv.Ar, s);
        Lv.t_from = (Lv.nc + ch - 1) / ch * ch;
        Lv.tvec.alloc((size_t)Lv.n);
        zero_on_stream(Lv.tvec.p, (size_t)Lv.n * sizeof(double));
        Lv.has_Ar = true;
        Lv.Az.prefer_gs_tiles = true;  // only the tile kernel hands the F pass's C-column product over
      }
    }
    trace_f.reset(), trace_f.reset(new TraceRange((fname + "vectors, C/F marks, permutation").c_str()));
    if (!Lv.d_diag.p || Lv.d_diag.n != (size_t)Lv.n) {
      Lv.d_diag.upload(Lv.diag);
      Lv.d_l1gs.upload(Lv.l1gs);
      Lv.d_l1jac.upload(Lv.l1jac);
    }
    if (!Lv.cf.empty()) {
      Lv.d_cf.alloc(Lv.cf.size());
      sk::ints_to_i8(Lv.cf.dev(), (long long)Lv.cf.size(), Lv.d_cf.p, s);
    }
    if (!Lv.perm.empty()) {
      Lv.d_perm.alloc(Lv.perm.size());
      MI_HIP(hipMemcpyAsync(Lv.d_perm.p, Lv.perm.dev(), Lv.perm.size() * sizeof(int), hipMemcpyDeviceToDevice, s));
    }
    Lv.u.alloc((size_t)Lv.n);
    Lv.f.alloc((size_t)Lv.n);
    Lv.tmp.alloc((size_t)Lv.n);
    Lv.snap.alloc((size_t)Lv.n);
    if (Lv.n) {
      zero_on_stream(Lv.u.p, (size_t)Lv.n * sizeof(double));
      zero_on_stream(Lv.f.p, (size_t)Lv.n * sizeof(double));
      zero_on_stream(Lv.tmp.p, (size_t)Lv.n * sizeof(double));
      zero_on_stream(Lv.snap.p, (size_t)Lv.n * sizeof(double));
    }
  }
  // complex smoother (smooth_type 5 = ILU) on levels < smooth_num_levels, never on the last level (which is the
  // coarse solve, or the hand-over to the redundant tail): one block-Jacobi ILU(0) of the level's diag block each
  for (size_t li = 0; li < L.size(); li++) {
    AmgLevel &Lv = L[li];
    Lv.smoother.reset();
    if (p.smooth_type != 5 || (int)li >= p.smooth_num_levels || li + 1 >= L.size()) continue;
    ensure_host((int)li);
    auto ilu = std::make_shared<IluSolver>();
    ilu->ilu_type = p.ilu_type;
    ilu->level_of_fill = p.ilu_level;
    ilu->tri_solve = p.ilu_tri_solve;
    ilu->lower_it = p.ilu_lower_it;
    ilu->upper_it = p.ilu_upper_it;
    ilu->max_iter = p.ilu_max_iter;
    ilu->tol = 0.0;
    ilu->print_level = (p.print_level > 0 && comm.rank == 0) ? 1 : 0;
    ilu->setup(*Lv.A);
    Lv.smoother = std::move(ilu);
  }
  AmgLevel &Lc = L.back();
  if (Lc.dense) {
    const size_t width = (size_t)comm.size * (size_t)Lc.slot;
    Lc.Cinv.upload(Lc.Cinv_host);
    Lc.fgather.alloc(width);
    Lc.fslot.alloc((size_t)Lc.slot);
    zero_on_stream(Lc.fslot.p, ((size_t)Lc.slot + 2) * sizeof(double));
    zero_on_stream(Lc.fgather.p, (width + 2) * sizeof(double));
  }
  if (tail) {
    tail->setup_device();
    const size_t ng = (size_t)tail_A->nrows;
    tail_fslot.alloc((size_t)tail_slot);
    tail_fgather.alloc((size_t)tail_slot * (size_t)comm.size);
    tail_f.alloc(ng);
    tail_e.alloc(ng);
    zero_on_stream(tail_fslot.p, ((size_t)tail_slot + 2) * sizeof(double));
    zero_on_stream(tail_fgather.p, ((size_t)tail_slot * (size_t)comm.size + 2) * sizeof(double));
    zero_on_stream(tail_e.p, (ng + 2) * sizeof(double));
    tail_map.upload(tail_map_host);
    AmgLevel &Lp = L[L.size() - 2];
    std::vector<int> pcol(Lp.Pm->col_map_offd.size());
    for (size_t q = 0; q < pcol.size(); q++) pcol[q] = (int)Lp.Pm->col_map_offd[q];
    tail_pcol.upload(pcol);
  }
  MI_HIP(hipDeviceSynchronize());
  sk::release_host_scratch();
  dev_pool_trim();  // the setup's transient buffers go back to the driver: the solve allocates its Krylov basis next
  {
    TraceRange tr("format: collapsed coarse tail");
    const double tc0 = wall_time();
    build_collapsed_tail();
    if (timing && collapsed_level >= 0)
      printf("   collapsed coarse tail: levels %d.. as one %d x %d map%s, %.3f s\n", collapsed_level, collapsed_n, collapsed_n,
             collapsed_level2 >= 0 ? (", and levels " + std::to_string(collapsed_level2) + ".. as one " +
                                      std::to_string(collapsed_n2) + " x " + std::to_string(collapsed_n2) + " map").c_str() : "",
             wall_time() - tc0);
  }
  is_setup = true;
  setup_seconds = wall_time() - t_setup_start;
  if (timing) {
    long long mp = 0, iu = 0, pm = 0, pu = 0, gr = 0, dr = 0;
    double tg = 0.0, td = 0.0, tw = 0.0;
    dev_arena_stats(&mp, &iu, &pm, &pu);
    dev_arena_times(&tg, &td, &gr, &dr, &tw);
    printf("   device arena (since the process started): %.1f GiB mapped (peak %.1f), %.1f GiB in use (peak %.1f); %lld chunks mapped, %.2f s inside "
           "hipMemCreate/Map (grow-ahead thread), requests waited %.2f s for it; %lld stream drains before a reuse, %.2f s\n",
           mp / 1073741824.0, pm / 1073741824.0, iu / 1073741824.0, pu / 1073741824.0, gr, tg, tw, dr, td);
    printf("   block-coded column lists (tiles that keep a 4-byte list / tiles; most 1024-id blocks in a tile):");
    for (size_t li = 0; li < L.size() && li < 6; li++) {
      const AmgLevel &Lv = L[li];
      if (!Lv.A->d_diag.xcache) continue;
      auto show = [](const char *nm, const DevCSR &M) {
        printf(" %s %d/%d (%d)%s", nm, M.ucode_wide_tiles, M.nblocks, M.ucode_max_blocks, M.ucode.p ? "" : " off");
      };
      printf(" L%zu[", li);
      show("A", Lv.A->d_diag);
      if (Lv.has_Az) show("Az", Lv.Az);
      if (Lv.has_Ar) show("Ar", Lv.Ar);
      if (Lv.Pm && Lv.Pm->d_diag.xcache) show("P", Lv.Pm->d_diag);
      if (Lv.Rm && Lv.Rm->d_diag.xcache) show("R", Lv.Rm->d_diag);
      printf("]");
    }
    printf("\n");
    printf("   value dictionaries (1-byte value stream):");
    for (size_t li = 0; li < L.size(); li++) {
      const AmgLevel &Lv = L[li];
      printf(" L%zu[%s%s%s%s%s]", li, Lv.A->d_diag.val8 ? "A" : "", Lv.has_Az && Lv.Az.val8 ? " Az" : "",
             Lv.has_Ar && Lv.Ar.val8 ? " Ar" : "", Lv.Pm && Lv.Pm->d_diag.val8 ? " P" : "", Lv.Rm && Lv.Rm->d_diag.val8 ? " R" : "");
    }
    printf("\n");
  }
  if (p.print_level > 0 && comm.rank == 0) {
    printf("mi_hypre BoomerAMG setup: %d levels, operator complexity %.3f, chunk %d, %.3f s\n", total_levels(),
           operator_complexity(), ch, setup_seconds);
    printf("   hierarchy build, wall per phase: strength %.2f  pmis %.2f  interp %.2f  galerkin %.2f  C-first ordering %.2f  (total %.2f) s\n",
           t_phase[0], t_phase[1], t_phase[2], t_phase[3], t_phase[4], t_phase[5]);
    for (int li = 0; li < total_levels(); li++) {
      int loc = 0;
      BoomerAMG &o = owner_of(li, loc);
      const AmgLevel &Lv = o.L[(size_t)loc];
      printf("   level %2d: local rows %10d  global rows %12lld  local nnz %12lld%s\n", li, Lv.n,
             (long long)Lv.A->global_rows(), (long long)(Lv.A->diag_nnz() + Lv.A->offd.nnz()),
             &o != this ? "  (whole level on every rank)" : "");
    }
  }
}

}  // namespace mi
