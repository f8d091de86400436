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
#include <cmath>
#include <cstring>

#include "amg.hpp"
#include "kernels.hpp"

namespace mi {

namespace {

constexpr int C_PT = 1, F_PT = -1, SF_PT = -3;
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

struct Strength {
  std::vector<int64_t> ia;
  std::vector<int> ja;
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
                  int pmax, HostCSR &P, int &nc_out) {
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
      if (cf[(size_t)i] == C_PT) {
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

}  // namespace

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

int BoomerAMG::chunk() const { return p.gs_chunk > 0 ? p.gs_chunk : ctx().gs_chunk; }

double BoomerAMG::operator_complexity() const {
  if (L.empty()) return 0.0;
  double tot = 0.0;
  for (const auto &l : L) tot += (double)(l.A->diag.nnz() + l.A->offd.nnz());
  const double base = (double)(L[0].A->diag.nnz() + L[0].A->offd.nnz());
  return base > 0 ? tot / base : 0.0;
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
  Comm &comm = current_comm();
  const size_t nlev = L.size();
  std::vector<std::vector<int>> pos(nlev);  // old -> new local row
  for (size_t l = 0; l < nlev; l++) {
    AmgLevel &Lv = L[l];
    if (Lv.cf.empty()) continue;
    const int n = Lv.A->nrows;
    pos[l].resize((size_t)n);
    Lv.perm.resize((size_t)n);
    int q = 0;
    for (int i = 0; i < n; i++)
      if (Lv.cf[(size_t)i] == C_PT) pos[l][(size_t)i] = q++;
    Lv.nc = q;
    for (int i = 0; i < n; i++)
      if (Lv.cf[(size_t)i] != C_PT) pos[l][(size_t)i] = q++;
    for (int i = 0; i < n; i++) Lv.perm[(size_t)pos[l][(size_t)i]] = i;
  }
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
    if (!pos[l].empty()) {
      ParCSR &A = *Lv.A;
      std::unique_ptr<ParCSR> An(new ParCSR());
      An->nrows = A.nrows;
      An->row_start = A.row_start;
      An->row_end = A.row_end;
      An->row_starts = A.row_starts;
      permute(A.diag, Lv.perm, pos[l].data(), An->diag);
      // halo columns: new global id = owner's start + owner's new position
      std::vector<int> ext_pos = A.halo_exchange_host_int(comm, pos[l]);
      const size_t next = A.col_map_offd.size();
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
      permute(A.offd, Lv.perm, next ? colpos.data() : nullptr, An->offd);
      An->offd.ncols = (int)next;
      An->col_map_offd = cm;
      An->build_halo_plan(comm);
      Lv.A_own = std::move(An);
      Lv.A = Lv.A_own.get();
      std::vector<int> cf2(Lv.cf.size());
      for (size_t q = 0; q < cf2.size(); q++) cf2[q] = Lv.cf[(size_t)Lv.perm[q]];
      Lv.cf.swap(cf2);
    }
    if (Lv.P.nrows > 0 && (!pos[l].empty() || (l + 1 < nlev && !pos[l + 1].empty()))) {
      HostCSR P2;
      permute(Lv.P, Lv.perm, (l + 1 < nlev && !pos[l + 1].empty()) ? pos[l + 1].data() : nullptr, P2);
      Lv.P = std::move(P2);
      host_transpose(Lv.P, Lv.R);
    }
  }
}

void BoomerAMG::setup_host(ParCSR &A0) {
  Comm &comm = current_comm();
  t_setup_start = wall_time();
  for (double &t : t_phase) t = 0.0;
  is_setup = false;
  L.clear();
  L.reserve((size_t)std::max(1, p.max_levels));
  L.emplace_back();
  L[0].A = &A0;
  MI_REQUIRE(!A0.row_starts.empty(), "BoomerAMGSetup: matrix is not assembled");
  if (p.print_level > 0 && comm.rank == 0 && p.coarsen_type != 8 && p.coarsen_type != 10)
    printf("mi_hypre BoomerAMG: coarsen_type %d is not restated; using PMIS (8)\n", p.coarsen_type);

  int l = 0;
  while (l < p.max_levels - 1 && L[(size_t)l].A->global_rows() > p.max_coarse_size) {
    ParCSR &A = *L[(size_t)l].A;
    const int n = A.nrows;
    Strength S;
    double tp0 = wall_time();
    strength(A, p.strong_threshold, p.max_row_sum, S);
    t_phase[0] += wall_time() - tp0;
    tp0 = wall_time();
    std::vector<int> cf;
    pmis(n, S, comm.rank, cf);
    t_phase[1] += wall_time() - tp0;
    long long nc_loc = 0;
    for (int i = 0; i < n; i++) nc_loc += (cf[(size_t)i] == C_PT);
    long long nc_glob = nc_loc;
    comm.allreduce_host(&nc_glob, 1, CommDType::I64, CommOp::SUM);
    if (nc_glob == 0 || nc_glob == A.global_rows() || nc_glob < p.min_coarse_size) break;

    AmgLevel &Lv = L[(size_t)l];
    int nc = 0;
    tp0 = wall_time();
    build_interp(A, S, cf, p.interp_type, p.trunc_factor, p.pmax_elmts, Lv.P, nc);
    Lv.cf = cf;
    host_transpose(Lv.P, Lv.R);
    t_phase[2] += wall_time() - tp0;
    tp0 = wall_time();

    // coarse partition
    std::vector<gidx> cstarts((size_t)comm.size + 1, 0);
    {
      std::vector<long long> all((size_t)comm.size);
      comm.allgather_host(&nc_loc, all.data(), sizeof(long long));
      for (int r = 0; r < comm.size; r++) cstarts[(size_t)r + 1] = cstarts[(size_t)r] + all[(size_t)r];
    }
    const gidx cstart = cstarts[(size_t)comm.rank];

    // P rows of the halo columns, in coarse GLOBAL ids
    const int next = (int)A.col_map_offd.size();
    HostCSR B;  // [P ; P_ext] over columns [0,nc) local + nc.. extended
    std::vector<gidx> ext_cgid;
    if (comm.size > 1) {
      std::vector<std::vector<char>> send(A.halo.send_peers.size());
      for (size_t i = 0; i < A.halo.send_peers.size(); i++) {
        std::vector<char> &buf = send[i];
        for (int k = A.halo.send_starts[i]; k < A.halo.send_starts[i + 1]; k++) {
          const int r = A.halo.send_map[(size_t)k];
          const int len = (int)(Lv.P.ia[(size_t)r + 1] - Lv.P.ia[(size_t)r]);
          const size_t off = buf.size();
          buf.resize(off + sizeof(int) + (size_t)len * (sizeof(gidx) + sizeof(double)));
          char *w = buf.data() + off;
          memcpy(w, &len, sizeof(int));
          w += sizeof(int);
          for (int q = 0; q < len; q++) {
            const gidx g = cstart + Lv.P.ja[(size_t)(Lv.P.ia[(size_t)r] + q)];
            memcpy(w, &g, sizeof(gidx));
            w += sizeof(gidx);
            memcpy(w, &Lv.P.a[(size_t)(Lv.P.ia[(size_t)r] + q)], sizeof(double));
            w += sizeof(double);
          }
        }
      }
      std::vector<int> from;
      std::vector<std::vector<char>> got;
      comm.exchange_host(A.halo.send_peers, send, from, got);
      std::vector<std::vector<std::pair<gidx, double>>> ext_rows((size_t)next);
      for (size_t i = 0; i < from.size(); i++) {
        size_t pi = 0;
        while (pi < A.halo.recv_peers.size() && A.halo.recv_peers[pi] != from[i]) pi++;
        MI_REQUIRE(pi < A.halo.recv_peers.size(), "unexpected P-row sender");
        const char *rp = got[i].data();
        for (int k = A.halo.recv_starts[pi]; k < A.halo.recv_starts[pi + 1]; k++) {
          int len;
          memcpy(&len, rp, sizeof(int));
          rp += sizeof(int);
          for (int q = 0; q < len; q++) {
            gidx g;
            double v;
            memcpy(&g, rp, sizeof(gidx));
            rp += sizeof(gidx);
            memcpy(&v, rp, sizeof(double));
            rp += sizeof(double);
            ext_rows[(size_t)k].push_back({g, v});
            ext_cgid.push_back(g);
          }
        }
      }
      std::sort(ext_cgid.begin(), ext_cgid.end());
      ext_cgid.erase(std::unique(ext_cgid.begin(), ext_cgid.end()), ext_cgid.end());
      B.nrows = n + next;
      B.ncols = nc + (int)ext_cgid.size();
      B.ia.assign((size_t)B.nrows + 1, 0);
      for (int i = 0; i < n; i++) B.ia[(size_t)i + 1] = Lv.P.ia[(size_t)i + 1];
      for (int k = 0; k < next; k++) B.ia[(size_t)(n + k) + 1] = B.ia[(size_t)(n + k)] + (int64_t)ext_rows[(size_t)k].size();
      B.ja = Lv.P.ja;
      B.a = Lv.P.a;
      B.ja.resize((size_t)B.nnz());
      B.a.resize((size_t)B.nnz());
      for (int k = 0; k < next; k++) {
        int64_t q = B.ia[(size_t)(n + k)];
        for (auto &pr : ext_rows[(size_t)k]) {
          B.ja[(size_t)q] =
              nc + (int)(std::lower_bound(ext_cgid.begin(), ext_cgid.end(), pr.first) - ext_cgid.begin());
          B.a[(size_t)q] = pr.second;
          q++;
        }
      }
    }

    HostCSR Ac;
    {
      HostCSR AP;
      if (comm.size > 1 && next > 0) {
        // Afull = [A_diag | A_offd] over the extended column space
        HostCSR Af;
        Af.nrows = n;
        Af.ncols = n + next;
        Af.ia.assign((size_t)n + 1, 0);
        for (int i = 0; i < n; i++)
          Af.ia[(size_t)i + 1] = Af.ia[(size_t)i] + (A.diag.ia[(size_t)i + 1] - A.diag.ia[(size_t)i]) +
                                 (A.offd.ia[(size_t)i + 1] - A.offd.ia[(size_t)i]);
        Af.ja.resize((size_t)Af.nnz());
        Af.a.resize((size_t)Af.nnz());
        parallel_for(n, [&](int64_t b, int64_t e, int) {
          for (int64_t i = b; i < e; i++) {
            int64_t q = Af.ia[(size_t)i];
            for (int64_t k = A.diag.ia[(size_t)i]; k < A.diag.ia[(size_t)i + 1]; k++, q++) {
              Af.ja[(size_t)q] = A.diag.ja[(size_t)k];
              Af.a[(size_t)q] = A.diag.a[(size_t)k];
            }
            for (int64_t k = A.offd.ia[(size_t)i]; k < A.offd.ia[(size_t)i + 1]; k++, q++) {
              Af.ja[(size_t)q] = n + A.offd.ja[(size_t)k];
              Af.a[(size_t)q] = A.offd.a[(size_t)k];
            }
          }
        });
        host_spgemm(Af, B, AP);
      } else {
        host_spgemm(A.diag, Lv.P, AP);
      }
      host_spgemm(Lv.R, AP, Ac);
    }

    // next level ParCSR: split columns at nc
    std::unique_ptr<ParCSR> An(new ParCSR());
    An->nrows = nc;
    An->row_start = cstart;
    An->row_end = cstart + nc;
    An->row_starts = cstarts;
    HostCSR &Dn = An->diag, &On = An->offd;
    Dn.nrows = On.nrows = nc;
    Dn.ncols = nc;
    On.ncols = (int)ext_cgid.size();
    Dn.ia.assign((size_t)nc + 1, 0);
    On.ia.assign((size_t)nc + 1, 0);
    for (int i = 0; i < nc; i++) {
      int nd = 0;
      for (int64_t k = Ac.ia[(size_t)i]; k < Ac.ia[(size_t)i + 1]; k++) nd += (Ac.ja[(size_t)k] < nc);
      Dn.ia[(size_t)i + 1] = Dn.ia[(size_t)i] + nd;
      On.ia[(size_t)i + 1] = On.ia[(size_t)i] + (Ac.ia[(size_t)i + 1] - Ac.ia[(size_t)i] - nd);
    }
    Dn.ja.resize((size_t)Dn.nnz());
    Dn.a.resize((size_t)Dn.nnz());
    On.ja.resize((size_t)On.nnz());
    On.a.resize((size_t)On.nnz());
    for (int i = 0; i < nc; i++) {
      int64_t pd = Dn.ia[(size_t)i], po = On.ia[(size_t)i];
      for (int64_t k = Ac.ia[(size_t)i]; k < Ac.ia[(size_t)i + 1]; k++) {
        if (Ac.ja[(size_t)k] < nc) {
          Dn.ja[(size_t)pd] = Ac.ja[(size_t)k];
          Dn.a[(size_t)pd++] = Ac.a[(size_t)k];
        } else {
          On.ja[(size_t)po] = Ac.ja[(size_t)k] - nc;
          On.a[(size_t)po++] = Ac.a[(size_t)k];
        }
      }
    }
    // drop halo columns that no longer appear (keeps col_map_offd tight)
    {
      std::vector<char> usedc(ext_cgid.size(), 0);
      for (int v : On.ja) usedc[(size_t)v] = 1;
      std::vector<int> remap(ext_cgid.size(), -1);
      std::vector<gidx> cm;
      for (size_t q = 0; q < ext_cgid.size(); q++)
        if (usedc[q]) {
          remap[q] = (int)cm.size();
          cm.push_back(ext_cgid[q]);
        }
      for (int &v : On.ja) v = remap[(size_t)v];
      On.ncols = (int)cm.size();
      An->col_map_offd = cm;
    }
    An->build_halo_plan(comm);
    t_phase[3] += wall_time() - tp0;
    L.emplace_back();
    L[(size_t)l + 1].A_own = std::move(An);
    L[(size_t)l + 1].A = L[(size_t)l + 1].A_own.get();
    l++;
  }

  {
    const double tp0 = wall_time();
    apply_cf_ordering();
    t_phase[4] += wall_time() - tp0;
  }

  // per-level norms (host), on the C-first ordered operators
  const int ch = chunk();
  for (size_t li = 0; li < L.size(); li++) {
    AmgLevel &Lv = L[li];
    ParCSR &A = *Lv.A;
    Lv.n = A.nrows;
    std::vector<int> cf_ext;
    if (!Lv.cf.empty()) cf_ext = A.halo_exchange_host_int(comm, Lv.cf);
    level_norms(A, Lv.cf, cf_ext, ch, Lv.diag, Lv.l1gs, Lv.l1jac);
  }

  // coarsest level: dense inverse (relax type 9), every rank holds its own rows
  AmgLevel &Lc = L.back();
  const gidx ng = Lc.A->global_rows();
  if (p.relax_type[2] == 9 && ng <= MAX_DENSE && ng > 0) {
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
  t_phase[5] = wall_time() - t_setup_start;
  host_ready = true;
}

void BoomerAMG::setup_device() {
  MI_REQUIRE(host_ready, "BoomerAMG: setup_device before setup_host");
  ensure_init();
  Comm &comm = current_comm();
  const int ch = chunk();
  for (size_t li = 0; li < L.size(); li++) {
    AmgLevel &Lv = L[li];
    if (li > 0 || !Lv.A->on_device) Lv.A->to_device();
    Lv.d_diag.upload(Lv.diag);
    Lv.d_l1gs.upload(Lv.l1gs);
    Lv.d_l1jac.upload(Lv.l1jac);
    if (!Lv.cf.empty()) {
      std::vector<signed char> c8(Lv.cf.size());
      for (size_t i = 0; i < c8.size(); i++) c8[i] = (signed char)Lv.cf[i];
      Lv.d_cf.upload(c8);
      Lv.d_perm.upload(Lv.perm);
      Lv.dP.upload(Lv.P);
      Lv.dR.upload(Lv.R);
    }
    Lv.u.alloc((size_t)Lv.n);
    Lv.f.alloc((size_t)Lv.n);
    Lv.tmp.alloc((size_t)Lv.n);
    Lv.snap.alloc((size_t)Lv.n);
    if (Lv.n) {
      MI_HIP(hipMemset(Lv.u.p, 0, (size_t)Lv.n * sizeof(double)));
      MI_HIP(hipMemset(Lv.f.p, 0, (size_t)Lv.n * sizeof(double)));
      MI_HIP(hipMemset(Lv.tmp.p, 0, (size_t)Lv.n * sizeof(double)));
      MI_HIP(hipMemset(Lv.snap.p, 0, (size_t)Lv.n * sizeof(double)));
    }
  }
  AmgLevel &Lc = L.back();
  if (Lc.dense) {
    const size_t width = (size_t)comm.size * (size_t)Lc.slot;
    Lc.Cinv.upload(Lc.Cinv_host);
    Lc.fgather.alloc(width);
    Lc.fslot.alloc((size_t)Lc.slot);
    MI_HIP(hipMemset(Lc.fslot.p, 0, ((size_t)Lc.slot + 2) * sizeof(double)));
    MI_HIP(hipMemset(Lc.fgather.p, 0, (width + 2) * sizeof(double)));
  }
  MI_HIP(hipDeviceSynchronize());
  is_setup = true;
  setup_seconds = wall_time() - t_setup_start;
  if (p.print_level > 0 && comm.rank == 0) {
    printf("mi_hypre BoomerAMG setup: %zu levels, operator complexity %.3f, chunk %d, %.3f s\n", L.size(),
           operator_complexity(), ch, setup_seconds);
    printf("   host phases: strength %.2f  pmis %.2f  interp %.2f  galerkin %.2f  C-first ordering %.2f  (host total %.2f) s\n",
           t_phase[0], t_phase[1], t_phase[2], t_phase[3], t_phase[4], t_phase[5]);
    for (size_t li = 0; li < L.size(); li++)
      printf("   level %2zu: local rows %10d  global rows %12lld  local nnz %12lld\n", li, L[li].n,
             (long long)L[li].A->global_rows(), (long long)(L[li].A->diag.nnz() + L[li].A->offd.nnz()));
  }
}

}  // namespace mi
