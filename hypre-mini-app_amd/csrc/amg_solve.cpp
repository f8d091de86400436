// BoomerAMG solve phase on the device: relaxation dispatch, V/W cycle
// (hypre_BoomerAMGCycle, par_cycle.c).  This is
// precondSolvePtr_ = HYPRE_BoomerAMGSolve of src/HypreSystem.cpp:324, called
// once per Arnoldi step from the GMRES loop.
//
// Every level works in its own C-first ordering (amg_setup.cpp,
// apply_cf_ordering) on its own vectors: a C pass sweeps rows [0, nc), an F pass
// rows [nc, n), so the two passes of a sweep together stream the level's matrix
// once.  The second pass of a C/F pair reads the first pass's rows straight
// from the scratch vector (composite source), and the pair ends with a pointer
// swap instead of a copy.
#include <algorithm>
#include <cmath>

#include "amg.hpp"
#include "kernels.hpp"
#include "profile.hpp"
#include "solvers.hpp"

namespace mi {

bool &zero_guess_hint() {
  static bool hint = false;
  return hint;
}

static int g_zero_skip = -1;
int zero_skip_mode() {
  if (g_zero_skip < 0) {
    const char *e = getenv("MI_HYPRE_GS_ZERO_SKIP");
    g_zero_skip = e ? std::max(0, std::min(3, atoi(e))) : 3;
  }
  return g_zero_skip;
}
void set_zero_skip_mode(int mode) { g_zero_skip = std::max(0, std::min(3, mode)); }

namespace {
// level 0 is timed under bench.py's id when that one is enabled, else under the per-level id
int relax_prof_id(int level, bool zero_guess) {
  if (zero_guess) return k::prof_level(k::PROF_LVL_RELAX0, level);  // not a full-operator sweep: own class
  KernelTimer *t = ctx().timer;
  if (level == 0 && t && t->enabled[k::PROF_RELAX_L0]) return k::PROF_RELAX_L0;
  return k::prof_level(k::PROF_LVL_RELAX, level);
}
struct GsKind {
  bool jacobi, l1, fwd, bwd;
  int two_stage;  // relax types 11 / 12: inner iterations of the two-stage Gauss-Seidel
};
GsKind classify(int type) {
  GsKind g{};
  g.two_stage = (type == 11) ? 1 : (type == 12) ? 2 : 0;
  if (g.two_stage) return g;
  g.jacobi = (type == 0 || type == 7 || type == 18);
  g.l1 = (type == 8 || type == 13 || type == 14 || type == 18);
  g.fwd = (type == 3 || type == 6 || type == 8 || type == 13);
  g.bwd = (type == 4 || type == 6 || type == 8 || type == 14);
  if (!g.jacobi && !g.fwd && !g.bwd) fail(4, "BoomerAMG: relax type " + std::to_string(type) + " is not supported");
  return g;
}

// coarsest-level direct solve (relax type 9); false when the level is too large
bool dense_solve(AmgLevel &Lv, Comm &comm, const double *f, double *u, hipStream_t s) {
  if (!Lv.dense) return false;
  if (comm.size == 1) {
    k::dense_matvec(Lv.Cinv.p, f, u, Lv.n, Lv.n, s);
  } else {
    k::copy(f, Lv.fslot.p, Lv.n, s);
    comm.allgather_dev(Lv.fslot.p, Lv.fgather.p, (size_t)Lv.slot * sizeof(double), s);
    ctx().n_allgather++;
    k::dense_matvec(Lv.Cinv.p, Lv.fgather.p, u, Lv.n, comm.size * Lv.slot, s);
  }
  return true;
}
// largest stretch of [row_begin, row_end) without halo entries, cut at the write-back units of the GS kernel
// that will sweep M; false when it is less than half of the range (not worth three launches)
bool interior_range(AmgLevel &Lv, const DevCSR &M, int ch, int row_begin, int row_end, int &ib, int &ie) {
  for (const auto &c : Lv.interior_cache)
    if (c.row_begin == row_begin && c.row_end == row_end && c.op == (const void *)&M) {
      ib = c.ib;
      ie = c.ie;
      return c.ok;
    }
  if (!Lv.halo_rows_ready) {
    const HostCSR &B = Lv.A->offd;
    Lv.halo_rows.clear();
    for (int i = 0; i < B.nrows && (size_t)i + 1 < B.ia.size(); i++)
      if (B.ia[(size_t)i + 1] > B.ia[(size_t)i]) Lv.halo_rows.push_back(i);
    Lv.halo_rows_ready = true;
  }
  const std::vector<int> &hr = Lv.halo_rows;
  const auto lo = std::lower_bound(hr.begin(), hr.end(), row_begin), hi = std::lower_bound(hr.begin(), hr.end(), row_end);
  // gaps between consecutive halo rows of the range (open interval (a, b) = rows a+1 .. b-1)
  int ga = row_begin - 1, gb = row_end;
  if (lo != hi) {
    int best = -1, prev = row_begin - 1;
    for (auto it = lo; it != hi; ++it) {
      if (*it - prev - 1 > best) best = *it - prev - 1, ga = prev, gb = *it;
      prev = *it;
    }
    if (row_end - prev - 1 > best) ga = prev, gb = row_end;
  }
  int b = ga + 1, e = gb;  // rows [b, e) have no halo entries
  const bool tiles = k::gs_uses_tiles(M, ch);
  if (b > row_begin) {
    if (tiles) {
      const auto t = std::lower_bound(M.rb_host.begin(), M.rb_host.end(), b);
      b = (t == M.rb_host.end()) ? e : *t;
    } else {
      b = (b + ch - 1) / ch * ch;
    }
  }
  if (e < row_end) {
    if (tiles) {
      const auto t = std::upper_bound(M.rb_host.begin(), M.rb_host.end(), e);
      e = (t == M.rb_host.begin()) ? b : *(t - 1);
    } else {
      e = e / ch * ch;
    }
  }
  const bool ok = e > b && (long long)(e - b) * 2 >= (long long)(row_end - row_begin);
  Lv.interior_cache.push_back({row_begin, row_end, (const void *)&M, b, e, ok});
  ib = b;
  ie = e;
  return ok;
}

// One hybrid-GS pass over rows [row_begin, row_end) with pre-sweep values lo (rows < split) / hi.  On N > 1
// ranks the pass needs the halo of those values: the neighbour exchange and the per-row halo contribution run on
// the side stream while the rows without halo entries are swept, the rest follows (MI_HYPRE_OVERLAP_HALO=0: in
// order).  zero_halo: the values are all zero -- no exchange at all.
void gs_pass(BoomerAMG &amg, AmgLevel &Lv, const DevCSR &M, bool zero_halo, const double *lo, const double *hi, int split,
             double *out, const double *f, const double *d, const signed char *cf, int points, int ch, const GsKind &g,
             double w, int row_begin, int row_end, int prof, int zero_from, double *tout = nullptr, int t_from = 0) {
  ParCSR &A = *Lv.A;
  Comm &comm = amg.my_comm();
  hipStream_t s = ctx().stream;
  if (zero_halo || comm.size == 1) {
    k::gs_hybrid(M, lo, hi, split, out, f, nullptr, d, cf, points, ch, g.fwd, g.bwd, w, row_begin, row_end, s, prof,
                 zero_from, tout, t_from);
    return;
  }
  static const bool overlap = !(getenv("MI_HYPRE_OVERLAP_HALO") && atoi(getenv("MI_HYPRE_OVERLAP_HALO")) == 0);
  const bool peers = !(A.halo.send_peers.empty() && A.halo.recv_peers.empty());
  int ib = 0, ie = 0;
  if (!overlap || !peers || !interior_range(Lv, M, ch, row_begin, row_end, ib, ie)) {
    const double *offc = A.offd_contrib(comm, lo, s, hi, split);
    k::gs_hybrid(M, lo, hi, split, out, f, offc, d, cf, points, ch, g.fwd, g.bwd, w, row_begin, row_end, s, prof,
                 zero_from, tout, t_from);
    ctx().n_gs_in_order++;
    return;
  }
  ctx().n_gs_overlapped++;
  hipStream_t cs = ctx().comm_stream;
  A.halo_pack(lo, s, hi, split);
  MI_HIP(hipEventRecord(ctx().ev_packed, s));
  // d_offc is zero outside the rows with halo entries, which the interior launch does not touch
  const double *offc = A.d_offd.nrows_c > 0 ? A.d_offc.p : nullptr;
  k::gs_hybrid(M, lo, hi, split, out, f, offc, d, cf, points, ch, g.fwd, g.bwd, w, ib, ie, s, prof, zero_from, tout, t_from);
  MI_HIP(hipStreamWaitEvent(cs, ctx().ev_packed, 0));
  A.halo_transfer(comm, cs);
  if (A.d_offd.nrows_c > 0) k::spmv_offd_set(A.d_offd, A.halo.d_xext.p, A.d_offc.p, cs);
  MI_HIP(hipEventRecord(ctx().ev_halo, cs));
  MI_HIP(hipStreamWaitEvent(s, ctx().ev_halo, 0));
  if (ib > row_begin)
    k::gs_hybrid(M, lo, hi, split, out, f, offc, d, cf, points, ch, g.fwd, g.bwd, w, row_begin, ib, s, prof, zero_from, tout,
                 t_from);
  if (row_end > ie)
    k::gs_hybrid(M, lo, hi, split, out, f, offc, d, cf, points, ch, g.fwd, g.bwd, w, ie, row_end, s, prof, zero_from, tout,
                 t_from);
}
}  // namespace

// one relaxation call on the level's current vector Lv.u (level ordering), in place
void BoomerAMG::relax(int level, int type, int points, const double *f, bool u_is_zero) {
  AmgLevel &Lv = L[(size_t)level];
  ParCSR &A = *Lv.A;
  Comm &comm = my_comm();
  hipStream_t s = ctx().stream;
  const int prof = relax_prof_id(level, u_is_zero && zero_skip_mode() > 0);
  double *u = Lv.u.p;
  if (type == 9) {
    if (dense_solve(Lv, comm, f, u, s)) return;
    type = p.relax_type[0];  // coarsest level too large for a dense solve
  }
  Lv.t_valid = false;
  const GsKind g = classify(type);
  const double w = p.relax_weight * p.outer_weight;
  const bool has_cf = Lv.has_cf && !Lv.cf.empty();
  const signed char *cf = has_cf ? Lv.d_cf.p : nullptr;
  if (!has_cf) points = 0;
  if (g.two_stage) {
    // par_relax.c hypre_BoomerAMGRelax11/12TwoStageGaussSeidel: r = w (f - A u); z_0 = D^-1 r; z_k = D^-1 L z_(k-1)
    // (L: strictly lower part of this rank's diag block); u += z_0 - z_1 (+ z_2).  The C/F marker plays no part.
    if (Lv.ts_work.n != (size_t)Lv.n) Lv.ts_work.alloc((size_t)Lv.n);
    A.matvec(comm, -w, u, w, f, Lv.tmp.p, s, prof);
    double *z = Lv.snap.p, *zn = Lv.ts_work.p;
    k::two_stage_first(Lv.tmp.p, Lv.d_diag.p, z, u, Lv.n, s);
    double sign = -1.0;
    for (int it = 0; it < g.two_stage; it++) {
      k::two_stage_lower(A.d_diag, Lv.d_diag.p, z, sign, zn, u, s);
      std::swap(z, zn);
      sign = -sign;
    }
    return;
  }
  if (g.jacobi) {
    // a zero vector has a zero halo: no exchange, no halo contribution
    const double *offc = u_is_zero ? nullptr : A.offd_contrib(comm, u, s);
    k::jacobi(A.d_diag, u, Lv.snap.p, f, offc, type == 18 ? Lv.d_l1jac.p : Lv.d_diag.p, cf, points, w, s, prof);
    std::swap(Lv.u.p, Lv.snap.p);  // every row was written
    return;
  }
  const int row_begin = (points == -1) ? Lv.nc : 0;
  const int row_end = (points == 1) ? Lv.nc : Lv.n;
  if (row_end <= row_begin) {
    // nothing to sweep HERE, but the neighbours' passes still expect this rank's values
    if (!u_is_zero) A.halo_exchange(comm, u, s);
    return;
  }
  const int ch = chunk();
  const bool zs = u_is_zero && zero_skip_mode() > 0;
  gs_pass(*this, Lv, (zs && Lv.has_Az && Lv.Az_chunk == ch) ? Lv.Az : A.d_diag, u_is_zero, u, u, 0, Lv.snap.p, f,
          g.l1 ? Lv.d_l1gs.p : Lv.d_diag.p, cf, points, ch, g, w, row_begin, row_end, prof, zs ? 0 : k::GS_NO_ZEROS);
  if (row_begin == 0 && row_end == Lv.n) {
    std::swap(Lv.u.p, Lv.snap.p);
  } else {  // lone C or F pass: only the swept chunks were written
    const long long r0 = (long long)(row_begin / ch) * ch;
    const long long r1 = std::min<long long>(((long long)row_end + ch - 1) / ch * ch, Lv.n);
    k::copy(Lv.snap.p + r0, u + r0, (int)(r1 - r0), s);
  }
}

// a C/F pair of hybrid-GS passes (first = +1: C then F, first = -1: F then C)
// without the intermediate copy
void BoomerAMG::relax_pair(int level, int type, int first, const double *f, bool u_is_zero) {
  AmgLevel &Lv = L[(size_t)level];
  const GsKind g = classify(type);
  if (g.jacobi || g.two_stage || Lv.cf.empty() || Lv.nc == 0 || Lv.nc == Lv.n) {
    relax(level, type, first, f, u_is_zero);
    relax(level, type, -first, f, false);
    return;
  }
  ParCSR &A = *Lv.A;
  const int prof = relax_prof_id(level, u_is_zero && zero_skip_mode() > 0);
  const double w = p.relax_weight * p.outer_weight;
  const double *d = g.l1 ? Lv.d_l1gs.p : Lv.d_diag.p;
  const int ch = chunk(), nc = Lv.nc, n = Lv.n;
  double *u = Lv.u.p, *sn = Lv.snap.p;
  // pass 1: everything is read from u; its swept rows land in snap
  // on a zero guess pass 1 gathers nothing, and pass 2 of a C-then-F pair only the C columns pass 1 wrote
  // (and both run on the level's zero-guess sub-operator, which leaves out what multiplies zeros)
  const bool zs = u_is_zero && zero_skip_mode() > 0;
  const int z1 = zs ? 0 : k::GS_NO_ZEROS, z2 = (zs && first == 1) ? nc : k::GS_NO_ZEROS;
  const bool use_Az = zs && Lv.has_Az && Lv.Az_chunk == ch;
  const DevCSR &A1 = use_Az ? Lv.Az : A.d_diag;
  const DevCSR &A2 = (use_Az && first == 1) ? Lv.Az : A.d_diag;
  // the F pass of a zero-guess C-then-F pair also leaves f - A_FC u_C for the residual that follows (tile kernel)
  Lv.t_valid = false;
  const bool give_t = use_Az && first == 1 && Lv.has_Ar && zero_skip_mode() > 2 && k::gs_uses_tiles(A2, ch);
  if (first == 1)
    gs_pass(*this, Lv, A1, u_is_zero, u, u, 0, sn, f, d, Lv.d_cf.p, 1, ch, g, w, 0, nc, prof, z1);
  else
    gs_pass(*this, Lv, A1, u_is_zero, u, u, 0, sn, f, d, Lv.d_cf.p, -1, ch, g, w, nc, n, prof, z1);
  // pass 2: the rows pass 1 updated are read from snap, the others from u
  const double *lo = (first == 1) ? sn : u;
  const double *hi = (first == 1) ? u : sn;
  if (first == 1 && give_t) {
    // (rows before t_from -- the C rows and the chunk that straddles nc -- keep f itself: the residual reads them
    // from f through its composite right-hand side, no copy)
    gs_pass(*this, Lv, A2, false, lo, hi, nc, sn, f, d, Lv.d_cf.p, -1, ch, g, w, nc, n, prof, z2, Lv.tvec.p, Lv.t_from);
    Lv.t_valid = true;
  } else if (first == 1)
    gs_pass(*this, Lv, A2, false, lo, hi, nc, sn, f, d, Lv.d_cf.p, -1, ch, g, w, nc, n, prof, z2);
  else
    gs_pass(*this, Lv, A2, false, lo, hi, nc, sn, f, d, Lv.d_cf.p, 1, ch, g, w, 0, nc, prof, z2);
  std::swap(Lv.u.p, Lv.snap.p);  // snap now holds every row
}

// which: 0 down, 1 up, 2 coarsest.  relax_order 1: C then F going down, F then
// C going up, all points on the coarsest level (hypre_BoomerAMGRelaxIF)
void BoomerAMG::relax_sweeps(int level, int which, const double *f, bool u_is_zero) {
  const int type = p.relax_type[which];
  const bool has_cf = L[(size_t)level].has_cf;
  if (L[(size_t)level].smoother && which != 2) {
    // complex smoother (par_cycle.c: smooth_num_levels > level, smooth_type 5): per sweep ilu_max_iter times
    // u += (LU)^-1 (f - A u); on a zero guess the first residual is f itself
    AmgLevel &Lv = L[(size_t)level];
    IluSolver &ilu = *Lv.smoother;
    Comm &comm = my_comm();
    hipStream_t s = ctx().stream;
    Lv.t_valid = false;
    bool zero = u_is_zero;
    for (int sw = 0; sw < p.num_sweeps[which]; sw++)
      for (int it = 0; it < ilu.max_iter; it++) {
        if (zero) {
          ilu.apply(f, Lv.u.p);
        } else {
          Lv.A->matvec(comm, -1.0, Lv.u.p, 1.0, f, Lv.tmp.p, s, k::prof_level(k::PROF_LVL_RELAX, level));
          ilu.apply(Lv.tmp.p, Lv.snap.p);
          k::axpy(1.0, Lv.snap.p, Lv.u.p, Lv.n, s);
        }
        zero = false;
      }
    return;
  }
  for (int sw = 0; sw < p.num_sweeps[which]; sw++) {
    const bool zero = u_is_zero && sw == 0;
    if (which == 2 || p.relax_order != 1 || !has_cf)
      relax(level, type, 0, f, zero);
    else
      relax_pair(level, type, which == 0 ? 1 : -1, f, zero);
  }
}

bool BoomerAMG::zero_cycle_ignores_u(int level) {
  static const bool enabled = !(getenv("MI_HYPRE_SKIP_ZERO_FILL") && atoi(getenv("MI_HYPRE_SKIP_ZERO_FILL")) == 0);
  if (!enabled || my_comm().size != 1) return false;  // N > 1: the second pass sends pre-sweep (zero) values of its halo rows
  const int nlev = (int)L.size();
  AmgLevel &Lv = L[(size_t)level];
  if (level == collapsed_level || level == collapsed_level2) return true;  // u = B f
  if (level == nlev - 1) {
    if (tail) return false;
    return p.relax_type[2] == 9 && Lv.dense && p.num_sweeps[2] > 0;  // u = C^-1 f
  }
  if (Lv.smoother || p.num_sweeps[0] < 1 || zero_skip_mode() < 1) return false;
  const int type = p.relax_type[0];
  if (type == 9 || type == 11 || type == 12) return false;
  const GsKind g = classify(type);
  if (g.jacobi || !(g.fwd || g.bwd)) return false;
  const int ch = chunk();
  const bool use_Az = Lv.has_Az && Lv.Az_chunk == ch;
  const bool has_cf = Lv.has_cf && !Lv.cf.empty();
  if (p.relax_order == 1 && has_cf && (Lv.nc == 0 || Lv.nc == Lv.n)) return false;  // degenerate pair: two lone passes
  // C-then-F pair: both passes run on Az when the level has one; a lone full sweep likewise
  const DevCSR &M = use_Az ? Lv.Az : Lv.A->d_diag;
  return k::gs_ignores_zero_vector(M, ch);
}

// one cycle on the level's own f (Lv.f) and u (Lv.u)
void BoomerAMG::cycle(int level, bool u_is_zero) {
  const int nlev = (int)L.size();
  if (level == 0 && collapsed_level >= 0 && collapsed_signature != cycle_signature())
    build_collapsed_tail();  // a cycle parameter changed after Setup: the tabulated maps are of another cycle
  AmgLevel &Lv = L[(size_t)level];
  if (level == collapsed_level2 && u_is_zero) {  // the tabulated map of this level's whole sub-cycle
    k::dense_matvec_t(collapsed_Bt2.p, Lv.f.p, Lv.u.p, collapsed_n2, ctx().stream);
    return;
  }
  if (level == collapsed_level && u_is_zero) {
    k::dense_matvec_t(collapsed_Bt.p, Lv.f.p, Lv.u.p, collapsed_n, ctx().stream);
    return;
  }
  if (level == nlev - 1) {
    if (tail)
      tail_cycle(u_is_zero);
    else
      relax_sweeps(level, 2, Lv.f.p, u_is_zero);
    return;
  }
  AmgLevel &Ln = L[(size_t)level + 1];
  Comm &comm = my_comm();
  hipStream_t s = ctx().stream;
  relax_sweeps(level, 0, Lv.f.p, u_is_zero && p.num_sweeps[0] > 0);
  // r = f - A u ; f_c = P^T r ; u_c = 0
  if (Lv.t_valid && Lv.has_Ar)  // F rows: f - A_FC u_C is in tvec already, only their F columns are left
    Lv.A->matvec(comm, -1.0, Lv.u.p, 1.0, Lv.tvec.p, Lv.tmp.p, s, k::prof_level(k::PROF_LVL_RESID, level), &Lv.Ar, Lv.f.p,
                 std::min(Lv.t_from, Lv.n));
  else
    Lv.A->matvec(comm, -1.0, Lv.u.p, 1.0, Lv.f.p, Lv.tmp.p, s, k::prof_level(k::PROF_LVL_RESID, level));
  Lv.t_valid = false;
  Lv.Rm->matvec(comm, 1.0, Lv.tmp.p, 0.0, nullptr, Ln.f.p, s, k::prof_level(k::PROF_LVL_RESTRICT, level));
  if (!zero_cycle_ignores_u(level + 1)) k::fill(Ln.u.p, Ln.n, 0.0, s);
  // the coarsest level is visited once per cycle; with a redundant tail the count continues into the tail
  const bool tail_next = tail && level + 1 == nlev - 1;
  const bool next_is_coarsest = tail_next ? tail->L.size() == 1 : level + 1 == nlev - 1;
  const int ncyc = (p.cycle_type == 2 && !next_is_coarsest) ? 2 : 1;
  for (int c = 0; c < ncyc; c++) cycle(level + 1, c == 0);
  // u += P e
  if (tail_next) {  // every rank holds the whole coarse correction: halo values without an exchange
    const int next = (int)Lv.Pm->col_map_offd.size();
    if (next) k::gather(tail_e.p, tail_pcol.p, Lv.Pm->halo.d_xext.p, next, s);
    Lv.Pm->matvec_ext_ready(1.0, Ln.u.p, 1.0, Lv.u.p, Lv.u.p, s);
  } else {
    Lv.Pm->matvec(comm, 1.0, Ln.u.p, 1.0, Lv.u.p, Lv.u.p, s, k::prof_level(k::PROF_LVL_PROLONG, level));
  }
  relax_sweeps(level, 1, Lv.f.p, false);
}

// tabulate the map f -> u of cycle(lt, true) by cycling the unit vectors (row j of Bt = column j of the map)
namespace {
// HIP events recorded inside a captured graph would be replayed into the same event objects: with any profiling class
// switched on the tabulation takes the plain launches
bool any_profile_class_on() {
  KernelTimer *t = ctx().timer;
  if (!t) return false;
  for (int i = 0; i < k::PROF_COUNT; i++)
    if (t->enabled[i]) return true;
  return false;
}
}  // namespace

void BoomerAMG::tabulate_cycle(int lt, DVec<double> &Bt) {
  AmgLevel &Lv = L[(size_t)lt];
  const int n = Lv.n;
  hipStream_t s = ctx().stream;
  Bt.alloc((size_t)n * (size_t)n);
  double *own_f = Lv.f.p;
  DVec<double> e((size_t)n);
  DVec<int> col(1);
  zero_on_stream(e.p, (size_t)n * sizeof(double));
  MI_HIP(hipMemsetAsync(col.p, 0, sizeof(int), s));
  Lv.f.p = e.p;
  // One column = unit vector, (fill,) cycle, store: ~10 launches of a few microseconds each, n columns -- tens of
  // thousands of dispatches whose cost is the host's launch path (0.7 s of a 5.3 s setup at 512^3, and the loop
  // rocprofv3's counter collection fell over in).  The column number lives on the device, so the sequence
  // is the same for every column: it is captured once as a HIP graph and replayed (MI_HYPRE_TAIL_GRAPH=0, or a capture
  // that fails: plain launches).  The first column always runs plainly -- anything a level allocates on first use
  // happens there, outside the capture.
  auto one_column = [&]() {
    k::tab_unit(e.p, col.p, s);
    if (!zero_cycle_ignores_u(lt)) k::fill(Lv.u.p, n, 0.0, s);
    cycle(lt, true);
    k::tab_store(Bt.p, Lv.u.p, n, col.p, s);
  };
  static const bool want_graph = !(getenv("MI_HYPRE_TAIL_GRAPH") && atoi(getenv("MI_HYPRE_TAIL_GRAPH")) == 0);
  try {
    int done = 0;
    if (n > 0) {
      one_column();
      done = 1;
    }
    hipGraphExec_t exec = nullptr;
    if (want_graph && n > 2 && !any_profile_class_on()) {
      MI_HIP(hipStreamSynchronize(s));
      hipGraph_t graph = nullptr;
      bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        try {
          one_column();
        } catch (...) {
          ok = false;
        }
        if (hipStreamEndCapture(s, &graph) != hipSuccess || !graph) ok = false;
      }
      if (ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        ok = false;
        exec = nullptr;
      }
      if (graph) (void)hipGraphDestroy(graph);
      if (!ok) {
        (void)hipGetLastError();
        exec = nullptr;
        // (the capture recorded no work: the column counter still says `done`)
      }
    }
    for (int j = done; j < n; j++) {
      if (exec)
        MI_HIP(hipGraphLaunch(exec, s));
      else
        one_column();
    }
    MI_HIP(hipStreamSynchronize(s));
    if (exec) (void)hipGraphExecDestroy(exec);
  } catch (...) {
    Lv.f.p = own_f;
    throw;
  }
  Lv.f.p = own_f;
}

std::vector<double> BoomerAMG::cycle_signature() const {
  std::vector<double> v;
  for (int k = 0; k < 3; k++) v.push_back(p.relax_type[k]), v.push_back(p.num_sweeps[k]);
  v.push_back(p.relax_order);
  v.push_back(p.relax_weight);
  v.push_back(p.outer_weight);
  v.push_back(p.cycle_type);
  v.push_back(chunk());
  v.push_back(zero_skip_mode());
  v.push_back(p.smooth_type);
  v.push_back(p.smooth_num_levels);
  v.push_back(p.smooth_num_sweeps);
  v.push_back(p.ilu_max_iter);
  return v;
}

void BoomerAMG::build_collapsed_tail() {
  collapsed_signature = cycle_signature();
  collapsed_level = collapsed_level2 = -1;
  collapsed_n = collapsed_n2 = 0;
  collapsed_Bt.release();
  collapsed_Bt2.release();
  static const long long max_rows = getenv("MI_HYPRE_DENSE_TAIL_ROWS") ? atoll(getenv("MI_HYPRE_DENSE_TAIL_ROWS")) : 1024;
  // (the second stage is an option since the end of round 4: tabulating the 4275 columns of level 7 at 512^3 costs 0.29 s of
  // a 2.9 s setup and saves ~9 short launches per cycle, which the solve does not show -- 669.1 / 670.6 ms without against
  // 670.0 / 672.2 ms with, alternating runs on one box)
  static const long long max_rows2 = getenv("MI_HYPRE_DENSE_TAIL_ROWS2") ? atoll(getenv("MI_HYPRE_DENSE_TAIL_ROWS2")) : 0;
  if (max_rows <= 0 || my_comm().size != 1 || tail) return;
  const int nlev = (int)L.size();
  int lt = -1;
  for (int l = 1; l + 1 < nlev; l++)  // the coarsest level alone is one launch already
    if (L[(size_t)l].n <= max_rows && L[(size_t)l].n > 0) {
      lt = l;
      break;
    }
  if (lt < 0) return;
  DVec<double> Bt;
  tabulate_cycle(lt, Bt);
  collapsed_Bt = std::move(Bt);
  collapsed_n = L[(size_t)lt].n;
  collapsed_level = lt;
  // second stage: the level above, through the map just built (one sub-cycle = its own ~9 launches + one dense
  // product: tabulating ~4000 columns costs a few tenths of a second; 512^3: level 7, 4303 rows, 148 MB)
  if (lt - 1 >= 1 && L[(size_t)lt - 1].n <= max_rows2) {
    DVec<double> Bt2;
    tabulate_cycle(lt - 1, Bt2);
    collapsed_Bt2 = std::move(Bt2);
    collapsed_n2 = L[(size_t)lt - 1].n;
    collapsed_level2 = lt - 1;
  }
}

// the stub level's right-hand side (this rank's slice) -> whole level on every rank -> one cycle of the
// redundant hierarchy -> this rank's slice of the correction
void BoomerAMG::tail_cycle(bool zero_guess) {
  AmgLevel &Lv = L.back();
  Comm &comm = my_comm();
  hipStream_t s = ctx().stream;
  const int ng = tail_A->nrows;
  if (Lv.n) k::copy(Lv.f.p, tail_fslot.p, Lv.n, s);
  comm.allgather_dev(tail_fslot.p, tail_fgather.p, (size_t)tail_slot * sizeof(double), s);
  ctx().n_allgather++;
  k::gather(tail_fgather.p, tail_map.p, tail_f.p, ng, s);
  tail->apply_global(tail_f.p, tail_e.p, zero_guess);
  if (Lv.n) k::copy(tail_e.p + tail_start, Lv.u.p, Lv.n, s);
}

// one cycle on vectors in the caller's (natural) ordering of this hierarchy's fine level; e is the initial
// guess unless zero_guess
void BoomerAMG::apply_global(const double *f, double *e, bool zero_guess) {
  hipStream_t s = ctx().stream;
  AmgLevel &L0 = L[0];
  const bool permuted = !L0.perm.empty();
  if (permuted)
    k::gather(f, L0.d_perm.p, L0.f.p, L0.n, s);
  else
    k::copy(f, L0.f.p, L0.n, s);
  if (zero_guess) {
    if (!zero_cycle_ignores_u(0)) k::fill(L0.u.p, L0.n, 0.0, s);
  } else if (permuted)
    k::gather(e, L0.d_perm.p, L0.u.p, L0.n, s);
  else
    k::copy(e, L0.u.p, L0.n, s);
  cycle(0, zero_guess);
  if (permuted)
    k::scatter_set(e, L0.d_perm.p, L0.u.p, L0.n, s);
  else
    k::copy(L0.u.p, e, L0.n, s);
}

void BoomerAMG::solve(ParCSR &A, ParVector &b, ParVector &x) {
  if (!is_setup) setup(A);
  MI_REQUIRE(x.ncomp == b.ncomp, "BoomerAMGSolve: b and x differ in their number of components");
  MI_REQUIRE(x.n == L[0].n && b.n == L[0].n, "BoomerAMGSolve: vector size does not match the matrix");
  Comm &comm = my_comm();
  hipStream_t s = ctx().stream;
  AmgLevel &L0 = L[0];
  const bool permuted = !L0.perm.empty();
  const int n = L0.n, nc = b.ncomp;  // a multivector is cycled component by component
  int it = 0;
  double rel = 0.0, bn = 0.0;
  if (p.tol > 0.0) bn = std::sqrt(par_dot_host(comm, b.all(), b.all(), b.len(), s));
  // a Krylov solver that has just zeroed x says so (krylov.cpp): the first cycle
  // then needs neither x nor its halo
  const bool zero_first = zero_guess_hint();
  zero_guess_hint() = false;
  while (it < p.max_iter) {
    const bool zero = zero_first && it == 0;
    for (int c = 0; c < nc; c++) {
      const double *bc = b.all() + (size_t)c * (size_t)n;
      double *xc = x.all() + (size_t)c * (size_t)n;
      // caller order -> level-0 (C-first) order
      if (it == 0 || nc > 1) {
        if (permuted)
          k::gather(bc, L0.d_perm.p, L0.f.p, n, s);
        else
          k::copy(bc, L0.f.p, n, s);
      }
      if (zero) {
        if (!zero_cycle_ignores_u(0)) k::fill(L0.u.p, n, 0.0, s);
      } else if (permuted)
        k::gather(xc, L0.d_perm.p, L0.u.p, n, s);
      else
        k::copy(xc, L0.u.p, n, s);
      cycle(0, zero);
      if (permuted)
        k::scatter_set(xc, L0.d_perm.p, L0.u.p, n, s);
      else
        k::copy(L0.u.p, xc, n, s);
    }
    it++;
    if (p.tol > 0.0) {
      double rr = 0.0;
      for (int c = 0; c < nc; c++) {
        const size_t o = (size_t)c * (size_t)n;
        A.matvec(comm, -1.0, x.all() + o, 1.0, b.all() + o, L0.tmp.p, s);
        rr += par_dot_host(comm, L0.tmp.p, L0.tmp.p, n, s);
      }
      const double rn = std::sqrt(rr);
      rel = (bn > 0.0) ? rn / bn : rn;
      if (p.print_level > 1 && comm.rank == 0) printf("    BoomerAMG cycle %3d   ||r||/||b|| = %e\n", it, rel);
      if (rel <= p.tol) break;
    }
  }
  num_iterations = it;
  final_rel_res = rel;
}

}  // namespace mi
