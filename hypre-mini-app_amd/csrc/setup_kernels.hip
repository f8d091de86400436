// Device kernels of the AMG setup phase: sparse matrix products (Galerkin
// operator), transposes, row permutations.  See setup_kernels.hpp.
//
// Bit-exactness contract: every floating-point result equals what the host
// code / the oracle computes -- products and sums are individually rounded
// (this file is built with -ffp-contract=off) and each output entry is summed
// by ONE thread in the reference order.  Everything else is integer work.
//
// SpGEMM layout: one sub-wave group (8 / 16 / 64 lanes, by the row's product
// count T) or one 256-thread block per output row.  The distinct output
// columns of the row are collected in an LDS hash set, ranked (counting sort by
// comparison), and then every lane owns output entries: it walks A's row in
// stored order and finds its column in B's sorted rows by binary search, so the
// sum is taken in exactly the reference order without any atomics on doubles.
#include "setup_kernels.hpp"

#include <algorithm>

#include "kernels.hpp"
#include "parcsr.hpp"

namespace mi {
namespace sk {
namespace {

constexpr int EMPTY = 0x7fffffff;
constexpr int BLK = 256;

// A launch may not exceed 2^32 - 1 threads per grid dimension.  Kernels that give a row 8 ... 256 lanes pass that with
// tens of millions of rows (27-point operator at 400^3: 64 M rows x 256 lanes -- the launch silently ran on the
// remainder modulo 2^32 and left most interpolation rows empty): such grids are folded into two dimensions, and the
// kernels number their workgroups with bid().  Workgroups beyond the work fall out at the kernels' bounds checks.
__device__ __forceinline__ long long bid() { return (long long)blockIdx.y * gridDim.x + blockIdx.x; }
inline dim3 grid_for(long long nblocks) {
  constexpr long long GX = 1 << 20;  // x 256 lanes = 2^28 threads per row of the grid
  if (nblocks < 1) nblocks = 1;
  if (nblocks <= GX) return dim3((unsigned)nblocks);
  return dim3((unsigned)GX, (unsigned)((nblocks + GX - 1) / GX));
}

__device__ __forceinline__ bool hs_insert(int *tab, int log_h, int j) {
  const unsigned mask = (1u << log_h) - 1u;
  unsigned s = ((unsigned)j * 2654435761u) >> (32 - log_h);
  while (true) {
    const int old = atomicCAS(&tab[s], EMPTY, j);
    if (old == EMPTY) return true;
    if (old == j) return false;
    s = (s + 1u) & mask;
  }
}

// value of C(i, j): products in the stored order of A's row, first assigned, rest added
__device__ __forceinline__ double row_dot(long long a0, long long a1, const int *__restrict__ Aja,
                                          const double *__restrict__ Aa, const long long *__restrict__ Bia,
                                          const int *__restrict__ Bja, const double *__restrict__ Ba, int j) {
  double acc = 0.0;
  bool first = true;
  for (long long ka = a0; ka < a1; ka++) {
    const int kr = Aja[ka];
    long long lo = Bia[kr];
    const long long end = Bia[kr + 1];
    long long hi = end;
    while (lo < hi) {
      const long long mid = (lo + hi) >> 1;
      if (Bja[mid] < j)
        lo = mid + 1;
      else
        hi = mid;
    }
    if (lo < end && Bja[lo] == j) {
      const double prod = Aa[ka] * Ba[lo];
      acc = first ? prod : acc + prod;
      first = false;
    }
  }
  return acc;
}

// ---------------------------------------------------------------- product counts and row bins
__global__ __launch_bounds__(BLK) void row_products_k(int n, const long long *__restrict__ Aia,
                                                      const int *__restrict__ Aja, const long long *__restrict__ Bia,
                                                      int *__restrict__ T, int *__restrict__ bin_count,
                                                      int *__restrict__ tmax) {
  __shared__ int hist[4];
  __shared__ int smax;
  if (threadIdx.x < 4) hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) smax = 0;
  __syncthreads();
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n) {
    long long t = 0;
    for (long long ka = Aia[i]; ka < Aia[i + 1]; ka++) {
      const int kr = Aja[ka];
      t += Bia[kr + 1] - Bia[kr];
    }
    const int ti = t > 0x3fffffff ? 0x3fffffff : (int)t;
    T[i] = ti;
    const int b = ti <= 32 ? 0 : ti <= 128 ? 1 : ti <= 512 ? 2 : 3;
    atomicAdd(&hist[b], 1);
    atomicMax(&smax, ti);
  }
  __syncthreads();
  if (threadIdx.x < 4 && hist[threadIdx.x]) atomicAdd(&bin_count[threadIdx.x], hist[threadIdx.x]);
  if (threadIdx.x == 0) atomicMax(tmax, smax);
}

// rows of every bin, contiguous per bin (order inside a bin is irrelevant: a row writes only its own output)
__global__ __launch_bounds__(BLK) void bin_fill_k(int n, const int *__restrict__ T, const int *__restrict__ bin_start,
                                                  int *__restrict__ bin_cursor, int *__restrict__ rows) {
  __shared__ int hist[4], base[4];
  if (threadIdx.x < 4) hist[threadIdx.x] = 0;
  __syncthreads();
  const long long i = bid() * BLK + threadIdx.x;
  int b = -1, off = 0;
  if (i < n) {
    const int ti = T[i];
    b = ti <= 32 ? 0 : ti <= 128 ? 1 : ti <= 512 ? 2 : 3;
    off = atomicAdd(&hist[b], 1);
  }
  __syncthreads();
  if (threadIdx.x < 4) base[threadIdx.x] = hist[threadIdx.x] ? atomicAdd(&bin_cursor[threadIdx.x], hist[threadIdx.x]) : 0;
  __syncthreads();
  if (b >= 0) rows[bin_start[b] + base[b] + off] = (int)i;
}

// Lanes of one wave that hand LDS data to each other between two steps (groups of G <= 64 lanes never straddle a wave):
// the wave's LDS operations execute in program order, so all that is needed is that the compiler keeps that order.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// Between the phases of a kernel whose rows are owned by groups of G lanes: a group inside one wave needs no workgroup
// barrier (with one, every wave of the workgroup waits for the slowest row of all of them at every phase).
template <int G>
__device__ __forceinline__ void group_sync() {
  if (G <= 64)
    wave_lds_sync();
  else
    __syncthreads();
}

// rows of one bin split by the EXACT number of distinct columns the symbolic pass found (order inside a part is irrelevant)
__global__ __launch_bounds__(BLK) void split_rows_k(int nlist, const int *__restrict__ rows, const int *__restrict__ nout,
                                                    int thresh, int *__restrict__ small, int *__restrict__ big,
                                                    int *__restrict__ counters) {
  __shared__ int cnt[2], base[2];
  if (threadIdx.x < 2) cnt[threadIdx.x] = 0;
  __syncthreads();
  const long long k = bid() * BLK + threadIdx.x;
  int row = 0, part = -1, off = 0;
  if (k < nlist) {
    row = rows[k];
    part = nout[row] <= thresh ? 0 : 1;
    off = atomicAdd(&cnt[part], 1);
  }
  __syncthreads();
  if (threadIdx.x < 2) base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&counters[threadIdx.x], cnt[threadIdx.x]) : 0;
  __syncthreads();
  if (part >= 0) (part == 0 ? small : big)[base[part] + off] = row;
}

// ---------------------------------------------------------------- SpGEMM, one group of G lanes per row
// S = lanes that share one row of B during the hash phase (power of two <= G)
template <int G, int CAP, bool NUMERIC>
__global__ __launch_bounds__(BLK) void spgemm_group_k(int nlist, const int *__restrict__ rows, int S,
                                                      const long long *__restrict__ Aia, const int *__restrict__ Aja,
                                                      const double *__restrict__ Aa, const long long *__restrict__ Bia,
                                                      const int *__restrict__ Bja, const double *__restrict__ Ba,
                                                      int *__restrict__ nout, const long long *__restrict__ Cia,
                                                      int *__restrict__ Cja, double *__restrict__ Ca) {
  constexpr int H = 2 * CAP;
  constexpr int LOGH = (H == 64) ? 6 : (H == 256) ? 8 : 10;
  static_assert(H == 64 || H == 256 || H == 1024, "table size");
  constexpr int GP = BLK / G;
  __shared__ int tab[GP][H];
  __shared__ int list[GP][CAP];
  __shared__ int cnt[GP];
  const int g = threadIdx.x / G, lane = threadIdx.x % G;
  const long long gi = bid() * GP + g;
  const bool active = gi < nlist;
  const int row = active ? rows[gi] : 0;
  for (int t = lane; t < H; t += G) tab[g][t] = EMPTY;
  if (lane == 0) cnt[g] = 0;
  group_sync<G>();
  long long a0 = 0, a1 = 0;
  if (active) {
    a0 = Aia[row];
    a1 = Aia[row + 1];
    const int sub = lane / S, sl = lane % S, nsub = G / S;
    for (long long ka = a0 + sub; ka < a1; ka += nsub) {
      const int kr = Aja[ka];
      const long long b1 = Bia[kr + 1];
      for (long long kb = Bia[kr] + sl; kb < b1; kb += S) {
        const int j = Bja[kb];
        if (hs_insert(tab[g], LOGH, j)) list[g][atomicAdd(&cnt[g], 1)] = j;
      }
    }
  }
  group_sync<G>();
  if (!active) return;
  const int no = cnt[g];
  if (!NUMERIC) {
    if (lane == 0) nout[row] = no;
    return;
  }
  const long long c0 = Cia[row];
  for (int r = lane; r < no; r += G) {
    const int j = list[g][r];
    int rank = 0;
    for (int t = 0; t < no; t++) rank += (list[g][t] < j);
    Cja[c0 + rank] = j;
    Ca[c0 + rank] = row_dot(a0, a1, Aja, Aa, Bia, Bja, Ba, j);
  }
}

// Numeric phase of the same product (round 4).  spgemm_group_k<.,.,true> finds every entry of C by its own walk over
// A's row with a binary search in each row of B: (entries of C) x (entries of A's row) x log(row of B) dependent loads --
// 13 x the products of a Galerkin row.  Here the group walks A's row ONCE, in stored order, one entry per step; the G
// lanes take the entries of that row of B (distinct columns), and each adds its product to the column's slot of an
// LDS hash table: first product assigned, the rest added, in the order of A's row -- the order of the host loop and of
// row_dot, so the sums are the same bits.  Steps are separated by wave_lds_sync (a step's slot may be the next step's).
template <int G, int CAP>
__global__ __launch_bounds__(BLK) void spgemm_accum_k(int nlist, const int *__restrict__ rows,
                                                      const long long *__restrict__ Aia, const int *__restrict__ Aja,
                                                      const double *__restrict__ Aa, const long long *__restrict__ Bia,
                                                      const int *__restrict__ Bja, const double *__restrict__ Ba,
                                                      const long long *__restrict__ Cia, int *__restrict__ Cja,
                                                      double *__restrict__ Ca) {
  constexpr int H = 2 * CAP;
  constexpr int LOGH = (H == 64) ? 6 : (H == 256) ? 8 : 10;
  static_assert(H == 64 || H == 256 || H == 1024, "table size");
  static_assert(G <= 64 && 64 % G == 0, "a group lives inside one wave");
  constexpr int GP = BLK / G;
  __shared__ double val[GP][H];
  __shared__ int tab[GP][H];
  __shared__ int keys[GP][CAP];             // discovered columns, in discovery order
  __shared__ unsigned short slots[GP][CAP];  // their table slots
  __shared__ int cnt[GP];
  const int g = threadIdx.x / G, lane = threadIdx.x % G;
  const long long gi = bid() * GP + g;
  const bool active = gi < nlist;
  const int row = active ? rows[gi] : 0;
  for (int t = lane; t < H; t += G) tab[g][t] = EMPTY;
  if (lane == 0) cnt[g] = 0;
  wave_lds_sync();
  if (!active) return;
  const long long a0 = Aia[row], a1 = Aia[row + 1];
  constexpr unsigned mask = (1u << LOGH) - 1u;
  auto add = [&](int j, double prod) {
    unsigned sl = ((unsigned)j * 2654435761u) >> (32 - LOGH);
    while (true) {
      const int old = atomicCAS(&tab[g][sl], EMPTY, j);
      if (old == EMPTY) {
        val[g][sl] = prod;
        const int p = atomicAdd(&cnt[g], 1);
        keys[g][p] = j;
        slots[g][p] = (unsigned short)sl;
        return;
      }
      if (old == j) {
        val[g][sl] = val[g][sl] + prod;
        return;
      }
      sl = (sl + 1u) & mask;
    }
  };
  // G entries of A's row at a time: lane l fetches entry l and the bounds of its row of B (two round trips for all of
  // them together); the steps then read them by shuffle, and the loads of step t + 1 are in flight while step t adds
  for (long long base = a0; base < a1; base += G) {
    const long long kal = base + lane;
    const bool has = kal < a1;
    const int krl = has ? Aja[kal] : 0;
    const double avl = has ? Aa[kal] : 0.0;
    const long long b0l = has ? Bia[krl] : 0, b1l = has ? Bia[krl + 1] : 0;
    const int nstep = (int)((a1 - base < (long long)G) ? (a1 - base) : (long long)G);
    long long nb0 = __shfl(b0l, 0, G), nb1 = __shfl(b1l, 0, G);
    int nj = 0;
    double nbv = 0.0;
    if (nb0 + lane < nb1) {
      nj = Bja[nb0 + lane];
      nbv = Ba[nb0 + lane];
    }
    for (int t = 0; t < nstep; t++) {
      const long long b0 = nb0, b1 = nb1;
      const int j = nj;
      const double bv = nbv;
      const double av = __shfl(avl, t, G);
      if (t + 1 < nstep) {
        nb0 = __shfl(b0l, t + 1, G);
        nb1 = __shfl(b1l, t + 1, G);
        if (nb0 + lane < nb1) {
          nj = Bja[nb0 + lane];
          nbv = Ba[nb0 + lane];
        }
      }
      if (b0 + lane < b1) add(j, av * bv);
      for (long long kb = b0 + lane + G; kb < b1; kb += G) add(Bja[kb], av * Ba[kb]);
      wave_lds_sync();
    }
  }
  const int no = cnt[g];
  const long long c0 = Cia[row];
  for (int r = lane; r < no; r += G) {
    const int j = keys[g][r];
    int rank = 0;
    for (int t = 0; t < no; t++) rank += (keys[g][t] < j);
    Cja[c0 + rank] = j;
    Ca[c0 + rank] = val[g][slots[g][r]];
  }
}

// ---------------------------------------------------------------- SpGEMM, one block per (long) row
// Hash set and column list live in LDS when the row's bound fits, otherwise in this block's slice of gscratch.
constexpr int BLK_LDS_CAP = 4096;
template <bool NUMERIC>
__global__ __launch_bounds__(BLK) void spgemm_block_k(int nlist, const int *__restrict__ rows,
                                                      const int *__restrict__ T, int m, int S, int *gscratch,
                                                      long long scratch_per_block, const long long *__restrict__ Aia,
                                                      const int *__restrict__ Aja, const double *__restrict__ Aa,
                                                      const long long *__restrict__ Bia, const int *__restrict__ Bja,
                                                      const double *__restrict__ Ba, int *__restrict__ nout,
                                                      const long long *__restrict__ Cia, int *__restrict__ Cja,
                                                      double *__restrict__ Ca) {
  __shared__ int s_tab[2 * BLK_LDS_CAP];
  __shared__ int s_list[BLK_LDS_CAP];
  __shared__ int s_cnt;
  const int tid = threadIdx.x;
  for (int li = blockIdx.x; li < nlist; li += gridDim.x) {
    const int row = rows[li];
    const int bound = min(T[row], m);  // distinct output columns of this row <= bound
    int log_h = 6;
    while ((1 << log_h) < 2 * bound) log_h++;
    const int H = 1 << log_h;
    int *tab, *list;
    if (bound <= BLK_LDS_CAP) {
      tab = s_tab;
      list = s_list;
    } else {
      tab = gscratch + (long long)blockIdx.x * scratch_per_block;
      list = tab + H;
    }
    for (int t = tid; t < H; t += BLK) tab[t] = EMPTY;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const long long a0 = Aia[row], a1 = Aia[row + 1];
    {
      const int sub = tid / S, sl = tid % S, nsub = BLK / S;
      for (long long ka = a0 + sub; ka < a1; ka += nsub) {
        const int kr = Aja[ka];
        const long long b1 = Bia[kr + 1];
        for (long long kb = Bia[kr] + sl; kb < b1; kb += S) {
          const int j = Bja[kb];
          if (hs_insert(tab, log_h, j)) list[atomicAdd(&s_cnt, 1)] = j;
        }
      }
    }
    __syncthreads();
    const int no = s_cnt;
    if (!NUMERIC) {
      if (tid == 0) nout[row] = no;
    } else {
      const long long c0 = Cia[row];
      for (int r = tid; r < no; r += BLK) {
        const int j = list[r];
        int rank = 0;
        for (int t = 0; t < no; t++) rank += (list[t] < j);
        Cja[c0 + rank] = j;
        Ca[c0 + rank] = row_dot(a0, a1, Aja, Aa, Bia, Bja, Ba, j);
      }
    }
    __syncthreads();  // table and counter are reused by the next row
  }
}

// ---------------------------------------------------------------- exclusive scan (int counts -> 64-bit offsets)
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLK * SCAN_ITEMS;

__global__ __launch_bounds__(BLK) void scan_tile_k(long long n, const int *__restrict__ in, long long *__restrict__ out,
                                                   long long *__restrict__ tile_sum) {
  __shared__ long long sh[BLK];
  const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  long long v[SCAN_ITEMS], tot = 0;
  for (int q = 0; q < SCAN_ITEMS; q++) {
    v[q] = (base + q < n) ? in[base + q] : 0;
    tot += v[q];
  }
  sh[threadIdx.x] = tot;
  __syncthreads();
  for (int d = 1; d < BLK; d <<= 1) {
    const long long add = (threadIdx.x >= d) ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += add;
    __syncthreads();
  }
  long long run = sh[threadIdx.x] - tot;  // exclusive prefix of this thread inside the tile
  for (int q = 0; q < SCAN_ITEMS; q++) {
    if (base + q < n) out[base + q] = run;
    run += v[q];
  }
  if (threadIdx.x == BLK - 1) tile_sum[blockIdx.x] = sh[BLK - 1];
}

// single block: exclusive scan of the tile sums in place; total -> tile_sum[ntiles]
__global__ __launch_bounds__(BLK) void scan_sums_k(long long ntiles, long long *__restrict__ tile_sum) {
  __shared__ long long sh[BLK];
  __shared__ long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (long long b0 = 0; b0 < ntiles; b0 += BLK) {
    const long long i = b0 + threadIdx.x;
    const long long v = (i < ntiles) ? tile_sum[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < BLK; d <<= 1) {
      const long long add = (threadIdx.x >= d) ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < ntiles) tile_sum[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 0) carry += sh[BLK - 1];
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_sum[ntiles] = carry;
}

__global__ __launch_bounds__(BLK) void scan_add_k(long long n, long long *__restrict__ out,
                                                  const long long *__restrict__ tile_sum, long long ntiles) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n) out[i] += tile_sum[i / SCAN_TILE];
  if (i == n) out[n] = tile_sum[ntiles];
}

// out[0..n] = exclusive scan of in[0..n); returns nothing (out[n] holds the total on the device)
void exclusive_scan(const int *in, long long *out, long long n, hipStream_t s) {
  const long long ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  DVec<long long> sums((size_t)ntiles + 1);
  if (ntiles > 0) {
    scan_tile_k<<<(unsigned)ntiles, BLK, 0, s>>>(n, in, out, sums.p);
    scan_sums_k<<<1, BLK, 0, s>>>(ntiles, sums.p);
  } else {
    MI_HIP(hipMemsetAsync(sums.p, 0, sizeof(long long), s));
  }
  scan_add_k<<<(unsigned)((n + 1 + BLK - 1) / BLK), BLK, 0, s>>>(n, out, sums.p, ntiles);
  MI_HIP(hipStreamSynchronize(s));  // sums is released on return
}

int pow2_at_most(double v, int cap) {
  int s = 1;
  while (s * 2 <= cap && (double)(s * 2) <= v) s *= 2;
  return s;
}


// value of lane gbase + q of the wave (q a compile-time constant of an unrolled loop): a group that is the whole wave reads
// it through a scalar register (v_readlane, no LDS crossbar), smaller groups shuffle
template <int G>
__device__ __forceinline__ double group_value(double v, int gbase, int q) {
  if (G == 64) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), q);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), q);
    return __hiloint2double(hi, lo);
  }
  return __shfl(v, gbase + q, 64);
}
// bit gbase + q of a ballot
__device__ __forceinline__ bool ballot_bit(unsigned long long m, int pos) { return (m >> pos) & 1ull; }

// ---------------------------------------------------------------- strength of connection
// G lanes share one row (G = 1 for short rows: thread per row).  With one thread per row a wave touches 64 rows
// whose entries are row-length x 12 B apart: on the 30-150-entry rows of the coarse levels every load fetched a
// line of its own (832 GB read for a 1.8 GB matrix on level 1 of 512^3, profiles/r01_pmc512_fetch_write.txt).  The
// lanes of a group read consecutive entries instead; sums that the host forms in stored order (row sum) are
// accumulated in stored order here too: every lane adds the G loaded values one after the other (shuffles), so the
// result is bit-identical; the kept columns are written in stored order through a ballot prefix.
// FILL = false: count per row; FILL = true: write the kept columns (stored order)
template <bool FILL, int G>
__global__ __launch_bounds__(BLK) void strength_k(int n, const long long *__restrict__ ia, const int *__restrict__ ja,
                                                  const double *__restrict__ a, double theta, double max_row_sum,
                                                  int *__restrict__ cnt, const long long *__restrict__ sia,
                                                  int *__restrict__ sja) {
  const long long i = (bid() * BLK + threadIdx.x) / G;
  const int sub = threadIdx.x % G, lane = threadIdx.x & 63, gbase = lane - sub;
  const bool live = i < n;
  const long long k0 = live ? ia[i] : 0, k1 = live ? ia[i + 1] : 0;
  double diag = 0.0, row_sum = 0.0, scale_pos = 0.0, scale_neg = 0.0;  // max / min over the off-diagonal entries
  // the longest row of the wave decides the trip count (shuffles need every lane)
  long long len = k1 - k0;
  for (int m = G; m < 64; m <<= 1) len = max(len, (long long)__shfl_xor((int)len, m, 64));
  for (long long t = 0; t < len; t += G) {
    const long long k = k0 + t + sub;
    const bool ok = k < k1;
    const double v = ok ? a[k] : 0.0;
    const int j = ok ? ja[k] : -1;
    const bool isd = ok && j == i;
    if (ok && !isd) {
      scale_pos = fmax(scale_pos, v);
      scale_neg = fmin(scale_neg, v);
    }
    // the row sum in stored order: every lane adds the G values one after the other.  Only the value travels per step
    // (round 4: the diagonal flag travelled too, by a third cross-lane read per step); which of them is the diagonal
    // comes out of one ballot.
    const unsigned long long dmask = (G > 1) ? __ballot(isd) : 0ull;
    const int rem = (int)((k1 - k0 - t) < (long long)G ? (k1 - k0 - t) : (long long)G);  // entries of this batch
#pragma unroll
    for (int q = 0; q < G; q++) {
      const double vq = (G > 1) ? group_value<G>(v, gbase, q) : v;
      if (q < rem) {
        row_sum += vq;
        if ((G > 1) ? ballot_bit(dmask, gbase + q) : isd) diag = vq;
      }
    }
  }
  for (int m = 1; m < G; m <<= 1) {
    scale_pos = fmax(scale_pos, __shfl_xor(scale_pos, m, 64));
    scale_neg = fmin(scale_neg, __shfl_xor(scale_neg, m, 64));
  }
  const double scale = (diag < 0) ? scale_pos : scale_neg;
  const bool all_weak = (fabs(row_sum) > fabs(diag) * max_row_sum) && (max_row_sum < 1.0);
  const double thr = theta * scale;
  int c = 0;
  long long q0 = (FILL && live) ? sia[i] : 0;
  for (long long t = 0; t < len; t += G) {
    const long long k = k0 + t + sub;
    const bool ok = k < k1 && !all_weak;
    bool strong = false;
    int j = -1;
    if (ok) {
      j = ja[k];
      const double v = a[k];
      strong = (j != i) && ((diag < 0) ? (v > thr) : (v < thr));
    }
    if (G == 1) {
      if (strong) {
        if (FILL) sja[q0] = j;
        q0++;
        c++;
      }
    } else {
      const unsigned long long bal = __ballot(strong);
      const unsigned long long mine = (G == 64) ? bal : ((bal >> gbase) & ((1ull << G) - 1ull));
      if (FILL && strong) sja[q0 + __popcll(mine & ((1ull << sub) - 1ull))] = j;
      const int add = __popcll(mine);
      q0 += add;
      c += add;
    }
  }
  if (!FILL && live && sub == 0) cnt[i] = c;
}

// ---------------------------------------------------------------- PMIS
constexpr int C_PT = 1, F_PT = -1, SF_PT = -3;

__device__ __forceinline__ unsigned long long mulmod31(unsigned long long x, unsigned long long y) {
  return (x * y) % 2147483647ULL;  // both < 2^31: the product fits 64 bits
}

__global__ __launch_bounds__(BLK) void pmis_init_k(int n, const long long *__restrict__ sia,
                                                   const int *__restrict__ incoming, int seed0,
                                                   double *__restrict__ measure, int *__restrict__ cf,
                                                   int *__restrict__ undecided) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n) return;
  // element i of the Park-Miller stream: seed0 * 16807^(i+1) mod (2^31 - 1)
  unsigned long long base = 16807ULL, acc = (unsigned long long)seed0, e = (unsigned long long)i + 1ULL;
  while (e) {
    if (e & 1ULL) acc = mulmod31(acc, base);
    base = mulmod31(base, base);
    e >>= 1;
  }
  const double m = (double)incoming[i] + (double)(int)acc / 2147483647;
  int c = 0;
  if (sia[i + 1] == sia[i])
    c = SF_PT;
  else if (m < 1.0)
    c = F_PT;
  cf[i] = c;
  measure[i] = c ? 0.0 : m;
  if (c == 0) atomicAdd(undecided, 1);
}

__global__ __launch_bounds__(BLK) void pmis_mark_k(int n, const int *__restrict__ cf, signed char *__restrict__ tmp) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n) tmp[i] = (cf[i] == 0);
}

__global__ __launch_bounds__(BLK) void pmis_compare_k(int n, const long long *__restrict__ sia,
                                                      const int *__restrict__ sja, const int *__restrict__ cf,
                                                      const double *__restrict__ measure,
                                                      signed char *__restrict__ tmp) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n || cf[i] != 0) return;
  const double mi = measure[i];
  bool lose = false;
  for (long long k = sia[i]; k < sia[i + 1]; k++) {
    const int j = sja[k];
    if (cf[j] != 0) continue;
    const double mj = measure[j];
    if (mi > mj)
      tmp[j] = 0;
    else if (mj > mi)
      lose = true;
  }
  if (lose) tmp[i] = 0;
}

__global__ __launch_bounds__(BLK) void pmis_select_k(int n, int *__restrict__ cf, const signed char *__restrict__ tmp) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n && cf[i] == 0 && tmp[i]) cf[i] = C_PT;
}

// undecided rows that depend on a C point become F; the others are counted for the next round
__global__ __launch_bounds__(BLK) void pmis_fpoints_k(int n, const long long *__restrict__ sia,
                                                      const int *__restrict__ sja, int *__restrict__ cf,
                                                      int *__restrict__ undecided) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n || cf[i] != 0) return;
  for (long long k = sia[i]; k < sia[i + 1]; k++)
    if (cf[sja[k]] == C_PT) {  // other rows only ever switch 0 -> F here, never to or from C
      cf[i] = F_PT;
      return;
    }
  atomicAdd(undecided, 1);
}

// ---------------------------------------------------------------- interpolation (ext+i / classical modified)
// One group of G lanes per row.  The interpolatory set of an F row (strong C neighbours, plus for ext+i the
// strong C neighbours of its strong F neighbours) is collected in an LDS hash keyed by the fine id with the
// DISCOVERY position of the host loop as value (atomicMin), ranked by that position, and from then on lane q
// owns entry q: it walks A's row in stored order and adds the same terms in the same order as the host loop.
constexpr int INTERP_SEQ = 65536;

// bound on the interpolatory set (T) and on the stored row of P (cap); bins on max(T, |S row|)
__global__ __launch_bounds__(BLK) void interp_bound_k(int n, const long long *__restrict__ sia,
                                                      const int *__restrict__ sja, const int *__restrict__ cf,
                                                      int ext, int pmax, int *__restrict__ T, int *__restrict__ cap,
                                                      int *__restrict__ is_c, int *__restrict__ bin_count,
                                                      int *__restrict__ tmax) {
  __shared__ int hist[4];
  __shared__ int smax;
  if (threadIdx.x < 4) hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) smax = 0;
  __syncthreads();
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n) {
    const int c = cf[i];
    long long t = 0;
    int cp = 0;
    if (c == C_PT) {
      cp = 1;
    } else if (c == F_PT) {
      long long trc = 0;
      for (long long k = sia[i]; k < sia[i + 1]; k++) {
        const int i1 = sja[k];
        const int c1 = cf[i1];
        if (c1 == C_PT) {
          trc++;
        } else if (c1 == F_PT && ext) {
          // only the C points of a strong F neighbour's row are candidates (counting the whole row put every row of
          // a 27-point operator -- 24 strong F neighbours x 26 -- into the one-workgroup-per-row bin: 19.6 s of the
          // 512^3 27-point setup)
          for (long long kk = sia[i1]; kk < sia[i1 + 1]; kk++) trc += (cf[sja[kk]] == C_PT);
        }
      }
      t = max(trc, sia[i + 1] - sia[i]);
      const long long rowcap = pmax > 0 ? min(trc, (long long)pmax) : trc;
      cp = rowcap > 0x3fffffff ? 0x3fffffff : (int)rowcap;
    }
    const int ti = t > 0x3fffffff ? 0x3fffffff : (int)t;
    T[i] = ti;
    cap[i] = cp;
    is_c[i] = (c == C_PT);
    const int b = ti <= 32 ? 0 : ti <= 128 ? 1 : ti <= 512 ? 2 : 3;
    atomicAdd(&hist[b], 1);
    atomicMax(&smax, ti);
  }
  __syncthreads();
  if (threadIdx.x < 4 && hist[threadIdx.x]) atomicAdd(&bin_count[threadIdx.x], hist[threadIdx.x]);
  if (threadIdx.x == 0) atomicMax(tmax, smax);
}

template <int CAP>
struct InterpGroup {
  int hkey[2 * CAP];
  int hval[2 * CAP];  // discovery position, later the entry's index q
  union {
    struct {
      int ckey[CAP];
      int cseq[CAP];  // discovery positions of the compacted entries, later the keep flags
    };
    double sfback[CAP];  // between the ranking and the truncation: per strong neighbour, its entry in column i
  };
  int ord[CAP];   // fine ids in discovery order
  double rv[CAP];
  double sfsum[CAP];
  signed char sfsgn[CAP];
  int cnt, nkept, nkeys;  // nkeys: distinct candidates so far (TRY instantiations only)
  double diag, scale;
};

template <int LOGH>
__device__ __forceinline__ int ig_find(const int *hkey, int key) {
  const unsigned mask = (1u << LOGH) - 1u;
  unsigned s = ((unsigned)key * 2654435761u) >> (32 - LOGH);
  while (true) {
    const int kk = hkey[s];
    if (kk == key) return (int)s;
    if (kk == EMPTY) return -1;
    s = (s + 1u) & mask;
  }
}

// -DMI_INTERP_STOP=<phase>: timing ablation, the kernel returns after that phase with empty rows (never in a shipped build)
#ifndef MI_INTERP_STOP
#define MI_INTERP_STOP 0
#endif
#define INTERP_STOP_AFTER(P)                       \
  if (MI_INTERP_STOP == (P)) {                     \
    if (work && lane == 0) len_out[i] = 0;         \
    return;                                        \
  }
// TRY: the tables are SMALLER than the row's bound (the bound counts candidates with their multiplicity -- 60-100 for a
// row of a 27-point operator that ends with ~20 distinct ones -- and LDS per row is what limits the rows in flight): the
// row is given up as soon as it turns out not to fit (more than CAP distinct candidates, or more than CAP strong
// connections) and marked len_out = -1; the caller runs the marked rows through the instantiation sized by the bound.
// A row that fits gives the same result either way (nothing below depends on the table size).
template <int G, int CAP, int BT, bool TRY = false>
__global__ __launch_bounds__(BT) void interp_group_k(int nlist, const int *__restrict__ rows, int ext,
                                                     const long long *__restrict__ Aia, const int *__restrict__ Aja,
                                                     const double *__restrict__ Aa, const long long *__restrict__ Sia,
                                                     const int *__restrict__ Sja, const int *__restrict__ cf,
                                                     const long long *__restrict__ f2c, double trunc_factor, int pmax,
                                                     const long long *__restrict__ slack_ia, int *__restrict__ Pj,
                                                     double *__restrict__ Pa, int *__restrict__ len_out) {
  constexpr int H = 2 * CAP;
  constexpr int LOGH = (H == 32) ? 5 : (H == 64) ? 6 : (H == 256) ? 8 : (H == 1024) ? 10 : 11;
  static_assert(H == 32 || H == 64 || H == 256 || H == 1024 || H == 2048, "table size");
  constexpr int GP = BT / G;
  __shared__ InterpGroup<CAP> grp[GP];
  InterpGroup<CAP> &L = grp[threadIdx.x / G];
  const int lane = threadIdx.x % G;
  const long long gi = bid() * GP + threadIdx.x / G;
  const bool active = gi < nlist;
  const int i = active ? rows[gi] : 0;
  const int mycf = active ? cf[i] : SF_PT;
  const bool work = active && mycf == F_PT;
  // trivial rows: C point -> (coarse id, 1.0); F without strong connections -> empty
  if (active && !work && lane == 0) {
    if (mycf == C_PT) {
      const long long o = slack_ia[i];
      Pj[o] = (int)f2c[i];
      Pa[o] = 1.0;
      len_out[i] = 1;
    } else {
      len_out[i] = 0;
    }
  }
  for (int t = lane; t < H; t += G) {
    L.hkey[t] = EMPTY;
    L.hval[t] = 0x7fffffff;
  }
  if (lane == 0) L.cnt = 0, L.nkeys = 0;
  group_sync<G>();
  const long long s0 = work ? Sia[i] : 0, s1 = work ? Sia[i + 1] : 0;
  const bool too_long = TRY && (s1 - s0 > (long long)CAP);  // the per-connection arrays hold CAP entries
  // ---- 1. interpolatory set with discovery positions
  if (work && !too_long) {
    for (long long k = s0 + lane; k < s1; k += G) {
      const int i1 = Sja[k];
      const int c1 = cf[i1];
      const int kl = (int)(k - s0);
      auto put = [&](int key, int seq) {
        const unsigned mask = (1u << LOGH) - 1u;
        unsigned sl = ((unsigned)key * 2654435761u) >> (32 - LOGH);
        if (TRY) {
          // at most CAP + G keys ever enter the table of 2 CAP slots (every lane looks at the count before it inserts),
          // so the probe below always meets the key or a free slot
          if (*(volatile int *)&L.nkeys > CAP) return;
        }
        while (true) {
          const int old = atomicCAS(&L.hkey[sl], EMPTY, key);
          if (old == EMPTY) {
            if (TRY) atomicAdd(&L.nkeys, 1);
            break;
          }
          if (old == key) break;
          sl = (sl + 1u) & mask;
        }
        atomicMin(&L.hval[sl], seq);
      };
      if (c1 == C_PT) {
        put(i1, kl * INTERP_SEQ);
      } else if (c1 == F_PT && ext) {
        // (four entries per round trip: ids together, then their marks together -- one entry at a time this walk was a
        // chain of 2 x |row| dependent loads)
        const long long t0 = Sia[i1], t1 = Sia[i1 + 1];
        for (long long kk = t0; kk < t1; kk += 4) {
          int k1[4], m1[4];
#pragma unroll
          for (int u = 0; u < 4; u++) k1[u] = (kk + u < t1) ? Sja[kk + u] : -1;
#pragma unroll
          for (int u = 0; u < 4; u++) m1[u] = (k1[u] >= 0) ? cf[k1[u]] : 0;
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (k1[u] >= 0 && m1[u] == C_PT) put(k1[u], kl * INTERP_SEQ + 1 + (int)(kk + u - t0));
        }
      }
    }
  }
  group_sync<G>();
  if (TRY && work && (too_long || L.nkeys > CAP)) {  // does not fit: the caller's second launch takes the row
    if (lane == 0) len_out[i] = -1;
    return;
  }
  INTERP_STOP_AFTER(1)
  // ---- 2. compaction
  if (work)
    for (int t = lane; t < H; t += G)
      if (L.hkey[t] != EMPTY) {
        const int p = atomicAdd(&L.cnt, 1);
        L.ckey[p] = L.hkey[t];
        L.cseq[p] = L.hval[t];
      }
  group_sync<G>();
  const int len = work ? L.cnt : 0;
  // ---- 3. discovery order: ord[rank] = key, hash value = rank
  if (work)
    for (int p = lane; p < len; p += G) {
      const int sq = L.cseq[p];
      int rank = 0;
      for (int u = 0; u < len; u++) rank += (L.cseq[u] < sq);
      L.ord[rank] = L.ckey[p];
      L.hval[ig_find<LOGH>(L.hkey, L.ckey[p])] = rank;
    }
  group_sync<G>();
  INTERP_STOP_AFTER(3)
  // ---- 4. per strong F neighbour: sign of its diagonal and the sum its connection is distributed over
  if (work)
    for (long long k = s0 + lane; k < s1; k += G) {
      const int i1 = Sja[k];
      const int kl = (int)(k - s0);
      double sum = 0.0, back = 0.0;
      signed char sg = 1;
      if (cf[i1] == F_PT) {
        const long long r0 = Aia[i1], r1 = Aia[i1 + 1];
        // its diagonal: rows are stored with ascending columns (the searches below rely on it as well)
        double dk = 0.0;
        {
          long long lo = r0, hi = r1;
          while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (Aja[mid] < i1)
              lo = mid + 1;
            else
              hi = mid;
          }
          if (lo < r1 && Aja[lo] == i1) dk = Aa[lo];
        }
        const double sgn = (dk < 0) ? -1.0 : 1.0;
        sg = (dk < 0) ? -1 : 1;
        // the row once, four entries per round trip, summed in stored order.  (The table holds C points only, so being
        // in it is the whole test; it used to be preceded by a load of the entry's mark.)
        for (long long kk = r0; kk < r1; kk += 4) {
          int c2[4];
          double w2[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            c2[u] = (kk + u < r1) ? Aja[kk + u] : i1;
            w2[u] = (kk + u < r1) ? Aa[kk + u] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int i2 = c2[u];
            if (i2 == i1) continue;
            const double v = w2[u];
            // the neighbour's entry in column i, for the accumulation of the diagonal (which used to walk this row again,
            // one neighbour after the other: the longest dependent chain of the kernel); magnitude 2 = present
            if (i2 == i) {
              back = v;
              sg = (signed char)(2 * sg);
            }
            if (!(sgn * v < 0)) continue;
            if ((ext && i2 == i) || ig_find<LOGH>(L.hkey, i2) >= 0) sum += v;
          }
        }
      }
      L.sfsum[kl] = sum;
      L.sfsgn[kl] = sg;
      L.sfback[kl] = back;
    }
  group_sync<G>();
  INTERP_STOP_AFTER(4)
  // ---- 5. weights.  The group walks row i ONCE, in stored order, one entry per step (round 4; before, every lane q walked
  // the row for its own entry and searched each strong F neighbour's row for it, and one lane did the same for the
  // diagonal): an interpolatory point adds its coefficient to its own weight; a strong F neighbour's row is spread over
  // the lanes, and every entry of it that is an interpolatory point of the right sign adds distribute * value to that
  // point's weight -- per weight at most one term per step, so each weight still sums its terms in the order of row i,
  // with the same operations as the host loop.  The diagonal's terms are lane 0's, in the same order.  G entries of the
  // row are fetched at a time (lane l: entry l, its mark, its position among the strong connections, the bounds of its
  // row); the steps read them by shuffle, and the first entries of the next step's row are in flight during a step.
  if constexpr (G <= 64) {
    if (work)
      for (int q = lane; q < len; q += G) L.rv[q] = 0.0;
    if (work && lane == 0) L.diag = 0.0;
    group_sync<G>();
    if (work) {
      const long long r0 = Aia[i], r1 = Aia[i + 1];
      // the diagonal's sum starts from a_ii (host loop): fetched before the walk
      for (long long k = r0 + lane; k < r1; k += G)
        if (Aja[k] == i) L.diag = Aa[k];
      group_sync<G>();
      double diagonal = L.diag;
      for (long long base = r0; base < r1; base += G) {
        const long long kal = base + lane;
        const bool has = kal < r1;
        const int i1l = has ? Aja[kal] : i;
        const double al = has ? Aa[kal] : 0.0;
        const int c1l = (has && i1l != i) ? cf[i1l] : 0;
        int kll = -1;  // position among the strong connections (ascending columns in both rows)
        long long b0l = 0, b1l = 0;
        if (has && i1l != i) {
          long long lo = s0, hi = s1;
          while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (Sja[mid] < i1l)
              lo = mid + 1;
            else
              hi = mid;
          }
          if (lo < s1 && Sja[lo] == i1l) kll = (int)(lo - s0);
          if (kll >= 0 && c1l == F_PT) {
            b0l = Aia[i1l];
            b1l = Aia[i1l + 1];
          }
        }
        const int nstep = (int)((r1 - base < (long long)G) ? (r1 - base) : (long long)G);
        long long nb0 = __shfl(b0l, 0, G), nb1 = __shfl(b1l, 0, G);
        int nj = 0;
        double nv = 0.0;
        if (nb0 + lane < nb1) {
          nj = Aja[nb0 + lane];
          nv = Aa[nb0 + lane];
        }
        for (int t = 0; t < nstep; t++) {
          const long long b0 = nb0, b1 = nb1;
          const int j0 = nj;
          const double v0 = nv;
          const int i1 = __shfl(i1l, t, G);
          const double aik = __shfl(al, t, G);
          const int c1 = __shfl(c1l, t, G);
          const int kl = __shfl(kll, t, G);
          if (t + 1 < nstep) {
            nb0 = __shfl(b0l, t + 1, G);
            nb1 = __shfl(b1l, t + 1, G);
            if (nb0 + lane < nb1) {
              nj = Aja[nb0 + lane];
              nv = Aa[nb0 + lane];
            }
          }
          if (i1 != i) {
            const int slot = (c1 == C_PT) ? ig_find<LOGH>(L.hkey, i1) : -1;
            if (slot >= 0) {  // interpolatory point
              if (lane == 0) {
                const int q = L.hval[slot];
                L.rv[q] = L.rv[q] + aik;
              }
            } else if (kl >= 0 && c1 == F_PT) {
              const double sum = L.sfsum[kl];
              if (sum != 0.0) {
                const double distribute = aik / sum;
                const int sg = L.sfsgn[kl];
                auto spread = [&](int i2, double v) {
                  if (!((sg < 0 ? -v : v) < 0)) return;
                  const int sl2 = ig_find<LOGH>(L.hkey, i2);
                  if (sl2 >= 0) {
                    const int q = L.hval[sl2];
                    L.rv[q] = L.rv[q] + distribute * v;
                  }
                };
                if (b0 + lane < b1) spread(j0, v0);
                for (long long kk = b0 + lane + G; kk < b1; kk += G) spread(Aja[kk], Aa[kk]);
                if (lane == 0 && ext && (sg == 2 || sg == -2)) {
                  const double v = L.sfback[kl];
                  if ((sg < 0 ? -v : v) < 0) diagonal += distribute * v;
                }
              } else if (lane == 0) {
                diagonal += aik;
              }
            } else if (lane == 0) {
              diagonal += aik;
            }
          }
          group_sync<G>();
        }
      }
      if (lane == 0) L.diag = diagonal;
    }
  } else {
    // one row per workgroup (up to 1024 candidates; groups wider than a wave cannot pass the row by shuffle): lane q
    // accumulates entry q by its own walk over row i, the last lane the diagonal
    if (work) {
      const long long r0 = Aia[i], r1 = Aia[i + 1];
      for (int q = lane; q < len; q += G) {
        const int myid = L.ord[q];
        double acc = 0.0;
        long long sp = s0;
        for (long long k = r0; k < r1; k++) {
          const int i1 = Aja[k];
          const bool strong = (sp < s1 && Sja[sp] == i1);
          const int kl = (int)(sp - s0);
          if (strong) sp++;
          if (i1 == i) continue;
          const double aik = Aa[k];
          if (i1 == myid) {
            acc += aik;
          } else if (strong && cf[i1] == F_PT) {
            const double sum = L.sfsum[kl];
            if (sum != 0.0) {
              const double distribute = aik / sum;
              long long lo = Aia[i1];
              const long long end = Aia[i1 + 1];
              long long hi = end;
              while (lo < hi) {
                const long long mid = (lo + hi) >> 1;
                if (Aja[mid] < myid)
                  lo = mid + 1;
                else
                  hi = mid;
              }
              if (lo < end && Aja[lo] == myid) {
                const double v = Aa[lo];
                if ((L.sfsgn[kl] < 0 ? -v : v) < 0) acc += distribute * v;
              }
            }
          }
        }
        L.rv[q] = acc;
      }
      if (lane == G - 1) {
        double diagonal = 0.0;
        for (long long k = r0; k < r1; k++)
          if (Aja[k] == i) diagonal = Aa[k];
        long long sp = s0;
        for (long long k = r0; k < r1; k++) {
          const int i1 = Aja[k];
          const bool strong = (sp < s1 && Sja[sp] == i1);
          const int kl = (int)(sp - s0);
          if (strong) sp++;
          if (i1 == i) continue;
          const double aik = Aa[k];
          const int c1 = cf[i1];
          if (c1 == C_PT && ig_find<LOGH>(L.hkey, i1) >= 0) continue;  // interpolatory point
          if (strong && c1 == F_PT) {
            const double sum = L.sfsum[kl];
            if (sum != 0.0) {
              if (ext) {
                const double distribute = aik / sum;
                const int sg = L.sfsgn[kl];
                if (sg == 2 || sg == -2) {
                  const double v = L.sfback[kl];
                  if ((sg < 0 ? -v : v) < 0) diagonal += distribute * v;
                }
              }
            } else {
              diagonal += aik;
            }
          } else {
            diagonal += aik;
          }
        }
        L.diag = diagonal;
      }
    }
  }
  group_sync<G>();
  if (work) {
    const double diagonal = L.diag;
    if (diagonal != 0.0)
      for (int q = lane; q < len; q += G) L.rv[q] = L.rv[q] / -diagonal;
  }
  group_sync<G>();
  INTERP_STOP_AFTER(5)
  // ---- 6. truncation: keep the pmax largest by (|p| descending, position ascending) among those >= factor*max
  if (work) {
    double maxabs = 0.0;
    if (trunc_factor > 0.0)
      for (int t = 0; t < len; t++) maxabs = fmax(maxabs, fabs(L.rv[t]));
    for (int q = lane; q < len; q += G) {
      const double aq = fabs(L.rv[q]);
      bool keep = (trunc_factor > 0.0) ? (aq >= trunc_factor * maxabs) : true;
      if (keep && pmax > 0) {
        int rank = 0;
        for (int t = 0; t < len; t++) {
          const double at = fabs(L.rv[t]);
          const bool kt = (trunc_factor > 0.0) ? (at >= trunc_factor * maxabs) : true;
          rank += (kt && (at > aq || (at == aq && t < q)));
        }
        keep = rank < pmax;
      }
      L.cseq[q] = keep;
    }
  }
  group_sync<G>();
  if (work && lane == 0) {
    double row_sum = 0.0, kept = 0.0;
    int nk = 0;
    for (int t = 0; t < len; t++) {
      row_sum += L.rv[t];
      if (L.cseq[t]) {
        kept += L.rv[t];
        nk++;
      }
    }
    L.scale = (kept != 0.0) ? row_sum / kept : 1.0;
    L.nkept = nk;
    len_out[i] = nk;
  }
  group_sync<G>();
  if (work) {
    const long long o = slack_ia[i];
    const double scale = L.scale;
    for (int q = lane; q < len; q += G) {
      if (!L.cseq[q]) continue;
      const int id = L.ord[q];
      int pos = 0;
      for (int t = 0; t < len; t++) pos += (L.cseq[t] && L.ord[t] < id);
      Pj[o + pos] = (int)f2c[id];
      Pa[o + pos] = L.rv[q] * scale;
    }
  }
}

__global__ __launch_bounds__(BLK) void compact_rows_k(int n, const long long *__restrict__ slack_ia,
                                                      const int *__restrict__ sj, const double *__restrict__ sa,
                                                      const long long *__restrict__ ia, int *__restrict__ dj,
                                                      double *__restrict__ da) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n) return;
  const long long s0 = slack_ia[i], d0 = ia[i];
  const int len = (int)(ia[i + 1] - d0);
  for (int e = 0; e < len; e++) {
    dj[d0 + e] = sj[s0 + e];
    da[d0 + e] = sa[s0 + e];
  }
}

__global__ __launch_bounds__(BLK) void sf_to_f_k(int n, int *__restrict__ cf) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n && cf[i] == SF_PT) cf[i] = F_PT;
}

// ---------------------------------------------------------------- row sort / transpose / permute
// dst row = src row sorted by column (columns are unique inside a row); group of G lanes per row
template <int G>
__global__ __launch_bounds__(BLK) void sort_rows_k(int n, const long long *__restrict__ ia, const int *__restrict__ sj,
                                                   const double *__restrict__ sa, int *__restrict__ dj,
                                                   double *__restrict__ da) {
  const long long row = (bid() * BLK + threadIdx.x) / G;
  const int lane = threadIdx.x % G;
  if (row >= n) return;
  const long long b = ia[row];
  const int len = (int)(ia[row + 1] - b);
  for (int e = lane; e < len; e += G) {
    const int c = sj[b + e];
    int rank = 0;
    for (int t = 0; t < len; t++) rank += (sj[b + t] < c);
    dj[b + rank] = c;
    da[b + rank] = sa[b + e];
  }
}

void sort_rows(int n, int64_t nnz, const long long *ia, const int *sj, const double *sa, int *dj, double *da,
               hipStream_t s) {
  if (n == 0 || nnz == 0) return;
  const double avg = (double)nnz / (double)n;
  const long long threads_needed = (long long)n * (avg <= 16 ? 8 : avg <= 48 ? 16 : 64);
  const dim3 grid = grid_for((threads_needed + BLK - 1) / BLK);
  if (avg <= 16)
    sort_rows_k<8><<<grid, BLK, 0, s>>>(n, ia, sj, sa, dj, da);
  else if (avg <= 48)
    sort_rows_k<16><<<grid, BLK, 0, s>>>(n, ia, sj, sa, dj, da);
  else
    sort_rows_k<64><<<grid, BLK, 0, s>>>(n, ia, sj, sa, dj, da);
}

__global__ __launch_bounds__(BLK) void col_count_k(long long nnz, const int *__restrict__ ja, int *__restrict__ cnt) {
  const long long k = bid() * BLK + threadIdx.x;
  if (k < nnz) atomicAdd(&cnt[ja[k]], 1);
}

__global__ __launch_bounds__(BLK) void transpose_fill_k(int n, const long long *__restrict__ ia,
                                                        const int *__restrict__ ja, const double *__restrict__ a,
                                                        const long long *__restrict__ tia, int *__restrict__ cursor,
                                                        int *__restrict__ tj, double *__restrict__ ta) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n) return;
  for (long long k = ia[i]; k < ia[i + 1]; k++) {
    const int j = ja[k];
    const long long p = tia[j] + atomicAdd(&cursor[j], 1);
    tj[p] = (int)i;
    ta[p] = a[k];
  }
}

__global__ __launch_bounds__(BLK) void perm_len_k(int n, const long long *__restrict__ ia, const int *__restrict__ perm,
                                                  int *__restrict__ len) {
  const long long q = bid() * BLK + threadIdx.x;
  if (q >= n) return;
  const int i = perm ? perm[q] : (int)q;
  len[q] = (int)(ia[i + 1] - ia[i]);
}

__global__ __launch_bounds__(BLK) void perm_copy_k(int n, const long long *__restrict__ ia, const int *__restrict__ ja,
                                                   const double *__restrict__ a, const int *__restrict__ perm,
                                                   const int *__restrict__ colpos, const long long *__restrict__ bia,
                                                   int *__restrict__ bj, double *__restrict__ ba) {
  const long long q = (bid() * BLK + threadIdx.x) / 8;
  const int lane = threadIdx.x % 8;
  if (q >= n) return;
  const int i = perm ? perm[q] : (int)q;
  const long long s0 = ia[i], d0 = bia[q];
  const int len = (int)(ia[i + 1] - s0);
  for (int e = lane; e < len; e += 8) {
    const int c = ja[s0 + e];
    bj[d0 + e] = colpos ? colpos[c] : c;
    ba[d0 + e] = a[s0 + e];
  }
}

// ---------------------------------------------------------------- solve-phase format on the device
__global__ __launch_bounds__(BLK) void ia_to_32_k(long long n1, const long long *__restrict__ ia, int *__restrict__ ia32) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n1) ia32[i] = (int)ia[i];
}
__global__ __launch_bounds__(BLK) void ia_to_64_k(long long n1, const int *__restrict__ ia32, long long *__restrict__ ia) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n1) ia[i] = ia32[i];
}

// ---- internal locality numbering (amg_setup.cpp: graph Voronoi cells): one round -- every unlabelled row (-1) takes
// the smallest label among its neighbours labelled in the previous round; labelled and excluded (-2) rows are copied
__global__ __launch_bounds__(BLK) void locality_round_k(int n, const long long *__restrict__ ia, const int *__restrict__ ja,
                                                        const int *__restrict__ in, int *__restrict__ out,
                                                        int *__restrict__ changed, int segshift) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  int li = in[i];
  if (li == -1) {
    int m = -1;
    const int sg = i >> segshift;
    const long long k1 = ia[i + 1];
    for (long long k = ia[i]; k < k1; k++) {
      const int j = ja[k];
      if ((j >> segshift) != sg) continue;  // cells do not cross segments
      const int lj = in[j];
      if (lj >= 0 && (m < 0 || lj < m)) m = lj;
    }
    if (m >= 0) {
      li = m;
      *changed = 1;  // every writer stores the same value
    }
  }
  out[i] = li;
}
__global__ __launch_bounds__(BLK) void locality_seed_k(int nseeds, const int *__restrict__ seeds, int *__restrict__ label) {
  const int k = blockIdx.x * BLK + threadIdx.x;
  if (k < nseeds) label[seeds[k]] = k;
}
__global__ __launch_bounds__(BLK) void invert_perm_k(int n, const int *__restrict__ order, int *__restrict__ pos) {
  const int q = blockIdx.x * BLK + threadIdx.x;
  if (q < n) pos[order[q]] = q;
}

// ---- non-Galerkin sparsification (one lane per row: a side-line of the setup, the arithmetic is the host's)
// (row0: the column of row 0's diagonal entry -- 0 for a square operator, the number of remote ids below the own range
// for a rank's rows in an extended column space; m is indexed by column)
__global__ __launch_bounds__(BLK) void ng_rowmax_k(int n, int row0, const long long *__restrict__ ia, const int *__restrict__ ja,
                                                   const double *__restrict__ a, double *__restrict__ m) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  double mx = 0.0;
  for (long long k = ia[i]; k < ia[i + 1]; k++)
    if (ja[k] != i + row0 && fabs(a[k]) > mx) mx = fabs(a[k]);
  m[i + row0] = mx;
}
template <bool FILL>
__global__ __launch_bounds__(BLK) void ng_drop_k(int n, int row0, const long long *__restrict__ ia, const int *__restrict__ ja,
                                                 const double *__restrict__ a, const double *__restrict__ m, double tol,
                                                 int *__restrict__ cnt, const long long *__restrict__ oia,
                                                 int *__restrict__ oja, double *__restrict__ oa) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  const int di = i + row0;
  const double mi = m[di];
  long long w = FILL ? oia[i] : 0, dpos = -1;
  int c = 0;
  double lump = 0.0;
  bool first = true;
  for (long long k = ia[i]; k < ia[i + 1]; k++) {
    const int j = ja[k];
    const double mj = m[j];
    const double lim = tol * (mi < mj ? mi : mj);
    if (j == di || !(fabs(a[k]) < lim)) {
      if (FILL) {
        if (j == di) dpos = w;
        oja[w] = j;
        oa[w++] = a[k];
      }
      c++;
    } else if (FILL) {
      lump = first ? a[k] : lump + a[k];
      first = false;
    }
  }
  if (FILL) {
    if (!first && dpos >= 0) oa[dpos] = oa[dpos] + lump;
  } else {
    cnt[i] = c;
  }
}

constexpr int LEN_BINS = 4096;
// every thread walks RUN consecutive rows and merges equal neighbours before it touches the (LDS) histogram:
// uniform row lengths (stencil matrices) would otherwise serialise on one counter
constexpr int HIST_RUN = 32;
__global__ __launch_bounds__(BLK) void rowlen_hist_k(int n, const long long *__restrict__ ia, int *__restrict__ hist) {
  __shared__ int sh[LEN_BINS];
  for (int t = threadIdx.x; t < LEN_BINS; t += BLK) sh[t] = 0;
  __syncthreads();
  const long long r0 = (bid() * BLK + threadIdx.x) * HIST_RUN;
  int cur = -1, cnt = 0;
  for (long long i = r0; i < r0 + HIST_RUN && i < n; i++) {
    const long long len = ia[i + 1] - ia[i];
    const int b = len >= LEN_BINS ? LEN_BINS - 1 : (int)len;
    if (b == cur) {
      cnt++;
    } else {
      if (cnt) atomicAdd(&sh[cur], cnt);
      cur = b;
      cnt = 1;
    }
  }
  if (cnt) atomicAdd(&sh[cur], cnt);
  __syncthreads();
  for (int t = threadIdx.x; t < LEN_BINS; t += BLK)
    if (sh[t]) atomicAdd(&hist[t], sh[t]);
}

// x cache of one row block (spmv_stream_xc): sorted unique columns of the block's entries and the
// block-local id of every entry.  One workgroup per block, bitonic sort in LDS.
template <int TILE>  // k::SPMV_TILE or k::SPMV_TILE_WIDE
__global__ __launch_bounds__(BLK) void xcache_block_k(int nb, const int *__restrict__ rb, const long long *__restrict__ ia,
                                                      const int *__restrict__ ja, int *__restrict__ ucnt,
                                                      int *__restrict__ uslack, unsigned short *__restrict__ lcol) {
  __shared__ int key[TILE];
  __shared__ int uniq[TILE];
  __shared__ int wsum[BLK];
  const int b = blockIdx.x;
  if (b >= nb) return;
  const int tid = threadIdx.x;
  const long long s0 = ia[rb[b]], e0 = ia[rb[b + 1]];
  const int cnt = (int)((e0 - s0) > (long long)TILE ? (long long)TILE : (e0 - s0));
  if (cnt >= TILE) {  // a single long row: direct gathers in the SpMV
    if (tid == 0) ucnt[b] = 0;
    return;
  }
  for (int t = tid; t < TILE; t += BLK) key[t] = (t < cnt) ? ja[s0 + t] : EMPTY;
  __syncthreads();
  for (int size = 2; size <= TILE; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = tid; t < TILE / 2; t += BLK) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const int a = key[lo], c = key[hi];
        if ((a > c) == up) {
          key[lo] = c;
          key[hi] = a;
        }
      }
      __syncthreads();
    }
  // unique: every thread owns TILE/BLK consecutive sorted keys
  constexpr int PER = TILE / BLK;
  int flags[PER], local = 0;
  for (int q = 0; q < PER; q++) {
    const int t = tid * PER + q;
    const int v = key[t];
    flags[q] = (v != EMPTY) && (t == 0 || key[t - 1] != v);
    local += flags[q];
  }
  wsum[tid] = local;
  __syncthreads();
  for (int d = 1; d < BLK; d <<= 1) {
    const int add = (tid >= d) ? wsum[tid - d] : 0;
    __syncthreads();
    wsum[tid] += add;
    __syncthreads();
  }
  int pos = wsum[tid] - local;
  for (int q = 0; q < PER; q++)
    if (flags[q]) {
      const int v = key[tid * PER + q];
      uniq[pos] = v;
      uslack[s0 + pos] = v;
      pos++;
    }
  const int nu = wsum[BLK - 1];
  __syncthreads();
  if (tid == 0) ucnt[b] = nu;
  for (int t = tid; t < cnt; t += BLK) {
    const int c = ja[s0 + t];
    int lo = 0, hi = nu;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (uniq[mid] < c)
        lo = mid + 1;
      else
        hi = mid;
    }
    lcol[s0 + t] = (unsigned short)lo;
  }
}

// in-chunk code bits of the lcol entries (tile Gauss-Seidel kernel): thread per row
__global__ __launch_bounds__(BLK) void lcol_code_k(int n, const long long *__restrict__ ia, const int *__restrict__ ja,
                                                   unsigned short *__restrict__ lcol) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n) return;
  for (long long q = ia[i]; q < ia[i + 1]; q++) {
    const int j = ja[q];
    if ((j >> 3) == (int)(i >> 3)) lcol[q] |= (unsigned short)(k::XC_INCH | ((j & 7) << k::XC_OFF_SHIFT));
  }
}

__global__ __launch_bounds__(BLK) void xcache_compact_k(int nb, const int *__restrict__ rb, const long long *__restrict__ ia,
                                                        const long long *__restrict__ uptr64,
                                                        const int *__restrict__ uslack, int *__restrict__ uptr,
                                                        int *__restrict__ ucols) {
  const int b = blockIdx.x;
  if (b >= nb) return;
  const long long u0 = uptr64[b];
  const int nu = (int)(uptr64[b + 1] - u0);
  const long long s0 = ia[rb[b]];
  for (int t = threadIdx.x; t < nu; t += BLK) ucols[u0 + t] = uslack[s0 + t];
  if (threadIdx.x == 0) {
    uptr[b] = (int)u0;
    if (b == nb - 1) uptr[nb] = (int)uptr64[nb];
  }
}

// ---------------------------------------------------------------- l1 norms
// G lanes per row, sums in stored order (see strength_k)
template <int G>
__global__ __launch_bounds__(BLK) void level_norms_k(int n, const long long *__restrict__ ia, const int *__restrict__ ja,
                                                     const double *__restrict__ a, const int *__restrict__ cf, int chunk,
                                                     double *__restrict__ diag, double *__restrict__ l1gs,
                                                     double *__restrict__ l1jac, const long long *__restrict__ oia,
                                                     const int *__restrict__ oja, const double *__restrict__ oa,
                                                     const int *__restrict__ cf_ext) {
  const long long i = (bid() * BLK + threadIdx.x) / G;
  const int sub = threadIdx.x % G, lane = threadIdx.x & 63, gbase = lane - sub;
  const bool live = i < n;
  const long long k0 = live ? ia[i] : 0, k1 = live ? ia[i + 1] : 0;
  const long long cs = live ? (i / chunk) * chunk : 0, ce = cs + chunk;
  const int mycf = (cf && live) ? cf[i] : 0;
  double d = 0.0, l1 = 0.0, full = 0.0;
  long long len = k1 - k0;
  for (int m = G; m < 64; m <<= 1) len = max(len, (long long)__shfl_xor((int)len, m, 64));
  for (long long t = 0; t < len; t += G) {
    const long long k = k0 + t + sub;
    const bool ok = k < k1;
    const double v = ok ? a[k] : 0.0;
    const int j = ok ? ja[k] : -1;
    // contribution of this entry to l1: |a| for the diagonal, |a|/2 for an out-of-chunk entry of the row's own
    // C/F type, nothing otherwise
    double h = 0.0;
    if (ok) {
      if (j == i)
        h = fabs(v);
      else if ((j < cs || j >= ce) && (!cf || cf[j] == mycf))
        h = 0.5 * fabs(v);
    }
    const bool isd = ok && j == i;
    // in stored order: every lane adds the G values one after the other.  Only the value travels per step; which
    // entries are the diagonal / count half comes out of ballots (round 4: h and the flag travelled too, five
    // cross-lane reads per step), and h is recomputed from the value: |v| resp. |v| / 2, the same bits.
    const bool half = ok && !isd && h != 0.0;
    const unsigned long long dmask = (G > 1) ? __ballot(isd) : 0ull;
    const unsigned long long hmask = (G > 1) ? __ballot(half) : 0ull;
    const int rem = (int)((k1 - k0 - t) < (long long)G ? (k1 - k0 - t) : (long long)G);
#pragma unroll
    for (int q = 0; q < G; q++) {
      const double vq = (G > 1) ? group_value<G>(v, gbase, q) : v;
      if (q < rem) {
        const bool dq = (G > 1) ? ballot_bit(dmask, gbase + q) : isd;
        const bool hq = (G > 1) ? ballot_bit(hmask, gbase + q) : half;
        const double av = fabs(vq);
        full += av;
        if (dq) {
          d = vq;
          if (av != 0.0) l1 += av;  // the host adds only the contributing entries, in stored order
        } else if (hq) {
          l1 += 0.5 * av;
        }
      }
    }
  }
  if (!live || sub != 0) return;
  if (oia) {  // halo block of the row (N > 1), after the diag block like the host loop
    for (long long k = oia[i]; k < oia[i + 1]; k++) {
      const double av = fabs(oa[k]);
      full += av;
      if (!cf || cf_ext[oja[k]] == mycf) l1 += 0.5 * av;
    }
  }
  if (l1 <= 4.0 / 3.0 * fabs(d)) l1 = fabs(d);
  if (d < 0) {
    l1 = -l1;
    full = -full;
  }
  diag[i] = d;
  l1gs[i] = l1;
  l1jac[i] = full;
}

// ---------------------------------------------------------------- ILU(0)
__global__ __launch_bounds__(BLK) void ilu_dpos_k(int n, const long long *__restrict__ ia, const int *__restrict__ ja,
                                                  long long *__restrict__ dpos) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n) return;
  long long d = -1;
  for (long long k = ia[i]; k < ia[i + 1]; k++)
    if (ja[k] == i) d = k;
  dpos[i] = d;
}

// one thread per row of the level: the oracle's IKJ loop, same operation order
__global__ __launch_bounds__(BLK) void ilu_factor_k(int nrows, const int *__restrict__ rows,
                                                    const long long *__restrict__ ia, const int *__restrict__ ja,
                                                    double *__restrict__ a, const long long *__restrict__ dpos) {
  const long long t = bid() * BLK + threadIdx.x;
  if (t >= nrows) return;
  const int i = rows[t];
  const long long e = ia[i + 1];
  for (long long kk = ia[i]; kk < e; kk++) {
    const int k = ja[kk];
    if (k >= i) break;
    const long long dk = dpos[k];
    if (dk < 0) continue;
    const double l = a[kk] / a[dk];
    a[kk] = l;
    long long pk = dk + 1;
    const long long ek = ia[k + 1];
    for (long long jj = kk + 1; jj < e; jj++) {
      const int j = ja[jj];
      while (pk < ek && ja[pk] < j) pk++;
      if (pk < ek && ja[pk] == j) a[jj] -= l * a[pk];
    }
  }
}

__global__ __launch_bounds__(BLK) void ilu_lower_k(int nrows, const int *__restrict__ rows,
                                                   const long long *__restrict__ ia, const int *__restrict__ ja,
                                                   const double *__restrict__ a, const double *__restrict__ b,
                                                   double *__restrict__ y) {
  const long long t = bid() * BLK + threadIdx.x;
  if (t >= nrows) return;
  const int i = rows ? rows[t] : (int)t;
  double s = b[i];
  for (long long k = ia[i]; k < ia[i + 1] && ja[k] < i; k++) s -= a[k] * y[ja[k]];
  y[i] = s;
}

__global__ __launch_bounds__(BLK) void ilu_upper_k(int nrows, const int *__restrict__ rows,
                                                   const long long *__restrict__ ia, const int *__restrict__ ja,
                                                   const double *__restrict__ a, const long long *__restrict__ dpos,
                                                   const double *__restrict__ y, double *__restrict__ x) {
  const long long t = bid() * BLK + threadIdx.x;
  if (t >= nrows) return;
  const int i = rows ? rows[t] : (int)t;
  double s = y[i];
  const long long d = dpos[i];
  for (long long k = d + 1; k < ia[i + 1]; k++) s -= a[k] * x[ja[k]];
  x[i] = (d >= 0) ? s / a[d] : s;
}

// Jacobi sweeps: every row reads the previous iterate
__global__ __launch_bounds__(BLK) void ilu_lower_jac_k(int n, const long long *__restrict__ ia,
                                                       const int *__restrict__ ja, const double *__restrict__ a,
                                                       const double *__restrict__ b, const double *__restrict__ in,
                                                       double *__restrict__ out) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n) return;
  double s = b[i];
  if (in)
    for (long long k = ia[i]; k < ia[i + 1] && ja[k] < i; k++) s -= a[k] * in[ja[k]];
  out[i] = s;
}

__global__ __launch_bounds__(BLK) void ilu_upper_jac_k(int n, const long long *__restrict__ ia,
                                                       const int *__restrict__ ja, const double *__restrict__ a,
                                                       const long long *__restrict__ dpos,
                                                       const double *__restrict__ b, const double *__restrict__ in,
                                                       double *__restrict__ out) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n) return;
  double s = b[i];
  const long long d = dpos[i];
  if (in)
    for (long long k = d + 1; k < ia[i + 1]; k++) s -= a[k] * in[ja[k]];
  out[i] = (d >= 0) ? s / a[d] : s;
}

// ---------------------------------------------------------------- zero-guess sub-operator
// which entries of row i a first sweep on a zero guess can touch: the row's own chunk, and for an F row
// (i >= nc) the C columns (< nc) the preceding C pass has just written
// MODE 1, the residual that follows that sweep: an F row at or beyond the first chunk boundary >= nc drops its
// C columns -- the F pass has just formed exactly that part of the row's product (every C column lies outside the
// row's chunk there) and hands it over as f - A_FC u_C; all other rows stay whole
template <bool FILL, int MODE, int G>
__global__ __launch_bounds__(BLK) void zero_guess_rows_k(int n, int nc, int chunk, const long long *__restrict__ ia,
                                                         const int *__restrict__ ja, const double *__restrict__ a,
                                                         int *__restrict__ cnt, const long long *__restrict__ zia,
                                                         int *__restrict__ zja, double *__restrict__ za) {
  // G lanes per row: coalesced reads, kept entries written in stored order through a ballot prefix (see strength_k)
  const int i = (int)((bid() * BLK + threadIdx.x) / G);
  const int sub = threadIdx.x % G, lane = threadIdx.x & 63, gbase = lane - sub;
  const bool live = i < n;
  const long long k0 = live ? ia[i] : 0, k1 = live ? ia[i + 1] : 0;
  const int c0 = live ? (i / chunk) * chunk : 0, c1 = c0 + chunk;
  const bool frow = i >= nc;
  const bool drops_c = i >= (nc + chunk - 1) / chunk * chunk;  // MODE 1
  long long o = (FILL && live) ? zia[i] : 0;
  int c = 0;
  int len = (int)(k1 - k0);
  for (int m = G; m < 64; m <<= 1) len = max(len, __shfl_xor(len, m, 64));
  for (int t = 0; t < len; t += G) {
    const long long k = k0 + t + sub;
    bool keep = false;
    int j = -1;
    double v = 0.0;
    if (k < k1) {
      j = ja[k];
      keep = (MODE == 0) ? ((j >= c0 && j < c1) || (frow && j < nc)) : !(drops_c && j < nc);
      if (FILL && keep) v = a[k];
    }
    if (G == 1) {
      if (keep) {
        if (FILL) {
          zja[o] = j;
          za[o] = v;
        }
        o++;
        c++;
      }
    } else {
      const unsigned long long bal = __ballot(keep);
      const unsigned long long mine = (G == 64) ? bal : ((bal >> gbase) & ((1ull << G) - 1ull));
      if (FILL && keep) {
        const long long w = o + __popcll(mine & ((1ull << sub) - 1ull));
        zja[w] = j;
        za[w] = v;
      }
      const int add = __popcll(mine);
      o += add;
      c += add;
    }
  }
  if (!FILL && live && sub == 0) cnt[i] = c;
}

// lanes per row for the row-cooperative setup kernels: the power of two next to half the mean row length
inline int row_group(long long nnz, long long n) {
  const double avg = n > 0 ? (double)nnz / (double)n : 0.0;
  int g = 1;
  while (g < 64 && (double)g * 1.5 < avg) g <<= 1;
  return g;
}
#define MI_ROW_GROUP_DISPATCH(G_, CALL)  \
  switch (G_) {                          \
    case 1: { constexpr int G = 1; CALL; } break;   \
    case 2: { constexpr int G = 2; CALL; } break;   \
    case 4: { constexpr int G = 4; CALL; } break;   \
    case 8: { constexpr int G = 8; CALL; } break;   \
    case 16: { constexpr int G = 16; CALL; } break; \
    case 32: { constexpr int G = 32; CALL; } break; \
    default: { constexpr int G = 64; CALL; } break; \
  }

}  // namespace

// ---------------------------------------------------------------- host-facing entry points
void DCsr::upload(const HostCSR &h, hipStream_t s) {
  ensure_init();
  nrows = h.nrows;
  ncols = h.ncols;
  nnz = h.nnz();
  ia.alloc((size_t)nrows + 1);
  ja.alloc((size_t)nnz);
  a.alloc((size_t)nnz);
  static_assert(sizeof(long long) == sizeof(int64_t), "row pointer width");
  if (h.ia.empty()) {
    MI_HIP(hipMemsetAsync(ia.p, 0, ((size_t)nrows + 1) * sizeof(long long), s));
  } else {
    MI_HIP(hipMemcpyAsync(ia.p, h.ia.data(), ((size_t)nrows + 1) * sizeof(long long), hipMemcpyHostToDevice, s));
  }
  if (nnz) {
    MI_HIP(hipMemcpyAsync(ja.p, h.ja.data(), (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, s));
    MI_HIP(hipMemcpyAsync(a.p, h.a.data(), (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, s));
  }
  MI_HIP(hipStreamSynchronize(s));
}

void DCsr::download(HostCSR &h, hipStream_t s) const {
  h.nrows = nrows;
  h.ncols = ncols;
  h.ia.resize((size_t)nrows + 1);
  h.ja.resize((size_t)nnz);
  h.a.resize((size_t)nnz);
  d2h(h.ia.data(), ia.p, ((size_t)nrows + 1) * sizeof(long long), s);
  if (nnz) {
    d2h(h.ja.data(), ja.p, (size_t)nnz * sizeof(int), s);
    d2h(h.a.data(), a.p, (size_t)nnz * sizeof(double), s);
  }
  MI_HIP(hipStreamSynchronize(s));
}

namespace {
struct Bins {
  int start[5] = {0, 0, 0, 0, 0};
  int tmax = 0;
};

template <bool NUMERIC>
void launch_bins(const Bins &bins, const int *rows, const int *T, int S_hint, const DCsr &A, const DCsr &B, int *nout,
                 const long long *Cia, int *Cja, double *Ca, int *gscratch, long long scratch_per_block, int block_grid,
                 hipStream_t s) {
  const int n0 = bins.start[1] - bins.start[0], n1 = bins.start[2] - bins.start[1], n2 = bins.start[3] - bins.start[2],
            n3 = bins.start[4] - bins.start[3];
  // MI_HYPRE_SPGEMM_ACCUM=0: the numeric phase by per-entry search (spgemm_group_k<.,.,true>), as before round 4
  static const bool accum = [] {
    const char *e = getenv("MI_HYPRE_SPGEMM_ACCUM");
    return !(e && atoi(e) == 0);
  }();
  if (NUMERIC && accum) {
    // the bins bound a row's PRODUCTS; its table only has to hold its distinct columns, which the symbolic pass has counted
    // (a Galerkin row of the 7-point benchmark: 377 products, 30 columns): rows whose count fits the next smaller table
    // take it -- 15 instead of 60 KB of LDS per workgroup, four times the rows in flight
    // The same for the rows beyond 512 products (one workgroup per row in the symbolic pass): up to 512 distinct columns
    // they run here too -- a Galerkin row of a coarse level has 500-1000 products and 100-150 columns; what is left
    // goes to spgemm_block_k below.
    DVec<int> part, part3;
    int hc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // bin 1 small/big, bin 2 small/big, bin 3 small/rest, rest of bin 3 mid/big
    int *s1 = nullptr, *b1 = nullptr, *s2 = nullptr, *b2 = nullptr, *s3 = nullptr, *r3 = nullptr, *m3 = nullptr, *g3 = nullptr;
    if (n1 || n2 || n3) {
      const size_t tot = (size_t)n1 + (size_t)n2 + (size_t)n3;
      part.alloc(2 * tot + 8);
      int *cnt = part.p + 2 * tot;
      MI_HIP(hipMemsetAsync(cnt, 0, 8 * sizeof(int), s));
      s1 = part.p, b1 = s1 + n1, s2 = b1 + n1, b2 = s2 + n2, s3 = b2 + n2, r3 = s3 + n3;
      if (n1) split_rows_k<<<grid_for(((long long)n1 + BLK - 1) / BLK), BLK, 0, s>>>(n1, rows + bins.start[1], nout, 32, s1, b1, cnt);
      if (n2) split_rows_k<<<grid_for(((long long)n2 + BLK - 1) / BLK), BLK, 0, s>>>(n2, rows + bins.start[2], nout, 128, s2, b2, cnt + 2);
      if (n3) split_rows_k<<<grid_for(((long long)n3 + BLK - 1) / BLK), BLK, 0, s>>>(n3, rows + bins.start[3], nout, 128, s3, r3, cnt + 4);
      d2h(hc, cnt, 6 * sizeof(int), s);
      MI_HIP(hipStreamSynchronize(s));
      if (hc[5]) {
        part3.alloc(2 * (size_t)hc[5]);
        m3 = part3.p, g3 = m3 + hc[5];
        split_rows_k<<<grid_for(((long long)hc[5] + BLK - 1) / BLK), BLK, 0, s>>>(hc[5], r3, nout, 512, m3, g3, cnt + 6);
        d2h(hc + 6, cnt + 6, 2 * sizeof(int), s);
        MI_HIP(hipStreamSynchronize(s));
      }
    }
    if (n0)
      spgemm_accum_k<8, 32><<<grid_for(((long long)n0 + 31) / 32), BLK, 0, s>>>(
          n0, rows + bins.start[0], A.ia.p, A.ja.p, A.a.p, B.ia.p, B.ja.p, B.a.p, Cia, Cja, Ca);
    if (hc[0])
      spgemm_accum_k<16, 32><<<grid_for(((long long)hc[0] + 15) / 16), BLK, 0, s>>>(
          hc[0], s1, A.ia.p, A.ja.p, A.a.p, B.ia.p, B.ja.p, B.a.p, Cia, Cja, Ca);
    if (hc[1])
      spgemm_accum_k<16, 128><<<grid_for(((long long)hc[1] + 15) / 16), BLK, 0, s>>>(
          hc[1], b1, A.ia.p, A.ja.p, A.a.p, B.ia.p, B.ja.p, B.a.p, Cia, Cja, Ca);
    if (hc[2])
      spgemm_accum_k<64, 128><<<grid_for(((long long)hc[2] + 3) / 4), BLK, 0, s>>>(
          hc[2], s2, A.ia.p, A.ja.p, A.a.p, B.ia.p, B.ja.p, B.a.p, Cia, Cja, Ca);
    if (hc[3])
      spgemm_accum_k<64, 512><<<grid_for(((long long)hc[3] + 3) / 4), BLK, 0, s>>>(
          hc[3], b2, A.ia.p, A.ja.p, A.a.p, B.ia.p, B.ja.p, B.a.p, Cia, Cja, Ca);
    if (hc[4])
      spgemm_accum_k<64, 128><<<grid_for(((long long)hc[4] + 3) / 4), BLK, 0, s>>>(
          hc[4], s3, A.ia.p, A.ja.p, A.a.p, B.ia.p, B.ja.p, B.a.p, Cia, Cja, Ca);
    if (hc[6])
      spgemm_accum_k<64, 512><<<grid_for(((long long)hc[6] + 3) / 4), BLK, 0, s>>>(
          hc[6], m3, A.ia.p, A.ja.p, A.a.p, B.ia.p, B.ja.p, B.a.p, Cia, Cja, Ca);
    if (hc[7])
      spgemm_block_k<true><<<(unsigned)std::min(hc[7], block_grid), BLK, 0, s>>>(
          hc[7], g3, T, B.ncols, S_hint, gscratch, scratch_per_block, A.ia.p, A.ja.p, A.a.p, B.ia.p, B.ja.p, B.a.p, nout,
          Cia, Cja, Ca);
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(s));  // `part` is released on return
  } else {
    if (n0)
      spgemm_group_k<8, 32, NUMERIC><<<grid_for(((long long)n0 + 31) / 32), BLK, 0, s>>>(
          n0, rows + bins.start[0], std::min(S_hint, 8), A.ia.p, A.ja.p, A.a.p, B.ia.p, B.ja.p, B.a.p, nout, Cia, Cja, Ca);
    if (n1)
      spgemm_group_k<16, 128, NUMERIC><<<grid_for(((long long)n1 + 15) / 16), BLK, 0, s>>>(
          n1, rows + bins.start[1], std::min(S_hint, 16), A.ia.p, A.ja.p, A.a.p, B.ia.p, B.ja.p, B.a.p, nout, Cia, Cja, Ca);
    if (n2)
      spgemm_group_k<64, 512, NUMERIC><<<grid_for(((long long)n2 + 3) / 4), BLK, 0, s>>>(
          n2, rows + bins.start[2], std::min(S_hint, 64), A.ia.p, A.ja.p, A.a.p, B.ia.p, B.ja.p, B.a.p, nout, Cia, Cja, Ca);
  }
  if (n3 && !(NUMERIC && accum))
    spgemm_block_k<NUMERIC><<<(unsigned)std::min(n3, block_grid), BLK, 0, s>>>(
        n3, rows + bins.start[3], T, B.ncols, S_hint, gscratch, scratch_per_block, A.ia.p, A.ja.p, A.a.p, B.ia.p,
        B.ja.p, B.a.p, nout, Cia, Cja, Ca);
  MI_HIP(hipGetLastError());
}
}  // namespace


namespace {
std::vector<int64_t> &host_scratch_i64() {
  static std::vector<int64_t> v;
  return v;
}
}  // namespace
void release_host_scratch() { std::vector<int64_t>().swap(host_scratch_i64()); }

namespace {
__global__ void load_module_k(int *p) {
  if (p) *p = 0;
}
}  // namespace
// The runtime loads a translation unit's device code at the first launch of any of its kernels (from the library file,
// which the first process on a machine reads from a cold disk cache): HYPRE_Init takes that, not the first setup.
void load_device_code(hipStream_t s) {
  load_module_k<<<1, 1, 0, s>>>(nullptr);
  MI_HIP(hipGetLastError());
}

namespace {
// tile schedule, one thread per super-block of k::TILE_SUPER_ROWS rows (k::tile_end is the host routine's step):
// FILL = false counts the super-block's tiles, FILL = true writes their ends behind start[sb]
template <bool FILL>
__global__ __launch_bounds__(BLK) void tile_schedule_k(int n, int nsb, const long long *__restrict__ ia, int row_cap,
                                                       int block_rows, int tile_entries, int *__restrict__ count,
                                                       const long long *__restrict__ start, int *__restrict__ rb,
                                                       int *__restrict__ unaligned) {
  const int sb = blockIdx.x * BLK + threadIdx.x;
  if (sb >= nsb) return;
  int r = sb * k::TILE_SUPER_ROWS;
  const int limit = (int)min((long long)n, (long long)r + k::TILE_SUPER_ROWS);
  bool aligned = true;
  int c = 0;
  long long w = FILL ? start[sb] + 1 : 0;
  while (r < limit) {
    const int e = k::tile_end_bisect(r, limit, ia, row_cap, block_rows, tile_entries, aligned);
    if (FILL) rb[w++] = e;
    c++;
    r = e;
  }
  if (!FILL) count[sb] = c;
  if (!FILL && !aligned) *unaligned = 1;
}
}  // namespace

// k::build_row_blocks on the device: rb (device) and blocks (host) = the tile boundaries, aligned = every tile starts on
// a multiple of 8 rows and was cut at whole chunks
void tile_schedule_device(int n, const long long *ia, int row_cap_in, int tile_entries, DVec<int> &rb,
                          std::vector<int> &blocks, bool &aligned, hipStream_t s) {
  const int block_rows = tile_entries == k::SPMV_TILE_WIDE ? k::SPMV_BLOCK_WIDE : k::SPMV_BLOCK;
  const int row_cap = std::max(row_cap_in, block_rows);
  const int nsb = (int)(((long long)n + k::TILE_SUPER_ROWS - 1) / k::TILE_SUPER_ROWS);
  aligned = true;
  if (nsb > 0) {
    DVec<int> cnt((size_t)nsb), unal(1);
    DVec<long long> st((size_t)nsb + 1);
    MI_HIP(hipMemsetAsync(unal.p, 0, sizeof(int), s));
    const unsigned g = (unsigned)((nsb + BLK - 1) / BLK);
    tile_schedule_k<false><<<g, BLK, 0, s>>>(n, nsb, ia, row_cap, block_rows, tile_entries, cnt.p, nullptr, nullptr, unal.p);
    exclusive_scan(cnt.p, st.p, nsb, s);
    long long nt = 0;
    int un = 0;
    d2h(&nt, st.p + nsb, sizeof(long long), s);
    d2h(&un, unal.p, sizeof(int), s);
    MI_HIP(hipStreamSynchronize(s));
    aligned = un == 0;
    rb.alloc((size_t)nt + 1);
    MI_HIP(hipMemsetAsync(rb.p, 0, sizeof(int), s));
    tile_schedule_k<true><<<g, BLK, 0, s>>>(n, nsb, ia, row_cap, block_rows, tile_entries, nullptr, st.p, rb.p, nullptr);
    blocks.resize((size_t)nt + 1);
    d2h(blocks.data(), rb.p, ((size_t)nt + 1) * sizeof(int), s);
    MI_HIP(hipStreamSynchronize(s));
  } else {
    blocks.assign(1, 0);
    rb.upload(blocks);
  }
}

void to_solve_format(DCsr &src, DevCSR &dst, hipStream_t s) {
  const int n = src.nrows;
  require_int32_block(src.nrows, 0, "solve format");
  dst.nrows = n;
  dst.ncols = src.ncols;
  dst.nnz = src.nnz;
  dst.ia.alloc((size_t)n + 1);
  dst.ia64.release();
  ia_to_32_k<<<(unsigned)((n + 1 + BLK - 1) / BLK), BLK, 0, s>>>((long long)n + 1, src.ia.p, dst.ia.p);  // low words
  // row-length percentile (GS kernel variant choice) from a device histogram
  dst.rowlen_p95 = 0;
  if (n) {
    DVec<int> hist(LEN_BINS);
    MI_HIP(hipMemsetAsync(hist.p, 0, LEN_BINS * sizeof(int), s));
    rowlen_hist_k<<<(unsigned)(((long long)n + (long long)BLK * HIST_RUN - 1) / ((long long)BLK * HIST_RUN)), BLK, 0, s>>>(
        n, src.ia.p, hist.p);
    std::vector<int> hh(LEN_BINS);
    d2h(hh.data(), hist.p, LEN_BINS * sizeof(int), s);
    MI_HIP(hipStreamSynchronize(s));
    const long long kth = (long long)((double)(n - 1) * 0.95);
    long long run = 0;
    for (int b = 0; b < LEN_BINS; b++) {
      run += hh[(size_t)b];
      if (run > kth) {
        dst.rowlen_p95 = b;
        break;
      }
    }
  }
  // greedy row-block schedule, super-block by super-block on the device (round 3 copied the row pointers to the host --
  // 1 GB per operator of 134 M rows -- and walked them there)
  bool aligned = true;
  dst.tile_entries = k::choose_tile_entries(dst.nnz, n);
  std::vector<int> blocks;
  tile_schedule_device(n, src.ia.p, dst.row_cap, dst.tile_entries, dst.rb, blocks, aligned, s);
  if (dst.row_cap > (dst.tile_entries == k::SPMV_TILE_WIDE ? k::SPMV_BLOCK_WIDE : k::SPMV_BLOCK)) aligned = false;  // such tiles are not for the tile Gauss-Seidel kernel
  dst.nblocks = (int)blocks.size() - 1;
  dst.rb_host = blocks;
  dst.gs_tiles = false;
  dst.max_tile_rows = 1;
  for (size_t b = 0; b + 1 < blocks.size(); b++) dst.max_tile_rows = std::max(dst.max_tile_rows, blocks[b + 1] - blocks[b]);
  dst.ja = std::move(src.ja);
  dst.a = std::move(src.a);
  static const int xc_min = getenv("MI_HYPRE_XCACHE_MIN") ? atoi(getenv("MI_HYPRE_XCACHE_MIN")) : 3;
  dst.xcache = n > 0 && (double)dst.nnz / (double)n >= (double)xc_min;
  if (dst.xcache) {
    const int nb = dst.nblocks;
    DVec<int> ucnt((size_t)nb), uslack((size_t)dst.nnz);
    dst.lcol.alloc((size_t)dst.nnz);
    MI_HIP(hipMemsetAsync(dst.lcol.p, 0, (size_t)dst.nnz * sizeof(unsigned short), s));
    if (dst.tile_entries == k::SPMV_TILE_WIDE)
      xcache_block_k<k::SPMV_TILE_WIDE><<<(unsigned)nb, BLK, 0, s>>>(nb, dst.rb.p, src.ia.p, dst.ja.p, ucnt.p, uslack.p, dst.lcol.p);
    else
      xcache_block_k<k::SPMV_TILE><<<(unsigned)nb, BLK, 0, s>>>(nb, dst.rb.p, src.ia.p, dst.ja.p, ucnt.p, uslack.p, dst.lcol.p);
    DVec<long long> uptr64((size_t)nb + 1);
    exclusive_scan(ucnt.p, uptr64.p, nb, s);
    long long tot = 0;
    d2h(&tot, uptr64.p + nb, sizeof(long long), s);
    MI_HIP(hipStreamSynchronize(s));
    dst.uptr.alloc((size_t)nb + 1);
    dst.ucols.alloc((size_t)tot);
    MI_REQUIRE(tot < 2147483647LL, "solve format: the tiles' column lists exceed 2^31 entries");
    xcache_compact_k<<<(unsigned)nb, BLK, 0, s>>>(nb, dst.rb.p, src.ia.p, uptr64.p, uslack.p, dst.uptr.p, dst.ucols.p);
    if (aligned && dst.nrows == dst.ncols) {
      lcol_code_k<<<(unsigned)((n + BLK - 1) / BLK), BLK, 0, s>>>(n, src.ia.p, dst.ja.p, dst.lcol.p);
      dst.gs_tiles = true;
    }
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(s));
  }
  k::build_tile_desc(dst, src.ia.p, s);
  MI_HIP(hipStreamSynchronize(s));
  if (dst.big()) {
    // beyond 2^31 entries: the setup paths that read this operator again (zero-guess sub-operators, host copies) need
    // the full row pointers; the solve runs on the tile kernels only
    MI_REQUIRE(dst.xcache, "solve format: an operator with 2^31 entries or more needs the x-cache tile format");
    dst.ia64 = std::move(src.ia);
  }
  src.release();
  MI_HIP(hipGetLastError());
}

namespace {
// 64-bit row pointers of an operator in the solve format: its own (big operators) or a widened temporary
const long long *wide_row_pointers(const DevCSR &A, DVec<long long> &tmp, hipStream_t s) {
  if (A.ia64.p) return A.ia64.p;
  MI_REQUIRE(!A.big(), "solve format: 64-bit row pointers of a big operator are missing");
  tmp.alloc((size_t)A.nrows + 1);
  ia_to_64_k<<<(unsigned)((A.nrows + 1 + BLK - 1) / BLK), BLK, 0, s>>>((long long)A.nrows + 1, A.ia.p, tmp.p);
  return tmp.p;
}
}  // namespace

void zero_guess_operator(const DevCSR &A, int nc, int chunk, DCsr &Z, hipStream_t s, int mode) {
  const int n = A.nrows;
  Z.release();
  Z.nrows = n;
  Z.ncols = A.ncols;
  Z.ia.alloc((size_t)n + 1);
  DVec<int> cnt((size_t)n);
  DVec<long long> ia_tmp;
  const long long *Aia = n ? wide_row_pointers(A, ia_tmp, s) : nullptr;
  const int rg = row_group(A.nnz, n);
  const dim3 grid = grid_for(((long long)n * rg + BLK - 1) / BLK);
  if (n && mode == 0) {
    MI_ROW_GROUP_DISPATCH(rg, (zero_guess_rows_k<false, 0, G><<<grid, BLK, 0, s>>>(n, nc, chunk, Aia, A.ja.p, A.a.p, cnt.p, nullptr, nullptr, nullptr)))
  } else if (n) {
    MI_ROW_GROUP_DISPATCH(rg, (zero_guess_rows_k<false, 1, G><<<grid, BLK, 0, s>>>(n, nc, chunk, Aia, A.ja.p, A.a.p, cnt.p, nullptr, nullptr, nullptr)))
  }
  exclusive_scan(cnt.p, Z.ia.p, n, s);
  long long total = 0;
  d2h(&total, Z.ia.p + n, sizeof(long long), s);
  MI_HIP(hipStreamSynchronize(s));
  Z.nnz = total;
  Z.ja.alloc((size_t)total);
  Z.a.alloc((size_t)total);
  if (n && total && mode == 0) {
    MI_ROW_GROUP_DISPATCH(rg, (zero_guess_rows_k<true, 0, G><<<grid, BLK, 0, s>>>(n, nc, chunk, Aia, A.ja.p, A.a.p, nullptr, Z.ia.p, Z.ja.p, Z.a.p)))
  } else if (n && total) {
    MI_ROW_GROUP_DISPATCH(rg, (zero_guess_rows_k<true, 1, G><<<grid, BLK, 0, s>>>(n, nc, chunk, Aia, A.ja.p, A.a.p, nullptr, Z.ia.p, Z.ja.p, Z.a.p)))
  }
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));
}

void solve_format_to_host(const DevCSR &src, HostCSR &h, hipStream_t s) {
  const int n = src.nrows;
  h.nrows = n;
  h.ncols = src.ncols;
  h.ia.assign((size_t)n + 1, 0);
  h.ja.resize((size_t)src.nnz);
  h.a.resize((size_t)src.nnz);
  if (src.ia.p) {
    DVec<long long> ia_tmp;
    const long long *ia64 = wide_row_pointers(src, ia_tmp, s);
    d2h(h.ia.data(), ia64, ((size_t)n + 1) * sizeof(long long), s);
    if (src.nnz) {
      d2h(h.ja.data(), src.ja.p, (size_t)src.nnz * sizeof(int), s);
      d2h(h.a.data(), src.a.p, (size_t)src.nnz * sizeof(double), s);
    }
    MI_HIP(hipStreamSynchronize(s));
  }
  if (src.rowmap.p && n) {  // stored row r is row rowmap[r] of the operator: hand the rows back in operator order
    std::vector<int> map((size_t)n);
    d2h(map.data(), src.rowmap.p, (size_t)n * sizeof(int), nullptr);
    HostCSR o;
    o.nrows = n;
    o.ncols = h.ncols;
    o.ia.assign((size_t)n + 1, 0);
    for (int r = 0; r < n; r++) o.ia[(size_t)map[(size_t)r] + 1] = h.ia[(size_t)r + 1] - h.ia[(size_t)r];
    for (int r = 0; r < n; r++) o.ia[(size_t)r + 1] += o.ia[(size_t)r];
    o.ja.resize(h.ja.size());
    o.a.resize(h.a.size());
    for (int r = 0; r < n; r++) {
      const int64_t len = h.ia[(size_t)r + 1] - h.ia[(size_t)r], from = h.ia[(size_t)r], to = o.ia[(size_t)map[(size_t)r]];
      std::copy(h.ja.begin() + from, h.ja.begin() + from + len, o.ja.begin() + to);
      std::copy(h.a.begin() + from, h.a.begin() + from + len, o.a.begin() + to);
    }
    h = std::move(o);
  }
}

void non_galerkin_row_maxima(const DCsr &A, int row0, double *m, hipStream_t s) {
  const int n = A.nrows;
  if (n) ng_rowmax_k<<<(unsigned)((n + BLK - 1) / BLK), BLK, 0, s>>>(n, row0, A.ia.p, A.ja.p, A.a.p, m);
  MI_HIP(hipGetLastError());
}

void sparsify_non_galerkin(DCsr &A, double tol, hipStream_t s, int row0, const double *maxima) {
  const int n = A.nrows;
  if (n == 0 || !(tol > 0.0)) return;
  MI_REQUIRE(maxima || A.nrows == A.ncols, "non-Galerkin sparsification: square operators only");
  const unsigned grid = (unsigned)((n + BLK - 1) / BLK);
  DVec<double> m_own;
  if (!maxima) {
    m_own.alloc((size_t)n);
    ng_rowmax_k<<<grid, BLK, 0, s>>>(n, 0, A.ia.p, A.ja.p, A.a.p, m_own.p);
    row0 = 0;
  }
  const double *mp = maxima ? maxima : m_own.p;
  DVec<int> cnt((size_t)n);
  ng_drop_k<false><<<grid, BLK, 0, s>>>(n, row0, A.ia.p, A.ja.p, A.a.p, mp, tol, cnt.p, nullptr, nullptr, nullptr);
  DCsr B;
  B.nrows = n;
  B.ncols = A.ncols;
  B.ia.alloc((size_t)n + 1);
  exclusive_scan(cnt.p, B.ia.p, n, s);
  long long total = 0;
  d2h(&total, B.ia.p + n, sizeof(long long), s);
  MI_HIP(hipStreamSynchronize(s));
  B.nnz = total;
  B.ja.alloc((size_t)total);
  B.a.alloc((size_t)total);
  ng_drop_k<true><<<grid, BLK, 0, s>>>(n, row0, A.ia.p, A.ja.p, A.a.p, mp, tol, nullptr, B.ia.p, B.ja.p, B.a.p);
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));
  A = std::move(B);
}

void from_solve_format(const DevCSR &src, DCsr &dst, hipStream_t s) {
  MI_REQUIRE(!src.rowmap.p, "from_solve_format: the operator's rows are stored in another order");
  const int n = src.nrows;
  dst.nrows = n;
  dst.ncols = src.ncols;
  dst.nnz = src.nnz;
  dst.ia.alloc((size_t)n + 1);
  dst.ja.alloc((size_t)src.nnz);
  dst.a.alloc((size_t)src.nnz);
  if (src.ia64.p)
    MI_HIP(hipMemcpyAsync(dst.ia.p, src.ia64.p, ((size_t)n + 1) * sizeof(long long), hipMemcpyDeviceToDevice, s));
  else
    ia_to_64_k<<<(unsigned)((n + 1 + BLK - 1) / BLK), BLK, 0, s>>>((long long)n + 1, src.ia.p, dst.ia.p);
  if (src.nnz) {
    MI_HIP(hipMemcpyAsync(dst.ja.p, src.ja.p, (size_t)src.nnz * sizeof(int), hipMemcpyDeviceToDevice, s));
    MI_HIP(hipMemcpyAsync(dst.a.p, src.a.p, (size_t)src.nnz * sizeof(double), hipMemcpyDeviceToDevice, s));
  }
  MI_HIP(hipGetLastError());
}

int locality_labels(const DCsr &A, const int *seeds_host, int nseeds, const unsigned char *exclude_host, int segshift,
                    int max_rounds, std::vector<int> &label_host, hipStream_t s) {
  const int n = A.nrows;
  label_host.assign((size_t)n, -1);
  if (n == 0) return 0;
  DVec<int> la((size_t)n), lb((size_t)n), dseeds((size_t)std::max(1, nseeds)), changed(1);
  if (exclude_host) {  // -2 for the rows that stay out
    for (int i = 0; i < n; i++)
      if (exclude_host[i]) label_host[(size_t)i] = -2;
    MI_HIP(hipMemcpyAsync(la.p, label_host.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, s));
  } else {
    MI_HIP(hipMemsetAsync(la.p, 0xFF, (size_t)n * sizeof(int), s));  // -1
  }
  if (nseeds) {
    MI_HIP(hipMemcpyAsync(dseeds.p, seeds_host, (size_t)nseeds * sizeof(int), hipMemcpyHostToDevice, s));
    locality_seed_k<<<(unsigned)((nseeds + BLK - 1) / BLK), BLK, 0, s>>>(nseeds, dseeds.p, la.p);
  }
  int *in = la.p, *out = lb.p;
  int rounds = 0;
  for (; rounds < max_rounds; rounds++) {
    MI_HIP(hipMemsetAsync(changed.p, 0, sizeof(int), s));
    locality_round_k<<<(unsigned)((n + BLK - 1) / BLK), BLK, 0, s>>>(n, A.ia.p, A.ja.p, in, out, changed.p, segshift);
    int ch = 0;
    d2h(&ch, changed.p, sizeof(int), s);
    MI_HIP(hipStreamSynchronize(s));
    std::swap(in, out);
    if (!ch) break;
  }
  d2h(label_host.data(), in, (size_t)n * sizeof(int), s);
  MI_HIP(hipStreamSynchronize(s));
  MI_HIP(hipGetLastError());
  return rounds;
}

// ---- the whole internal numbering on the device (single rank, nothing excluded): seeds, label rounds, cells ranked by
// their smallest row, rows by cell rank in natural order.  Same result as hs::locality_order (amg_setup.cpp), which
// stays the routine of the host-only and the multi-rank setup (tests/test_locality_order.py compares the two).
namespace {
__global__ __launch_bounds__(BLK) void loc_seed_flag_k(int n, unsigned long long cluster, int segshift, int *__restrict__ flag,
                                                       int *__restrict__ seg_has) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  unsigned long long z = (unsigned long long)i + 0x9E3779B97F4A7C15ULL;  // splitmix64, as hs::locality_seeds
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z ^= z >> 31;
  const int f = (z % cluster == 0) ? 1 : 0;
  flag[i] = f;
  if (f) seg_has[i >> segshift] = 1;  // (every writer stores the same value)
}
__global__ __launch_bounds__(BLK) void loc_seg_fix_k(int nseg, int segshift, const int *__restrict__ seg_has, int *__restrict__ flag) {
  const int sg = blockIdx.x * BLK + threadIdx.x;
  if (sg < nseg && !seg_has[sg]) flag[(size_t)sg << segshift] = 1;  // a segment without a seed: its first row
}
__global__ __launch_bounds__(BLK) void loc_seed_label_k(int n, const int *__restrict__ flag, const long long *__restrict__ rank,
                                                        int *__restrict__ label) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i < n) label[i] = flag[i] ? (int)rank[i] : -1;
}
__global__ __launch_bounds__(BLK) void loc_minrow_k(int n, const int *__restrict__ label, int *__restrict__ minrow) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i < n && label[i] >= 0) atomicMin(minrow + label[i], i);
}
// key = rank of the row's cell (unreached rows: the last cluster), counted per key
__global__ __launch_bounds__(BLK) void loc_key_count_k(int n, int nseeds, const int *__restrict__ label,
                                                       const int *__restrict__ cell_rank, int *__restrict__ key,
                                                       int *__restrict__ count) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  const int l = label[i];
  const int kq = l >= 0 ? cell_rank[l] : nseeds;
  key[i] = kq;
  atomicAdd(count + kq, 1);
}
__global__ __launch_bounds__(BLK) void loc_scatter_k(int n, const int *__restrict__ key, const long long *__restrict__ start,
                                                     int *__restrict__ cursor, int *__restrict__ order) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  const int kq = key[i];
  order[start[kq] + atomicAdd(cursor + kq, 1)] = i;  // any order inside the cell: sorted next
}
// one workgroup per cell: its rows ascending (bitonic sort in LDS); cells beyond LOC_SORT_CAP rows are left to
// loc_big_* below
constexpr int LOC_SORT_CAP = 4096;
__global__ __launch_bounds__(BLK) void loc_sort_cells_k(int ncells, const long long *__restrict__ start, int *__restrict__ order) {
  __shared__ int v[LOC_SORT_CAP];
  const int c = blockIdx.x;
  if (c >= ncells) return;
  const long long b = start[c];
  const int len = (int)(start[c + 1] - b);
  if (len <= 1 || len > LOC_SORT_CAP) return;
  int m = 1;
  while (m < len) m <<= 1;
  for (int k = threadIdx.x; k < m; k += BLK) v[k] = k < len ? order[b + k] : 0x7fffffff;
  __syncthreads();
  for (int size = 2; size <= m; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < (m >> 1); t += BLK) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool up = ((lo & size) == 0);
        const int a = v[lo], bb = v[hi];
        if ((a > bb) == up) {
          v[lo] = bb;
          v[hi] = a;
        }
      }
      __syncthreads();
    }
  for (int k = threadIdx.x; k < len; k += BLK) order[b + k] = v[k];
}
__global__ __launch_bounds__(BLK) void loc_key_eq_k(int n, const int *__restrict__ key, int kq, int *__restrict__ flag) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i < n) flag[i] = key[i] == kq;
}
__global__ __launch_bounds__(BLK) void loc_big_place_k(int n, const int *__restrict__ flag, const long long *__restrict__ rank,
                                                       long long base, int *__restrict__ order) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i < n && flag[i]) order[base + rank[i]] = i;
}
}  // namespace

bool locality_order_device(const DCsr &A, int segshift, int cluster, int max_rounds, DVec<int> &order, int &nseeds,
                           int &rounds, hipStream_t s) {
  const int n = A.nrows;
  nseeds = rounds = 0;
  order.alloc((size_t)n);
  if (n == 0) return true;
  const unsigned gn = (unsigned)((n + BLK - 1) / BLK);
  const int nseg = (int)((((long long)n - 1) >> segshift) + 1);
  DVec<int> la((size_t)n), lb((size_t)n);
  {
    DVec<int> flag((size_t)n), seg_has((size_t)nseg);
    DVec<long long> rank((size_t)n + 1);
    MI_HIP(hipMemsetAsync(seg_has.p, 0, (size_t)nseg * sizeof(int), s));
    loc_seed_flag_k<<<gn, BLK, 0, s>>>(n, (unsigned long long)cluster, segshift, flag.p, seg_has.p);
    loc_seg_fix_k<<<(unsigned)((nseg + BLK - 1) / BLK), BLK, 0, s>>>(nseg, segshift, seg_has.p, flag.p);
    exclusive_scan(flag.p, rank.p, n, s);
    long long ns = 0;
    d2h(&ns, rank.p + n, sizeof(long long), s);
    MI_HIP(hipStreamSynchronize(s));
    nseeds = (int)ns;
    loc_seed_label_k<<<gn, BLK, 0, s>>>(n, flag.p, rank.p, la.p);
  }
  int *in = la.p, *out = lb.p;
  DVec<int> changed(1);
  for (; rounds < max_rounds; rounds++) {
    MI_HIP(hipMemsetAsync(changed.p, 0, sizeof(int), s));
    locality_round_k<<<gn, BLK, 0, s>>>(n, A.ia.p, A.ja.p, in, out, changed.p, segshift);
    int ch = 0;
    d2h(&ch, changed.p, sizeof(int), s);
    MI_HIP(hipStreamSynchronize(s));
    std::swap(in, out);
    if (!ch) break;
  }
  // cells ranked by their smallest row (a cell always holds its seed, so every cell has one); the sort of the ~n/512
  // (smallest row, cell) pairs is host work of a millisecond
  DVec<int> minrow((size_t)nseeds);
  {
    std::vector<int> init((size_t)nseeds, n);
    MI_HIP(hipMemcpyAsync(minrow.p, init.data(), (size_t)nseeds * sizeof(int), hipMemcpyHostToDevice, s));
    MI_HIP(hipStreamSynchronize(s));
  }
  loc_minrow_k<<<gn, BLK, 0, s>>>(n, in, minrow.p);
  MI_HIP(hipStreamSynchronize(s));
  const std::vector<int> mr = minrow.to_host();
  std::vector<int> cells((size_t)nseeds), crank((size_t)nseeds);
  for (int c = 0; c < nseeds; c++) cells[(size_t)c] = c;
  std::sort(cells.begin(), cells.end(), [&](int a, int b) { return mr[(size_t)a] != mr[(size_t)b] ? mr[(size_t)a] < mr[(size_t)b] : a < b; });
  for (int q = 0; q < nseeds; q++) crank[(size_t)cells[(size_t)q]] = q;
  DVec<int> dcrank;
  dcrank.upload(crank);
  const int ncells = nseeds + 1;  // + the unreached rest
  DVec<int> key((size_t)n), count((size_t)ncells), cursor((size_t)ncells);
  DVec<long long> start((size_t)ncells + 1);
  MI_HIP(hipMemsetAsync(count.p, 0, (size_t)ncells * sizeof(int), s));
  MI_HIP(hipMemsetAsync(cursor.p, 0, (size_t)ncells * sizeof(int), s));
  loc_key_count_k<<<gn, BLK, 0, s>>>(n, nseeds, in, dcrank.p, key.p, count.p);
  exclusive_scan(count.p, start.p, ncells, s);
  loc_scatter_k<<<gn, BLK, 0, s>>>(n, key.p, start.p, cursor.p, order.p);
  loc_sort_cells_k<<<(unsigned)ncells, BLK, 0, s>>>(ncells, start.p, order.p);
  // cells too large for the LDS sort (rare: the unreached rest of a disconnected graph): their rows by one flagged scan each
  MI_HIP(hipStreamSynchronize(s));
  const std::vector<int> hc = count.to_host();
  std::vector<int> big;
  for (int c = 0; c < ncells; c++)
    if (hc[(size_t)c] > LOC_SORT_CAP) big.push_back(c);
  if (big.size() > 8) return false;  // (the caller takes the host routine)
  if (!big.empty()) {
    std::vector<long long> hs_ = start.to_host();
    DVec<int> flag((size_t)n);
    DVec<long long> rank((size_t)n + 1);
    for (int c : big) {
      loc_key_eq_k<<<gn, BLK, 0, s>>>(n, key.p, c, flag.p);
      exclusive_scan(flag.p, rank.p, n, s);
      loc_big_place_k<<<gn, BLK, 0, s>>>(n, flag.p, rank.p, hs_[(size_t)c], order.p);
    }
  }
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));
  return true;
}

void invert_permutation(const int *order, int n, int *pos, hipStream_t s) {
  if (n == 0) return;
  invert_perm_k<<<(unsigned)((n + BLK - 1) / BLK), BLK, 0, s>>>(n, order, pos);
  MI_HIP(hipGetLastError());
}

void level_norms(const DCsr &A, const int *cf, int chunk, double *diag, double *l1gs, double *l1jac, hipStream_t s,
                 const DCsr *halo, const int *cf_ext) {
  const int n = A.nrows;
  if (n == 0) return;
  const int rg = row_group(A.nnz, n);
  const dim3 grid = grid_for(((long long)n * rg + BLK - 1) / BLK);
  const long long *oia = (halo && halo->nnz > 0) ? halo->ia.p : nullptr;
  const int *oja = oia ? halo->ja.p : nullptr;
  const double *oa = oia ? halo->a.p : nullptr;
  MI_ROW_GROUP_DISPATCH(rg, (level_norms_k<G><<<grid, BLK, 0, s>>>(n, A.ia.p, A.ja.p, A.a.p, cf, chunk, diag, l1gs, l1jac, oia, oja, oa, cf_ext)))
  MI_HIP(hipGetLastError());
}

void strength(const DCsr &A, double theta, double max_row_sum, DCsr &S, hipStream_t s) {
  const int n = A.nrows;
  S.release();
  S.nrows = n;
  S.ncols = A.ncols;
  S.ia.alloc((size_t)n + 1);
  DVec<int> cnt((size_t)n);
  const int rg = row_group(A.nnz, n);
  const dim3 grid = grid_for(((long long)n * rg + BLK - 1) / BLK);
  if (n) {
    MI_ROW_GROUP_DISPATCH(rg, (strength_k<false, G><<<grid, BLK, 0, s>>>(n, A.ia.p, A.ja.p, A.a.p, theta, max_row_sum, cnt.p, nullptr, nullptr)))
  }
  exclusive_scan(cnt.p, S.ia.p, n, s);
  long long total = 0;
  d2h(&total, S.ia.p + n, sizeof(long long), s);
  MI_HIP(hipStreamSynchronize(s));
  S.nnz = total;
  S.ja.alloc((size_t)total);
  if (n && total) {
    MI_ROW_GROUP_DISPATCH(rg, (strength_k<true, G><<<grid, BLK, 0, s>>>(n, A.ia.p, A.ja.p, A.a.p, theta, max_row_sum, nullptr, S.ia.p, S.ja.p)))
  }
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));
}

void pmis(const DCsr &S, int seed, DVec<int> &cf, hipStream_t s) {
  const int n = S.nrows;
  cf.alloc((size_t)n);
  if (n == 0) return;
  const unsigned grid = (unsigned)((n + BLK - 1) / BLK);
  DVec<int> incoming((size_t)n), counter(1);
  DVec<double> measure((size_t)n);
  DVec<signed char> tmp((size_t)n);
  MI_HIP(hipMemsetAsync(incoming.p, 0, (size_t)n * sizeof(int), s));
  MI_HIP(hipMemsetAsync(counter.p, 0, sizeof(int), s));
  if (S.nnz) col_count_k<<<grid_for((S.nnz + BLK - 1) / BLK), BLK, 0, s>>>(S.nnz, S.ja.p, incoming.p);
  pmis_init_k<<<grid, BLK, 0, s>>>(n, S.ia.p, incoming.p, seed ? seed : 13579, measure.p, cf.p, counter.p);
  int undecided = 0;
  d2h(&undecided, counter.p, sizeof(int), s);
  MI_HIP(hipStreamSynchronize(s));
  int rounds = 0;
  while (undecided > 0) {
    MI_REQUIRE(++rounds <= 10000, "PMIS does not terminate");
    pmis_mark_k<<<grid, BLK, 0, s>>>(n, cf.p, tmp.p);
    pmis_compare_k<<<grid, BLK, 0, s>>>(n, S.ia.p, S.ja.p, cf.p, measure.p, tmp.p);
    pmis_select_k<<<grid, BLK, 0, s>>>(n, cf.p, tmp.p);
    MI_HIP(hipMemsetAsync(counter.p, 0, sizeof(int), s));
    pmis_fpoints_k<<<grid, BLK, 0, s>>>(n, S.ia.p, S.ja.p, cf.p, counter.p);
    d2h(&undecided, counter.p, sizeof(int), s);
    MI_HIP(hipStreamSynchronize(s));
  }
  MI_HIP(hipGetLastError());
}

bool interp(const DCsr &A, const DCsr &S, DVec<int> &cf, int interp_type, double trunc_factor, int pmax, DCsr &P,
            int &nc, hipStream_t s) {
  MI_REQUIRE(interp_type == 6 || interp_type == 0, "device interpolation: type 6 (ext+i) or 0 (classical modified)");
  const int n = A.nrows;
  const int ext = interp_type == 6;
  P.release();
  if (n == 0) return false;
  const unsigned grid_rows = (unsigned)((n + BLK - 1) / BLK);
  DVec<int> T((size_t)n), cap((size_t)n), is_c((size_t)n), rows((size_t)n), len((size_t)n), meta(16);
  MI_HIP(hipMemsetAsync(meta.p, 0, 16 * sizeof(int), s));
  interp_bound_k<<<grid_rows, BLK, 0, s>>>(n, S.ia.p, S.ja.p, cf.p, ext, pmax, T.p, cap.p, is_c.p, meta.p, meta.p + 4);
  int hmeta[16];
  d2h(hmeta, meta.p, sizeof(hmeta), s);
  MI_HIP(hipStreamSynchronize(s));
  if (hmeta[3] > 0 && hmeta[4] > 1024) return false;  // a row may exceed the largest LDS table
  Bins bins;
  for (int b = 0; b < 4; b++) bins.start[b + 1] = bins.start[b] + hmeta[b];
  for (int b = 0; b < 4; b++) hmeta[8 + b] = bins.start[b], hmeta[12 + b] = 0;
  MI_HIP(hipMemcpyAsync(meta.p + 8, hmeta + 8, 8 * sizeof(int), hipMemcpyHostToDevice, s));
  bin_fill_k<<<grid_rows, BLK, 0, s>>>(n, T.p, meta.p + 8, meta.p + 12, rows.p);
  // coarse numbering and the slack row pointers
  DVec<long long> f2c((size_t)n + 1), slack_ia((size_t)n + 1);
  exclusive_scan(is_c.p, f2c.p, n, s);
  exclusive_scan(cap.p, slack_ia.p, n, s);
  long long tot[2];
  d2h(&tot[0], f2c.p + n, sizeof(long long), s);
  d2h(&tot[1], slack_ia.p + n, sizeof(long long), s);
  MI_HIP(hipStreamSynchronize(s));
  nc = (int)tot[0];
  DVec<int> sj((size_t)tot[1]);
  DVec<double> sa((size_t)tot[1]);
  const int n0 = bins.start[1] - bins.start[0], n1 = bins.start[2] - bins.start[1], n2 = bins.start[3] - bins.start[2],
            n3 = bins.start[4] - bins.start[3];
  if (n3)  // up to 1024 candidates: one workgroup per row
    interp_group_k<256, 1024, 256><<<grid_for(n3), 256, 0, s>>>(
        n3, rows.p + bins.start[3], ext, A.ia.p, A.ja.p, A.a.p, S.ia.p, S.ja.p, cf.p, f2c.p, trunc_factor, pmax,
        slack_ia.p, sj.p, sa.p, len.p);
  if (n0) {
    // the kernel's rows in flight are bounded by LDS (1.5 KB per row with 32-entry tables: 3 workgroups per CU), and what a
    // row costs is ~30 dependent round trips: rows with at most 16 candidates (C points, rows of a 7-point operator) take
    // half-size tables = twice the rows in flight
    DVec<int> part0((size_t)2 * (size_t)n0 + 2);
    int *cnt0 = part0.p + 2 * (size_t)n0;
    int h0[2] = {0, 0};
    MI_HIP(hipMemsetAsync(cnt0, 0, 2 * sizeof(int), s));
    split_rows_k<<<grid_for(((long long)n0 + BLK - 1) / BLK), BLK, 0, s>>>(n0, rows.p + bins.start[0], T.p, 16, part0.p,
                                                                         part0.p + n0, cnt0);
    d2h(h0, cnt0, sizeof(h0), s);
    MI_HIP(hipStreamSynchronize(s));
    if (h0[0])
      interp_group_k<8, 16, 256><<<grid_for(((long long)h0[0] + 31) / 32), 256, 0, s>>>(
          h0[0], part0.p, ext, A.ia.p, A.ja.p, A.a.p, S.ia.p, S.ja.p, cf.p, f2c.p, trunc_factor, pmax, slack_ia.p, sj.p,
          sa.p, len.p);
    if (h0[1])
      interp_group_k<8, 32, 256><<<grid_for(((long long)h0[1] + 31) / 32), 256, 0, s>>>(
          h0[1], part0.p + n0, ext, A.ia.p, A.ja.p, A.a.p, S.ia.p, S.ja.p, cf.p, f2c.p, trunc_factor, pmax, slack_ia.p,
          sj.p, sa.p, len.p);
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(s));  // part0 is released at the end of this block
  }
  if (n1) {
    // bound 33 ... 128: first with 32-entry tables (a quarter of the LDS, four times the rows in flight); the rows that do
    // not fit come back marked and go through the tables sized by the bound.  MI_HYPRE_INTERP_TRY=0: all of them there.
    static const bool try_small = !(getenv("MI_HYPRE_INTERP_TRY") && atoi(getenv("MI_HYPRE_INTERP_TRY")) == 0);
    const int *big = rows.p + bins.start[1];
    int nbig = n1;
    DVec<int> part1;
    if (try_small) {
      static_assert(16 <= 32, "CAP + G keys must stay below the 2 CAP slots of the table");
      interp_group_k<16, 32, 256, true><<<grid_for(((long long)n1 + 15) / 16), 256, 0, s>>>(
          n1, rows.p + bins.start[1], ext, A.ia.p, A.ja.p, A.a.p, S.ia.p, S.ja.p, cf.p, f2c.p, trunc_factor, pmax,
          slack_ia.p, sj.p, sa.p, len.p);
      part1.alloc((size_t)2 * (size_t)n1 + 2);
      int *cnt1 = part1.p + 2 * (size_t)n1;
      int h1[2] = {0, 0};
      MI_HIP(hipMemsetAsync(cnt1, 0, 2 * sizeof(int), s));
      split_rows_k<<<grid_for(((long long)n1 + BLK - 1) / BLK), BLK, 0, s>>>(n1, rows.p + bins.start[1], len.p, -1, part1.p,
                                                                           part1.p + n1, cnt1);
      d2h(h1, cnt1, sizeof(h1), s);
      MI_HIP(hipStreamSynchronize(s));
      big = part1.p;  // the rows marked -1
      nbig = h1[0];
    }
    if (nbig)
      interp_group_k<16, 128, 128><<<grid_for(((long long)nbig + 7) / 8), 128, 0, s>>>(
          nbig, big, ext, A.ia.p, A.ja.p, A.a.p, S.ia.p, S.ja.p, cf.p, f2c.p, trunc_factor, pmax, slack_ia.p, sj.p, sa.p,
          len.p);
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(s));  // part1 is released at the end of this block
  }
  if (n2)
    interp_group_k<64, 512, 128><<<grid_for(((long long)n2 + 1) / 2), 128, 0, s>>>(
        n2, rows.p + bins.start[2], ext, A.ia.p, A.ja.p, A.a.p, S.ia.p, S.ja.p, cf.p, f2c.p, trunc_factor, pmax,
        slack_ia.p, sj.p, sa.p, len.p);
  MI_HIP(hipGetLastError());
  P.nrows = n;
  P.ncols = nc;
  P.ia.alloc((size_t)n + 1);
  exclusive_scan(len.p, P.ia.p, n, s);
  long long total = 0;
  d2h(&total, P.ia.p + n, sizeof(long long), s);
  MI_HIP(hipStreamSynchronize(s));
  P.nnz = total;
  P.ja.alloc((size_t)total);
  P.a.alloc((size_t)total);
  compact_rows_k<<<grid_rows, BLK, 0, s>>>(n, slack_ia.p, sj.p, sa.p, P.ia.p, P.ja.p, P.a.p);
  sf_to_f_k<<<grid_rows, BLK, 0, s>>>(n, cf.p);
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));
  return true;
}

void spgemm(const DCsr &A, const DCsr &B, DCsr &C, hipStream_t s) {
  MI_REQUIRE(A.ncols == B.nrows, "spgemm: inner dimensions differ");
  const int n = A.nrows;
  C.release();
  C.nrows = n;
  C.ncols = B.ncols;
  C.ia.alloc((size_t)n + 1);
  if (n == 0 || A.nnz == 0 || B.nnz == 0) {
    MI_HIP(hipMemsetAsync(C.ia.p, 0, ((size_t)n + 1) * sizeof(long long), s));
    C.ja.alloc(0);
    C.a.alloc(0);
    MI_HIP(hipStreamSynchronize(s));
    return;
  }
  const unsigned grid_rows = (unsigned)((n + BLK - 1) / BLK);
  DVec<int> T((size_t)n), rows((size_t)n), nout((size_t)n), meta(16);
  MI_HIP(hipMemsetAsync(meta.p, 0, 16 * sizeof(int), s));
  // meta: [0..3] bin counts, [4] max T, [8..11] bin starts, [12..15] fill cursors
  row_products_k<<<grid_rows, BLK, 0, s>>>(n, A.ia.p, A.ja.p, B.ia.p, T.p, meta.p, meta.p + 4);
  int hmeta[16];
  d2h(hmeta, meta.p, sizeof(hmeta), s);
  MI_HIP(hipStreamSynchronize(s));
  Bins bins;
  for (int b = 0; b < 4; b++) bins.start[b + 1] = bins.start[b] + hmeta[b];
  bins.tmax = hmeta[4];
  for (int b = 0; b < 4; b++) hmeta[8 + b] = bins.start[b], hmeta[12 + b] = 0;
  MI_HIP(hipMemcpyAsync(meta.p + 8, hmeta + 8, 8 * sizeof(int), hipMemcpyHostToDevice, s));
  bin_fill_k<<<grid_rows, BLK, 0, s>>>(n, T.p, meta.p + 8, meta.p + 12, rows.p);
  // lanes per B row in the hash phase ~ B's mean row length
  const int S_hint = pow2_at_most((double)B.nnz / std::max(1, B.nrows), 64);
  // long rows whose bound exceeds the LDS table use a per-block slice of global scratch
  DVec<int> gscratch;
  long long per_block = 0;
  int block_grid = 2048;
  const int n3 = bins.start[4] - bins.start[3];
  if (n3) {
    const long long bound = std::min<long long>(bins.tmax, B.ncols);
    if (bound > BLK_LDS_CAP) {
      long long H = 64;
      while (H < 2 * bound) H *= 2;
      per_block = H + bound;
      const long long budget = 1LL << 29;  // ints (2 GiB)
      block_grid = (int)std::max<long long>(32, std::min<long long>(2048, budget / per_block));
      block_grid = std::min(block_grid, n3);
      gscratch.alloc((size_t)(per_block * block_grid));
    }
  }
  launch_bins<false>(bins, rows.p, T.p, S_hint, A, B, nout.p, nullptr, nullptr, nullptr, gscratch.p, per_block,
                     block_grid, s);
  exclusive_scan(nout.p, C.ia.p, n, s);
  long long total = 0;
  d2h(&total, C.ia.p + n, sizeof(long long), s);
  MI_HIP(hipStreamSynchronize(s));
  C.nnz = total;
  C.ja.alloc((size_t)total);
  C.a.alloc((size_t)total);
  launch_bins<true>(bins, rows.p, T.p, S_hint, A, B, nout.p, C.ia.p, C.ja.p, C.a.p, gscratch.p, per_block, block_grid,
                    s);
  MI_HIP(hipStreamSynchronize(s));
}

void transpose(const DCsr &A, DCsr &T, hipStream_t s) {
  T.release();
  T.nrows = A.ncols;
  T.ncols = A.nrows;
  T.nnz = A.nnz;
  T.ia.alloc((size_t)T.nrows + 1);
  T.ja.alloc((size_t)A.nnz);
  T.a.alloc((size_t)A.nnz);
  DVec<int> cnt((size_t)T.nrows);
  if (T.nrows) MI_HIP(hipMemsetAsync(cnt.p, 0, (size_t)T.nrows * sizeof(int), s));
  if (A.nnz) col_count_k<<<grid_for((A.nnz + BLK - 1) / BLK), BLK, 0, s>>>(A.nnz, A.ja.p, cnt.p);
  exclusive_scan(cnt.p, T.ia.p, T.nrows, s);
  if (A.nnz == 0) return;
  MI_HIP(hipMemsetAsync(cnt.p, 0, (size_t)T.nrows * sizeof(int), s));
  DVec<int> tj((size_t)A.nnz);
  DVec<double> ta((size_t)A.nnz);
  transpose_fill_k<<<(unsigned)((A.nrows + BLK - 1) / BLK), BLK, 0, s>>>(A.nrows, A.ia.p, A.ja.p, A.a.p, T.ia.p, cnt.p,
                                                                         tj.p, ta.p);
  sort_rows(T.nrows, T.nnz, T.ia.p, tj.p, ta.p, T.ja.p, T.a.p, s);
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));
}

void permute(const DCsr &A, const int *perm, const int *colpos, DCsr &B, hipStream_t s) {
  B.release();
  const int n = A.nrows;
  B.nrows = n;
  B.ncols = A.ncols;
  B.nnz = A.nnz;
  B.ia.alloc((size_t)n + 1);
  B.ja.alloc((size_t)A.nnz);
  B.a.alloc((size_t)A.nnz);
  DVec<int> len((size_t)n);
  if (n) perm_len_k<<<(unsigned)((n + BLK - 1) / BLK), BLK, 0, s>>>(n, A.ia.p, perm, len.p);
  exclusive_scan(len.p, B.ia.p, n, s);
  if (A.nnz == 0) return;
  DVec<int> tj((size_t)A.nnz);
  DVec<double> ta((size_t)A.nnz);
  perm_copy_k<<<grid_for(((long long)n * 8 + BLK - 1) / BLK), BLK, 0, s>>>(n, A.ia.p, A.ja.p, A.a.p, perm, colpos,
                                                                             B.ia.p, tj.p, ta.p);
  sort_rows(n, A.nnz, B.ia.p, tj.p, ta.p, B.ja.p, B.a.p, s);
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));
}

void extract_rows(const DCsr &A, const int *rows, int nout, const int *colpos, DCsr &B, hipStream_t s) {
  B.release();
  B.nrows = nout;
  B.ncols = A.ncols;
  B.ia.alloc((size_t)nout + 1);
  DVec<int> len((size_t)nout);
  if (nout) perm_len_k<<<(unsigned)((nout + BLK - 1) / BLK), BLK, 0, s>>>(nout, A.ia.p, rows, len.p);
  exclusive_scan(len.p, B.ia.p, nout, s);
  long long total = 0;
  d2h(&total, B.ia.p + nout, sizeof(long long), s);
  MI_HIP(hipStreamSynchronize(s));
  B.nnz = total;
  B.ja.alloc((size_t)total);
  B.a.alloc((size_t)total);
  if (total == 0) return;
  DVec<int> tj((size_t)total);
  DVec<double> ta((size_t)total);
  perm_copy_k<<<grid_for(((long long)nout * 8 + BLK - 1) / BLK), BLK, 0, s>>>(nout, A.ia.p, A.ja.p, A.a.p, rows, colpos,
                                                                                B.ia.p, tj.p, ta.p);
  sort_rows(nout, total, B.ia.p, tj.p, ta.p, B.ja.p, B.a.p, s);
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));
}

void ilu_diag_positions(const DCsr &A, long long *dpos, hipStream_t s) {
  if (A.nrows) ilu_dpos_k<<<(unsigned)((A.nrows + BLK - 1) / BLK), BLK, 0, s>>>(A.nrows, A.ia.p, A.ja.p, dpos);
  MI_HIP(hipGetLastError());
}
void ilu_factor_level(DCsr &LU, const long long *dpos, const int *rows, int nrows, hipStream_t s) {
  if (nrows) ilu_factor_k<<<(unsigned)((nrows + BLK - 1) / BLK), BLK, 0, s>>>(nrows, rows, LU.ia.p, LU.ja.p, LU.a.p, dpos);
}
void ilu_lower_level(const DCsr &LU, const long long *dpos, const int *rows, int nrows, const double *b, double *y,
                     hipStream_t s) {
  (void)dpos;
  if (nrows) ilu_lower_k<<<(unsigned)((nrows + BLK - 1) / BLK), BLK, 0, s>>>(nrows, rows, LU.ia.p, LU.ja.p, LU.a.p, b, y);
}
void ilu_upper_level(const DCsr &LU, const long long *dpos, const int *rows, int nrows, const double *y, double *x,
                     hipStream_t s) {
  if (nrows)
    ilu_upper_k<<<(unsigned)((nrows + BLK - 1) / BLK), BLK, 0, s>>>(nrows, rows, LU.ia.p, LU.ja.p, LU.a.p, dpos, y, x);
}
void ilu_lower_jacobi(const DCsr &LU, const long long *dpos, const double *b, const double *in, double *out,
                      hipStream_t s) {
  (void)dpos;
  const int n = LU.nrows;
  if (n) ilu_lower_jac_k<<<(unsigned)((n + BLK - 1) / BLK), BLK, 0, s>>>(n, LU.ia.p, LU.ja.p, LU.a.p, b, in, out);
}
void ilu_upper_jacobi(const DCsr &LU, const long long *dpos, const double *b, const double *in, double *out,
                      hipStream_t s) {
  const int n = LU.nrows;
  if (n) ilu_upper_jac_k<<<(unsigned)((n + BLK - 1) / BLK), BLK, 0, s>>>(n, LU.ia.p, LU.ja.p, LU.a.p, dpos, b, in, out);
}

// ---------------------------------------------------------------- distributed setup on the device
// (amg_setup_dist.cpp, BoomerAMG::build_distributed_device).  A rank's piece of a level lives in an EXTENDED index
// space [remote ids below the own range | own range | remote ids above], ascending in the global id, so that stored
// order, discovery order and every floating-point sum are those of the single-rank kernels above, which then run
// unchanged on the extended matrices.  What is new here are the changes of index space (monotone three-piece column
// maps with small tables for the remote part), stacking row blocks, and the PMIS rounds restricted to the own rows.
namespace {

__device__ __forceinline__ int ext_map_col(const ExtColMap &m, int c) {
  if (c < m.nb_old) return m.below ? m.below[c] : -1;
  const int o = c - m.nb_old;
  if (o < m.n_own) {
    if (!m.keep_own) return -1;
    return m.own_tab ? m.own_tab[o] : m.own_new0 + o;
  }
  return m.above ? m.above[o - m.n_own] : -1;
}

// 8 lanes per output row; FILL = false: kept entries per row, FILL = true: copy them (stored order kept)
template <bool FILL>
__global__ __launch_bounds__(BLK) void select_rows_k(int nout, const int *__restrict__ rows, int row0,
                                                     const long long *__restrict__ ia, const int *__restrict__ ja,
                                                     const double *__restrict__ a, ExtColMap m, int *__restrict__ len,
                                                     const long long *__restrict__ bia, int *__restrict__ bj,
                                                     double *__restrict__ ba) {
  const long long q = (bid() * BLK + threadIdx.x) / 8;
  const int lane = threadIdx.x % 8, gbase = (threadIdx.x & 63) - lane;
  if (q >= nout) return;  // whole groups leave together
  const int i = rows ? rows[q] : row0 + (int)q;
  const long long s0 = ia[i];
  const int n = (int)(ia[i + 1] - s0);
  long long d0 = FILL ? bia[q] : 0;
  int kept = 0;
  for (int e0 = 0; e0 < n; e0 += 8) {
    const int e = e0 + lane;
    int c = -1;
    if (e < n) c = ext_map_col(m, ja[s0 + e]);
    const unsigned long long bal = __ballot(c >= 0);
    const unsigned mine = (unsigned)((bal >> gbase) & 0xffull);
    if (FILL && c >= 0) {
      const long long w = d0 + __popc(mine & ((1u << lane) - 1u));
      bj[w] = c;
      ba[w] = a[s0 + e];
    }
    const int add = __popc(mine);
    d0 += add;
    kept += add;
  }
  if (!FILL && lane == 0) len[q] = kept;
}

__global__ __launch_bounds__(BLK) void shift_ia_k(long long n1, const long long *__restrict__ src, long long add,
                                                  long long *__restrict__ dst) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n1) dst[i] = src[i] + add;
}

__global__ __launch_bounds__(BLK) void hstack_len_k(int n, const long long *__restrict__ aia, const long long *__restrict__ bia,
                                                    int *__restrict__ len) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n) len[i] = (int)(aia[i + 1] - aia[i]) + (int)(bia[i + 1] - bia[i]);
}
__global__ __launch_bounds__(BLK) void hstack_copy_k(int n, const long long *__restrict__ aia, const int *__restrict__ aja,
                                                     const double *__restrict__ aa, const long long *__restrict__ bia,
                                                     const int *__restrict__ bja, const double *__restrict__ ba, int shift,
                                                     const long long *__restrict__ cia, int *__restrict__ cj,
                                                     double *__restrict__ ca) {
  const long long i = (bid() * BLK + threadIdx.x) / 8;
  const int lane = threadIdx.x % 8;
  if (i >= n) return;
  const long long a0 = aia[i], b0 = bia[i], c0 = cia[i];
  const int la = (int)(aia[i + 1] - a0), lb = (int)(bia[i + 1] - b0);
  for (int e = lane; e < la; e += 8) {
    cj[c0 + e] = aja[a0 + e];
    ca[c0 + e] = aa[a0 + e];
  }
  for (int e = lane; e < lb; e += 8) {
    cj[c0 + la + e] = bja[b0 + e] + shift;
    ca[c0 + la + e] = ba[b0 + e];
  }
}

// ---- PMIS rounds on the rows [row0, row0 + n) of the extended strength graph; vectors span the extended space
__global__ __launch_bounds__(BLK) void pmisd_count_k(int n, int row0, const long long *__restrict__ sia,
                                                     const int *__restrict__ sja, int *__restrict__ cnt) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n) return;
  for (long long k = sia[row0 + i]; k < sia[row0 + i + 1]; k++) atomicAdd(&cnt[sja[k]], 1);
}
__global__ __launch_bounds__(BLK) void pmisd_init_k(int n, int row0, long long gid0, const long long *__restrict__ sia,
                                                    const int *__restrict__ incoming, int seed0,
                                                    double *__restrict__ measure, int *__restrict__ cf,
                                                    int *__restrict__ undecided) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n) return;
  const long long x = row0 + i;
  // element gid0 + i of the ONE global Park-Miller stream (see pmis_init_k)
  unsigned long long base = 16807ULL, acc = (unsigned long long)seed0, e = (unsigned long long)(gid0 + i) + 1ULL;
  while (e) {
    if (e & 1ULL) acc = mulmod31(acc, base);
    base = mulmod31(base, base);
    e >>= 1;
  }
  const double m = (double)incoming[x] + (double)(int)acc / 2147483647;
  int c = 0;
  if (sia[x + 1] == sia[x])
    c = SF_PT;
  else if (m < 1.0)
    c = F_PT;
  cf[x] = c;
  measure[x] = c ? 0.0 : m;
  if (c == 0) atomicAdd(undecided, 1);
}
__global__ __launch_bounds__(BLK) void pmisd_compare_k(int n, int row0, const long long *__restrict__ sia,
                                                       const int *__restrict__ sja, const int *__restrict__ cf,
                                                       const double *__restrict__ measure,
                                                       signed char *__restrict__ tmp) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n) return;
  const long long x = row0 + i;
  if (cf[x] != 0) return;
  const double mi = measure[x];
  bool lose = false;
  for (long long k = sia[x]; k < sia[x + 1]; k++) {
    const int j = sja[k];
    if (cf[j] != 0) continue;
    const double mj = measure[j];
    if (mi > mj)
      tmp[j] = 0;
    else if (mj > mi)
      lose = true;
  }
  if (lose) tmp[x] = 0;
}
__global__ __launch_bounds__(BLK) void pmisd_select_k(int n, int row0, int *__restrict__ cf,
                                                      const signed char *__restrict__ tmp) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n && cf[row0 + i] == 0 && tmp[row0 + i]) cf[row0 + i] = C_PT;
}
__global__ __launch_bounds__(BLK) void pmisd_fpoints_k(int n, int row0, const long long *__restrict__ sia,
                                                       const int *__restrict__ sja, int *__restrict__ cf,
                                                       int *__restrict__ undecided) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n) return;
  const long long x = row0 + i;
  if (cf[x] != 0) return;
  for (long long k = sia[x]; k < sia[x + 1]; k++)
    if (cf[sja[k]] == C_PT) {
      cf[x] = F_PT;
      return;
    }
  atomicAdd(undecided, 1);
}

template <class T>
__global__ __launch_bounds__(BLK) void gather_k(int n, const T *__restrict__ src, const int *__restrict__ idx, int shift,
                                                T *__restrict__ dst) {
  const long long k = bid() * BLK + threadIdx.x;
  if (k < n) dst[k] = src[idx[k] + shift];
}
__global__ __launch_bounds__(BLK) void scatter_add_k(int n, int *__restrict__ dst, const int *__restrict__ idx, int shift,
                                                     const int *__restrict__ v) {
  const long long k = bid() * BLK + threadIdx.x;
  if (k < n && v[k] != 0) atomicAdd(&dst[idx[k] + shift], v[k]);
}
__global__ __launch_bounds__(BLK) void scatter_zero_k(int n, signed char *__restrict__ dst, const int *__restrict__ idx,
                                                      int shift, const signed char *__restrict__ v) {
  const long long k = bid() * BLK + threadIdx.x;
  if (k < n && v[k] == 0) dst[idx[k] + shift] = 0;
}
__global__ __launch_bounds__(BLK) void is_c_k(int n, const int *__restrict__ cf, int *__restrict__ flag) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n) flag[i] = (cf[i] == C_PT);
}
__global__ __launch_bounds__(BLK) void fill_cgid_k(int n, const int *__restrict__ cf, const long long *__restrict__ rank,
                                                   long long first, long long *__restrict__ cg) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n) cg[i] = (cf[i] == C_PT) ? first + rank[i] : -1;
}
__global__ __launch_bounds__(BLK) void rows_outside_k(int n, const long long *__restrict__ ia, const int *__restrict__ ja,
                                                      int c0, int c1, int *__restrict__ flag) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n) return;
  int f = 0;
  for (long long k = ia[i]; k < ia[i + 1] && !f; k++) f = (ja[k] < c0 || ja[k] >= c1);
  flag[i] = f;
}
__global__ __launch_bounds__(BLK) void compact_fill_k(int n, const int *__restrict__ flag, const long long *__restrict__ pos,
                                                      int *__restrict__ list) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n && flag[i]) list[pos[i]] = (int)i;
}
__global__ __launch_bounds__(BLK) void mark_cols_k(long long nnz, const int *__restrict__ ja, unsigned char *__restrict__ used) {
  const long long k = bid() * BLK + threadIdx.x;
  if (k < nnz) used[ja[k]] = 1;
}
__global__ __launch_bounds__(BLK) void cfirst_pos_k(int n, const int *__restrict__ cf, const long long *__restrict__ crank,
                                                    int nc, int *__restrict__ pos, int *__restrict__ perm) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i >= n) return;
  const int q = (cf[i] == C_PT) ? (int)crank[i] : nc + (int)(i - crank[i]);
  pos[i] = q;
  perm[q] = (int)i;
}
__global__ __launch_bounds__(BLK) void remap_cols_k(long long nnz, int *__restrict__ ja, ExtColMap m, int *__restrict__ bad) {
  const long long k = bid() * BLK + threadIdx.x;
  if (k >= nnz) return;
  const int c = ext_map_col(m, ja[k]);
  if (c < 0) *bad = 1;
  ja[k] = c;
}
__global__ __launch_bounds__(BLK) void add_const_k(int n, int *__restrict__ v, int add) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n) v[i] += add;
}

}  // namespace

void select_rows(const DCsr &A, const int *rows, int row0, int nout, const ExtColMap &m, int new_ncols, bool sort,
                 DCsr &B, hipStream_t s) {
  B.release();
  B.nrows = nout;
  B.ncols = new_ncols;
  B.ia.alloc((size_t)nout + 1);
  DVec<int> len((size_t)nout);
  const dim3 grid = grid_for(((long long)nout * 8 + BLK - 1) / BLK);
  if (nout) select_rows_k<false><<<grid, BLK, 0, s>>>(nout, rows, row0, A.ia.p, A.ja.p, A.a.p, m, len.p, nullptr, nullptr, nullptr);
  exclusive_scan(len.p, B.ia.p, nout, s);
  long long total = 0;
  d2h(&total, B.ia.p + nout, sizeof(long long), s);
  MI_HIP(hipStreamSynchronize(s));
  B.nnz = total;
  B.ja.alloc((size_t)total);
  B.a.alloc((size_t)total);
  if (total == 0) return;
  if (!sort) {
    select_rows_k<true><<<grid, BLK, 0, s>>>(nout, rows, row0, A.ia.p, A.ja.p, A.a.p, m, nullptr, B.ia.p, B.ja.p, B.a.p);
  } else {
    DVec<int> tj((size_t)total);
    DVec<double> ta((size_t)total);
    select_rows_k<true><<<grid, BLK, 0, s>>>(nout, rows, row0, A.ia.p, A.ja.p, A.a.p, m, nullptr, B.ia.p, tj.p, ta.p);
    sort_rows(nout, total, B.ia.p, tj.p, ta.p, B.ja.p, B.a.p, s);
    MI_HIP(hipGetLastError());
    MI_HIP(hipStreamSynchronize(s));  // tj / ta are released on return
  }
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));
}

void remap_columns(DCsr &A, const ExtColMap &m, int new_ncols, hipStream_t s) {
  DVec<int> bad(1);
  MI_HIP(hipMemsetAsync(bad.p, 0, sizeof(int), s));
  if (A.nnz) remap_cols_k<<<grid_for((A.nnz + BLK - 1) / BLK), BLK, 0, s>>>(A.nnz, A.ja.p, m, bad.p);
  int hbad = 0;
  d2h(&hbad, bad.p, sizeof(int), s);
  MI_HIP(hipStreamSynchronize(s));
  MI_REQUIRE(hbad == 0, "remap_columns: a column has no image in the new index space");
  A.ncols = new_ncols;
}

void vconcat(const DCsr *const *parts, int nparts, DCsr &C, hipStream_t s) {
  C.release();
  long long rows = 0, nnz = 0;
  int ncols = 0;
  for (int q = 0; q < nparts; q++) {
    rows += parts[q]->nrows;
    nnz += parts[q]->nnz;
    ncols = std::max(ncols, parts[q]->ncols);
  }
  MI_REQUIRE(rows < 2147483647LL, "vconcat: more than 2^31 rows");
  C.nrows = (int)rows;
  C.ncols = ncols;
  C.nnz = nnz;
  C.ia.alloc((size_t)rows + 1);
  C.ja.alloc((size_t)nnz);
  C.a.alloc((size_t)nnz);
  long long r = 0, e = 0;
  for (int q = 0; q < nparts; q++) {
    const DCsr &P = *parts[q];
    if (P.nrows) {
      const long long n1 = (long long)P.nrows + (q == nparts - 1 ? 1 : 0);
      shift_ia_k<<<grid_for((n1 + BLK - 1) / BLK), BLK, 0, s>>>(n1, P.ia.p, e, C.ia.p + r);
    }
    if (P.nnz) {
      MI_HIP(hipMemcpyAsync(C.ja.p + e, P.ja.p, (size_t)P.nnz * sizeof(int), hipMemcpyDeviceToDevice, s));
      MI_HIP(hipMemcpyAsync(C.a.p + e, P.a.p, (size_t)P.nnz * sizeof(double), hipMemcpyDeviceToDevice, s));
    }
    r += P.nrows;
    e += P.nnz;
  }
  // the closing row pointer when the last part has no rows (or there are no parts)
  if (nparts == 0 || parts[nparts - 1]->nrows == 0)
    MI_HIP(hipMemcpyAsync(C.ia.p + rows, &nnz, sizeof(long long), hipMemcpyHostToDevice, s));
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));
}

void hstack(const DCsr &A, const DCsr &B, DCsr &C, hipStream_t s) {
  MI_REQUIRE(A.nrows == B.nrows, "hstack: row counts differ");
  C.release();
  const int n = A.nrows;
  C.nrows = n;
  C.ncols = A.ncols + B.ncols;
  C.nnz = A.nnz + B.nnz;
  C.ia.alloc((size_t)n + 1);
  C.ja.alloc((size_t)C.nnz);
  C.a.alloc((size_t)C.nnz);
  DVec<int> len((size_t)n);
  if (n) hstack_len_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, A.ia.p, B.ia.p, len.p);
  exclusive_scan(len.p, C.ia.p, n, s);
  if (n && C.nnz)
    hstack_copy_k<<<grid_for(((long long)n * 8 + BLK - 1) / BLK), BLK, 0, s>>>(n, A.ia.p, A.ja.p, A.a.p, B.ia.p, B.ja.p, B.a.p,
                                                                                A.ncols, C.ia.p, C.ja.p, C.a.p);
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));
}

void pmis_dist_counts(const DCsr &S, int row0, int n, int *cnt, hipStream_t s) {
  if (n) pmisd_count_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, row0, S.ia.p, S.ja.p, cnt);
  MI_HIP(hipGetLastError());
}
int pmis_dist_init(const DCsr &S, int row0, int n, long long gid0, int seed, const int *cnt, double *measure, int *cf,
                   int *counter, hipStream_t s) {
  MI_HIP(hipMemsetAsync(counter, 0, sizeof(int), s));
  if (n)
    pmisd_init_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, row0, gid0, S.ia.p, cnt, seed ? seed : 13579, measure,
                                                                          cf, counter);
  int undecided = 0;
  d2h(&undecided, counter, sizeof(int), s);
  MI_HIP(hipStreamSynchronize(s));
  return undecided;
}
void pmis_dist_compare(const DCsr &S, int row0, int n, int ne, const int *cf, const double *measure, signed char *tmp,
                       hipStream_t s) {
  if (ne) MI_HIP(hipMemsetAsync(tmp, 1, (size_t)ne, s));
  if (n) pmisd_compare_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, row0, S.ia.p, S.ja.p, cf, measure, tmp);
  MI_HIP(hipGetLastError());
}
void pmis_dist_select(int row0, int n, int *cf, const signed char *tmp, hipStream_t s) {
  if (n) pmisd_select_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, row0, cf, tmp);
  MI_HIP(hipGetLastError());
}
int pmis_dist_fpoints(const DCsr &S, int row0, int n, int *cf, int *counter, hipStream_t s) {
  MI_HIP(hipMemsetAsync(counter, 0, sizeof(int), s));
  if (n) pmisd_fpoints_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, row0, S.ia.p, S.ja.p, cf, counter);
  int undecided = 0;
  d2h(&undecided, counter, sizeof(int), s);
  MI_HIP(hipStreamSynchronize(s));
  return undecided;
}

void gather_elems(const void *src, const int *idx, int shift, int n, int elem_bytes, void *dst, hipStream_t s) {
  if (n == 0) return;
  const dim3 grid = grid_for(((long long)n + BLK - 1) / BLK);
  if (elem_bytes == 1)
    gather_k<signed char><<<grid, BLK, 0, s>>>(n, (const signed char *)src, idx, shift, (signed char *)dst);
  else if (elem_bytes == 4)
    gather_k<int><<<grid, BLK, 0, s>>>(n, (const int *)src, idx, shift, (int *)dst);
  else if (elem_bytes == 8)
    gather_k<long long><<<grid, BLK, 0, s>>>(n, (const long long *)src, idx, shift, (long long *)dst);
  else
    MI_REQUIRE(false, "gather_elems: element size");
  MI_HIP(hipGetLastError());
}
void scatter_add_int(int *dst, const int *idx, int shift, const int *v, int n, hipStream_t s) {
  if (n) scatter_add_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, dst, idx, shift, v);
  MI_HIP(hipGetLastError());
}
void scatter_zero_flags(signed char *dst, const int *idx, int shift, const signed char *v, int n, hipStream_t s) {
  if (n) scatter_zero_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, dst, idx, shift, v);
  MI_HIP(hipGetLastError());
}

namespace {
__global__ __launch_bounds__(BLK) void ints_to_i8_k(long long n, const int *__restrict__ in, signed char *__restrict__ out) {
  const long long i = bid() * BLK + threadIdx.x;
  if (i < n) out[i] = (signed char)in[i];
}
}  // namespace
void ints_to_i8(const int *in, long long n, signed char *out, hipStream_t s) {
  if (n) ints_to_i8_k<<<grid_for((n + BLK - 1) / BLK), BLK, 0, s>>>(n, in, out);
  MI_HIP(hipGetLastError());
}
long long count_c_points(const int *cf, int n, DVec<long long> &rank, hipStream_t s) {
  rank.alloc((size_t)n + 1);
  DVec<int> flag((size_t)n);
  if (n) is_c_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, cf, flag.p);
  exclusive_scan(flag.p, rank.p, n, s);
  long long nc = 0;
  d2h(&nc, rank.p + n, sizeof(long long), s);
  MI_HIP(hipStreamSynchronize(s));
  return nc;
}
void fill_coarse_ids(const int *cf, const long long *rank, int n, long long first, long long *cg, hipStream_t s) {
  if (n) fill_cgid_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, cf, rank, first, cg);
  MI_HIP(hipGetLastError());
}
void cfirst_order(const int *cf, const long long *crank, int n, int nc, int *pos, int *perm, hipStream_t s) {
  if (n) cfirst_pos_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, cf, crank, nc, pos, perm);
  MI_HIP(hipGetLastError());
}
int rows_with_columns_outside(const DCsr &P, int c0, int c1, std::vector<int> &rows_host, hipStream_t s) {
  const int n = P.nrows;
  rows_host.clear();
  if (n == 0) return 0;
  DVec<int> flag((size_t)n);
  DVec<long long> pos((size_t)n + 1);
  rows_outside_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, P.ia.p, P.ja.p, c0, c1, flag.p);
  exclusive_scan(flag.p, pos.p, n, s);
  long long cnt = 0;
  d2h(&cnt, pos.p + n, sizeof(long long), s);
  MI_HIP(hipStreamSynchronize(s));
  if (cnt == 0) return 0;
  DVec<int> list((size_t)cnt);
  compact_fill_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, flag.p, pos.p, list.p);
  rows_host.resize((size_t)cnt);
  d2h(rows_host.data(), list.p, (size_t)cnt * sizeof(int), s);
  MI_HIP(hipStreamSynchronize(s));
  return (int)cnt;
}
void mark_used_columns(const DCsr &A, DVec<unsigned char> &used, hipStream_t s) {
  used.alloc((size_t)A.ncols);
  if (A.ncols) MI_HIP(hipMemsetAsync(used.p, 0, (size_t)A.ncols, s));
  if (A.nnz) mark_cols_k<<<grid_for((A.nnz + BLK - 1) / BLK), BLK, 0, s>>>(A.nnz, A.ja.p, used.p);
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));
}
void add_to_ints(int *v, int n, int add, hipStream_t s) {
  if (n && add) add_const_k<<<grid_for(((long long)n + BLK - 1) / BLK), BLK, 0, s>>>(n, v, add);
  MI_HIP(hipGetLastError());
}

}  // namespace sk
}  // namespace mi
