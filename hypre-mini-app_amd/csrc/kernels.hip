// Hand-written HIP kernels for gfx950 (MI355X, wave64, 256 CUs in 8 XCDs).
//
// Everything on this path is HBM-bandwidth bound sparse/vector work (arithmetic
// intensity ~0.13 flop/B): no MFMA.  The rules that matter are coalesced 16-byte
// streaming loads of the matrix, LDS staging of the per-entry products so that
// short rows reduce without divergence, __shfl wave reductions, and a
// workgroup->row-range map that keeps each XCD's L2 on its own slice of x.
#include <algorithm>

#include "kernels.hpp"

#include <cstring>
#include "profile.hpp"

namespace mi {
namespace k {

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an
// L2).  Remap so that XCD g walks the contiguous row-block range
// [g*chunk, (g+1)*chunk): neighbouring rows (and the x planes they gather) stay
// in one L2.  Speed only -- any placement gives the same result.
// Streams that are read exactly once per kernel in whole cache lines per load instruction (matrix values
// and column ids of the row-block SpMV) use non-temporal loads so that they do not evict the gathered
// vector from L2.  (Not in the GS kernel: its per-row segments share lines between load instructions, and
// non-temporal lines are not kept in L1 -- measured 35 % slower.)
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef int i2_t __attribute__((ext_vector_type(2)));
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
typedef unsigned char uc2_t __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ T nt_load(const T *p) {
  return __builtin_nontemporal_load(p);
}

// The tile kernels (spmv_stream_xc, gs_tile_k) read their matrix stream (values or value indices, 16-bit column words)
// and their column lists exactly once, in whole lines per load instruction: non-temporal loads too, so that what stays
// in L2 is the gathered vector, which neighbouring tiles share.  A/B at 512^3 on one box (profiles/r03_ab_stream_nt.txt):
// stream alone -0.6 % per solve, with the lists -1 %, level-0 SpMV 2.77 -> 2.73 ms; same bits.  -DMI_STREAM_NT=0 /
// -DMI_LIST_NT=0 build the plain loads.  (Also tried, no effect either way: non-temporal loads of the per-row vectors
// f / d / b, non-temporal stores of the swept vector.)
#if !defined(MI_STREAM_NT) || MI_STREAM_NT
#define STREAM_LOAD(p) nt_load(p)
#else
#define STREAM_LOAD(p) (*(p))
#endif
#if !defined(MI_LIST_NT) || MI_LIST_NT
#define LIST_LOAD(p) nt_load(p)
#else
#define LIST_LOAD(p) (*(p))
#endif

// Timing ablations of the tile kernels (profiles/debug/ablate_table.py; results are WRONG with any bit set, never in a
// shipped build): -DMI_ABLATE=<bits>  1: every gather reads one of x's first 64 entries (the list is still loaded)
// 2: no column-list load  4: no matrix-stream loads  8: no LDS product phase / row sums (SpMV), no 8x8 sweep (GS)
#ifndef MI_ABLATE
#define MI_ABLATE 0
#endif

__device__ __forceinline__ int xcd_remap(int bid, int chunk) { return (bid & 7) * chunk + (bid >> 3); }

// ---------------------------------------------------------------------------
// LDS-staged row-block SpMV ("CSR-stream") with fused epilogues.
//   phase 1: the workgroup streams its contiguous slice of (col,val) with
//            16-byte/8-byte coalesced loads, gathers x, writes products to LDS
//   phase 2: G lanes per row (G = 1..64, uniform per workgroup) sum the row's
//            products out of LDS; G > 1 finishes with __shfl_down
// A block that holds exactly one row longer than the tile takes the
// whole-workgroup path instead.
// EPI 0: y = alpha*s + beta*b          (matvec / residual / restrict / prolong)
// EPI 1: masked Jacobi  y = x_i + w*(f - offc - s)/d on selected rows, else x_i
// ---------------------------------------------------------------------------
struct EpiArgs {
  double alpha, beta;      // EPI 0
  const double *b;         // EPI 0: b ; EPI 1: f
  const double *offc;      // EPI 1 (nullable)
  const double *d;         // EPI 1
  const signed char *cf;   // EPI 1 (nullable)
  int points;              // EPI 1
  const int *rowmap;       // EPI 0 (nullable): stored row r is row rowmap[r] of y and b (DevCSR::rowmap)
  const double *b_lo;      // EPI 0 (nullable): rows < b_split take their b entry from here (a composite right-hand
  int b_split;             //   side: BoomerAMG's residual after a zero-guess sweep, f for the C rows, f - A_FC u_C after)
};
__device__ __forceinline__ double epi_b(const EpiArgs &e, int ro) { return ((e.b_lo && ro < e.b_split) ? e.b_lo : e.b)[ro]; }

template <int EPI>
__device__ __forceinline__ void epilogue(int r, double s, const double *__restrict__ x, double *__restrict__ y,
                                         const EpiArgs &e) {
  if (EPI == 0) {
    // y is written once and next read by another kernel: keep it out of L2's way
    const int ro = e.rowmap ? e.rowmap[r] : r;
    __builtin_nontemporal_store((e.beta == 0.0) ? e.alpha * s : e.alpha * s + e.beta * epi_b(e, ro), y + ro);
  } else {
    const double xi = x[r];
    double out = xi;
    const bool sel = (e.points == 0) || (e.cf == nullptr) || (e.cf[r] == e.points);
    const double d = e.d[r];
    if (sel && d != 0.0) {
      double res = e.b[r] - s;
      if (e.offc) res -= e.offc[r];
      out = xi + e.alpha * res / d;
    }
    y[r] = out;
  }
}

// TAG only names the instantiation: TAG 1 = the level-0 operator, so that the
// rocprofv3 kernel statistics carry a row for exactly the launches bench.py times
template <int EPI, int TAG>
__global__ __launch_bounds__(SPMV_BLOCK) void spmv_stream(int nb, int xchunk, const int *__restrict__ rb,
                                                          const int *__restrict__ ia, const int *__restrict__ ja,
                                                          const double *__restrict__ av, const double *__restrict__ x,
                                                          double *__restrict__ y, EpiArgs e) {
  __shared__ double prod[SPMV_TILE];
  const int blk = xcd_remap(blockIdx.x, xchunk);
  if (blk >= nb) return;
  const int tid = threadIdx.x;
  const int r0 = rb[blk], r1 = rb[blk + 1];
  const int base = ia[r0], end = ia[r1];
  if (end - base >= SPMV_TILE) {
    // one long row: every lane strides over it, two-level reduction
    double s = 0.0;
    for (int k = base + tid; k < end; k += SPMV_BLOCK) s += av[k] * x[ja[k]];
    s = wave_sum(s);
    if ((tid & 63) == 0) prod[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) epilogue<EPI>(r0, prod[0] + prod[1] + prod[2] + prod[3], x, y, e);
    return;
  }
  // phase 1: aligned pairs -> double2 / int2 loads (arrays carry 2 pad entries)
  const int base_al = base & ~1;
  const int cnt = end - base_al;  // <= SPMV_TILE
  for (int k = 2 * tid; k < cnt; k += 2 * SPMV_BLOCK) {
    const d2_t v = nt_load(reinterpret_cast<const d2_t *>(av + base_al + k));
    const i2_t c = nt_load(reinterpret_cast<const i2_t *>(ja + base_al + k));
    const bool ok0 = (base_al + k >= base);
    const bool ok1 = (base_al + k + 1 < end);
    const double x0 = ok0 ? x[c.x] : 0.0;
    const double x1 = ok1 ? x[c.y] : 0.0;
    prod[k] = v.x * x0;
    if (k + 1 < SPMV_TILE) prod[k + 1] = v.y * x1;
  }
  __syncthreads();
  // phase 2
  const int nr = r1 - r0;
  int G = 1;
  while (G < 64 && nr * G * 2 <= SPMV_BLOCK) G <<= 1;
  const int lane = tid & (G - 1);
  for (int rr = tid / G; rr < nr; rr += SPMV_BLOCK / G) {
    const int r = r0 + rr;
    const int s0 = ia[r] - base_al, s1 = ia[r + 1] - base_al;
    double s = 0.0;
    for (int k = s0 + lane; k < s1; k += G) s += prod[k];
    for (int off = G >> 1; off > 0; off >>= 1) s += __shfl_down(s, off, G);
    if (lane == 0) epilogue<EPI>(r, s, x, y, e);
  }
}

// Same kernel with an LDS x cache, for levels with long rows (coarse AMG levels).
// There the plain version spends ~45 % of its time on the x gathers: 64
// consecutive entries hit 30-60 different lines, neighbouring rows re-request
// the same lines a moment later, and with 32 waves per CU the 32 KB L1 cannot
// hold them, so every lane becomes an L2 transaction.  Here each workgroup first
// gathers the sorted unique columns of its row block ONCE into LDS (adjacent
// lanes -> ascending addresses, well coalesced) and the entries then carry
// 16-bit block-local ids (which also shrinks the index stream from 4 to 2 B).
//
// Load order: the waves of this kernel wait ~84 % of their cycles with ~7 tiles resident per CU, i.e. a tile
// lives ~12 us, most of it in dependent round trips.  Vector-memory results are counted in issue order, so
// waiting for a LATE-issued short load also waits for every stream load issued before it.  The tile's row
// range, entry range and column-list range therefore arrive in one 32-byte (scalar) descriptor load, the
// column list is requested FIRST (its gathers are the longest chain), then the matrix stream and the row
// pointers of the row sums; the x gathers start as soon as the column ids are back, while the stream is still
// in flight.  (Worth ~0.5 % of the 512^3 solve over the rb -> ia -> stream / uptr -> ucols -> x order: the
// kernel is not bound by this chain alone -- VALU, LDS and the L1 gather path are each 25-30 % busy.)
// VAL8: the operator has a value dictionary (DevCSR::vidx / vlut): the stream is one byte per value, looked up in
// a 2 KB LDS copy of the table
template <int EPI, int TAG, bool VAL8, int BLOCK>
__global__ __launch_bounds__(BLOCK) void spmv_stream_xc(int nb, int xchunk, const int *__restrict__ tdesc,
                                                             const int *__restrict__ ia, const int *__restrict__ ja,
                                                             const double *__restrict__ av,
                                                             const int *__restrict__ ucols,
                                                             const unsigned short *__restrict__ lcol,
                                                             const double *__restrict__ x, double *__restrict__ y,
                                                             EpiArgs e, const unsigned char *__restrict__ vidx,
                                                             const double *__restrict__ vlut,
                                                             const unsigned short *__restrict__ ucode,
                                                             const int *__restrict__ ubase) {
  constexpr int TILE = 8 * BLOCK;  // TILE / TILE_WIDE
  __shared__ double prod[TILE];
  __shared__ double slut[VAL8 ? 256 : 1];
  double *xs = prod;
  const int blk = xcd_remap(blockIdx.x, xchunk);
  if (blk >= nb) return;
  const int tid = threadIdx.x;
  if (MI_ABLATE & 128) {  // launch only
    if (tid == 0 && blk == nb + 1) y[0] = 0.0;
    return;
  }
  // (the table entry waits in a register until the LDS writes before the first barrier: stored here, the kernel began
  // with a load, a wait and a write before anything else was requested)
  double lutv = 0.0;
  if (VAL8 && tid < 256) lutv = vlut[tid];
  const int4 d0 = reinterpret_cast<const int4 *>(tdesc)[2 * blk];
  const int4 d1 = reinterpret_cast<const int4 *>(tdesc)[2 * blk + 1];
  const int r0 = d0.x, r1 = d0.y, len = d0.w, u0 = d1.x, nu = d1.y;
  if (MI_ABLATE & 64) {  // launch + descriptor
    if (tid == 0 && r0 + len + u0 + nu + d1.z == -12345) y[0] = lutv;
    return;
  }
  // the tile's entry range starts at a 64-bit offset (operators beyond 2^31 entries); everything below is tile-local
  const long long base64 = ((long long)d1.z << 32) | (long long)(unsigned)d0.z;
  if (len >= TILE) {
    double s = 0.0;
    for (long long k = base64 + tid; k < base64 + len; k += BLOCK) s += av[k] * x[ja[k]];
    s = wave_sum(s);
    if ((tid & 63) == 0) prod[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
      double tot = prod[0] + prod[1] + prod[2] + prod[3];
      for (int wv = 4; wv < BLOCK / 64; wv++) tot += prod[wv];
      epilogue<EPI>(r0, tot, x, y, e);
    }
    return;
  }
  constexpr int NU = TILE / BLOCK;
  int uc[NU];
  int myblock = 0;
  const bool coded = ucode && d1.w >= 0;  // (uniform per tile) a tile with more than 64 blocks keeps a 4-byte list, in the side list
  if (ucode && !coded) ucols += (-d1.w - 1) - u0;
  if (coded) {  // block-coded list: 6-bit selector of one of the tile's <= 64 column blocks + 10-bit offset
    myblock = LIST_LOAD(ubase + d1.w + (tid & 63));  // (padded: the 64 ints behind any tile's first block exist)
#pragma unroll
    for (int q = 0; q < NU; q++) {
      const int k = tid + q * BLOCK;
      uc[q] = 0;
      if (k < nu) uc[q] = LIST_LOAD(ucode + u0 + k);
    }
  } else {
#pragma unroll
    for (int q = 0; q < NU; q++) {
      const int k = tid + q * BLOCK;
      // (nothing that needs the loaded id inside the branch: written as a select, the compiler put the id's 64-bit
      // extension there and with it a wait after every load -- up to eight SERIAL round trips per tile; check the ISA
      // when this line changes: no s_waitcnt between the list loads)
      uc[q] = 0;
      if (!(MI_ABLATE & 2) && k < nu) uc[q] = LIST_LOAD(ucols + u0 + k);
    }
  }
  const long long base_al64 = base64 & ~1LL;
  const int base = (int)(base64 - base_al64), base_al = 0;  // tile-local: the aligned start is 0, the first entry 0 or 1
  const int end = base + len;
  const int cnt = end;
  const unsigned ia_off = (unsigned)base_al64;  // low word: (unsigned)ia[row] - ia_off = the row's tile-local offset
  av += base_al64;
  lcol += base_al64;
  if (VAL8) vidx += base_al64;
  constexpr int NIT = TILE / (2 * BLOCK);
  d2_t vv[NIT];
  unsigned cw[NIT];  // two 16-bit column words, as loaded
  unsigned vw[NIT];  // two value indices, as loaded, one register each (nothing is unpacked or packed inside the
                     // branches: that makes the compiler wait for each load where it stands)
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int k = 2 * tid + it * 2 * BLOCK;
    cw[it] = 0;
    vw[it] = 0;
    if (MI_ABLATE & 4) vv[it] = d2_t{1.0, 1.0};
    if (!(MI_ABLATE & 4) && k < cnt) {
      if (VAL8)
        vw[it] = STREAM_LOAD(reinterpret_cast<const unsigned short *>(vidx + base_al + k));
      else
        vv[it] = STREAM_LOAD(reinterpret_cast<const d2_t *>(av + base_al + k));
      cw[it] = STREAM_LOAD(reinterpret_cast<const unsigned *>(lcol + base_al + k));
    }
  }
  // the gathers: ALL of them in flight before the first LDS write, and before the loads below that depend on each other.
  // (Written as `if (k < nu) xs[k] = x[uc[q]]`, every load sat in a branch of its own with its wait and its LDS write:
  // up to eight serial round trips per tile.  Lanes beyond the list read x[0] -- uc is 0 there -- one cached sector.)
  if (coded) {  // (after the stream loads have been issued: the codes are waited for here)
#pragma unroll
    for (int q = 0; q < NU; q++) uc[q] = (__shfl(myblock, uc[q] >> 10, 64) << 10) | (uc[q] & 1023);
  }
  double xv[NU];
#pragma unroll
  for (int q = 0; q < NU; q++) xv[q] = x[(MI_ABLATE & 1) ? min(uc[q], 63) : uc[q]];
  // the entry range of the first row a thread sums (operators that the tile Gauss-Seidel kernel also sweeps have
  // <= BLOCK rows per tile: one row per thread; SpMV-only operators with short rows -- P, R, the residual
  // sub-operator -- get up to 4 x BLOCK rows to fill their tiles)
  const int nr = r1 - r0;
  int G = 1;
  while (G < 64 && nr * G * 2 <= BLOCK) G <<= 1;
  const int lane = tid & (G - 1);
  const int rr = tid / G;
  int s0 = 0, s1 = 0;
  double bpre = 0.0;  // EPI 0 with beta != 0: the b entry of this thread's row, requested with the other loads
  int ro = r0 + rr;   // EPI 0: where this thread's row lands in y (operators stored in another row order)
  if (rr < nr) {
    if (!(MI_ABLATE & 16)) {
      s0 = (int)((unsigned)ia[r0 + rr] - ia_off);
      s1 = (int)((unsigned)ia[r0 + rr + 1] - ia_off);
    }
    if (EPI == 0 && e.rowmap && lane == 0) ro = e.rowmap[r0 + rr];
    if (EPI == 0 && e.beta != 0.0 && lane == 0) bpre = epi_b(e, ro);
  }
  if (VAL8 && tid < 256) slut[tid] = lutv;  // visible after the barrier below
#pragma unroll
  for (int q = 0; q < NU; q++) {
    const int k = tid + q * BLOCK;
    if (k < nu) xs[k] = xv[q];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int k = 2 * tid + it * 2 * BLOCK;
    if (k < cnt) {
      const bool ok0 = (base_al + k >= base);
      const bool ok1 = (base_al + k + 1 < end);
      if (VAL8) {
        vv[it].x = slut[vw[it] & 0xffu];
        vv[it].y = slut[vw[it] >> 8];
      }
      vv[it].x = ok0 ? vv[it].x * xs[cw[it] & XC_ID_MASK] : 0.0;
      vv[it].y = ok1 ? vv[it].y * xs[(cw[it] >> 16) & XC_ID_MASK] : 0.0;
    }
  }
  if (MI_ABLATE & 8) {
    double s = (double)(s1 - s0);
#pragma unroll
    for (int it = 0; it < NIT; it++) s += vv[it].x + vv[it].y;
    if (rr < nr && lane == 0 && (!(MI_ABLATE & 32) || s == -12345.0)) __builtin_nontemporal_store(e.alpha * s + e.beta * bpre, y + ro);
    return;
  }
  __syncthreads();  // every x-cache read is done: the array becomes the product buffer
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int k = 2 * tid + it * 2 * BLOCK;
    if (k < cnt) {
      prod[k] = vv[it].x;
      prod[k + 1] = vv[it].y;
    }
  }
  __syncthreads();
  if (rr < nr) {
    double s = 0.0;
    for (int k = s0 + lane; k < s1; k += G) s += prod[k];
    for (int off = G >> 1; off > 0; off >>= 1) s += __shfl_down(s, off, G);
    if (lane == 0) {
      if (EPI == 0)
        __builtin_nontemporal_store((e.beta == 0.0) ? e.alpha * s : e.alpha * s + e.beta * bpre, y + ro);
      else
        epilogue<EPI>(r0 + rr, s, x, y, e);
    }
  }
  // tiles with more rows than threads (G == 1 there): the remaining rows, one lane each
  for (int r2 = tid + BLOCK; r2 < nr; r2 += BLOCK) {
    const int a0 = (int)((unsigned)ia[r0 + r2] - ia_off), a1 = (int)((unsigned)ia[r0 + r2 + 1] - ia_off);
    double s = 0.0;
    for (int k = a0; k < a1; k++) s += prod[k];
    epilogue<EPI>(r0 + r2, s, x, y, e);
  }
}

__global__ __launch_bounds__(256) void tile_desc_k(int nb, const int *__restrict__ rb, const long long *__restrict__ ia,
                                                   const int *__restrict__ uptr, int *__restrict__ desc,
                                                   const long long *__restrict__ bptr) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nb) return;
  const int r0 = rb[b], r1 = rb[b + 1];
  int4 *d = reinterpret_cast<int4 *>(desc) + 2 * (size_t)b;
  const int u0 = uptr ? uptr[b] : 0, u1 = uptr ? uptr[b + 1] : 0;
  const long long base = ia[r0], len = ia[r1] - base;
  // a single row longer than any tile (>= 2^31 entries it cannot be: such rows do not exist) keeps its true length
  d[0] = make_int4(r0, r1, (int)(unsigned)(base & 0xffffffffLL), (int)(len > 0x7fffffffLL ? 0x7fffffffLL : len));
  d[1] = make_int4(u0, u1 - u0, (int)(base >> 32), bptr ? (int)bptr[b] : 0);  // (>= 0: coded tile; < 0: -(offset in the side list) - 1)
}

// Block-coded column lists (DevCSR::ucode / ubase): one wave per tile walks the tile's sorted unique columns in steps of
// 64, a column opens a new block when its id >> 10 differs from its predecessor's.  FILL = false counts the blocks,
// FILL = true writes the block numbers behind bptr[tile] and the 16-bit codes.
template <bool FILL>
__global__ __launch_bounds__(256) void ucode_blocks_k(int ntiles, const int *__restrict__ uptr, const int *__restrict__ ucols,
                                                      int *__restrict__ nblk, const long long *__restrict__ bptr,
                                                      int *__restrict__ ubase, unsigned short *__restrict__ ucode,
                                                      int *__restrict__ maxblk, int *__restrict__ wide_list) {
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (tile >= ntiles) return;
  const int u0 = uptr[tile], u1 = uptr[tile + 1];
  if (FILL && bptr[tile] < 0) {  // a tile with more than 64 blocks: its 4-byte list moves to the compact side list
    const long long w0 = -bptr[tile] - 1;
    for (int k = u0 + lane; k < u1; k += 64) wide_list[w0 + (k - u0)] = ucols[k];
    return;
  }
  int run = 0, prev = -1;
  for (int k0 = u0; k0 < u1; k0 += 64) {
    const int k = k0 + lane;
    const int c = k < u1 ? ucols[k] : -1;
    const int blk = k < u1 ? (c >> 10) : -2;
    int left = __shfl_up(blk, 1, 64);
    if (lane == 0) left = prev;
    const bool isnew = k < u1 && blk != left;
    const unsigned long long m = __ballot(isnew);
    const int sel = run + __popcll(m & ((2ull << lane) - 1ull)) - 1;  // blocks opened up to and including this lane
    if (FILL) {
      if (isnew) ubase[bptr[tile] + sel] = blk;
      if (k < u1) ucode[k] = (unsigned short)(((sel & 63) << 10) | (c & 1023));
    }
    run += __popcll(m);
    prev = __shfl(blk, 63, 64);
  }
  if (!FILL && lane == 0) {
    nblk[tile] = run;
    atomicMax(maxblk, run);
  }
}

// compressed-row off-diagonal block: one lane per stored row (halo rows are few
// and short); MODE 0: y[row] += alpha*s ; MODE 1: y[row] = s
template <int MODE>
__global__ __launch_bounds__(256) void spmv_offd_k(int nrc, const int *__restrict__ rows, const int *__restrict__ ia,
                                                   const int *__restrict__ ja, const double *__restrict__ av,
                                                   const double *__restrict__ xext, double alpha,
                                                   double *__restrict__ y) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= nrc) return;
  double s = 0.0;
  for (int q = ia[k]; q < ia[k + 1]; q++) s += av[q] * xext[ja[q]];
  const int r = rows[k];
  if (MODE == 0)
    y[r] += alpha * s;
  else
    y[r] = s;
}

// ---------------------------------------------------------------------------
// Hybrid Gauss-Seidel: lane c owns the chunk of `chunk` consecutive rows
// [c*chunk, ...), sweeps it sequentially (forward and/or backward) with its own
// running values in LDS, and reads every other chunk's PRE-sweep value from
// u_old.  This is HYPRE's hybrid smoother with chunks in the role of its
// threads/ranks (par_relax.c) and is bitwise independent of
// scheduling because u_old is never written.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(GS_BLOCK) void gs_hybrid_k(int n, int chunk0, int nchunks, int chunk, const int *__restrict__ ia,
                                                        const int *__restrict__ ja, const double *__restrict__ av,
                                                        const signed char *__restrict__ cf, int points,
                                                        const double *__restrict__ dd, const double *__restrict__ f,
                                                        const double *__restrict__ offc,
                                                        const double *__restrict__ u_lo,
                                                        const double *__restrict__ u_hi, int split,
                                                        double *__restrict__ u_new, int fwd, int bwd, double w) {
#define UOLD(j) (((j) < split ? u_lo : u_hi)[(j)])
  extern __shared__ double ucur[];  // [chunk][GS_BLOCK]
  const int tid = threadIdx.x;
  const long long c = chunk0 + (long long)blockIdx.x * GS_BLOCK + tid;
  if (c >= nchunks) return;
  const int cs = (int)(c * chunk);
  const int len = min(chunk, n - cs);
  for (int t = 0; t < len; t++) ucur[t * GS_BLOCK + tid] = UOLD(cs + t);
  for (int dir = 0; dir < 2; dir++) {
    if (dir == 0 ? !fwd : !bwd) continue;
    for (int t = 0; t < len; t++) {
      const int tt = (dir == 0) ? t : len - 1 - t;
      const int i = cs + tt;
      if (points != 0 && cf != nullptr && cf[i] != points) continue;
      const double d = dd[i];
      if (d == 0.0) continue;
      double res = f[i];
      if (offc) res -= offc[i];
      const int k1 = ia[i + 1];
      for (int k = ia[i]; k < k1; k++) {
        const int j = ja[k];
        const unsigned o = (unsigned)(j - cs);
        const double v = (o < (unsigned)len) ? ucur[o * GS_BLOCK + tid] : UOLD(j);
        res -= av[k] * v;
      }
      ucur[tt * GS_BLOCK + tid] += w * res / d;
    }
  }
  for (int t = 0; t < len; t++) u_new[cs + t] = ucur[t * GS_BLOCK + tid];
#undef UOLD
}

// ---------------------------------------------------------------------------
// Hybrid Gauss-Seidel, cooperative form (the one that runs when chunk == 8).
// A group of LPC lanes (8/16/32/64, chosen from the level's mean row length)
// owns one chunk of R = 8 consecutive rows; a wave holds 64/LPC chunks, so for
// the 7-point fine level the wave's 64 lanes read 64 consecutive rows' entries.
//   phase A  every row of the chunk is loaded up front: lane g takes entries
//            g, g+LPC, ... of each row (adjacent lanes -> adjacent addresses),
//            and gathers u_old for the columns outside the chunk.  Nothing in
//            this phase depends on the sweep, so all loads are in flight at once.
//   phase B  the 8 rows are swept in order (forward and/or backward).  The
//            chunk's running values live one per lane (lane g holds row g) and
//            are read with __shfl; the row's dot product is a __shfl_xor
//            butterfly over the group.
// u_old is never written, so the result does not depend on scheduling.
// ---------------------------------------------------------------------------
// 64-bit DPP move (two dword moves); CTRL is a DPP control word
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// sum over an aligned group of LPC lanes, result in every lane of the group.
// quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140: cross-lane moves in the VALU, no LDS crossbar; only the
// 32- and 64-lane steps go through ds_bpermute.
template <int LPC>
__device__ __forceinline__ double group_sum(double v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  if (LPC >= 16) v += dpp_mov<0x140>(v);
  if (LPC >= 32) v += __shfl_xor(v, 16, 64);
  if (LPC >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

template <int LPC, int E>
__global__ __launch_bounds__(256) void gs_group_k(int n, int chunk0, int nchunks, const int *__restrict__ ia,
                                                  const int *__restrict__ ja, const double *__restrict__ av,
                                                  const signed char *__restrict__ cf, int points,
                                                  const double *__restrict__ dd, const double *__restrict__ f,
                                                  const double *__restrict__ offc,
                                                  const double *__restrict__ u_lo,
                                                  const double *__restrict__ u_hi, int split,
                                                  double *__restrict__ u_new, int fwd, int bwd, double w,
                                                  int zero_from) {
#define UOLD(j) (((j) < split ? u_lo : u_hi)[(j)])
  constexpr int R = 8;
  constexpr int CPW = 64 / LPC;
  const int lane = threadIdx.x & 63;
  const int g = lane & (LPC - 1);
  const int gbase = lane & ~(LPC - 1);
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long long c = chunk0 + wave * CPW + (lane / LPC);
  const bool live = c < nchunks;
  const int cs = live ? (int)(c * R) : 0;
  const int len = live ? min(R, n - cs) : 0;

  // owner lane g < len holds the state of row cs+g
  double myu = 0.0, myrhs = 0.0, myd = 0.0;
  bool mysel = false;
  int my_k0 = 0, my_k1 = 0;
  if (g < len) {
    const int i = cs + g;
    my_k0 = ia[i];
    const int a1 = ia[i + 1];
    const int mark = (points != 0 && cf != nullptr) ? (int)cf[i] : points;
    myu = (i >= zero_from) ? 0.0 : UOLD(i);
    myd = dd[i];
    myrhs = f[i];
    if (offc) myrhs -= offc[i];
    const bool rowsel = (mark == points);
    my_k1 = rowsel ? a1 : my_k0;  // unselected rows load nothing
    mysel = rowsel && myd != 0.0;
  }
  // one division per row, outside the sweep (the sweep's VALU work is what bounds this kernel)
  const double mywd = mysel ? w / myd : 0.0;
  // phase A: three waves of independent loads (row pointers + marker, entries,
  // gathers).  Per entry one double survives: a_ij*u_old[j] for a column outside
  // the chunk, a_ij itself for a column inside it; the in-chunk offset (or 15)
  // of the 8 rows is packed 4 bits each into one register per strip.
  int k0s[R], k1s[R];
  unsigned longmask = 0;
#pragma unroll
  for (int t = 0; t < R; t++) {
    k0s[t] = __shfl(my_k0, gbase + t, 64);
    k1s[t] = __shfl(my_k1, gbase + t, 64);
  }
  double wv[R][E];
  int cols[R][E];
#pragma unroll
  for (int t = 0; t < R; t++) {
    if (k1s[t] - k0s[t] > LPC * E) longmask |= 1u << t;
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int k = k0s[t] + g + e * LPC;
      const bool ok = k < k1s[t];
      wv[t][e] = ok ? av[k] : 0.0;
      cols[t][e] = ok ? ja[k] : cs;
    }
  }
  unsigned code[E];
#pragma unroll
  for (int e = 0; e < E; e++) code[e] = 0;
#pragma unroll
  for (int t = 0; t < R; t++) {
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int j = cols[t][e];
      const unsigned oo = (unsigned)(j - cs);
      const bool inch = oo < (unsigned)len;
      // gather issued without waiting on the in-chunk test; columns >= zero_from hold zeros by contract
      const double x = (j >= zero_from) ? 0.0 : UOLD(j);
      code[e] |= (inch ? oo : 15u) << (4 * t);
      if (!inch) wv[t][e] *= x;
    }
  }
  // phase B
#pragma unroll
  for (int dir = 0; dir < 2; dir++) {
    if (dir == 0 ? !fwd : !bwd) continue;
#pragma unroll
    for (int tt = 0; tt < R; tt++) {
      const int t = (dir == 0) ? tt : R - 1 - tt;
      double part = 0.0;
#pragma unroll
      for (int e = 0; e < E; e++) {
        const unsigned o = (code[e] >> (4 * t)) & 15u;
        const double cur = __shfl(myu, gbase + (int)(o & 7u), 64);
        part += (o == 15u) ? wv[t][e] : wv[t][e] * cur;
      }
      if (longmask & (1u << t)) {  // entries beyond the preloaded strips (group-uniform branch)
        const int i = cs + t;
        const int k1 = ia[i + 1];
        for (int k = ia[i] + LPC * E + g; k - g < k1; k += LPC) {
          double v = 0.0, x = 0.0;
          int o = -1;
          if (k < k1) {
            v = av[k];
            const int j = ja[k];
            const unsigned oo = (unsigned)(j - cs);
            if (oo < (unsigned)len)
              o = (int)oo;
            else if (j < zero_from)
              x = UOLD(j);
          }
          const double cur = __shfl(myu, gbase + (o < 0 ? 0 : o), 64);
          part += v * (o < 0 ? x : cur);
        }
      }
      const double sum = group_sum<LPC>(part);
      if (g == t && mysel) myu += (myrhs - sum) * mywd;
    }
  }
  if (g < len) u_new[cs + g] = myu;
#undef UOLD
}


// ---------------------------------------------------------------------------
// Hybrid Gauss-Seidel, dense-chunk formulation (the default for chunk = 8).
//
// gs_group_k above sweeps the 8 rows of a chunk with one cross-lane reduction and
// one division per row and direction; counters showed its VALU pipes saturated
// (955 VALU instructions per wave, 100 % busy) while HBM idled at ~2 TB/s.  Here
// everything that does not depend on the sweep order leaves the serial part:
//   * entries with a column outside the chunk multiply the snapshot of u: their
//     8 row sums are reduced ONCE over the group by a reduce-scatter butterfly
//     (8 -> 4 -> 2 -> 1 values per lane while the lane count halves) and serve the
//     forward and the backward sweep;
//   * entries inside the chunk (diagonal included) are scattered into a dense
//     8x8 block in LDS, one block per chunk; owner lane t then holds row t of the
//     block in registers;
//   * rows longer than the preloaded strips are finished in the same pre-sweep
//     phase (no slow path inside the sweep);
//   * the sweep itself is 8 fused multiply-adds per row on the owner lanes, one
//     multiplication by the precomputed w/d, and one broadcast of the new value.
// ---------------------------------------------------------------------------
// LDS accesses of one wave are processed in issue order; this only keeps the compiler from moving them
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL>
__device__ __forceinline__ double dpp_xchg(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// value of the partner lane for the exchange step with distance M inside an aligned group:
// M = 1, 2: xor; 4: mirror inside 8 lanes; 8: mirror inside 16 lanes (both pair the two halves); 16, 32: xor
template <int M>
__device__ __forceinline__ double partner(double v) {
  if (M == 1) return dpp_xchg<0xB1>(v);
  if (M == 2) return dpp_xchg<0x4E>(v);
  if (M == 4) return dpp_xchg<0x141>(v);
  if (M == 8) return dpp_xchg<0x140>(v);
  return __shfl_xor(v, M, 64);
}

// reduce-scatter step: NV values per lane -> NV/2; lanes whose bit M is clear keep the lower half of the rows
template <int M, int NV>
__device__ __forceinline__ void rs_step(double *p, int g) {
  const bool hi = (g & M) != 0;
#pragma unroll
  for (int q = 0; q < NV / 2; q++) {
    const double keep = hi ? p[NV / 2 + q] : p[q];
    const double send = hi ? p[q] : p[NV / 2 + q];
    p[q] = keep + partner<M>(send);
  }
}

template <int LPC, int E>
__global__ __launch_bounds__(256) void gs_dense_k(int n, int chunk0, int nchunks, const int *__restrict__ ia,
                                                  const int *__restrict__ ja, const double *__restrict__ av,
                                                  const signed char *__restrict__ cf, int points,
                                                  const double *__restrict__ dd, const double *__restrict__ f,
                                                  const double *__restrict__ offc,
                                                  const double *__restrict__ u_lo,
                                                  const double *__restrict__ u_hi, int split,
                                                  double *__restrict__ u_new, int fwd, int bwd, double w,
                                                  int /*zero_from: not used, the vector holds real zeros*/) {
#define UOLD(j) (((j) < split ? u_lo : u_hi)[(j)])
  constexpr int R = 8;
  constexpr int CPW = 64 / LPC;
  constexpr int GPB = 256 / LPC;
  __shared__ double Cs[GPB][R][R];
  const int lane = threadIdx.x & 63;
  const int g = lane & (LPC - 1);
  const int gbase = lane & ~(LPC - 1);
  const int grp = threadIdx.x / LPC;
  const long long wave = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const long long c = chunk0 + wave * CPW + (lane / LPC);
  const bool live = c < nchunks;
  const int cs = live ? (int)(c * R) : 0;
  const int len = live ? min(R, n - cs) : 0;

  // owner lane g < len holds the state of row cs+g
  double myu = 0.0, myrhs = 0.0, wd = 0.0;
  int my_k0 = 0, my_k1 = 0;
  if (g < len) {
    const int i = cs + g;
    my_k0 = ia[i];
    const int a1 = ia[i + 1];
    const int mark = (points != 0 && cf != nullptr) ? (int)cf[i] : points;
    myu = UOLD(i);
    const double myd = dd[i];
    myrhs = f[i];
    if (offc) myrhs -= offc[i];
    const bool rowsel = (mark == points);
    my_k1 = rowsel ? a1 : my_k0;  // unselected rows load nothing
    if (rowsel && myd != 0.0) wd = w / myd;
  }
  {
    double *blk = &Cs[grp][0][0];
#pragma unroll
    for (int q = g; q < R * R; q += LPC) blk[q] = 0.0;
  }
  int k0s[R], k1s[R];
#pragma unroll
  for (int t = 0; t < R; t++) {
    k0s[t] = __shfl(my_k0, gbase + t, 64);
    k1s[t] = __shfl(my_k1, gbase + t, 64);
  }
  double wv[R][E];
  int cols[R][E];
#pragma unroll
  for (int t = 0; t < R; t++) {
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int k = k0s[t] + g + e * LPC;
      const bool ok = k < k1s[t];
      wv[t][e] = ok ? av[k] : 0.0;
      cols[t][e] = ok ? ja[k] : -1;
    }
  }
  wave_sync_lds();  // the block is zero before the scatter below (a chunk's block is private to its wave)
  double p[R];
#pragma unroll
  for (int t = 0; t < R; t++) {
    p[t] = 0.0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int j = cols[t][e];
      const unsigned oo = (unsigned)(j - cs);
      const bool inch = (j >= 0) && oo < (unsigned)len;
      const double x = UOLD(j < 0 ? cs : j);  // unconditional gather: no load waits on a branch
      if (inch)
        Cs[grp][t][oo] = wv[t][e];
      else
        p[t] += wv[t][e] * x;
    }
    if (k1s[t] - k0s[t] > LPC * E) {  // the rest of a long row (uniform inside the group)
      for (int k = k0s[t] + LPC * E + g; k < k1s[t]; k += LPC) {
        const double a = av[k];
        const int j = ja[k];
        const unsigned oo = (unsigned)(j - cs);
        if (oo < (unsigned)len)
          Cs[grp][t][oo] = a;
        else
          p[t] += a * UOLD(j);
      }
    }
  }
  // row sums of the out-of-chunk part: reduce-scatter over the group, then the owners fetch theirs
  rs_step<LPC / 2, 8>(p, g);
  if (LPC >= 16) rs_step<(LPC >= 16 ? LPC / 4 : 1), 4>(p, g); else rs_step<2, 4>(p, g);
  if (LPC >= 32) rs_step<(LPC >= 32 ? LPC / 8 : 1), 2>(p, g); else if (LPC == 16) rs_step<2, 2>(p, g); else rs_step<1, 2>(p, g);
  double S = p[0];  // full sum of row g / (LPC / 8) once the low bits are folded in
  if (LPC >= 16) {
    S += partner<1>(S);
    if (LPC >= 32) S += partner<2>(S);
    if (LPC >= 64) S += partner<4>(S);
  }
  if (LPC > 8) S = __shfl(S, gbase + (g & 7) * (LPC / 8), 64);
  wave_sync_lds();  // the dense block is complete
  double crow[R], uc[R];
  {
    const int r = g & 7;
#pragma unroll
    for (int j = 0; j < R; j++) crow[j] = Cs[grp][r][j];
  }
#pragma unroll
  for (int j = 0; j < R; j++) uc[j] = __shfl(myu, gbase + j, 64);
#pragma unroll
  for (int dir = 0; dir < 2; dir++) {
    if (dir == 0 ? !fwd : !bwd) continue;
#pragma unroll
    for (int tt = 0; tt < R; tt++) {
      const int t = (dir == 0) ? tt : R - 1 - tt;
      double s = S;
#pragma unroll
      for (int j = 0; j < R; j++) s += crow[j] * uc[j];
      const double nu = myu + (myrhs - s) * wd;  // unselected rows: wd == 0
      double b;
      if (LPC == 64) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(nu), t);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(nu), t);
        b = __hiloint2double(hi, lo);
      } else {
        b = __shfl(nu, gbase + t, 64);
      }
      uc[t] = b;
      if (g == t) myu = b;
    }
  }
  if (g < len) u_new[cs + g] = myu;
#undef UOLD
}


// ---------------------------------------------------------------------------
// Hybrid Gauss-Seidel on the SpMV tiles (levels with an x cache).
//
// The chunk kernels above read the matrix row by row and gather u through L1;
// on the coarse levels they wait on memory most of the time (53 % L1 misses,
// 2 TB/s).  This kernel reads like spmv_stream_xc -- one workgroup per tile of
// <= 256 rows / < 2048 entries, matrix stream in aligned pairs, the tile's
// unique columns gathered once (coalesced) into LDS -- and then sweeps the
// tile's chunks: 8 lanes per chunk, lane g owns row g, sums its out-of-chunk
// products out of LDS, picks its in-chunk coefficients by the code bits of the
// 16-bit column entries, and runs the dense 8x8 sweep of gs_dense_k.
// ---------------------------------------------------------------------------
template <bool VAL8, int BLOCK>
__global__ __launch_bounds__(BLOCK, 2048 / BLOCK) void gs_tile_k(int blk0, int nblk, const int *__restrict__ tdesc,
                                                        const int *__restrict__ ia, const double *__restrict__ av,
                                                        const int *__restrict__ ucols,
                                                        const unsigned short *__restrict__ lcol,
                                                        const signed char *__restrict__ cf, int points,
                                                        const double *__restrict__ dd, const double *__restrict__ f,
                                                        const double *__restrict__ offc,
                                                        const double *__restrict__ u_lo,
                                                        const double *__restrict__ u_hi, int split,
                                                        double *__restrict__ u_new, int fwd, int bwd, double w,
                                                        int row_begin, int row_end, int zero_from,
                                                        double *__restrict__ tout, int t_from,
                                                        const unsigned char *__restrict__ vidx,
                                                        const double *__restrict__ vlut,
                                                        const unsigned short *__restrict__ ucode,
                                                        const int *__restrict__ ubase) {
#define UOLD(j) (((j) < split ? u_lo : u_hi)[(j)])
  constexpr int TILE = 8 * BLOCK;        // TILE / TILE_WIDE
  __shared__ double buf[TILE];           // x cache, then products / in-chunk coefficients
  // the entries' lcol words.  The value dictionary (see spmv_stream_xc) lives in the first 2 KB of the same bytes,
  // which are only written as `code` after the last lookup (second barrier) -- no LDS beyond the 20 KB that allow
  // 8 workgroups per CU.  One 8-byte-aligned byte array with two typed views.
  __shared__ __attribute__((aligned(8))) unsigned char code_bytes[TILE * sizeof(unsigned short)];
  static_assert(TILE * sizeof(unsigned short) >= 256 * sizeof(double), "the value table must fit the code array");
  unsigned short *code = reinterpret_cast<unsigned short *>(code_bytes);
  double *slut = reinterpret_cast<double *>(code_bytes);
  if ((int)blockIdx.x >= nblk) return;
  const int blk = blk0 + blockIdx.x;
  const int tid = threadIdx.x;
  double lutv = 0.0;  // stored to LDS with the x cache (see spmv_stream_xc)
  if (VAL8 && tid < 256) lutv = vlut[tid];
  // one descriptor load, then the loads in the order of their dependent chains (see spmv_stream_xc): column
  // list, matrix stream, the gathers as soon as the column ids are back, then the per-row data (C/F mark -> row
  // selected? -> divisor, right-hand side, row pointers: loads that wait for each other)
  const int4 d0 = reinterpret_cast<const int4 *>(tdesc)[2 * blk];
  const int4 d1 = reinterpret_cast<const int4 *>(tdesc)[2 * blk + 1];
  const int r0 = d0.x, r1 = d0.y, len = d0.w, u0 = d1.x, nu = d1.y;
  const long long base64 = ((long long)d1.z << 32) | (long long)(unsigned)d0.z;  // see spmv_stream_xc
  // columns >= zero_from hold zeros by contract (first sweep on a zero guess): nothing to gather there --
  // the unique columns ascend, so for zero_from == 0 not even the ids are read
  const bool all_zero = zero_from <= 0;
  constexpr int NU = TILE / BLOCK;
  int ucid[NU];
  int myblock = 0;
  const bool coded = ucode && d1.w >= 0;  // (uniform per tile, see spmv_stream_xc)
  if (ucode && !coded) ucols += (-d1.w - 1) - u0;
  if (!all_zero && coded) {
    myblock = LIST_LOAD(ubase + d1.w + (tid & 63));
#pragma unroll
    for (int q = 0; q < NU; q++) {
      const int k = tid + q * BLOCK;
      ucid[q] = 0;
      if (k < nu) ucid[q] = LIST_LOAD(ucode + u0 + k);
    }
  } else if (!all_zero) {
#pragma unroll
    for (int q = 0; q < NU; q++) {
      const int k = tid + q * BLOCK;
      ucid[q] = 0;
      if (!(MI_ABLATE & 2) && k < nu) ucid[q] = LIST_LOAD(ucols + u0 + k);
    }
  }
  const long long base_al64 = base64 & ~1LL;
  const int base = (int)(base64 - base_al64), base_al = 0;  // tile-local from here on
  const int end = base + len;
  const int cnt = end;
  const unsigned ia_off = (unsigned)base_al64;
  av += base_al64;
  lcol += base_al64;
  if (VAL8) vidx += base_al64;
  constexpr int NIT = TILE / (2 * BLOCK);
  d2_t vv[NIT];
  unsigned cw[NIT], vw[NIT];  // two 16-bit column words / two value indices, as loaded (see spmv_stream_xc)
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int k = 2 * tid + it * 2 * BLOCK;
    cw[it] = 0;
    vw[it] = 0;
    if (MI_ABLATE & 4) vv[it] = d2_t{1.0, 1.0};
    if (!(MI_ABLATE & 4) && k < cnt) {
      if (VAL8)
        vw[it] = STREAM_LOAD(reinterpret_cast<const unsigned short *>(vidx + base_al + k));
      else
        vv[it] = STREAM_LOAD(reinterpret_cast<const d2_t *>(av + base_al + k));
      cw[it] = STREAM_LOAD(reinterpret_cast<const unsigned *>(lcol + base_al + k));
    }
  }
  // the gathers in flight together, before the first LDS write and before the per-row chain (see spmv_stream_xc); lanes
  // beyond the list and columns from zero_from on read entry 0 and are overwritten / not stored.  With 8-byte values in
  // registers all NU at once would cost 74 VGPRs (6 instead of 8 waves per SIMD): that variant takes them in two
  // batches -- the first covers tiles of up to 4 * BLOCK unique columns, i.e. nearly all of them.
  constexpr int GB = VAL8 ? NU : NU / 2;
  double xv[GB];
  if (!all_zero && coded) {
#pragma unroll
    for (int q = 0; q < NU; q++) ucid[q] = (__shfl(myblock, ucid[q] >> 10, 64) << 10) | (ucid[q] & 1023);
  }
  if (!all_zero) {
#pragma unroll
    for (int q = 0; q < GB; q++) {
      const int j = (ucid[q] >= zero_from) ? 0 : ((MI_ABLATE & 1) ? min(ucid[q], 63) : ucid[q]);
      xv[q] = UOLD(j);
    }
  }
  // LPR lanes per row, as many as this tile's row count leaves room for (uniform in the workgroup);
  // 8 * LPR consecutive lanes = one chunk; all lanes of a row carry its state, the first one writes it back
  const int nr = r1 - r0;
  const int LPR = nr <= BLOCK / 8 ? 8 : nr <= BLOCK / 4 ? 4 : nr <= BLOCK / 2 ? 2 : 1;
  const int rl = tid / LPR, sub = tid % LPR;
  const int i = r0 + rl;
  double myu = 0.0, myrhs = 0.0, wd = 0.0;
  bool rowsel = false;
  int s0 = 0, s1 = 0;
  if (rl < nr) {
    const int mark = (points != 0 && cf != nullptr) ? (int)cf[i] : points;
    myu = (i >= zero_from) ? 0.0 : UOLD(i);
    rowsel = (mark == points) && i >= row_begin && i < row_end;
    if (rowsel) {
      const double myd = dd[i];
      myrhs = f[i];
      if (offc) myrhs -= offc[i];
      if (myd != 0.0) wd = w / myd;
      s0 = (int)((unsigned)ia[i] - ia_off);
      s1 = (int)((unsigned)ia[i + 1] - ia_off);
    }
  }
  if (VAL8 && tid < 256) slut[tid] = lutv;
  if (!all_zero) {
#pragma unroll
    for (int q = 0; q < GB; q++) {
      const int k = tid + q * BLOCK;
      if (k < nu) buf[k] = (ucid[q] >= zero_from) ? 0.0 : xv[q];
    }
    if (GB < NU && nu > GB * BLOCK) {  // (uniform) the rare tile with more unique columns than the first batch holds
#pragma unroll
      for (int q = GB; q < NU; q++) {
        const int j = (ucid[q] >= zero_from) ? 0 : ucid[q];
        xv[q - GB] = UOLD(j);
      }
#pragma unroll
      for (int q = GB; q < NU; q++) {
        const int k = tid + q * BLOCK;
        if (k < nu) buf[k] = (ucid[q] >= zero_from) ? 0.0 : xv[q - GB];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int k = 2 * tid + it * 2 * BLOCK;
    if (k < cnt) {
      const bool ok0 = (base_al + k >= base);
      const bool ok1 = (base_al + k + 1 < end);
      const unsigned c0 = cw[it] & 0xffffu, c1 = cw[it] >> 16;
      if (VAL8) {
        vv[it].x = slut[vw[it] & 0xffu];
        vv[it].y = slut[vw[it] >> 8];
      }
      // out-of-chunk: product with the snapshot value; in-chunk: the coefficient itself
      const double x0 = all_zero ? 0.0 : buf[c0 & XC_ID_MASK], x1 = all_zero ? 0.0 : buf[c1 & XC_ID_MASK];
      vv[it].x = ok0 ? ((c0 & XC_INCH) ? vv[it].x : vv[it].x * x0) : 0.0;
      vv[it].y = ok1 ? ((c1 & XC_INCH) ? vv[it].y : vv[it].y * x1) : 0.0;
    }
  }
  __syncthreads();  // every x-cache read is done
#pragma unroll
  for (int it = 0; it < NIT; it++) {
    const int k = 2 * tid + it * 2 * BLOCK;
    if (k < cnt) {
      buf[k] = vv[it].x;
      buf[k + 1] = vv[it].y;
      code[k] = (unsigned short)(cw[it] & 0xffffu);
      code[k + 1] = (unsigned short)(cw[it] >> 16);
    }
  }
  __syncthreads();
  // per row: out-of-chunk sum (the row's LPR lanes share the entries) and the row of the dense 8x8 chunk
  // block.  Columns ascend and a chunk is 8 consecutive ids, so the in-chunk entries of a row are ONE run
  // [kin, kin + popcount(mask)) of its entries, in ascending in-chunk offset: the loop only notes where the
  // run starts and which offsets occur; the coefficients are then read straight from the product buffer.
  double S = 0.0;
  unsigned inmask = 0;
  int kin = TILE;
  if (rowsel) {
    for (int k = s0 + sub; k < s1; k += LPR) {
      const unsigned c = code[k];
      const double v = buf[k];
      const bool in = (c & XC_INCH) != 0;
      S += in ? 0.0 : v;
      inmask |= in ? (1u << ((c >> XC_OFF_SHIFT) & 7)) : 0u;
      kin = in ? min(kin, k) : kin;
    }
  }
  for (int m = 1; m < LPR; m <<= 1) {
    S += __shfl_xor(S, m, 64);
    inmask |= (unsigned)__shfl_xor((int)inmask, m, 64);
    kin = min(kin, __shfl_xor(kin, m, 64));
  }
  // F pass on a zero guess: S is exactly the row's product with the C values -- f - S is what the residual that
  // follows needs from those columns (BoomerAMG::cycle)
  // (f is read again here rather than kept in two registers through the kernel: 66 -> 64 VGPRs = 8 waves per SIMD)
  if (tout && rowsel && sub == 0 && i >= t_from) tout[i] = f[i] - S;
  double crow[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const bool has = (inmask >> j) & 1u;
    const int pos = kin + __popc(inmask & ((1u << j) - 1u));
    crow[j] = has ? buf[has ? pos : 0] : 0.0;
  }
  // the sweep: a chunk = 8 rows = 8 * LPR consecutive lanes (tiles start on a multiple of 8 rows)
  const int lane = tid & 63;
  const int g = (lane / LPR) & 7;
  const int gbase = lane & ~(8 * LPR - 1);
  double uc[8];
#pragma unroll
  for (int j = 0; j < 8; j++) uc[j] = __shfl(myu, gbase + j * LPR, 64);
  // The phase is VALU throughput (profiles/r04_ablation_tile_kernels.txt: 16 steps x (8 double FMAs + update + cross-lane
  // read) on every wave = a quarter of a pass).  Every lane forms ITS row's sum at every step although only row t's is
  // used at step t; what can be shared without touching the order of the additions (S, then columns 0..7 ascending):
  // * forward, rows above t already hold their new values when row t's turn comes, and they are the FIRST terms of its
  //   sum: `acc` = S + the terms of the columns swept so far is carried along (one FMA per step), step t adds the 8 - t
  //   remaining terms to a copy of it -- 44 instead of 64 FMAs.  (Backward the new values are the LAST terms; the prefix
  //   over the old ones has a different length in every lane, and masking it costs what it saves: all 8 terms there.)
  // * the row's own value at its step is uc[t] (the lane's copy of the chunk's values), so `myu` need not follow the
  //   sweep: it is picked out of uc[] once at the end instead of a select per step.
  // Same operations on the same operands for the row whose turn it is; the other lanes' results were never used.
#if !defined(MI_GS_SHARED_PREFIX) || MI_GS_SHARED_PREFIX
  if (!(MI_ABLATE & 8)) {
    if (fwd) {
      double acc = S;
#pragma unroll
      for (int t = 0; t < 8; t++) {
        double sacc = acc;
#pragma unroll
        for (int j = t; j < 8; j++) sacc += crow[j] * uc[j];
        const double nu2 = uc[t] + (myrhs - sacc) * wd;  // unselected rows: wd == 0
        const double b = __shfl(nu2, gbase + t * LPR, 64);
        uc[t] = b;
        acc += crow[t] * b;
      }
    }
    if (bwd) {
#pragma unroll
      for (int tt = 0; tt < 8; tt++) {
        const int t = 7 - tt;
        double sacc = S;
#pragma unroll
        for (int j = 0; j < 8; j++) sacc += crow[j] * uc[j];
        const double nu2 = uc[t] + (myrhs - sacc) * wd;
        uc[t] = __shfl(nu2, gbase + t * LPR, 64);
      }
    }
    if (fwd || bwd) {
#pragma unroll
      for (int j = 0; j < 8; j++) myu = (g == j) ? uc[j] : myu;
    }
  }
#else
#pragma unroll
  for (int dir = 0; dir < 2; dir++) {
    if ((MI_ABLATE & 8) || (dir == 0 ? !fwd : !bwd)) continue;
#pragma unroll
    for (int tt = 0; tt < 8; tt++) {
      const int t = (dir == 0) ? tt : 7 - tt;
      double sacc = S;
#pragma unroll
      for (int j = 0; j < 8; j++) sacc += crow[j] * uc[j];
      const double nu2 = myu + (myrhs - sacc) * wd;  // unselected rows: wd == 0
      const double b = __shfl(nu2, gbase + t * LPR, 64);
      uc[t] = b;
      if (g == t) myu = b;
    }
  }
#endif
  if (rl < nr && sub == 0) u_new[i] = myu;
#undef UOLD
}

// ---------------------------------------------------------------------------
// BLAS-1
// ---------------------------------------------------------------------------
// Second stage of the two-stage reductions inside the first-stage kernel: every block leaves its partial sum, takes a
// ticket, and the block that draws the last one adds the partials up -- in a fixed order
// (thread t takes partials t, t + 256, ...; wave shuffles; (w0 + w1) + (w2 + w3)), so the result does not depend on
// which block comes last and equals the two-launch form bit for bit.  One launch per inner product instead of two
// (a GMRES solve of m steps issues m(m+1)/2 + 2m + 3 of them).  The ticket counter is reset by the last block;
// reductions are serialised on the library stream, so one counter serves them all.
// Ordering without a release fence: an agent-scope release would write this XCD's L2 back for EVERY block (the
// vector the kernel has just stored included: measured +18 % on the fused Gram-Schmidt step).  Only the partial has
// to be visible before the ticket, and both are agent-scope atomics executed at the device's coherence point: the
// partial is stored by an exchange whose RETURN the lane waits for (s_waitcnt vmcnt(0): the operation has been
// performed) before it draws the ticket; the last block reads the partials with agent-scope atomic loads after its
// own ticket came back.
__device__ __forceinline__ void finish_reduction(double block_sum, double *__restrict__ partials,
                                                 unsigned *__restrict__ ticket, double *__restrict__ out, double *ws) {
  __shared__ unsigned s_last;
  const int tid = threadIdx.x;
  if (tid == 0) {
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(partials + blockIdx.x);
    const unsigned long long old =
        __hip_atomic_exchange(slot, (unsigned long long)__double_as_longlong(block_sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(old) : "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  const int nb = (int)gridDim.x;
  double s = 0.0;
  for (int i = tid; i < nb; i += 256) s += __hip_atomic_load(partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  s = wave_sum(s);
  __syncthreads();  // ws is reused
  if ((tid & 63) == 0) ws[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    out[0] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(256) void dot_partial_k(const double *__restrict__ x, const double *__restrict__ y, int n,
                                                     double *__restrict__ partials, unsigned *__restrict__ ticket,
                                                     double *__restrict__ out) {
  __shared__ double ws[4];
  const int tid = threadIdx.x;
  double s = 0.0;
  const long long stride = (long long)gridDim.x * 512;
  long long i = ((long long)blockIdx.x * 256 + tid) * 2;
  for (; i + 1 < n; i += stride) {
    const d2_t a = nt_load(reinterpret_cast<const d2_t *>(x + i));
    const d2_t b = nt_load(reinterpret_cast<const d2_t *>(y + i));
    s += a.x * b.x + a.y * b.y;
  }
  if (i < n) s += x[i] * y[i];
  s = wave_sum(s);
  if ((tid & 63) == 0) ws[tid >> 6] = s;
  __syncthreads();
  finish_reduction((ws[0] + ws[1]) + (ws[2] + ws[3]), partials, ticket, out, ws);
}

// fused modified-Gram-Schmidt step: y += a*xa (a = scale * *alpha_dev), then the
// partial dot <xd, y_new> (xd == nullptr: <y_new, y_new>) in the same pass.
// Same per-thread accumulation pattern as dot_partial_k, so the result is
// bit-identical to axpy followed by dot.
__global__ __launch_bounds__(256) void axpy_dot_partial_k(const double *__restrict__ alpha_dev, double scale,
                                                          const double *__restrict__ xa, double *__restrict__ y,
                                                          const double *__restrict__ xd, int n,
                                                          double *__restrict__ partials, unsigned *__restrict__ ticket,
                                                          double *__restrict__ out) {
  __shared__ double ws[4];
  const int tid = threadIdx.x;
  const double a = scale * alpha_dev[0];
  double s = 0.0;
  const long long stride = (long long)gridDim.x * 512;
  long long i = ((long long)blockIdx.x * 256 + tid) * 2;
  for (; i + 1 < n; i += stride) {
    // every operand is streamed exactly once: non-temporal accesses keep them out of L2's way
    const d2_t xv = nt_load(reinterpret_cast<const d2_t *>(xa + i));
    d2_t yv = nt_load(reinterpret_cast<const d2_t *>(y + i));
    yv.x += a * xv.x;
    yv.y += a * xv.y;
    __builtin_nontemporal_store(yv, reinterpret_cast<d2_t *>(y + i));
    d2_t dv = yv;
    if (xd) dv = nt_load(reinterpret_cast<const d2_t *>(xd + i));
    s += dv.x * yv.x + dv.y * yv.y;
  }
  if (i < n) {
    const double yn = y[i] + a * xa[i];
    y[i] = yn;
    s += (xd ? xd[i] : yn) * yn;
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) ws[tid >> 6] = s;
  __syncthreads();
  finish_reduction((ws[0] + ws[1]) + (ws[2] + ws[3]), partials, ticket, out, ws);
}



// ---- block Gram-Schmidt pieces of COGMRES: up to MASS_NV inner products off one vector in one pass, and the
// matching block update.  The per-thread accumulation pattern is dot_partial_k's, so every result equals the
// separate dot's bit for bit; the update adds the terms in ascending j like repeated axpys.
struct MassPtrs {
  const double *p[MASS_NV];
};

__global__ __launch_bounds__(256) void mass_dot_partial_k(MassPtrs P, int m, const double *__restrict__ w, int n,
                                                          double *__restrict__ partials) {
  __shared__ double ws[MASS_NV][4];
  const int tid = threadIdx.x;
  double acc[MASS_NV];
#pragma unroll
  for (int j = 0; j < MASS_NV; j++) acc[j] = 0.0;
  const long long stride = (long long)gridDim.x * 512;
  long long i = ((long long)blockIdx.x * 256 + tid) * 2;
  for (; i + 1 < n; i += stride) {
    const double2 b = *reinterpret_cast<const double2 *>(w + i);
#pragma unroll
    for (int j = 0; j < MASS_NV; j++)
      if (j < m) {
        const double2 a = *reinterpret_cast<const double2 *>(P.p[j] + i);
        acc[j] += a.x * b.x + a.y * b.y;
      }
  }
  if (i < n) {
#pragma unroll
    for (int j = 0; j < MASS_NV; j++)
      if (j < m) acc[j] += P.p[j][i] * w[i];
  }
#pragma unroll
  for (int j = 0; j < MASS_NV; j++) {
    const double v = wave_sum(acc[j]);
    if ((tid & 63) == 0) ws[j][tid >> 6] = v;
  }
  __syncthreads();
  if (tid < m) partials[(size_t)tid * RED_MAX_BLOCKS + blockIdx.x] = (ws[tid][0] + ws[tid][1]) + (ws[tid][2] + ws[tid][3]);
}

__global__ __launch_bounds__(256) void reduce_final_multi_k(const double *__restrict__ partials, int nb,
                                                            double *__restrict__ out) {
  __shared__ double ws[4];
  const int tid = threadIdx.x;
  const double *mine = partials + (size_t)blockIdx.x * RED_MAX_BLOCKS;
  double s = 0.0;
  for (int i = tid; i < nb; i += 256) s += mine[i];
  s = wave_sum(s);
  if ((tid & 63) == 0) ws[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// w += sum_j (scale * coef[j]) * p_j, terms added in ascending j
__global__ __launch_bounds__(256) void mass_axpy_k(MassPtrs P, int m, const double *__restrict__ coef, double scale,
                                                   double *__restrict__ w, int n) {
  double c[MASS_NV];
#pragma unroll
  for (int j = 0; j < MASS_NV; j++) c[j] = (j < m) ? scale * coef[j] : 0.0;
  const long long stride = (long long)gridDim.x * 512;
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  for (; i + 1 < n; i += stride) {
    double2 wv = *reinterpret_cast<double2 *>(w + i);
#pragma unroll
    for (int j = 0; j < MASS_NV; j++)
      if (j < m) {
        const double2 a = *reinterpret_cast<const double2 *>(P.p[j] + i);
        wv.x += c[j] * a.x;
        wv.y += c[j] * a.y;
      }
    *reinterpret_cast<double2 *>(w + i) = wv;
  }
  if (i < n) {
    double v = w[i];
#pragma unroll
    for (int j = 0; j < MASS_NV; j++)
      if (j < m) v += c[j] * P.p[j][i];
    w[i] = v;
  }
}

// w = c_0 p_0 (INIT) or w += c_0 p_0, then += c_1 p_1, ... in this order: the chain copy / scale / axpy / axpy ... of the
// GMRES solution update in one pass over up to MASS_NV vectors (same operations per element in the same order)
struct MassCoef {
  double c[MASS_NV];
};
template <bool INIT>
__global__ __launch_bounds__(256) void lin_comb_k(MassPtrs P, int m, MassCoef C, double *__restrict__ w, int n) {
  const long long stride = (long long)gridDim.x * 512;
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  for (; i + 1 < n; i += stride) {
    double2 wv;
    if (INIT) {
      const d2_t a = nt_load(reinterpret_cast<const d2_t *>(P.p[0] + i));
      wv.x = C.c[0] * a.x;
      wv.y = C.c[0] * a.y;
    } else {
      wv = *reinterpret_cast<double2 *>(w + i);
    }
#pragma unroll
    for (int j = INIT ? 1 : 0; j < MASS_NV; j++)
      if (j < m) {
        const d2_t a = nt_load(reinterpret_cast<const d2_t *>(P.p[j] + i));
        wv.x += C.c[j] * a.x;
        wv.y += C.c[j] * a.y;
      }
    *reinterpret_cast<double2 *>(w + i) = wv;
  }
  if (i < n) {
    double v = INIT ? C.c[0] * P.p[0][i] : w[i];
#pragma unroll
    for (int j = INIT ? 1 : 0; j < MASS_NV; j++)
      if (j < m) v += C.c[j] * P.p[j][i];
    w[i] = v;
  }
}

// y += (scale * (alpha_dev ? *alpha_dev : 1)) * x
__global__ __launch_bounds__(256) void axpy_k(const double *__restrict__ alpha_dev, double scale,
                                              const double *__restrict__ x, double *__restrict__ y, int n) {
  const double a = scale * (alpha_dev ? alpha_dev[0] : 1.0);
  const long long stride = (long long)gridDim.x * 512;
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  for (; i + 1 < n; i += stride) {
    const double2 xv = *reinterpret_cast<const double2 *>(x + i);
    double2 yv = *reinterpret_cast<double2 *>(y + i);
    yv.x += a * xv.x;
    yv.y += a * xv.y;
    *reinterpret_cast<double2 *>(y + i) = yv;
  }
  if (i < n) y[i] += a * x[i];
}

// MODE 0: x *= scale ; MODE 1: x *= 1/sqrt(*sumsq_dev) when *sumsq_dev > 0
template <int MODE>
__global__ __launch_bounds__(256) void scale_k(const double *__restrict__ sumsq_dev, double scale,
                                               double *__restrict__ x, int n) {
  double a = scale;
  if (MODE == 1) {
    const double t = sumsq_dev[0];
    if (!(t > 0.0)) return;
    a = 1.0 / sqrt(t);
  }
  const long long stride = (long long)gridDim.x * 512;
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  for (; i + 1 < n; i += stride) {
    double2 v = *reinterpret_cast<double2 *>(x + i);
    v.x *= a;
    v.y *= a;
    *reinterpret_cast<double2 *>(x + i) = v;
  }
  if (i < n) x[i] *= a;
}

__global__ __launch_bounds__(256) void fill_k(double *__restrict__ x, int n, double v) {
  const long long stride = (long long)gridDim.x * 512;
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  for (; i + 1 < n; i += stride) *reinterpret_cast<double2 *>(x + i) = make_double2(v, v);
  if (i < n) x[i] = v;
}

__global__ __launch_bounds__(256) void gather_k(const double *__restrict__ x, const int *__restrict__ map,
                                                double *__restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = x[map[i]];
}

__global__ __launch_bounds__(256) void gather2_k(const double *__restrict__ lo, const double *__restrict__ hi, int split,
                                                 const int *__restrict__ map, double *__restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const int j = map[i];
    out[i] = (j < split ? lo : hi)[j];
  }
}

template <int ADD>
__global__ __launch_bounds__(256) void scatter_k(double *__restrict__ x, const int *__restrict__ idx,
                                                 const double *__restrict__ vals, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    if (ADD)
      x[idx[i]] += vals[i];  // caller guarantees unique indices
    else
      x[idx[i]] = vals[i];
  }
}

// u = M f for the coarsest level (relax type 9: M = dense inverse, n small)
// two-stage Gauss-Seidel (relax types 11 / 12): z = r / d, u += z; then z_out = (L z_in) / d with L the strictly
// lower part of the block, u += sign * z_out.  One lane per row: these smoothers are not on the benchmark path.
__global__ __launch_bounds__(256) void two_stage_first_k(int n, const double *__restrict__ r, const double *__restrict__ d,
                                                         double *__restrict__ z, double *__restrict__ u) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double zi = (d[i] != 0.0) ? r[i] / d[i] : 0.0;
  z[i] = zi;
  u[i] += zi;
}
__global__ __launch_bounds__(256) void two_stage_lower_k(int n, const int *__restrict__ ia, const int *__restrict__ ja,
                                                         const double *__restrict__ a, const double *__restrict__ d,
                                                         const double *__restrict__ zin, double sign,
                                                         double *__restrict__ zout, double *__restrict__ u) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int k = ia[i]; k < ia[i + 1]; k++) {
    const int j = ja[k];
    if (j < i) s += a[k] * zin[j];
  }
  const double zi = (d[i] != 0.0) ? s / d[i] : 0.0;
  zout[i] = zi;
  u[i] += sign * zi;
}

__global__ __launch_bounds__(256) void dense_matvec_k(const double *__restrict__ M, const double *__restrict__ f,
                                                      double *__restrict__ u, int n, int m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int j = 0; j < m; j++) s += M[(size_t)i * m + j] * f[j];
  u[i] = s;
}

// u[i] = sum_j Mt[j*n + i] * f[j]: a workgroup of 1024 lanes owns 16 consecutive rows i (lane & 15) and splits the
// columns j over its 64 lane groups (128-byte reads of Mt's rows; ~n/64 loads per lane, all independent: the kernel
// is a latency chain, not a stream -- 47 workgroups and 12 loads per lane for the 753-row tail of the 512^3
// hierarchy); the 64 partial sums of a row are added in group order out of LDS -- a fixed order, the result does
// not depend on scheduling
__global__ __launch_bounds__(1024) void dense_matvec_t_k(const double *__restrict__ Mt, const double *__restrict__ f,
                                                         double *__restrict__ u, int n) {
  __shared__ double part[64][17];
  const int r = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + r;
  double s = 0.0;
  if (i < n)
    for (int j = g; j < n; j += 64) s += Mt[(size_t)j * (size_t)n + i] * f[j];
  part[g][r] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    double t = part[0][r];
#pragma unroll 8
    for (int w = 1; w < 64; w++) t += part[w][r];
    u[i] = t;
  }
}

// peer-store neighbour exchange: see kernels.hpp.  blockIdx.x walks the transfers' workgroups in order.
__global__ __launch_bounds__(1024) void ipc_exchange_k(IpcBatch B, unsigned long long spin_limit, int *error_flag) {
  __shared__ int s_ok;
  int blk = (int)blockIdx.x, ti = 0;
  while (ti < B.n && blk >= B.t[ti].nblocks) blk -= B.t[ti++].nblocks;
  if (ti >= B.n) return;
  const IpcTransfer &T = B.t[ti];
  const int tid = threadIdx.x;
  if (tid == 0) {
    int ok = 1;
    const unsigned long long t0 = wall_clock64();
    // a rank whose flag is already up does not wait the full bound again (a dead transport would otherwise
    // cost bound x exchanges before the end-of-solve check sees it): 1 ms per wait from then on
    const unsigned long long limit = __hip_atomic_load(error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? min(spin_limit, 100000ull) : spin_limit;
    while (__hip_atomic_load(T.wait_word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < T.wait_value) {
      if (wall_clock64() - t0 > limit) {  // the peer never arrived: say so and fall through (no hang)
        ok = 0;
        atomicExch(error_flag, 1);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
    s_ok = ok;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // what the peer stored before its flag is visible
  if (s_ok) {
    const unsigned long long n16 = T.bytes / 16;
    const bool al = ((((unsigned long long)T.src) | ((unsigned long long)T.dst)) & 15ull) == 0;
    const unsigned long long stride = (unsigned long long)T.nblocks * 1024ull;
    if (al) {
      const d2_t *sp = reinterpret_cast<const d2_t *>(T.src);
      d2_t *dp = reinterpret_cast<d2_t *>(T.dst);
      for (unsigned long long i = (unsigned long long)blk * 1024ull + tid; i < n16; i += stride) dp[i] = sp[i];
      const unsigned char *sb = reinterpret_cast<const unsigned char *>(T.src);
      unsigned char *db = reinterpret_cast<unsigned char *>(T.dst);
      for (unsigned long long i = n16 * 16 + (unsigned long long)blk * 1024ull + tid; i < T.bytes; i += stride) db[i] = sb[i];
    } else {
      const unsigned char *sb = reinterpret_cast<const unsigned char *>(T.src);
      unsigned char *db = reinterpret_cast<unsigned char *>(T.dst);
      for (unsigned long long i = (unsigned long long)blk * 1024ull + tid; i < T.bytes; i += stride) db[i] = sb[i];
    }
  }
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // this workgroup's stores are out
    const unsigned done = __hip_atomic_fetch_add(T.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (unsigned)T.nblocks - 1) {
      __hip_atomic_store(T.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
      __hip_atomic_store(T.post_word, T.post_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Workgroups of the streaming vector kernels.  Measured at 512^3 (134 M doubles per vector, profiles/
// r03_vector_grid_512.txt): the fused Gram-Schmidt step runs at 3.8 / 6.0 / 6.4 / 6.2 / 5.4 / 5.2 / 4.9 TB/s with
// 256 / 512 / 768 / 1024 / 2048 / 4096 / 16384 workgroups of 256 lanes -- three per CU stream best (fewer, longer
// streams keep DRAM pages open); more only adds concurrent streams.  MI_HYPRE_VEC_BLOCKS overrides.
// peer-store all-reduce: see kernels.hpp.  One workgroup of 64 lanes; lane r < size talks to rank r.
__global__ __launch_bounds__(64) void ipc_poison_k(double *buf, int count, const int *error_flag) {
  if ((int)threadIdx.x < count && __hip_atomic_load(error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    buf[threadIdx.x] = __longlong_as_double(0x7ff8000000000000LL);
}
__global__ __launch_bounds__(64) void ipc_allreduce_k(IpcAllreduce A, unsigned long long spin_limit, int *error_flag) {
  __shared__ double mine[IPC_AR_MAX];
  const int r = threadIdx.x;
  const int par = (int)(A.seq & 1ull);
  // A rank whose error flag is up (one of its bounded waits expired: its data are no longer what the algorithm thinks)
  // contributes NaN: EVERY rank then sees NaN in the SAME reduction and the Krylov loops' NaN test (gmres.c's IEEE check,
  // krylov.cpp) takes all of them out of the loop at the same point -- the collectives that are not bounded (all-gathers
  // of the wrapped communicator) stay matched, and the end-of-solve gate (capi.cpp) reports the failure on every rank.
  const bool dead = __hip_atomic_load(error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  if (r < A.count) mine[r] = dead ? __longlong_as_double(0x7ff8000000000000LL) : A.buf[r];
  __syncthreads();
  if (r < A.size && r != A.rank) {
    double *dst = A.peer_slots[r] + ((size_t)par * A.size + A.rank) * IPC_AR_MAX;
    for (int q = 0; q < A.count; q++) __hip_atomic_store(dst + q, mine[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(A.peer_flags[r] + (size_t)par * A.size + A.rank, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (r < A.size && r != A.rank) {
    const unsigned long long *fl = A.my_flags + (size_t)par * A.size + r;
    const unsigned long long t0 = wall_clock64();
    const unsigned long long limit = __hip_atomic_load(error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? min(spin_limit, 100000ull) : spin_limit;  // (see ipc_exchange_k)
    while (__hip_atomic_load(fl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < A.seq) {
      if (wall_clock64() - t0 > limit) {
        atomicExch(error_flag, 1);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  if (r < A.count) {
    double sum = 0.0;
    for (int q = 0; q < A.size; q++) {  // rank order: identical bits on every rank
      const double v = (q == A.rank) ? mine[r]
                                     : __hip_atomic_load(A.my_slots + ((size_t)par * A.size + q) * IPC_AR_MAX + r,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      sum = (q == 0) ? v : sum + v;
    }
    // a sum over slots that never arrived must not pass for a number: NaN ends the Krylov loop's tests on every rank that
    // timed out, and the end-of-solve check (capi.cpp transport_gate) reports it on all of them
    if (__hip_atomic_load(error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) sum = __longlong_as_double(0x7ff8000000000000LL);
    A.buf[r] = sum;
  }
}

inline int vec_grid_cap() {
  static const int cap = getenv("MI_HYPRE_VEC_BLOCKS") ? std::max(1, std::min(RED_MAX_BLOCKS, atoi(getenv("MI_HYPRE_VEC_BLOCKS")))) : 768;
  return cap;
}
inline int vec_grid(int n) {
  long long want = ((long long)n + 511) / 512;
  if (want < 1) want = 1;
  if (want > vec_grid_cap()) want = vec_grid_cap();
  return (int)want;
}

}  // namespace

// ---------------------------------------------------------------- host side

// Tiles of the SpMV / tile Gauss-Seidel kernels: <= 256 rows and < SPMV_TILE entries.  Whenever no 8-row
// chunk exceeds a tile the blocks begin and end on multiples of 8 rows (the hybrid-GS chunks), which the
// tile Gauss-Seidel kernel needs; *chunk_aligned says whether that held for the whole matrix.
int choose_tile_entries(int64_t nnz, int nrows) {
  static const double wide_min = getenv("MI_HYPRE_WIDE_TILE_MIN_ROWLEN") ? atof(getenv("MI_HYPRE_WIDE_TILE_MIN_ROWLEN")) : 100.0;
  // very short rows (the fine level's zero-guess sub-operator: 3.5 entries per row) leave a 256-row tile half empty:
  // MI_HYPRE_WIDE_TILE_MAX_SHORT = x gives operators with at most x entries per row the 512-row tiles too (experiment)
  static const double short_max = getenv("MI_HYPRE_WIDE_TILE_MAX_SHORT") ? atof(getenv("MI_HYPRE_WIDE_TILE_MAX_SHORT")) : 0.0;
  // (only operators that get the x-cache format: the plain stream kernel has 2048-entry tiles only)
  static const int xc_min = getenv("MI_HYPRE_XCACHE_MIN") ? atoi(getenv("MI_HYPRE_XCACHE_MIN")) : 3;
  if (nrows > 0 && short_max > 0.0 && (double)nnz / (double)nrows <= short_max && (double)nnz / (double)nrows >= (double)xc_min)
    return SPMV_TILE_WIDE;
  return (nrows > 0 && wide_min > 0.0 && (double)nnz / (double)nrows >= wide_min) ? SPMV_TILE_WIDE : SPMV_TILE;
}

std::vector<int> build_row_blocks(int nrows, const int64_t *ia, bool *chunk_aligned, int row_cap, int tile_entries) {
  // rows per tile: at most one per thread of the kernel that runs the tile (256, or 512 for the wide tiles) when a
  // Gauss-Seidel kernel sweeps the operator; SpMV-only operators pass a larger cap
  const int block_rows = tile_entries == SPMV_TILE_WIDE ? SPMV_BLOCK_WIDE : SPMV_BLOCK;
  if (row_cap < block_rows) row_cap = block_rows;
  std::vector<int> rb;
  rb.reserve((size_t)nrows / 200 + 2);
  rb.push_back(0);
  bool aligned = true;
  int r = 0;
  while (r < nrows) {
    const int limit = (int)std::min<long long>(nrows, ((long long)r / TILE_SUPER_ROWS + 1) * TILE_SUPER_ROWS);
    const int e = tile_end(r, limit, ia, row_cap, block_rows, tile_entries, aligned);
    rb.push_back(e);
    r = e;
  }
  if (chunk_aligned) *chunk_aligned = aligned;
  return rb;
}

// MI_HYPRE_GS_TILE: 0 never, 1 every level with tiles, unset: levels with mean row length > 5
// (MI_HYPRE_GS_TILE_AVG; 512^3 7-pt: the fine level gains 6 % over the shuffle kernel, its zero-guess
// sub-operator with 3.5 entries per row does not care)
static bool gs_tile_mode(const DevCSR &A) {
  static int v = -2;
  if (v == -2) {
    const char *e = getenv("MI_HYPRE_GS_TILE");
    v = e ? atoi(e) : -1;
  }
  if (v == 0) return false;
  if (v == 1 || A.prefer_gs_tiles) return true;
  static const double thr = getenv("MI_HYPRE_GS_TILE_AVG") ? atof(getenv("MI_HYPRE_GS_TILE_AVG")) : 5.0;
  return (double)A.nnz / (double)std::max(1, A.nrows) > thr;
}
static bool gs_use_old() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("MI_HYPRE_GS_OLD");
    v = (e && atoi(e)) ? 1 : 0;
  }
  return v == 1;
}
static bool gs_force_generic() {
  static int v = -1;
  if (v < 0) {
    const char *e = getenv("MI_HYPRE_GS_GENERIC");
    v = (e && atoi(e)) ? 1 : 0;
  }
  return v == 1;
}

// returns the name of the instantiation it launched (as rocprofv3's kernel statistics spell it)
static const char *launch_stream(int epi, const DevCSR &A, const double *x, double *y, const EpiArgs &e, hipStream_t s,
                                 bool level0 = false) {
  if (A.nrows == 0) return "";
  MI_REQUIRE(A.xcache || !A.big(), "an operator with 2^31 entries or more must be in the x-cache tile format");
  const char *name = "";
  const int nb = A.nblocks;
  const int xchunk = (nb + 7) / 8;
  const dim3 grid(xchunk * 8), block(SPMV_BLOCK);
  if (A.xcache && A.tile_entries == SPMV_TILE_WIDE) {
    const dim3 wide(SPMV_BLOCK_WIDE);
#define XC_LAUNCH_W(EPI_, V8_)                                                                                       \
  name = "spmv_stream_xc<" #EPI_ ", 0, " #V8_ ", 512>";                                                              \
  hipLaunchKernelGGL((spmv_stream_xc<EPI_, 0, V8_, SPMV_BLOCK_WIDE>), grid, wide, 0, s, nb, xchunk, A.tdesc.p, A.ia.p, \
                     A.ja.p, A.a.p, A.ucols.p, A.lcol.p, x, y, e, A.vidx.p, A.vlut.p, A.ucode.p, A.ubase.p)
    if (A.val8) {
      if (epi == 0) {
        XC_LAUNCH_W(0, true);
      } else {
        XC_LAUNCH_W(1, true);
      }
    } else {
      if (epi == 0) {
        XC_LAUNCH_W(0, false);
      } else {
        XC_LAUNCH_W(1, false);
      }
    }
#undef XC_LAUNCH_W
  } else if (A.xcache) {
#define XC_LAUNCH(EPI_, TAG_, V8_)                                                                                  \
  name = "spmv_stream_xc<" #EPI_ ", " #TAG_ ", " #V8_ ", 256>";                                                       \
  hipLaunchKernelGGL((spmv_stream_xc<EPI_, TAG_, V8_, SPMV_BLOCK>), grid, block, 0, s, nb, xchunk, A.tdesc.p, A.ia.p, A.ja.p, \
                     A.a.p, A.ucols.p, A.lcol.p, x, y, e, A.vidx.p, A.vlut.p, A.ucode.p, A.ubase.p)
    if (A.val8) {
      if (epi == 0 && level0) {
        XC_LAUNCH(0, 1, true);
      } else if (epi == 0) {
        XC_LAUNCH(0, 0, true);
      } else {
        XC_LAUNCH(1, 0, true);
      }
    } else {
      if (epi == 0 && level0) {
        XC_LAUNCH(0, 1, false);
      } else if (epi == 0) {
        XC_LAUNCH(0, 0, false);
      } else {
        XC_LAUNCH(1, 0, false);
      }
    }
#undef XC_LAUNCH
  } else if (epi == 0 && level0) {
    name = "spmv_stream<0, 1>";
    hipLaunchKernelGGL((spmv_stream<0, 1>), grid, block, 0, s, nb, xchunk, A.rb.p, A.ia.p, A.ja.p, A.a.p, x, y, e);
  } else if (epi == 0) {
    name = "spmv_stream<0, 0>";
    hipLaunchKernelGGL((spmv_stream<0, 0>), grid, block, 0, s, nb, xchunk, A.rb.p, A.ia.p, A.ja.p, A.a.p, x, y, e);
  } else {
    name = "spmv_stream<1, 0>";
    hipLaunchKernelGGL((spmv_stream<1, 0>), grid, block, 0, s, nb, xchunk, A.rb.p, A.ia.p, A.ja.p, A.a.p, x, y, e);
  }
  MI_HIP(hipGetLastError());
  return name;
}

namespace {
// vidx[k] = position of a[k] in the sorted table of bit patterns; *fail counts values that are not in it
__global__ __launch_bounds__(256) void value_index_k(long long n, const double *__restrict__ a,
                                                     const long long *__restrict__ table, int nt,
                                                     unsigned char *__restrict__ vidx, int *__restrict__ fail) {
  const long long stride = (long long)gridDim.x * 256;
  int bad = 0;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
    const long long bits = __double_as_longlong(a[k]);
    int lo = 0, hi = nt;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (table[mid] < bits)
        lo = mid + 1;
      else
        hi = mid;
    }
    if (lo < nt && table[lo] == bits)
      vidx[k] = (unsigned char)lo;
    else
      bad = 1;
  }
  if (bad) atomicAdd(fail, 1);
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void strided_sample_k(const double *__restrict__ a, size_t stride, int n, double *__restrict__ out) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < n) out[k] = a[(size_t)k * stride];
}
}  // namespace

static int g_value_dict = -1;
bool value_dictionary_enabled() {
  if (g_value_dict < 0) g_value_dict = (getenv("MI_HYPRE_VALUE_DICT") && atoi(getenv("MI_HYPRE_VALUE_DICT")) == 0) ? 0 : 1;
  return g_value_dict == 1;
}
void set_value_dictionary(bool on) { g_value_dict = on ? 1 : 0; }

// Value dictionary of an operator in the solve format: distinct values (bit patterns) of a sample, at most 256;
// one pass then encodes every entry or finds one that is not in the table (the operator keeps the plain stream).
// MI_HYPRE_VALUE_DICT=0 switches it off.
void build_value_dictionary(DevCSR &A, hipStream_t s) {
  A.val8 = false;
  A.vidx.release();
  A.vlut.release();
  if (!value_dictionary_enabled() || !A.xcache || A.nnz < (1 << 16) || !A.a.p) return;
  // a strided sample over the whole array (the leading rows alone are not representative: the C rows of a
  // zero-guess sub-operator hold nothing but their diagonal); enough to see > 256 distinct values at once
  const size_t sample = (size_t)std::min<int64_t>(A.nnz, 1 << 16);
  const size_t stride = (size_t)A.nnz / sample;
  std::vector<double> hs(sample);
  {
    // (a kernel, not hipMemcpy2DAsync: the strided source may span several physical chunks of the device arena, which
    // the 2-D copy refuses with "invalid argument")
    DVec<double> dsample(sample);
    hipLaunchKernelGGL(strided_sample_k, dim3((unsigned)((sample + 255) / 256)), dim3(256), 0, s, A.a.p, stride, (int)sample, dsample.p);
    MI_HIP(hipGetLastError());
    d2h(hs.data(), dsample.p, sample * sizeof(double), s);
    MI_HIP(hipStreamSynchronize(s));
  }
  std::vector<long long> bits(sample);
  std::memcpy(bits.data(), hs.data(), sample * sizeof(double));
  std::sort(bits.begin(), bits.end());
  bits.erase(std::unique(bits.begin(), bits.end()), bits.end());
  if (bits.size() > 256) return;
  const int nt = (int)bits.size();
  DVec<long long> dtab;
  dtab.upload(bits);
  DVec<int> fail(1);
  MI_HIP(hipMemsetAsync(fail.p, 0, sizeof(int), s));
  A.vidx.alloc((size_t)A.nnz);
  hipLaunchKernelGGL(value_index_k, dim3(4096), dim3(256), 0, s, (long long)A.nnz, A.a.p, dtab.p, nt, A.vidx.p, fail.p);
  int nfail = 0;
  d2h(&nfail, fail.p, sizeof(int), s);
  MI_HIP(hipStreamSynchronize(s));
  if (nfail) {  // a value beyond the sample's 256: plain stream
    A.vidx.release();
    return;
  }
  std::vector<double> lut(256, 0.0);
  std::memcpy(lut.data(), bits.data(), (size_t)nt * sizeof(double));
  A.vlut.upload(lut);
  A.val8 = true;
}

// block-coded column lists of an x-cache operator (DevCSR::ucode / ubase) from its 4-byte lists (MI_HYPRE_UCODE=0: keep
// those).  Per tile: at most 64 blocks of 1024 ids -> coded, bptr[tile] = start of its blocks in ubase; more (tiles at the
// seams of the internal numbering's cells and segments: a few per cent at 512^3) -> the tile keeps a 4-byte list, moved to a
// compact side list (A.ucols afterwards), bptr[tile] = -(its offset there) - 1.
static void build_block_coded_lists(DevCSR &A, DVec<long long> &bptr, hipStream_t s) {
  A.ucode.release();
  A.ubase.release();
  A.n_unique = (long long)A.ucols.n;
  A.ucode_max_blocks = 0;
  A.ucode_wide_tiles = 0;
  // OFF by default: measured at 512^3 (profiles/r04_ab_block_coded_lists.txt) the decode -- a wave-wide block table and one
  // ds_bpermute per column between the list load and the gather -- costs the tile Gauss-Seidel kernel and the residual SpMVs
  // of levels 1-2 more (+3..12 %) than the 2 bytes per unique column save; only the level-0 dictionary SpMV gains (-5 %)
  static const bool on = getenv("MI_HYPRE_UCODE") && atoi(getenv("MI_HYPRE_UCODE")) != 0;
  if (!on || !A.xcache || A.nblocks <= 0 || !A.uptr.p || A.ucols.n == 0) return;
  const int nt = A.nblocks;
  DVec<int> nblk((size_t)nt), maxblk(1);
  MI_HIP(hipMemsetAsync(maxblk.p, 0, sizeof(int), s));
  const dim3 grid((unsigned)((nt + 3) / 4));
  hipLaunchKernelGGL(ucode_blocks_k<false>, grid, dim3(256), 0, s, nt, A.uptr.p, A.ucols.p, nblk.p, nullptr, nullptr, nullptr, maxblk.p, nullptr);
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));  // (DVec::to_host copies on the null stream, which does not wait for this one)
  const std::vector<int> hn = nblk.to_host();   // (tiles: at most a few hundred thousand; through the pinned staging buffer)
  const std::vector<int> hu = A.uptr.to_host();
  std::vector<long long> hp((size_t)nt + 1, 0);
  long long nb_tot = 0, wide_tot = 0;
  for (int t = 0; t < nt; t++) {
    A.ucode_max_blocks = std::max(A.ucode_max_blocks, hn[(size_t)t]);
    if (hn[(size_t)t] <= 64) {
      hp[(size_t)t] = nb_tot;
      nb_tot += hn[(size_t)t];
    } else {
      hp[(size_t)t] = -wide_tot - 1;
      wide_tot += hu[(size_t)t + 1] - hu[(size_t)t];
      A.ucode_wide_tiles++;
    }
  }
  if (wide_tot * 2 > (long long)A.ucols.n) return;  // mostly wide tiles: not worth two formats -- the 4-byte lists stay
  bptr.alloc((size_t)nt + 1);
  MI_HIP(hipMemcpyAsync(bptr.p, hp.data(), ((size_t)nt + 1) * sizeof(long long), hipMemcpyHostToDevice, s));
  MI_HIP(hipStreamSynchronize(s));
  A.ubase.alloc((size_t)nb_tot, 64);  // (a wave loads 64 ints from a tile's first block on: padded)
  MI_HIP(hipMemsetAsync(A.ubase.p, 0, ((size_t)nb_tot + 64) * sizeof(int), s));
  A.ucode.alloc(A.ucols.n);
  DVec<int> wide((size_t)wide_tot);
  hipLaunchKernelGGL(ucode_blocks_k<true>, grid, dim3(256), 0, s, nt, A.uptr.p, A.ucols.p, nullptr, bptr.p, A.ubase.p, A.ucode.p, nullptr, wide.p);
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));
  A.ucols = std::move(wide);  // what is left of the 4-byte lists: the wide tiles'
}

void build_tile_desc(DevCSR &A, const long long *ia64, hipStream_t s) {
  A.tdesc.release();
  if (A.nblocks <= 0 || !A.rb.p) return;
  DVec<long long> bptr;
  build_block_coded_lists(A, bptr, s);
  A.tdesc.alloc((size_t)A.nblocks * 8);
  hipLaunchKernelGGL(tile_desc_k, dim3((unsigned)((A.nblocks + 255) / 256)), dim3(256), 0, s, A.nblocks, A.rb.p, ia64,
                     A.xcache ? A.uptr.p : nullptr, A.tdesc.p, A.ucode.p ? bptr.p : nullptr);
  MI_HIP(hipGetLastError());
  MI_HIP(hipStreamSynchronize(s));  // (bptr goes away)
  build_value_dictionary(A, s);
}

void spmv(const DevCSR &A, const double *x, double alpha, double beta, const double *b, double *y, hipStream_t s,
          int prof, const double *b_lo, int b_split) {
  EpiArgs e{};
  e.alpha = alpha;
  e.beta = beta;
  e.b = b;
  e.b_lo = b_lo;
  e.b_split = b_split;
  e.rowmap = A.rowmap.p;
  prof_begin(prof, s);
  prof_name(prof, launch_stream(0, A, x, y, e, s, prof == PROF_SPMV_L0));
  prof_end(prof, s);
}

void jacobi(const DevCSR &A, const double *u_old, double *u_new, const double *f, const double *offc, const double *d,
            const signed char *cf, int points, double w, hipStream_t s, int prof) {
  EpiArgs e{};
  e.alpha = w;
  e.b = f;
  e.offc = offc;
  e.d = d;
  e.cf = cf;
  e.points = points;
  prof_begin(prof, s);
  prof_name(prof, launch_stream(1, A, u_old, u_new, e, s));
  prof_end(prof, s);
}

void spmv_offd_add(const DevOffd &B, const double *xext, double alpha, double *y, hipStream_t s) {
  if (B.nrows_c == 0) return;
  hipLaunchKernelGGL(spmv_offd_k<0>, dim3((B.nrows_c + 255) / 256), dim3(256), 0, s, B.nrows_c, B.rows.p, B.ia.p,
                     B.ja.p, B.a.p, xext, alpha, y);
  MI_HIP(hipGetLastError());
}
void spmv_offd_set(const DevOffd &B, const double *xext, double *out, hipStream_t s) {
  if (B.nrows_c == 0) return;
  hipLaunchKernelGGL(spmv_offd_k<1>, dim3((B.nrows_c + 255) / 256), dim3(256), 0, s, B.nrows_c, B.rows.p, B.ia.p,
                     B.ja.p, B.a.p, xext, 1.0, out);
  MI_HIP(hipGetLastError());
}

bool gs_uses_tiles(const DevCSR &A, int chunk) {
  return chunk == 8 && A.gs_tiles && A.xcache && gs_tile_mode(A) && !gs_force_generic() && !gs_use_old();
}

// whether a sweep of A with zero_from == 0 never reads the pre-sweep vector (the tile kernel and the shuffle kernel
// skip every gather and take 0 for the rows' own values; the dense-chunk and the generic kernel read real zeros)
bool gs_ignores_zero_vector(const DevCSR &A, int chunk) {
  if (A.nrows == 0) return false;
  if (gs_uses_tiles(A, chunk)) return true;
  if (chunk != 8 || gs_force_generic()) return false;
  return (double)A.nnz / (double)A.nrows <= 8.0;  // gs_group_k (see the dispatch in gs_hybrid)
}

void gs_hybrid(const DevCSR &A, const double *u_lo, const double *u_hi, int split, double *out, const double *f,
               const double *offc, const double *d, const signed char *cf, int points, int chunk, bool fwd, bool bwd,
               double w, int row_begin, int row_end, hipStream_t s, int prof, int zero_from, double *tout,
               int t_from) {
  if (A.nrows == 0 || row_end <= row_begin) return;
  MI_REQUIRE(chunk >= 1 && chunk <= GS_MAX_CHUNK, "hybrid GS chunk out of range");
  // chunks that intersect [row_begin, row_end); pre-sweep values come from u_lo
  // (rows < split) / u_hi (rows >= split), the swept chunks' rows go to out
  const long long c0 = row_begin / chunk;
  const long long c1 = ((long long)row_end + chunk - 1) / chunk;
  const long long nch = c1 - c0;
  prof_begin(prof, s);
  if (gs_uses_tiles(A, chunk)) {
    // tiles that hold the chunks [c0, c1): first tile with rb[b+1] > c0*8, last tile with rb[b] < c1*8
    const std::vector<int> &rbh = A.rb_host;
    const int first_row = (int)(c0 * 8), last_row = (int)std::min<long long>(c1 * 8, A.nrows);
    const int b0 = (int)(std::upper_bound(rbh.begin(), rbh.end(), first_row) - rbh.begin()) - 1;
    const int b1 = (int)(std::lower_bound(rbh.begin(), rbh.end(), last_row) - rbh.begin());
#define GS_TILE_LAUNCH(V8_, BLOCK_)                                                                                  \
  prof_name(prof, V8_ ? (BLOCK_ == 512 ? "gs_tile_k<true, 512>" : "gs_tile_k<true, 256>")                             \
                      : (BLOCK_ == 512 ? "gs_tile_k<false, 512>" : "gs_tile_k<false, 256>"));                         \
  hipLaunchKernelGGL((gs_tile_k<V8_, BLOCK_>), dim3((unsigned)(b1 - b0)), dim3(BLOCK_), 0, s, b0, b1 - b0, A.tdesc.p, \
                     A.ia.p, A.a.p, A.ucols.p, A.lcol.p, cf, points, d, f, offc, u_lo, u_hi, split, out, fwd ? 1 : 0,  \
                     bwd ? 1 : 0, w, first_row, last_row, zero_from, tout, t_from, V8_ ? A.vidx.p : nullptr,          \
                     V8_ ? A.vlut.p : nullptr, A.ucode.p, A.ubase.p)
    const bool wide = A.tile_entries == SPMV_TILE_WIDE;
    if (b1 > b0 && A.val8 && wide) {
      GS_TILE_LAUNCH(true, SPMV_BLOCK_WIDE);
    } else if (b1 > b0 && A.val8) {
      GS_TILE_LAUNCH(true, SPMV_BLOCK);
    } else if (b1 > b0 && wide) {
      GS_TILE_LAUNCH(false, SPMV_BLOCK_WIDE);
    } else if (b1 > b0) {
      GS_TILE_LAUNCH(false, SPMV_BLOCK);
    }
#undef GS_TILE_LAUNCH
  } else if (chunk == 8 && !gs_force_generic()) {
    MI_REQUIRE(!A.big(), "an operator with 2^31 entries or more is swept by the tile Gauss-Seidel kernel only");
    prof_name(prof, "gs_group_k / gs_dense_k (chunk kernels)");
    const double avg = (double)A.nnz / (double)A.nrows;
    const int p95 = A.rowlen_p95;
#define GS_LAUNCH_K(KERNEL, LPC, E)                                                                             \
  {                                                                                                             \
    const long long waves = (nch + (64 / LPC) - 1) / (64 / LPC);                                                \
    hipLaunchKernelGGL((KERNEL<LPC, E>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, A.nrows, (int)c0,   \
                       (int)c1, A.ia.p, A.ja.p, A.a.p, cf, points, d, f, offc, u_lo, u_hi, split, out,          \
                       fwd ? 1 : 0, bwd ? 1 : 0, w, zero_from);                                                 \
  }
#define GS_LAUNCH(LPC, E) \
  if (gs_use_old()) GS_LAUNCH_K(gs_group_k, LPC, E) else GS_LAUNCH_K(gs_dense_k, LPC, E)
    // Measured per level (256^3 / 512^3 Laplacian hierarchies, profiles/compare_gs.*):
    //   mean row length <= 8 (the fine level): the shuffle kernel with 8 lanes per chunk is memory-bound
    //   already (3.9 TB/s) and beats the dense-chunk kernel, whose barriers and LDS block it does not need;
    //   longer rows: the dense-chunk kernel wins 10-18 %; for means up to 32 two chunks per wave (32 lanes
    //   each, rows beyond 32 entries finished in the pre-sweep loop) beat one chunk per wave.
    if (avg <= 8.0) {
      if (p95 <= 8) GS_LAUNCH_K(gs_group_k, 8, 1) else GS_LAUNCH_K(gs_group_k, 8, 2)
    } else if (avg <= 16.0) {
      if (p95 <= 16) GS_LAUNCH(16, 1) else GS_LAUNCH(16, 2)
    } else if (avg <= 32.0) {
      if (gs_use_old()) {
        if (p95 <= 32) GS_LAUNCH_K(gs_group_k, 32, 1) else GS_LAUNCH_K(gs_group_k, 64, 1)
      } else {
        if (p95 <= 64) GS_LAUNCH_K(gs_dense_k, 32, 1) else GS_LAUNCH_K(gs_dense_k, 32, 2)
      }
    } else {
      if (p95 <= 64) GS_LAUNCH(64, 1) else if (p95 <= 128) GS_LAUNCH(64, 2) else GS_LAUNCH(64, 4)
    }
#undef GS_LAUNCH_K
#undef GS_LAUNCH
  } else {
    MI_REQUIRE(!A.big(), "an operator with 2^31 entries or more is swept by the tile Gauss-Seidel kernel only");
    const size_t lds = (size_t)chunk * GS_BLOCK * sizeof(double);
    prof_name(prof, "gs_hybrid_k");
    hipLaunchKernelGGL(gs_hybrid_k, dim3((unsigned)((nch + GS_BLOCK - 1) / GS_BLOCK)), dim3(GS_BLOCK), lds, s,
                       A.nrows, (int)c0, (int)c1, chunk, A.ia.p, A.ja.p, A.a.p, cf, points, d, f, offc, u_lo, u_hi,
                       split, out, fwd ? 1 : 0, bwd ? 1 : 0, w);
  }
  MI_HIP(hipGetLastError());
  prof_end(prof, s);
}

// workgroups of the inner-product kernels (MI_HYPRE_DOT_BLOCKS, at most RED_MAX_BLOCKS * MASS_NV partial slots)
static int dot_grid(int n) {
  static const int cap = getenv("MI_HYPRE_DOT_BLOCKS") ? std::max(1, std::min(RED_MAX_BLOCKS * MASS_NV, atoi(getenv("MI_HYPRE_DOT_BLOCKS")))) : vec_grid_cap();
  long long want = ((long long)n + 511) / 512;
  if (want < 1) want = 1;
  if (want > cap) want = cap;
  return (int)want;
}

void dot(const double *x, const double *y, int n, double *out_dev, hipStream_t s) {
  const int g = dot_grid(n);
  double *partials = ctx().red_partials.p;
  prof_begin(PROF_DOT, s);
  hipLaunchKernelGGL(dot_partial_k, dim3(g), dim3(256), 0, s, x, y, n, partials, ctx().red_ticket.p, out_dev);
  prof_end(PROF_DOT, s);
  MI_HIP(hipGetLastError());
}

void axpy_dot(const double *alpha_dev, double scale_, const double *xa, double *y, const double *xd, int n,
              double *out_dev, hipStream_t s) {
  const int g = dot_grid(n);
  double *partials = ctx().red_partials.p;
  prof_begin(PROF_DOT, s);
  hipLaunchKernelGGL(axpy_dot_partial_k, dim3(g), dim3(256), 0, s, alpha_dev, scale_, xa, y, xd, n, partials,
                     ctx().red_ticket.p, out_dev);
  prof_end(PROF_DOT, s);
  MI_HIP(hipGetLastError());
}

void mass_dot(const double *const *vecs, int m, const double *w, int n, double *out_dev, hipStream_t s) {
  const int g = vec_grid(n);
  double *partials = ctx().red_partials.p;
  prof_begin(PROF_DOT, s);
  for (int j0 = 0; j0 < m; j0 += MASS_NV) {
    const int mm = std::min(MASS_NV, m - j0);
    MassPtrs P;
    for (int j = 0; j < MASS_NV; j++) P.p[j] = vecs[j0 + (j < mm ? j : 0)];
    hipLaunchKernelGGL(mass_dot_partial_k, dim3(g), dim3(256), 0, s, P, mm, w, n, partials);
    hipLaunchKernelGGL(reduce_final_multi_k, dim3(mm), dim3(256), 0, s, partials, g, out_dev + j0);
  }
  prof_end(PROF_DOT, s);
  MI_HIP(hipGetLastError());
}

void mass_axpy(const double *const *vecs, int m, const double *coef_dev, double scale_, double *w, int n,
               hipStream_t s) {
  if (n == 0) return;
  prof_begin(PROF_AXPY, s);
  for (int j0 = 0; j0 < m; j0 += MASS_NV) {
    const int mm = std::min(MASS_NV, m - j0);
    MassPtrs P;
    for (int j = 0; j < MASS_NV; j++) P.p[j] = vecs[j0 + (j < mm ? j : 0)];
    hipLaunchKernelGGL(mass_axpy_k, dim3(vec_grid(n)), dim3(256), 0, s, P, mm, coef_dev + j0, scale_, w, n);
  }
  prof_end(PROF_AXPY, s);
  MI_HIP(hipGetLastError());
}

void lin_comb(const double *const *vecs, const double *coef_host, int m, bool init, double *w, int n, hipStream_t s) {
  if (n == 0 || m == 0) return;
  prof_begin(PROF_AXPY, s);
  for (int j0 = 0; j0 < m; j0 += MASS_NV) {
    const int mm = std::min(MASS_NV, m - j0);
    MassPtrs P;
    MassCoef C;
    for (int j = 0; j < MASS_NV; j++) {
      P.p[j] = vecs[j0 + (j < mm ? j : 0)];
      C.c[j] = j < mm ? coef_host[j0 + j] : 0.0;
    }
    if (init && j0 == 0)
      hipLaunchKernelGGL(lin_comb_k<true>, dim3(vec_grid(n)), dim3(256), 0, s, P, mm, C, w, n);
    else
      hipLaunchKernelGGL(lin_comb_k<false>, dim3(vec_grid(n)), dim3(256), 0, s, P, mm, C, w, n);
  }
  prof_end(PROF_AXPY, s);
  MI_HIP(hipGetLastError());
}

void axpy(double alpha, const double *x, double *y, int n, hipStream_t s) {
  if (n == 0) return;
  prof_begin(PROF_AXPY, s);
  hipLaunchKernelGGL(axpy_k, dim3(vec_grid(n)), dim3(256), 0, s, (const double *)nullptr, alpha, x, y, n);
  prof_end(PROF_AXPY, s);
  MI_HIP(hipGetLastError());
}
void axpy_dev(const double *alpha_dev, double scale_, const double *x, double *y, int n, hipStream_t s) {
  if (n == 0) return;
  prof_begin(PROF_AXPY, s);
  hipLaunchKernelGGL(axpy_k, dim3(vec_grid(n)), dim3(256), 0, s, alpha_dev, scale_, x, y, n);
  prof_end(PROF_AXPY, s);
  MI_HIP(hipGetLastError());
}
void scale(double alpha, double *x, int n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(scale_k<0>, dim3(vec_grid(n)), dim3(256), 0, s, (const double *)nullptr, alpha, x, n);
  MI_HIP(hipGetLastError());
}
// x *= 1 / sqrt(*sumsq_dev) as scale_k<1>, and -- first wave of the first workgroup, before it touches x -- the `count`
// doubles at `slots` are stored into host memory the device can write (hipHostMalloc) followed by the number `seq` in a
// flag word, system-scope release: the host reads the Hessenberg column of an Arnoldi step by polling that word, while
// this kernel is still streaming the new basis vector, instead of a copy kernel and a stream synchronisation behind it
// (one idle queue of ~70 us per step, profiles/r03_gaps_512.txt).
__global__ __launch_bounds__(256) void scale_post_k(const double *__restrict__ sumsq_dev, double *__restrict__ x, int n,
                                                    const double *__restrict__ slots, int count,
                                                    double *__restrict__ host_out, unsigned long long *__restrict__ host_flag,
                                                    unsigned long long seq) {
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    for (int j = threadIdx.x; j < count; j += 64)
      __hip_atomic_store(host_out + j, slots[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // the wave's stores are out before the flag
    if (threadIdx.x == 0) __hip_atomic_store(host_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const double t = sumsq_dev[0];
  if (!(t > 0.0)) return;
  const double a = 1.0 / sqrt(t);
  const long long stride = (long long)gridDim.x * 512;
  long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
  for (; i + 1 < n; i += stride) {
    double2 v = *reinterpret_cast<double2 *>(x + i);
    v.x *= a;
    v.y *= a;
    *reinterpret_cast<double2 *>(x + i) = v;
  }
  if (i < n) x[i] *= a;
}
void scale_inv_sqrt_post(const double *sumsq_dev, double *x, int n, const double *slots, int count, double *host_out,
                         unsigned long long *host_flag, unsigned long long seq, hipStream_t s) {
  // (n == 0: a rank without rows still posts)
  hipLaunchKernelGGL(scale_post_k, dim3(n ? vec_grid(n) : 1), dim3(256), 0, s, sumsq_dev, x, n, slots, count, host_out,
                     host_flag, seq);
  MI_HIP(hipGetLastError());
}
void scale_inv_sqrt_dev(const double *sumsq_dev, double *x, int n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(scale_k<1>, dim3(vec_grid(n)), dim3(256), 0, s, sumsq_dev, 1.0, x, n);
  MI_HIP(hipGetLastError());
}
// Tabulation of a linear map column by column inside ONE captured graph (BoomerAMG::tabulate_cycle): the column number
// lives in device memory, so that every replay of the graph is the same sequence of launches.
//   tab_unit_k : e = unit vector number *col (the previous column's 1 is cleared)
//   tab_store_k: column *col of Bt = u, then ++*col (one workgroup: n is at most a few thousand)
__global__ __launch_bounds__(64) void tab_unit_k(double *__restrict__ e, const int *__restrict__ col) {
  if (threadIdx.x == 0) {
    const int j = *col;
    if (j > 0) e[j - 1] = 0.0;
    e[j] = 1.0;
  }
}
__global__ __launch_bounds__(256) void tab_store_k(double *__restrict__ Bt, const double *__restrict__ u, int n, int *__restrict__ col) {
  const int j = *col;
  for (int i = threadIdx.x; i < n; i += 256) Bt[(size_t)j * (size_t)n + i] = u[i];
  __syncthreads();
  if (threadIdx.x == 0) *col = j + 1;
}
// test support (HYPRE_MI_ArenaSelfTest): bytes of p[0..n) that differ from `tag`: their number and the first and last offset
__global__ __launch_bounds__(256) void bytes_differ_k(const unsigned char *__restrict__ p, size_t n, unsigned char tag,
                                                      unsigned long long *__restrict__ out) {
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  unsigned long long cnt = 0, first = ~0ull, last = 0;
  for (; i < n; i += (size_t)gridDim.x * 256)
    if (p[i] != tag) {
      cnt++;
      if (i < first) first = i;
      if (i > last) last = i;
    }
  if (cnt) {
    atomicAdd(out, cnt);
    atomicMin(out + 1, first);
    atomicMax(out + 2, last);
  }
}
__global__ __launch_bounds__(256) void fill_bytes_k(unsigned char *__restrict__ p, size_t n, unsigned char tag) {
  size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * 256) p[i] = tag;
}
void fill_bytes(unsigned char *p, size_t n, unsigned char tag, hipStream_t s) {
  hipLaunchKernelGGL(fill_bytes_k, dim3(2048), dim3(256), 0, s, p, n, tag);
  MI_HIP(hipGetLastError());
}
void bytes_differ(const unsigned char *p, size_t n, unsigned char tag, unsigned long long *out3, hipStream_t s) {
  hipLaunchKernelGGL(bytes_differ_k, dim3(2048), dim3(256), 0, s, p, n, tag, out3);
  MI_HIP(hipGetLastError());
}
void tab_unit(double *e, const int *col, hipStream_t s) {
  hipLaunchKernelGGL(tab_unit_k, dim3(1), dim3(64), 0, s, e, col);
  MI_HIP(hipGetLastError());
}
void tab_store(double *Bt, const double *u, int n, int *col, hipStream_t s) {
  hipLaunchKernelGGL(tab_store_k, dim3(1), dim3(256), 0, s, Bt, u, n, col);
  MI_HIP(hipGetLastError());
}
namespace {
__global__ void load_module_k(int *p) {
  if (p) *p = 0;
}
}  // namespace
void load_device_code(hipStream_t s) {  // see sk::load_device_code
  load_module_k<<<1, 1, 0, s>>>(nullptr);
  MI_HIP(hipGetLastError());
}
void fill(double *x, int n, double v, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(fill_k, dim3(vec_grid(n)), dim3(256), 0, s, x, n, v);
  MI_HIP(hipGetLastError());
}
void copy(const double *x, double *y, int n, hipStream_t s) {
  if (n == 0 || x == y) return;
  MI_HIP(hipMemcpyAsync(y, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
}
void gather2(const double *lo, const double *hi, int split, const int *map, double *out, int n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(gather2_k, dim3((n + 255) / 256), dim3(256), 0, s, lo, hi, split, map, out, n);
  MI_HIP(hipGetLastError());
}
void gather(const double *x, const int *map, double *out, int n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(gather_k, dim3((n + 255) / 256), dim3(256), 0, s, x, map, out, n);
  MI_HIP(hipGetLastError());
}
void scatter_set(double *x, const int *idx, const double *vals, int n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(scatter_k<0>, dim3((n + 255) / 256), dim3(256), 0, s, x, idx, vals, n);
  MI_HIP(hipGetLastError());
}
void scatter_add(double *x, const int *idx, const double *vals, int n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(scatter_k<1>, dim3((n + 255) / 256), dim3(256), 0, s, x, idx, vals, n);
  MI_HIP(hipGetLastError());
}
void two_stage_first(const double *r, const double *d, double *z, double *u, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(two_stage_first_k, dim3((n + 255) / 256), dim3(256), 0, s, n, r, d, z, u);
}
void two_stage_lower(const DevCSR &A, const double *d, const double *zin, double sign, double *zout, double *u,
                     hipStream_t s) {
  if (A.nrows <= 0) return;
  MI_REQUIRE(!A.big(), "two-stage Gauss-Seidel: operators with 2^31 entries or more are not supported");
  hipLaunchKernelGGL(two_stage_lower_k, dim3((A.nrows + 255) / 256), dim3(256), 0, s, A.nrows, A.ia.p, A.ja.p, A.a.p, d,
                     zin, sign, zout, u);
}

void ipc_exchange(const IpcBatch &b, unsigned long long spin_limit, int *error_flag, hipStream_t s) {
  int nb = 0;
  for (int i = 0; i < b.n; i++) nb += b.t[i].nblocks;
  if (nb == 0) return;
  hipLaunchKernelGGL(ipc_exchange_k, dim3((unsigned)nb), dim3(1024), 0, s, b, spin_limit, error_flag);
  MI_HIP(hipGetLastError());
}

void ipc_allreduce(const IpcAllreduce &a, unsigned long long spin_limit, int *error_flag, hipStream_t s) {
  hipLaunchKernelGGL(ipc_allreduce_k, dim3(1), dim3(64), 0, s, a, spin_limit, error_flag);
  MI_HIP(hipGetLastError());
}

void ipc_poison(double *buf, int count, const int *error_flag, hipStream_t s) {
  // (larger counts: the first 64 values are enough to turn every sum into NaN)
  hipLaunchKernelGGL(ipc_poison_k, dim3(1), dim3(64), 0, s, buf, std::min(count, 64), error_flag);
  MI_HIP(hipGetLastError());
}

void dense_matvec_t(const double *Mt, const double *f, double *u, int n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(dense_matvec_t_k, dim3((n + 15) / 16), dim3(1024), 0, s, Mt, f, u, n);
  MI_HIP(hipGetLastError());
}

void dense_matvec(const double *M, const double *f, double *u, int n, int m, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(dense_matvec_k, dim3((n + 255) / 256), dim3(256), 0, s, M, f, u, n, m);
  MI_HIP(hipGetLastError());
}

}  // namespace k
}  // namespace mi
