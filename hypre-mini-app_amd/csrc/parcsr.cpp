#include "parcsr.hpp"
#include "setup_kernels.hpp"

#include <algorithm>
#include <cstring>

#include "kernels.hpp"

namespace mi {

bool is_device_pointer(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // plain host memory: clear the sticky error
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

void ParVector::init(gidx s, gidx e, int nc) {
  start = s;
  end = e;
  n = (int)(e - s);
  ncomp = nc < 1 ? 1 : nc;
  cur = 0;
  d.alloc((size_t)n * ncomp);
  // stream-ordered: a null-stream memset is NOT ordered against the library's non-blocking stream and could
  // land after the first kernel that writes this vector (GMRES creates its basis vectors inside the solve)
  if (n) zero_on_stream(d.p, (size_t)n * ncomp * sizeof(double));
}

// nnz: entries of a block whose offsets stay 32-bit (the halo block; a single row); 0 = not checked -- the diagonal
// block's entry offsets are 64-bit (tile descriptors carry the high word, DevCSR)
void require_int32_block(int64_t nrows, int64_t nnz, const char *what) {
  if (nrows >= MAX_BLOCK_ENTRIES)
    fail(4, std::string(what) + ": " + std::to_string(nrows) +
                " local rows exceed the 32-bit local row ids of the solve format; split the rows over more ranks");
  if (nnz >= MAX_BLOCK_ENTRIES)
    fail(4, std::string(what) + ": " + std::to_string(nnz) + " entries exceed the 32-bit entry offsets of this block "
                "(limit " + std::to_string(MAX_BLOCK_ENTRIES) + "); split the rows over more ranks");
}

// ------------------------------------------------------------------ IJ assembly
// HYPRE_IJMatrixAssemble semantics: entries with a column in
// [jlower, jupper] go to diag, the rest to offd; for one (row, col) the batches
// are applied in submission order, Set overwrites and AddTo accumulates.
void assemble_parcsr(Comm &comm, gidx ilower, gidx iupper, gidx jlower, gidx jupper,
                     std::vector<IJEntryBatch> &batches, ParCSR &out) {
  require_int32_block(iupper - ilower + 1, 0, "IJMatrixAssemble");
  const int nrows = (int)(iupper - ilower + 1);
  const int ncols_loc = (int)(jupper - jlower + 1);
  std::vector<int64_t> ia((size_t)nrows + 1, 0);
  int64_t total = 0;
  for (auto &b : batches) total += (int64_t)b.rows.size();
  std::vector<gidx> cj;
  std::vector<double> cv;
  std::vector<char> cadd;
  // Fast path: the entries arrive in row order (generators, row-sorted files) -- they already are the CSR
  // entry arrays, only the row pointers have to be found.  Checked in parallel; anything else takes the
  // counting sort below.
  bool in_row_order = total > 0;
  {
    gidx prev_last = ilower;
    for (auto &b : batches) {
      if (b.rows.empty()) continue;
      if (b.rows.front() < prev_last) in_row_order = false;
      prev_last = b.rows.back();
    }
    if (in_row_order)
      for (auto &b : batches) {
        const int64_t m = (int64_t)b.rows.size();
        std::vector<char> bad((size_t)host_threads() + 1, 0);
        parallel_for(m, [&](int64_t k0, int64_t k1, int t) {
          for (int64_t k = k0; k < k1; k++) {
            const gidx r = b.rows[(size_t)k];
            if (r < ilower || r > iupper || (k > 0 && b.rows[(size_t)k - 1] > r)) {
              bad[(size_t)t] = 1;
              return;
            }
          }
        });
        for (char f : bad) in_row_order = in_row_order && !f;
        if (!in_row_order) break;
      }
  }
  if (in_row_order) {
    // row pointers: ia[r + 1] = one past the last entry of row r
    std::vector<int64_t> last((size_t)nrows, -1);
    int64_t off = 0;
    for (auto &b : batches) {
      const int64_t m = (int64_t)b.rows.size();
      parallel_for(m, [&](int64_t k0, int64_t k1, int) {
        for (int64_t k = k0; k < k1; k++)
          if (k + 1 == m || b.rows[(size_t)k + 1] != b.rows[(size_t)k]) {
            int64_t &slot = last[(size_t)(b.rows[(size_t)k] - ilower)];
            slot = std::max(slot, off + k + 1);  // a row may continue in a later batch: batches are visited in order
          }
      });
      off += m;
    }
    for (int i = 0; i < nrows; i++) ia[(size_t)i + 1] = (last[(size_t)i] >= 0) ? last[(size_t)i] : ia[(size_t)i];
    bool same_add = true;
    for (auto &b : batches) same_add = same_add && (b.add == batches.front().add);
    if (batches.size() == 1) {
      cj.swap(batches[0].cols);
      cv.swap(batches[0].vals);
    } else {
      cj.resize((size_t)total);
      cv.resize((size_t)total);
      int64_t o = 0;
      for (auto &b : batches) {
        std::copy(b.cols.begin(), b.cols.end(), cj.begin() + o);
        std::copy(b.vals.begin(), b.vals.end(), cv.begin() + o);
        o += (int64_t)b.rows.size();
      }
    }
    cadd.resize((size_t)total);
    {
      int64_t o = 0;
      for (auto &b : batches) {
        const int64_t m = (int64_t)b.rows.size();
        const char f = b.add ? 1 : 0;
        parallel_for(m, [&](int64_t k0, int64_t k1, int) { memset(cadd.data() + o + k0, f, (size_t)(k1 - k0)); });
        o += m;
      }
    }
    (void)same_add;
    for (auto &b : batches) {
      std::vector<gidx>().swap(b.rows);
      std::vector<gidx>().swap(b.cols);
      std::vector<double>().swap(b.vals);
    }
  } else {
    for (auto &b : batches)
      for (size_t k = 0; k < b.rows.size(); k++) {
        const gidx r = b.rows[k] - ilower;
        if (r < 0 || r >= nrows) fail(4, "IJMatrix: row " + std::to_string(b.rows[k]) + " is not owned by this rank");
        ia[(size_t)r + 1]++;
      }
    for (int i = 0; i < nrows; i++) ia[(size_t)i + 1] += ia[(size_t)i];
    cj.resize((size_t)total);
    cv.resize((size_t)total);
    cadd.resize((size_t)total);
    std::vector<int64_t> pos(ia.begin(), ia.end() - 1);
    for (auto &b : batches) {
      for (size_t k = 0; k < b.rows.size(); k++) {
        const int64_t p = pos[(size_t)(b.rows[k] - ilower)]++;
        cj[(size_t)p] = b.cols[k];
        cv[(size_t)p] = b.vals[k];
        cadd[(size_t)p] = b.add ? 1 : 0;
      }
      std::vector<gidx>().swap(b.rows);
      std::vector<gidx>().swap(b.cols);
      std::vector<double>().swap(b.vals);
    }
  }
  batches.clear();
  // per row: stable sort by column, fold duplicates in submission order
  std::vector<int> ndiag((size_t)nrows, 0), noffd((size_t)nrows, 0), rowlen((size_t)nrows, 0);
  parallel_for(nrows, [&](int64_t b, int64_t e, int) {
    std::vector<int> perm;
    std::vector<gidx> tj;
    std::vector<double> tv;
    std::vector<char> ta;
    for (int64_t i = b; i < e; i++) {
      const int64_t s = ia[(size_t)i], len = ia[(size_t)i + 1] - s;
      bool sorted = true;
      for (int64_t k = 1; k < len; k++)
        if (cj[(size_t)(s + k)] <= cj[(size_t)(s + k - 1)]) {
          sorted = false;
          break;
        }
      int64_t m = len;
      if (!sorted) {
        perm.resize((size_t)len);
        for (int64_t k = 0; k < len; k++) perm[(size_t)k] = (int)k;
        std::stable_sort(perm.begin(), perm.end(),
                         [&](int x, int y) { return cj[(size_t)(s + x)] < cj[(size_t)(s + y)]; });
        tj.resize((size_t)len);
        tv.resize((size_t)len);
        ta.resize((size_t)len);
        for (int64_t k = 0; k < len; k++) {
          tj[(size_t)k] = cj[(size_t)(s + perm[(size_t)k])];
          tv[(size_t)k] = cv[(size_t)(s + perm[(size_t)k])];
          ta[(size_t)k] = cadd[(size_t)(s + perm[(size_t)k])];
        }
        m = 0;
        for (int64_t k = 0; k < len; k++) {
          if (m > 0 && cj[(size_t)(s + m - 1)] == tj[(size_t)k]) {
            cv[(size_t)(s + m - 1)] = ta[(size_t)k] ? cv[(size_t)(s + m - 1)] + tv[(size_t)k] : tv[(size_t)k];
          } else {
            cj[(size_t)(s + m)] = tj[(size_t)k];
            cv[(size_t)(s + m)] = tv[(size_t)k];
            m++;
          }
        }
      }
      rowlen[(size_t)i] = (int)m;
      int nd = 0;
      for (int64_t k = 0; k < m; k++) {
        const gidx c = cj[(size_t)(s + k)];
        if (c >= jlower && c <= jupper) nd++;
      }
      ndiag[(size_t)i] = nd;
      noffd[(size_t)i] = (int)m - nd;
    }
  });
  HostCSR &D = out.diag, &O = out.offd;
  D.nrows = O.nrows = nrows;
  D.ncols = ncols_loc;
  D.ia.assign((size_t)nrows + 1, 0);
  O.ia.assign((size_t)nrows + 1, 0);
  for (int i = 0; i < nrows; i++) {
    D.ia[(size_t)i + 1] = D.ia[(size_t)i] + ndiag[(size_t)i];
    O.ia[(size_t)i + 1] = O.ia[(size_t)i] + noffd[(size_t)i];
  }
  require_int32_block(nrows, O.nnz(), "IJMatrixAssemble (off-diagonal block)");  // the diagonal block: 64-bit offsets
  D.ja.resize((size_t)D.nnz());
  D.a.resize((size_t)D.nnz());
  O.ja.resize((size_t)O.nnz());
  O.a.resize((size_t)O.nnz());
  std::vector<gidx> ogid((size_t)O.nnz());
  parallel_for(nrows, [&](int64_t b, int64_t e, int) {
    for (int64_t i = b; i < e; i++) {
      const int64_t s = ia[(size_t)i];
      int64_t pd = D.ia[(size_t)i], po = O.ia[(size_t)i];
      for (int k = 0; k < rowlen[(size_t)i]; k++) {
        const gidx c = cj[(size_t)(s + k)];
        if (c >= jlower && c <= jupper) {
          D.ja[(size_t)pd] = (int)(c - jlower);
          D.a[(size_t)pd] = cv[(size_t)(s + k)];
          pd++;
        } else {
          ogid[(size_t)po] = c;
          O.a[(size_t)po] = cv[(size_t)(s + k)];
          po++;
        }
      }
    }
  });
  std::vector<gidx>().swap(cj);
  std::vector<double>().swap(cv);
  std::vector<char>().swap(cadd);
  out.col_map_offd = ogid;
  std::sort(out.col_map_offd.begin(), out.col_map_offd.end());
  out.col_map_offd.erase(std::unique(out.col_map_offd.begin(), out.col_map_offd.end()), out.col_map_offd.end());
  for (size_t k = 0; k < ogid.size(); k++)
    O.ja[k] = (int)(std::lower_bound(out.col_map_offd.begin(), out.col_map_offd.end(), ogid[k]) -
                    out.col_map_offd.begin());
  O.ncols = (int)out.col_map_offd.size();
  out.row_start = ilower;
  out.row_end = iupper + 1;
  out.nrows = nrows;
  // global partition
  std::vector<gidx> starts((size_t)comm.size);
  gidx mine = ilower;
  comm.allgather_host(&mine, starts.data(), sizeof(gidx));
  gidx gend = iupper + 1;
  comm.allreduce_host(&gend, 1, CommDType::I64, CommOp::MAX);
  out.row_starts = starts;
  out.row_starts.push_back(gend);
  static unsigned long long stamp = 0;
  out.assembly_stamp = ++stamp;
}

// ------------------------------------------------------------------ halo plan + device mirror
void ParCSR::build_halo_plan(Comm &comm) {
  halo = HaloPlan();
  if (comm.size == 1) {
    MI_REQUIRE(col_map_offd.empty(), "matrix has columns outside the single rank's range");
  } else {
    // receive side: col_map_offd is sorted and the partition is contiguous, so
    // the halo columns of one owner are contiguous in x_ext
    const std::vector<gidx> &cpart = col_partition();
    const gidx my_c0 = cpart[(size_t)comm.rank], my_c1 = cpart[(size_t)comm.rank + 1];
    halo.recv_starts.push_back(0);
    for (size_t k = 0; k < col_map_offd.size(); k++) {
      const gidx g = col_map_offd[k];
      const int owner = (int)(std::upper_bound(cpart.begin(), cpart.end(), g) - cpart.begin()) - 1;
      MI_REQUIRE(owner >= 0 && owner < comm.size && owner != comm.rank, "halo column without an owner");
      if (halo.recv_peers.empty() || halo.recv_peers.back() != owner) {
        if (!halo.recv_peers.empty()) halo.recv_starts.push_back((int)k);
        halo.recv_peers.push_back(owner);
      }
    }
    if (!halo.recv_peers.empty()) halo.recv_starts.push_back((int)col_map_offd.size());
    // tell each owner which of its rows we need
    std::vector<std::vector<char>> req(halo.recv_peers.size());
    for (size_t i = 0; i < halo.recv_peers.size(); i++) {
      const int b = halo.recv_starts[i], e = halo.recv_starts[i + 1];
      req[i].resize((size_t)(e - b) * sizeof(gidx));
      memcpy(req[i].data(), col_map_offd.data() + b, req[i].size());
    }
    std::vector<int> from;
    std::vector<std::vector<char>> got;
    comm.exchange_host(halo.recv_peers, req, from, got);
    halo.send_starts.push_back(0);
    for (size_t i = 0; i < from.size(); i++) {
      const size_t cnt = got[i].size() / sizeof(gidx);
      const gidx *g = reinterpret_cast<const gidx *>(got[i].data());
      for (size_t k = 0; k < cnt; k++) {
        MI_REQUIRE(g[k] >= my_c0 && g[k] < my_c1, "peer requested an entry this rank does not own");
        halo.send_map.push_back((int)(g[k] - my_c0));
      }
      halo.send_peers.push_back(from[i]);
      halo.send_starts.push_back((int)halo.send_map.size());
    }
  }
}

void ParCSR::to_device() {
  ensure_init();
  MI_REQUIRE(!host_diag_stale, "to_device: the host copy of the diag block was not built");
  // large blocks: raw upload, then row-block schedule inputs, x cache and Gauss-Seidel code bits on the device
  // (sk::to_solve_format); small ones through the host builder (DevCSR::upload) -- same result
  static const long long dev_min = getenv("MI_HYPRE_DEVICE_FORMAT_MIN_NNZ") ? atoll(getenv("MI_HYPRE_DEVICE_FORMAT_MIN_NNZ")) : 4000000;
  if (diag.nnz() >= dev_min) {
    sk::DCsr raw;
    raw.upload(diag, ctx().stream);
    sk::to_solve_format(raw, d_diag, ctx().stream);
  } else {
    d_diag.upload(diag);
  }
  to_device_halo();
}

void ParCSR::to_device_halo() {
  ensure_init();
  d_offd.upload(nrows, offd);
  halo.d_send_map.upload(halo.send_map);
  halo.d_send_buf.alloc(halo.send_map.size());
  halo.d_xext.alloc(col_map_offd.size());
  if (!col_map_offd.empty()) {
    d_offc.alloc((size_t)nrows);
    zero_on_stream(d_offc.p, (size_t)nrows * sizeof(double));
  }
  on_device = true;
}

void ParCSR::halo_pack(const double *x, hipStream_t s, const double *x_hi, int split) {
  if (halo.nsend() == 0) return;
  if (x_hi)
    k::gather2(x, x_hi, split, halo.d_send_map.p, halo.d_send_buf.p, halo.nsend(), s);
  else
    k::gather(x, halo.d_send_map.p, halo.d_send_buf.p, halo.nsend(), s);
}

void ParCSR::halo_transfer(Comm &comm, hipStream_t s) {
  std::vector<PeerBuf> sb, rb;
  for (size_t i = 0; i < halo.send_peers.size(); i++)
    sb.push_back({halo.send_peers[i], halo.d_send_buf.p + halo.send_starts[i],
                  (size_t)(halo.send_starts[i + 1] - halo.send_starts[i]) * sizeof(double)});
  for (size_t i = 0; i < halo.recv_peers.size(); i++)
    rb.push_back({halo.recv_peers[i], halo.d_xext.p + halo.recv_starts[i],
                  (size_t)(halo.recv_starts[i + 1] - halo.recv_starts[i]) * sizeof(double)});
  comm.exchange_dev(sb, rb, s);
  ctx().n_halo_exchange++;
}

void ParCSR::halo_exchange(Comm &comm, const double *x, hipStream_t s, const double *x_hi, int split) {
  if (comm.size == 1) return;
  if (halo.send_peers.empty() && halo.recv_peers.empty()) return;
  halo_pack(x, s, x_hi, split);
  halo_transfer(comm, s);
}

std::vector<int> ParCSR::halo_exchange_host_int(Comm &comm, const std::vector<int> &local) const {
  std::vector<int> ext(col_map_offd.size(), 0);
  if (comm.size == 1) return ext;
  std::vector<std::vector<char>> send(halo.send_peers.size());
  for (size_t i = 0; i < halo.send_peers.size(); i++) {
    const int b = halo.send_starts[i], e = halo.send_starts[i + 1];
    send[i].resize((size_t)(e - b) * sizeof(int));
    int *p = reinterpret_cast<int *>(send[i].data());
    for (int k = b; k < e; k++) p[k - b] = local[(size_t)halo.send_map[(size_t)k]];
  }
  std::vector<int> from;
  std::vector<std::vector<char>> got;
  comm.exchange_host(halo.send_peers, send, from, got);
  for (size_t i = 0; i < from.size(); i++) {
    size_t pi = 0;
    while (pi < halo.recv_peers.size() && halo.recv_peers[pi] != from[i]) pi++;
    MI_REQUIRE(pi < halo.recv_peers.size(), "unexpected halo sender");
    const int b = halo.recv_starts[pi], e = halo.recv_starts[pi + 1];
    MI_REQUIRE(got[i].size() == (size_t)(e - b) * sizeof(int), "halo message size mismatch");
    memcpy(ext.data() + b, got[i].data(), got[i].size());
  }
  return ext;
}

void ParCSR::matvec(Comm &comm, double alpha, const double *x, double beta, const double *b, double *y,
                    hipStream_t s, int prof, const DevCSR *diag_op, const double *b_lo, int b_split) {
  MI_REQUIRE(on_device, "matrix not assembled");
  const bool halo_on = comm.size > 1 && d_offd.nrows_c > 0;
  // The diag-block product does not need the halo: the neighbour exchange runs beside it on the side stream
  // (pack on s -> event -> exchange on comm_stream -> event -> halo-block product on s).
  // MI_HYPRE_OVERLAP_HALO=0: everything in order on s.
  static const bool overlap = !(getenv("MI_HYPRE_OVERLAP_HALO") && atoi(getenv("MI_HYPRE_OVERLAP_HALO")) == 0);
  const bool mine = comm.size > 1 && !(halo.send_peers.empty() && halo.recv_peers.empty());
  if (mine && !overlap) halo_exchange(comm, x, s);
  if (mine && overlap) {
    ctx().n_matvec_overlapped++;
    halo_pack(x, s, nullptr, 0);
    MI_HIP(hipEventRecord(ctx().ev_packed, s));
  }
  k::spmv(diag_op ? *diag_op : d_diag, x, alpha, beta, b, y, s, prof, b_lo, b_split);
  if (mine && overlap) {
    hipStream_t cs = ctx().comm_stream;
    MI_HIP(hipStreamWaitEvent(cs, ctx().ev_packed, 0));
    halo_transfer(comm, cs);
    MI_HIP(hipEventRecord(ctx().ev_halo, cs));
    MI_HIP(hipStreamWaitEvent(s, ctx().ev_halo, 0));
  }
  if (halo_on) k::spmv_offd_add(d_offd, halo.d_xext.p, alpha, y, s);
}

void ParCSR::matvec_ext_ready(double alpha, const double *x, double beta, const double *b, double *y, hipStream_t s) {
  MI_REQUIRE(on_device, "matrix not assembled");
  k::spmv(d_diag, x, alpha, beta, b, y, s, -1);
  if (d_offd.nrows_c > 0) k::spmv_offd_add(d_offd, halo.d_xext.p, alpha, y, s);
}

const double *ParCSR::offd_contrib(Comm &comm, const double *x, hipStream_t s, const double *x_hi, int split) {
  if (comm.size == 1) return nullptr;
  halo_exchange(comm, x, s, x_hi, split);
  if (d_offd.nrows_c == 0) return nullptr;
  k::spmv_offd_set(d_offd, halo.d_xext.p, d_offc.p, s);
  return d_offc.p;
}

void par_dot(Comm &comm, const double *x, const double *y, int n, double *out_dev, hipStream_t s) {
  k::dot(x, y, n, out_dev, s);
  if (comm.size > 1) comm.allreduce_dev(out_dev, 1, CommDType::F64, CommOp::SUM, s), ctx().n_allreduce++;
}

double par_dot_host(Comm &comm, const double *x, const double *y, int n, hipStream_t s) {
  Ctx &c = ctx();
  double *slot = c.red_out.p + 255;
  par_dot(comm, x, y, n, slot, s);
  d2h(c.h_pinned + 255, slot, sizeof(double), s);
  MI_HIP(hipStreamSynchronize(s));
  return c.h_pinned[255];
}

}  // namespace mi
