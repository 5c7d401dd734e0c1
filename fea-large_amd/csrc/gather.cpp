// gather.cpp -- one-time host construction of the maps of the GATHER assembly
// (kernels_gather.hip) for linear tetrahedra.
//
// The reference scatters every element matrix into a growing sparse matrix
// (sp_matrix_element_add, fea_solver.c:966,1055).  The mesh topology never
// changes, so the scatter is inverted once, here: for every off-diagonal 3x3
// block (a, b) of the matrix the list of (element, local row node, local column
// node) triples that contribute to it, grouped by chunks of consecutive block
// rows so that one workgroup finds everything it needs in one record:
//   header   rows, CSR range, counts
//   nodes    the nodes the chunk's elements touch, owned rows first
//   elems    the distinct elements touching the rows: 4 chunk-local node ids
//   tpos     per block thread: tile position (CSR order inside the chunk) of its block (a, b) and, when b is a
//            row of the chunk too, of the mirror block (b, a) = its transpose (one thread serves both)
//   rows     per row: first tile position, diagonal position, first residual thread
//   vlist    per residual thread a slice of ONE row's (element, local node) visits
//   clist    per block thread its contributions, ascending element order
// Every list is stored thread-minor ("transposed"): word k of thread t sits at
// [k][t], so a wave reads 64 consecutive words.
#include "feahip_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {
template <class F>
void par_chunks(int n, F f)
{
  unsigned hw = std::thread::hardware_concurrency();
  int nt = (int)std::min<unsigned>(hw ? hw : 4, 32);
  if (n < 512) nt = 1;
  if (nt <= 1) { f(0, n); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) {
    int lo = (int)((long long)n * t / nt), hi = (int)((long long)n * (t + 1) / nt);
    th.emplace_back([=] { f(lo, hi); });
  }
  for (auto &x : th) x.join();
}
inline int up(int v, int m) { return (v + m - 1) / m * m; }
}  // namespace

// u16 offsets inside the "rows" section
#define G_RS 0                        // rstart[MAX_ROWS + 1]
#define G_RD 20                       // rdiag[MAX_ROWS]
#define G_VF 40                       // vfirst[MAX_ROWS + 1]
#define G_ROWS_U16 64

void build_host_gather(int N, int E, const int *conn, const HostPattern &hp, int row_lo, int row_hi, HostGather &out)
{
  (void)E;
  out.ok = false; out.nchunks = 0; out.blob.clear(); out.first_row.clear();
  if (row_lo < 0 || row_hi > N || row_lo >= row_hi) return;
  // limits of one chunk; the element target keeps three workgroups' records in one CU's LDS
  int max_rows = FEA_G_MAX_ROWS, max_elems = 216, alpha = 24;
  if (const char *e = getenv("FEAHIP_GATHER_ROWS")) max_rows = std::max(1, std::min(FEA_G_MAX_ROWS, atoi(e)));
  if (const char *e = getenv("FEAHIP_GATHER_ELEMS")) max_elems = std::max(8, std::min(FEA_G_MAX_ELEMS, atoi(e)));
  if (const char *e = getenv("FEAHIP_GATHER_ALPHA")) alpha = std::max(0, atoi(e));
  const int nrows_all = row_hi - row_lo;

  // ---- pass A: chunk boundaries.  cost[i][l-1] = distinct elements touching rows [i, i+l) (0xFFFF: does not
  // fit).  Every element evaluation a chunk makes is work, so the partition that minimises their total (plus a
  // per-chunk overhead alpha) is found by a shortest-path recurrence over the rows; it finds the natural
  // clusters of whatever numbering the mesh came with (bricks, lines) instead of cutting through them.
  const int L = max_rows;
  std::vector<uint16_t> cost((size_t)nrows_all * L, 0xFFFFu);
  {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::min<unsigned>(hw ? hw : 4, 32);
    if (nrows_all < 4096) nt = 1;
    std::vector<std::thread> th;
    auto work = [&](int lo, int hi) {
      std::vector<int> nstamp((size_t)N, -1);
      for (int i = lo; i < hi; ++i) {
        const int r0 = row_lo + i;
        int nel = 0, nnod = 0, noffd = 0, nvis = 0;
        for (int l = 1; l <= L && r0 + l <= row_hi; ++l) {
          const int r = r0 + l - 1;
          if (nstamp[r] != r0) { nstamp[r] = r0; ++nnod; }
          for (int q = hp.incptr[r]; q < hp.incptr[r + 1]; ++q) {
            const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu), la = (int)(hp.inc_rows[q] >> 28);
            bool fresh = true;                     // new to the window unless another of its nodes is a row of it
            for (int k = 0; k < 4; ++k) {
              const int g = conn[(size_t)e * 4 + k];
              if (k != la && g >= r0 && g < r) fresh = false;
            }
            if (!fresh) continue;
            ++nel;
            for (int k = 0; k < 4; ++k) {
              const int g = conn[(size_t)e * 4 + k];
              if (nstamp[g] != r0) { nstamp[g] = r0; ++nnod; }
            }
          }
          noffd += hp.rowptr[r + 1] - hp.rowptr[r] - 1;
          nvis += hp.incptr[r + 1] - hp.incptr[r];
          const bool fits = nel <= (l > 1 ? max_elems : FEA_G_MAX_ELEMS) && nnod <= FEA_G_MAX_NODES &&
                            noffd <= FEA_G_THREADS && nvis <= 4 * FEA_G_THREADS;
          if (!fits) break;                        // every longer window fails too
          cost[(size_t)i * L + (l - 1)] = (uint16_t)nel;
        }
      }
    };
    if (nt <= 1) work(0, nrows_all);
    else {
      for (int t = 0; t < nt; ++t) {
        const int lo = (int)((long long)nrows_all * t / nt), hi = (int)((long long)nrows_all * (t + 1) / nt);
        th.emplace_back([=] { work(lo, hi); });
      }
      for (auto &x : th) x.join();
    }
  }
  {
    std::vector<long long> best((size_t)nrows_all + 1, -1);
    std::vector<unsigned char> from((size_t)nrows_all + 1, 0);
    best[0] = 0;
    for (int j = 1; j <= nrows_all; ++j) {
      long long b = -1; int bl = 0;
      for (int l = 1; l <= L && l <= j; ++l) {
        const uint16_t c = cost[(size_t)(j - l) * L + (l - 1)];
        if (c == 0xFFFFu || best[j - l] < 0) continue;
        const long long v = best[j - l] + c + alpha;
        if (b < 0 || v < b) { b = v; bl = l; }
      }
      if (b < 0) return;                           // a single row does not fit: no gather assembly for this mesh
      best[j] = b; from[j] = (unsigned char)bl;
    }
    std::vector<int> cuts;
    for (int j = nrows_all; j > 0; j -= from[j]) cuts.push_back(row_lo + j);
    cuts.push_back(row_lo);
    out.first_row.assign(cuts.rbegin(), cuts.rend());
  }
  std::vector<uint16_t>().swap(cost);
  const int nch = (int)out.first_row.size() - 1;

  // ---- pass B: per-chunk lists (parallel), first into per-chunk vectors to learn the depths
  struct Local {
    GatherHeader h;
    std::vector<int> nodes;
    std::vector<uint32_t> elems, tpos;
    std::vector<uint16_t> rows, vlist, clist;
  };
  std::vector<Local> loc((size_t)nch);
  std::vector<char> bad((size_t)nch, 0);
  par_chunks(nch, [&](int lo, int hi) {
    std::vector<int> el, halo, tid_of;
    std::vector<std::vector<uint16_t>> lists;
    for (int p = lo; p < hi; ++p) {
      Local &L = loc[p];
      const int r0 = out.first_row[p], r1 = out.first_row[p + 1], nrows = r1 - r0;
      const int b0 = hp.rowptr[r0], nb = hp.rowptr[r1] - b0;
      el.clear(); halo.clear();
      for (int q = hp.incptr[r0]; q < hp.incptr[r1]; ++q) el.push_back((int)(hp.inc_rows[q] & 0x0FFFFFFFu));
      std::sort(el.begin(), el.end());
      el.erase(std::unique(el.begin(), el.end()), el.end());
      for (int e : el)
        for (int k = 0; k < 4; ++k) {
          const int g = conn[(size_t)e * 4 + k];
          if (g < r0 || g >= r1) halo.push_back(g);
        }
      std::sort(halo.begin(), halo.end());
      halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
      const int nnode = nrows + (int)halo.size(), nelem = (int)el.size();
      if (nnode > FEA_G_MAX_NODES || nelem > FEA_G_MAX_ELEMS || nrows > FEA_G_MAX_ROWS) { bad[p] = 1; continue; }
      L.nodes.resize((size_t)nnode);
      for (int i = 0; i < nrows; ++i) L.nodes[i] = r0 + i;
      std::copy(halo.begin(), halo.end(), L.nodes.begin() + nrows);
      auto lid = [&](int g) -> int {
        if (g >= r0 && g < r1) return g - r0;
        return nrows + (int)(std::lower_bound(halo.begin(), halo.end(), g) - halo.begin());
      };
      L.elems.resize((size_t)nelem);
      for (int i = 0; i < nelem; ++i) {
        uint32_t w = 0;
        for (int k = 0; k < 4; ++k) w |= (uint32_t)lid(conn[(size_t)el[i] * 4 + k]) << (8 * k);
        L.elems[i] = w;
      }
      // block threads: the off-diagonal blocks in CSR order; a block whose column is a LOWER row of the same chunk
      // has no thread of its own, it is the transpose of its mirror block
      tid_of.assign((size_t)nb, -1);
      L.tpos.clear();
      L.rows.assign(G_ROWS_U16, 0);
      for (int a = r0; a < r1; ++a) {
        L.rows[G_RS + (a - r0)] = (uint16_t)(hp.rowptr[a] - b0);
        L.rows[G_RD + (a - r0)] = (uint16_t)(hp.diag[a] - b0);
        for (int q = hp.rowptr[a]; q < hp.rowptr[a + 1]; ++q) {
          const int b = hp.colidx[q];
          if (b == a || (b >= r0 && b < a)) continue;
          uint32_t w = (uint32_t)(q - b0) | 0xFFFF0000u;
          if (b > a && b < r1) {                   // mirror (b, a): position of a in row b
            const int *cb = hp.colidx.data() + hp.rowptr[b], *ce = hp.colidx.data() + hp.rowptr[b + 1];
            const int m = hp.rowptr[b] + (int)(std::lower_bound(cb, ce, a) - cb) - b0;
            w = (uint32_t)(q - b0) | ((uint32_t)m << 16);
          }
          tid_of[q - b0] = (int)L.tpos.size();
          L.tpos.push_back(w);
        }
      }
      L.rows[G_RS + nrows] = (uint16_t)nb;
      const int ntask = (int)L.tpos.size();
      if (ntask > FEA_G_THREADS) { bad[p] = 1; continue; }
      lists.assign((size_t)ntask, std::vector<uint16_t>());
      for (int a = r0; a < r1; ++a) {
        const int *cb = hp.colidx.data() + hp.rowptr[a], *ce = hp.colidx.data() + hp.rowptr[a + 1];
        for (int q = hp.incptr[a]; q < hp.incptr[a + 1]; ++q) {
          const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu), la = (int)(hp.inc_rows[q] >> 28);
          const int le = (int)(std::lower_bound(el.begin(), el.end(), e) - el.begin());
          for (int lb = 0; lb < 4; ++lb) {
            if (lb == la) continue;
            const int b = conn[(size_t)e * 4 + lb];
            if (b == a) continue;                 // degenerate element (repeated node): no off-diagonal block
            const int pos = hp.rowptr[a] + (int)(std::lower_bound(cb, ce, b) - cb) - b0;
            if (tid_of[pos] < 0) continue;        // served by the mirror block's thread
            lists[(size_t)tid_of[pos]].push_back((uint16_t)(le | (la << 8) | (lb << 10)));
          }
        }
      }
      int depth = 0;
      for (auto &l : lists) depth = std::max(depth, (int)l.size());
      const int dwords = (depth + 1) / 2, cstride = FEA_G_THREADS;
      // an empty slot points at the all-zero record the kernel keeps behind the last element: no branch in the sum
      L.clist.assign((size_t)dwords * 2 * cstride, (uint16_t)nelem);
      for (int t = 0; t < ntask; ++t)
        for (size_t k = 0; k < lists[t].size(); ++k)
          L.clist[((size_t)(k / 2) * cstride + t) * 2 + (k & 1)] = lists[t][k];
      // residual threads: slices of vdepth visits of one row
      int vdepth = 1;
      for (;; ++vdepth) {
        int need = 0;
        for (int a = r0; a < r1; ++a) need += (hp.incptr[a + 1] - hp.incptr[a] + vdepth - 1) / vdepth;
        if (need <= FEA_G_THREADS) break;
      }
      int nvthr = 0;
      for (int a = r0; a < r1; ++a) {
        L.rows[G_VF + (a - r0)] = (uint16_t)nvthr;
        nvthr += (hp.incptr[a + 1] - hp.incptr[a] + vdepth - 1) / vdepth;
      }
      L.rows[G_VF + nrows] = (uint16_t)nvthr;
      const int vstride = FEA_G_THREADS;
      L.vlist.assign((size_t)vdepth * vstride, (uint16_t)nelem);
      for (int a = r0; a < r1; ++a) {
        const int t0 = L.rows[G_VF + (a - r0)];
        int k = 0;
        for (int q = hp.incptr[a]; q < hp.incptr[a + 1]; ++q, ++k) {
          const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu), la = (int)(hp.inc_rows[q] >> 28);
          const int le = (int)(std::lower_bound(el.begin(), el.end(), e) - el.begin());
          L.vlist[(size_t)(k % vdepth) * vstride + t0 + k / vdepth] = (uint16_t)(le | (la << 8));
        }
      }
      GatherHeader &h = L.h;
      memset(&h, 0, sizeof(h));
      h.r0 = r0; h.r1 = r1; h.b0 = b0; h.nb = nb; h.nnode = nnode; h.nelem = nelem; h.noffd = ntask;
      h.depth = dwords; h.nvthr = nvthr; h.vdepth = vdepth;
    }
  });
  for (int p = 0; p < nch; ++p)
    if (bad[p]) return;

  // ---- layout: fixed section offsets, sized by the largest chunk
  int m_v = 0, m_c = 0, g_nodes = 0, g_elems = 0, g_tile = 0;
  GatherLayout &lay = out.lay;
  memset(&lay, 0, sizeof(lay));
  for (const Local &L : loc) {
    m_v = std::max(m_v, (int)L.vlist.size() * 2); m_c = std::max(m_c, (int)L.clist.size() * 2);
    g_nodes = std::max(g_nodes, L.h.nnode); g_elems = std::max(g_elems, L.h.nelem); g_tile = std::max(g_tile, L.h.nb);
    lay.max_tasks = std::max(lay.max_tasks, L.h.noffd); lay.max_depth = std::max(lay.max_depth, L.h.depth);
    lay.max_vthr = std::max(lay.max_vthr, L.h.nvthr); lay.max_vdepth = std::max(lay.max_vdepth, L.h.vdepth);
  }
  lay.max_nodes = up(g_nodes, 2); lay.max_elems = g_elems + 1; lay.max_tile = g_tile;   // + the zero record
  lay.o_nodes = 64;
  lay.o_elems = lay.o_nodes + up(4 * FEA_G_MAX_NODES, 64);
  lay.o_bpos = lay.o_elems + up(4 * FEA_G_THREADS, 64);
  lay.o_rows = lay.o_bpos + up(4 * FEA_G_THREADS, 64);
  lay.o_vlist = lay.o_rows + up(2 * G_ROWS_U16, 64);
  lay.o_clist = lay.o_vlist + up(m_v, 64);
  lay.stride = up(lay.o_clist + m_c, 128);
  if ((long long)nch * lay.stride > 0x7FFFFFFF00LL) return;
  out.blob.assign((size_t)nch * lay.stride, 0);
  par_chunks(nch, [&](int lo, int hi) {
    for (int p = lo; p < hi; ++p) {
      const Local &L = loc[p];
      unsigned char *rec = out.blob.data() + (size_t)p * lay.stride;
      memcpy(rec, &L.h, sizeof(GatherHeader));
      memcpy(rec + lay.o_nodes, L.nodes.data(), L.nodes.size() * 4);
      memcpy(rec + lay.o_elems, L.elems.data(), L.elems.size() * 4);
      memcpy(rec + lay.o_bpos, L.tpos.data(), L.tpos.size() * 4);
      memcpy(rec + lay.o_rows, L.rows.data(), L.rows.size() * 2);
      memcpy(rec + lay.o_vlist, L.vlist.data(), L.vlist.size() * 2);
      memcpy(rec + lay.o_clist, L.clist.data(), L.clist.size() * 2);
    }
  });
  out.nchunks = nch;
  out.ok = true;
}
