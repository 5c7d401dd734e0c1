// gather10.cpp -- one-time host construction of the maps of the GATHER assembly
// of elements with several Gauss points: 10-node tetrahedra, 8-node bricks
// (kernels_gather10.hip).
//
// The reference integrates a 30x30 element matrix Gauss point by Gauss point
// and scatters it (fea_solver.c:887-1068, sp_matrix_element_add :966,1055).
// As for the 4-node element (gather.cpp) the scatter is inverted once: a chunk
// of consecutive block rows gets one record that lists
//   header   rows, CSR range, counts, the rows of every write-out pass
//   elems    its distinct elements, as indices into the rank's element list (the state kernel's output order)
//   rows     per row: first tile position, diagonal position, first residual lane
//   tpos     per thread and block slot: tile position of the block (a, b) and of its mirror (b, a) when b is a
//            row of the chunk too (one thread serves both, the mirror is the transpose)
//   flist    per visit lane a slice of ONE row's (element, local node) visits: residual and diagonal block
//   clist    per thread and block slot the (element, local row node, local column node) contributions
// Blocks are dealt to the threads longest list first, so the 64 blocks a wave works on at a time have lists of
// (nearly) the same length.  Lists are stored thread-minor: a wave reads 64 consecutive words.
#include "feahip_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {
template <class F>
void par_chunks10(int n, F f)
{
  unsigned hw = std::thread::hardware_concurrency();
  int nt = (int)std::min<unsigned>(hw ? hw : 4, 32);
  if (n < 256) nt = 1;
  if (nt <= 1) { f(0, n); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) {
    int lo = (int)((long long)n * t / nt), hi = (int)((long long)n * (t + 1) / nt);
    th.emplace_back([=] { f(lo, hi); });
  }
  for (auto &x : th) x.join();
}
inline int up10(int v, int m) { return (v + m - 1) / m * m; }
}  // namespace

#define Q_RS 0
#define Q_RD 66
#define Q_FF 130
#define Q_MAX_TASKS (FEA_Q_SLOTS * FEA_Q_THREADS)
#define Q_FENT 8                        // visits per residual lane at most (4 words)

void build_host_gather10(int N, int E, int npe, const int *conn, const HostPattern &hp, int row_lo, int row_hi, HostGather10 &out)
{
  (void)E; (void)N;
  out.ok = false; out.nchunks = 0; out.blob.clear(); out.first_row.clear(); out.npe = npe;
  if (npe < 2 || npe > 15) return;                     // 4-bit local node ids
  if (row_lo < 0 || row_hi > N || row_lo >= row_hi) return;
  // limits of one chunk: two workgroups' records (496 bytes per element) in one CU's LDS
  int max_rows = FEA_Q_MAX_ROWS, max_elems = 127, alpha = 8;
  if (const char *e = getenv("FEAHIP_GATHER10_ROWS")) max_rows = std::max(1, std::min(FEA_Q_MAX_ROWS, atoi(e)));
  if (const char *e = getenv("FEAHIP_GATHER10_ELEMS")) max_elems = std::max(4, std::min(FEA_Q_MAX_ELEMS, atoi(e)));
  if (const char *e = getenv("FEAHIP_GATHER10_ALPHA")) alpha = std::max(0, atoi(e));
  const int tile_blocks = (max_elems * (3 * npe + 1) * 16) / 72 - 1;   // the K tile takes the records' place
  const int nrows_all = row_hi - row_lo;

  // ---- pass A: chunk boundaries by the shortest-path recurrence of gather.cpp (cost = element evaluations)
  const int L = max_rows;
  std::vector<uint16_t> cost((size_t)nrows_all * L, 0xFFFFu);
  par_chunks10(nrows_all, [&](int lo, int hi) {
    for (int i = lo; i < hi; ++i) {
      const int r0 = row_lo + i;
      int nel = 0, ntask = 0, nb = 0, nfl = 0;
      for (int l = 1; l <= L && r0 + l <= row_hi; ++l) {
        const int r = r0 + l - 1;
        for (int q = hp.incptr[r]; q < hp.incptr[r + 1]; ++q) {
          const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu), la = (int)(hp.inc_rows[q] >> 28);
          bool fresh = true;
          for (int k = 0; k < npe; ++k) {
            const int g = conn[(size_t)e * npe + k];
            if (k != la && g >= r0 && g < r) fresh = false;
          }
          if (fresh) ++nel;
        }
        const int *cb = hp.colidx.data() + hp.rowptr[r], *ce = hp.colidx.data() + hp.rowptr[r + 1];
        const int rowlen = (int)(ce - cb);
        ntask += rowlen - 1 - (int)(std::lower_bound(cb, ce, r) - std::lower_bound(cb, ce, r0));
        nb += rowlen;
        nfl += std::max(1, (hp.incptr[r + 1] - hp.incptr[r] + Q_FENT - 1) / Q_FENT);
        const bool fits = nel <= (l > 1 ? max_elems : FEA_Q_MAX_ELEMS) &&
                          ntask <= Q_MAX_TASKS && nfl <= FEA_Q_FLANES && rowlen <= tile_blocks / 2 &&
                          nb <= (FEA_Q_MAX_PASS - 2) * tile_blocks && nb < 0xFFFF;
        if (!fits) break;
        cost[(size_t)i * L + (l - 1)] = (uint16_t)nel;
      }
    }
  });
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
      if (b < 0) return;                           // a single row does not fit
      best[j] = b; from[j] = (unsigned char)bl;
    }
    std::vector<int> cuts;
    for (int j = nrows_all; j > 0; j -= from[j]) cuts.push_back(row_lo + j);
    cuts.push_back(row_lo);
    out.first_row.assign(cuts.rbegin(), cuts.rend());
  }
  std::vector<uint16_t>().swap(cost);
  const int nch = (int)out.first_row.size() - 1;

  // the rank's elements: everything touching its rows, ascending; the state kernel evaluates them in this order
  {
    out.elist.clear();
    for (int q = hp.incptr[row_lo]; q < hp.incptr[row_hi]; ++q) out.elist.push_back((int)(hp.inc_rows[q] & 0x0FFFFFFFu));
    std::sort(out.elist.begin(), out.elist.end());
    out.elist.erase(std::unique(out.elist.begin(), out.elist.end()), out.elist.end());
  }
  // ---- pass B: per-chunk lists
  struct Local {
    Gather10Header h;
    std::vector<uint32_t> elems, tpos;
    std::vector<uint16_t> rows, flist;
    std::vector<std::vector<uint16_t>> lists;        // per task, in thread order (task i: thread i % 256, slot i / 256)
  };
  std::vector<Local> loc((size_t)nch);
  std::vector<char> bad((size_t)nch, 0);
  const int elem_cap = std::max(max_elems, 1);
  par_chunks10(nch, [&](int lo, int hi) {
    std::vector<int> el, tid_of, order;
    std::vector<uint32_t> tp;
    std::vector<std::vector<uint16_t>> lists;
    for (int p = lo; p < hi; ++p) {
      Local &Lc = loc[p];
      const int r0 = out.first_row[p], r1 = out.first_row[p + 1], nrows = r1 - r0;
      const int b0 = hp.rowptr[r0], nb = hp.rowptr[r1] - b0;
      el.clear();
      for (int q = hp.incptr[r0]; q < hp.incptr[r1]; ++q) el.push_back((int)(hp.inc_rows[q] & 0x0FFFFFFFu));
      std::sort(el.begin(), el.end());
      el.erase(std::unique(el.begin(), el.end()), el.end());
      const int nelem = (int)el.size();
      if (nelem > FEA_Q_MAX_ELEMS || nrows > FEA_Q_MAX_ROWS || nb >= 0xFFFF) { bad[p] = 1; continue; }
      auto lelem = [&](int e) { return (int)(std::lower_bound(el.begin(), el.end(), e) - el.begin()); };
      tid_of.assign((size_t)nb, -1);
      tp.clear();
      Lc.rows.assign(FEA_Q_ROWS_U16, 0);
      for (int a = r0; a < r1; ++a) {
        Lc.rows[Q_RS + (a - r0)] = (uint16_t)(hp.rowptr[a] - b0);
        Lc.rows[Q_RD + (a - r0)] = (uint16_t)(hp.diag[a] - b0);
        for (int q = hp.rowptr[a]; q < hp.rowptr[a + 1]; ++q) {
          const int b = hp.colidx[q];
          if (b == a || (b >= r0 && b < a)) continue;
          uint32_t w = (uint32_t)(q - b0) | 0xFFFF0000u;
          if (b > a && b < r1) {
            const int *cb = hp.colidx.data() + hp.rowptr[b], *ce = hp.colidx.data() + hp.rowptr[b + 1];
            const int m = hp.rowptr[b] + (int)(std::lower_bound(cb, ce, a) - cb) - b0;
            w = (uint32_t)(q - b0) | ((uint32_t)m << 16);
          }
          tid_of[q - b0] = (int)tp.size();
          tp.push_back(w);
        }
      }
      Lc.rows[Q_RS + nrows] = (uint16_t)nb;
      const int ntask = (int)tp.size();
      if (ntask > Q_MAX_TASKS) { bad[p] = 1; continue; }
      lists.assign((size_t)ntask, std::vector<uint16_t>());
      for (int a = r0; a < r1; ++a) {
        const int *cb = hp.colidx.data() + hp.rowptr[a], *ce = hp.colidx.data() + hp.rowptr[a + 1];
        for (int q = hp.incptr[a]; q < hp.incptr[a + 1]; ++q) {
          const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu), la = (int)(hp.inc_rows[q] >> 28);
          const int le = lelem(e);
          for (int lb = 0; lb < npe; ++lb) {
            if (lb == la) continue;
            const int b = conn[(size_t)e * npe + lb];
            if (b == a) continue;                 // repeated node: no off-diagonal block
            const int pos = hp.rowptr[a] + (int)(std::lower_bound(cb, ce, b) - cb) - b0;
            if (tid_of[pos] < 0) continue;        // served by the mirror block's thread
            lists[(size_t)tid_of[pos]].push_back((uint16_t)(le | (la << 7) | (lb << 11)));
          }
        }
      }
      // longest lists first (ties: CSR order), dealt to the waves in runs of 64 so that the 64 blocks a wave works
      // on at a time have lists of (nearly) one length; the runs go over the waves back and forth
      order.resize((size_t)ntask);
      for (int i = 0; i < ntask; ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return lists[x].size() > lists[y].size(); });
      Lc.tpos.assign((size_t)Q_MAX_TASKS, 0xFFFFFFFFu);          // no block: neither position is ever in a pass
      Lc.lists.assign((size_t)Q_MAX_TASKS, std::vector<uint16_t>());
      Gather10Header &h = Lc.h;
      memset(&h, 0, sizeof(h));
      bool too_long = false;
      for (int i = 0; i < ntask; ++i) {
        const int run = i / 64, s = run / FEA_Q_WAVES, wv = (s & 1) ? FEA_Q_WAVES - 1 - (run % FEA_Q_WAVES) : (run % FEA_Q_WAVES);
        const int slot = s * FEA_Q_THREADS + wv * 64 + (i & 63);           // thread wv*64 + i%64, block slot s
        Lc.tpos[slot] = tp[order[i]];
        Lc.lists[slot].swap(lists[order[i]]);
        const int len = (int)Lc.lists[slot].size();
        if (len > 250) too_long = true;
        h.cnt[FEA_Q_WAVES * s + wv] = (unsigned char)std::max((int)h.cnt[FEA_Q_WAVES * s + wv], std::min(len, 250));
        h.sw[s] = (unsigned char)std::max((int)h.sw[s], (std::min(len, 250) + 1) / 2);
      }
      if (too_long) { bad[p] = 1; continue; }
      // write-out passes: whole rows, as many as fit the tile
      {
        int np = 0, a = 0;
        h.prow[0] = 0;
        while (a < nrows) {
          int b = a, blocks = 0;
          while (b < nrows) {
            const int len = hp.rowptr[r0 + b + 1] - hp.rowptr[r0 + b];
            if (blocks + len > tile_blocks) break;
            blocks += len; ++b;
          }
          if (b == a || np >= FEA_Q_MAX_PASS) { np = -1; break; }
          h.prow[++np] = (unsigned char)b;
          a = b;
        }
        if (np < 0) { bad[p] = 1; continue; }
        h.npass = np;
      }
      // residual lanes: slices of 2*fdw visits of one row
      int fdw = 1;
      for (;; ++fdw) {
        int need = 0;
        for (int a = r0; a < r1; ++a) need += std::max(1, (hp.incptr[a + 1] - hp.incptr[a] + 2 * fdw - 1) / (2 * fdw));
        if (need <= FEA_Q_FLANES) break;
        if (2 * fdw >= Q_FENT) { fdw = -1; break; }
      }
      if (fdw < 0) { bad[p] = 1; continue; }
      int nft = 0;
      for (int a = r0; a < r1; ++a) {
        Lc.rows[Q_FF + (a - r0)] = (uint16_t)nft;
        nft += std::max(1, (hp.incptr[a + 1] - hp.incptr[a] + 2 * fdw - 1) / (2 * fdw));
      }
      Lc.rows[Q_FF + nrows] = (uint16_t)nft;
      Lc.flist.assign((size_t)fdw * 2 * FEA_Q_FLANES, (uint16_t)0xFFFFu);   // no visit: the all-zero record, once its slot is known
      for (int a = r0; a < r1; ++a) {
        const int t0 = Lc.rows[Q_FF + (a - r0)];
        int k = 0;
        for (int q = hp.incptr[a]; q < hp.incptr[a + 1]; ++q, ++k) {
          const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu), la = (int)(hp.inc_rows[q] >> 28);
          const int lane = t0 + k / (2 * fdw), j = k % (2 * fdw);
          Lc.flist[((size_t)(j / 2) * FEA_Q_FLANES + lane) * 2 + (j & 1)] = (uint16_t)(lelem(e) | (la << 7));
        }
      }
      Lc.elems.assign((size_t)nelem, 0);            // index into the rank's element list = the state kernel's output order
      for (int i = 0; i < nelem; ++i)
        Lc.elems[i] = (uint32_t)(std::lower_bound(out.elist.begin(), out.elist.end(), el[i]) - out.elist.begin());
      h.r0 = r0; h.r1 = r1; h.b0 = b0; h.nb = nb; h.nnode = 0; h.nelem = nelem; h.ntask = ntask;
      h.nft = nft; h.fdw = fdw;
    }
  });
  for (int p = 0; p < nch; ++p)
    if (bad[p]) return;

  Gather10Layout &lay = out.lay;
  memset(&lay, 0, sizeof(lay));
  for (const Local &Lc : loc) {
    lay.max_elems = std::max(lay.max_elems, Lc.h.nelem);
    int cw = 0;
    for (int s = 0; s < FEA_Q_SLOTS; ++s) cw += Lc.h.sw[s];
    lay.max_cw = std::max(lay.max_cw, cw);
    lay.max_fdw = std::max(lay.max_fdw, Lc.h.fdw);
  }
  lay.max_elems = std::max(lay.max_elems, elem_cap);   // the K tile takes the records' place: sized by the limit the passes were cut for
  lay.max_cw = std::max(lay.max_cw, 1);
  lay.tile_blocks = tile_blocks;
  lay.o_nodes = 0;
  lay.o_elems = (int)sizeof(Gather10Header);
  lay.o_rows = lay.o_elems + up10(4 * (FEA_Q_MAX_ELEMS + 1), 64);
  lay.o_tpos = lay.o_rows + up10(2 * FEA_Q_ROWS_U16, 64);
  lay.o_flist = lay.o_tpos + 4 * Q_MAX_TASKS;
  lay.o_clist = lay.o_flist + up10(4 * lay.max_fdw * FEA_Q_FLANES, 64);
  lay.stride = up10(lay.o_clist + 4 * lay.max_cw * FEA_Q_THREADS, 128);
  if ((long long)nch * lay.stride > 0x7FFFFFFF00LL) return;
  out.blob.assign((size_t)nch * lay.stride, 0);
  par_chunks10(nch, [&](int lo, int hi) {
    for (int p = lo; p < hi; ++p) {
      const Local &Lc = loc[p];
      unsigned char *rec = out.blob.data() + (size_t)p * lay.stride;
      memcpy(rec, &Lc.h, sizeof(Gather10Header));
      memcpy(rec + lay.o_elems, Lc.elems.data(), Lc.elems.size() * 4);
      memcpy(rec + lay.o_rows, Lc.rows.data(), Lc.rows.size() * 2);
      memcpy(rec + lay.o_tpos, Lc.tpos.data(), Lc.tpos.size() * 4);
      uint16_t *fl = reinterpret_cast<uint16_t *>(rec + lay.o_flist);
      const uint16_t zslot = (uint16_t)lay.max_elems;                     // record max_elems of the LDS tile stays all-zero
      for (int i = 0; i < 2 * lay.max_fdw * FEA_Q_FLANES; ++i) fl[i] = zslot;
      for (size_t i = 0; i < Lc.flist.size(); ++i)
        if (Lc.flist[i] != 0xFFFFu) fl[i] = Lc.flist[i];
      // clist: sw[0] rows of 256 words for the threads' first blocks, then sw[1] rows for their second ones, ...
      uint16_t *cl = reinterpret_cast<uint16_t *>(rec + lay.o_clist);
      for (int i = 0; i < 2 * lay.max_cw * FEA_Q_THREADS; ++i) cl[i] = zslot;
      int row0 = 0;
      for (int s = 0; s < FEA_Q_SLOTS; ++s) {
        for (int t = 0; t < FEA_Q_THREADS; ++t) {
          const std::vector<uint16_t> &l = Lc.lists[(size_t)s * FEA_Q_THREADS + t];
          for (size_t k = 0; k < l.size(); ++k) cl[(((size_t)row0 + k / 2) * FEA_Q_THREADS + t) * 2 + (k & 1)] = l[k];
        }
        row0 += Lc.h.sw[s];
      }
    }
  });
  out.nchunks = nch;
  out.total_evals = 0;
  for (const Local &Lc : loc) out.total_evals += Lc.h.nelem;
  {
    long long d = 0;
    for (int a = row_lo; a < row_hi; ++a)
      for (int q = hp.incptr[a]; q < hp.incptr[a + 1]; ++q) {
        const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu), la = (int)(hp.inc_rows[q] >> 28);
        bool first = true;
        for (int k = 0; k < npe; ++k) {
          const int g = conn[(size_t)e * npe + k];
          if (k != la && g >= row_lo && g < a) first = false;
        }
        d += first;
      }
    out.distinct_elems = d;
  }
  out.ok = true;
}

// what the maps say, row by row (gather.cpp: feahip_host_assembly_digest)
void gather10_row_digest(const HostGather10 &hg, const HostPattern &hp, const int *conn, unsigned long long *rowhash,
                         unsigned long long (*hash)(int, int, const int *, int, int))
{
  const Gather10Layout &lay = hg.lay;
  for (int p = 0; p < hg.nchunks; ++p) {
    const unsigned char *rec = hg.blob.data() + (size_t)p * lay.stride;
    const Gather10Header &h = *reinterpret_cast<const Gather10Header *>(rec);
    const uint32_t *elems = reinterpret_cast<const uint32_t *>(rec + lay.o_elems);
    const uint32_t *tpos = reinterpret_cast<const uint32_t *>(rec + lay.o_tpos);
    const uint16_t *cl = reinterpret_cast<const uint16_t *>(rec + lay.o_clist);
    auto row_of = [&](int pos) {
      int a = h.r0;
      while (a + 1 < h.r1 && hp.rowptr[a + 1] - h.b0 <= pos) ++a;
      return a;
    };
    int row0 = 0;
    for (int s = 0; s < FEA_Q_SLOTS; ++s) {
      for (int t = 0; t < FEA_Q_THREADS; ++t) {
        const uint32_t tw = tpos[s * FEA_Q_THREADS + t];
        if (tw == 0xFFFFFFFFu) continue;
        const int bpos = (int)(tw & 0xFFFFu), mpos = (int)(tw >> 16);
        const int a = row_of(bpos), b = hp.colidx[h.b0 + bpos];
        for (int k = 0; k < 2 * h.sw[s]; ++k) {
          const uint16_t w = cl[(((size_t)row0 + k / 2) * FEA_Q_THREADS + t) * 2 + (k & 1)];
          const int le = w & 127, la = (w >> 7) & 15, lb = (w >> 11) & 15;
          if (le == lay.max_elems) continue;
          int g[16];
          for (int j = 0; j < hg.npe; ++j) g[j] = conn[(size_t)hg.elist[elems[le]] * hg.npe + j];
          rowhash[a] += hash(a, b, g, la, lb);
          if (mpos != 0xFFFF) rowhash[b] += hash(b, a, g, lb, la);
        }
      }
      row0 += h.sw[s];
    }
  }
}
