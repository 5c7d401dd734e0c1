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
// the 16-lane group a ds_read_b128 serves thread t in (MI355X_MICROARCH: {0-3,12-15,20-27}, {4-11,16-19,28-31}, and
// the same pattern in the upper half of the wave), numbered over the whole workgroup
inline int b128_group(int t)
{
  static const unsigned char g32[32] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1};
  return (t >> 6) * 4 + ((t >> 5) & 1) * 2 + g32[t & 31];
}
}  // namespace

// u16 offsets inside the "rows" section
#define G_RS 0                        // rstart[MAX_ROWS + 1]
#define G_RD 66                       // rdiag[MAX_ROWS]
#define G_VF 130                      // vfirst[MAX_ROWS + 1]
#define G_ROWS_U16 200
#define G_TASK_THREADS FEA_G_TASK_THREADS   // block and residual threads; the remaining waves sum the diagonal blocks
#define G_SLOT(w) ((w) & 1023u)
#define G_LA(w) (((w) >> 10) & 3)
#define G_LB(w) (((w) >> 12) & 3)
#define G_NGROUPS (FEA_G_THREADS / 16) // 16-lane groups of a workgroup's ds_read_b128

void build_host_gather(int N, int E, const int *conn, const HostPattern &hp, int row_lo, int row_hi, HostGather &out)
{
  (void)E;
  out.ok = false; out.nchunks = 0; out.blob.clear(); out.first_row.clear();
  if (row_lo < 0 || row_hi > N || row_lo >= row_hi) return;
  // limits of one chunk: its rows, and the element records one CU's LDS holds (a lattice's bricks stop at the row
  // limit with FEA_G_ELEMS_TARGET elements whatever the element limit is -- its partition is the same for 672 and 719,
  // checked on the 31^3, 40^3 and 66^3 blocks; an unstructured mesh uses the room: 2-3 % fewer chunks)
  int max_rows = FEA_G_MAX_ROWS, max_elems = FEA_G_BIG == 1 ? FEA_G_MAX_ELEMS : FEA_G_ELEMS_TARGET, alpha = 24;
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
          {                                        // block threads: blocks whose column is a lower row of the window are mirrors
            const int *cb = hp.colidx.data() + hp.rowptr[r], *ce = hp.colidx.data() + hp.rowptr[r + 1];
            noffd += (int)(ce - cb) - 1 - (int)(std::lower_bound(cb, ce, r) - std::lower_bound(cb, ce, r0));
          }
          nvis += hp.incptr[r + 1] - hp.incptr[r];
          const bool fits = nel <= (l > 1 ? max_elems : FEA_G_MAX_ELEMS) && nnod <= FEA_G_MAX_NODES &&
                            noffd <= G_TASK_THREADS && nvis <= 4 * G_TASK_THREADS;
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
    std::vector<uint16_t> rows, vlist, clist, dlist;
  };
  std::vector<Local> loc((size_t)nch);
  std::vector<char> bad((size_t)nch, 0);
  par_chunks(nch, [&](int lo, int hi) {
    std::vector<int> el, nd, tid_of, eslot, nslot, order;
    std::vector<std::vector<uint16_t>> lists, dl;
    struct Read { uint16_t set; uint8_t off; };          // one LDS read of an element's record: (lane group, step, kind) and piece
    std::vector<std::vector<Read>> reads;
    std::vector<uint8_t> occ;
    for (int p = lo; p < hi; ++p) {
      Local &L = loc[p];
      const int r0 = out.first_row[p], r1 = out.first_row[p + 1], nrows = r1 - r0;
      const int b0 = hp.rowptr[r0], nb = hp.rowptr[r1] - b0;
      el.clear(); nd.clear();
      for (int q = hp.incptr[r0]; q < hp.incptr[r1]; ++q) el.push_back((int)(hp.inc_rows[q] & 0x0FFFFFFFu));
      std::sort(el.begin(), el.end());
      el.erase(std::unique(el.begin(), el.end()), el.end());
      for (int a = r0; a < r1; ++a) nd.push_back(a);          // a node no element refers to still owns a (diagonal) row
      for (int e : el)
        for (int k = 0; k < 4; ++k) nd.push_back(conn[(size_t)e * 4 + k]);
      std::sort(nd.begin(), nd.end());
      nd.erase(std::unique(nd.begin(), nd.end()), nd.end());
      const int nnode = (int)nd.size(), nelem = (int)el.size();
      const int nslots = 16 * ((nelem + 1 + 15) / 16), nnslots = 16 * ((nnode + 15) / 16);
      if (nnslots > FEA_G_MAX_NODES || nslots > FEA_G_MAX_SLOTS || nrows > FEA_G_MAX_ROWS) { bad[p] = 1; continue; }
      auto lnode = [&](int g) { return (int)(std::lower_bound(nd.begin(), nd.end(), g) - nd.begin()); };
      auto lelem = [&](int e) { return (int)(std::lower_bound(el.begin(), el.end(), e) - el.begin()); };
      // blocks with a thread: the off-diagonal blocks in CSR order; a block whose column is a LOWER row of the same chunk
      // has no thread of its own, it is the transpose of its mirror block.  blk_of[pos] = index of the block at tile
      // position pos; which THREAD serves it is decided below, once the lists are known.
      tid_of.assign((size_t)nb, -1);
      std::vector<uint32_t> btpos;
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
          tid_of[q - b0] = (int)btpos.size();
          btpos.push_back(w);
        }
      }
      L.rows[G_RS + nrows] = (uint16_t)nb;
      const int nblk = (int)btpos.size();
      if (nblk > G_TASK_THREADS) { bad[p] = 1; continue; }
      // contributions per block, and per row the visits of its diagonal block (four lanes of the last waves
      // per row: a row's diagonal block is summed from the records like any other block, K_aa = sum_e K_aa^e)
      std::vector<std::vector<uint16_t>> blists((size_t)nblk);
      dl.assign((size_t)4 * FEA_G_MAX_ROWS, std::vector<uint16_t>());
      for (int a = r0; a < r1; ++a) {
        const int *cb = hp.colidx.data() + hp.rowptr[a], *ce = hp.colidx.data() + hp.rowptr[a + 1];
        int kv = 0;
        for (int q = hp.incptr[a]; q < hp.incptr[a + 1]; ++q, ++kv) {
          const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu), la = (int)(hp.inc_rows[q] >> 28);
          const int le = lelem(e);
          dl[(size_t)4 * (a - r0) + (kv & 3)].push_back((uint16_t)(le | (la << 10)));
          for (int lb = 0; lb < 4; ++lb) {
            if (lb == la) continue;
            const int b = conn[(size_t)e * 4 + lb];
            if (b == a) continue;                 // degenerate element (repeated node): no off-diagonal block
            const int pos = hp.rowptr[a] + (int)(std::lower_bound(cb, ce, b) - cb) - b0;
            if (tid_of[pos] < 0) continue;        // served by the mirror block's thread
            blists[(size_t)tid_of[pos]].push_back((uint16_t)(le | (la << 10) | (lb << 12)));
          }
        }
      }
      // ---- which thread serves which block.  A wave walks its lists to the depth of its LONGEST one (empty entries
      // read the all-zero record), and the gather phase lasts as long as its busiest SIMD: in CSR order every wave
      // of a Kuhn block mixes blocks of 4 and of 6 contributions and walks 6, and ten block waves over four SIMDs
      // are 3 + 3 + 2 + 2.  So (i) the blocks are sorted by list length, a wave holds lists of (nearly) one length and
      // stops at its own depth (GatherHeader::wdepth); (ii) the waves are dealt to the wave slots so that the four
      // SIMDs carry equal sums of depths -- a workgroup's waves go to the SIMDs cyclically, slot w runs on SIMD
      // (w + start) mod 4, and the slots s, s+4, s+8 of the block waves share one (longest wave first, to the
      // lightest SIMD with a free slot); (iii) inside a wave sixteen consecutive lanes (one group of the tile
      // phase's ds_write_b64) get tile positions that differ mod 16, mirrors too where possible: a block is nine
      // doubles, so two blocks meet in a bank exactly when their positions agree mod 16.
      std::vector<int> thr_blk((size_t)G_TASK_THREADS, -1);      // thread -> block
      int wdepth[G_TASK_THREADS / 64] = {0};
      int ntask = 0;
      {
        constexpr int NBW = G_TASK_THREADS / 64;
        auto words = [&](int i) { return ((int)blists[i].size() + 1) / 2; };
        std::vector<int> ord((size_t)nblk);
        for (int i = 0; i < nblk; ++i) ord[i] = i;
        std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return words(x) > words(y); });
        // sixteens with distinct tile positions mod 16, class by class (a class = one list length in words)
        std::vector<int> seq;
        seq.reserve((size_t)nblk);
        for (size_t c0 = 0; c0 < ord.size();) {
          size_t c1 = c0;
          while (c1 < ord.size() && words(ord[c1]) == words(ord[c0])) ++c1;
          std::vector<int> bucket[16];
          for (size_t k = c0; k < c1; ++k) bucket[btpos[ord[k]] & 15u].push_back(ord[k]);
          size_t left = c1 - c0;
          while (left) {
            // finish the sixteen the previous class may have left open, then whole sixteens
            const int room = 16 - (int)(seq.size() & 15);
            int rs[16];
            for (int r = 0; r < 16; ++r) rs[r] = r;
            std::stable_sort(rs, rs + 16, [&](int x, int y) { return bucket[x].size() > bucket[y].size(); });
            unsigned mused = 0;
            int taken = 0;
            for (int pass = 0; pass < 2 && taken < room && left; ++pass)      // pass 1: a second block of a residue, if the sixteen would stay short
              for (int q = 0; q < 16 && taken < room && left; ++q) {
                std::vector<int> &bk = bucket[rs[q]];
                if (bk.empty()) continue;
                size_t pick = 0;                                              // prefer a mirror position not yet in the sixteen
                for (size_t c = 0; c < bk.size() && c < 8; ++c) {
                  const unsigned mp = btpos[bk[c]] >> 16;
                  if (mp == 0xFFFFu || !((mused >> (mp & 15u)) & 1u)) { pick = c; break; }
                }
                const unsigned mp = btpos[bk[pick]] >> 16;
                if (mp != 0xFFFFu) mused |= 1u << (mp & 15u);
                seq.push_back(bk[pick]);
                bk.erase(bk.begin() + (long)pick);
                ++taken; --left;
              }
          }
          c0 = c1;
        }
        const int nwaves = (nblk + 63) / 64;
        std::vector<int> wcost((size_t)nwaves, 0), word((size_t)nwaves);
        for (int w = 0; w < nwaves; ++w) {
          for (int l = 0; l < 64 && w * 64 + l < nblk; ++l) wcost[w] = std::max(wcost[w], words(seq[(size_t)w * 64 + l]));
          word[w] = w;
        }
        std::stable_sort(word.begin(), word.end(), [&](int x, int y) { return wcost[x] > wcost[y]; });
        int load[4] = {0, 0, 0, 0}, used[4] = {0, 0, 0, 0};
        for (int w : word) {
          int bs = -1;
          for (int sd = 0; sd < 4; ++sd)
            if (used[sd] * 4 + sd < NBW && (bs < 0 || load[sd] < load[bs])) bs = sd;
          const int slot = used[bs] * 4 + bs;
          ++used[bs]; load[bs] += wcost[w];
          wdepth[slot] = wcost[w];
          for (int l = 0; l < 64 && w * 64 + l < nblk; ++l) thr_blk[(size_t)slot * 64 + l] = seq[(size_t)w * 64 + l];
          ntask = std::max(ntask, slot * 64 + std::min(64, nblk - w * 64));
        }
      }
      L.tpos.assign((size_t)ntask, 0xFFFFFFFFu);                  // a thread without a block: no tile position, empty lists
      lists.assign((size_t)ntask, std::vector<uint16_t>());
      for (int t = 0; t < ntask; ++t)
        if (thr_blk[t] >= 0) { L.tpos[t] = btpos[thr_blk[t]]; lists[t] = blists[thr_blk[t]]; }
      int depth = 0, ddepth = 0;
      for (auto &l : lists) depth = std::max(depth, (int)l.size());
      for (auto &l : dl) ddepth = std::max(ddepth, (int)l.size());
      const int dwords = (depth + 1) / 2, ddwords = (ddepth + 1) / 2;

      // ---- LDS bank schedule.  A record is 13 pieces of 16 bytes (kernels_gather.hip) and every read of it is a
      // ds_read_b128, served in groups of 16 lanes over 16 bank slots: lanes of a group that read different
      // addresses in one slot serialise.  The slot of piece `off` of the record in element slot s is (13 s + off)
      // mod 16 = (off - 3 s) mod 16, so which slots collide is decided by s mod 16 alone: choose that residue per
      // element greedily against the reads already placed (most-read elements first), then once more with
      // everything in place.  PMC before: 41 % of the kernel's LDS cycles were bank conflicts.
      reads.assign((size_t)nelem, std::vector<Read>());
      const int nsteps = std::max(2 * dwords, 2 * ddwords);
      auto add_reads = [&](int lane, int step, uint16_t w, bool diag) {
        const int le = (int)G_SLOT(w), la = G_LA(w), lb = G_LB(w);
        const int grp = b128_group(lane);
        // kinds: 0 P_a, 1 Z_a, 2 P_b, 3 Q_b, 4 Z_b, 5 VV   (diagonal visit: P_a, Z_a, Q_a, VV)
        const int offs[6] = {la, 8 + la, diag ? 4 + la : lb, diag ? -1 : 4 + lb, diag ? -1 : 8 + lb, 12};
        for (int kind = 0; kind < 6; ++kind)
          if (offs[kind] >= 0) reads[le].push_back({(uint16_t)((grp * nsteps + step) * 6 + kind), (uint8_t)offs[kind]});
      };
      for (int t = 0; t < ntask; ++t)
        for (size_t k = 0; k < lists[t].size(); ++k) add_reads(t, (int)k, lists[t][k], false);
      for (int l = 0; l < 4 * nrows; ++l)
        for (size_t k = 0; k < dl[l].size(); ++k) add_reads(G_TASK_THREADS + l, (int)k, dl[l][k], true);
      const int nsets = G_NGROUPS * nsteps * 6;
      occ.assign((size_t)nsets * 16, 0);
      eslot.assign((size_t)nelem, -1);
      std::vector<int> cap(16, nslots / 16), res((size_t)nelem, -1);
      order.resize((size_t)nelem);
      for (int i = 0; i < nelem; ++i) order[i] = i;
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return reads[a].size() > reads[b].size(); });
      // an element read twice in one set at one piece is one address (a broadcast): count it once
      for (auto &rv : reads) {
        std::sort(rv.begin(), rv.end(), [](const Read &a, const Read &b) { return a.set != b.set ? a.set < b.set : a.off < b.off; });
        rv.erase(std::unique(rv.begin(), rv.end(), [](const Read &a, const Read &b) { return a.set == b.set && a.off == b.off; }), rv.end());
      }
      auto place = [&](int e, int r, int sign) {
        for (const Read &rd : reads[e]) occ[(size_t)rd.set * 16 + ((rd.off - 3 * r) & 15)] += sign;
      };
      auto best_residue = [&](int e) {
        int br = -1; long bc = 0;
        for (int r = 0; r < 16; ++r) {
          if (cap[r] <= 0) continue;
          long c = 0;
          for (const Read &rd : reads[e]) c += occ[(size_t)rd.set * 16 + ((rd.off - 3 * r) & 15)];
          if (br < 0 || c < bc) { br = r; bc = c; }
        }
        return br;
      };
      --cap[15];                                    // one slot stays empty: the all-zero record of the unused list slots
      for (int pass = 0; pass < 2; ++pass)
        for (int e : order) {
          if (res[e] >= 0) { place(e, res[e], -1); ++cap[res[e]]; }
          const int r = best_residue(e);
          res[e] = r; --cap[r]; place(e, r, +1);
        }
      ++cap[15];
      {
        std::vector<int> next(16);
        for (int r = 0; r < 16; ++r) next[r] = r;
        for (int e = 0; e < nelem; ++e) { eslot[e] = next[res[e]]; next[res[e]] += 16; }
      }
      std::vector<char> used((size_t)nslots, 0);
      for (int e = 0; e < nelem; ++e) used[eslot[e]] = 1;
      int zslot = 0;
      while (used[zslot]) ++zslot;

      // node slots: the state phase reads the four nodes of the element in slot t from lane t, again 16 lanes per
      // group over 16 slots of 16 bytes; same greedy
      nslot.assign((size_t)nnode, -1);
      {
        std::vector<std::vector<uint16_t>> nreads((size_t)nnode);
        for (int e = 0; e < nelem; ++e)
          for (int k = 0; k < 4; ++k) nreads[lnode(conn[(size_t)el[e] * 4 + k])].push_back((uint16_t)(b128_group(eslot[e]) * 4 + k));
        std::vector<uint8_t> nocc((size_t)G_NGROUPS * 4 * 16, 0);    // lane groups x 4 node positions x 16 slots
        std::vector<int> ncap(16, nnslots / 16), nres((size_t)nnode, -1), nord((size_t)nnode);
        for (int i = 0; i < nnode; ++i) { nord[i] = i; std::sort(nreads[i].begin(), nreads[i].end()); nreads[i].erase(std::unique(nreads[i].begin(), nreads[i].end()), nreads[i].end()); }
        std::stable_sort(nord.begin(), nord.end(), [&](int a, int b) { return nreads[a].size() > nreads[b].size(); });
        for (int pass = 0; pass < 2; ++pass)
          for (int i : nord) {
            if (nres[i] >= 0) { for (uint16_t st : nreads[i]) --nocc[(size_t)st * 16 + nres[i]]; ++ncap[nres[i]]; }
            int br = -1; long bc = 0;
            for (int r = 0; r < 16; ++r) {
              if (ncap[r] <= 0) continue;
              long c = 0;
              for (uint16_t st : nreads[i]) c += nocc[(size_t)st * 16 + r];
              if (br < 0 || c < bc) { br = r; bc = c; }
            }
            nres[i] = br; --ncap[br];
            for (uint16_t st : nreads[i]) ++nocc[(size_t)st * 16 + br];
          }
        std::vector<int> next(16);
        for (int r = 0; r < 16; ++r) next[r] = r;
        for (int i = 0; i < nnode; ++i) { nslot[i] = next[nres[i]]; next[nres[i]] += 16; }
      }

      // ---- emit with the slots
      L.nodes.assign((size_t)nnslots, 0);           // unused slots: node 0 (their coordinates are loaded and never read)
      for (int i = 0; i < nnode; ++i) L.nodes[nslot[i]] = nd[i];
      L.elems.assign((size_t)nslots, 0xFFFFFFFFu);  // unused slots stay all-zero records
      for (int i = 0; i < nelem; ++i) {
        uint32_t w = 0;
        for (int k = 0; k < 4; ++k) w |= (uint32_t)nslot[lnode(conn[(size_t)el[i] * 4 + k])] << (8 * k);
        L.elems[eslot[i]] = w;
      }
      auto reslot = [&](uint16_t w) { return (uint16_t)((w & 0xFC00u) | (uint16_t)eslot[G_SLOT(w)]); };
      L.clist.assign((size_t)dwords * 2 * FEA_G_THREADS, (uint16_t)zslot);
      for (int t = 0; t < ntask; ++t)
        for (size_t k = 0; k < lists[t].size(); ++k)
          L.clist[((size_t)(k / 2) * FEA_G_THREADS + t) * 2 + (k & 1)] = reslot(lists[t][k]);
      L.dlist.assign((size_t)ddwords * 2 * FEA_G_DIAG_LANES, (uint16_t)zslot);
      for (int l = 0; l < 4 * nrows; ++l)
        for (size_t k = 0; k < dl[l].size(); ++k)
          L.dlist[((size_t)(k / 2) * FEA_G_DIAG_LANES + l) * 2 + (k & 1)] = reslot(dl[l][k]);
      // residual threads (waves 0-2): slices of vdepth visits of one row
      int vdepth = 1;
      for (;; ++vdepth) {
        int need = 0;
        for (int a = r0; a < r1; ++a) need += (hp.incptr[a + 1] - hp.incptr[a] + vdepth - 1) / vdepth;
        if (need <= G_TASK_THREADS) break;
      }
      int nvthr = 0;
      for (int a = r0; a < r1; ++a) {
        L.rows[G_VF + (a - r0)] = (uint16_t)nvthr;
        nvthr += (hp.incptr[a + 1] - hp.incptr[a] + vdepth - 1) / vdepth;
      }
      L.rows[G_VF + nrows] = (uint16_t)nvthr;
      L.vlist.assign((size_t)vdepth * FEA_G_THREADS, (uint16_t)zslot);
      for (int a = r0; a < r1; ++a) {
        const int t0 = L.rows[G_VF + (a - r0)];
        int k = 0;
        for (int q = hp.incptr[a]; q < hp.incptr[a + 1]; ++q, ++k) {
          const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu), la = (int)(hp.inc_rows[q] >> 28);
          L.vlist[(size_t)(k % vdepth) * FEA_G_THREADS + t0 + k / vdepth] = (uint16_t)(eslot[lelem(e)] | (la << 10));
        }
      }
      GatherHeader &h = L.h;
      memset(&h, 0, sizeof(h));
      h.r0 = r0; h.r1 = r1; h.b0 = b0; h.nb = nb; h.nnode = nnslots; h.nelem = nslots; h.noffd = ntask;
      h.depth = dwords; h.nvthr = nvthr; h.vdepth = vdepth; h.ddepth = ddwords;
      for (int w = 0; w < G_TASK_THREADS / 64; ++w) h.wdepth[w >> 2] |= (unsigned)std::min(wdepth[w], 255) << (8 * (w & 3));
    }
  });
  for (int p = 0; p < nch; ++p)
    if (bad[p]) return;

  // ---- layout: fixed section offsets, sized by the largest chunk
  int m_v = 0, m_c = 0, m_d = 0, g_nodes = 0, g_elems = 0, g_tile = 0;
  GatherLayout &lay = out.lay;
  memset(&lay, 0, sizeof(lay));
  for (const Local &L : loc) {
    m_v = std::max(m_v, (int)L.vlist.size() * 2); m_c = std::max(m_c, (int)L.clist.size() * 2); m_d = std::max(m_d, (int)L.dlist.size() * 2);
    g_nodes = std::max(g_nodes, L.h.nnode); g_elems = std::max(g_elems, L.h.nelem); g_tile = std::max(g_tile, L.h.nb);
    lay.max_tasks = std::max(lay.max_tasks, L.h.noffd); lay.max_depth = std::max(lay.max_depth, L.h.depth);
    lay.max_vthr = std::max(lay.max_vthr, L.h.nvthr); lay.max_vdepth = std::max(lay.max_vdepth, L.h.vdepth);
    lay.max_ddepth = std::max(lay.max_ddepth, L.h.ddepth);
  }
  lay.max_nodes = g_nodes; lay.max_elems = g_elems; lay.max_tile = g_tile;
  lay.o_nodes = 64;
  lay.o_elems = lay.o_nodes + up(4 * FEA_G_MAX_NODES, 64);
  lay.o_bpos = lay.o_elems + up(4 * FEA_G_THREADS, 64);
  lay.o_rows = lay.o_bpos + up(4 * FEA_G_THREADS, 64);
  lay.o_vlist = lay.o_rows + up(2 * G_ROWS_U16, 64);
  lay.o_dlist = lay.o_vlist + up(m_v, 64);
  lay.o_clist = lay.o_dlist + up(m_d, 64);
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
      memcpy(rec + lay.o_dlist, L.dlist.data(), L.dlist.size() * 2);
      memcpy(rec + lay.o_clist, L.clist.data(), L.clist.size() * 2);
    }
  });
  // a chunk whose successor has the same map words (chunk-local indices only: the interior bricks of a structured
  // block are all alike) says so in its header: the kernel then keeps the words in registers instead of loading them
  out.same_as_previous = 0;
  for (int p = 0; p + 1 < nch; ++p) {
    const unsigned char *r0 = out.blob.data() + (size_t)p * lay.stride, *r1 = r0 + lay.stride;
    const GatherHeader &h0 = *reinterpret_cast<const GatherHeader *>(r0), &h1 = *reinterpret_cast<const GatherHeader *>(r1);
    const bool same = h0.nelem == h1.nelem && h0.noffd == h1.noffd && h0.depth == h1.depth && h0.nvthr == h1.nvthr &&
                      h0.vdepth == h1.vdepth && h0.ddepth == h1.ddepth && h0.r1 - h0.r0 == h1.r1 - h1.r0 &&
                      memcmp(r0 + lay.o_elems, r1 + lay.o_elems, (size_t)(lay.stride - lay.o_elems)) == 0;
    if (same) { reinterpret_cast<GatherHeader *>(out.blob.data() + (size_t)p * lay.stride)->flags |= 1; ++out.same_as_previous; }
  }
  out.nchunks = nch;
  out.total_evals = 0;
  for (const Local &L : loc) {
    const uint32_t *ev = L.elems.data();
    for (size_t i = 0; i < L.elems.size(); ++i) out.total_evals += ev[i] != 0xFFFFFFFFu;
  }
  {                                                  // distinct elements touching rows [row_lo, row_hi)
    long long d = 0;
    for (int a = row_lo; a < row_hi; ++a)
      for (int q = hp.incptr[a]; q < hp.incptr[a + 1]; ++q) {
        const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu), la = (int)(hp.inc_rows[q] >> 28);
        bool first = true;                           // counted at its lowest-numbered node inside the range
        for (int k = 0; k < 4; ++k) {
          const int g = conn[(size_t)e * 4 + k];
          if (k != la && g >= row_lo && g < a) first = false;
        }
        d += first;
      }
    out.distinct_elems = d;
  }
  out.ok = true;
}

// ---------------------------------------------------------------------------
// Host-only digest of what the assembly maps of one rank say, row by row: for every block row a rank's maps cover,
// a hash of the set of (row, column, element nodes, local row node, local column node) contributions they list
// (mirror blocks expanded).  The maps of different shards are cut differently (the gather chunks of a rank start
// at its first row), but what they SAY about a row must not depend on the cut: tests/test_host.py checks that the
// digests of the ranks of a sharded run add up to the digest of the unsharded run.  No device is touched.
// ---------------------------------------------------------------------------
namespace {
inline unsigned long long mix(unsigned long long h, unsigned long long v)
{
  h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
  return h * 0xBF58476D1CE4E5B9ull;
}
inline unsigned long long contribution_hash(int a, int b, const int g[4], int la, int lb)
{
  unsigned long long h = 0x1234567ull;
  h = mix(h, (unsigned long long)a); h = mix(h, (unsigned long long)b);
  for (int k = 0; k < 4; ++k) h = mix(h, (unsigned long long)g[k]);
  h = mix(h, (unsigned long long)(la * 4 + lb));
  return h;
}
}  // namespace

void gather_row_digest(const HostGather &hg, const HostPattern &hp, unsigned long long *rowhash)
{
  const GatherLayout &lay = hg.lay;
  for (int p = 0; p < hg.nchunks; ++p) {
    const unsigned char *rec = hg.blob.data() + (size_t)p * lay.stride;
    const GatherHeader &h = *reinterpret_cast<const GatherHeader *>(rec);
    const int *nodes = reinterpret_cast<const int *>(rec + lay.o_nodes);
    const uint32_t *elems = reinterpret_cast<const uint32_t *>(rec + lay.o_elems);
    const uint32_t *tpos = reinterpret_cast<const uint32_t *>(rec + lay.o_bpos);
    const uint16_t *clist = reinterpret_cast<const uint16_t *>(rec + lay.o_clist);
    const uint16_t *dlist = reinterpret_cast<const uint16_t *>(rec + lay.o_dlist);
    auto element_nodes = [&](int slot, int g[4]) {
      for (int k = 0; k < 4; ++k) g[k] = nodes[(elems[slot] >> (8 * k)) & 255u];
    };
    auto row_of = [&](int pos) {                      // tile position -> global row
      int a = h.r0;
      while (a + 1 < h.r1 && hp.rowptr[a + 1] - h.b0 <= pos) ++a;
      return a;
    };
    for (int t = 0; t < h.noffd; ++t) {
      if (tpos[t] == 0xFFFFFFFFu) continue;             // a thread without a block
      const int bpos = (int)(tpos[t] & 0xFFFFu), mpos = (int)(tpos[t] >> 16);
      const int a = row_of(bpos), b = hp.colidx[h.b0 + bpos];
      for (int k = 0; k < 2 * h.depth; ++k) {
        const uint16_t w = clist[((size_t)(k / 2) * FEA_G_THREADS + t) * 2 + (k & 1)];
        const int slot = (int)G_SLOT(w), la = G_LA(w), lb = G_LB(w);
        if (elems[slot] == 0xFFFFFFFFu) continue;     // empty list slot
        int g[4];
        element_nodes(slot, g);
        rowhash[a] += contribution_hash(a, b, g, la, lb);
        if (mpos != 0xFFFF) rowhash[b] += contribution_hash(b, a, g, lb, la);
      }
    }
    for (int l = 0; l < 4 * (h.r1 - h.r0); ++l)
      for (int k = 0; k < 2 * h.ddepth; ++k) {
        const uint16_t w = dlist[((size_t)(k / 2) * FEA_G_DIAG_LANES + l) * 2 + (k & 1)];
        const int slot = (int)G_SLOT(w), la = G_LA(w);
        if (elems[slot] == 0xFFFFFFFFu) continue;
        int g[4];
        element_nodes(slot, g);
        const int a = h.r0 + (l >> 2);
        rowhash[a] += contribution_hash(a, a, g, la, la);
      }
  }
}

void gather10_row_digest(const HostGather10 &hg, const HostPattern &hp, const int *conn, unsigned long long *rowhash,
                         unsigned long long (*hash)(int, int, const int *, int, int));     // gather10.cpp
namespace {
int hash_npe = 10;           // nodes per element of the mesh being digested (host-only, single-threaded)
unsigned long long contribution_hash10(int a, int b, const int *g, int la, int lb)
{
  unsigned long long h = 0x1234567ull;
  h = mix(h, (unsigned long long)a); h = mix(h, (unsigned long long)b);
  for (int k = 0; k < hash_npe; ++k) h = mix(h, (unsigned long long)g[k]);
  return mix(h, (unsigned long long)(la * 16 + lb));
}
}  // namespace

void quad_row_digest(const HostQuad &hq, const HostPattern &hp, unsigned long long *rowhash)
{
  for (const QuadDesc &d : hq.desc) {
    for (int p = 0; p < d.npair; ++p) {
      const uint32_t w = hq.qpair[(size_t)d.pair_off + p];
      const int eli = (int)(w & 63u), la = (int)((w >> 6) & 15u), lb = (int)((w >> 10) & 15u), pos = (int)((w >> 14) & 255u);
      const int a = d.r0 + (int)((w >> 22) & 15u), b = hp.colidx[d.b0 + pos];
      const uint32_t *we = hq.qelem.data() + ((size_t)d.elem_off + eli) * 3;
      unsigned long long h = 0x1234567ull;
      h = mix(h, (unsigned long long)a); h = mix(h, (unsigned long long)b);
      for (int k = 0; k < 10; ++k) h = mix(h, (unsigned long long)hq.qnode[(size_t)d.node_off + ((we[k / 4] >> (8 * (k % 4))) & 255u)]);
      h = mix(h, (unsigned long long)(la * 16 + lb));
      rowhash[a] += h;
    }
  }
}

extern "C" int feahip_host_assembly_digest(int n_nodes, int n_elems, int npe, const int *elements, int rank, int nranks,
                                           unsigned long long *rowhash, int *rows)
{
  if (!elements || !rowhash || !rows || n_nodes <= 0 || n_elems <= 0 || nranks < 1 || rank < 0 || rank >= nranks) return FEAHIP_EINVAL;
  HostPattern hp;
  std::string err;
  int rc = build_host_pattern(n_nodes, n_elems, npe, elements, hp, err);
  if (rc) return rc;
  int row0, row1;
  shard_row_range(hp.chunk, rank, nranks, row0, row1);
  rows[0] = row0; rows[1] = row1;
  for (int a = 0; a < n_nodes; ++a) rowhash[a] = 0;
  if (npe == 4) {
    HostGather hg;
    build_host_gather(n_nodes, n_elems, elements, hp, row0, row1, hg);
    if (!hg.ok) return FEAHIP_EINVAL;
    gather_row_digest(hg, hp, rowhash);
  } else {
    // what AUTO runs for 10-node elements (kernels_assemble.hip): the gather maps when they build and an element
    // does not fall into too many chunks (12), the shared-state maps otherwise.  Both list every off-diagonal
    // contribution (the gather maps' mirror blocks are expanded); neither digest covers the diagonal blocks.
    bool done = false;
    if (npe == 10 || npe == 8) {
      HostGather10 hg;
      hash_npe = npe;
      build_host_gather10(n_nodes, n_elems, npe, elements, hp, row0, row1, hg);
      if (hg.ok && (double)hg.total_evals <= 12.0 * (double)hg.distinct_elems) {
        gather10_row_digest(hg, hp, elements, rowhash, contribution_hash10);
        done = true;
      }
    }
    if (!done) {
      if (hp.super_achunk.empty()) return FEAHIP_EINVAL;
      const int nchunks = (int)hp.chunk.size() - 1;
      const int nsuper = (nchunks + FEA_SUPER_CHUNKS - 1) / FEA_SUPER_CHUNKS;
      const int s0 = (int)((long long)nsuper * rank / nranks), s1 = (int)((long long)nsuper * (rank + 1) / nranks);
      HostQuad hq;
      build_host_quad(n_nodes, n_elems, npe, elements, hp, hp.super_achunk[s0], hp.super_achunk[s1], hq);
      if (!hq.ok) return FEAHIP_EINVAL;
      quad_row_digest(hq, hp, rowhash);
    }
  }
  return FEAHIP_OK;
}

// Host-only (no device): what the gather maps (4-node, or 10-node / 8-node) look like for a mesh in the numbering it is given --
// stats[0..7] = chunks, element evaluations, distinct elements, rows, chunks repeating their predecessor's words, map
// bytes, chunks with block lists / diagonal lists longer than a thread's registers hold (4-node only); rows_hist[FEA_G_MAX_ROWS + 1] (may be null) = chunks by row count.  What the numbering of an unstructured mesh
// is judged by before a device sees it (tools/gather_stats.py).
extern "C" int feahip_host_gather_stats(int n_nodes, int n_elems, int npe, const int *elements, long long *stats, int *rows_hist)
{
  if (!elements || !stats || n_nodes <= 0 || n_elems <= 0 || (npe != 4 && npe != 10 && npe != 8)) return FEAHIP_EINVAL;
  HostPattern hp;
  std::string err;
  int rc = build_host_pattern(n_nodes, n_elems, npe, elements, hp, err);
  if (rc) return rc;
  const std::vector<int> *first_row;
  stats[6] = stats[7] = 0;
  HostGather hg;
  HostGather10 hq;
  if (npe == 4) {
    build_host_gather(n_nodes, n_elems, elements, hp, 0, n_nodes, hg);
    if (!hg.ok) return FEAHIP_EINVAL;
    stats[0] = hg.nchunks; stats[1] = hg.total_evals; stats[2] = hg.distinct_elems;
    stats[4] = hg.same_as_previous; stats[5] = (long long)hg.blob.size();
    for (int p = 0; p < hg.nchunks; ++p) {            // lists longer than the words a thread keeps in registers are walked out of memory
      const GatherHeader &gh = *reinterpret_cast<const GatherHeader *>(hg.blob.data() + (size_t)p * hg.lay.stride);
      stats[6] += gh.depth > FEA_G_REGW; stats[7] += gh.ddepth > FEA_G_REGW;
    }
    if (getenv("FEAHIP_NUMBERING_VERBOSE")) {         // chunks by the words of their longest block list / diagonal list
      int hd[17] = {0}, hdd[17] = {0};
      for (int p = 0; p < hg.nchunks; ++p) {
        const GatherHeader &gh = *reinterpret_cast<const GatherHeader *>(hg.blob.data() + (size_t)p * hg.lay.stride);
        ++hd[std::min(gh.depth, 16)]; ++hdd[std::min(gh.ddepth, 16)];
      }
      for (int k = 0; k <= 16; ++k) if (hd[k] || hdd[k]) fprintf(stderr, "gather maps: %d words: %d chunks by block list, %d by diagonal list\n", k, hd[k], hdd[k]);
    }
    first_row = &hg.first_row;
  } else {
    build_host_gather10(n_nodes, n_elems, npe, elements, hp, 0, n_nodes, hq);
    if (!hq.ok) return FEAHIP_EINVAL;
    stats[0] = hq.nchunks; stats[1] = hq.total_evals; stats[2] = hq.distinct_elems;
    stats[4] = 0; stats[5] = (long long)hq.blob.size();
    first_row = &hq.first_row;
  }
  stats[3] = n_nodes;
  if (rows_hist) {
    for (int l = 0; l <= FEA_G_MAX_ROWS; ++l) rows_hist[l] = 0;
    for (size_t p = 0; p + 1 < first_row->size(); ++p) ++rows_hist[std::min((*first_row)[p + 1] - (*first_row)[p], FEA_G_MAX_ROWS)];
  }
  return FEAHIP_OK;
}

#ifdef FEAHIP_DEBUG
// diagnostic build only, host only: one chunk's map record and the layout, for the LDS bank model (dbg/lds_model.py)
extern "C" int feahip_debug_gather_record_host(int n_nodes, int n_elems, const int *elements, int chunk, int *layout_ints,
                                               unsigned char *record, int *nchunks)
{
  static HostGather hg;                       // cached between the sizing call and the copying call
  static int cached_n = -1, cached_e = -1;
  if (cached_n != n_nodes || cached_e != n_elems) {
    HostPattern hp;
    std::string err;
    if (build_host_pattern(n_nodes, n_elems, 4, elements, hp, err)) return FEAHIP_EINVAL;
    build_host_gather(n_nodes, n_elems, elements, hp, 0, n_nodes, hg);
    if (!hg.ok) return FEAHIP_EINVAL;
    cached_n = n_nodes; cached_e = n_elems;
  }
  if (chunk < 0) chunk = hg.nchunks / 2;
  memcpy(layout_ints, &hg.lay, sizeof(GatherLayout));
  if (nchunks) *nchunks = hg.nchunks;
  if (record) memcpy(record, hg.blob.data() + (size_t)chunk * hg.lay.stride, hg.lay.stride);
  return (int)(sizeof(GatherLayout) / sizeof(int));
}
#endif
