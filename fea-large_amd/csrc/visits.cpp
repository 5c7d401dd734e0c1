// visits.cpp -- one-time host construction of the maps of the staged-visit assembly of linear tetrahedra
// (kernels_visit.hip): per assembly chunk the nodes its elements touch and one 8-byte record per (row, element) visit,
// scheduled against the LDS banks.
#include "feahip_internal.h"
#include <algorithm>
#include <cstdlib>
#include <thread>

namespace {
template <class F>
void par_for(int n, F f)
{
  unsigned hw = std::thread::hardware_concurrency();
  int nt = (int)std::min<unsigned>(hw ? hw : 4, 32);
  if (n < 4096) nt = 1;
  if (nt <= 1) { f(0, n); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) {
    int lo = (int)((long long)n * t / nt), hi = (int)((long long)n * (t + 1) / nt);
    th.emplace_back([=] { f(lo, hi); });
  }
  for (auto &x : th) x.join();
}
}  // namespace

// ---------------------------------------------------------------------------
// visit records for kernels_visit.hip
// ---------------------------------------------------------------------------
void build_host_visits(int N, int E, const int *conn, const HostPattern &hp, HostVisits &out)
{
  (void)N;
  out.ok = false;
  if (hp.achunk.size() < 2) return;
  const int np = (int)hp.achunk.size() - 1;
  out.desc.resize((size_t)np);
  // Records are laid out in whole passes of 64 lanes per chunk: the lanes a chunk leaves idle (its visit count
  // is rarely a multiple of 64) are spread over the 16-lane LDS groups instead of trailing the last pass, which
  // gives the bank-aware schedule below room (a group of 14-15 visits needs 14-15 distinct residues out of 16,
  // not 16 out of 16).  Idle lanes carry the null record (slots = 0xFFFFFFFF).
  std::vector<long long> voff((size_t)np + 1, 0);
  for (int p = 0; p < np; ++p) {
    const int n = hp.incptr[hp.achunk[p + 1]] - hp.incptr[hp.achunk[p]];
    voff[p + 1] = voff[p] + 64LL * ((n + 63) / 64);
  }
  if (voff[np] > 0x7FFFFFFFLL) return;
  out.vrec.assign((size_t)voff[np] * 2, 0xFFFFFFFFu);
  out.vnode.assign((size_t)np * FEA_VISIT_MAX_NODES, 0);
  std::vector<char> bad((size_t)np, 0);
  const char *ord = getenv("FEAHIP_VISIT_ORDER");
  const bool interleave_rows = !(ord && ord[0] == '1');
  const bool bank_aware = !(ord && (ord[0] == '0' || ord[0] == '1'));
  const int nsweeps = (ord && ord[0] == '3') ? 2 : 0;
  par_for(np, [&](int lo, int hi) {
    std::vector<int> halo;
    for (int p = lo; p < hi; ++p) {
      const int r0 = hp.achunk[p], r1 = hp.achunk[p + 1];
      const int p0 = hp.incptr[r0], p1 = hp.incptr[r1];
      const int b0 = hp.rowptr[r0];
      VisitDesc &d = out.desc[p];
      d.r0 = r0; d.r1 = r1; d.b0 = b0; d.nb = hp.rowptr[r1] - b0;
      d.node_off = p * FEA_VISIT_MAX_NODES; d.visit_off = (int)voff[p]; d.nvisit = p1 - p0;
      // owned rows first (chunk-local id = row - r0), then the other nodes ascending
      halo.clear();
      for (int q = p0; q < p1; ++q) {
        const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu);
        for (int k = 0; k < 4; ++k) {
          const int g = conn[(size_t)e * 4 + k];
          if (g < r0 || g >= r1) halo.push_back(g);
        }
      }
      std::sort(halo.begin(), halo.end());
      halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
      const int nown = r1 - r0;
      d.nnode = nown + (int)halo.size();
      if (d.nnode > FEA_VISIT_MAX_NODES || d.nvisit > FEA_VISIT_MAX_VISITS || d.nb > FEA_ACHUNK_BLOCKS ||
          nown > FEA_ACHUNK_ROWS) { bad[p] = 1; continue; }
      int *vn = out.vnode.data() + (size_t)p * FEA_VISIT_MAX_NODES;
      for (int r = r0; r < r1; ++r) vn[r - r0] = r;
      std::copy(halo.begin(), halo.end(), vn + nown);
      // chunk-local id of a halo node: position in `halo` -> id (identity order unless renumbered below)
      std::vector<int> halo_id(halo.size());
      for (size_t h = 0; h < halo.size(); ++h) halo_id[h] = nown + (int)h;
      auto lid = [&](int g) {
        if (g >= r0 && g < r1) return g - r0;
        return halo_id[(size_t)(std::lower_bound(halo.begin(), halo.end(), g) - halo.begin())];
      };
      struct V { uint32_t w; int row; int slot[3]; int node[3]; int round; int order[3]; };
      std::vector<V> vs;
      vs.reserve((size_t)d.nvisit);
      for (int r = r0; r < r1; ++r)
        for (int q = hp.incptr[r]; q < hp.incptr[r + 1]; ++q) {
          V v; v.w = hp.inc_rows[q]; v.row = r - r0; v.round = 0;
          const int e = (int)(v.w & 0x0FFFFFFFu), la = (int)(v.w >> 28);
          const int *cb = hp.colidx.data() + hp.rowptr[r], *ce = hp.colidx.data() + hp.rowptr[r + 1];
          int m = 0;
          for (int k = 0; k < 4; ++k) {
            if (k == la) continue;
            v.node[m] = k;                                                   // local index in the stored element
            v.slot[m] = (int)(std::lower_bound(cb, ce, conn[(size_t)e * 4 + k]) - cb);
            ++m;
          }
          v.order[0] = 0; v.order[1] = 1; v.order[2] = 2;
          vs.push_back(v);
        }
      static const int perms[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {2, 1, 0}, {1, 0, 2}};
      if (bank_aware) {
        // Schedule of the chunk's visits against LDS bank conflicts.  A pass
        // of the kernel = 64 visits, each adding its three off-diagonal blocks
        // in three steps of nine ds_add_f64.  The LDS serves a 64-lane f64
        // atomic as four groups of 16 consecutive lanes, and inside a group
        // two lanes collide when their addresses agree mod 16 doubles
        // (profiles/r01_microbench_lds_f64_atomic.txt: 8.3 clk conflict-free,
        // 26 clk for random blocks).  A block at tile position p sits at
        // doubles 9p..9p+8, and 9 is odd: lanes collide iff their blocks agree
        // mod 16.  So the visits are dealt to groups of 16 lanes, and each
        // visit's three columns to the three steps, so that inside a group the
        // blocks of one step are distinct mod 16: greedy, least collisions
        // first, then a pass of pairwise swaps for the visits still colliding.
        const int n = (int)vs.size();
        const int ngroups = 4 * ((n + 63) / 64);                       // every 16-lane group of the chunk's passes
        const int gcap = (n + ngroups - 1) / ngroups;                  // visits per group, evenly (<= 16)
        // pos = tile position of the block; two lanes on one block (same address) serialise
        // harder than two blocks on one bank, so that costs 4 collisions
        std::vector<int> pos((size_t)n * 3);
        for (int i = 0; i < n; ++i)
          for (int m = 0; m < 3; ++m) pos[(size_t)i * 3 + m] = hp.rowptr[r0 + vs[i].row] - b0 + vs[i].slot[m];
        std::vector<uint8_t> cnt((size_t)ngroups * 3 * 16, 0), cnta((size_t)ngroups * 3 * 256, 0);
        auto CA = [&](int g, int st, int ps) -> uint8_t & { return cnta[((size_t)g * 3 + st) * 256 + ps]; };
        std::vector<int> gsize((size_t)ngroups, 0), gof((size_t)n, -1), pof((size_t)n, 0);
        auto C = [&](int g, int st, int res) -> uint8_t & { return cnt[((size_t)g * 3 + st) * 16 + res]; };
        auto cap = [&](int) { return gcap; };
        auto cost_in = [&](int i, int g, int pi) {
          int cst = 0;
          for (int st = 0; st < 3; ++st) {
            const int ps = pos[(size_t)i * 3 + perms[pi][st]];
            cst += C(g, st, ps & 15) + 3 * CA(g, st, ps);
          }
          return cst;
        };
        auto put = [&](int i, int g, int pi, int delta) {
          for (int st = 0; st < 3; ++st) {
            const int ps = pos[(size_t)i * 3 + perms[pi][st]];
            C(g, st, ps & 15) = (uint8_t)(C(g, st, ps & 15) + delta);
            CA(g, st, ps) = (uint8_t)(CA(g, st, ps) + delta);
          }
          gsize[g] += delta;
        };
        // deal order: round-robin over the rows, so a group mixes rows evenly
        std::vector<int> order; order.reserve((size_t)n);
        {
          std::vector<std::vector<int>> per((size_t)nown);
          for (int i = 0; i < n; ++i) per[(size_t)vs[i].row].push_back(i);
          for (size_t k = 0;; ++k) {
            bool any = false;
            for (auto &pr : per) if (k < pr.size()) { order.push_back(pr[k]); any = true; }
            if (!any) break;
          }
        }
        for (int i : order) {
          int bg = -1, bp = 0, bc = 1 << 30;
          for (int g = 0; g < ngroups; ++g) {
            if (gsize[g] >= cap(g)) continue;
            for (int pi = 0; pi < 6; ++pi) {
              const int cst = cost_in(i, g, pi) * 64 + gsize[g];
              if (cst < bc) { bc = cst; bg = g; bp = pi; }
            }
          }
          gof[i] = bg; pof[i] = bp; put(i, bg, bp, +1);
        }
        // pairwise swaps for the visits that still collide
        for (int sweep = 0; sweep < nsweeps; ++sweep) {
          bool changed = false;
          for (int i = 0; i < n; ++i) {
            put(i, gof[i], pof[i], -1);
            const int ci = cost_in(i, gof[i], pof[i]);
            put(i, gof[i], pof[i], +1);
            if (ci == 0) continue;
            int best_j = -1, best_pi = 0, best_pj = 0, best_gain = 0;
            for (int j = 0; j < n; ++j) {
              if (gof[j] == gof[i]) continue;
              const int gi = gof[i], gj = gof[j];
              put(i, gi, pof[i], -1); put(j, gj, pof[j], -1);
              const int before = cost_in(i, gi, pof[i]) + cost_in(j, gj, pof[j]);
              int ai = 1 << 30, api = 0, aj = 1 << 30, apj = 0;
              for (int pi = 0; pi < 6; ++pi) {
                const int x = cost_in(i, gj, pi); if (x < ai) { ai = x; api = pi; }
                const int y = cost_in(j, gi, pi); if (y < aj) { aj = y; apj = pi; }
              }
              put(i, gi, pof[i], +1); put(j, gj, pof[j], +1);
              const int gain = before - (ai + aj);
              if (gain > best_gain) { best_gain = gain; best_j = j; best_pi = api; best_pj = apj; }
            }
            if (best_j >= 0) {
              const int j = best_j, gi = gof[i], gj = gof[j];
              put(i, gi, pof[i], -1); put(j, gj, pof[j], -1);
              gof[i] = gj; pof[i] = best_pi; gof[j] = gi; pof[j] = best_pj;
              put(i, gj, best_pi, +1); put(j, gi, best_pj, +1);
              changed = true;
            }
          }
          if (!changed) break;
        }
        for (int i = 0; i < n; ++i) { vs[i].round = gof[i]; for (int st = 0; st < 3; ++st) vs[i].order[st] = perms[pof[i]][st]; }
        std::stable_sort(vs.begin(), vs.end(), [](const V &x, const V &y) { return x.round < y.round; });
      } else {
        // Legacy schedule (FEAHIP_VISIT_ORDER=0/1).  A pass of the kernel = 64 visits, each
        // adding its three off-diagonal blocks in three steps; lanes that add to
        // the same (row, column) block in the same step serialise in the LDS
        // (~11 clk per extra lane).  So (1) the visits of every row are split
        // over the passes keeping each column's count per pass low, and (2) each
        // visit's three column nodes are ordered so that, step by step, the
        // visits of one row in one pass hit different columns (greedy
        // edge-colouring of the visit x column graph).
        const int nrounds = (d.nvisit + 63) / 64;
        // (1) pass of every visit
        std::vector<int> rsize((size_t)nrounds, 0);
        std::vector<uint8_t> deg((size_t)nrounds * nown * 256, 0);
        auto D = [&](int rd, int row, int slot) -> uint8_t & { return deg[((size_t)rd * nown + row) * 256 + slot]; };
        const int cap = (d.nvisit + nrounds - 1) / nrounds;
        for (auto &v : vs) {
          int best = -1, bestcost = 1 << 30;
          for (int rd = 0; rd < nrounds; ++rd) {
            if (rsize[rd] >= std::min(64, cap)) continue;
            int mx = 0;
            for (int m = 0; m < 3; ++m) mx = std::max(mx, (int)D(rd, v.row, v.slot[m]));
            const int cost = mx * 1024 + rsize[rd];
            if (cost < bestcost) { bestcost = cost; best = rd; }
          }
          if (best < 0) best = (int)(std::min_element(rsize.begin(), rsize.end()) - rsize.begin());
          v.round = best; rsize[best]++;
          for (int m = 0; m < 3; ++m) D(best, v.row, v.slot[m])++;
        }
        // (2) step of every column inside its visit
        std::vector<uint8_t> used((size_t)nrounds * nown * 3 * 256, 0);
        auto U = [&](int rd, int row, int step, int slot) -> uint8_t & { return used[(((size_t)rd * nown + row) * 3 + step) * 256 + slot]; };
        for (auto &v : vs) {
          int bestp = 0, bestc = 1 << 30;
          for (int pi = 0; pi < 6; ++pi) {
            int cst = 0;
            for (int st = 0; st < 3; ++st) cst += U(v.round, v.row, st, v.slot[perms[pi][st]]);
            if (cst < bestc) { bestc = cst; bestp = pi; }
          }
          for (int st = 0; st < 3; ++st) { v.order[st] = perms[bestp][st]; U(v.round, v.row, st, v.slot[v.order[st]])++; }
        }
        // emit pass by pass
        std::stable_sort(vs.begin(), vs.end(), [](const V &x, const V &y) { return x.round < y.round; });
        if (interleave_rows) {
          // inside a pass, deal the visits round-robin over the rows
          std::vector<V> tmp2; tmp2.reserve(vs.size());
          size_t b = 0;
          while (b < vs.size()) {
            size_t e2 = b;
            while (e2 < vs.size() && vs[e2].round == vs[b].round) ++e2;
            std::vector<std::vector<V>> per((size_t)nown);
            for (size_t i = b; i < e2; ++i) per[(size_t)vs[i].row].push_back(vs[i]);
            for (size_t k = 0;; ++k) {
              bool any = false;
              for (auto &pr : per) if (k < pr.size()) { tmp2.push_back(pr[k]); any = true; }
              if (!any) break;
            }
            b = e2;
          }
          vs.swap(tmp2);
        }
      }
      std::vector<int> fill((size_t)4 * ((vs.size() + 63) / 64), 0), lane_of(vs.size());
      for (int i = 0; i < (int)vs.size(); ++i)
        lane_of[(size_t)i] = bank_aware ? 16 * vs[i].round + fill[(size_t)vs[i].round]++ : i;   // inside its group, or packed
      if (bank_aware && !halo.empty()) {
        // Chunk-local ids of the halo nodes against bank conflicts of the coordinate reads.  A lane reads its four
        // nodes with ds_read_b128 (48-byte records: 16-byte slot = (3 id + part) mod 16, so two distinct nodes of one
        // read collide iff their ids agree mod 16); the LDS serves such a read in groups of 16 lanes
        // ({0-3,12-15,20-27}, {4-11,16-19,28-31} of every 32).  Owned rows keep id = row; halo nodes take the free id
        // (up to 63) whose residue is least used by the nodes they share a (lane group, node position) read with.
        static const uint8_t grp_of[32] = {0,0,0,0,1,1,1,1,1,1,1,1,0,0,0,0,1,1,1,1,0,0,0,0,0,0,0,0,1,1,1,1};
        const int nhalf = (int)(fill.size() / 4) * 2, nsets = nhalf * 2 * 4;
        std::vector<std::vector<int>> members((size_t)nsets);               // global node ids, distinct
        for (int i = 0; i < (int)vs.size(); ++i) {
          const V &v = vs[i];
          const int e = (int)(v.w & 0x0FFFFFFFu), la = (int)(v.w >> 28), ln = lane_of[(size_t)i];
          const int nd[4] = {la, v.node[v.order[0]], v.node[v.order[1]], v.node[v.order[2]]};
          for (int k = 0; k < 4; ++k) {
            auto &m = members[(size_t)(((ln >> 5) * 2 + grp_of[ln & 31]) * 4 + k)];
            const int g = conn[(size_t)e * 4 + nd[k]];
            if (std::find(m.begin(), m.end(), g) == m.end()) m.push_back(g);
          }
        }
        std::vector<uint8_t> cnt((size_t)nsets * 16, 0);
        std::vector<std::vector<int>> sets_of(halo.size());
        for (int sidx = 0; sidx < nsets; ++sidx)
          for (int g : members[(size_t)sidx]) {
            if (g >= r0 && g < r1) cnt[(size_t)sidx * 16 + ((g - r0) & 15)]++;
            else sets_of[(size_t)(std::lower_bound(halo.begin(), halo.end(), g) - halo.begin())].push_back(sidx);
          }
        std::vector<int> byload(halo.size());
        for (size_t h = 0; h < halo.size(); ++h) byload[h] = (int)h;
        std::stable_sort(byload.begin(), byload.end(), [&](int a, int b) { return sets_of[(size_t)a].size() > sets_of[(size_t)b].size(); });
        bool used[FEA_VISIT_MAX_NODES] = {false};
        int top = nown;
        for (int h : byload) {
          int cost[16] = {0};
          for (int sidx : sets_of[(size_t)h]) for (int r = 0; r < 16; ++r) cost[r] += cnt[(size_t)sidx * 16 + r];
          int best = -1;
          for (int idc = nown; idc < FEA_VISIT_MAX_NODES; ++idc)
            if (!used[idc] && (best < 0 || cost[idc & 15] < cost[best & 15])) best = idc;
          used[best] = true; halo_id[(size_t)h] = best; top = std::max(top, best + 1);
          for (int sidx : sets_of[(size_t)h]) cnt[(size_t)sidx * 16 + (best & 15)]++;
        }
        // node list in id order; unused ids read an owned row (harmless)
        for (int idc = nown; idc < top; ++idc) vn[idc] = r0;
        for (size_t h = 0; h < halo.size(); ++h) vn[halo_id[h]] = halo[h];
        d.nnode = top;
      }
      for (int i = 0; i < (int)vs.size(); ++i) {
        const V &v = vs[i];
        const int e = (int)(v.w & 0x0FFFFFFFu), la = (int)(v.w >> 28);
        int perm[4] = {la, v.node[v.order[0]], v.node[v.order[1]], v.node[v.order[2]]};
        int inv = 0;                                     // parity of the renumbering (orientation flips when odd)
        for (int x = 0; x < 4; ++x) for (int y = x + 1; y < 4; ++y) inv += perm[x] > perm[y];
        uint32_t ids = 0, sl = (uint32_t)(inv & 1);
        for (int k = 0; k < 4; ++k) ids |= (uint32_t)lid(conn[(size_t)e * 4 + perm[k]]) << (8 * k);   // row node first
        // tile position of the block (row start + column slot): the kernel needs no row table in its passes
        for (int k = 1; k < 4; ++k) sl |= (uint32_t)(hp.rowptr[r0 + v.row] - b0 + v.slot[v.order[k - 1]]) << (8 * k);
        const int at = lane_of[(size_t)i];
        out.vrec[((size_t)d.visit_off + at) * 2] = ids;
        out.vrec[((size_t)d.visit_off + at) * 2 + 1] = sl;
      }
      d.nvisit = 64 * ((d.nvisit + 63) / 64);            // the kernel walks whole passes; idle lanes hold the null record
    }
  });
  for (int p = 0; p < np; ++p)
    if (bad[p]) { out.desc.clear(); out.vrec.clear(); out.vnode.clear(); return; }
  if (hp.max_rowlen > 255) { out.desc.clear(); out.vrec.clear(); out.vnode.clear(); return; }
  out.ok = true;
}
// ---------------------------------------------------------------------------
// maps of the shared-state assembly of 10-node elements (kernels_quad.hip)
// ---------------------------------------------------------------------------
// chunks [p_lo, p_hi) of the assembly partition only: a rank builds (and uploads) the maps of the rows it owns,
// with offsets relative to its own arrays -- at 50M quadratic tets the pair list of the whole mesh (4.5e9 words)
// neither fits 32-bit offsets nor belongs on every rank
void build_host_quad(int N, int E, int npe, const int *conn, const HostPattern &hp, int p_lo, int p_hi, HostQuad &out)
{
  (void)N; (void)E;
  out.ok = false;
  if (hp.achunk.size() < 2 || npe != 10 || hp.max_rowlen > 255) return;
  if (p_lo < 0 || p_hi > (int)hp.achunk.size() - 1 || p_lo > p_hi) return;
  const int *achunk = hp.achunk.data() + p_lo;
  const int np = p_hi - p_lo;
  out.desc.resize((size_t)np);
  // sizes first (prefix sums), then fill in parallel
  std::vector<int> nel((size_t)np, 0), nnd((size_t)np, 0);
  par_for(np, [&](int lo, int hi) {
    std::vector<int> el, nd;
    for (int p = lo; p < hi; ++p) {
      const int r0 = achunk[p], r1 = achunk[p + 1];
      el.clear(); nd.clear();
      for (int q = hp.incptr[r0]; q < hp.incptr[r1]; ++q) el.push_back((int)(hp.inc_rows[q] & 0x0FFFFFFFu));
      std::sort(el.begin(), el.end());
      el.erase(std::unique(el.begin(), el.end()), el.end());
      for (int e : el) for (int k = 0; k < npe; ++k) nd.push_back(conn[(size_t)e * npe + k]);
      std::sort(nd.begin(), nd.end());
      nel[p] = (int)el.size();
      nnd[p] = (int)(std::unique(nd.begin(), nd.end()) - nd.begin());
    }
  });
  size_t eo = 0, po = 0, no = 0;
  for (int p = 0; p < np; ++p) {
    const int r0 = achunk[p], r1 = achunk[p + 1];
    QuadDesc &d = out.desc[p];
    d.r0 = r0; d.r1 = r1; d.b0 = hp.rowptr[r0]; d.nb = hp.rowptr[r1] - d.b0;
    d.elem_off = (int)eo; d.nelem = nel[p];
    d.pair_off = (int)po; d.npair = (hp.incptr[r1] - hp.incptr[r0]) * (npe - 1);
    d.node_off = (int)no; d.nnode = nnd[p];
    eo += (size_t)d.nelem; po += (size_t)d.npair; no += (size_t)d.nnode;
    if (d.nelem > FEA_QUAD_ELEMS || d.nb > FEA_QUAD_BLOCKS || d.nnode > FEA_QUAD_NODES || r1 - r0 > FEA_CHUNK_ROWS ||
        hp.incptr[r1] - hp.incptr[r0] > FEA_QUAD_VISITS || po > 0x7FFFFFFFull) { out.desc.clear(); return; }
  }
  out.qelem.assign(eo * 3, 0u); out.qpair.resize(po); out.qnode.resize(no);
  par_for(np, [&](int lo, int hi) {
    std::vector<int> el, nd;
    for (int p = lo; p < hi; ++p) {
      const QuadDesc &d = out.desc[p];
      el.clear(); nd.clear();
      for (int q = hp.incptr[d.r0]; q < hp.incptr[d.r1]; ++q) el.push_back((int)(hp.inc_rows[q] & 0x0FFFFFFFu));
      std::sort(el.begin(), el.end());
      el.erase(std::unique(el.begin(), el.end()), el.end());
      for (int e : el) for (int k = 0; k < npe; ++k) nd.push_back(conn[(size_t)e * npe + k]);
      std::sort(nd.begin(), nd.end());
      nd.erase(std::unique(nd.begin(), nd.end()), nd.end());
      std::copy(nd.begin(), nd.end(), out.qnode.begin() + d.node_off);
      for (int i = 0; i < d.nelem; ++i) {
        uint8_t b[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < npe; ++k)
          b[k] = (uint8_t)(std::lower_bound(nd.begin(), nd.end(), conn[(size_t)el[i] * npe + k]) - nd.begin());
        const int n0 = conn[(size_t)el[i] * npe];
        b[10] = (n0 >= d.r0 && n0 < d.r1) ? 1 : 0;
        uint32_t *w = out.qelem.data() + ((size_t)d.elem_off + i) * 3;
        for (int k = 0; k < 12; ++k) w[k / 4] |= (uint32_t)b[k] << (8 * (k % 4));
      }
      // pairs, visit-major: the npe-1 lanes of a visit read the same state entry (LDS broadcast)
      size_t w = (size_t)d.pair_off;
      for (int r = d.r0; r < d.r1; ++r) {
        const int *cb = hp.colidx.data() + hp.rowptr[r], *ce = hp.colidx.data() + hp.rowptr[r + 1];
        const uint32_t rowpos = (uint32_t)(hp.rowptr[r] - d.b0);
        for (int q = hp.incptr[r]; q < hp.incptr[r + 1]; ++q) {
          const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu), la = (int)(hp.inc_rows[q] >> 28);
          const uint32_t eli = (uint32_t)(std::lower_bound(el.begin(), el.end(), e) - el.begin());
          bool first = true;
          for (int lb = 0; lb < npe; ++lb) {
            if (lb == la) continue;
            const uint32_t slot = (uint32_t)(std::lower_bound(cb, ce, conn[(size_t)e * npe + lb]) - cb);
            out.qpair[w++] = eli | ((uint32_t)la << 6) | ((uint32_t)lb << 10) | ((rowpos + slot) << 14) |
                             ((uint32_t)(r - d.r0) << 22) | (first ? (1u << 26) : 0u);
            first = false;
          }
        }
      }
    }
  });
  out.ok = true;
}
