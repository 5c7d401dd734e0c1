// patches.cpp -- one-time host construction of the maps the PATCH assembly
// kernel (kernels_patch.hip) walks.
//
// A patch is one chunk of consecutive block rows.  For every patch:
//   pnode   the unique nodes its elements touch (their coordinates are
//           staged in LDS once per launch),
//   pelem   the unique elements that touch its rows, as 4 patch-local node
//           ids each (every element state is evaluated once per patch, not
//           once per (element, node) visit),
//   pent    for every off-diagonal block of the patch, the list of
//           (element, local row node, local column node) contributions that
//           sum to it, in ascending patch-element order (fixed summation
//           order => bitwise reproducible assembly).
// Diagonal blocks have no entries: they are minus the sum of their row.
#include "feahip_internal.h"
#include <algorithm>
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

struct Local {            // one patch, built independently
  std::vector<int> nodes;
  std::vector<uint16_t> elems, ents, bptr;
  bool ok = true;
};
}  // namespace

void build_host_patches(int N, int E, const int *conn, const HostPattern &hp, HostPatches &out)
{
  (void)N; (void)E;
  const int np = (int)hp.chunk.size() - 1;
  std::vector<Local> loc((size_t)np);
  par_for(np, [&](int lo, int hi) {
    std::vector<int> el, order;
    std::vector<std::pair<int, uint16_t>> tmp;      // (block, entry)
    for (int p = lo; p < hi; ++p) {
      Local &L = loc[p];
      const int r0 = hp.chunk[p], r1 = hp.chunk[p + 1];
      const int b0 = hp.rowptr[r0], nb = hp.rowptr[r1] - b0;
      // unique elements of the patch
      el.clear();
      for (int q = hp.incptr[r0]; q < hp.incptr[r1]; ++q) el.push_back((int)(hp.inc[q] & 0x0FFFFFFFu));
      std::sort(el.begin(), el.end());
      el.erase(std::unique(el.begin(), el.end()), el.end());
      const int ne = (int)el.size();
      // spread every batch of 64 elements over the whole patch: stride walk
      // with a stride coprime to ne (keeps the per-batch work of the block
      // owners even)
      order.assign(el.begin(), el.end());
      if (ne > 64) {
        int stride = (int)(ne * 0.6180339887) | 1;
        auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
        while (gcd(stride, ne) != 1) stride += 2;
        for (int i = 0; i < ne; ++i) order[i] = el[(int)(((long long)i * stride) % ne)];
      }
      // unique nodes
      L.nodes.clear();
      for (int e : order) for (int k = 0; k < 4; ++k) L.nodes.push_back(conn[(size_t)e * 4 + k]);
      std::sort(L.nodes.begin(), L.nodes.end());
      L.nodes.erase(std::unique(L.nodes.begin(), L.nodes.end()), L.nodes.end());
      auto lnode = [&](int g) { return (int)(std::lower_bound(L.nodes.begin(), L.nodes.end(), g) - L.nodes.begin()); };
      L.elems.resize((size_t)ne * 4);
      for (int i = 0; i < ne; ++i)
        for (int k = 0; k < 4; ++k) L.elems[(size_t)i * 4 + k] = (uint16_t)lnode(conn[(size_t)order[i] * 4 + k]);
      // contributions, bucketed by block
      tmp.clear();
      for (int i = 0; i < ne; ++i) {
        const int *c = conn + (size_t)order[i] * 4;
        for (int la = 0; la < 4; ++la) {
          const int a = c[la];
          if (a < r0 || a >= r1) continue;         // row owned by another patch
          const int *cb = hp.colidx.data() + hp.rowptr[a], *ce = hp.colidx.data() + hp.rowptr[a + 1];
          bool first = true;
          for (int lb = 0; lb < 4; ++lb) {
            if (lb == la) continue;
            const int blk = hp.rowptr[a] - b0 + (int)(std::lower_bound(cb, ce, c[lb]) - cb);
            tmp.emplace_back(blk, (uint16_t)(i | (la << 11) | (lb << 13) | ((first ? 1 : 0) << 15)));
            first = false;
          }
        }
      }
      std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int, uint16_t> &x, const std::pair<int, uint16_t> &y) { return x.first < y.first; });
      L.ents.resize(tmp.size());
      L.bptr.assign((size_t)nb + 1, 0);
      for (size_t i = 0; i < tmp.size(); ++i) { L.ents[i] = tmp[i].second; L.bptr[tmp[i].first + 1]++; }
      for (int b = 0; b < nb; ++b) L.bptr[b + 1] = (uint16_t)(L.bptr[b + 1] + L.bptr[b]);
      L.ok = ne <= FEA_PATCH_MAX_ELEMS && (int)L.nodes.size() <= FEA_PATCH_MAX_NODES &&
             (int)tmp.size() <= FEA_PATCH_MAX_ENTRIES && nb <= FEA_CHUNK_BLOCKS;
    }
  });
  out.ok = true;
  out.desc.resize((size_t)np);
  size_t no = 0, eo = 0, to = 0, bo = 0;
  for (int p = 0; p < np; ++p) {
    const Local &L = loc[p];
    if (!L.ok) out.ok = false;
    PatchDesc &d = out.desc[p];
    d.r0 = hp.chunk[p]; d.r1 = hp.chunk[p + 1];
    d.b0 = hp.rowptr[d.r0]; d.nb = hp.rowptr[d.r1] - d.b0;
    d.node_off = (int)no; d.nnode = (int)L.nodes.size();
    d.elem_off = (int)eo; d.nelem = (int)(L.elems.size() / 4);
    d.ent_off = (int)to; d.nent = (int)L.ents.size();
    d.bptr_off = (int)bo; d.pad = 0;
    no += L.nodes.size(); eo += L.elems.size() / 4; to += L.ents.size(); bo += L.bptr.size();
    if (no > 0x7FFFFFFFull || to > 0x7FFFFFFFull) { out.ok = false; break; }
  }
  if (!out.ok) { out.desc.clear(); return; }
  out.pnode.resize(no); out.pelem.resize(eo * 4); out.pent.resize(to); out.pbptr.resize(bo);
  par_for(np, [&](int lo, int hi) {
    for (int p = lo; p < hi; ++p) {
      const Local &L = loc[p];
      const PatchDesc &d = out.desc[p];
      std::copy(L.nodes.begin(), L.nodes.end(), out.pnode.begin() + d.node_off);
      std::copy(L.elems.begin(), L.elems.end(), out.pelem.begin() + (size_t)d.elem_off * 4);
      std::copy(L.ents.begin(), L.ents.end(), out.pent.begin() + d.ent_off);
      std::copy(L.bptr.begin(), L.bptr.end(), out.pbptr.begin() + d.bptr_off);
    }
  });
}

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
  out.vrec.assign((size_t)E * 4 * 2, 0);
  out.vnode.assign((size_t)np * FEA_VISIT_MAX_NODES, 0);
  std::vector<char> bad((size_t)np, 0);
  par_for(np, [&](int lo, int hi) {
    std::vector<int> halo;
    std::vector<uint32_t> vis;
    for (int p = lo; p < hi; ++p) {
      const int r0 = hp.achunk[p], r1 = hp.achunk[p + 1];
      const int p0 = hp.incptr[r0], p1 = hp.incptr[r1];
      const int b0 = hp.rowptr[r0];
      VisitDesc &d = out.desc[p];
      d.r0 = r0; d.r1 = r1; d.b0 = b0; d.nb = hp.rowptr[r1] - b0;
      d.node_off = p * FEA_VISIT_MAX_NODES; d.visit_off = p0; d.nvisit = p1 - p0;
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
          nown > FEA_CHUNK_ROWS) { bad[p] = 1; continue; }
      int *vn = out.vnode.data() + (size_t)p * FEA_VISIT_MAX_NODES;
      for (int r = r0; r < r1; ++r) vn[r - r0] = r;
      std::copy(halo.begin(), halo.end(), vn + nown);
      auto lid = [&](int g) {
        if (g >= r0 && g < r1) return g - r0;
        return nown + (int)(std::lower_bound(halo.begin(), halo.end(), g) - halo.begin());
      };
      // visits dealt round-robin over the rows: the lanes of one pass then work
      // on as many different rows as the chunk has (few same-address LDS adds)
      vis.clear();
      int maxlen = 0;
      for (int r = r0; r < r1; ++r) maxlen = std::max(maxlen, hp.incptr[r + 1] - hp.incptr[r]);
      for (int k = 0; k < maxlen; ++k)
        for (int r = r0; r < r1; ++r)
          if (k < hp.incptr[r + 1] - hp.incptr[r]) vis.push_back(hp.inc_rows[hp.incptr[r] + k]);
      for (int v = 0; v < (int)vis.size(); ++v) {
        const int e = (int)(vis[v] & 0x0FFFFFFFu), la = (int)(vis[v] >> 28);
        const int a = conn[(size_t)e * 4 + la];
        const int *cb = hp.colidx.data() + hp.rowptr[a], *ce = hp.colidx.data() + hp.rowptr[a + 1];
        uint32_t ids = 0, sl = 0;
        for (int k = 0; k < 4; ++k) {
          const int g = conn[(size_t)e * 4 + (k ^ la)];                 // row node first
          ids |= (uint32_t)lid(g) << (8 * k);
          if (k) sl |= (uint32_t)(std::lower_bound(cb, ce, g) - cb) << (8 * k);
        }
        out.vrec[(size_t)(p0 + v) * 2] = ids;
        out.vrec[(size_t)(p0 + v) * 2 + 1] = sl;
      }
    }
  });
  for (int p = 0; p < np; ++p)
    if (bad[p]) { out.desc.clear(); out.vrec.clear(); out.vnode.clear(); return; }
  if (hp.max_rowlen > 255) { out.desc.clear(); out.vrec.clear(); out.vnode.clear(); return; }
  out.ok = true;
}
