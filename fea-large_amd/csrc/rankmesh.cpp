// rankmesh.cpp -- what ONE rank of a sharded run holds (host, once per context).
//
// The reference keeps one row-wise store of the whole matrix (fea_solver.c:444-448) in one process.  A rank of a
// sharded run here owns a slab of nodes -- a range of LIBRARY node ids (renumber.cpp: compact cells in slabs across
// the longest axis) -- and holds nothing but its slab: the elements that touch its nodes, those elements' nodes (its
// own first, then the halo nodes in library order), the block rows of its own nodes (pattern built from its own
// elements only: every element around an owned node is local, so an owned row is complete), K and the vectors over
// its local nodes.  Everything is indexed locally; the halo plan (who sends which rows to whom) is derived on every
// rank from its own elements: rank r needs my node a exactly when a shares an element with a node r owns, which both
// sides see.  No global pattern, no global K index space: the 50M-element TET10 block of BASELINE.json configs[4]
// (1.9e9 3x3 blocks, 140 GB of values) is eight independent index spaces of 2.4e8 blocks.
#include "feahip_internal.h"
#include <algorithm>
#include <cstring>

// library ids [g0, g1) of rank `rank`: equal node counts, rounded to a multiple of the numbering's FULL cell (renumber.cpp:
// 4 x 4 x 4 nodes, 48 half-grid nodes for 10-node elements).  The nodes of a cell are consecutive ids, so while the cells
// before the cut are full ones the cut falls between two cells and no gather chunk is split between two ranks; behind
// the first PARTIAL cell (the last cell of every line of a block whose node count is not a multiple of four) the
// multiples no longer coincide with cell boundaries and a cut may fall inside a cell -- that costs the two ranks one
// irregular chunk each, nothing else.  Cutting at whole layers of cells instead would make the slab faces planar but
// the ranks unequal (the 10M block has 100 layers: 12 or 13 per rank of eight, 4 % of the strong-scaling efficiency).
static int rank_cut(int N, int npe, int k, int nranks)
{
  constexpr int gc[3] = {FEA_G_CELL};
  const int cell = npe == 10 ? 48 : gc[0] * gc[1] * gc[2];
  if (k <= 0) return 0;
  if (k >= nranks) return N;
  const long long t = (long long)N * k / nranks;
  const long long c = (t + cell / 2) / cell * cell;
  return (int)std::min<long long>(std::max<long long>(c, 0), N);
}
void rank_row_range(int N, int npe, int rank, int nranks, int &g0, int &g1)
{
  g0 = rank_cut(N, npe, rank, nranks); g1 = rank_cut(N, npe, rank + 1, nranks);
}

int build_rank_mesh(int rank, int nranks, int N, int E, int npe, const int *elements, const double *nodes0,
                    int n_presc, const int *presc_node, const int *presc_type, const double *presc_values,
                    RankMesh &out, std::string &err)
{
  if (nranks < 1 || rank < 0 || rank >= nranks) { err = "bad (rank, nranks)"; return FEAHIP_EINVAL; }
  for (long long i = 0; i < (long long)E * npe; ++i)
    if (elements[i] < 0 || elements[i] >= N) { err = "element refers to a node outside the mesh"; return FEAHIP_EINVAL; }
  std::vector<int> lib;                                        // library id of every caller node
  locality_numbering(N, E, npe, elements, nodes0, lib);        // (the identity where the mesh gives no basis for one)
  out.n_global = N; out.rank = rank; out.nranks = nranks; out.npe = npe;
  rank_row_range(N, npe, rank, nranks, out.lib0, out.lib1);
  const int g0 = out.lib0, g1 = out.lib1;
  std::vector<int> cuts((size_t)nranks + 1);
  for (int k = 0; k <= nranks; ++k) cuts[k] = rank_cut(N, npe, k, nranks);
  auto owner_of = [&](int libid) { return (int)(std::upper_bound(cuts.begin(), cuts.end(), libid) - cuts.begin()) - 1; };
  // local elements: every element with an owned node, in the caller's order
  out.elem_global.clear();
  std::vector<char> is_local((size_t)N, 0);
  for (int e = 0; e < E; ++e) {
    bool mine = false;
    for (int k = 0; k < npe; ++k) { const int l = lib[elements[(size_t)e * npe + k]]; if (l >= g0 && l < g1) { mine = true; break; } }
    if (!mine) continue;
    out.elem_global.push_back(e);
    for (int k = 0; k < npe; ++k) is_local[elements[(size_t)e * npe + k]] = 1;
  }
  // local nodes: owned in library order, then halo in library order
  std::vector<std::pair<int, int>> own, halo;                  // (library id, caller id)
  for (int a = 0; a < N; ++a) {
    const int l = lib[a];
    if (l >= g0 && l < g1) own.emplace_back(l, a);             // (an owned node no element touches keeps its diagonal row)
    else if (is_local[a]) halo.emplace_back(l, a);
  }
  std::sort(own.begin(), own.end()); std::sort(halo.begin(), halo.end());
  out.n_own = (int)own.size();
  const int nl = (int)(own.size() + halo.size());
  out.node_global.resize((size_t)nl); out.node_lib.resize((size_t)nl);
  std::vector<int> local_of((size_t)N, -1);
  for (int i = 0; i < nl; ++i) {
    const auto &p = i < out.n_own ? own[i] : halo[i - out.n_own];
    out.node_lib[i] = p.first; out.node_global[i] = p.second; local_of[p.second] = i;
  }
  out.elements.resize(out.elem_global.size() * (size_t)npe);
  for (size_t i = 0; i < out.elem_global.size(); ++i)
    for (int k = 0; k < npe; ++k) out.elements[i * npe + k] = local_of[elements[(size_t)out.elem_global[i] * npe + k]];
  out.nodes0.resize((size_t)nl * 3);
  for (int i = 0; i < nl; ++i)
    for (int j = 0; j < 3; ++j) out.nodes0[(size_t)i * 3 + j] = nodes0[(size_t)out.node_global[i] * 3 + j];
  out.presc_node.clear(); out.presc_type.clear(); out.presc_values.clear();
  for (int i = 0; i < n_presc; ++i) {
    if (presc_node[i] < 0 || presc_node[i] >= N) { err = "prescribed node id out of range"; return FEAHIP_EINVAL; }
    const int l = local_of[presc_node[i]];
    if (l < 0) continue;                                        // another rank's business
    out.presc_node.push_back(l); out.presc_type.push_back(presc_type[i]);
    for (int j = 0; j < 3; ++j) out.presc_values.push_back(presc_values[(size_t)i * 3 + j]);
  }
  // halo plan in local ids.  recv: my halo nodes by owner (ascending library id inside a peer); send: my owned nodes that
  // share an element with a node of that peer (the peer's halo nodes I own: the same set, seen from its elements)
  std::vector<std::vector<int>> send((size_t)nranks), recv((size_t)nranks);
  for (int i = out.n_own; i < nl; ++i) recv[(size_t)owner_of(out.node_lib[i])].push_back(i);
  {
    std::vector<std::vector<int>> tmp((size_t)nranks);
    const size_t ne = out.elem_global.size();
    for (size_t e = 0; e < ne; ++e) {
      const int *c = out.elements.data() + e * npe;
      int owners[16]; int no = 0;
      for (int k = 0; k < npe; ++k)
        if (c[k] >= out.n_own) {
          const int r = owner_of(out.node_lib[c[k]]);
          bool seen = false;
          for (int q = 0; q < no; ++q) seen = seen || owners[q] == r;
          if (!seen && no < 16) owners[no++] = r;
        }
      if (!no) continue;
      for (int k = 0; k < npe; ++k)
        if (c[k] < out.n_own)
          for (int q = 0; q < no; ++q) tmp[(size_t)owners[q]].push_back(c[k]);
    }
    for (int r = 0; r < nranks; ++r) {
      std::sort(tmp[r].begin(), tmp[r].end());
      tmp[r].erase(std::unique(tmp[r].begin(), tmp[r].end()), tmp[r].end());
      send[r] = tmp[r];                                         // local owned ids ascend with their library ids
    }
  }
  ShardPlan &pl = out.plan;
  pl = ShardPlan();
  pl.rank = rank; pl.nranks = nranks; pl.row0 = 0; pl.row1 = out.n_own;
  pl.send_off.push_back(0); pl.recv_off.push_back(0);
  for (int r = 0; r < nranks; ++r) {
    if (r == rank || (send[r].empty() && recv[r].empty())) continue;
    pl.peer.push_back(r);
    pl.send_idx.insert(pl.send_idx.end(), send[r].begin(), send[r].end());
    pl.recv_idx.insert(pl.recv_idx.end(), recv[r].begin(), recv[r].end());
    pl.send_off.push_back((int)pl.send_idx.size()); pl.recv_off.push_back((int)pl.recv_idx.size());
  }
  return FEAHIP_OK;
}
