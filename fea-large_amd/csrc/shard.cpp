// shard.cpp -- host-only plan of the row-sharded (multi-GPU) solve.
//
// The reference is one process and one thread; sharding is new (SURVEY.md
// 8e).  Nodes -- hence block rows of K and the entries of f, u, x -- are owned
// by exactly one rank, in contiguous ranges that end at "super" boundaries of
// the chunk partition (slabs across the bar's long axis with the generator's
// y-slowest numbering).  A rank assembles every element that touches its rows
// (ghost elements are recomputed, so assembly needs no exchange) and, for the
// SpMV of the linear solve and for the node update, needs the values at the
// halo nodes: the column nodes of its rows that another rank owns.  The block
// pattern is symmetric, so what rank q needs from rank r is exactly the set
// of r's nodes that are adjacent to q's nodes: both sides derive the same
// ascending lists without talking to each other.
#include "feahip_internal.h"
#include <algorithm>

void shard_row_range(const std::vector<int> &chunk, int rank, int nranks, int &row0, int &row1)
{
  const int nchunks = (int)chunk.size() - 1;
  const int nsuper = (nchunks + FEA_SUPER_CHUNKS - 1) / FEA_SUPER_CHUNKS;
  const int s0 = (int)((long long)nsuper * rank / nranks), s1 = (int)((long long)nsuper * (rank + 1) / nranks);
  row0 = chunk[std::min(nchunks, s0 * FEA_SUPER_CHUNKS)];
  row1 = chunk[std::min(nchunks, s1 * FEA_SUPER_CHUNKS)];
}

void build_shard_plan(const std::vector<int> &rowptr, const std::vector<int> &colidx,
                      const std::vector<int> &chunk, int rank, int nranks, ShardPlan &plan)
{
  plan.rank = rank; plan.nranks = nranks;
  std::vector<int> start((size_t)nranks + 1);
  for (int k = 0; k < nranks; ++k) { int a, b; shard_row_range(chunk, k, nranks, a, b); start[k] = a; start[k + 1] = b; }
  plan.row0 = start[rank]; plan.row1 = start[rank + 1];
  std::vector<std::vector<int>> send((size_t)nranks), recv((size_t)nranks);
  for (int i = plan.row0; i < plan.row1; ++i)
    for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) {
      const int j = colidx[q];
      if (j >= plan.row0 && j < plan.row1) continue;
      const int owner = (int)(std::upper_bound(start.begin(), start.end(), j) - start.begin()) - 1;
      send[owner].push_back(i);
      recv[owner].push_back(j);
    }
  plan.peer.clear(); plan.send_off.assign(1, 0); plan.recv_off.assign(1, 0);
  plan.send_idx.clear(); plan.recv_idx.clear();
  for (int k = 0; k < nranks; ++k) {
    if (send[k].empty() && recv[k].empty()) continue;
    auto uniq = [](std::vector<int> &v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); };
    uniq(send[k]); uniq(recv[k]);
    plan.peer.push_back(k);
    plan.send_idx.insert(plan.send_idx.end(), send[k].begin(), send[k].end());
    plan.recv_idx.insert(plan.recv_idx.end(), recv[k].begin(), recv[k].end());
    plan.send_off.push_back((int)plan.send_idx.size());
    plan.recv_off.push_back((int)plan.recv_idx.size());
  }
}
