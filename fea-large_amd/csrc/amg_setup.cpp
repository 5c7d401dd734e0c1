// amg_setup.cpp -- topology of the aggregation hierarchy, host, once.
#include "amg.h"
#include <algorithm>

namespace {

// Greedy aggregation of the node graph (Vanek et al.): a node all of whose
// neighbours are still free seeds an aggregate made of itself and them; the
// rest join a neighbouring aggregate.
void aggregate(int N, const std::vector<int> &rowptr, const std::vector<int> &colidx, std::vector<int> &agg, int &nagg)
{
  agg.assign((size_t)N, -1);
  nagg = 0;
  for (int i = 0; i < N; ++i) {
    if (agg[i] >= 0) continue;
    bool free_nb = true;
    for (int q = rowptr[i]; q < rowptr[i + 1] && free_nb; ++q) free_nb = agg[colidx[q]] < 0;
    if (!free_nb) continue;
    for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) agg[colidx[q]] = nagg;
    agg[i] = nagg++;
  }
  std::vector<int> size((size_t)nagg, 0);
  for (int i = 0; i < N; ++i) if (agg[i] >= 0) size[agg[i]]++;
  std::vector<int> join((size_t)N, -1);
  for (int i = 0; i < N; ++i) {
    if (agg[i] >= 0) continue;
    int best = -1;
    for (int q = rowptr[i]; q < rowptr[i + 1]; ++q) {
      const int a = agg[colidx[q]];
      if (a >= 0 && (best < 0 || size[a] < size[best])) best = a;
    }
    join[i] = best;
    if (best >= 0) size[best]++;
  }
  for (int i = 0; i < N; ++i) {
    if (agg[i] >= 0) continue;
    if (join[i] >= 0) agg[i] = join[i];
    else agg[i] = nagg++;                 // isolated
  }
}

void make_chunks(const std::vector<int> &rowptr, int N, std::vector<int> &chunk)
{
  chunk.clear();
  chunk.push_back(0);
  int rows = 0, blocks = 0;
  for (int a = 0; a < N; ++a) {
    const int len = rowptr[a + 1] - rowptr[a];
    if (rows > 0 && (rows == FEA_CHUNK_ROWS || blocks + len > FEA_CHUNK_BLOCKS)) { chunk.push_back(a); rows = 0; blocks = 0; }
    rows++; blocks += len;
  }
  chunk.push_back(N);
}

}  // namespace

bool build_host_amg(const std::vector<int> &rowptr0, const std::vector<int> &colidx0, std::vector<HostAmgLevel> &out)
{
  out.clear();
  HostAmgLevel L;
  L.N = (int)rowptr0.size() - 1;
  L.rowptr = rowptr0; L.colidx = colidx0;
  for (;;) {
    const int N = L.N;
    for (int a = 0; a < N; ++a)
      if (L.rowptr[a + 1] - L.rowptr[a] > FEA_CHUNK_BLOCKS) return false;     // SpMV chunk limit
    L.diag.resize((size_t)N);
    for (int a = 0; a < N; ++a) {
      const int *cb = L.colidx.data() + L.rowptr[a], *ce = L.colidx.data() + L.rowptr[a + 1];
      L.diag[a] = L.rowptr[a] + (int)(std::lower_bound(cb, ce, a) - cb);
    }
    make_chunks(L.rowptr, N, L.chunk);
    if (N <= 1500 || out.size() >= 6) { L.Nc = 0; out.push_back(L); break; }
    int nagg = 0;
    aggregate(N, L.rowptr, L.colidx, L.agg, nagg);
    if (nagg * 2 > N) { L.Nc = 0; L.agg.clear(); out.push_back(L); break; }     // no longer coarsening
    L.Nc = nagg;
    // aggregate -> nodes
    L.aptr.assign((size_t)nagg + 1, 0);
    for (int i = 0; i < N; ++i) L.aptr[L.agg[i] + 1]++;
    for (int a = 0; a < nagg; ++a) L.aptr[a + 1] += L.aptr[a];
    L.anodes.resize((size_t)N);
    { std::vector<int> fill(L.aptr.begin(), L.aptr.end() - 1); for (int i = 0; i < N; ++i) L.anodes[fill[L.agg[i]]++] = i; }
    // coarse pattern
    HostAmgLevel C;
    C.N = nagg;
    C.rowptr.assign((size_t)nagg + 1, 0);
    std::vector<int> tmp;
    std::vector<std::vector<int>> rows((size_t)nagg);
    for (int I = 0; I < nagg; ++I) {
      tmp.clear();
      for (int p = L.aptr[I]; p < L.aptr[I + 1]; ++p) {
        const int i = L.anodes[p];
        for (int q = L.rowptr[i]; q < L.rowptr[i + 1]; ++q) tmp.push_back(L.agg[L.colidx[q]]);
      }
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      rows[I] = tmp;
      C.rowptr[I + 1] = C.rowptr[I] + (int)tmp.size();
    }
    C.colidx.resize((size_t)C.rowptr[nagg]);
    for (int I = 0; I < nagg; ++I) std::copy(rows[I].begin(), rows[I].end(), C.colidx.begin() + C.rowptr[I]);
    // which fine blocks sum into which coarse block
    const int nnzb = L.rowptr[N], nnzc = C.rowptr[nagg];
    std::vector<int> cmap((size_t)nnzb);
    L.cbrow.resize((size_t)nnzb);
    for (int i = 0; i < N; ++i) {
      const int I = L.agg[i];
      const int *cb = C.colidx.data() + C.rowptr[I], *ce = C.colidx.data() + C.rowptr[I + 1];
      for (int q = L.rowptr[i]; q < L.rowptr[i + 1]; ++q) {
        cmap[q] = C.rowptr[I] + (int)(std::lower_bound(cb, ce, L.agg[L.colidx[q]]) - cb);
        L.cbrow[q] = i;
      }
    }
    L.cbptr.assign((size_t)nnzc + 1, 0);
    for (int q = 0; q < nnzb; ++q) L.cbptr[cmap[q] + 1]++;
    for (int k = 0; k < nnzc; ++k) L.cbptr[k + 1] += L.cbptr[k];
    L.cblist.resize((size_t)nnzb);
    { std::vector<int> fill(L.cbptr.begin(), L.cbptr.end() - 1); for (int q = 0; q < nnzb; ++q) L.cblist[fill[cmap[q]]++] = q; }
    out.push_back(L);
    L = C;
  }
  return out.size() >= 2;
}
