// amg_setup.cpp -- topology and geometry of the aggregation hierarchy, host, once.
#include "amg.h"
#include <algorithm>
#include <cstdlib>

namespace {

// Greedy aggregation of a graph (Vanek et al.): a vertex all of whose
// neighbours are still free seeds an aggregate made of itself and them; the
// rest join the smallest neighbouring aggregate.
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

void finish_pattern(HostAmgLevel &L)
{
  const int N = L.N;
  L.diag.resize((size_t)N);
  for (int a = 0; a < N; ++a) {
    const int *cb = L.colidx.data() + L.rowptr[a], *ce = L.colidx.data() + L.rowptr[a + 1];
    L.diag[a] = L.rowptr[a] + (int)(std::lower_bound(cb, ce, a) - cb);
  }
  make_chunks(L.rowptr, N, L.chunk);
}

}  // namespace

// Level 0: one block row per mesh node ("site"), all of translation type.
// Level l >= 1: two block rows per site (aggregate of the level above): its
// translation and its rotation about the aggregate's centroid.
//
// A rank of a sharded solve builds the hierarchy of ITS diagonal block: nodes
// [own0, own1) and the couplings among them.  Level 0 keeps the global
// numbering of the context's arrays (nodes outside the range have agg = -1 and
// appear in no list); everything below is local to the rank, so the
// preconditioner -- block-Jacobi over the ranks, a W-cycle inside each --
// needs no communication.
static bool build_levels(const std::vector<int> &rowptr0, const std::vector<int> &colidx0, const std::vector<double> &pos0,
                         std::vector<HostAmgLevel> &out);

bool build_host_amg(const std::vector<int> &rowptr, const std::vector<int> &colidx, const std::vector<double> &pos,
                    int own0, int own1, std::vector<HostAmgLevel> &out)
{
  const int N = (int)rowptr.size() - 1, nloc = own1 - own0;
  if (own0 == 0 && own1 == N) return build_levels(rowptr, colidx, pos, out);
  // the rank's diagonal block as a graph of its own
  std::vector<int> lrow((size_t)nloc + 1, 0), lcol, lq;
  for (int i = own0; i < own1; ++i) {
    for (int q = rowptr[i]; q < rowptr[i + 1]; ++q)
      if (colidx[q] >= own0 && colidx[q] < own1) { lcol.push_back(colidx[q] - own0); lq.push_back(q); }
    lrow[i - own0 + 1] = (int)lcol.size();
  }
  std::vector<double> lpos(pos.begin() + (size_t)own0 * 3, pos.begin() + (size_t)own1 * 3);
  if (!build_levels(lrow, lcol, lpos, out)) return false;
  // level 0 back to the numbering of the context's arrays
  HostAmgLevel &L = out[0];
  std::vector<int> agg((size_t)N, -1), cbrow((size_t)rowptr[N], -1);
  std::vector<double> doff((size_t)N * 3, 0.0);
  for (int i = 0; i < nloc; ++i) {
    agg[own0 + i] = L.agg[i];
    for (int d = 0; d < 3; ++d) doff[(size_t)(own0 + i) * 3 + d] = L.doff[(size_t)i * 3 + d];
  }
  for (size_t k = 0; k < L.cbrow.size(); ++k) cbrow[lq[k]] = L.cbrow[k] + own0;
  for (int &v : L.anodes) v += own0;
  for (int &v : L.cblist) v = lq[v];
  L.agg.swap(agg); L.doff.swap(doff); L.cbrow.swap(cbrow);
  L.N = N; L.S = N;                      // the device loops over the context's rows; rows without an aggregate are skipped
  return true;
}

static bool build_levels(const std::vector<int> &rowptr0, const std::vector<int> &colidx0, const std::vector<double> &pos0,
                         std::vector<HostAmgLevel> &out)
{
  out.clear();
  // block rows of the level the hierarchy stops at (FEAHIP_AMG_COARSEST: tuning).  The W-cycle's over-correction counts on
  // inexact coarse solves, and what serves it best is depth, not sweeps: 10M-tet block, stop at 1 500 rows with 12
  // Jacobi sweeps there (4 levels) 95 CG iterations / 0.243 s per Newton iteration, 48 sweeps 96 iterations, stop at 200
  // rows (5 levels) with 12 sweeps 77 / 0.232 s, with 2 sweeps 79 / 0.196 s
  int coarsest_rows = 200;
  if (const char *e = getenv("FEAHIP_AMG_COARSEST")) coarsest_rows = std::max(2, atoi(e));
  HostAmgLevel L;
  L.N = (int)rowptr0.size() - 1;
  L.S = L.N;
  L.rowptr = rowptr0; L.colidx = colidx0;
  L.pos = pos0;
  // site graph of the current level (level 0: the block pattern itself)
  std::vector<int> sg_rowptr = rowptr0, sg_colidx = colidx0;
  for (;;) {
    const int N = L.N, S = L.S;
    const bool paired = (N == 2 * S);                                      // false only on level 0
    finish_pattern(L);
    if (N <= coarsest_rows || out.size() >= 6) { L.Sc = 0; out.push_back(L); break; }
    int nagg = 0;
    std::vector<int> sagg;
    aggregate(S, sg_rowptr, sg_colidx, sagg, nagg);
    if (nagg * 2 > S) { L.Sc = 0; out.push_back(L); break; }               // no longer coarsening
    // coarse site graph
    std::vector<int> c_rowptr((size_t)nagg + 1, 0), c_colidx;
    {
      std::vector<int> sptr((size_t)nagg + 1, 0), slist((size_t)S);
      for (int s = 0; s < S; ++s) sptr[sagg[s] + 1]++;
      for (int a = 0; a < nagg; ++a) sptr[a + 1] += sptr[a];
      { std::vector<int> fill(sptr.begin(), sptr.end() - 1); for (int s = 0; s < S; ++s) slist[fill[sagg[s]]++] = s; }
      std::vector<int> tmp;
      std::vector<std::vector<int>> rows((size_t)nagg);
      for (int I = 0; I < nagg; ++I) {
        tmp.clear();
        for (int p = sptr[I]; p < sptr[I + 1]; ++p) {
          const int s = slist[p];
          for (int q = sg_rowptr[s]; q < sg_rowptr[s + 1]; ++q) tmp.push_back(sagg[sg_colidx[q]]);
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        rows[I] = tmp;
        c_rowptr[I + 1] = c_rowptr[I] + (int)tmp.size();
      }
      c_colidx.resize((size_t)c_rowptr[nagg]);
      for (int I = 0; I < nagg; ++I) std::copy(rows[I].begin(), rows[I].end(), c_colidx.begin() + c_rowptr[I]);
    }
    bool too_wide = false;
    for (int I = 0; I < nagg; ++I) if (2 * (c_rowptr[I + 1] - c_rowptr[I]) > FEA_CHUNK_BLOCKS) too_wide = true;
    if (too_wide) { L.Sc = 0; out.push_back(L); break; }                    // keep this level as the coarsest
    L.Sc = nagg;
    // centroids, offsets
    std::vector<double> cpos((size_t)nagg * 3, 0.0);
    {
      std::vector<int> cnt((size_t)nagg, 0);
      for (int s = 0; s < S; ++s) { cnt[sagg[s]]++; for (int d = 0; d < 3; ++d) cpos[(size_t)sagg[s] * 3 + d] += L.pos[(size_t)s * 3 + d]; }
      for (int a = 0; a < nagg; ++a) for (int d = 0; d < 3; ++d) cpos[(size_t)a * 3 + d] /= (double)cnt[a];
    }
    auto site_of = [&](int i) { return paired ? i / 2 : i; };
    L.agg.resize((size_t)N); L.doff.resize((size_t)N * 3);
    for (int i = 0; i < N; ++i) {
      const int s = site_of(i), a = sagg[s];
      L.agg[i] = a;
      for (int d = 0; d < 3; ++d) L.doff[(size_t)i * 3 + d] = L.pos[(size_t)s * 3 + d] - cpos[(size_t)a * 3 + d];
    }
    L.aptr.assign((size_t)nagg + 1, 0);
    for (int i = 0; i < N; ++i) L.aptr[L.agg[i] + 1]++;
    for (int a = 0; a < nagg; ++a) L.aptr[a + 1] += L.aptr[a];
    L.anodes.resize((size_t)N);
    { std::vector<int> fill(L.aptr.begin(), L.aptr.end() - 1); for (int i = 0; i < N; ++i) L.anodes[fill[L.agg[i]]++] = i; }
    // fine blocks -> coarse site pair
    const int nnzb = L.rowptr[N], npair = c_rowptr[nagg];
    std::vector<int> cmap((size_t)nnzb);
    L.cbrow.resize((size_t)nnzb);
    for (int i = 0; i < N; ++i) {
      const int I = L.agg[i];
      const int *cb = c_colidx.data() + c_rowptr[I], *ce = c_colidx.data() + c_rowptr[I + 1];
      for (int q = L.rowptr[i]; q < L.rowptr[i + 1]; ++q) {
        cmap[q] = c_rowptr[I] + (int)(std::lower_bound(cb, ce, L.agg[L.colidx[q]]) - cb);
        L.cbrow[q] = i;
      }
    }
    L.cbptr.assign((size_t)npair + 1, 0);
    for (int q = 0; q < nnzb; ++q) L.cbptr[cmap[q] + 1]++;
    for (int k = 0; k < npair; ++k) L.cbptr[k + 1] += L.cbptr[k];
    L.cblist.resize((size_t)nnzb);
    { std::vector<int> fill(L.cbptr.begin(), L.cbptr.end() - 1); for (int q = 0; q < nnzb; ++q) L.cblist[fill[cmap[q]]++] = q; }
    // coarse level: two block rows per site, (translation, rotation)
    HostAmgLevel C;
    C.S = nagg; C.N = 2 * nagg;
    C.pos = cpos;
    C.type.resize((size_t)C.N);
    C.rowptr.assign((size_t)C.N + 1, 0);
    for (int I = 0; I < nagg; ++I) {
      const int len = c_rowptr[I + 1] - c_rowptr[I];
      C.type[2 * I] = 0; C.type[2 * I + 1] = 1;
      C.rowptr[2 * I + 1] = C.rowptr[2 * I] + 2 * len;
      C.rowptr[2 * I + 2] = C.rowptr[2 * I + 1] + 2 * len;
    }
    C.colidx.resize((size_t)C.rowptr[C.N]);
    L.prow.resize((size_t)npair);
    for (int I = 0; I < nagg; ++I) for (int t = c_rowptr[I]; t < c_rowptr[I + 1]; ++t) L.prow[t] = I;
    for (int I = 0; I < nagg; ++I)
      for (int sg = 0; sg < 2; ++sg) {
        const int base = C.rowptr[2 * I + sg];
        for (int t = 0; t < c_rowptr[I + 1] - c_rowptr[I]; ++t)
          for (int rho = 0; rho < 2; ++rho) {
            const int k = base + 2 * t + rho;
            C.colidx[k] = 2 * c_colidx[c_rowptr[I] + t] + rho;
          }
      }
    out.push_back(L);
    L = C;
    sg_rowptr.swap(c_rowptr); sg_colidx.swap(c_colidx);
  }
  return out.size() >= 2;
}
