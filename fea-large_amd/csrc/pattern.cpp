// pattern.cpp -- one-time host construction of the sparsity pattern and the
// node->element incidence maps from the element->node map.
//
// The reference grows its matrix dynamically inside sp_matrix_element_add
// (fea_solver.c:966,1055) and converts it to Yale form before every solve
// (:304).  The mesh topology never changes, so here the full symmetric
// block pattern is built once: block row a holds one 3x3 block per node that
// shares an element with node a, columns sorted ascending (which is the
// column order sp_matrix_yale has).
#include "feahip_internal.h"
#include <algorithm>
#include <thread>
#include <utility>

namespace {

template <class F>
void parallel_ranges(int n, F f)
{
  unsigned hw = std::thread::hardware_concurrency();
  int nt = (int)std::min<unsigned>(hw ? hw : 4, 32);
  if (n < 65536) nt = 1;
  if (nt <= 1) { f(0, n, 0); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t) {
    int lo = (int)((long long)n * t / nt), hi = (int)((long long)n * (t + 1) / nt);
    th.emplace_back([=] { f(lo, hi, t); });
  }
  for (auto &x : th) x.join();
}

}  // namespace

int build_host_pattern(int N, int E, int npe, const int *conn, HostPattern &hp,
                       std::string &err, int row_break)
{
  if (E >= (1 << 28)) { err = "too many elements for the packed incidence word"; return FEAHIP_EINVAL; }
  for (long long i = 0; i < (long long)E * npe; ++i)
    if (conn[i] < 0 || conn[i] >= N) {
      err = "element " + std::to_string(i / npe) + " refers to node " +
            std::to_string(conn[i]) + " outside [0," + std::to_string(N) + ")";
      return FEAHIP_EINVAL;
    }

  // node -> element incidence, elements ascending inside every node
  hp.incptr.assign((size_t)N + 1, 0);
  for (long long i = 0; i < (long long)E * npe; ++i) hp.incptr[conn[i] + 1]++;
  for (int a = 0; a < N; ++a) hp.incptr[a + 1] += hp.incptr[a];
  hp.inc.resize((size_t)E * npe);
  {
    std::vector<int> fill(hp.incptr.begin(), hp.incptr.end() - 1);
    for (int e = 0; e < E; ++e)
      for (int k = 0; k < npe; ++k) {
        int a = conn[(size_t)e * npe + k];
        hp.inc[fill[a]++] = (uint32_t)e | ((uint32_t)k << 28);
      }
  }

  // neighbour sets -> block rows
  hp.rowptr.assign((size_t)N + 1, 0);
  auto row_nodes = [&](int a, std::vector<int> &tmp) {
    tmp.clear();
    for (int p = hp.incptr[a]; p < hp.incptr[a + 1]; ++p) {
      int e = (int)(hp.inc[p] & 0x0FFFFFFFu);
      for (int k = 0; k < npe; ++k) tmp.push_back(conn[(size_t)e * npe + k]);
    }
    if (tmp.empty()) tmp.push_back(a);   // isolated node keeps its diagonal
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
  };
  parallel_ranges(N, [&](int lo, int hi, int) {
    std::vector<int> tmp;
    for (int a = lo; a < hi; ++a) { row_nodes(a, tmp); hp.rowptr[a + 1] = (int)tmp.size(); }
  });
  long long tot = 0;
  int maxlen = 0;
  for (int a = 0; a < N; ++a) {
    maxlen = std::max(maxlen, hp.rowptr[a + 1]);
    tot += hp.rowptr[a + 1];
    // block numbers are ints; VALUE indices (block x 9 + entry) are size_t everywhere they are formed.  (Until round 4
    // this refused 2^31 / 9 blocks -- 2^31 scalar non-zeros -- which one rank of eight of BASELINE configs[4] exceeds.)
    if (tot > 0x7FFFFF00LL) { err = "pattern exceeds 32-bit block indexing"; return FEAHIP_EINVAL; }
    hp.rowptr[a + 1] = (int)tot;
  }
  hp.max_rowlen = maxlen;
  hp.colidx.resize((size_t)tot);
  parallel_ranges(N, [&](int lo, int hi, int) {
    std::vector<int> tmp;
    for (int a = lo; a < hi; ++a) {
      row_nodes(a, tmp);
      std::copy(tmp.begin(), tmp.end(), hp.colidx.begin() + hp.rowptr[a]);
    }
  });

  // chunks of consecutive rows, each small enough for one wave's LDS tile
  hp.chunk.clear();
  hp.chunk.push_back(0);
  int rows = 0, blocks = 0;
  for (int a = 0; a < N; ++a) {
    int len = hp.rowptr[a + 1] - hp.rowptr[a];
    // row_break (a rank's sub-mesh: its first halo row): chunks, supers and assembly chunks break there, so that
    // "the rows this rank owns" is a whole number of each
    if (rows > 0 && (rows == FEA_CHUNK_ROWS || blocks + len > FEA_CHUNK_BLOCKS || a == row_break)) {
      hp.chunk.push_back(a);
      rows = 0; blocks = 0;
    }
    rows++; blocks += len;
  }
  hp.chunk.push_back(N);
  const int nchunks = (int)hp.chunk.size() - 1;
  hp.break_chunk = nchunks;
  if (row_break > 0 && row_break < N)
    hp.break_chunk = (int)(std::lower_bound(hp.chunk.begin(), hp.chunk.end(), row_break) - hp.chunk.begin());
  // supers: FEA_SUPER_CHUNKS chunks each, a new one at the break
  std::vector<std::pair<int, int>> supers;
  for (int part = 0; part < 2; ++part) {
    const int lo = part ? hp.break_chunk : 0, hi = part ? nchunks : hp.break_chunk;
    for (int s0 = lo; s0 < hi; s0 += FEA_SUPER_CHUNKS) supers.emplace_back(s0, std::min(hi, s0 + FEA_SUPER_CHUNKS));
  }
  hp.break_super = (int)supers.size();
  for (size_t k = 0; k < supers.size(); ++k)
    if (supers[k].first >= hp.break_chunk) { hp.break_super = (int)k; break; }

  // position of the diagonal block of every row
  hp.diag.resize((size_t)N);
  parallel_ranges(N, [&](int lo, int hi, int) {
    for (int a = lo; a < hi; ++a) {
      const int *cb = hp.colidx.data() + hp.rowptr[a];
      const int *ce = hp.colidx.data() + hp.rowptr[a + 1];
      hp.diag[a] = hp.rowptr[a] + (int)(std::lower_bound(cb, ce, a) - cb);
    }
  });

  // Assembly partition of the staged kernel: finer chunks (visits, nodes and
  // blocks bounded so a wave's LDS footprint stays ~10 KB), nested in supers of
  // FEA_SUPER_CHUNKS SpMV chunks so both partitions share the shard boundaries.
  hp.inc_rows = hp.inc;
  hp.achunk.clear(); hp.super_achunk.clear();
  if (npe == 4) {
    std::vector<int> stamp((size_t)N, -1);
    int serial = 0;
    for (const auto &sp : supers) {
      const int s0 = sp.first, s1 = sp.second;
      hp.super_achunk.push_back((int)hp.achunk.size());
      const int ra = hp.chunk[s0], rb = hp.chunk[s1];
      int r0 = ra;
      while (r0 < rb) {
        int r = r0, nblk = 0, nvis = 0, nnod = 0;
        ++serial;
        for (; r < rb && r - r0 < FEA_ACHUNK_ROWS; ++r) {
          const int len = hp.rowptr[r + 1] - hp.rowptr[r], vis = hp.incptr[r + 1] - hp.incptr[r];
          // nodes this row would add
          int add = 0;
          std::vector<int> fresh;
          for (int q = hp.incptr[r]; q < hp.incptr[r + 1]; ++q) {
            const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu);
            for (int k = 0; k < 4; ++k) {
              const int g = conn[(size_t)e * 4 + k];
              if (stamp[g] != serial) { stamp[g] = serial; fresh.push_back(g); ++add; }
            }
          }
          if (r > r0 && (nblk + len > FEA_ACHUNK_BLOCKS || nvis + vis > FEA_VISIT_MAX_VISITS ||
                         nnod + add > FEA_VISIT_MAX_NODES)) {
            for (int g : fresh) stamp[g] = -1;     // not taken
            break;
          }
          nblk += len; nvis += vis; nnod += add;
        }
        hp.achunk.push_back(r0);
        r0 = r;
      }
    }
    hp.super_achunk.push_back((int)hp.achunk.size());
    hp.achunk.push_back(N);
  } else if (npe <= 16) {
    // Assembly partition of the shared-state kernel (kernels_quad.hip): rows
    // while the K tile, the (row, element, column) pairs and the distinct
    // elements of the chunk fit one wave's registers and LDS.
    std::vector<int> stamp((size_t)E, -1), nstamp((size_t)N, -1);
    int serial = 0;
    bool ok = true;
    for (size_t sk = 0; sk < supers.size() && ok; ++sk) {
      const int s0 = supers[sk].first, s1 = supers[sk].second;
      hp.super_achunk.push_back((int)hp.achunk.size());
      const int ra = hp.chunk[s0], rb = hp.chunk[s1];
      int r0 = ra;
      while (r0 < rb) {
        int r = r0, nblk = 0, npair = 0, nel = 0, nnod = 0;
        ++serial;
        for (; r < rb && r - r0 < FEA_CHUNK_ROWS; ++r) {
          const int len = hp.rowptr[r + 1] - hp.rowptr[r], vis = hp.incptr[r + 1] - hp.incptr[r];
          std::vector<int> fresh, nfresh;
          for (int q = hp.incptr[r]; q < hp.incptr[r + 1]; ++q) {
            const int e = (int)(hp.inc_rows[q] & 0x0FFFFFFFu);
            if (stamp[e] != serial) {
              stamp[e] = serial; fresh.push_back(e);
              for (int k = 0; k < npe; ++k) {
                const int g = conn[(size_t)e * npe + k];
                if (nstamp[g] != serial) { nstamp[g] = serial; nfresh.push_back(g); }
              }
            }
          }
          // more pairs than one round of the kernel holds only when a single row has them
          const bool over = nblk + len > FEA_QUAD_BLOCKS || nel + (int)fresh.size() > FEA_QUAD_ELEMS ||
                            nnod + (int)nfresh.size() > FEA_QUAD_NODES ||
                            (r > r0 && npair + vis * (npe - 1) > FEA_QUAD_PAIRS);
          if (over) {
            if (r == r0) ok = false;               // a single row does not fit: no shared-state assembly for this mesh
            for (int e : fresh) stamp[e] = -1;     // not taken
            for (int g : nfresh) nstamp[g] = -1;
            break;
          }
          nblk += len; npair += vis * (npe - 1); nel += (int)fresh.size(); nnod += (int)nfresh.size();
        }
        if (!ok) break;
        hp.achunk.push_back(r0);
        r0 = r;
      }
    }
    if (ok) {
      hp.super_achunk.push_back((int)hp.achunk.size());
      hp.achunk.push_back(N);
    } else {
      hp.achunk.clear(); hp.super_achunk.clear();
    }
  }

  // Inside a chunk the (row, element) visits are dealt round-robin over the
  // rows: the 64 lanes of one pass then work on as many different rows as the
  // chunk has, which keeps LDS adds to one address few (same-address
  // ds_add_f64 serialise, ~11 clk per extra lane on gfx950).
  parallel_ranges(nchunks, [&](int lo, int hi, int) {
    std::vector<uint32_t> tmp;
    for (int ch = lo; ch < hi; ++ch) {
      const int r0 = hp.chunk[ch], r1 = hp.chunk[ch + 1];
      const int p0 = hp.incptr[r0], p1 = hp.incptr[r1];
      tmp.assign(hp.inc.begin() + p0, hp.inc.begin() + p1);
      int out = p0, maxlen = 0;
      for (int r = r0; r < r1; ++r) maxlen = std::max(maxlen, hp.incptr[r + 1] - hp.incptr[r]);
      for (int k = 0; k < maxlen; ++k)
        for (int r = r0; r < r1; ++r)
          if (k < hp.incptr[r + 1] - hp.incptr[r]) hp.inc[out++] = tmp[hp.incptr[r] - p0 + k];
    }
  });

  // slot of every (visit, local column node) inside the visit's block row
  if (maxlen <= 255) {
    hp.incslot.resize((size_t)E * npe * npe);
    parallel_ranges((int)((long long)E * npe > 0x7FFFFFFF ? 0x7FFFFFFF : (long long)E * npe), [&](int lo, int hi, int) {
      for (int p = lo; p < hi; ++p) {
        const int e = (int)(hp.inc[p] & 0x0FFFFFFFu), la = (int)(hp.inc[p] >> 28);
        const int a = conn[(size_t)e * npe + la];
        const int *cb = hp.colidx.data() + hp.rowptr[a];
        const int *ce = hp.colidx.data() + hp.rowptr[a + 1];
        for (int k = 0; k < npe; ++k) {
          // 4-node elements are visited with local node k XOR la in position k
          // (row node first, kernels_assemble.hip); slots are stored in that order
          int b = conn[(size_t)e * npe + (npe == 4 ? (k ^ la) : k)];
          hp.incslot[(size_t)p * npe + k] = (uint8_t)(std::lower_bound(cb, ce, b) - cb);
        }
      }
    });
  } else {
    hp.incslot.clear();   // row-owner assembly unavailable; atomic path only
  }
  return FEAHIP_OK;
}
