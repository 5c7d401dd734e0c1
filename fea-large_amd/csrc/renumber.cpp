// renumber.cpp -- library-side locality numbering of the nodes (host, once per context).
//
// The reference keeps the nodes in deck order (sexp_loader.c:170-215 stores them as they come; TetGen or
// lexicographic ids) and its matrix rows follow (node * 3 + axis, fea_solver.c:377-384).  The assembly kernels here
// own runs of CONSECUTIVE block rows and evaluate every element that touches them, so what a run of consecutive ids
// looks like in space decides how often an element is evaluated: 1.75 times when 64 consecutive ids are a compact
// 4 x 4 x 4 cluster, 2.9 times when they are lines of a lexicographic numbering.  feahip_create therefore numbers
// the nodes itself and runs everything in that numbering; every entry of the ABI that takes or returns node-indexed
// data (coordinates, forces, solution, prescribed node ids, the Yale matrix, SpMV vectors) translates, so the caller
// only ever sees its own indexing (bit-exact connectivity / dof indexing: node * 3 + axis of the CALLER's node).
//
// The numbering: cells of bx x by x bz node spacings laid over the bounding box (spacing per axis = median of the
// non-zero coordinate differences between the nodes of an element: exact on a lattice; cell counts checked and the
// cells rescaled when the mesh is not one); cells in lexicographic order with the longest axis of the box slowest
// (a contiguous id range stays a slab across the long axis: what the row shard cuts); nodes inside a cell in the same
// order of their sub-positions, ties by original id.  On a structured block (exact, or with noise: origin and spacing
// are fitted to the nodes) this reproduces a brick numbering exactly.  A mesh whose nodes sit on no lattice on any axis
// (a TetGen deck) is numbered by recursive coordinate bisection into leaves of the chunk length instead (rcb_order
// below; FEAHIP_NUMBERING_RCB=0 keeps the cells, which are a bucket sort into boxes of ~bx*by*bz nodes there).
// Deterministic in (coordinates, connectivity).
#include "feahip_internal.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace {
double median_inplace(std::vector<double> &v)
{
  if (v.empty()) return 0.0;
  const size_t m = v.size() / 2;
  std::nth_element(v.begin(), v.begin() + m, v.end());
  return v[m];
}

// Unstructured meshes: recursive coordinate bisection into leaves of exactly `leaf` nodes.  The bucket sort below gives a
// lattice its bricks; on a TetGen mesh its cells hold anything between a few and a hundred nodes, a run of 64
// consecutive ids straddles two or three of them, and the chunks the gather maps cut out of it (gather.cpp, pass A)
// end at ~50 rows with 1.90 evaluations per element.  Here every range of nodes is split across the longest side of
// ITS bounding box at a multiple of the leaf size, so every leaf but the last of a range is full and as close to a cube
// as the nodes allow; the leaves follow one another in the order of the tree (neighbours in space stay neighbours in
// L2), and the first cuts of a slender body are across its long axis: id ranges are slabs, as the row shard wants.
// Inside a leaf the bisection goes on by halves down to single nodes (the first 32, 48, 56 ids of a leaf are compact
// too).  Deterministic and independent of the caller's ids (ties by the other coordinates; only coincident points fall back
// to the id); the threads work on disjoint ranges.
struct RcbRange { int lo, hi; };
int rcb_split(const double *X, int *idx, RcbRange r, int leaf)
{
  const int m = r.hi - r.lo;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = r.lo; i < r.hi; ++i)
    for (int k = 0; k < 3; ++k) { const double x = X[(size_t)idx[i] * 3 + k]; lo[k] = std::min(lo[k], x); hi[k] = std::max(hi[k], x); }
  int ax = 1;                                                  // ties: y, z, x -- the order of the reference's bar
  if (hi[2] - lo[2] > (hi[ax] - lo[ax]) * (1.0 + 1e-9)) ax = 2;
  if (hi[0] - lo[0] > (hi[ax] - lo[ax]) * (1.0 + 1e-9)) ax = 0;
  const int left = m > leaf ? ((m + leaf - 1) / leaf / 2) * leaf : m / 2;
  const int ax1 = (ax + 1) % 3, ax2 = (ax + 2) % 3;            // ties (the nodes of a flat face): by the other coordinates, then by id
  std::nth_element(idx + r.lo, idx + r.lo + left, idx + r.hi, [&](int a, int b) {
    const double *pa = X + (size_t)a * 3, *pb = X + (size_t)b * 3;
    if (pa[ax] != pb[ax]) return pa[ax] < pb[ax];
    if (pa[ax1] != pb[ax1]) return pa[ax1] < pb[ax1];
    if (pa[ax2] != pb[ax2]) return pa[ax2] < pb[ax2];
    return a < b;
  });
  return left;
}
void rcb_order(const double *X, int *idx, int n, int leaf)
{
  unsigned hw = std::thread::hardware_concurrency();
  const int nt = n < (1 << 16) ? 1 : (int)std::min<unsigned>(hw ? hw : 4, 32);
  std::vector<RcbRange> todo(1, RcbRange{0, n});
  while (nt > 1 && (int)todo.size() < 8 * nt) {                // the top of the tree, level by level
    std::vector<RcbRange> next;
    bool any = false;
    for (const RcbRange &r : todo) {
      if (r.hi - r.lo <= 4 * leaf) { next.push_back(r); continue; }
      const int left = rcb_split(X, idx, r, leaf);
      next.push_back({r.lo, r.lo + left}); next.push_back({r.lo + left, r.hi});
      any = true;
    }
    todo.swap(next);
    if (!any) break;
  }
  auto finish = [&](RcbRange top) {
    std::vector<RcbRange> stack(1, top);
    while (!stack.empty()) {
      const RcbRange r = stack.back();
      stack.pop_back();
      if (r.hi - r.lo <= 1) continue;
      const int left = rcb_split(X, idx, r, leaf);
      stack.push_back({r.lo + left, r.hi});
      stack.push_back({r.lo, r.lo + left});
    }
  };
  if (nt <= 1) { for (const RcbRange &r : todo) finish(r); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t] { for (size_t i = (size_t)t; i < todo.size(); i += (size_t)nt) finish(todo[i]); });
  for (auto &x : th) x.join();
}

// The leaf size for 4-node tetrahedra.  A chunk is bounded in ELEMENTS as well as rows (the records of the elements around
// its rows fill the LDS: FEA_G_MAX_ELEMS), and where a mesh is dense (the inside of the reference's TetGen deck: ~750
// elements around 64 nodes, ~450 at its faces) a leaf of 64 does not fit: pass A (gather.cpp) cuts it into 52 + 12
// rows, and the 10M mesh ends with 53 rows per chunk.  A leaf a few rows SHORT of the row limit leaves pass A room to
// move a boundary: a leaf that is too dense hands its last rows to its neighbour.  How short is decided by pass A
// itself: after one bisection with leaves of 64, two blocks of 8 192 consecutive ids (compact regions: unions of
// subtrees, at one and two thirds of the id range) are bisected again with every candidate size, cut out as
// sub-meshes (their nodes first, the nodes of the elements around them behind) and handed to the pattern and gather
// builders; the size with the fewest chunks over both blocks wins, ties to the fewer element evaluations.  On the
// TetGen deck's corner tetrahedra, 10M elements (gpurun_out/r4_h6): leaves of 60 give 34 325 chunks and 0.851 ms,
// of 62 33 217 and 0.835, of 63 32 690 and 0.830; the cells gave 41 393 and 0.950.  Measured and dropped on the way:
// leaves bounded by the element incidences of their nodes (a weight limit in the bisection: the incidences predict
// the distinct elements too loosely); the prefixes of the 64-leaves as a model of shorter leaves (a prefix is half +
// quarter + ... of a leaf, a slab, not a cube); a greedy walk as a model of pass A (it cuts 60.7 rows whatever the
// leaf); the share of leaves whose elements do not fit against the rows of room (it picks 60 where 63 is better).
int rcb_pick_leaf(int N, int E, const int *conn, const double *X, const int *order, int leaf0)
{
  const int block = 8192;
  if (N < 4 * block) return leaf0;                             // small meshes: the row limit itself
  std::vector<int> incptr((size_t)N + 1, 0);
  for (size_t q = 0; q < (size_t)E * 4; ++q) ++incptr[(size_t)conn[q] + 1];
  for (int a = 0; a < N; ++a) incptr[(size_t)a + 1] += incptr[a];
  std::vector<int> inc((size_t)E * 4), fill(incptr.begin(), incptr.end() - 1);
  for (int e = 0; e < E; ++e)
    for (int k = 0; k < 4; ++k) inc[(size_t)fill[conn[(size_t)e * 4 + k]]++] = e;
  std::vector<int>().swap(fill);
  const int cand[] = {64, 63, 62, 61, 60, 58, 56, 52, 48};
  const int ncand = (int)(sizeof(cand) / sizeof(cand[0]));
  std::vector<long long> chunks((size_t)ncand, 0), evals((size_t)ncand, 0);
  std::vector<char> failed((size_t)ncand, 0);
  std::vector<std::thread> th;
  const int nt = 3;                                            // each keeps an [N] and an [E] scratch array
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      std::vector<int> ids((size_t)block), local((size_t)N, -1), estamp((size_t)E, -1), lconn, touched;
      for (int ci = t; ci < ncand; ci += nt) {
        const int L = cand[ci];
        if (L > leaf0) { failed[ci] = 1; continue; }
        for (int b = 1; b <= 2 && !failed[ci]; ++b) {
          const int start = (int)((long long)(N - block) * b / 3), stamp = ci * 4 + b;
          for (int i = 0; i < block; ++i) ids[i] = order[(size_t)start + i];
          std::sort(ids.begin(), ids.end());                    // the result must not depend on the order the block came in
          rcb_order(X, ids.data(), block, L);
          // the block as a sub-mesh: its nodes first in the new order, the other nodes of the elements around them behind
          touched.clear(); lconn.clear();
          int nloc = block;
          for (int i = 0; i < block; ++i) { local[ids[i]] = i; touched.push_back(ids[i]); }
          for (int i = 0; i < block; ++i)
            for (int q = incptr[ids[i]]; q < incptr[ids[i] + 1]; ++q) {
              const int e = inc[q];
              if (estamp[e] == stamp) continue;
              estamp[e] = stamp;
              for (int k = 0; k < 4; ++k) {
                const int g = conn[(size_t)e * 4 + k];
                if (local[g] < 0) { local[g] = nloc++; touched.push_back(g); }
                lconn.push_back(local[g]);
              }
            }
          HostPattern hp;
          HostGather hg;
          std::string err;
          if (build_host_pattern(nloc, (int)(lconn.size() / 4), 4, lconn.data(), hp, err) == 0)
            build_host_gather(nloc, (int)(lconn.size() / 4), lconn.data(), hp, 0, block, hg);
          for (int g : touched) local[g] = -1;
          if (!hg.ok) { failed[ci] = 1; break; }
          chunks[ci] += hg.nchunks; evals[ci] += hg.total_evals;
        }
      }
    });
  for (auto &x : th) x.join();
  int best = -1;
  for (int ci = 0; ci < ncand; ++ci) {
    if (failed[ci]) continue;
    if (getenv("FEAHIP_NUMBERING_VERBOSE")) fprintf(stderr, "rcb leaf %d: %lld chunks, %lld evaluations on the sample blocks\n", cand[ci], chunks[ci], evals[ci]);
    if (best < 0 || chunks[ci] < chunks[best] || (chunks[ci] == chunks[best] && evals[ci] < evals[best])) best = ci;
  }
  return best < 0 ? leaf0 : cand[best];
}
}  // namespace

// new_of_old[N].  Returns false (identity left in place) when the mesh gives no basis for a numbering (degenerate
// box, fewer nodes than a few cells).
bool locality_numbering(int N, int E, int npe, const int *conn, const double *X /*[N][3]*/, std::vector<int> &new_of_old)
{
  new_of_old.resize((size_t)N);
  for (int a = 0; a < N; ++a) new_of_old[a] = a;
  // nodes of a cell along (fastest, slowest, middle) axis: the chunk shapes the gather kernels want
  // (64 rows for 4-node tetrahedra and 8-node bricks; 48 rows of the half-spacing grid for 10-node tetrahedra)
  int cell[3] = {FEA_G_CELL};
  if (npe == 10) { cell[0] = 3; cell[1] = 4; cell[2] = 4; }
  const int target = cell[0] * cell[1] * cell[2];
  if (N < 4 * target || E <= 0) return false;
  double lo[3] = {X[0], X[1], X[2]}, hi[3] = {X[0], X[1], X[2]};
  for (int a = 0; a < N; ++a)
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], X[(size_t)a * 3 + k]); hi[k] = std::max(hi[k], X[(size_t)a * 3 + k]); }
  double ext[3];
  for (int k = 0; k < 3; ++k) { ext[k] = hi[k] - lo[k]; if (!(ext[k] > 0.0) || !std::isfinite(ext[k])) return false; }
  // axis roles: slowest = longest extent, then the middle, fastest = shortest; ties keep (y, z, x) -- the order of
  // the reference's bar and of mesh.brick_numbering
  int ax[3] = {1, 2, 0};                                     // candidates in tie order: slowest first
  std::stable_sort(ax, ax + 3, [&](int a, int b) { return ext[a] > ext[b] * (1.0 + 1e-9); });
  const int a_slow = ax[0], a_mid = ax[1], a_fast = ax[2];
  // spacing per axis from a sample of the elements
  double h[3];
  std::vector<double> dsample[3];                            // non-zero coordinate differences between the nodes of an element, per axis
  {
    const int stride = std::max(1, E / 200000);
    std::vector<double> (&d)[3] = dsample;
    for (int e = 0; e < E; e += stride)
      for (int p = 0; p < npe; ++p)
        for (int q = p + 1; q < npe; ++q) {
          const int a = conn[(size_t)e * npe + p], b = conn[(size_t)e * npe + q];
          for (int k = 0; k < 3; ++k) {
            const double v = std::fabs(X[(size_t)a * 3 + k] - X[(size_t)b * 3 + k]);
            if (v > 1e-9 * ext[k]) d[k].push_back(v);
          }
        }
    for (int k = 0; k < 3; ++k) { std::vector<double> w(d[k]); h[k] = median_inplace(w); if (!(h[k] > 0.0)) return false; }
  }
  // A lattice with noise (a structured block whose nodes were moved by a fraction of the spacing).  The median above
  // is the spacing of an exact lattice only: noise turns the zero differences into small ones and drags it down (10M
  // block, +-0.2 h: 0.84 h, 0.78 h and 0.50 h on the three axes), cells of four such spacings drift against the node
  // planes, the chunks stop being bricks and stop repeating (1.93 element evaluations per element instead of 1.73,
  // 36 056 chunks with no two alike in a row instead of 28 682 with 82 % repeats, 0.80 ms instead of 0.65).  So origin
  // and spacing of every axis are FITTED to the nodes: from a few candidate spacings (the median; the median of the
  // upper cluster of the differences; half of it -- the mid-edge planes of 10-node elements) the plane index of every
  // node k_i = round((x_i - o) / h) and (o, h) by least squares of x_i on k_i, three rounds; a candidate counts when
  // the nodes DO sit on its planes (rms distance < 0.2 h: uniformly scattered nodes give 0.29) and its planes are IN
  // USE (nine in ten between the first and the last hold a node: a lattice fits any finer grid as well); the finest
  // that counts wins.  An exact lattice returns itself, a TetGen mesh keeps the median spacing and the box corner.
  double lat_o[3] = {lo[0], lo[1], lo[2]};
  int fitted = 0;                                             // axes whose nodes sit on the planes of a lattice
  for (int k = 0; k < 3; ++k) {
    double cand[3] = {h[k], 0.0, 0.0};
    {
      std::vector<double> &v = dsample[k];
      std::vector<double> w(v);
      const size_t i95 = (size_t)(0.95 * (double)(w.size() - 1));
      std::nth_element(w.begin(), w.begin() + (long)i95, w.end());
      const double p95 = w[i95];
      std::vector<double> up;
      for (double x : v) if (x > 0.5 * p95) up.push_back(x);
      if (!up.empty()) { cand[1] = median_inplace(up); cand[2] = 0.5 * cand[1]; }
    }
    double best_h = 0.0, best_o = 0.0;
    for (int ci = 0; ci < 3; ++ci) {
      double o = lo[k], hh = cand[ci];
      if (!(hh > 0.0)) continue;
      bool ok = true;
      for (int round = 0; round < 3 && ok; ++round) {
        double sk = 0, sx = 0, skk = 0, skx = 0;
        for (int a = 0; a < N; ++a) {
          const double x = X[(size_t)a * 3 + k], q = std::floor((x - o) / hh + 0.5);
          sk += q; sx += x; skk += q * q; skx += q * x;
        }
        const double den = (double)N * skk - sk * sk;
        if (!(den > 0.0)) { ok = false; break; }
        const double h2 = ((double)N * skx - sk * sx) / den, o2 = (sx - h2 * sk) / (double)N;
        if (!(h2 > 0.7 * hh && h2 < 1.4 * hh)) { ok = false; break; }
        hh = h2; o = o2;
      }
      if (!ok) continue;
      const long long q0 = (long long)std::floor((lo[k] - o) / hh + 0.5), q1 = (long long)std::floor((hi[k] - o) / hh + 0.5);
      if (q1 - q0 < 1 || q1 - q0 > (1 << 22)) continue;
      std::vector<char> used((size_t)(q1 - q0 + 1), 0);
      double ss = 0;
      for (int a = 0; a < N; ++a) {
        const double x = X[(size_t)a * 3 + k], r = (x - o) / hh, q = std::floor(r + 0.5);
        ss += (r - q) * (r - q);
        const long long qi = (long long)q - q0;
        if (qi >= 0 && qi <= q1 - q0) used[(size_t)qi] = 1;
      }
      long long nused = 0;
      for (char u : used) nused += u;
      if (std::sqrt(ss / (double)N) < 0.2 && 10 * nused >= 9 * (q1 - q0 + 1) && (best_h == 0.0 || hh < best_h)) { best_h = hh; best_o = o + hh * (double)q0; }
    }
    if (best_h > 0.0) { h[k] = best_h; lat_o[k] = best_o; ++fitted; }
  }
  if (fitted == 0) {                                          // no lattice on any axis: bisection instead of cells
    // 4-node tetrahedra by default (the corner tetrahedra of the reference's TetGen deck at 10M elements: 0.950 ->
    // 0.838 ms per assembly, gpurun_out/r4_h2); 10-node and 8-node elements only on request: their chunks are bounded
    // by elements (127), not rows, and the deck's own 10-node mesh came out even (3.03 -> 2.94 evaluations per
    // element, 31 964 -> 32 539 chunks, 2.345 -> 2.365 ms)
    const char *e = getenv("FEAHIP_NUMBERING_RCB");
    if (e ? atoi(e) != 0 : npe == 4) {
      std::vector<int> order((size_t)N);
      for (int a = 0; a < N; ++a) order[a] = a;
      // leaves: the rows of a chunk, fewer where 4-node elements are dense (rcb_pick_leaf)
      int leaf = FEA_G_MAX_ROWS;
      const char *w = getenv("FEAHIP_NUMBERING_RCB_LEAF");
      if (w) leaf = std::max(8, atoi(w));
      rcb_order(X, order.data(), N, leaf);
      if (npe == 4 && !w) {
        const int pick = rcb_pick_leaf(N, E, conn, X, order.data(), leaf);
        if (pick != leaf) {
          for (int a = 0; a < N; ++a) order[a] = a;
          rcb_order(X, order.data(), N, pick);
        }
      }
      for (int r = 0; r < N; ++r) new_of_old[order[r]] = r;
      return true;
    }
  }
  int cn[3];                                                  // nodes of a cell along x, y, z
  cn[a_fast] = cell[0]; cn[a_slow] = cell[1]; cn[a_mid] = cell[2];
  double scale = 1.0;
  std::vector<unsigned long long> key((size_t)N);
  for (int attempt = 0; attempt < 4; ++attempt) {
    double cs[3], org[3];
    long long nc[3];
    bool ok = true;
    for (int k = 0; k < 3; ++k) {
      cs[k] = h[k] * cn[k] * scale;                           // cell size
      org[k] = lat_o[k] - 0.5 * h[k] * scale;                 // node planes of a lattice sit inside the cells, not on their faces
      nc[k] = (long long)std::floor((hi[k] - org[k]) / cs[k]) + 1;
      if (nc[k] < 1 || nc[k] > (1 << 18)) ok = false;
    }
    if (!ok) return false;
    // key = (cell slow, cell mid, cell fast, sub slow, sub mid, sub fast): 18 bits per cell index, 3 bits per sub index
    for (int a = 0; a < N; ++a) {
      long long ci[3], si[3];
      for (int k = 0; k < 3; ++k) {
        const double r = (X[(size_t)a * 3 + k] - org[k]) / cs[k];
        ci[k] = std::min<long long>(std::max<long long>((long long)std::floor(r), 0), nc[k] - 1);
        const long long sub = (long long)std::floor((r - (double)ci[k]) * cn[k]);
        si[k] = std::min<long long>(std::max<long long>(sub, 0), cn[k] - 1);
      }
      key[a] = ((unsigned long long)ci[a_slow] << 45) | ((unsigned long long)ci[a_mid] << 27) | ((unsigned long long)ci[a_fast] << 9) |
               ((unsigned long long)si[a_slow] << 6) | ((unsigned long long)si[a_mid] << 3) | (unsigned long long)si[a_fast];
    }
    // occupancy: mean nodes per non-empty cell against the target; a lattice gives the target (boundary cells a little
    // less), an unstructured or graded mesh does not: rescale the cells (uniformly) and try again
    std::vector<unsigned long long> cells((size_t)N);
    for (int a = 0; a < N; ++a) cells[a] = key[a] >> 9;
    std::sort(cells.begin(), cells.end());
    const long long nonempty = (long long)(std::unique(cells.begin(), cells.end()) - cells.begin());
    const double mean = (double)N / (double)nonempty;
    if (mean >= 0.4 * target && mean <= 1.5 * target) break;     // (small lattices have many partial boundary cells)
    if (attempt == 3) break;
    scale *= std::cbrt((double)target / mean);
  }
  std::vector<int> order((size_t)N);
  for (int a = 0; a < N; ++a) order[a] = a;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
  for (int r = 0; r < N; ++r) new_of_old[order[r]] = r;
  return true;
}
