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
// order of their sub-positions, ties by original id.  On a structured block this reproduces a brick numbering
// exactly; on an unstructured mesh it is a bucket sort into compact boxes of ~bx*by*bz nodes.  Deterministic in
// (coordinates, connectivity).
#include "feahip_internal.h"
#include <algorithm>
#include <cmath>
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
  {
    const int stride = std::max(1, E / 200000);
    std::vector<double> d[3];
    for (int e = 0; e < E; e += stride)
      for (int p = 0; p < npe; ++p)
        for (int q = p + 1; q < npe; ++q) {
          const int a = conn[(size_t)e * npe + p], b = conn[(size_t)e * npe + q];
          for (int k = 0; k < 3; ++k) {
            const double v = std::fabs(X[(size_t)a * 3 + k] - X[(size_t)b * 3 + k]);
            if (v > 1e-9 * ext[k]) d[k].push_back(v);
          }
        }
    for (int k = 0; k < 3; ++k) { h[k] = median_inplace(d[k]); if (!(h[k] > 0.0)) return false; }
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
      org[k] = lo[k] - 0.5 * h[k] * scale;                    // node planes of a lattice sit inside the cells, not on their faces
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
