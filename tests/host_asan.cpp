// host-only sanitizer run over the map builders (pattern, visits, gather, quad, numbering, rank sub-mesh, multigrid setup)
#include "feahip_internal.h"
#include "amg.h"
#include <cstdio>
static const int P[6][3] = {{0,1,2},{0,2,1},{1,0,2},{1,2,0},{2,0,1},{2,1,0}};
int main()
{
  for (int quad = 0; quad < 2; ++quad) {
    const int nx = quad ? 9 : 24, ny = quad ? 20 : 60, nz = quad ? 8 : 22, m = quad ? 2 : 1;
    const int gx = m * nx + 1, gy = m * ny + 1, gz = m * nz + 1;
    auto id = [&](int i, int j, int k) { return (j * gz + k) * gx + i; };
    std::vector<int> conn; std::vector<double> pos((size_t)gx * gy * gz * 3);
    for (int j = 0; j < gy; ++j) for (int k = 0; k < gz; ++k) for (int i = 0; i < gx; ++i) {
      double *p = &pos[(size_t)id(i, j, k) * 3]; p[0] = i; p[1] = j; p[2] = k; }
    static const int ED[6][2] = {{0,1},{1,2},{0,2},{0,3},{1,3},{2,3}};
    for (int j = 0; j < ny; ++j) for (int k = 0; k < nz; ++k) for (int i = 0; i < nx; ++i)
      for (int p = 0; p < 6; ++p) {
        int c[3] = {i * m, j * m, k * m}; int v[4][3];
        for (int d = 0; d < 3; ++d) v[0][d] = c[d];
        for (int s = 0; s < 3; ++s) { c[P[p][s]] += m; for (int d = 0; d < 3; ++d) v[s + 1][d] = c[d]; }
        for (int s = 0; s < 4; ++s) conn.push_back(id(v[s][0], v[s][1], v[s][2]));
        if (quad) for (auto &e : ED) conn.push_back(id((v[e[0]][0] + v[e[1]][0]) / 2, (v[e[0]][1] + v[e[1]][1]) / 2, (v[e[0]][2] + v[e[1]][2]) / 2));
      }
    const int npe = quad ? 10 : 4, N = gx * gy * gz, E = (int)conn.size() / npe;
    HostPattern hp; std::string err;
    if (build_host_pattern(N, E, npe, conn.data(), hp, err)) { printf("pattern: %s\n", err.c_str()); return 1; }
    if (!quad) {
      HostVisits hv; build_host_visits(N, E, conn.data(), hp, hv);
      HostGather hg; build_host_gather(N, E, conn.data(), hp, 0, N, hg);
      HostGather hs; build_host_gather(N, E, conn.data(), hp, N / 3, 2 * N / 3, hs);      // a rank's rows only
      printf("tet4: N=%d E=%d chunks=%zu achunks=%zu visits=%d gather=%d/%d gather chunks=%d evaluations per element=%.2f\n", N, E,
             hp.chunk.size() - 1, hp.achunk.size() - 1, (int)hv.ok, (int)hg.ok, (int)hs.ok, hg.nchunks,
             hg.distinct_elems ? (double)hg.total_evals / (double)hg.distinct_elems : 0.0);
    } else {
      const int na = (int)hp.achunk.size() - 1;
      HostQuad hq; build_host_quad(N, E, npe, conn.data(), hp, 0, na, hq);
      HostQuad hs; build_host_quad(N, E, npe, conn.data(), hp, na / 3, 2 * na / 3, hs);      // a rank's chunks only
      printf("tet10: N=%d E=%d chunks=%zu achunks=%zu quad=%d/%d pairs=%zu\n", N, E, hp.chunk.size() - 1, hp.achunk.size() - 1, (int)hq.ok, (int)hs.ok, hq.qpair.size());
      HostGather10 g10; build_host_gather10(N, E, 10, conn.data(), hp, 0, N, g10);
      HostGather10 g10s; build_host_gather10(N, E, 10, conn.data(), hp, N / 3, 2 * N / 3, g10s);      // a rank's rows only
      printf("tet10: gather10=%d/%d chunks=%d in %.2f chunks per element\n", (int)g10.ok, (int)g10s.ok, g10.nchunks,
             g10.distinct_elems ? (double)g10.total_evals / (double)g10.distinct_elems : 0.0);
    }
    std::vector<HostAmgLevel> lv;
    const bool ok = build_host_amg(hp.rowptr, hp.colidx, pos, 0, N, lv);
    {                                       // a rank's diagonal block: rows of the middle third
      std::vector<HostAmgLevel> part;
      const bool okp = build_host_amg(hp.rowptr, hp.colidx, pos, N / 3, 2 * N / 3, part);
      long long inside = 0;
      if (okp) for (int i = 0; i < N; ++i) inside += part[0].agg[i] >= 0 ? 1 : 0;
      printf("  amg shard [%d,%d): ok=%d levels=%zu rows with an aggregate %lld\n", N / 3, 2 * N / 3, (int)okp, part.size(), inside);
      if (okp && inside != 2 * N / 3 - N / 3) return 2;
    }
    printf("  amg: ok=%d levels=%zu", (int)ok, lv.size());
    for (auto &L : lv) printf(" [N=%d S=%d Sc=%d nnzb=%zu]", L.N, L.S, L.Sc, L.colidx.size());
    printf("\n");
    {                                       // the library's numbering and one rank's sub-mesh with its own pattern
      std::vector<int> lib;
      const bool ren = locality_numbering(N, E, npe, conn.data(), pos.data(), lib);
      RankMesh rm;
      const int rc = build_rank_mesh(1, 3, N, E, npe, conn.data(), pos.data(), 0, nullptr, nullptr, nullptr, rm, err);
      HostPattern hl;
      const int rc2 = rc ? rc : build_host_pattern((int)rm.node_global.size(), (int)rm.elem_global.size(), npe, rm.elements.data(), hl, err, rm.n_own);
      printf("  numbering=%d rank mesh: rc=%d/%d owned %d of %zu local nodes, %zu elements, peers %zu\n", (int)ren, rc, rc2, rm.n_own,
             rm.node_global.size(), rm.elem_global.size(), rm.plan.peer.size());
      if (rc || rc2) return 3;
    }
  }
  return 0;
}
