// kernels_gather.hip -- GATHER assembly of linear tetrahedra: stiffness and
// residual with no atomics anywhere (replaces fea_solver.c:873-883 and
// :863-870 for TETRAHEDRA4 meshes; algebra of fem_device.h).
//
// The reference scatters (3n)^2 values per element into a growing sparse matrix.
// The row-owner kernels of kernels_visit.hip turned that into LDS atomics: one
// lane per (row node, element) visit, the element state re-evaluated by each of
// its four visits, 27 + 3 ds_add_f64 per visit -- and the LDS pipe (~8 clk per
// f64 atomic wave-instruction, more with bank conflicts) ended up bounding the
// kernel at 27 % of the HBM roofline.  Here the scatter is inverted on the host
// once (gather.cpp) and a 256-thread workgroup owns a chunk of consecutive
// block rows:
//   0. the chunk's map record and the coordinates of the nodes its elements
//      touch are loaded (two dependent latencies), coordinates into LDS;
//   1. thread <-> element: every element that touches the chunk's rows is
//      evaluated ONCE per chunk (J^-1, F^-1, sigma, tangent coefficients) and
//      parked in LDS as { g_b, t_b = vol (m1 g_b + sigma g_b), vol l1, vol m1 };
//   2. thread <-> off-diagonal block (a, b): walks the block's contribution list
//      (element, local a, local b), reads g_a, g_b, t_b, vl, vm from the record
//      and sums K_ab = sum_e [ vl g_a (x) g_b + vm g_b (x) g_a + (g_a . t_b) I ]
//      in registers, in ascending element order;
//      thread <-> slice of a row's visits: partial sums of the residual
//      -(t_a - vm g_a) = -vol sigma g_a (fea_solver.c:1096-1109);
//   3. finished blocks and residual partials go to an LDS tile (aliasing the
//      records); the diagonal block of every row is minus the sum of the row's
//      other blocks (shape functions sum to one), the residual of a row the sum
//      of its partials;
//   4. the rows leave LDS as one contiguous, coalesced stream of 16-byte
//      stores: every CSR value is written exactly once.
// Every sum has a fixed order: the assembly is bitwise reproducible.
#include "gather_device.h"
#ifndef G_NT_ROWS
#define G_NT_ROWS 1                // the rows of K as non-temporal stores (measurement switch)
#endif
// Persistent form: a workgroup walks a run of consecutive chunks and keeps the next chunk's loads in flight under
// the current chunk's arithmetic --
//   * the map words and the node coordinates of chunk i+1 (and the node ids of chunk i+2) are requested before the
//     state phase of chunk i and sit in registers; the coordinates move into the LDS tile after chunk i's gather
//     phase (the tile is dead from the end of the state phase), a whole state + gather phase after their request;
//   * the finished rows of chunk i are stored last and nobody waits for them.
template <bool DOK, bool DOF, bool NH>
__global__ __launch_bounds__(FEA_G_THREADS, FEA_G_BIG ? 4 : 3)
void k_assemble_gather(GatherArgs A, int run_len)
{
  extern __shared__ double2 g_smem[];
  double2 *sC = g_smem;                                       // per node slot 3 x 16 bytes: (x0,x1) (x2,X0) (X1,X2)
  double *sT = reinterpret_cast<double *>(g_smem + A.lay.max_nodes * 3);   // element records; later the K tile and the residual partials
  const int t = threadIdx.x;
#ifdef FEAHIP_DEBUG
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), real0 = __builtin_amdgcn_s_memrealtime();
#endif
  // XCD-aware order: workgroups b and b+8 share an L2; each XCD gets a contiguous eighth of the runs so that
  // neighbouring chunks re-read each other's halo coordinates from the same L2 (speed only)
  const int nruns = (A.nchunks + run_len - 1) / run_len;
  const int per = ((int)gridDim.x + 7) >> 3;
  const int ridx = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (ridx >= nruns) return;
  int chunk = ridx * run_len;
  const int cend = min(A.nchunks, chunk + run_len);
  constexpr int REC = DOK ? GREC : GREC_F;
  const size_t stride = (size_t)A.lay.stride;
  const unsigned char *rec = A.maps + (size_t)(A.chunk0 + chunk) * stride;
  const bool node_lane = t < A.lay.max_nodes;
  const int wslot = __builtin_amdgcn_readfirstlane(t >> 6);   // this wave's slot in the workgroup, provably wave-uniform

  // the Gauss weight is read ONCE: a load inside the loop would be the youngest memory operation when the state
  // phase needs it, and waiting for it waits for every prefetch issued before it (in-order counter)
  const double gauss_w = A.tab->w[0];
  // ---- prologue: maps and coordinates of the first chunk, nothing to hide behind
  GMaps mn;                                                   // "next": the chunk about to be worked on
  g_load_maps<DOK, DOF>(A.lay, G_ABL(64) ? A.maps + (size_t)A.chunk0 * stride : rec, t, mn);      // 64: every chunk reads the first chunk's map words (timing: what de-duplicated maps could save)
  GatherHeader hn = *reinterpret_cast<const GatherHeader *>(rec);
  int node1 = reinterpret_cast<const int *>(rec + A.lay.o_nodes)[t & (FEA_G_MAX_NODES - 1)];       // node slot t of the chunk whose coordinates are loaded next
  if (node_lane) {
    const double *gx = A.x + (size_t)node1 * 4, *gX = A.X0 + (size_t)node1 * 4;
    const double2 a0 = *reinterpret_cast<const double2 *>(gx), a1 = *reinterpret_cast<const double2 *>(gx + 2);
    const double2 c0 = *reinterpret_cast<const double2 *>(gX), c1 = *reinterpret_cast<const double2 *>(gX + 2);
    sC[t * 3] = a0; sC[t * 3 + 1] = make_double2(a1.x, c0.x); sC[t * 3 + 2] = make_double2(c0.y, c1.x);
  }
  {
    const int cn = min(chunk + 1, cend - 1);
    node1 = reinterpret_cast<const int *>(A.maps + (size_t)(A.chunk0 + cn) * stride + A.lay.o_nodes)[t & (FEA_G_MAX_NODES - 1)];
  }
  // nothing is pending when the loop is entered: otherwise the loop header inherits "node1 may still be in
  // flight" and the compiler waits for everything (s_waitcnt vmcnt(0)) at the top of EVERY iteration
  asm volatile("" : : "v"(mn.eids), "v"(mn.tpos), "v"(mn.cw[0]), "v"(mn.cw[1]), "v"(mn.cw[2]), "v"(mn.cw[3]), "v"(mn.cw[4]), "v"(mn.cw[5]),
               "v"(mn.vw[0]), "v"(mn.vw[1]), "v"(mn.kd), "v"(mn.vb), "v"(mn.ve), "v"(node1), "v"(gauss_w));
  G_BARRIER();

  for (;;) {
    G_STAMP(0);
    const bool more = chunk + 1 < cend;
    const GatherHeader h = hn;
    int hword;
    double2 ca0, ca1, cc0, cc1;
    const GMaps m = mn;
    const int nrows = h.r1 - h.r0;
    // contribution words THIS wave's block threads walk: the host sorts the blocks by list length and balances the waves
    // over the SIMDs (gather.cpp), so a wave stops at the depth of its own longest list, not of the chunk's
    const unsigned wdw = wslot < 4 ? h.wdepth[0] : (wslot < 8 ? h.wdepth[1] : h.wdepth[2]);     // (no dynamic index: the header stays in scalar registers)
    const int wd = wslot < G_TASK_THREADS / 64 ? (int)((wdw >> (8 * (wslot & 3))) & 255u) : 0;
    rec = A.maps + (size_t)(A.chunk0 + chunk) * stride;

    __builtin_amdgcn_s_setprio(PRIO_STATE);
    // ---- phase 1: one state evaluation per element of the chunk
    if (t < h.nelem && m.eids != 0xFFFFFFFFu) {
      const unsigned eids = m.eids;
      const int nd[4] = {(int)(eids & 255u), (int)((eids >> 8) & 255u), (int)((eids >> 16) & 255u), (int)(eids >> 24)};
      double xe[4][3], Xe[4][3];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double2 *cc = sC + nd[k] * 3;                 // 16-byte reads: the bank slot follows the node slot mod 16
        const double2 p0 = cc[0], p1 = cc[1], p2 = cc[2];
        xe[k][0] = p0.x; xe[k][1] = p0.y; xe[k][2] = p1.x;
        Xe[k][0] = p1.y; Xe[k][1] = p2.x; Xe[k][2] = p2.y;
      }
      double R[REC];
      double detJ;
      if (G_ABL(8)) { detJ = 1.0; for (int q = 0; q < REC; ++q) R[q] = xe[q & 3][q % 3] + Xe[(q >> 2) & 3][q % 3]; }
      else detJ = NH ? lintet_record_nh<DOK>(xe, Xe, gauss_w, A.lambda, A.mu, R)
                     : lintet_record_a5<DOK>(xe, Xe, gauss_w, A.lambda, A.mu, R);
      if (!(detJ > 0.0)) {                             // rare, kept off the fast path
        if (DOK) {                                     // counted by the chunk that owns its lowest-numbered node
          const int *gn = reinterpret_cast<const int *>(rec + A.lay.o_nodes);
          const int g0 = min(min(gn[nd[0]], gn[nd[1]]), min(gn[nd[2]], gn[nd[3]]));
          if (g0 >= h.r0 && g0 < h.r1) atomicAdd(A.bad, 1);
        }
      }
      if (detJ == 0.0) {                               // fea_solver.c:697: no gradient, no contribution: an all-zero record
        double2 *o = reinterpret_cast<double2 *>(sT + t * REC);
#pragma unroll
        for (int q = 0; q < REC / 2; ++q) o[q] = make_double2(0.0, 0.0);
      } else if (!G_ABL(1)) g_store_record<DOK>(sT + t * REC, R);
      else sT[t * REC] = R[0] + R[25 % REC];
    } else if (t < h.nelem) {                          // unused slot: the all-zero record empty list slots point at
      double2 *o = reinterpret_cast<double2 *>(sT + t * REC);
#pragma unroll
      for (int q = 0; q < REC / 2; ++q) o[q] = make_double2(0.0, 0.0);
    }
    // ---- next chunk's loads, a gather phase ahead of their first use: header, map words, node coordinates (by the
    // node ids requested one chunk earlier), and the node ids of the chunk after it.  Clamped indices instead of
    // branches (a load under a branch is waited for at the join).
    // Issued HERE, after the state evaluation: at the top of the chunk the CU's memory pipeline (one in-order queue)
    // still holds the previous chunk's rows, 69 KB leaving at the CU's share of the HBM write rate, and a wave had to
    // get its thirteen loads into that queue before it could start evaluating (~2 300 of the state phase's 6 900
    // cycles, by the in-kernel stamps).  By now the rows are gone; the waves without elements issue theirs at once.
    // (Issued BEFORE the rows instead, the loads keep the queue's entries for their whole latency and it is the row
    // stores that cannot be issued: 8 500 cycles of write-out instead of 1 700.)
    {
      const int c1 = min(chunk + 1, cend - 1), c2 = min(chunk + 2, cend - 1);
      const unsigned char *rec1 = A.maps + (size_t)(A.chunk0 + c1) * stride;
      // the header of the next chunk as a VECTOR load (lane l holds word l, read back with v_readlane): a scalar
      // load shares its counter with the LDS, and every barrier's wait for the LDS would wait for it as well
      hword = reinterpret_cast<const int *>(rec1)[t & 15];
      // the interior bricks of a structured block have IDENTICAL map words (chunk-local indices only): the host flags a
      // chunk whose successor's words equal its own, and the words then simply stay in their registers -- no loads, no
      // HBM traffic (the maps were 0.6 of the 0.8 GB a launch fetched), a shorter issue queue at the end of this phase
      if (!(h.flags & 1) || G_ABL(64)) g_load_maps<DOK, DOF>(A.lay, G_ABL(64) ? A.maps + (size_t)A.chunk0 * stride : rec1, t, mn);
      // coordinates and node ids: the waves that hold node slots only (a wave-uniform branch: the other waves issue
      // nothing -- every load a wave issues here queues behind the previous chunk's rows)
      if (wslot < FEA_G_MAX_NODES / 64) {
        const size_t n1 = (size_t)(node_lane ? node1 : 0);
        ca0 = *reinterpret_cast<const double2 *>(A.x + n1 * 4); ca1.x = A.x[n1 * 4 + 2];
        cc0 = *reinterpret_cast<const double2 *>(A.X0 + n1 * 4); cc1.x = A.X0[n1 * 4 + 2];
        node1 = reinterpret_cast<const int *>(A.maps + (size_t)(A.chunk0 + c2) * stride + A.lay.o_nodes)[t & (FEA_G_MAX_NODES - 1)];
      }
    }

    G_LDS_DRAIN();                                     // the asm record stores
    G_BARRIER();                                       // records visible; the coordinate tile is dead
    G_STAMP(1);

    __builtin_amdgcn_s_setprio(PRIO_GATHER);
    // ---- phase 2: block sums and residual partials, out of the records
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, accd = 0;
    if (DOK && t < h.noffd) {
      // the reads of the next contribution are in flight while the current one is summed; a list of n words is
      // walked as 2n entries whatever it holds (empty entries read the all-zero record)
#define G_FETCH(w) (G_ABL(4) ? g_fetch_cheap(sT, w) : g_fetch(sT, w))
#define G_APPLY(r) do { if (G_ABL(2)) g_apply_cheap(r, acc); else g_apply(r, acc, accd); } while (0)
#pragma unroll
      for (int k = 0; k < FEA_G_REGW; ++k)
        if (k < wd) { { const GRead r = G_FETCH(m.cw[k] & 0xFFFFu); G_APPLY(r); } { const GRead r = G_FETCH(m.cw[k] >> 16); G_APPLY(r); } }
      for (int k = FEA_G_REGW; k < wd; ++k) {               // blocks with more than 12 contributions
        const unsigned w = reinterpret_cast<const unsigned *>(rec + A.lay.o_clist)[k * FEA_G_THREADS + t];
        g_consume(sT, w & 0xFFFFu, acc, accd); g_consume(sT, w >> 16, acc, accd);
      }
      g_finish(acc, accd);
    }
    double dg[6] = {0, 0, 0, 0, 0, 0};
    double fa[3] = {0, 0, 0};
    if (DOK && t >= G_TASK_THREADS && t - G_TASK_THREADS < 4 * nrows) {
#pragma unroll
      for (int k = 0; k < FEA_G_REGW; ++k)
        if (k < h.ddepth) { g_consume_diag<DOF>(sT, m.cw[k] & 0xFFFFu, dg, fa); g_consume_diag<DOF>(sT, m.cw[k] >> 16, dg, fa); }
      for (int k = FEA_G_REGW; k < h.ddepth; ++k) {        // nodes with more than 48 elements around them
        const unsigned w = reinterpret_cast<const unsigned *>(rec + A.lay.o_dlist)[k * FEA_G_DIAG_LANES + t - G_TASK_THREADS];
        g_consume_diag<DOF>(sT, w & 0xFFFFu, dg, fa); g_consume_diag<DOF>(sT, w >> 16, dg, fa);
      }
    }
    if (!DOK && t < h.nvthr) {                         // residual alone: slices of a row's visits on every wave
      g_visit<DOK>(sT, m.vw[0], fa);
      if (h.vdepth > 1) g_visit<DOK>(sT, m.vw[1], fa);
      for (int v = 2; v < h.vdepth; ++v)
        g_visit<DOK>(sT, reinterpret_cast<const unsigned short *>(rec + A.lay.o_vlist)[v * FEA_G_THREADS + t], fa);
    }
    G_STAMP(2);
    __builtin_amdgcn_s_setprio(PRIO_OUT);
    G_BARRIER();                                       // the records are dead: their space becomes the tile
    G_STAMP(3);

    // the tile sits at an LDS offset with the parity of the chunk's first global value, so LDS and HBM agree on
    // 16-byte alignment in the write-out
    const int odd = h.b0 & 1;
    double *sK = sT + odd;
    double *sF = DOK ? sT + ((A.lay.max_tile * 9 + 3) & ~1) : sT;
    if (DOK && t < h.noffd && !G_ABL(16)) {
      const int bpos = (int)(m.tpos & 0xFFFFu), mpos = (int)(m.tpos >> 16);
      if (bpos != 0xFFFF) {                            // (a thread slot the host left without a block)
#pragma unroll
        for (int q = 0; q < 9; ++q) sK[bpos * 9 + q] = acc[q];
      }
      if (mpos != 0xFFFF) {                            // K_ba = K_ab' (fea_solver.c:1249 relies on the same symmetry)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) sK[mpos * 9 + 3 * j + i] = acc[3 * i + j];
      }
    }
    if (DOK && t >= G_TASK_THREADS) {                  // the four partial sums of a row's diagonal block meet in its first lane
#pragma unroll
      for (int q = 0; q < 6; ++q) dg[q] = g_quad_sum(dg[q]);
      if (DOF) {
#pragma unroll
        for (int q = 0; q < 3; ++q) fa[q] = g_quad_sum(fa[q]);
      }
      if ((t & 3) == 0 && t - G_TASK_THREADS < 4 * nrows) {
        double *o = sK + m.kd * 9;
        o[0] = dg[0]; o[1] = dg[1]; o[2] = dg[2]; o[3] = dg[1]; o[4] = dg[3]; o[5] = dg[4]; o[6] = dg[2]; o[7] = dg[4]; o[8] = dg[5];
      }
    }
    if (!DOK && t < h.nvthr) { sF[t * 3] = fa[0]; sF[t * 3 + 1] = fa[1]; sF[t * 3 + 2] = fa[2]; }
    // The prefetched words of the next chunk are "used" HERE, before this chunk's row stores are issued: the
    // hardware counts loads and stores in one in-order counter and the number of stores is not a compile-time
    // constant, so a first use after the stores would make the compiler wait for everything in flight, the
    // stores just issued included (s_waitcnt vmcnt(0): measured, a third of the chunk time).  Here only the
    // prefetches themselves are younger, they were requested a state + gather phase ago, and nothing ever waits
    // for a store.
    G_STAMP(4);
    asm volatile("" : : "v"(mn.eids), "v"(mn.tpos), "v"(mn.cw[0]), "v"(mn.cw[1]), "v"(mn.cw[2]), "v"(mn.cw[3]), "v"(mn.cw[4]), "v"(mn.cw[5]),
                 "v"(mn.vw[0]), "v"(mn.vw[1]), "v"(mn.kd), "v"(mn.vb), "v"(mn.ve), "v"(node1), "v"(hword));
    if (more && node_lane) {                           // next chunk's coordinates: the tile has been dead since the state phase
      sC[t * 3] = ca0; sC[t * 3 + 1] = make_double2(ca1.x, cc0.x); sC[t * 3 + 2] = make_double2(cc0.y, cc1.x);
    }
    hn.r0 = __builtin_amdgcn_readlane(hword, 0); hn.r1 = __builtin_amdgcn_readlane(hword, 1);
    hn.b0 = __builtin_amdgcn_readlane(hword, 2); hn.nb = __builtin_amdgcn_readlane(hword, 3);
    hn.nnode = __builtin_amdgcn_readlane(hword, 4); hn.nelem = __builtin_amdgcn_readlane(hword, 5);
    hn.noffd = __builtin_amdgcn_readlane(hword, 6); hn.depth = __builtin_amdgcn_readlane(hword, 7);
    hn.nvthr = __builtin_amdgcn_readlane(hword, 8); hn.vdepth = __builtin_amdgcn_readlane(hword, 9);
    hn.ddepth = __builtin_amdgcn_readlane(hword, 10);
    hn.wdepth[0] = (unsigned)__builtin_amdgcn_readlane(hword, 11); hn.wdepth[1] = (unsigned)__builtin_amdgcn_readlane(hword, 12);
    hn.wdepth[2] = (unsigned)__builtin_amdgcn_readlane(hword, 13); hn.flags = __builtin_amdgcn_readlane(hword, 14);
    G_BARRIER();
    G_STAMP(5);
    if (!DOK) {                                        // f_a = sum of the row's partials
      const int ft = t;
      const int fr = ft / 3, fi = ft - 3 * fr;
      if (fr < nrows) {
        double a = 0;
        for (int k = m.vb; k < m.ve; ++k) a += sF[k * 3 + fi];
        A.f[(size_t)(h.r0 + fr) * 3 + fi] = a;
      }
    }
    if (DOK && DOF && t >= G_TASK_THREADS && (t & 3) == 0 && t - G_TASK_THREADS < 4 * nrows) {
      // the row's residual, stored with the rows (after the wait for the prefetches above: no store before it)
      double *fo = A.f + (size_t)(h.r0 + ((t - G_TASK_THREADS) >> 2)) * 3;
      fo[0] = fa[0]; fo[1] = fa[1]; fo[2] = fa[2];
    }
    if (DOK) {
      // ---- phase 3: stream the finished rows out, 16-byte LDS reads and HBM stores
      double *Kd = A.K + (size_t)h.b0 * 9;
      const int total = h.nb * 9;
      if (odd && t == 0) Kd[0] = sK[0];
      const int npair = (total - odd) >> 1;
      for (int j = t; j < npair; j += FEA_G_THREADS) {
        const int p = odd + 2 * j;
        if (G_ABL(32) && j >= 64) break;               // timing experiment: one 1 KB store per chunk instead of all rows
#if G_NT_ROWS
        __builtin_nontemporal_store(*reinterpret_cast<const g_v2d *>(sK + p), reinterpret_cast<g_v2d *>(Kd + p));   // written once, read by other kernels only
#else
        *reinterpret_cast<double2 *>(Kd + p) = *reinterpret_cast<const double2 *>(sK + p);
#endif
      }
      if (((total - odd) & 1) && t == 0) Kd[total - 1] = sK[total - 1];
    }
    G_BARRIER();                                       // the tile is free again (its reads are done)
    G_STAMP(6);
#ifdef FEAHIP_DEBUG
    if (A.stamps) for (int i = 0; i < 6; ++i) sa[i] += st[i + 1] - st[i];
#endif
    if (!more) break;
    ++chunk;
  }
#ifdef FEAHIP_DEBUG
  if (A.stamps && (t & 63) == 0) {                     // one line per wave: [run][wave][8]
    unsigned long long *o = A.stamps + ((size_t)ridx * (FEA_G_THREADS / 64) + (t >> 6)) * 8;
    for (int i = 0; i < 6; ++i) o[i] = sa[i];
    o[6] = __builtin_amdgcn_s_memtime() - clk0;          // shader cycles of this run ...
    o[7] = __builtin_amdgcn_s_memrealtime() - real0;     // ... and 100 MHz ticks: their ratio is the in-kernel clock
  }
#endif
}

int ensure_gather(feahip_ctx *c)
{
  if (c->have_gather && c->gather_row0 == c->row0 && c->gather_row1 == c->row1) return FEAHIP_OK;
  if (!c->h_pat || c->h_conn.empty() || !c->linear_tet || c->G != 1) return FEAHIP_OK;
  // a new row range (re-shard): the old maps describe rows K no longer holds (release_k re-allocates the window), so
  // they go before anything else can launch them -- also when the new range is known not to fit, or turns out not to
  if (c->d_gmaps) { (void)hipFree(c->d_gmaps); c->d_gmaps = nullptr; }
  c->have_gather = false; c->ngchunks = 0; c->gather_row0 = c->gather_row1 = -1;
  if (c->gather_failed && c->gather_fail_row0 == c->row0 && c->gather_fail_row1 == c->row1) return FEAHIP_OK;
  HostGather hg;
  build_host_gather(c->N, c->E, c->h_conn.data(), *c->h_pat, c->row0, c->row1, hg);
  if (!hg.ok) { c->gather_failed = true; c->gather_fail_row0 = c->row0; c->gather_fail_row1 = c->row1; return FEAHIP_OK; }   // this row range only: another shard of the same context may fit
  FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_gmaps, hg.blob.size() ? hg.blob.size() : 1));
  FEA_HIP_CHECK(c, hipMemcpy(c->d_gmaps, hg.blob.data(), hg.blob.size(), hipMemcpyHostToDevice));
  if (!c->gather_lay) c->gather_lay = new GatherLayout();
  *c->gather_lay = hg.lay;
  c->ngchunks = hg.nchunks;
  c->gather_row0 = c->row0; c->gather_row1 = c->row1;
  c->gather_bytes = (long long)hg.blob.size();
  c->gather_evals_per_element = hg.distinct_elems ? (double)hg.total_evals / (double)hg.distinct_elems : 0.0;
  c->gather_same_words = hg.same_as_previous;
  c->have_gather = true;
  return FEAHIP_OK;
}

#ifdef FEAHIP_DEBUG
// diagnostic build only: one chunk's map record and the layout, for the host-side LDS bank model (dbg/lds_model.py)
extern "C" int feahip_debug_gather_record(feahip_ctx *c, int chunk, int *layout_ints, unsigned char *record)
{
  int rc = ensure_gather(c);
  if (rc || !c->have_gather) return FEAHIP_ESTATE;
  if (chunk < 0) chunk = c->ngchunks / 2;
  memcpy(layout_ints, c->gather_lay, sizeof(GatherLayout));
  if (record) (void)hipMemcpy(record, c->d_gmaps + (size_t)chunk * c->gather_lay->stride, c->gather_lay->stride, hipMemcpyDeviceToHost);
  return (int)(sizeof(GatherLayout) / sizeof(int));
}
#endif

int launch_assemble_gather(feahip_ctx *c, bool doK, bool doF)
{
  GatherArgs A;
  A.chunk0 = 0; A.nchunks = c->ngchunks; A.model = c->model; A.lambda = c->lambda; A.mu = c->mu;
  A.tab = c->d_table; A.maps = c->d_gmaps; A.lay = *c->gather_lay; A.X0 = c->d_X0; A.x = c->d_x;
  A.K = c->d_K; A.f = c->d_f; A.bad = c->d_flag + 1; A.stamps = nullptr; A.ablate = 0;
  if (c->ngchunks <= 0) return FEAHIP_OK;
#ifdef FEAHIP_DEBUG
  static unsigned long long *d_stamps = nullptr;
  static int stamps_cap = 0;
  { const char *e = getenv("FEAHIP_GATHER_ABLATE"); A.ablate = e ? atoi(e) : 0; }
  const char *dbg = getenv("FEAHIP_GATHER_STAMPS");
  if (dbg && atoi(dbg)) {
    if (!d_stamps || stamps_cap < c->ngchunks) {
      if (d_stamps) (void)hipFree(d_stamps);
      (void)hipMalloc((void **)&d_stamps, sizeof(unsigned long long) * 8 * (FEA_G_THREADS / 64) * (size_t)c->ngchunks);
      (void)hipMemset(d_stamps, 0, sizeof(unsigned long long) * 8 * (FEA_G_THREADS / 64) * (size_t)c->ngchunks);
      stamps_cap = c->ngchunks;
    }
    A.stamps = d_stamps;
  }
#endif
  // chunks per workgroup run: one workgroup is resident per CU, so the runs are cut to give every CU the same number of
  // them, two per CU (FEAHIP_GATHER_RUN: a fixed run length instead; tuning only, results unchanged)
  static int run_env = -1;
  if (run_env < 0) { const char *e = getenv("FEAHIP_GATHER_RUN"); run_env = e && atoi(e) > 0 ? atoi(e) : 0; }
  static int ncu_of[64];
  int &ncu = ncu_of[c->device & 63];
  if (ncu <= 0) { hipDeviceProp_t p; ncu = (hipGetDeviceProperties(&p, c->device) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }
  // One workgroup is resident per CU: with k runs per CU a launch lasts ceil(runs / CUs) rounds of run_len chunks.  Two runs
  // per CU even out the lighter boundary chunks on a whole mesh (28 682 chunks: 2 x 57); a rank of eight has 3 585 chunks
  // and 2 x 8 = 16 chunk times where one run of 15 does -- whichever of k = 1, 2 gives the shorter launch is taken.
  int run_len = 16;
  if (run_env) run_len = run_env;
  else if (FEA_G_BIG == 1) {
    long best = -1;
    for (int k = 2; k >= 1; --k) {
      const int rl = std::max(1, (c->ngchunks + k * ncu - 1) / (k * ncu)), nr = (c->ngchunks + rl - 1) / rl;
      const long cost = (long)((nr + ncu - 1) / ncu) * rl;
      if (best < 0 || cost < best) { best = cost; run_len = rl; }
    }
  }
  const int nruns = (c->ngchunks + run_len - 1) / run_len;
  const dim3 grid((nruns + 7) & ~7), blk(FEA_G_THREADS);
  // LDS: coordinates (48 bytes per node slot) | element records, later the K tile (+1 double of alignment slack) and the residual partials
  const int regK = std::max(A.lay.max_elems * GREC, ((A.lay.max_tile * 9 + 3) & ~1) + 3 * FEA_G_THREADS);
  const int regF = std::max(A.lay.max_elems * GREC_F, 3 * FEA_G_THREADS);
  const int ldsK = A.lay.max_nodes * 48 + ((regK + 1) & ~1) * 8, ldsF = A.lay.max_nodes * 48 + ((regF + 1) & ~1) * 8;
  const bool nh = c->model == FEAHIP_MODEL_COMPRESSIBLE_NEOHOOKEAN;
#define G_LAUNCH(K, F, M, LDS)                                                                                         \
  do {                                                                                                               \
    FEA_HIP_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_assemble_gather<K, F, M>),                \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS));                         \
    hipLaunchKernelGGL((k_assemble_gather<K, F, M>), grid, blk, LDS, c->stream, A, run_len);                                  \
  } while (0)
  if (doK && doF) { if (nh) G_LAUNCH(true, true, true, ldsK); else G_LAUNCH(true, true, false, ldsK); }
  else if (doK)   { if (nh) G_LAUNCH(true, false, true, ldsK); else G_LAUNCH(true, false, false, ldsK); }
  else            { if (nh) G_LAUNCH(false, true, true, ldsF); else G_LAUNCH(false, true, false, ldsF); }
#undef G_LAUNCH
  FEA_HIP_CHECK(c, hipGetLastError());
#ifdef FEAHIP_DEBUG
  if (A.stamps) {
    static int calls = 0;
    if (++calls == 50) {
      (void)hipStreamSynchronize(c->stream);
      constexpr int NW = FEA_G_THREADS / 64;
      std::vector<unsigned long long> hst((size_t)c->ngchunks * 8 * NW);
      (void)hipMemcpy(hst.data(), A.stamps, hst.size() * 8, hipMemcpyDeviceToHost);
      double sum[NW][8] = {};
      for (int i = 0; i < nruns; ++i)
        for (int w = 0; w < NW; ++w)
          for (int q = 0; q < 8; ++q) sum[w][q] += (double)hst[((size_t)i * NW + w) * 8 + q];
      fprintf(stderr, "[gather stamps] in-kernel clock %.0f MHz (s_memtime / s_memrealtime x 100 MHz over a run), run = %.0f shader cycles for %d chunks\n",
              sum[0][7] > 0 ? 100.0 * sum[0][6] / sum[0][7] : 0.0, sum[0][6] / nruns, run_len);
      for (int w = 0; w < NW; w += (NW > 4 ? 5 : 1))
        fprintf(stderr, "[gather stamps K=%d F=%d wave %d, per chunk] state %.0f  gather %.0f  barrier B %.0f  tile writes %.0f  wait for the prefetched words + coordinates + barrier C %.0f  rows out (+ barrier D) %.0f cycles\n",
                (int)doK, (int)doF, w, sum[w][0] / c->ngchunks, sum[w][1] / c->ngchunks, sum[w][2] / c->ngchunks, sum[w][3] / c->ngchunks,
                sum[w][4] / c->ngchunks, sum[w][5] / c->ngchunks);
    }
  }
#endif
  return FEAHIP_OK;
}
