// kernels_visit.hip -- stiffness / residual assembly of linear tetrahedra
// with node coordinates and connectivity staged through LDS.
//
// Same row-owner idea as kernels_assemble.hip (one wave owns a chunk of
// consecutive block rows, walks the (row node, element) visits of those rows,
// sums the row's 3x3 blocks in an LDS tile with ds_add_f64, writes every CSR
// value once).  What changes is where the data come from.  The visit kernel
// of kernels_assemble.hip chases inc -> conn -> coordinates -> rowptr through
// global memory in every pass of 64 visits: about twelve dependent memory
// latencies per chunk against ~2 us of arithmetic, so the wave sits stalled.
// Here the host has prepared, per chunk,
//   vnode : the nodes its elements touch, owned rows first  (coalesced read)
//   vrec  : per visit 4 chunk-local node ids, row node first, and the tile
//           positions of its 3 blocks                 (8 B, coalesced read)
// so the wave (1) gathers the coordinates of ~70 nodes into LDS once, (2)
// reads everything else from LDS: three dependent latencies per chunk.
// The global element->node map is not read at all.
// Replaces fea_solver.c:873-883 / 863-870 for TETRAHEDRA4 meshes.
#include "fem_device.h"
#include <cstdlib>

struct VisitArgs {
  int chunk0, nchunks, model;
  double lambda, mu;
  const ElemTable *tab;
  const VisitDesc *desc;
  const int *vnode;
  const uint2 *vrec;
  const double *X0, *x;          // [N][4]
  const int *rowptr, *diag;
  double *K, *f;
  int *bad;
  int nvisits;                   // length of vrec
  int nrows_total;               // block rows of the matrix
  int dbg;                       // timing experiments only (FEAHIP_DBG): 1 = conflict-free K adds, 2 = no f adds, 4 = phase stamps
  unsigned long long *stamps;    // [chunk][8] s_memtime stamps when dbg & 4
};

// DBG: the timing experiments of DESIGN.md section 4 (FEAHIP_DBG); the production instantiations carry none of it
template <bool DOK, bool DOF, bool DBG>
__global__ __launch_bounds__(64)
void k_assemble_visit(VisitArgs A)
{
  __shared__ double sC[FEA_VISIT_MAX_NODES * 6];       // x, X0 of the chunk's nodes
  __shared__ double sK[DOK ? FEA_ACHUNK_BLOCKS * 9 + 2 : 2];
  __shared__ double sF[4][FEA_ACHUNK_ROWS * 3 + 3];        // 4 replicas (a row's visits spread over them), padded off the same banks
  __shared__ int sRow[FEA_ACHUNK_ROWS + 1];             // first block of every row, relative to b0
  __shared__ int sDiag[FEA_ACHUNK_ROWS];
  const int lane = threadIdx.x;
  unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
  if (DBG && A.dbg & 4) t0 = __builtin_amdgcn_s_memtime();
  // node lists have a fixed stride per chunk, so they are fetched together with
  // the descriptor (one dependent latency less); unused tail entries are 0
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so
  // workgroup b and b+8 share an L2.  Give each XCD a contiguous eighth of the
  // chunks: neighbouring chunks then re-read each other's halo coordinates
  // from the same L2 (placement is a speed matter only).
  const int nwg = gridDim.x, per = (nwg + 7) >> 3;
  int cidx = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (DBG && A.dbg & 8) cidx = blockIdx.x;
  if (cidx >= A.nchunks) return;   // the grid is padded to a multiple of 8
  const int chunk = A.chunk0 + cidx;
  const int vn0 = A.vnode[(size_t)chunk * FEA_VISIT_MAX_NODES + lane];
  const VisitDesc d = A.desc[chunk];
  const int nrows = d.r1 - d.r0;
  // the tile sits at an LDS offset with the parity of the chunk's first global
  // value, so LDS and HBM agree on 16-byte alignment in the write-out
  const int odd = d.b0 & 1;
  double *sKt = sK + odd;

  uint2 rec = make_uint2(0, 0);
  if (lane < d.nvisit) rec = A.vrec[d.visit_off + lane];
  if (lane <= nrows) sRow[lane] = A.rowptr[d.r0 + lane] - d.b0;
  if (lane < nrows) sDiag[lane] = A.diag[d.r0 + lane] - d.b0;
  if (lane < d.nnode) {
    const size_t n = (size_t)vn0;
    const double2 a0 = *reinterpret_cast<const double2 *>(A.x + n * 4);
    const double2 a1 = *reinterpret_cast<const double2 *>(A.x + n * 4 + 2);
    const double2 c0 = *reinterpret_cast<const double2 *>(A.X0 + n * 4);
    const double2 c1 = *reinterpret_cast<const double2 *>(A.X0 + n * 4 + 2);
    double *o = sC + lane * 6;
    o[0] = a0.x; o[1] = a0.y; o[2] = a1.x; o[3] = c0.x; o[4] = c0.y; o[5] = c1.x;
  }
  if (DOK)
    for (int t = lane; 2 * t < d.nb * 9 + odd; t += 64)              // 16-byte stores from the aligned base of the tile
      reinterpret_cast<double2 *>(sK)[t] = make_double2(0.0, 0.0);
  if (DOF)
    for (int t = lane; t < 4 * (FEA_ACHUNK_ROWS * 3 + 3); t += 64) (&sF[0][0])[t] = 0.0;
  __syncthreads();
  if (DBG && A.dbg & 4) t1 = __builtin_amdgcn_s_memtime();

  for (int p = lane; p - lane < d.nvisit; p += 64) {
    // record of the next pass, in flight while this one computes
    uint2 nxt = make_uint2(0, 0);
    if (p + 64 < d.nvisit) nxt = A.vrec[d.visit_off + p + 64];
    if (p < d.nvisit && rec.y != 0xFFFFFFFFu) {           // 0xFFFFFFFF: idle lane of the schedule
      const unsigned ids = rec.x, sl = rec.y;
      const int n0 = ids & 255u, n1 = (ids >> 8) & 255u, n2 = (ids >> 16) & 255u, n3 = ids >> 24;
      const int nd[4] = {n0, n1, n2, n3};
      double xe[4][3], Xe[4][3];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double *cc = sC + nd[k] * 6;
#pragma unroll
        for (int j = 0; j < 3; ++j) { xe[k][j] = cc[j]; Xe[k][j] = cc[3 + j]; }
      }
      GPState<4> s;
      gp_state<4, true, false>(xe, Xe, A.tab, 0, A.model, A.lambda, A.mu, s);
      // the host may have renumbered the element with an odd permutation (bit 0 of sl):
      // the sign of det J is then the opposite of the stored element's
      if (!(((sl & 1u) ? -s.detJ : s.detJ) > 0.0) && DOK) {   // rare: count it from its lowest-numbered node only
        const int *gn = A.vnode + (size_t)chunk * FEA_VISIT_MAX_NODES;
        const int g0 = gn[n0];
        if (g0 < gn[n1] && g0 < gn[n2] && g0 < gn[n3]) atomicAdd(A.bad, 1);
      }
      if (s.detJ != 0.0) {                               // fea_solver.c:697: no gradient otherwise
        const double ga[3] = {s.g[0][0], s.g[0][1], s.g[0][2]};
        RowVecs rv;
        row_vectors(ga, s.sig, s.l1, s.m1, s.vol, rv);
        if (DOF && !(DBG && A.dbg & 2)) {
#pragma unroll
          for (int i = 0; i < 3; ++i)
            __hip_atomic_fetch_add(&sF[lane & 3][n0 * 3 + i], -rv.s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (DOK) {
#pragma unroll
          for (int k = 1; k < 4; ++k) {                  // the diagonal block comes from the row sum
            double blk[9];
            block_row(rv, s.g[k], blk);
            double *dst = sKt + (int)((sl >> (8 * k)) & 255u) * 9;    // tile position of block (row n0, column k)
            if (DBG && A.dbg & 1) dst = sKt + lane * 9;
#pragma unroll
            for (int q = 0; q < 9; ++q)
              __hip_atomic_fetch_add(dst + q, blk[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    rec = nxt;
  }
  __syncthreads();
  if (DBG && A.dbg & 4) t2 = __builtin_amdgcn_s_memtime();

  if (DOK) {
    // K_aa = -sum_{b != a} K_ab (shape functions sum to one).  The diagonal
    // block was never added to (still zero), so the whole row is summed;
    // four independent partial sums keep the LDS reads pipelined.
    for (int t = lane; t < nrows * 9; t += 64) {
      const int r = t / 9, q = t % 9;
      const int kb = sRow[r], ke = sRow[r + 1], kd = sDiag[r];
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      int k = kb;
      for (; k + 3 < ke; k += 4) {
        a0 += sKt[k * 9 + q]; a1 += sKt[(k + 1) * 9 + q]; a2 += sKt[(k + 2) * 9 + q]; a3 += sKt[(k + 3) * 9 + q];
      }
      for (; k < ke; ++k) a0 += sKt[k * 9 + q];
      sKt[kd * 9 + q] = -((a0 + a1) + (a2 + a3));
    }
    __syncthreads();
    if (DBG && A.dbg & 4) t3 = __builtin_amdgcn_s_memtime();
    // stream the finished rows out: 16-byte LDS reads and HBM stores
    double *Kd = A.K + (size_t)d.b0 * 9;
    const int total = d.nb * 9;
    if (odd && lane == 0) Kd[0] = sKt[0];
    const int npair = (total - odd) >> 1;
    for (int t = lane; t < npair; t += 64) {
      const int j = odd + 2 * t;
      *reinterpret_cast<double2 *>(Kd + j) = *reinterpret_cast<const double2 *>(sKt + j);
    }
    if (((total - odd) & 1) && lane == 0) Kd[total - 1] = sKt[total - 1];
  }
  if (DOF) {
    double *fd = A.f + (size_t)d.r0 * 3;
    for (int t = lane; t < nrows * 3; t += 64) fd[t] = (sF[0][t] + sF[1][t]) + (sF[2][t] + sF[3][t]);
  }
  if (DBG && A.dbg & 4) {
    __builtin_amdgcn_s_waitcnt(0);
    t4 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
      unsigned long long *o = A.stamps + (size_t)cidx * 8;
      o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t4 - t3; o[4] = t4 - t0; o[5] = t0; o[6] = t4; o[7] = t5;
    }
  }
}

int launch_assemble_visit(feahip_ctx *c, bool doK, bool doF)
{
  VisitArgs A;
  A.chunk0 = c->achunk0; A.nchunks = c->nachunks_local; A.model = c->model;
  A.lambda = c->lambda; A.mu = c->mu; A.tab = c->d_table; A.desc = c->d_vdesc; A.vnode = c->d_vnode;
  A.vrec = reinterpret_cast<const uint2 *>(c->d_vrec); A.X0 = c->d_X0; A.x = c->d_x;
  A.rowptr = c->d_rowptr; A.diag = c->d_diag; A.nvisits = c->nvisit_records; A.nrows_total = c->N; A.K = c->d_K; A.f = c->d_f; A.bad = c->d_flag + 1;
  A.dbg = 0; A.stamps = nullptr;
#ifdef FEAHIP_DEBUG
  // Diagnostic build only (make debug -> libfeahip_dbg.so): FEAHIP_DBG selects the timing experiments of DESIGN.md,
  // whose K and f are meaningless.  The production library has neither the lookup nor the instantiations.
  { const char *e = getenv("FEAHIP_DBG"); A.dbg = e ? atoi(e) : 0; }
  static unsigned long long *d_stamps = nullptr;
  static int stamps_cap = 0;
  if ((A.dbg & 4) && (!d_stamps || stamps_cap < c->nachunks)) {
    if (d_stamps) (void)hipFree(d_stamps);
    (void)hipMalloc((void **)&d_stamps, sizeof(unsigned long long) * 8 * (size_t)c->nachunks);
    stamps_cap = c->nachunks;
  }
  A.stamps = d_stamps;
#endif
  if (c->nachunks_local <= 0) return FEAHIP_OK;
  const dim3 grid((c->nachunks_local + 7) & ~7), blk(64);
#ifdef FEAHIP_DEBUG
  if (A.dbg) {
    if (doK && doF) hipLaunchKernelGGL((k_assemble_visit<true, true, true>), grid, blk, 0, c->stream, A);
    else if (doK)   hipLaunchKernelGGL((k_assemble_visit<true, false, true>), grid, blk, 0, c->stream, A);
    else            hipLaunchKernelGGL((k_assemble_visit<false, true, true>), grid, blk, 0, c->stream, A);
  } else
#endif
  {
    if (doK && doF) hipLaunchKernelGGL((k_assemble_visit<true, true, false>), grid, blk, 0, c->stream, A);
    else if (doK)   hipLaunchKernelGGL((k_assemble_visit<true, false, false>), grid, blk, 0, c->stream, A);
    else            hipLaunchKernelGGL((k_assemble_visit<false, true, false>), grid, blk, 0, c->stream, A);
  }
  FEA_HIP_CHECK(c, hipGetLastError());
#ifdef FEAHIP_DEBUG
  if (A.dbg & 4) {                                  // diagnostic build path: phase shares, never a timing
    static int printed = 0;
    (void)hipStreamSynchronize(c->stream);
    if (printed++ == 3) {
      std::vector<unsigned long long> h((size_t)c->nachunks_local * 8);
      (void)hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost);
      double sum[5] = {0, 0, 0, 0, 0};
      unsigned long long tmin = ~0ull, tmax = 0;
      for (int i = 0; i < c->nachunks_local; ++i) {
        for (int k = 0; k < 5; ++k) sum[k] += (double)h[(size_t)i * 8 + k];
        if (h[(size_t)i * 8 + 5] < tmin) tmin = h[(size_t)i * 8 + 5];
        if (h[(size_t)i * 8 + 6] > tmax) tmax = h[(size_t)i * 8 + 6];
      }
      fprintf(stderr, "[feahip stamps] K=%d F=%d chunks=%d  mean cycles: setup %.0f  rounds %.0f  diag %.0f  writeout %.0f  total %.0f  | kernel span %llu ticks\n",
              (int)doK, (int)doF, c->nachunks_local, sum[0] / c->nachunks_local, sum[1] / c->nachunks_local,
              sum[2] / c->nachunks_local, sum[3] / c->nachunks_local, sum[4] / c->nachunks_local, tmax - tmin);
    }
  }
#endif
  return FEAHIP_OK;
}
