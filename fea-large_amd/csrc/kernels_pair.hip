// kernels_pair.hip -- paired-visit assembly of linear tetrahedra.
//
// Same machinery as kernels_visit.hip (a wave owns a chunk of block rows,
// node coordinates staged in LDS, blocks summed in an LDS tile with
// ds_add_f64, diagonal block from the row sum, one coalesced write-out), but
// the unit of work of a lane is a PAIR of elements around the row node a that
// share a face through a: (a,p,q,r) and (a,p,q,s).  Both contribute to the
// blocks (a,p) and (a,q): the lane sums those in registers and adds each to
// the tile once; the second element costs one new node from LDS, not four.
// Per element visit: 18 LDS adds instead of 27, 7.5 LDS reads instead of 12 --
// the LDS pipe is what bounds the visit kernel (profiles/: 70-80 % busy).
// Host side: build_host_pairs (patches.cpp).
#include "fem_device.h"
#include <cstdlib>

struct PairArgs {
  int chunk0, nchunks, model;
  double lambda, mu;
  const ElemTable *tab;
  const VisitDesc *desc;
  const int *vnode;
  const uint4 *prec;
  const double *X0, *x;          // [N][4]
  const int *rowptr, *diag;
  double *K, *f;
  int *bad;
};

__device__ __forceinline__ void lds_add9(double *dst, const double (&v)[9])
{
#pragma unroll
  for (int q = 0; q < 9; ++q) __hip_atomic_fetch_add(dst + q, v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <bool DOK, bool DOF>
__global__ __launch_bounds__(64, 3)     // 3 waves per SIMD: at most 168 VGPRs
void k_assemble_pair(PairArgs A)
{
  __shared__ double sC[FEA_VISIT_MAX_NODES * 6];       // x, X0 of the chunk's nodes
  __shared__ double sK[DOK ? FEA_ACHUNK_BLOCKS * 9 + 2 : 2];
  __shared__ double sF[4][FEA_CHUNK_ROWS * 3 + 3];
  __shared__ int sRow[FEA_CHUNK_ROWS + 1];
  __shared__ int sDiag[FEA_CHUNK_ROWS];
  const int lane = threadIdx.x;
  const int nwg = gridDim.x, per = (nwg + 7) >> 3;     // XCD-aware order, see kernels_visit.hip
  const int cidx = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (cidx >= A.nchunks) return;
  const int chunk = A.chunk0 + cidx;
  const int vn0 = A.vnode[(size_t)chunk * FEA_VISIT_MAX_NODES + lane];
  const VisitDesc d = A.desc[chunk];
  const int nrows = d.r1 - d.r0;
  const int odd = d.b0 & 1;
  double *sKt = sK + odd;

  uint4 rec = make_uint4(0, 0, 0, 0);
  if (lane < d.nvisit) rec = A.prec[d.visit_off + lane];
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
    for (int t = lane; t < d.nb * 9; t += 64) sKt[t] = 0.0;
  if (DOF)
    for (int t = lane; t < 4 * (FEA_CHUNK_ROWS * 3 + 3); t += 64) (&sF[0][0])[t] = 0.0;
  __syncthreads();

  for (int p = lane; p - lane < d.nvisit; p += 64) {
    uint4 nxt = make_uint4(0, 0, 0, 0);
    if (p + 64 < d.nvisit) nxt = A.prec[d.visit_off + p + 64];
    if (p < d.nvisit) {
      const unsigned ids = rec.x, sl = rec.z, flags = rec.y >> 8;
      const int na = ids & 255u, np_ = (ids >> 8) & 255u, nq = (ids >> 16) & 255u, nr = ids >> 24, ns = rec.y & 255u;
      const int nd[4] = {na, np_, nq, nr};
      double xe[4][3], Xe[4][3];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double *cc = sC + nd[k] * 6;
#pragma unroll
        for (int j = 0; j < 3; ++j) { xe[k][j] = cc[j]; Xe[k][j] = cc[3 + j]; }
      }
      double accP[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, accQ[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, fa[3] = {0, 0, 0};
      const int rowoff = sRow[na] * 9;
      const int *gn = A.vnode + (size_t)chunk * FEA_VISIT_MAX_NODES;
      // ---- element A = (a, p, q, r) ----
      {
        GPState<4> s;
        gp_state<4, true, false>(xe, Xe, A.tab, 0, A.model, A.lambda, A.mu, s);
        if (!(((flags & 1u) ? -s.detJ : s.detJ) > 0.0) && DOK) {     // rare: report from its lowest-numbered node
          const int g0 = gn[na];
          if (g0 < gn[np_] && g0 < gn[nq] && g0 < gn[nr]) atomicAdd(A.bad, 1);
        }
        if (s.detJ != 0.0) {
          RowVecs rv;
          row_vectors(s.g[0], s.sig, s.l1, s.m1, s.vol, rv);
          if (DOF) { fa[0] -= rv.s[0]; fa[1] -= rv.s[1]; fa[2] -= rv.s[2]; }
          if (DOK) {
            double blk[9];
            block_row(rv, s.g[1], accP);
            block_row(rv, s.g[2], accQ);
            block_row(rv, s.g[3], blk);
            lds_add9(sKt + rowoff + (int)((sl >> 16) & 255u) * 9, blk);
          }
        }
      }
      // ---- element B = (a, p, q, s): one new node ----
      if (flags & 4u) {
        const double *cc = sC + ns * 6;
#pragma unroll
        for (int j = 0; j < 3; ++j) { xe[3][j] = cc[j]; Xe[3][j] = cc[3 + j]; }
        GPState<4> s;
        gp_state<4, true, false>(xe, Xe, A.tab, 0, A.model, A.lambda, A.mu, s);
        if (!(((flags & 2u) ? -s.detJ : s.detJ) > 0.0) && DOK) {
          const int g0 = gn[na];
          if (g0 < gn[np_] && g0 < gn[nq] && g0 < gn[ns]) atomicAdd(A.bad, 1);
        }
        if (s.detJ != 0.0) {
          RowVecs rv;
          row_vectors(s.g[0], s.sig, s.l1, s.m1, s.vol, rv);
          if (DOF) { fa[0] -= rv.s[0]; fa[1] -= rv.s[1]; fa[2] -= rv.s[2]; }
          if (DOK) {
            double bp[9], bq[9], blk[9];
            block_row(rv, s.g[1], bp);
            block_row(rv, s.g[2], bq);
            block_row(rv, s.g[3], blk);
#pragma unroll
            for (int q = 0; q < 9; ++q) { accP[q] += bp[q]; accQ[q] += bq[q]; }
            lds_add9(sKt + rowoff + (int)(sl >> 24) * 9, blk);
          }
        }
      }
      if (DOK) {
        lds_add9(sKt + rowoff + (int)(sl & 255u) * 9, accP);
        lds_add9(sKt + rowoff + (int)((sl >> 8) & 255u) * 9, accQ);
      }
      if (DOF) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
          __hip_atomic_fetch_add(&sF[lane & 3][na * 3 + i], fa[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    rec = nxt;
  }
  __syncthreads();

  if (DOK) {
    // K_aa = -sum_{b != a} K_ab: the diagonal block was never added to
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
}

int launch_assemble_pair(feahip_ctx *c, bool doK, bool doF)
{
  PairArgs A;
  A.chunk0 = c->achunk0; A.nchunks = c->nachunks_local; A.model = c->model;
  A.lambda = c->lambda; A.mu = c->mu; A.tab = c->d_table; A.desc = c->d_pairdesc; A.vnode = c->d_vnode;
  A.prec = reinterpret_cast<const uint4 *>(c->d_prec); A.X0 = c->d_X0; A.x = c->d_x;
  A.rowptr = c->d_rowptr; A.diag = c->d_diag; A.K = c->d_K; A.f = c->d_f; A.bad = c->d_flag + 1;
  if (c->nachunks_local <= 0) return FEAHIP_OK;
  const dim3 grid((c->nachunks_local + 7) & ~7), blk(64);
  if (doK && doF) hipLaunchKernelGGL((k_assemble_pair<true, true>), grid, blk, 0, c->stream, A);
  else if (doK)   hipLaunchKernelGGL((k_assemble_pair<true, false>), grid, blk, 0, c->stream, A);
  else            hipLaunchKernelGGL((k_assemble_pair<false, true>), grid, blk, 0, c->stream, A);
  FEA_HIP_CHECK(c, hipGetLastError());
  return FEAHIP_OK;
}
