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

// ---------------------------------------------------------------------------
// The same work, software-pipelined over a run of consecutive chunks.
//
// k_assemble_visit spends 38 % of a wave's life in front of its first FMA: the
// descriptor / node list come back (one HBM latency), then the coordinates
// they point to (a second one), and the registers and LDS of the wave sit idle
// meanwhile -- the CU is short of waves in their arithmetic phase, not of
// LDS or VALU cycles (PMC: VALU 48 % busy; conflict-free LDS adds buy 7 %).
// Here one wave walks `run_len` consecutive chunks and keeps the loads of the
// next chunk in flight under the work of the current one:
//   * while chunk i is in its passes: node list + descriptor of chunk i+1
//     (one VGPR, scalar registers) and the visit records of its first pass;
//   * as soon as the passes of chunk i are over (the coordinate tile is dead):
//     the coordinates of chunk i+1 go HBM -> LDS with global_load_lds (no
//     registers held), and land while the wave sums the diagonal blocks and
//     streams the rows of chunk i out.
// A workgroup is a single wave: its LDS operations execute in program order,
// so the phases need no barrier -- and must not use __syncthreads(), which
// would drain the LDS-DMA loads early (s_waitcnt vmcnt(0)).
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef int v8i __attribute__((ext_vector_type(8)));

// LDS-DMA load of 16 bytes per active lane: lane l's bytes land at lds_base + 16 l.
// Inline asm on purpose: with the builtin the compiler drains the load
// (s_waitcnt vmcnt(0)) before the next instruction that reuses its address
// registers, i.e. at once; an asm load is outside its bookkeeping and is
// waited for by FEA_VMEM_DRAIN() where the pipeline wants it.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_base)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_base) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void *)p;
}

#define FEA_LDS_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define FEA_VMEM_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

template <bool DOK, bool DOF, bool DBG>
__global__ __launch_bounds__(64)
void k_assemble_run(VisitArgs A, int run_len)
{
  // coordinates of the chunk's nodes, four 16-byte pieces per node, each piece
  // lane-linear as global_load_lds writes it: (x0,x1) (x2,-) (X0,X1) (X2,-)
  __shared__ double2 sC[4][FEA_VISIT_MAX_NODES];
  __shared__ double sK[DOK ? FEA_ACHUNK_BLOCKS * 9 + 2 : 2];
  __shared__ double sF[2][FEA_ACHUNK_ROWS * 3 + 3];
  __shared__ int sRow[FEA_ACHUNK_ROWS + 1];
  __shared__ int sDiag[FEA_ACHUNK_ROWS];
  const int lane = threadIdx.x;
  const int nruns = (A.nchunks + run_len - 1) / run_len;
  const int per = ((int)gridDim.x + 7) >> 3;
  const int ridx = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);       // XCD-aware, as above
  if (ridx >= nruns) return;
  int chunk = __builtin_amdgcn_readfirstlane(A.chunk0 + ridx * run_len);
  const int cend = min(A.chunk0 + A.nchunks, chunk + run_len);

  // ---- prologue: first chunk of the run, nothing to hide behind
  VisitDesc d = A.desc[chunk];
  {
    const int vn0 = A.vnode[(size_t)chunk * FEA_VISIT_MAX_NODES + lane];
    if (lane < d.nnode) {
      const double *gx = A.x + (size_t)vn0 * 4, *gX = A.X0 + (size_t)vn0 * 4;
      glds16(gx, lds_addr(&sC[0][0]));
      glds16(gx + 2, lds_addr(&sC[1][0]));
      glds16(gX, lds_addr(&sC[2][0]));
      glds16(gX + 2, lds_addr(&sC[3][0]));
    }
  }
  uint2 rec = make_uint2(0, 0);
  if (lane < d.nvisit) rec = A.vrec[d.visit_off + lane];
  {
    const int nrows = d.r1 - d.r0;
    if (lane <= nrows) sRow[lane] = A.rowptr[d.r0 + lane] - d.b0;
    if (lane < nrows) sDiag[lane] = A.diag[d.r0 + lane] - d.b0;
  }
  if (DOK)
    for (int t = lane; t < FEA_ACHUNK_BLOCKS * 9 + 2; t += 64) sK[t] = 0.0;
  if (DOF)
    for (int t = lane; t < 2 * (FEA_ACHUNK_ROWS * 3 + 3); t += 64) (&sF[0][0])[t] = 0.0;
  FEA_VMEM_DRAIN();
  FEA_LDS_ORDER();

  unsigned long long a_rounds = 0, a_tail = 0, a_drain = 0, ta = 0, tb = 0, tc = 0, td = 0;
  for (;;) {
    if (DBG && (A.dbg & 4)) ta = __builtin_amdgcn_s_memtime();
    const bool more = chunk + 1 < cend;
    const int nrows = d.r1 - d.r0;
    const int odd = d.b0 & 1;            // LDS and HBM agree on 16-byte alignment in the write-out
    double *sKt = sK + odd;
    // next chunk: node list and descriptor, in flight during the passes.  All
    // prefetches are unconditional loads from clamped indices -- a load under
    // a branch makes the compiler wait for it at the join.
    const int cn = more ? chunk + 1 : chunk;
    const int vn1 = A.vnode[(size_t)cn * FEA_VISIT_MAX_NODES + lane];
    // its rows start where this chunk's end (d.r1): row pointers and diagonal positions, raw
    const int rrow = min(d.r1 + min(lane, FEA_ACHUNK_ROWS), A.nrows_total);
    const int rp = A.rowptr[rrow];
    const int dg = A.diag[min(rrow, A.nrows_total - 1)];
    // (descriptor: an explicit scalar load -- after the first store the compiler
    // falls back to a vector load + readfirstlane and waits for it on the spot)
    v8i dnv;
    asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(dnv) : "s"(A.desc + cn) : "memory");

    for (int p = 0; p < d.nvisit; p += 64) {
      // records of the next pass -- of this chunk, or the first of the next one
      // (the visits of consecutive chunks are consecutive in vrec)
      const int gi = d.visit_off + min(p + 64, d.nvisit) + lane;
      const uint2 nxt = A.vrec[min(gi, A.nvisits - 1)];
      if (p + lane < d.nvisit && rec.y != 0xFFFFFFFFu) {
        const unsigned ids = rec.x, sl = rec.y;
        const int n0 = ids & 255u, n1 = (ids >> 8) & 255u, n2 = (ids >> 16) & 255u, n3 = ids >> 24;
        const int nd[4] = {n0, n1, n2, n3};
        double xe[4][3], Xe[4][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double2 a = sC[0][nd[k]], c2 = sC[2][nd[k]];
          xe[k][0] = a.x; xe[k][1] = a.y; xe[k][2] = sC[1][nd[k]].x;
          Xe[k][0] = c2.x; Xe[k][1] = c2.y; Xe[k][2] = sC[3][nd[k]].x;
        }
        GPState<4> s;
        gp_state<4, true, false>(xe, Xe, A.tab, 0, A.model, A.lambda, A.mu, s);
        if (!(((sl & 1u) ? -s.detJ : s.detJ) > 0.0) && DOK) {   // rare: count it from its lowest-numbered node only
          const int *gn = A.vnode + (size_t)chunk * FEA_VISIT_MAX_NODES;
          const int g0 = gn[n0];
          if (g0 < gn[n1] && g0 < gn[n2] && g0 < gn[n3]) atomicAdd(A.bad, 1);
        }
        if (s.detJ != 0.0) {                               // fea_solver.c:697: no gradient otherwise
          const double ga[3] = {s.g[0][0], s.g[0][1], s.g[0][2]};
          RowVecs rv;
          row_vectors(ga, s.sig, s.l1, s.m1, s.vol, rv);
          if (DOF) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
              __hip_atomic_fetch_add(&sF[lane & 1][n0 * 3 + i], -rv.s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
          if (DOK) {
#pragma unroll
            for (int k = 1; k < 4; ++k) {
              double blk[9];
              block_row(rv, s.g[k], blk);
              double *dst = sKt + (int)((sl >> (8 * k)) & 255u) * 9;
              if (DBG && (A.dbg & 1)) dst = sKt + lane * 9;
#pragma unroll
              for (int q = 0; q < 9; ++q)
                __hip_atomic_fetch_add(dst + q, blk[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
      }
      rec = nxt;
    }
    // every read of the coordinate tile has returned, and so has the descriptor
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dnv) : : "memory");
    if (DBG && (A.dbg & 4)) tb = __builtin_amdgcn_s_memtime();
    VisitDesc dn;
    dn.r0 = dnv[0]; dn.r1 = dnv[1]; dn.b0 = dnv[2]; dn.nb = dnv[3];
    dn.node_off = dnv[4]; dn.nnode = dnv[5]; dn.visit_off = dnv[6]; dn.nvisit = dnv[7];

    // The tail phases index by `tl`, a copy of the lane id the compiler cannot
    // see through: otherwise their address arithmetic is hoisted out of the
    // chunk loop and stays live across the passes (+16 VGPRs: one wave less per SIMD).
    int tl = lane;
    asm volatile("" : "+v"(tl));
    // coordinates of the next chunk: HBM -> LDS behind the rest of this one
    asm volatile("" : : "v"(vn1), "v"(rp), "v"(dg));      // the compiler waits for them here, not behind the LDS-DMA loads
    if (more) {
      if (tl < dn.nnode) {
        const double *gx = A.x + (size_t)vn1 * 4, *gX = A.X0 + (size_t)vn1 * 4;
        glds16(gx, lds_addr(&sC[0][0]));
        glds16(gx + 2, lds_addr(&sC[1][0]));
        glds16(gX, lds_addr(&sC[2][0]));
        glds16(gX + 2, lds_addr(&sC[3][0]));
      }
    }

    if (DOK) {
      // K_aa = -sum_{b != a} K_ab (shape functions sum to one); the diagonal
      // block was never added to, so the whole row is summed
      for (int t = tl; t < nrows * 9; t += 64) {
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
      FEA_LDS_ORDER();
      // stream the finished rows out (16-byte LDS reads and HBM stores) and clear the tile behind them
      double *Kd = A.K + (size_t)d.b0 * 9;
      const int total = d.nb * 9;
      if (odd && tl == 0) { Kd[0] = sKt[0]; sKt[0] = 0.0; }
      const int npair = (total - odd) >> 1;
      for (int t = tl; t < npair; t += 64) {
        const int j = odd + 2 * t;
        *reinterpret_cast<double2 *>(Kd + j) = *reinterpret_cast<const double2 *>(sKt + j);
        *reinterpret_cast<double2 *>(sKt + j) = make_double2(0.0, 0.0);
      }
      if (((total - odd) & 1) && tl == 0) { Kd[total - 1] = sKt[total - 1]; sKt[total - 1] = 0.0; }
    }
    if (DOF) {
      double *fd = A.f + (size_t)d.r0 * 3;
      for (int t = tl; t < nrows * 3; t += 64) {
        fd[t] = sF[0][t] + sF[1][t];
        sF[0][t] = 0.0; sF[1][t] = 0.0;
      }
    }
    if (DBG && (A.dbg & 4)) { FEA_LDS_ORDER(); tc = __builtin_amdgcn_s_memtime(); a_rounds += tb - ta; a_tail += tc - tb; }
    if (!more) break;
    FEA_LDS_ORDER();
    {
      const int nr1 = dn.r1 - dn.r0;
      if (tl <= nr1) sRow[tl] = rp - dn.b0;
      if (tl < nr1) sDiag[tl] = dg - dn.b0;
    }
    FEA_VMEM_DRAIN();                    // the coordinate tile of the next chunk has landed
    FEA_LDS_ORDER();
    if (DBG && (A.dbg & 4)) { td = __builtin_amdgcn_s_memtime(); a_drain += td - tc; }
    d = dn;
    ++chunk;
  }
  if (DBG && (A.dbg & 4) && lane == 0) {
    unsigned long long *o = A.stamps + (size_t)ridx * 8;
    o[0] = a_rounds; o[1] = a_tail; o[2] = a_drain; o[3] = 0; o[4] = a_rounds + a_tail + a_drain;
  }
}

int launch_assemble_visit(feahip_ctx *c, bool doK, bool doF, bool pipelined)
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
  static int run_len = -1;           // chunks per wave of the pipelined kernel (FEAHIP_RUN: tuning only)
  if (run_len < 0) { const char *e = getenv("FEAHIP_RUN"); run_len = e && atoi(e) > 0 ? atoi(e) : 6; }
  if (pipelined && !(A.dbg & ~5)) {
    const int nruns = (c->nachunks_local + run_len - 1) / run_len;
    const dim3 rgrid((nruns + 7) & ~7), rblk(64);
#ifdef FEAHIP_DEBUG
    if (A.dbg) {
      if (doK && doF) hipLaunchKernelGGL((k_assemble_run<true, true, true>), rgrid, rblk, 0, c->stream, A, run_len);
      else if (doK)   hipLaunchKernelGGL((k_assemble_run<true, false, true>), rgrid, rblk, 0, c->stream, A, run_len);
      else            hipLaunchKernelGGL((k_assemble_run<false, true, true>), rgrid, rblk, 0, c->stream, A, run_len);
    } else
#endif
    {
      if (doK && doF) hipLaunchKernelGGL((k_assemble_run<true, true, false>), rgrid, rblk, 0, c->stream, A, run_len);
      else if (doK)   hipLaunchKernelGGL((k_assemble_run<true, false, false>), rgrid, rblk, 0, c->stream, A, run_len);
      else            hipLaunchKernelGGL((k_assemble_run<false, true, false>), rgrid, rblk, 0, c->stream, A, run_len);
    }
    FEA_HIP_CHECK(c, hipGetLastError());
#ifdef FEAHIP_DEBUG
    if (A.dbg & 4) {                                  // diagnostic path: phase shares, never a timing
      static int printed = 0;
      (void)hipStreamSynchronize(c->stream);
      if (printed++ == 3) {
        std::vector<unsigned long long> h((size_t)nruns * 8);
        (void)hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost);
        double sum[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < nruns; ++i)
          for (int q = 0; q < 5; ++q) sum[q] += (double)h[(size_t)i * 8 + q];
        fprintf(stderr, "[feahip run stamps] run_len=%d  mean cycles per chunk: passes %.0f  tail %.0f  drain %.0f  total %.0f\n",
                run_len, sum[0] / c->nachunks_local, sum[1] / c->nachunks_local, sum[2] / c->nachunks_local, sum[4] / c->nachunks_local);
      }
    }
#endif
    return FEAHIP_OK;
  }
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
