// kernels_patch.hip -- PATCH assembly of linear tetrahedra (the 1M / 10M-tet
// BASELINE configurations): stiffness (+ residual) with no atomics at all.
//
// One wavefront owns a patch = a chunk of consecutive block rows, and works
// entirely out of LDS:
//   1. the coordinates (x, X0) of every node the patch's elements touch are
//      gathered into LDS once (patch-local node ids are 16-bit);
//   2. in batches of 64, every element that touches the patch gets its state
//      evaluated ONCE by one lane (J, grad N, F, sigma, tangent coefficients
//      -- fem_device.h) and parked in an LDS tile as
//      { g_b, t_b = vol (m1 g_b + sigma g_b), vol l1, vol m1 };
//   3. lane k owns the off-diagonal blocks k and k+64 of the patch and walks
//      the precomputed list of (element, a, b) contributions of each, reading
//      the tile and accumulating K_ab = g_a (x) vl g_b + vm g_b (x) g_a
//      + (g_a . t_b) I in registers, in a fixed order (bitwise reproducible);
//   4. the diagonal block of every row is minus the sum of the row's other
//      blocks (shape functions sum to one), the residual of a row node is the
//      sum of -(t_a - vm g_a) over its elements; the finished rows leave LDS as
//      one contiguous, coalesced stream.  Every CSR value is written once.
// Replaces fea_solver.c:873-883 (+ :863-870) for TETRAHEDRA4 meshes.
#include "fem_device.h"

struct PatchArgs {
  int patch0, npatches, model;
  double lambda, mu;
  const ElemTable *tab;
  const PatchDesc *desc;
  const int *pnode;
  const uint16_t *pelem, *pent, *pbptr;
  const double *X0, *x;          // [N][4]
  const int *rowptr, *diag;
  double *K, *f;
  int *bad;
};

#define PT_STRIDE 26             // doubles per element in the tile

template <bool DOF>
__device__ __forceinline__ void consume(const uint16_t *sE, const double *sT, int &pos, const int end,
                                        const int base, double (&acc)[9], double (&fa)[3])
{
  while (pos < end) {
    const unsigned w = sE[pos];
    const int el = (int)(w & 0x7FFu) - base;
    if (el >= 64) break;                       // belongs to a later batch
    const int la = (w >> 11) & 3, lb = (w >> 13) & 3;
    const double *T = sT + el * PT_STRIDE;
    const double ga0 = T[la * 3], ga1 = T[la * 3 + 1], ga2 = T[la * 3 + 2];
    const double gb0 = T[lb * 3], gb1 = T[lb * 3 + 1], gb2 = T[lb * 3 + 2];
    const double tb0 = T[12 + lb * 3], tb1 = T[12 + lb * 3 + 1], tb2 = T[12 + lb * 3 + 2];
    const double vl = T[24], vm = T[25];
    const double h0 = vl * gb0, h1 = vl * gb1, h2 = vl * gb2;
    const double m0 = vm * gb0, m1 = vm * gb1, m2 = vm * gb2;
    const double d = ga0 * tb0 + ga1 * tb1 + ga2 * tb2;
    acc[0] += ga0 * h0 + ga0 * m0 + d; acc[1] += ga0 * h1 + ga1 * m0;     acc[2] += ga0 * h2 + ga2 * m0;
    acc[3] += ga1 * h0 + ga0 * m1;     acc[4] += ga1 * h1 + ga1 * m1 + d; acc[5] += ga1 * h2 + ga2 * m1;
    acc[6] += ga2 * h0 + ga0 * m2;     acc[7] += ga2 * h1 + ga1 * m2;     acc[8] += ga2 * h2 + ga2 * m2 + d;
    if (DOF && (w >> 15)) {
      fa[0] -= T[12 + la * 3] - vm * ga0;
      fa[1] -= T[12 + la * 3 + 1] - vm * ga1;
      fa[2] -= T[12 + la * 3 + 2] - vm * ga2;
    }
    ++pos;
  }
}

template <bool DOF>
__global__ __launch_bounds__(64)
void k_assemble_patch(PatchArgs A)
{
  __shared__ double sC[FEA_PATCH_MAX_NODES * 6];
  __shared__ double sT[64 * PT_STRIDE];        // element tile; later the K tile + row partials of f
  __shared__ uint16_t sE[FEA_PATCH_MAX_ENTRIES];
  __shared__ uint16_t sB[FEA_CHUNK_BLOCKS + 2];
  const int lane = threadIdx.x;
  const PatchDesc d = A.desc[A.patch0 + blockIdx.x];

  for (int i = lane; i < d.nent; i += 64) sE[i] = A.pent[d.ent_off + i];
  for (int i = lane; i <= d.nb; i += 64) sB[i] = A.pbptr[d.bptr_off + i];
  for (int i = lane; i < d.nnode; i += 64) {
    const size_t n = (size_t)A.pnode[d.node_off + i];
    const double2 a0 = *reinterpret_cast<const double2 *>(A.x + n * 4);
    const double2 a1 = *reinterpret_cast<const double2 *>(A.x + n * 4 + 2);
    const double2 c0 = *reinterpret_cast<const double2 *>(A.X0 + n * 4);
    const double2 c1 = *reinterpret_cast<const double2 *>(A.X0 + n * 4 + 2);
    double *o = sC + i * 6;
    o[0] = a0.x; o[1] = a0.y; o[2] = a1.x; o[3] = c0.x; o[4] = c0.y; o[5] = c1.x;
  }
  __syncthreads();

  int pos0 = 0, end0 = 0, pos1 = 0, end1 = 0;
  if (lane < d.nb) { pos0 = sB[lane]; end0 = sB[lane + 1]; }
  if (lane + 64 < d.nb) { pos1 = sB[lane + 64]; end1 = sB[lane + 65]; }
  double acc0[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double f0[3] = {0, 0, 0}, f1[3] = {0, 0, 0};

  for (int base = 0; base < d.nelem; base += 64) {
    const int e = base + lane;
    if (e < d.nelem) {
      const ushort4 id = *reinterpret_cast<const ushort4 *>(A.pelem + (size_t)(d.elem_off + e) * 4);
      const int nd[4] = {id.x, id.y, id.z, id.w};
      double xe[4][3], Xe[4][3];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double *cc = sC + nd[k] * 6;
#pragma unroll
        for (int j = 0; j < 3; ++j) { xe[k][j] = cc[j]; Xe[k][j] = cc[3 + j]; }
      }
      GPState<4> s;
      gp_state<4, true>(xe, Xe, A.tab, 0, A.model, A.lambda, A.mu, s);
      double *T = sT + lane * PT_STRIDE;
      if (s.detJ == 0.0) {                     // no gradient, no contribution (fea_solver.c:697)
#pragma unroll
        for (int q = 0; q < PT_STRIDE; ++q) T[q] = 0.0;
      } else {
        const double vm = s.vol * s.m1;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            T[b * 3 + i] = s.g[b][i];
            T[12 + b * 3 + i] = vm * s.g[b][i] +
                s.vol * (s.sig[i][0] * s.g[b][0] + s.sig[i][1] * s.g[b][1] + s.sig[i][2] * s.g[b][2]);
          }
        T[24] = s.vol * s.l1;
        T[25] = vm;
      }
      if (!(s.detJ > 0.0)) {                   // report each bad element once: by the patch owning its node 0
        const int g0 = A.pnode[d.node_off + nd[0]];
        if (g0 >= d.r0 && g0 < d.r1) atomicAdd(A.bad, 1);
      }
    }
    __syncthreads();
    consume<DOF>(sE, sT, pos0, end0, base, acc0, f0);
    consume<DOF>(sE, sT, pos1, end1, base, acc1, f1);
    __syncthreads();
  }

  // finished off-diagonal blocks -> K tile (aliases the element tile)
  double *tK = sT, *tF = sT + FEA_CHUNK_BLOCKS * 9;
  if (lane < d.nb) {
#pragma unroll
    for (int q = 0; q < 9; ++q) tK[lane * 9 + q] = acc0[q];
    if (DOF) { tF[lane * 3] = f0[0]; tF[lane * 3 + 1] = f0[1]; tF[lane * 3 + 2] = f0[2]; }
  }
  if (lane + 64 < d.nb) {
#pragma unroll
    for (int q = 0; q < 9; ++q) tK[(lane + 64) * 9 + q] = acc1[q];
    if (DOF) { tF[(lane + 64) * 3] = f1[0]; tF[(lane + 64) * 3 + 1] = f1[1]; tF[(lane + 64) * 3 + 2] = f1[2]; }
  }
  __syncthreads();
  const int nrows = d.r1 - d.r0;
  for (int t = lane; t < nrows * 9; t += 64) {
    const int r = d.r0 + t / 9, q = t % 9;
    const int kb = A.rowptr[r] - d.b0, ke = A.rowptr[r + 1] - d.b0, kd = A.diag[r] - d.b0;
    double a = 0;
    for (int k = kb; k < ke; ++k) a += (k == kd) ? 0.0 : tK[k * 9 + q];
    tK[kd * 9 + q] = -a;
  }
  if (DOF) {
    for (int t = lane; t < nrows * 3; t += 64) {
      const int r = d.r0 + t / 3, i = t % 3;
      const int kb = A.rowptr[r] - d.b0, ke = A.rowptr[r + 1] - d.b0;
      double a = 0;
      for (int k = kb; k < ke; ++k) a += tF[k * 3 + i];
      A.f[(size_t)d.r0 * 3 + t] = a;
    }
  }
  __syncthreads();
  double *Kd = A.K + (size_t)d.b0 * 9;
  for (int t = lane; t < d.nb * 9; t += 64) Kd[t] = tK[t];
}

int launch_assemble_patch(feahip_ctx *c, bool doF)
{
  PatchArgs A;
  A.patch0 = c->chunk0; A.npatches = c->nchunks_local; A.model = c->model;
  A.lambda = c->lambda; A.mu = c->mu; A.tab = c->d_table; A.desc = c->d_pdesc; A.pnode = c->d_pnode;
  A.pelem = c->d_pelem; A.pent = c->d_pent; A.pbptr = c->d_pbptr; A.X0 = c->d_X0; A.x = c->d_x;
  A.rowptr = c->d_rowptr; A.diag = c->d_diag; A.K = c->d_K; A.f = c->d_f; A.bad = c->d_flag + 1;
  if (c->nchunks_local <= 0) return FEAHIP_OK;
  if (doF) hipLaunchKernelGGL(k_assemble_patch<true>, dim3(c->nchunks_local), dim3(64), 0, c->stream, A);
  else hipLaunchKernelGGL(k_assemble_patch<false>, dim3(c->nchunks_local), dim3(64), 0, c->stream, A);
  FEA_HIP_CHECK(c, hipGetLastError());
  return FEAHIP_OK;
}
