// kernels_quad.hip -- stiffness + residual assembly of 10-node tetrahedra
// (the reference's element, fea_solver.c:873-883 / :887-1068 / :1072-1114)
// with the Gauss-point state shared between the visits of a chunk.
//
// The generic row-owner kernel (kernels_assemble.hip) gives a lane one (row
// node, element) visit: for a 10-node element that lane evaluates the state of
// every Gauss point itself (10x redundantly over the element's nodes), adds
// 9 blocks x 9 doubles to the LDS tile PER Gauss point (405 ds_add_f64 per
// visit at G = 5), and a chunk of 2-4 rows has only ~20 visits for 64 lanes.
// Here a workgroup of four waves owns a chunk of block rows (four, so that the
// 37 KB of tiles below are shared by enough waves to keep the SIMDs busy: one
// wave per chunk left 1.2 waves per SIMD waiting on their own dependent FP64
// chains, VALU 32 % busy), and the work is cut twice:
//   phase 1  lane <-> (element of the chunk, Gauss point): the state -- inverse
//            Jacobian of the current configuration, stress, tangent
//            coefficients, w|det J| -- once per chunk, into an LDS tile
//            (structure of arrays: lane-linear writes, broadcast reads).  Node
//            coordinates, the elements' local node ids and the shape-function
//            table of the batch are staged in LDS first: phase 1 touches no
//            global memory;
//   phase 2  lane <-> (row node a, element, column node b): the spatial
//            gradients g_a, g_b = J^-T dN from the state entry, K_ab summed
//            over the Gauss points of the batch in registers, then 9
//            ds_add_f64 into the wave's K tile; the 9 lanes of a visit read
//            the same state entry.
// Gauss points go through in batches of as many as fit 128 state entries, so
// the 27-point rule runs in the same LDS.  The diagonal blocks come from the
// row sums (shape functions sum to one) and the rows are streamed out: every
// CSR value written once.
#include "fem_device.h"
#include <cstdlib>
#include <vector>

struct QuadArgs {
  int chunk0, nchunks, model, G;
  double lambda, mu;
  const ElemTable *tab;
  const QuadDesc *desc;
  const uint32_t *qelem, *qpair;
  const int *qnode;
  const double *X0, *x;
  const int *rowptr, *diag;
  double *K, *f;
  int *bad;
  unsigned long long *stamps;   // diagnostic build only
};
#ifdef FEAHIP_DEBUG
#define Q_STAMP(i) do { if (A.stamps) { const unsigned long long _t = __builtin_amdgcn_s_memtime(); qa[i] += _t - qt; qt = _t; } } while (0)
#else
#define Q_STAMP(i) do { } while (0)
#endif

// rows of the state tile
#define QS_JI 0       // 9: inverse Jacobian Ji[i][m] at 3i+m   (g_a[i] = sum_m Ji[i][m] dN[m][a])
#define QS_SIG 9      // 6: 00 01 02 11 12 22
#define QS_L1 15
#define QS_M1 16
#define QS_VOL 17
#define QS_ROWS 18
#define QUAD_NT 256                    // threads per chunk
#define QUAD_ENTRIES 128              // (element, Gauss point) entries per batch
#define FEA_QUAD_BATCH_GAUSS 8        // Gauss points per batch at most (size of the table slice in LDS)

// workgroup barrier that orders LDS only (kernels_gather.hip): __syncthreads() would also drain the prefetches
#define Q_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)

// what a thread holds of the NEXT chunk while the current one is worked on
struct QNext {
  int hword;                // word (lane % 10) of the chunk's descriptor
  int node;                 // node slot `lane` of the chunk
  uint32_t eword, pw0, pw1; // element word `lane`, pair words `lane` and `lane + QUAD_NT`
  int rp, dg;               // rowptr / diag of row `lane` of the chunk
};

// Persistent form (round 2; the stamps of the one-chunk-per-workgroup kernel showed 37 % of a chunk's 22 k cycles
// in front of its first FMA -- descriptor, then node list, then coordinates: three dependent memory round trips
// -- with chunks of three rows and sixteen elements).  A workgroup walks a run of consecutive chunks and keeps the
// next chunk's loads in flight under the current chunk's batches:
//   top of chunk i      descriptor of chunk i+2 (a vector load, read back with v_readlane when chunk i ends); node
//                       ids, element words, pair words, row pointers of chunk i+1 (its descriptor came one chunk ago)
//   after the last      the coordinate tile is dead: coordinates of chunk i+1 by the node ids that have arrived,
//   state stage         into registers
//   before the stores   everything prefetched is "used" (no first use behind the store burst: kernels_gather.hip)
//   after the stores    the registers move into the LDS tiles of chunk i+1
template <int NPE, bool DOF>
__global__ __launch_bounds__(QUAD_NT)
void k_assemble_quad(QuadArgs A, int run_len)
{
  __shared__ double sS[QS_ROWS][QUAD_ENTRIES];
  __shared__ double sx[FEA_QUAD_NODES * 3], sX[FEA_QUAD_NODES * 3];     // current / reference coordinates of the chunk's nodes
  __shared__ uint32_t sE[FEA_QUAD_ELEMS * 3];
  __shared__ double sTw[FEA_QUAD_BATCH_GAUSS];                          // table slice of the batch (whole table if it fits)
  __shared__ double sTd[FEA_QUAD_BATCH_GAUSS][3][NPE];
  __shared__ double sK[FEA_QUAD_BLOCKS * 9 + 2];
  __shared__ double sF[FEA_CHUNK_ROWS * 3];
  __shared__ int sRow[FEA_CHUNK_ROWS + 1];
  __shared__ int sDiag[FEA_CHUNK_ROWS];
  const int lane = threadIdx.x;
  const int nruns = (A.nchunks + run_len - 1) / run_len;
  const int per = ((int)gridDim.x + 7) >> 3;
  const int ridx = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);     // XCD-contiguous run order (kernels_visit.hip)
  if (ridx >= nruns) return;
  int chunk = A.chunk0 + ridx * run_len;
  const int cend = min(A.chunk0 + A.nchunks, chunk + run_len);
#ifdef FEAHIP_DEBUG
  unsigned long long qa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, qt = __builtin_amdgcn_s_memtime();
#endif
  const bool whole_table = A.G <= FEA_QUAD_BATCH_GAUSS;
  if (whole_table) {
    for (int t = lane; t < A.G; t += QUAD_NT) sTw[t] = A.tab->w[t];
    for (int t = lane; t < A.G * 3 * NPE; t += QUAD_NT) (&sTd[0][0][0])[t] = A.tab->dN[t / (3 * NPE)][(t / NPE) % 3][t % NPE];
  }

  // ---- prologue: the first chunk of the run, nothing to hide behind
  QuadDesc d = A.desc[chunk];
  QuadDesc dn = A.desc[min(chunk + 1, cend - 1)];
  uint32_t pw0 = 0, pw1 = 0;
  {
    if (lane < d.nnode) {
      const size_t n = (size_t)A.qnode[(size_t)d.node_off + lane];
      const double2 a0 = *reinterpret_cast<const double2 *>(A.x + n * 4), c0 = *reinterpret_cast<const double2 *>(A.X0 + n * 4);
      const double a2 = A.x[n * 4 + 2], c2 = A.X0[n * 4 + 2];
      sx[lane * 3] = a0.x; sx[lane * 3 + 1] = a0.y; sx[lane * 3 + 2] = a2;
      sX[lane * 3] = c0.x; sX[lane * 3 + 1] = c0.y; sX[lane * 3 + 2] = c2;
    }
    if (lane < d.nelem * 3) sE[lane] = A.qelem[(size_t)d.elem_off * 3 + lane];
    if (lane <= d.r1 - d.r0) sRow[lane] = A.rowptr[d.r0 + lane] - d.b0;
    if (lane < d.r1 - d.r0) sDiag[lane] = A.diag[d.r0 + lane] - d.b0;
    if (lane < d.npair) pw0 = A.qpair[(size_t)d.pair_off + lane];
    if (lane + QUAD_NT < d.npair) pw1 = A.qpair[(size_t)d.pair_off + lane + QUAD_NT];
  }
  asm volatile("" : : "v"(pw0), "v"(pw1));              // nothing pending when the loop is entered
  Q_STAMP(0);

  for (;;) {
    const bool more = chunk + 1 < cend;
    const int nrows = d.r1 - d.r0;
    const int odd = d.b0 & 1;
    double *sKt = sK + odd;
    for (int t = lane; t < d.nb * 9 + odd; t += QUAD_NT) sK[t] = 0.0;
    if (DOF)
      for (int t = lane; t < nrows * 3; t += QUAD_NT) sF[t] = 0.0;

    // ---- next chunk's loads (clamped indices instead of branches: a load under a branch is waited for at the join)
    QNext nx;
    {
      const int c2 = min(chunk + 2, cend - 1);
      nx.hword = reinterpret_cast<const int *>(A.desc + c2)[min(lane & 63, 9)];     // word k in lane k of EVERY wave (v_readlane is per wave)
      nx.node = A.qnode[(size_t)dn.node_off + min(lane, dn.nnode - 1)];
      nx.eword = A.qelem[(size_t)dn.elem_off * 3 + min(lane, dn.nelem * 3 - 1)];
      nx.pw0 = A.qpair[(size_t)dn.pair_off + min(lane, dn.npair - 1)];
      nx.pw1 = A.qpair[(size_t)dn.pair_off + min(lane + QUAD_NT, dn.npair - 1)];
      nx.rp = A.rowptr[dn.r0 + min(lane, dn.r1 - dn.r0)];
      nx.dg = A.diag[dn.r0 + min(lane, dn.r1 - dn.r0 - 1)];
    }
    double2 nxa = make_double2(0, 0), nxc = make_double2(0, 0);
    double nxa2 = 0, nxc2 = 0;

    const int ne = d.nelem;
    int gb = QUAD_ENTRIES / ne;
    gb = gb < 1 ? 1 : (gb > A.G ? A.G : gb);
    gb = gb > FEA_QUAD_BATCH_GAUSS ? FEA_QUAD_BATCH_GAUSS : gb;
    gb = (A.G + (A.G + gb - 1) / gb - 1) / ((A.G + gb - 1) / gb);          // same number of batches, evenly filled
    for (int g0 = 0; g0 < A.G; g0 += gb) {
      const int ng = (A.G - g0 < gb) ? (A.G - g0) : gb;
      const int tb = whole_table ? g0 : 0;               // first row of the batch in the LDS table
      Q_BARRIER();                                       // the previous batch has been read (first batch: the tiles are staged)
      if (!whole_table) {
        for (int t = lane; t < ng; t += QUAD_NT) sTw[t] = A.tab->w[g0 + t];
        for (int t = lane; t < ng * 3 * NPE; t += QUAD_NT) (&sTd[0][0][0])[t] = A.tab->dN[g0 + t / (3 * NPE)][(t / NPE) % 3][t % NPE];
        __syncthreads();
      }
      Q_STAMP(1);
      // ---- phase 1: state of (element, Gauss point) entries
      if (lane < ne * ng) {
        const int el = lane / ng, tg = lane % ng;
        const uint32_t e0 = sE[el * 3], e1 = sE[el * 3 + 1], e2 = sE[el * 3 + 2];
        const int nd[10] = {(int)(e0 & 255u), (int)((e0 >> 8) & 255u), (int)((e0 >> 16) & 255u), (int)(e0 >> 24),
                            (int)(e1 & 255u), (int)((e1 >> 8) & 255u), (int)((e1 >> 16) & 255u), (int)(e1 >> 24),
                            (int)(e2 & 255u), (int)((e2 >> 8) & 255u)};
        // J = dx/dxi, M = dX/dxi (both as sum_k dN[.][k] (x) coordinates of node k)
        double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
        for (int k = 0; k < NPE; ++k) {
          const double xc[3] = {sx[nd[k] * 3], sx[nd[k] * 3 + 1], sx[nd[k] * 3 + 2]};
          const double Xc[3] = {sX[nd[k] * 3], sX[nd[k] * 3 + 1], sX[nd[k] * 3 + 2]};
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const double dn_ = sTd[tb + tg][i][k];
#pragma unroll
            for (int j = 0; j < 3; ++j) { J[i][j] += dn_ * xc[j]; M[i][j] += dn_ * Xc[j]; }
          }
        }
        double Ji[3][3], detJ;
        fd_inv3(J, Ji, detJ);
        // F^-1 = sum_k X_k (x) g_k with g_k = Ji dN_k  =>  Finv[i][j] = sum_m M[m][i] Ji[j][m]
        double Fi[3][3], F[3][3], detFi;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) Fi[i][j] = M[0][i] * Ji[j][0] + M[1][i] * Ji[j][1] + M[2][i] * Ji[j][2];
        fd_inv3(Fi, F, detFi);
        double sig[3][3], l1, m1;
        fd_constitutive(F, A.model, A.lambda, A.mu, sig, l1, m1);
        if (!(detJ > 0.0) && ((e2 >> 16) & 1u)) atomicAdd(A.bad, 1);       // once per (element, Gauss point) of the mesh
        const bool dead = detJ == 0.0;                                      // fea_solver.c:697: no gradient then
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int m = 0; m < 3; ++m) sS[QS_JI + 3 * i + m][lane] = dead ? 0.0 : Ji[i][m];
        sS[QS_SIG + 0][lane] = dead ? 0.0 : sig[0][0]; sS[QS_SIG + 1][lane] = dead ? 0.0 : sig[0][1];
        sS[QS_SIG + 2][lane] = dead ? 0.0 : sig[0][2]; sS[QS_SIG + 3][lane] = dead ? 0.0 : sig[1][1];
        sS[QS_SIG + 4][lane] = dead ? 0.0 : sig[1][2]; sS[QS_SIG + 5][lane] = dead ? 0.0 : sig[2][2];
        sS[QS_L1][lane] = dead ? 0.0 : l1;
        sS[QS_M1][lane] = dead ? 0.0 : m1;
        sS[QS_VOL][lane] = dead ? 0.0 : sTw[tb + tg] * fabs(detJ);
      }
      Q_STAMP(2);
      Q_BARRIER();
      Q_STAMP(3);
      if (g0 + gb >= A.G) {
        // the coordinate tile is dead from here on: request the next chunk's coordinates (its node ids were
        // requested at the top of this chunk, at least one state stage ago)
        const size_t n = (size_t)nx.node;
        nxa = *reinterpret_cast<const double2 *>(A.x + n * 4); nxa2 = A.x[n * 4 + 2];
        nxc = *reinterpret_cast<const double2 *>(A.X0 + n * 4); nxc2 = A.X0[n * 4 + 2];
      }
      // ---- phase 2: blocks of (row node, element, column node) pairs over the batch, then into the K tile
      for (int p0 = lane; p0 < d.npair; p0 += QUAD_NT) {
        const uint32_t w = p0 < QUAD_NT ? pw0 : (p0 < 2 * QUAD_NT ? pw1 : A.qpair[(size_t)d.pair_off + p0]);
        const int el = w & 63u, la = (w >> 6) & 15u, lb = (w >> 10) & 15u;
        double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, fa[3] = {0, 0, 0};
        for (int t = 0; t < ng; ++t) {
          const int ent = el * ng + t;
          double Ji[3][3], ga[3], gbv[3], sig[3][3];
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int m = 0; m < 3; ++m) Ji[i][m] = sS[QS_JI + 3 * i + m][ent];
          const double da[3] = {sTd[tb + t][0][la], sTd[tb + t][1][la], sTd[tb + t][2][la]};
          const double db[3] = {sTd[tb + t][0][lb], sTd[tb + t][1][lb], sTd[tb + t][2][lb]};
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            ga[i] = Ji[i][0] * da[0] + Ji[i][1] * da[1] + Ji[i][2] * da[2];
            gbv[i] = Ji[i][0] * db[0] + Ji[i][1] * db[1] + Ji[i][2] * db[2];
          }
          sig[0][0] = sS[QS_SIG + 0][ent]; sig[0][1] = sig[1][0] = sS[QS_SIG + 1][ent]; sig[0][2] = sig[2][0] = sS[QS_SIG + 2][ent];
          sig[1][1] = sS[QS_SIG + 3][ent]; sig[1][2] = sig[2][1] = sS[QS_SIG + 4][ent]; sig[2][2] = sS[QS_SIG + 5][ent];
          RowVecs rv;
          row_vectors(ga, sig, sS[QS_L1][ent], sS[QS_M1][ent], sS[QS_VOL][ent], rv);
          double blk[9];
          block_row(rv, gbv, blk);
#pragma unroll
          for (int q = 0; q < 9; ++q) acc[q] += blk[q];
          if (DOF) {
#pragma unroll
            for (int i = 0; i < 3; ++i) fa[i] -= rv.s[i];
          }
        }
        double *dst = sKt + (int)((w >> 14) & 255u) * 9;
#pragma unroll
        for (int q = 0; q < 9; ++q)
          __hip_atomic_fetch_add(dst + q, acc[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (DOF && ((w >> 26) & 1u)) {
          const int rl = (w >> 22) & 15u;
#pragma unroll
          for (int i = 0; i < 3; ++i)
            __hip_atomic_fetch_add(&sF[rl * 3 + i], fa[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
    Q_STAMP(4);
    Q_BARRIER();
    Q_STAMP(5);
    // K_aa = -sum_{b != a} K_ab: the diagonal block was never added to
    for (int t = lane; t < nrows * 9; t += QUAD_NT) {
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
    // everything prefetched is used HERE, before the row stores are issued (in-order memory counter: a first use
    // behind the stores would wait for them)
    asm volatile("" : : "v"(nx.hword), "v"(nx.node), "v"(nx.eword), "v"(nx.pw0), "v"(nx.pw1), "v"(nx.rp), "v"(nx.dg),
                 "v"(nxa.x), "v"(nxa.y), "v"(nxc.x), "v"(nxc.y), "v"(nxa2), "v"(nxc2));
    Q_BARRIER();
    double *Kd = A.K + (size_t)d.b0 * 9;
    const int total = d.nb * 9;
    if (odd && lane == 0) Kd[0] = sKt[0];
    const int npair2 = (total - odd) >> 1;
    for (int t = lane; t < npair2; t += QUAD_NT) {
      const int j = odd + 2 * t;
      *reinterpret_cast<double2 *>(Kd + j) = *reinterpret_cast<const double2 *>(sKt + j);
    }
    if (((total - odd) & 1) && lane == 0) Kd[total - 1] = sKt[total - 1];
    if (DOF) {
      double *fd = A.f + (size_t)d.r0 * 3;
      for (int t = lane; t < nrows * 3; t += QUAD_NT) fd[t] = sF[t];
    }
    Q_STAMP(6);
    if (!more) break;
    Q_BARRIER();                                       // the tiles of this chunk have been read
    // ---- the next chunk becomes the current one: registers -> LDS tiles, descriptor of the one after it
    d = dn;
    dn.r0 = __builtin_amdgcn_readlane(nx.hword, 0); dn.r1 = __builtin_amdgcn_readlane(nx.hword, 1);
    dn.b0 = __builtin_amdgcn_readlane(nx.hword, 2); dn.nb = __builtin_amdgcn_readlane(nx.hword, 3);
    dn.elem_off = __builtin_amdgcn_readlane(nx.hword, 4); dn.nelem = __builtin_amdgcn_readlane(nx.hword, 5);
    dn.pair_off = __builtin_amdgcn_readlane(nx.hword, 6); dn.npair = __builtin_amdgcn_readlane(nx.hword, 7);
    dn.node_off = __builtin_amdgcn_readlane(nx.hword, 8); dn.nnode = __builtin_amdgcn_readlane(nx.hword, 9);
    if (lane < d.nnode) {
      sx[lane * 3] = nxa.x; sx[lane * 3 + 1] = nxa.y; sx[lane * 3 + 2] = nxa2;
      sX[lane * 3] = nxc.x; sX[lane * 3 + 1] = nxc.y; sX[lane * 3 + 2] = nxc2;
    }
    if (lane < d.nelem * 3) sE[lane] = nx.eword;
    if (lane <= d.r1 - d.r0) sRow[lane] = nx.rp - d.b0;
    if (lane < d.r1 - d.r0) sDiag[lane] = nx.dg - d.b0;
    pw0 = nx.pw0; pw1 = nx.pw1;
    ++chunk;
  }
#ifdef FEAHIP_DEBUG
  if (A.stamps && (lane & 63) == 0) {
    unsigned long long *o = A.stamps + ((size_t)ridx * 4 + (lane >> 6)) * 8;
    for (int i = 0; i < 7; ++i) o[i] = qa[i];
  }
#endif
}

int launch_assemble_quad(feahip_ctx *c, bool doF)
{
  if (c->npe != 10) { c->err = "shared-state assembly is built for 10-node elements"; return FEAHIP_EINVAL; }
  QuadArgs A;
  A.chunk0 = 0; A.nchunks = c->quad_n; A.model = c->model; A.G = c->G;      // the maps hold this rank's chunks only
  A.lambda = c->lambda; A.mu = c->mu; A.tab = c->d_table; A.desc = c->d_qdesc; A.qelem = c->d_qelem; A.qpair = c->d_qpair;
  A.qnode = c->d_qnode; A.X0 = c->d_X0; A.x = c->d_x; A.rowptr = c->d_rowptr; A.diag = c->d_diag;
  A.K = c->d_K; A.f = c->d_f; A.bad = c->d_flag + 1; A.stamps = nullptr;
  if (c->quad_n <= 0) return FEAHIP_OK;
#ifdef FEAHIP_DEBUG
  static unsigned long long *d_stamps = nullptr;
  static int cap = 0;
  if (getenv("FEAHIP_QUAD_STAMPS")) {
    if (!d_stamps || cap < c->quad_n) { if (d_stamps) (void)hipFree(d_stamps); (void)hipMalloc((void **)&d_stamps, 8 * 32 * (size_t)c->quad_n); cap = c->quad_n; }
    (void)hipMemset(d_stamps, 0, 8 * 32 * (size_t)c->quad_n);
    A.stamps = d_stamps;
  }
#endif
  static int run_len = -1;           // chunks per workgroup run (FEAHIP_QUAD_RUN: tuning only, results unchanged)
  if (run_len < 0) { const char *e = getenv("FEAHIP_QUAD_RUN"); run_len = e && atoi(e) > 0 ? atoi(e) : 16; }
  const int nruns = (c->quad_n + run_len - 1) / run_len;
  const dim3 grid((nruns + 7) & ~7), blk(QUAD_NT);
  if (doF) hipLaunchKernelGGL((k_assemble_quad<10, true>), grid, blk, 0, c->stream, A, run_len);
  else     hipLaunchKernelGGL((k_assemble_quad<10, false>), grid, blk, 0, c->stream, A, run_len);
  FEA_HIP_CHECK(c, hipGetLastError());
#ifdef FEAHIP_DEBUG
  if (A.stamps) {
    static int calls = 0;
    if (++calls == 8) {
      (void)hipStreamSynchronize(c->stream);
      std::vector<unsigned long long> h((size_t)c->quad_n * 32);
      (void)hipMemcpy(h.data(), A.stamps, h.size() * 8, hipMemcpyDeviceToHost);
      for (int w = 0; w < 4; ++w) {
        double sum[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < nruns; ++i) for (int q = 0; q < 7; ++q) sum[q] += (double)h[((size_t)i * 4 + w) * 8 + q];
        fprintf(stderr, "[quad stamps wave %d, per chunk] setup %.0f  batch-top(sync+table) %.0f  state %.0f  barrier %.0f  pairs %.0f  barrier %.0f  diag+writeout %.0f\n",
                w, sum[0] / c->quad_n, sum[1] / c->quad_n, sum[2] / c->quad_n, sum[3] / c->quad_n, sum[4] / c->quad_n, sum[5] / c->quad_n, sum[6] / c->quad_n);
      }
    }
  }
#endif
  return FEAHIP_OK;
}
