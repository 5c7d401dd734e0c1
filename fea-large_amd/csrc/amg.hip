// amg.hip -- device side of the aggregation multigrid preconditioner (amg.h).
#include "amg.h"
#include "dpp_device.h"
#include <cmath>
#include <cstdlib>
#include <cstring>

// from kernels_solve.hip
void enq_spmv_arrays(hipStream_t stream, int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx,
                     const double *K, const double *xv, double *yv);
void enq_spmv_jacobi(hipStream_t stream, int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx, const double *K,
                     const float *K32, const unsigned short *K16, const double *xin, double *xout, const double *r, const double *minv,
                     double omega, double *part);
void enq_spmv_arrays_f32(hipStream_t stream, int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx,
                         const float *K, const double *xv, double *yv);
void enq_spmv_arrays_bf16(hipStream_t stream, int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx,
                          const unsigned short *K, const double *xv, double *yv);

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------

// Coarse block (I kind s, J kind r) = sum over the fine blocks (i, j) of the
// aggregate pair, in list order (deterministic), of  P_is' K_ij P_jr, where a
// translation row i has P_i0 = I, P_i1 = R(d_i) (u = t + w x d, R(d) w = w x d)
// and a rotation row has P_i0 = 0, P_i1 = I.  R(d)' B crosses every column of
// B with d from the left, B R(d) every row.  On level 0 the prescribed dofs are
// left out of the coarse space: their rows and columns of K_ij are skipped.
__device__ __forceinline__ void cross3(const double *d, double a0, double a1, double a2, double &o0, double &o1, double &o2)
{
  o0 = d[1] * a2 - d[2] * a1; o1 = d[2] * a0 - d[0] * a2; o2 = d[0] * a1 - d[1] * a0;
}
#define FEA_GAL_LANES 16                // lanes per aggregate pair
template <class TIN, class TOUT>
__global__ __launch_bounds__(256)
void k_galerkin(int npair, const int *prow, const int *crowptr, const int *cbptr, const int *cblist,
                const TIN *Kf, TOUT *Kc, const int *cbrow, const int *colidx_f, const uint8_t *type_f,
                const double *doff, const uint8_t *mask)
{
  // sixteen lanes per aggregate pair (I, J): lane s takes the fine blocks s, s + 16, ... of the pair's list, the four
  // coarse blocks (kind s of I, kind r of J) meet in a fixed butterfly (deterministic).  One thread per pair read the
  // 72-byte fine blocks one after the other: 9.7 ms for the 26M blocks of the 10M-tet block's level 0, 5.4 ms for the
  // 60 000 pairs of the levels below (too few threads to fill the chip)
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int kp = gid / FEA_GAL_LANES, sub = gid % FEA_GAL_LANES;
  const bool on = kp < npair;
  double acc[4][9];
#pragma unroll
  for (int w = 0; w < 4; ++w)
#pragma unroll
    for (int q = 0; q < 9; ++q) acc[w][q] = 0.0;
  const int pb = on ? cbptr[kp] : 0, pe = on ? cbptr[kp + 1] : 0;
  for (int p = pb + sub; p < pe; p += FEA_GAL_LANES) {
    const int q = cblist[p];
    const int i = cbrow[q], j = colidx_f[q];
    const int ti = type_f ? type_f[i] : 0, tj = type_f ? type_f[j] : 0;
    double m[9];
    const TIN *bq = Kf + (size_t)q * 9;
#pragma unroll
    for (int e = 0; e < 9; ++e) m[e] = (double)bq[e];
    if (mask) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb)
          if (mask[3 * i + a] || mask[3 * j + bb]) m[3 * a + bb] = 0.0;
    }
    double ml[9];                                // left factor of the rotation kind: R(d_i)' m, or m for a rotation row
#pragma unroll
    for (int e = 0; e < 9; ++e) ml[e] = m[e];
    if (ti == 0) {
      const double *d = doff + (size_t)i * 3;
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) cross3(d, m[cc], m[3 + cc], m[6 + cc], ml[cc], ml[3 + cc], ml[6 + cc]);
    }
    const double *dj = doff + (size_t)j * 3;
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
      if (sg == 0 && ti == 1) continue;          // a rotation row has no translation part
      const double *x = sg == 0 ? m : ml;
      if (tj == 0) {
#pragma unroll
        for (int e = 0; e < 9; ++e) acc[sg * 2][e] += x[e];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          double c0, c1, c2;
          cross3(dj, x[3 * r], x[3 * r + 1], x[3 * r + 2], c0, c1, c2);
          acc[sg * 2 + 1][3 * r] += c0; acc[sg * 2 + 1][3 * r + 1] += c1; acc[sg * 2 + 1][3 * r + 2] += c2;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 9; ++e) acc[sg * 2 + 1][e] += x[e];
      }
    }
  }
#pragma unroll
  for (int w = 0; w < 4; ++w)
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      acc[w][e] = dpp_row16_sum(acc[w][e]);
    }
  if (!on || sub >= 4) return;
  const int I = prow[kp];
  const int base0 = crowptr[2 * I], base1 = crowptr[2 * I + 1];
  const int t = kp - base0 / 4;                 // position of J in I's aggregate row (the site graph has a quarter of the blocks)
  const int k = sub == 0 ? base0 + 2 * t : (sub == 1 ? base0 + 2 * t + 1 : (sub == 2 ? base1 + 2 * t : base1 + 2 * t + 1));
  TOUT *o = Kc + (size_t)k * 9;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    const double v = sub == 0 ? acc[0][e] : (sub == 1 ? acc[1][e] : (sub == 2 ? acc[2][e] : acc[3][e]));
    o[e] = (TOUT)v;
  }
}

// inverse of the diagonal 3x3 blocks; a singular block (aggregate made of
// prescribed dofs only) gets the identity
template <class TK>
__global__ void k_block_inverse(int a0, int N, const int *diag, const TK *K, double *minv)
{
  const int a = a0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= N) return;
  double d[9];
  for (int t = 0; t < 9; ++t) d[t] = (double)K[(size_t)diag[a] * 9 + t];
  // rows/columns that are entirely zero (masked dofs on the coarse levels) become identity rows
  for (int i = 0; i < 3; ++i)
    if (d[3 * i + i] == 0.0) d[3 * i + i] = 1.0;
  const double det = d[0] * (d[4] * d[8] - d[5] * d[7]) - d[1] * (d[3] * d[8] - d[5] * d[6]) + d[2] * (d[3] * d[7] - d[4] * d[6]);
  double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (det != 0.0 && det == det) {
    const double id = 1.0 / det;
    m[0] = (d[4] * d[8] - d[5] * d[7]) * id; m[1] = (d[2] * d[7] - d[1] * d[8]) * id; m[2] = (d[1] * d[5] - d[2] * d[4]) * id;
    m[3] = (d[5] * d[6] - d[3] * d[8]) * id; m[4] = (d[0] * d[8] - d[2] * d[6]) * id; m[5] = (d[2] * d[3] - d[0] * d[5]) * id;
    m[6] = (d[3] * d[7] - d[4] * d[6]) * id; m[7] = (d[1] * d[6] - d[0] * d[7]) * id; m[8] = (d[0] * d[4] - d[1] * d[3]) * id;
  }
  for (int t = 0; t < 9; ++t) minv[(size_t)a * 9 + t] = m[t];
}

// The two smoothing updates, lane <-> scalar dof: a wave takes 21 nodes per step (63 lanes), a 256-thread block 252 nodes
// in three steps.  Every vector is read as one contiguous run per instruction, and so is D^-1 (row i of node a's block
// is the three doubles at 3 (3a + i)); the three components of a node's residual meet by shuffles.  (A lane per node
// read its nine inverse values with a 72-byte stride between lanes: 51 + 60 us of a 2.0 ms cycle on level 0.)
#define SM_NODES 252
template <bool NEXT>
__device__ __forceinline__ void smooth_body(int a0, int N, double omega, const double *minv, const double *r, const double *y, double *x)
{
  const int lane = threadIdx.x & 63, i = lane % 3;
  const long long base = (long long)a0 + (long long)blockIdx.x * SM_NODES + (threadIdx.x >> 6) * 21;
#pragma unroll
  for (int st = 0; st < 3; ++st) {
    const long long nb = base + st * 84;
    const bool on = lane < 63 && nb + lane / 3 < N;
    const size_t k = (size_t)nb * 3 + lane;
    double t = 0, m0 = 0, m1 = 0, m2 = 0, xo = 0;
    if (on) {
      const double *m = minv + 3 * k;
      m0 = m[0]; m1 = m[1]; m2 = m[2];
      t = NEXT ? r[k] - y[k] : r[k];
      if (NEXT) xo = x[k];
    }
    const double t0 = __shfl(t, lane - i), t1 = __shfl(t, lane - i + 1), t2 = __shfl(t, lane - i + 2);
    if (on) x[k] = xo + omega * (m0 * t0 + m1 * t1 + m2 * t2);
  }
}
// x = omega D^-1 r                       (first smoothing sweep from x = 0)
__global__ __launch_bounds__(256)
void k_smooth_first(int a0, int N, double omega, const double *minv, const double *r, double *x)
{
  smooth_body<false>(a0, N, omega, minv, r, nullptr, x);
}
// x += omega D^-1 (r - y)                (y = K x)
__global__ __launch_bounds__(256)
void k_smooth_next(int a0, int N, double omega, const double *minv, const double *r, const double *y, double *x)
{
  smooth_body<true>(a0, N, omega, minv, r, y, x);
}

// r_c = P' (r - K x): translation row of aggregate A = sum of the residuals of
// its translation rows, rotation row = sum of their moments d x res plus the
// residuals of its rotation rows.  Sixteen lanes share an aggregate: member
// p goes to lane p mod 16, the six partial sums meet in a fixed butterfly
// (deterministic), so the member gathers of one aggregate are in flight together.
__global__ __launch_bounds__(256)
void k_restrict(int nagg, const int *aptr, const int *anodes, const uint8_t *type_f, const double *doff,
                const double *r, const double *y, const uint8_t *mask, double *rc)
{
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int A = gid >> 4, sub = gid & 15;
  double s[6] = {0, 0, 0, 0, 0, 0};
  if (A < nagg) {
    for (int p = aptr[A] + sub; p < aptr[A + 1]; p += 16) {
      const int i = anodes[p];
      const size_t k = (size_t)i * 3;
      double a0 = r[k] - y[k], a1 = r[k + 1] - y[k + 1], a2 = r[k + 2] - y[k + 2];
      if (mask) { if (mask[k]) a0 = 0; if (mask[k + 1]) a1 = 0; if (mask[k + 2]) a2 = 0; }
      if (type_f && type_f[i]) { s[3] += a0; s[4] += a1; s[5] += a2; }
      else {
        s[0] += a0; s[1] += a1; s[2] += a2;
        double m0, m1, m2;
        cross3(doff + k, a0, a1, a2, m0, m1, m2);
        s[3] += m0; s[4] += m1; s[5] += m2;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) s[q] = dpp_row16_sum(s[q]);
  if (A < nagg && sub < 6) rc[(size_t)A * 6 + sub] = s[sub];
}

// x += over * P x_c : translation row i gets t + w x d_i, rotation row gets w
__global__ void k_prolong(int N, const int *agg, const uint8_t *type_f, const double *doff, const double *xc,
                          const uint8_t *mask, double over, double *x)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N || agg[i] < 0) return;                     // rows of other ranks have no aggregate here
  const double *c6 = xc + (size_t)agg[i] * 6;
  const size_t k = (size_t)i * 3;
  double u0, u1, u2;
  if (type_f && type_f[i]) { u0 = c6[3]; u1 = c6[4]; u2 = c6[5]; }
  else {
    const double dd[3] = {doff[k], doff[k + 1], doff[k + 2]};
    // w x d
    u0 = c6[0] + (c6[4] * dd[2] - c6[5] * dd[1]);
    u1 = c6[1] + (c6[5] * dd[0] - c6[3] * dd[2]);
    u2 = c6[2] + (c6[3] * dd[1] - c6[4] * dd[0]);
  }
  if (!mask || !mask[k]) x[k] += over * u0;
  if (!mask || !mask[k + 1]) x[k + 1] += over * u1;
  if (!mask || !mask[k + 2]) x[k + 2] += over * u2;
}

// power iteration helpers for lambda_max(D^-1 K)
__global__ void k_apply_minv(int a0, int N, const double *minv, const double *y, double *v)
{
  const int a = a0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= N) return;
  const double *m = minv + (size_t)a * 9;
  const double y0 = y[(size_t)a * 3], y1 = y[(size_t)a * 3 + 1], y2 = y[(size_t)a * 3 + 2];
  for (int i = 0; i < 3; ++i) v[(size_t)a * 3 + i] = m[3 * i] * y0 + m[3 * i + 1] * y1 + m[3 * i + 2] * y2;
}
__global__ void k_fill_pattern(int t0, int t1, int n, double *v)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) v[t] = (t >= t0 && t < t1) ? 1.0 + 0.37 * (double)((t * 2654435761u) >> 24) / 256.0 : 0.0;   // fixed pseudo-random start
}
// copies of the fine matrix for the smoother of level 0 (the CG itself multiplies with the double one): float, or
// bfloat16 rounded to nearest even from the float
__global__ void k_to_f32(size_t n, const double *src, float *dst)
{
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) dst[t] = (float)src[t];
}
// (n rows of three values -> rows of four, 8 bytes: three values and a pad.  The pads of a block's first two rows carry
// the low and the high half of its column index: the smoother's product reads 24 bytes per block and no index array)
__global__ void k_to_bf16(size_t nrows3, const double *src, const int *col, unsigned short *dst)
{
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nrows3; t += (size_t)gridDim.x * blockDim.x) {
    unsigned short o[4] = {0, 0, 0, 0};
    for (int j = 0; j < 3; ++j) {
      const unsigned u = __float_as_uint((float)src[t * 3 + j]);
      o[j] = (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
    }
    const unsigned c = (unsigned)col[t / 3];
    const int rib = (int)(t % 3);
    o[3] = rib == 0 ? (unsigned short)(c & 0xFFFFu) : (rib == 1 ? (unsigned short)(c >> 16) : (unsigned short)0);
    reinterpret_cast<uint2 *>(dst)[t] = make_uint2((unsigned)o[0] | ((unsigned)o[1] << 16), (unsigned)o[2] | ((unsigned)o[3] << 16));
  }
}

// |v|^2 in two stages with a fixed grid and order (deterministic)
__global__ __launch_bounds__(256)
void k_norm2_partial(int n, const double *v, double *part)
{
  __shared__ double s[256];
  double a = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) a += v[i] * v[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}
__global__ __launch_bounds__(256)
void k_norm2_final(int nparts, const double *part, double *out)
{
  __shared__ double s[256];
  double a = 0;
  for (int i = threadIdx.x; i < nparts; i += 256) a += part[i];
  s[threadIdx.x] = a;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w]; __syncthreads(); }
  if (threadIdx.x == 0) out[0] = s[0];
}
// v /= sqrt(*nrm2) with the squared norm read from the device (the power iteration below never visits the host); a norm
// that is zero or not a number leaves v alone -- the host sees the same value at the end and takes the fallback
__global__ void k_scale_by_norm(int n, const double *nrm2, double *v)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const double q = *nrm2;
  if (t < n && q > 0.0) v[t] *= 1.0 / sqrt(q);
}

// ---------------------------------------------------------------------------
// The small end of the hierarchy in ONE launch.  On the 10M-tet block the levels are 1 782 133 / 140 714 / 5 760 / 270 /
// 10 block rows; a W-cycle visits level 3 eight times and level 4 sixteen times per application, and every step
// there (a product over 5 260 blocks, a restriction to 5 aggregates) is a kernel that takes its 4-5 us of launch and
// drain whatever it does: 120 of the ~250 launches of one application, 0.5 of its 2.5 ms.  k_amg_tail runs the
// whole subtree below a level of at most FEA_TAIL_ROWS rows in one 1024-thread workgroup: the same steps in the same
// order (smooth, product, restrict, recurse, prolong, product, smooth), __syncthreads between them, the vectors
// where the separate kernels keep them (L2-resident), four lanes per block row in the products.
// ---------------------------------------------------------------------------
#define FEA_TAIL_T 1024
#define FEA_TAIL_MAXL 4
#define FEA_TAIL_ROWS 640             // 23 doubles of LDS per block row of the subtree: 640 + 64 + ... rows fit 160 KB
#define FEA_TAIL_LDS_BLOCKS 1024       // a level with at most this many blocks keeps its (float) matrix and column indices in LDS too
struct TailLevel {
  int N, Nc, nagg, nnzb;
  const int *rowptr, *colidx;
  const float *K32; const double *K;
  const double *minv; double omega;
  const int *agg, *aptr, *anodes; const uint8_t *type; const double *doff;
  double *r, *x;                        // global: the level the launch is entered at reads r here and leaves x here
  int o_v;                              // LDS, in doubles: r at o_v, x at o_v + 3N, y at o_v + 6N
  int o_K;                              // LDS, in doubles: nnzb * 9 floats, then nnzb column indices; -1: the matrix stays in L2
  int o_aux;                            // LDS, in doubles: minv 9N, doff 3N, then ints rowptr N+1, agg N, type N, anodes N, aptr nagg+1
};
// the read-only per-row arrays of a level, staged in LDS by k_amg_tail (every step would otherwise pay an L2 round trip
// for them: ~1 us x 19 steps per visit of the entry level)
struct TailAux { const double *minv, *doff; const int *rowptr, *agg, *type, *anodes, *aptr; };
__device__ __forceinline__ TailAux t_aux(const TailLevel &L, double *smem)
{
  TailAux a;
  a.minv = smem + L.o_aux; a.doff = a.minv + 9 * L.N;
  a.rowptr = reinterpret_cast<const int *>(a.doff + 3 * L.N); a.agg = a.rowptr + L.N + 1; a.type = a.agg + L.N;
  a.anodes = a.type + L.N; a.aptr = a.anodes + L.N;
  return a;
}
#define T_AUX_DOUBLES(N, nagg) (12 * (N) + (4 * (N) + (nagg) + 2 + 1) / 2 + 1)
struct TailArgs {
  TailLevel lv[FEA_TAIL_MAXL];
  int gamma[FEA_TAIL_MAXL]; double over[FEA_TAIL_MAXL];
  int nl, sweeps;
  unsigned long long *stamps;            // diagnostic build only: [level][8] accumulated s_memtime ticks per phase kind, [32] launches
  const double *blob;                    // the levels' read-only arrays and small matrices in their LDS layout (k_tail_pack), or null
  const float *ell; const int *goff;     // entry level's matrix by (group of 16 rows, slot, piece, lane) and the groups' first slots, or null
  int ngroups, o_goff;                   // o_goff: LDS doubles offset of the ngroups + 1 ints
  const double *cop; int cop_n, o_cop;   // the coarsest level's procedure as a dense n x n operator (transposed), its LDS offset; or null / -1
  int blob_lo, blob_n;                   // LDS doubles [blob_lo, blob_lo + blob_n): one contiguous copy per launch
  int lds_doubles;                       // all of it
};
#ifdef FEAHIP_DEBUG
__device__ unsigned long long g_tail_prev;
#define TS(l, cat) do { if (A.stamps && threadIdx.x == 0) { const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
                          atomicAdd(A.stamps + (l) * 8 + (cat), _t - g_tail_prev); g_tail_prev = _t; } } while (0)
#else
#define TS(l, cat) do { } while (0)
#endif
#define T_R(L) (smem + (L).o_v)
#define T_X(L) (smem + (L).o_v + 3 * (L).N)
#define T_Y(L) (smem + (L).o_v + 6 * (L).N)

__device__ __forceinline__ void t_smooth_first(const TailLevel &L, double *smem)
{
  const double *r = T_R(L); double *x = T_X(L);
  const TailAux X = t_aux(L, smem);
  for (int a = threadIdx.x; a < L.N; a += FEA_TAIL_T) {
    const double *m = X.minv + a * 9;
    const double r0 = r[a * 3], r1 = r[a * 3 + 1], r2 = r[a * 3 + 2];
    for (int i = 0; i < 3; ++i) x[a * 3 + i] = L.omega * (m[3 * i] * r0 + m[3 * i + 1] * r1 + m[3 * i + 2] * r2);
  }
}
__device__ __forceinline__ void t_smooth_next(const TailLevel &L, double *smem)
{
  const double *r = T_R(L), *y = T_Y(L); double *x = T_X(L);
  const TailAux X = t_aux(L, smem);
  for (int a = threadIdx.x; a < L.N; a += FEA_TAIL_T) {
    const double *m = X.minv + a * 9;
    const double t0 = r[a * 3] - y[a * 3], t1 = r[a * 3 + 1] - y[a * 3 + 1], t2 = r[a * 3 + 2] - y[a * 3 + 2];
    for (int i = 0; i < 3; ++i) x[a * 3 + i] += L.omega * (m[3 * i] * t0 + m[3 * i + 1] * t1 + m[3 * i + 2] * t2);
  }
}
// y = K x: four neighbouring lanes share a block row, their partial sums meet in two shuffles (fixed order)
__device__ __forceinline__ void t_spmv(const TailLevel &L, double *smem)
{
  const int t = threadIdx.x, sub = t & 3;
  const double *x = T_X(L); double *y = T_Y(L);
  const float *sK = L.o_K >= 0 ? reinterpret_cast<const float *>(smem + L.o_K) : (const float *)nullptr;
  const int *sCol = sK ? reinterpret_cast<const int *>(sK + (size_t)L.nnzb * 9) : (const int *)nullptr;
  const TailAux X = t_aux(L, smem);
  for (int row0 = 0; row0 < L.N; row0 += FEA_TAIL_T / 4) {
    const int row = row0 + (t >> 2);
    double a0 = 0, a1 = 0, a2 = 0;
    if (row < L.N) {
      const int kb = X.rowptr[row], ke = X.rowptr[row + 1];
      if (sK) {
        for (int k = kb + sub; k < ke; k += 4) {
          const int col = sCol[k];
          const float *vp = sK + k * 9;
          const double x0 = x[col * 3], x1 = x[col * 3 + 1], x2 = x[col * 3 + 2];
          a0 += (double)vp[0] * x0 + (double)vp[1] * x1 + (double)vp[2] * x2;
          a1 += (double)vp[3] * x0 + (double)vp[4] * x1 + (double)vp[5] * x2;
          a2 += (double)vp[6] * x0 + (double)vp[7] * x1 + (double)vp[8] * x2;
        }
      } else if (L.K32) {
        // from L2: the loads of eight blocks (a row of up to 32) are issued before the first is used -- one round trip
        // per product instead of one per block
        for (int k0 = kb + sub; k0 < ke; k0 += 32) {
          float v[8][9]; int col[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = k0 + 4 * u, kc = k < ke ? k : kb;
            col[u] = L.colidx[kc];
            const float *vp = L.K32 + (size_t)kc * 9;
#pragma unroll
            for (int q = 0; q < 9; ++q) v[u][q] = vp[q];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (k0 + 4 * u < ke) {
              const double x0 = x[col[u] * 3], x1 = x[col[u] * 3 + 1], x2 = x[col[u] * 3 + 2];
              a0 += (double)v[u][0] * x0 + (double)v[u][1] * x1 + (double)v[u][2] * x2;
              a1 += (double)v[u][3] * x0 + (double)v[u][4] * x1 + (double)v[u][5] * x2;
              a2 += (double)v[u][6] * x0 + (double)v[u][7] * x1 + (double)v[u][8] * x2;
            }
          }
        }
      } else {
        for (int k = kb + sub; k < ke; k += 4) {
          const int col = L.colidx[k];
          const double *vp = L.K + (size_t)k * 9;
          const double x0 = x[col * 3], x1 = x[col * 3 + 1], x2 = x[col * 3 + 2];
          a0 += vp[0] * x0 + vp[1] * x1 + vp[2] * x2;
          a1 += vp[3] * x0 + vp[4] * x1 + vp[5] * x2;
          a2 += vp[6] * x0 + vp[7] * x1 + vp[8] * x2;
        }
      }
    }
    a0 = dpp_quad_sum(a0); a1 = dpp_quad_sum(a1); a2 = dpp_quad_sum(a2);
    if (row < L.N && sub < 3) y[row * 3 + sub] = sub == 0 ? a0 : (sub == 1 ? a1 : a2);
  }
}
// r_c = P' (r - y), x += over P x_c: k_restrict / k_prolong for a workgroup
__device__ __forceinline__ void t_restrict(const TailLevel &L, const TailLevel &C, double *smem)
{
  const double *r = T_R(L), *y = T_Y(L); double *rc = T_R(C);
  const TailAux X = t_aux(L, smem);
  for (int base = 0; base < L.nagg * 16; base += FEA_TAIL_T) {
    const int gid = base + (int)threadIdx.x;
    const int A = gid >> 4, sub = gid & 15;
    double s[6] = {0, 0, 0, 0, 0, 0};
    if (A < L.nagg) {
      for (int p = X.aptr[A] + sub; p < X.aptr[A + 1]; p += 16) {
        const int i = X.anodes[p];
        const int k = i * 3;
        const double a0 = r[k] - y[k], a1 = r[k + 1] - y[k + 1], a2 = r[k + 2] - y[k + 2];
        if (X.type[i]) { s[3] += a0; s[4] += a1; s[5] += a2; }
        else {
          s[0] += a0; s[1] += a1; s[2] += a2;
          double m0, m1, m2;
          cross3(X.doff + k, a0, a1, a2, m0, m1, m2);
          s[3] += m0; s[4] += m1; s[5] += m2;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) s[q] = dpp_row16_sum(s[q]);
    if (A < L.nagg && sub < 6) rc[A * 6 + sub] = s[sub];
  }
}
__device__ __forceinline__ void t_prolong(const TailLevel &L, const TailLevel &C, double over, double *smem)
{
  const double *xc = T_X(C); double *x = T_X(L);
  const TailAux X = t_aux(L, smem);
  for (int i = threadIdx.x; i < L.N; i += FEA_TAIL_T) {
    const double *c6 = xc + X.agg[i] * 6;
    const int k = i * 3;
    double u0, u1, u2;
    if (X.type[i]) { u0 = c6[3]; u1 = c6[4]; u2 = c6[5]; }
    else {
      const double dd[3] = {X.doff[k], X.doff[k + 1], X.doff[k + 2]};
      u0 = c6[0] + (c6[4] * dd[2] - c6[5] * dd[1]);
      u1 = c6[1] + (c6[5] * dd[0] - c6[3] * dd[2]);
      u2 = c6[2] + (c6[3] * dd[1] - c6[4] * dd[0]);
    }
    x[k] += over * u0; x[k + 1] += over * u1; x[k + 2] += over * u2;
  }
}
// ---- the entry level's product out of a lane-major copy of its matrix ----------------------------------------------
// Lane (row, sub) of t_spmv multiplies blocks kb + sub, kb + sub + 4, ... of its row.  From the CSR array that is a
// 36-byte gather per block: ~1 200 load instructions of one CU's texture unit per product, 6.2 us, 19 of a launch's 33.
// k_tail_relayout writes the same blocks in the order the lanes read them: groups of 16 rows (one wave), per group as
// many slots as its longest row needs, a slot = three 16-byte pieces (values 0-3, 4-7, value 8 + column) of 64 lanes
// each; absent blocks are zeros on column 0.  Same blocks, same order of summation, 1 KB per load instruction.
__global__ void k_tail_relayout(int N, int ngroups, const int *goff, const int *rowptr, const int *colidx, const float *K32, float *ell)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int nslots = goff[ngroups];
  if (t >= nslots * 64) return;
  const int s = t >> 6, lane = t & 63;
  int g = 0;
  while (g + 1 < ngroups && goff[g + 1] <= s) ++g;
  const int u = s - goff[g], row = 16 * g + (lane >> 2), sub = lane & 3;
  float v[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (row < N) {
    const int k = rowptr[row] + sub + 4 * u;
    if (k < rowptr[row + 1]) {
      for (int q = 0; q < 9; ++q) v[q] = K32[(size_t)k * 9 + q];
      v[9] = __int_as_float(colidx[k]);
    }
  }
  float4 *o = reinterpret_cast<float4 *>(ell);
  for (int j = 0; j < 3; ++j) o[((size_t)s * 3 + j) * 64 + lane] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}
__device__ __forceinline__ void t_spmv_ell(const TailArgs &A, const TailLevel &L, double *smem)
{
  const int t = threadIdx.x, lane = t & 63, sub = lane & 3;
  const double *x = T_X(L); double *y = T_Y(L);
  const int *goff = reinterpret_cast<const int *>(smem + A.o_goff);
  const float4 *ell = reinterpret_cast<const float4 *>(A.ell);
  for (int g = t >> 6; g < A.ngroups; g += FEA_TAIL_T / 64) {
    const int row = 16 * g + (lane >> 2);
    const int s0 = goff[g], ns = goff[g + 1] - s0;
    double a0 = 0, a1 = 0, a2 = 0;
    for (int u0 = 0; u0 < ns; u0 += 4) {                 // four slots (twelve loads) in flight
      float4 p[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int su = s0 + (u0 + u < ns ? u0 + u : u0);
#pragma unroll
        for (int j = 0; j < 3; ++j) p[u][j] = ell[((size_t)su * 3 + j) * 64 + lane];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u0 + u < ns) {                               // (uniform)
          const int col = __float_as_int(p[u][2].y);
          const double x0 = x[col * 3], x1 = x[col * 3 + 1], x2 = x[col * 3 + 2];
          a0 += (double)p[u][0].x * x0 + (double)p[u][0].y * x1 + (double)p[u][0].z * x2;
          a1 += (double)p[u][0].w * x0 + (double)p[u][1].x * x1 + (double)p[u][1].y * x2;
          a2 += (double)p[u][1].z * x0 + (double)p[u][1].w * x1 + (double)p[u][2].x * x2;
        }
      }
    }
    a0 = dpp_quad_sum(a0); a1 = dpp_quad_sum(a1); a2 = dpp_quad_sum(a2);
    if (row < L.N && sub < 3) y[row * 3 + sub] = sub == 0 ? a0 : (sub == 1 ? a1 : a2);
  }
}
template <int D>
__device__ void t_cycle(const TailArgs &A, int l, double *smem)
{
  const TailLevel &L = A.lv[l];
  if ((L.Nc == 0 || D == 0) && L.N <= 16) {
    // a coarsest level of at most 16 block rows (10 on the 10M-tet block, visited 16 times per cycle) is the work of
    // one wave: its 1 + 2 * sweeps steps follow each other in program order (a wave's LDS operations complete in
    // order), the other fifteen waves wait at ONE barrier instead of taking part in five
    if (A.cop && threadIdx.x < 64) {
      // ... and the whole procedure is ONE linear map of 3N <= 48 numbers (k_tail_coarse_op): a dense product out of LDS
      const int n = A.cop_n, i = threadIdx.x;
      const double *B = smem + A.o_cop, *r = T_R(L);
      if (i < n) {
        double a = 0;
        for (int j = 0; j < n; ++j) a += B[j * n + i] * r[j];
        T_X(L)[i] = a;
      }
    } else if (threadIdx.x < 64) {
      t_smooth_first(L, smem);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      for (int s = 0; s < A.sweeps; ++s) {
        t_spmv(L, smem); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        t_smooth_next(L, smem); __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      }
    }
    __syncthreads(); TS(l, 1);
    return;
  }
  t_smooth_first(L, smem);
  __syncthreads(); TS(l, 1);
  if (L.Nc == 0 || D == 0) {                            // the coarsest level: a few damped Jacobi sweeps
    for (int s = 0; s < A.sweeps; ++s) { t_spmv(L, smem); __syncthreads(); TS(l, 2); t_smooth_next(L, smem); __syncthreads(); TS(l, 1); }
    return;
  }
  const bool ell = l == 0 && A.ell != nullptr;          // (uniform)
  for (int g = 0; g < A.gamma[l]; ++g) {
    if (ell) t_spmv_ell(A, L, smem); else t_spmv(L, smem);
    __syncthreads(); TS(l, 2);
    t_restrict(L, A.lv[l + 1], smem); __syncthreads(); TS(l, 3);
    if (D > 0) t_cycle<(D > 0 ? D - 1 : 0)>(A, l + 1, smem);
    t_prolong(L, A.lv[l + 1], A.over[l], smem); __syncthreads(); TS(l, 4);
  }
  if (ell) t_spmv_ell(A, L, smem); else t_spmv(L, smem);
  __syncthreads(); TS(l, 2);
  t_smooth_next(L, smem); __syncthreads(); TS(l, 1);
}
// the levels' read-only arrays and the small matrices, laid out as the kernel's LDS holds them, written to `base`
// (k_amg_tail without a blob: its LDS; k_tail_pack: the global blob the launches then copy in one go)
__device__ __forceinline__ void t_stage(const TailArgs &A, double *base)
{
  for (int l = 0; l < A.nl; ++l) {
    const TailLevel &L = A.lv[l];
    double *minv = base + L.o_aux, *doff = minv + 9 * L.N;
    int *rowptr = reinterpret_cast<int *>(doff + 3 * L.N), *agg = rowptr + L.N + 1, *type = agg + L.N, *anodes = type + L.N,
        *aptr = anodes + L.N;
    for (int i = threadIdx.x; i < 9 * L.N; i += FEA_TAIL_T) minv[i] = L.minv[i];
    for (int i = threadIdx.x; i <= L.N; i += FEA_TAIL_T) rowptr[i] = L.rowptr[i];
    if (L.Nc) {
      for (int i = threadIdx.x; i < 3 * L.N; i += FEA_TAIL_T) doff[i] = L.doff[i];
      for (int i = threadIdx.x; i < L.N; i += FEA_TAIL_T) { agg[i] = L.agg[i]; type[i] = (int)L.type[i]; anodes[i] = L.anodes[i]; }
      for (int i = threadIdx.x; i <= L.nagg; i += FEA_TAIL_T) aptr[i] = L.aptr[i];
    }
  }
  if (A.cop) {
    double *sc = base + A.o_cop;
    for (int i = threadIdx.x; i < A.cop_n * A.cop_n; i += FEA_TAIL_T) sc[i] = A.cop[i];
  }
  if (A.goff) {
    int *sg = reinterpret_cast<int *>(base + A.o_goff);
    for (int i = threadIdx.x; i <= A.ngroups; i += FEA_TAIL_T) sg[i] = A.goff[i];
  }
  for (int l = 0; l < A.nl; ++l) {                      // small matrices too
    const TailLevel &L = A.lv[l];
    if (L.o_K < 0) continue;
    float *sK = reinterpret_cast<float *>(base + L.o_K);
    int *sCol = reinterpret_cast<int *>(sK + (size_t)L.nnzb * 9);
    for (int i = threadIdx.x; i < L.nnzb * 9; i += FEA_TAIL_T) sK[i] = L.K32[i];
    for (int i = threadIdx.x; i < L.nnzb; i += FEA_TAIL_T) sCol[i] = L.colidx[i];
  }
}
// The coarsest level's procedure -- x = omega D^-1 r, then `sweeps` times x += omega D^-1 (r - K x) -- is a fixed linear
// map x = B r of 3N <= 48 numbers: B_0 = omega D^-1, B_{s+1} = B_s + omega D^-1 (I - K B_s).  Formed densely once per
// numeric setup (one workgroup, LDS), stored transposed; a visit of that level is then one dense product instead of
// 1 + 2 sweeps steps that each wait for the one before (2.8 us per visit, sixteen visits per cycle).
#define T_COP_MAX 48
__global__ __launch_bounds__(256)
void k_tail_coarse_op(TailLevel L, int sweeps, double *cop)
{
  __shared__ double sA[T_COP_MAX * T_COP_MAX], sB[T_COP_MAX * T_COP_MAX], sT[T_COP_MAX * T_COP_MAX];
  const int n = 3 * L.N, t = threadIdx.x;
  for (int i = t; i < n * n; i += 256) { sA[i] = 0.0; sB[i] = 0.0; }
  __syncthreads();
  for (int k = t; k < L.nnzb; k += 256) {                // dense K (a thread per block; blocks of a row are distinct columns)
    int row = 0;
    while (L.rowptr[row + 1] <= k) ++row;
    const int col = L.colidx[k];
    for (int q = 0; q < 9; ++q)
      sA[(3 * row + q / 3) * n + 3 * col + q % 3] = L.K32 ? (double)L.K32[(size_t)k * 9 + q] : L.K[(size_t)k * 9 + q];
  }
  for (int i = t; i < L.N * 9; i += 256) {               // B_0 = omega D^-1 (block diagonal)
    const int a = i / 9, q = i % 9;
    sB[(3 * a + q / 3) * n + 3 * a + q % 3] = L.omega * L.minv[i];
  }
  __syncthreads();
  for (int s = 0; s < sweeps; ++s) {
    for (int e = t; e < n * n; e += 256) {               // T = I - K B
      const int i = e / n, j = e % n;
      double a = i == j ? 1.0 : 0.0;
      for (int m = 0; m < n; ++m) a -= sA[i * n + m] * sB[m * n + j];
      sT[e] = a;
    }
    __syncthreads();
    for (int e = t; e < n * n; e += 256) {               // B += omega D^-1 T (the 3x3 block row of node i / 3)
      const int i = e / n, j = e % n, a = i / 3;
      const double *m = L.minv + (size_t)a * 9 + 3 * (i % 3);
      sB[e] += L.omega * (m[0] * sT[(3 * a) * n + j] + m[1] * sT[(3 * a + 1) * n + j] + m[2] * sT[(3 * a + 2) * n + j]);
    }
    __syncthreads();
  }
  for (int e = t; e < n * n; e += 256) cop[(e % n) * n + e / n] = sB[e];    // transposed: lane i of the product reads consecutive words
}
__global__ __launch_bounds__(FEA_TAIL_T)
void k_tail_pack(TailArgs A, double *blob)
{
  t_stage(A, blob - A.blob_lo);
}
__global__ __launch_bounds__(FEA_TAIL_T)
void k_amg_tail(TailArgs A)
{
  extern __shared__ __attribute__((aligned(16))) double tail_smem[];
  double *smem = tail_smem;
#ifdef FEAHIP_DEBUG
  if (A.stamps && threadIdx.x == 0) { g_tail_prev = __builtin_amdgcn_s_memtime(); atomicAdd(A.stamps + 32, 1ull); }
#endif
  if (A.blob) {
    // one contiguous copy, every load in flight before the first wait (the per-array loops paid one L2 round trip each:
    // 3.9 us of a 33.6 us launch, 632 launches per solve of data that changes once per Newton iteration)
    typedef double tl_v2d __attribute__((ext_vector_type(2)));
    const tl_v2d *src = reinterpret_cast<const tl_v2d *>(A.blob);
    tl_v2d *dst = reinterpret_cast<tl_v2d *>(smem + A.blob_lo);
    for (int i = threadIdx.x; i < A.blob_n / 2; i += FEA_TAIL_T) dst[i] = src[i];
  } else t_stage(A, smem);
  {
    const TailLevel &L = A.lv[0];
    double *r = T_R(L);
    for (int i = threadIdx.x; i < 3 * L.N; i += FEA_TAIL_T) r[i] = L.r[i];
  }
  __syncthreads(); TS(0, 0);
  t_cycle<FEA_TAIL_MAXL - 1>(A, 0, smem);
  {
    const TailLevel &L = A.lv[0];
    const double *x = T_X(L);
    for (int i = threadIdx.x; i < 3 * L.N; i += FEA_TAIL_T) L.x[i] = x[i];
  }
  TS(0, 5);
}

// ---------------------------------------------------------------------------
// hierarchy
// ---------------------------------------------------------------------------
static AmgHierarchy *H(feahip_ctx *c) { return reinterpret_cast<AmgHierarchy *>(c->amg); }
static TailArgs tail_args(feahip_ctx *c);

template <class T>
static int up(feahip_ctx *c, T **dst, const std::vector<T> &v, long long &bytes)
{
  FEA_HIP_CHECK(c, hipMalloc((void **)dst, sizeof(T) * (v.size() ? v.size() : 1)));
  if (!v.empty()) FEA_HIP_CHECK(c, hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
  bytes += (long long)(sizeof(T) * v.size());
  return FEAHIP_OK;
}
static int zeros(feahip_ctx *c, double **dst, size_t n, long long &bytes)
{
  FEA_HIP_CHECK(c, hipMalloc((void **)dst, sizeof(double) * (n ? n : 1)));
  FEA_HIP_CHECK(c, hipMemset(*dst, 0, sizeof(double) * (n ? n : 1)));
  bytes += (long long)(sizeof(double) * n);
  return FEAHIP_OK;
}

int amg_create(feahip_ctx *c)
{
  if (c->amg) return FEAHIP_OK;
  std::vector<HostAmgLevel> hl;
  std::vector<double> pos((size_t)c->N * 3);
  {                                   // aggregate geometry from the reference configuration
    std::vector<double> pad((size_t)c->N * 4);
    FEA_HIP_CHECK(c, hipMemcpy(pad.data(), c->d_X0, sizeof(double) * pad.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < c->N; ++i) for (int d = 0; d < 3; ++d) pos[(size_t)i * 3 + d] = pad[(size_t)i * 4 + d];
  }
  if (!build_host_amg(c->h_rowptr, c->h_colidx, pos, c->row0, c->row1, hl)) {
    c->err = "multigrid hierarchy unavailable for this mesh (too small, or a coarse row exceeds the SpMV chunk)";
    return FEAHIP_EINVAL;
  }
  AmgHierarchy *h = new AmgHierarchy();
  c->amg = h;
  h->row0 = c->row0; h->row1 = c->row1;
  { const char *e = getenv("FEAHIP_AMG_GAMMA"); if (e) h->gamma = atoi(e); }
  { const char *e = getenv("FEAHIP_AMG_F32"); h->coarse_f32 = !(e && atoi(e) == 0); }
  { const char *e = getenv("FEAHIP_AMG_FINE_BITS"); if (e && (atoi(e) == 32 || atoi(e) == 64)) h->fine_bits = atoi(e); }
  { const char *e = getenv("FEAHIP_AMG_OVER"); if (e) h->over = atof(e); }
  { const char *e = getenv("FEAHIP_AMG_GAMMA_FROM"); if (e) h->gamma_from = atoi(e); }
  { const char *e = getenv("FEAHIP_AMG_GAMMA_UNTIL"); if (e) h->gamma_until = atoi(e); }
  { const char *e = getenv("FEAHIP_AMG_TAIL_BLOB"); h->tail_blob = !(e && atoi(e) == 0); }
  { const char *e = getenv("FEAHIP_AMG_TAIL_ELL"); h->tail_ell = !(e && atoi(e) == 0); }
  { const char *e = getenv("FEAHIP_AMG_TAIL_COP"); h->tail_cop = !(e && atoi(e) == 0); }
  { const char *e = getenv("FEAHIP_AMG_FUSED_POST"); h->fused_post = e && atoi(e) != 0; }
  { const char *e = getenv("FEAHIP_AMG_SWEEPS"); if (e) h->coarse_sweeps = atoi(e); }
  // the levels the one-workgroup kernel takes: from the first level below the finest of at most FEA_TAIL_ROWS rows
  h->tail_from = -1;
  { const char *e = getenv("FEAHIP_AMG_TAIL");
    if (!(e && atoi(e) == 0))
      for (size_t l = 1; l < hl.size(); ++l)
        if (hl[l].N <= FEA_TAIL_ROWS && (int)(hl.size() - l) <= FEA_TAIL_MAXL) {
          long long need = 0;
          for (size_t k = l; k < hl.size(); ++k) need += 9LL * hl[k].N + T_AUX_DOUBLES(hl[k].N, hl[k].Sc);
          if (need * 8 <= 150 * 1024) h->tail_from = (int)l;
          break;
        } }
  int rc;
  h->lv.resize(hl.size());
  for (size_t l = 0; l < hl.size(); ++l) {
    AmgLevel &L = h->lv[l];
    const HostAmgLevel &S = hl[l];
    L.N = S.N; L.nnzb = (int)S.colidx.size(); L.nchunks = (int)S.chunk.size() - 1; L.Nc = 2 * S.Sc;
    if (l == 0) {
      L.rowptr = c->d_rowptr; L.colidx = c->d_colidx; L.diag = c->d_diag; L.chunk = c->d_chunk; L.K = c->d_K;
      L.nchunks = c->nchunks;
      if (h->fine_bits == 32) {
        FEA_HIP_CHECK(c, hipMalloc((void **)&L.K32, sizeof(float) * ((size_t)c->nnzb * 9 + 4)));
        FEA_HIP_CHECK(c, hipMemset(L.K32, 0, sizeof(float) * ((size_t)c->nnzb * 9 + 4)));
        h->bytes += (long long)(sizeof(float) * (size_t)c->nnzb * 9);
      } else if (h->fine_bits == 16) {
        FEA_HIP_CHECK(c, hipMalloc((void **)&L.K16, sizeof(unsigned short) * ((size_t)c->nnzb * 12 + 8)));
        FEA_HIP_CHECK(c, hipMemset(L.K16, 0, sizeof(unsigned short) * ((size_t)c->nnzb * 12 + 8)));
        h->bytes += (long long)(sizeof(unsigned short) * (size_t)c->nnzb * 12);
      }
    } else {
      L.owns_matrix = true;
      if ((rc = up(c, &L.rowptr, S.rowptr, h->bytes))) return rc;
      if ((rc = up(c, &L.colidx, S.colidx, h->bytes))) return rc;
      if ((rc = up(c, &L.diag, S.diag, h->bytes))) return rc;
      if ((rc = up(c, &L.chunk, S.chunk, h->bytes))) return rc;
      if (h->coarse_f32) {
        FEA_HIP_CHECK(c, hipMalloc((void **)&L.K32, sizeof(float) * ((size_t)L.nnzb * 9 + 4)));
        FEA_HIP_CHECK(c, hipMemset(L.K32, 0, sizeof(float) * ((size_t)L.nnzb * 9 + 4)));
        h->bytes += (long long)(sizeof(float) * (size_t)L.nnzb * 9);
      } else if ((rc = zeros(c, &L.K, (size_t)L.nnzb * 9 + 2, h->bytes))) return rc;
      if ((rc = zeros(c, &L.r, (size_t)L.N * 3, h->bytes))) return rc;
      if ((rc = zeros(c, &L.x, (size_t)L.N * 3, h->bytes))) return rc;
      if ((rc = zeros(c, &L.y, (size_t)L.N * 3, h->bytes))) return rc;
      if ((rc = up(c, &L.type, S.type, h->bytes))) return rc;
    }
    if ((rc = zeros(c, &L.minv, (size_t)L.N * 9, h->bytes))) return rc;
    if (S.Sc > 0) {
      if ((rc = up(c, &L.agg, S.agg, h->bytes))) return rc;
      if ((rc = up(c, &L.doff, S.doff, h->bytes))) return rc;
      if ((rc = up(c, &L.aptr, S.aptr, h->bytes))) return rc;
      if ((rc = up(c, &L.anodes, S.anodes, h->bytes))) return rc;
      if ((rc = up(c, &L.cbptr, S.cbptr, h->bytes))) return rc;
      if ((rc = up(c, &L.cblist, S.cblist, h->bytes))) return rc;
      if ((rc = up(c, &L.prow, S.prow, h->bytes))) return rc;
      if ((rc = up(c, &L.cbrow, S.cbrow, h->bytes))) return rc;
    }
  }
  if ((rc = zeros(c, &h->d_z, (size_t)c->ndof, h->bytes))) return rc;
  if ((rc = zeros(c, &h->d_pw, (size_t)c->ndof, h->bytes))) return rc;
  if ((rc = zeros(c, &h->d_lam, (size_t)64, h->bytes))) return rc;
  if (h->tail_from >= 0 && h->tail_ell && h->coarse_f32) {
    const HostAmgLevel &S = hl[(size_t)h->tail_from];
    const int G = (S.N + 15) / 16;
    std::vector<int> goff((size_t)G + 1, 0);
    int worst = 0;
    for (int g = 0; g < G; ++g) {
      int ns = 0;
      for (int r = 16 * g; r < std::min(S.N, 16 * g + 16); ++r) ns = std::max(ns, (S.rowptr[(size_t)r + 1] - S.rowptr[(size_t)r] + 3) / 4);
      goff[(size_t)g + 1] = goff[(size_t)g] + ns; worst = std::max(worst, ns);
    }
    if (worst <= 32) {                                  // (rows of more than 128 blocks have a chunk of their own elsewhere; here they keep the CSR product)
      h->tail_groups = G; h->tail_slots = goff[(size_t)G];
      if ((rc = up(c, &h->d_tail_goff, goff, h->bytes))) return rc;
      FEA_HIP_CHECK(c, hipMalloc((void **)&h->d_tail_ell, sizeof(float) * 12 * 64 * (size_t)std::max(h->tail_slots, 1)));
      h->bytes += (long long)(sizeof(float) * 12 * 64 * (size_t)h->tail_slots);
    }
  }
  if (h->tail_from >= 0 && h->tail_cop && hl.back().N <= 16 && hl.back().Sc == 0 && (int)hl.size() - 1 > h->tail_from) {
    h->tail_cop_n = 3 * hl.back().N;
    if ((rc = zeros(c, &h->d_tail_cop, (size_t)h->tail_cop_n * h->tail_cop_n, h->bytes))) return rc;
  }
  if (h->tail_from >= 0 && h->tail_blob) {
    const TailArgs A = tail_args(c);
    if ((rc = zeros(c, &h->d_tail_blob, (size_t)A.blob_n, h->bytes))) return rc;
  }
  return FEAHIP_OK;
}

void amg_destroy(feahip_ctx *c)
{
  AmgHierarchy *h = H(c);
  if (!h) return;
  for (AmgLevel &L : h->lv) {
    void *own[] = {L.minv, L.agg, L.doff, L.aptr, L.anodes, L.cbptr, L.cblist, L.prow, L.cbrow, L.r, L.x, L.y, L.type};
    for (void *p : own) if (p) (void)hipFree(p);
    if (L.owns_matrix) { void *m[] = {L.rowptr, L.colidx, L.diag, L.chunk, L.K, L.K32}; for (void *p : m) if (p) (void)hipFree(p); }
    else { if (L.K32) (void)hipFree(L.K32); if (L.K16) (void)hipFree(L.K16); }
  }
  if (h->d_z) (void)hipFree(h->d_z);
  if (h->d_pw) (void)hipFree(h->d_pw);
  if (h->d_lam) (void)hipFree(h->d_lam);
  if (h->d_tail_blob) (void)hipFree(h->d_tail_blob);
  if (h->d_tail_ell) (void)hipFree(h->d_tail_ell);
  if (h->d_tail_cop) (void)hipFree(h->d_tail_cop);
  if (h->d_tail_goff) (void)hipFree(h->d_tail_goff);
  delete h;
  c->amg = nullptr;
}

#define G256(n) dim3(((n) + 255) / 256 > 0 ? ((n) + 255) / 256 : 1), dim3(256), 0, c->stream

// rows and SpMV chunks a level works on: everything below level 0 is the rank's
// own; on level 0 the rank's rows of the context's arrays
struct LevelRange { int a0, a1, ch0, nch; };
static LevelRange level_range(feahip_ctx *c, int l)
{
  AmgHierarchy *h = H(c);
  const AmgLevel &L = h->lv[l];
  if (l == 0) return {h->row0, h->row1, c->chunk0, c->nchunks_local};
  return {0, L.N, 0, L.nchunks};
}
#define GROWS(R) G256((R).a1 - (R).a0)
#define GSMOOTH(R) dim3(((R).a1 - (R).a0 + SM_NODES - 1) / SM_NODES > 0 ? ((R).a1 - (R).a0 + SM_NODES - 1) / SM_NODES : 1), dim3(256), 0, c->stream

static void level_spmv(feahip_ctx *c, const AmgLevel &L, const LevelRange &R, const double *x, double *y)
{
  if (L.K16) enq_spmv_arrays_bf16(c->stream, R.ch0, R.nch, L.chunk, L.rowptr, L.colidx, L.K16, x, y);
  else if (L.K32) enq_spmv_arrays_f32(c->stream, R.ch0, R.nch, L.chunk, L.rowptr, L.colidx, L.K32, x, y);
  else enq_spmv_arrays(c->stream, R.ch0, R.nch, L.chunk, L.rowptr, L.colidx, L.K, x, y);
}

// coarse matrices, block inverses and the Jacobi damping of every level, for the current K
static int amg_numeric(feahip_ctx *c)
{
  AmgHierarchy *h = H(c);
  const int nl = (int)h->lv.size();
  if (nl > 64) { c->err = "multigrid: more than 64 levels"; return FEAHIP_ESTATE; }
  for (int l = 0; l < nl; ++l) {
    AmgLevel &L = h->lv[l];
    const LevelRange R = level_range(c, l);
    if (l == 0 && (L.K32 || L.K16)) {                    // the rank's rows only
      const size_t q0 = (size_t)c->h_rowptr[(size_t)h->row0] * 9, q1 = (size_t)c->h_rowptr[(size_t)h->row1] * 9;
      if (L.K16) hipLaunchKernelGGL(k_to_bf16, dim3(4096), dim3(256), 0, c->stream, (q1 - q0) / 3, (const double *)L.K + q0, L.colidx + q0 / 9, L.K16 + q0 / 3 * 4);
      else hipLaunchKernelGGL(k_to_f32, dim3(4096), dim3(256), 0, c->stream, q1 - q0, (const double *)L.K + q0, L.K32 + q0);
    }
    if (l > 0 && L.K32) hipLaunchKernelGGL(k_block_inverse<float>, GROWS(R), R.a0, R.a1, L.diag, L.K32, L.minv);
    else hipLaunchKernelGGL(k_block_inverse<double>, GROWS(R), R.a0, R.a1, L.diag, L.K, L.minv);
    if (L.Nc > 0) {
      AmgLevel &C = h->lv[l + 1];
      const uint8_t *gm = l == 0 ? c->d_dofmask : (const uint8_t *)nullptr;
#define GALERKIN(TI, TO, KI, KO) hipLaunchKernelGGL((k_galerkin<TI, TO>), G256((size_t)(C.nnzb / 4) * FEA_GAL_LANES), C.nnzb / 4, L.prow, C.rowptr, L.cbptr, \
                                                    L.cblist, KI, KO, L.cbrow, L.colidx, L.type, L.doff, gm)
      if (l > 0 && L.K32 && C.K32) GALERKIN(float, float, L.K32, C.K32);
      else if (C.K32) GALERKIN(double, float, L.K, C.K32);
      else GALERKIN(double, double, L.K, C.K);
#undef GALERKIN
    }
    // lambda_max(D^-1 K) by a few power iterations -> omega = 4 / (3 lambda_max).  The iteration stays on the device:
    // the squared norm of every step lands in d_lam[l] and the next step scales by it there (k_scale_by_norm); the
    // host reads all levels' last norms once, below -- one synchronisation per numeric setup instead of one per step
    // and level (forty on the 10M-tet block).  Same arithmetic, same omega to the bit as the loop that went through
    // the host; measured: no difference in the Newton iteration either (0.1575 / 0.1586 against 0.1577 / 0.1559 s,
    // gpurun_out/r4_h5) -- the forty round trips cost less than the spread between two runs.
    double *v = (l == 0) ? h->d_pw : L.x, *y = (l == 0) ? c->d_q : L.y;
    const int n = L.N * 3;
    hipLaunchKernelGGL(k_fill_pattern, G256(n), 3 * R.a0, 3 * R.a1, n, v);
    for (int it = 0; it < 8; ++it) {
      level_spmv(c, L, R, v, y);
      hipLaunchKernelGGL(k_apply_minv, GROWS(R), R.a0, R.a1, L.minv, y, v);
      const int nb = n >= 256 * 1024 ? 1024 : (n + 255) / 256;
      hipLaunchKernelGGL(k_norm2_partial, dim3(nb), dim3(256), 0, c->stream, n, v, c->d_part);
      hipLaunchKernelGGL(k_norm2_final, dim3(1), dim3(256), 0, c->stream, nb, c->d_part, h->d_lam + l);
      if (it < 7) hipLaunchKernelGGL(k_scale_by_norm, G256(n), n, h->d_lam + l, v);
    }
  }
  {
    double nrm2[64];
    FEA_HIP_CHECK(c, hipMemcpyAsync(nrm2, h->d_lam, sizeof(double) * (size_t)nl, hipMemcpyDeviceToHost, c->stream));
    FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    for (int l = 0; l < nl; ++l) {
      const double lam = (nrm2[l] > 0 && nrm2[l] == nrm2[l]) ? sqrt(nrm2[l]) : 2.0;   // v was normalised before the last product
      h->lv[l].omega = 4.0 / (3.0 * 1.1 * lam);     // 10 % margin: the power iteration approaches lambda_max from below
    }
  }
  if (h->d_tail_ell) {                                   // the tail's entry-level matrix for this K, lane-major
    const AmgLevel &L = h->lv[(size_t)h->tail_from];
    if (L.K32) hipLaunchKernelGGL(k_tail_relayout, G256(h->tail_slots * 64), L.N, h->tail_groups, h->d_tail_goff, L.rowptr, L.colidx, L.K32, h->d_tail_ell);
  }
  if (h->d_tail_cop) {                                   // the coarsest level's procedure as one operator, for this K
    const TailArgs A = tail_args(c);
    hipLaunchKernelGGL(k_tail_coarse_op, dim3(1), dim3(256), 0, c->stream, A.lv[A.nl - 1], h->coarse_sweeps, h->d_tail_cop);
  }
  if (h->d_tail_blob) {                                  // the tail's read-only arrays for this K, in its LDS layout
    const TailArgs A = tail_args(c);
    hipLaunchKernelGGL(k_tail_pack, dim3(1), dim3(FEA_TAIL_T), 0, c->stream, A, h->d_tail_blob);
  }
  FEA_HIP_CHECK(c, hipGetLastError());
  h->numeric_valid = true;
  return FEAHIP_OK;
}

// x_l = cycle(r_l) from a zero initial guess: pre-smooth, `gamma` coarse
// corrections (gamma = 1: V-cycle, 2: W-cycle), post-smooth.  Every step is a
// fixed linear map and the cycle is symmetric in the K inner product
// (I - M K = S' (I - P B P' K)^gamma S), which is what CG needs.  The coarse
// correction of plain aggregation is too small by a mesh-independent factor;
// `over` scales it (over-correction).
// the subtree from level tail_from down, in one launch: reads its r, leaves its x (what amg_cycle does for that level)
// arguments and LDS layout of the tail: the vectors of all levels first, then -- contiguous, even offsets -- the
// read-only arrays of all levels and the small matrices: that second part is what k_tail_pack writes to the blob
static TailArgs tail_args(feahip_ctx *c)
{
  AmgHierarchy *h = H(c);
  TailArgs A;
  memset(&A, 0, sizeof(A));
  const int nl = (int)h->lv.size() - h->tail_from;
  int off = 0;                                          // LDS layout, in doubles
  for (int k = 0; k < nl; ++k) {
    const int l = h->tail_from + k;
    const AmgLevel &L = h->lv[l];
    TailLevel &T = A.lv[k];
    T.N = L.N; T.Nc = L.Nc; T.nagg = L.Nc / 2; T.nnzb = L.nnzb;
    T.rowptr = L.rowptr; T.colidx = L.colidx; T.K32 = L.K32; T.K = L.K; T.minv = L.minv; T.omega = L.omega;
    T.agg = L.agg; T.aptr = L.aptr; T.anodes = L.anodes; T.type = L.type; T.doff = L.doff;
    T.r = L.r; T.x = L.x;
    T.o_v = off; off += 9 * L.N;
    T.o_K = -1;
    A.gamma[k] = (l < h->gamma_from || l >= h->gamma_until) ? 1 : h->gamma;
    A.over[k] = A.gamma[k] >= 2 ? h->over : fmin(h->over, 1.0);
  }
  off += off & 1;
  A.blob_lo = off;
  for (int k = 0; k < nl; ++k) { const AmgLevel &L = h->lv[h->tail_from + k]; A.lv[k].o_aux = off; off += T_AUX_DOUBLES(L.N, L.Nc / 2); }
  A.o_goff = off; off += (h->tail_groups + 2) / 2 + 1;
  A.cop = h->d_tail_cop; A.cop_n = h->tail_cop_n; A.o_cop = -1;
  if (A.cop && (off + A.cop_n * A.cop_n) * 8 <= 156 * 1024) { A.o_cop = off; off += A.cop_n * A.cop_n; }
  else A.cop = nullptr;
  for (int k = nl - 1; k >= 0; --k) {                   // small matrices too, the most visited first, while they fit
    const AmgLevel &L = h->lv[h->tail_from + k];
    const int need = (L.nnzb * 10 + 1) / 2 + 1;
    if (L.K32 && L.nnzb <= FEA_TAIL_LDS_BLOCKS && (off + need) * 8 <= 150 * 1024) { A.lv[k].o_K = off; off += need; }
  }
  off += off & 1;
  A.blob_n = off - A.blob_lo;
  // the entry level's product reads the lane-major copy unless that level's matrix sits in LDS anyway
  const bool ell = h->d_tail_ell && A.lv[0].o_K < 0;
  A.ell = ell ? h->d_tail_ell : (const float *)nullptr; A.goff = ell ? h->d_tail_goff : (const int *)nullptr;
  A.ngroups = ell ? h->tail_groups : 0;
  A.lds_doubles = off;
  A.nl = nl; A.sweeps = h->coarse_sweeps;
  A.blob = h->d_tail_blob;
  return A;
}

static void launch_tail(feahip_ctx *c)
{
  TailArgs A = tail_args(c);
  const int off = A.lds_doubles;
  A.stamps = nullptr;
#ifdef FEAHIP_DEBUG
  static unsigned long long *d_stamps = nullptr;
  if (getenv("FEAHIP_TAIL_STAMPS")) {
    if (!d_stamps) { (void)hipMalloc((void **)&d_stamps, 8 * 40); (void)hipMemset(d_stamps, 0, 8 * 40); }
    A.stamps = d_stamps;
    static int calls = 0;
    if (++calls % 2000 == 0) {
      unsigned long long hs[40];
      (void)hipMemcpy(hs, d_stamps, sizeof(hs), hipMemcpyDeviceToHost);
      const double n = (double)(hs[32] ? hs[32] : 1);
      const char *nm[6] = {"stage", "smooth", "spmv", "restrict", "prolong", "writeback"};
      fprintf(stderr, "[tail stamps, ticks of s_memtime per launch over %.0f launches]", n);
      for (int l = 0; l < 4; ++l) for (int k = 0; k < 6; ++k) if (hs[l * 8 + k]) fprintf(stderr, " L%d.%s %.1f", l, nm[k], (double)hs[l * 8 + k] / n);
      fprintf(stderr, "\n");
    }
  }
#endif
  const int lds = off * 8;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_amg_tail), hipFuncAttributeMaxDynamicSharedMemorySize, lds);   // per device, per size
  hipLaunchKernelGGL(k_amg_tail, dim3(1), dim3(FEA_TAIL_T), lds, c->stream, A);
}

// returns where the level's result is: x or y (the post-smoothing sweep is fused with its product and writes the other
// vector); `part`: partial sums of result . r from that last launch (level 0: the CG's r.z), or null
static double *amg_cycle(feahip_ctx *c, int l, const double *r, double *x, double *y, double *part = nullptr)
{
  AmgHierarchy *h = H(c);
  AmgLevel &L = h->lv[l];
  const uint8_t *mask = l == 0 ? c->d_dofmask : (const uint8_t *)nullptr;
  const LevelRange R = level_range(c, l);
  hipLaunchKernelGGL(k_smooth_first, GSMOOTH(R), R.a0, R.a1, L.omega, L.minv, r, x);
  if (L.Nc == 0 && l > 0 && (h->coarse_sweeps & 1) == 0) {
    // coarsest level: product and damped Jacobi update in one launch per sweep, ping-pong between x and y
    double *a = x, *b = y;
    for (int s = 0; s < h->coarse_sweeps; ++s) {
      enq_spmv_jacobi(c->stream, 0, L.nchunks, L.chunk, L.rowptr, L.colidx, L.K, L.K32, nullptr, a, b, r, L.minv, L.omega, nullptr);
      double *t = a; a = b; b = t;
    }
    return x;                                          // an even number of sweeps ends in x
  }
  if (L.Nc == 0) {
    for (int s = 0; s < h->coarse_sweeps; ++s) {
      level_spmv(c, L, R, x, y);
      hipLaunchKernelGGL(k_smooth_next, GSMOOTH(R), R.a0, R.a1, L.omega, L.minv, r, y, x);
    }
    return x;
  }
  AmgLevel &C = h->lv[l + 1];
  const int gamma = (l < h->gamma_from || l >= h->gamma_until) ? 1 : h->gamma;
  for (int g = 0; g < gamma; ++g) {
    level_spmv(c, L, R, x, y);
    hipLaunchKernelGGL(k_restrict, G256((C.N / 2) * 16), C.N / 2, L.aptr, L.anodes, L.type, L.doff, r, y, mask, C.r);
    const double *xc = C.x;
    if (l + 1 == h->tail_from) launch_tail(c);
    else xc = amg_cycle(c, l + 1, C.r, C.x, C.y);
    // over-correction only where the correction is applied twice: (I - aE)^2 >= 0 for any a <= 2, while a single
    // over-corrected step can flip the sign of the preconditioner on part of the spectrum (seen: 6 492 iterations)
    hipLaunchKernelGGL(k_prolong, G256(L.N), L.N, L.agg, L.type, L.doff, xc, mask, gamma >= 2 ? h->over : fmin(h->over, 1.0), x);
  }
  if (!h->fused_post) {
    level_spmv(c, L, R, x, y);
    hipLaunchKernelGGL(k_smooth_next, GSMOOTH(R), R.a0, R.a1, L.omega, L.minv, r, y, x);
    return x;
  }
  enq_spmv_jacobi(c->stream, R.ch0, R.nch, L.chunk, L.rowptr, L.colidx, L.K, L.K32, L.K16, x, y, r, L.minv, L.omega, part);
  return y;
}

static double *amg_vcycle(feahip_ctx *c, const double *r0, double *z0)
{
  return amg_cycle(c, 0, r0, z0, c->d_q);
}

// ---- interface to the PCG loop (kernels_solve.hip) ---------------------------
// hierarchy for the context's current row range + numeric setup for the current K
int amg_prepare(feahip_ctx *c)
{
  int rc;
  AmgHierarchy *h = H(c);
  if (h && (h->row0 != c->row0 || h->row1 != c->row1)) { amg_destroy(c); h = nullptr; }     // the shard changed
  if (!h && (rc = amg_create(c))) return rc;
  h = H(c);
  if (h->numeric_valid && h->num_epoch == c->k_epoch && h->num_bc == c->k_bc) return FEAHIP_OK;   // same K as last time
  if ((rc = amg_numeric(c))) return rc;
  h->num_epoch = c->k_epoch; h->num_bc = c->k_bc;
  return FEAHIP_OK;
}

// z = M^-1 r on the rank's rows (z stays zero elsewhere); q is used as scratch
double *amg_result(feahip_ctx *c) { return H(c)->result; }
double *amg_apply(feahip_ctx *c, const double *r)
{
  AmgHierarchy *h = H(c);
  h->result = amg_vcycle(c, r, h->d_z);
  return h->result;
}
