// kernels_solve.hip -- everything between assembly and the next assembly:
// prescribed-displacement handling, the linear solve, node updates.
//
//   solver_apply_prescribed_bc        fea_solver.c:1200-1257
//   solver_solve_slae (CG / PCG)      fea_solver.c:245-321 (libspmatrix)
//   cdot(f,u)                         dense_matrix.c:16-23 at fea_solver.c:208
//   solver_update_nodes_with_*        fea_solver.c:1259-1284
//
// Matrix: block CSR, 3x3 blocks row-major, block columns sorted.  Vectors:
// 3N doubles.  All reductions are two-stage with a fixed grid and a fixed
// order, so repeated runs give identical bits.
#include <type_traits>
#include "feahip_internal.h"
#include <map>

// ------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// the same sum in every lane (butterfly: a fixed order as well)
__device__ __forceinline__ double wave_sum_all(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// sum over the 256 threads of a block; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double *scratch /*[4]*/)
{
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0) r = scratch[0] + scratch[1] + scratch[2] + scratch[3];
  __syncthreads();
  return r;
}

// every block re-reduces the producer kernel's partial sums, in fixed order;
// result broadcast to all threads
__device__ __forceinline__ double reduce_partials(const double *part, int n, double *scratch /*[5]*/)
{
  double v = 0;
  for (int i = threadIdx.x; i < n; i += 256) v += part[i];
  v = block_sum(v, scratch);
  if (threadIdx.x == 0) scratch[4] = v;
  __syncthreads();
  v = scratch[4];
  __syncthreads();
  return v;
}

// ------------------------------------------------------------------------
// prescribed displacements
// ------------------------------------------------------------------------
__device__ __forceinline__ int bc_find_block(const int *rowptr, const int *colidx, int row, int col)
{
  int lo = rowptr[row], hi = rowptr[row + 1] - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (colidx[mid] < col) lo = mid + 1; else hi = mid;
  }
  return (colidx[lo] == col) ? lo : -1;
}

// f[r] -= K[r,c] * p  over the stored rows r of column c (fea_solver.c:1250-1252).
// Only free rows matter: a constrained row's entry is overwritten by
// f[c] = K[c,c]*p (:1256) whatever the processing order.  A rank of a sharded
// solve touches the rows it owns, [a0, a1), and nothing else.
__global__ void k_bc_rhs(int n_cdof, const int *cdof, const double *cval, double lambda,
                         const int *rowptr, const int *colidx, const double *K,
                         const uint8_t *mask, double *f, int a0, int a1)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_cdof) return;
  const double p = cval[t] * lambda;
  if (p == 0.0) return;
  const int c = cdof[t], cn = c / 3, cj = c % 3;
  for (int q = rowptr[cn]; q < rowptr[cn + 1]; ++q) {
    const int b = colidx[q];
    if (b < a0 || b >= a1) continue;
    const int tb = bc_find_block(rowptr, colidx, b, cn);   // block (b, cn)
    if (tb < 0) continue;
    for (int i = 0; i < 3; ++i) {
      const int r = 3 * b + i;
      if (!mask[r]) atomicAdd(f + r, -K[(size_t)tb * 9 + 3 * i + cj] * p);
    }
  }
}

// sp_matrix_cross_cancellation + f[c] = K[c,c]*p (fea_solver.c:1254-1256)
__global__ void k_bc_cancel(int n_cdof, const int *cdof, const double *cval, double lambda,
                            const int *rowptr, const int *colidx, double *K, double *f, int a0, int a1)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_cdof) return;
  const int c = cdof[t], cn = c / 3, ci = c % 3;
  const bool own_row = cn >= a0 && cn < a1;
  for (int q = rowptr[cn]; q < rowptr[cn + 1]; ++q) {
    const int b = colidx[q];
    double *blk = K + (size_t)q * 9;             // block (cn, b): row ci
    if (own_row)
      for (int j = 0; j < 3; ++j)
        if (!(b == cn && j == ci)) blk[3 * ci + j] = 0.0;
    if (b >= a0 && b < a1) {
      const int tb = bc_find_block(rowptr, colidx, b, cn);   // block (b, cn): column ci
      if (tb >= 0) {
        double *tblk = K + (size_t)tb * 9;
        for (int i = 0; i < 3; ++i)
          if (!(b == cn && i == ci)) tblk[3 * i + ci] = 0.0;
      }
    }
    if (b == cn && own_row) f[c] = blk[3 * ci + ci] * (cval[t] * lambda);
  }
}

// solver_update_node_with_bc (fea_solver.c:1259-1266); x is [N][4]
__global__ void k_nodes_bc(int n_cdof, const int *cdof, const double *cval, double lambda, double *x)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_cdof) return;
  const int c = cdof[t];
  atomicAdd(x + (size_t)(c / 3) * 4 + c % 3, cval[t] * lambda);
}

// solver_update_nodes_with_solution (fea_solver.c:1270-1279)
__global__ void k_nodes_add(int ndof, const double *u, double *x)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ndof) return;
  x[(size_t)(t / 3) * 4 + t % 3] += u[t];
}

int launch_apply_bc(feahip_ctx *c, double lambda)
{
  if (c->n_cdof == 0) return FEAHIP_OK;
  const int grid = (c->n_cdof + 255) / 256;
  if (lambda != 0.0)
    hipLaunchKernelGGL(k_bc_rhs, dim3(grid), dim3(256), 0, c->stream, c->n_cdof, c->d_cdof, c->d_cval,
                       lambda, c->d_rowptr, c->d_colidx, c->d_K, c->d_dofmask, c->d_f, c->row0, c->row1);
  hipLaunchKernelGGL(k_bc_cancel, dim3(grid), dim3(256), 0, c->stream, c->n_cdof, c->d_cdof, c->d_cval,
                     lambda, c->d_rowptr, c->d_colidx, c->d_K, c->d_f, c->row0, c->row1);
  FEA_HIP_CHECK(c, hipGetLastError());
  return FEAHIP_OK;
}

int launch_update_nodes_bc(feahip_ctx *c, double lambda)
{
  if (c->n_cdof == 0) return FEAHIP_OK;
  hipLaunchKernelGGL(k_nodes_bc, dim3((c->n_cdof + 255) / 256), dim3(256), 0, c->stream,
                     c->n_cdof, c->d_cdof, c->d_cval, lambda, c->d_x);
  FEA_HIP_CHECK(c, hipGetLastError());
  return FEAHIP_OK;
}

int launch_update_nodes_solution(feahip_ctx *c, const double *d_uv)
{
  hipLaunchKernelGGL(k_nodes_add, dim3((c->ndof + 255) / 256), dim3(256), 0, c->stream,
                     c->ndof, d_uv, c->d_x);
  FEA_HIP_CHECK(c, hipGetLastError());
  return FEAHIP_OK;
}

// ------------------------------------------------------------------------
// SpMV  y = K x  (+ optional partial sums of dotwith . y)
//
// One wave owns a chunk of consecutive block rows (the assembly's chunks): a
// contiguous run of at most 128 3x3 blocks.  Lane k takes blocks k and k+64:
// it loads its blocks' nine values, their column index and the three x
// entries they multiply -- all 26 loads of a lane are independent and in
// flight together, nothing is staged -- and leaves the 3-vector K_k x_col(k)
// in LDS; lane (row,i) then adds up its row's partial products in block
// order.  No atomics, fixed summation order; HBM sees each matrix byte once
// (the 72-byte blocks of neighbouring lanes share cache lines, so the nine
// strided loads of a wave hit L1 after the first touch).
// ------------------------------------------------------------------------
// nine values of block kk as doubles.  bf16_t: the block is stored as three rows of four bfloat16 (three values and a
// pad: 24 bytes, three 8-byte loads instead of nine 2-byte ones)
struct bf16_t { unsigned short v[4]; };
#ifndef FEA_SPMV_STAGED
#define FEA_SPMV_STAGED 1      // bit 0: double matrices, 1: float, 2: bfloat16 (measured: only the 72-byte blocks gain)
#endif
typedef double spmv_v2d __attribute__((ext_vector_type(2)));
template <class TK>
__device__ __forceinline__ void load_block9(const TK *K, size_t kk, double (&v)[9])
{
  const TK *vp = K + kk * 9;
#pragma unroll
  for (int q = 0; q < 9; ++q) v[q] = (double)vp[q];
}
template <>
__device__ __forceinline__ void load_block9<bf16_t>(const bf16_t *K, size_t kk, double (&v)[9])
{
  const uint2 *vp = reinterpret_cast<const uint2 *>(K) + kk * 3;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const uint2 w = vp[i];
    v[3 * i] = (double)__uint_as_float(w.x << 16);
    v[3 * i + 1] = (double)__uint_as_float(w.x & 0xFFFF0000u);
    v[3 * i + 2] = (double)__uint_as_float(w.y << 16);
  }
}
// nine values and the column index of block kk.  bf16_t blocks carry their index in the pads of rows 0 and 1
// (k_to_bf16, amg.hip): no index load
template <class TK>
__device__ __forceinline__ int load_block9_col(const TK *K, const int *colidx, size_t kk, double (&v)[9])
{
  load_block9<TK>(K, kk, v);
  return colidx[kk];
}
template <>
__device__ __forceinline__ int load_block9_col<bf16_t>(const bf16_t *K, const int *, size_t kk, double (&v)[9])
{
  const uint2 *vp = reinterpret_cast<const uint2 *>(K) + kk * 3;
  uint2 w[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    w[i] = vp[i];
    v[3 * i] = (double)__uint_as_float(w[i].x << 16);
    v[3 * i + 1] = (double)__uint_as_float(w[i].x & 0xFFFF0000u);
    v[3 * i + 2] = (double)__uint_as_float(w[i].y << 16);
  }
  return (int)((w[0].y >> 16) | (w[1].y & 0xFFFF0000u));
}
// JAC: y = x + omega D^-1 (r - K x) instead of y = K x (one damped block-Jacobi sweep fused with its product: the rows of
// a chunk -- at most FEA_CHUNK_ROWS = 16, i.e. 48 lanes -- exchange their three residual components by shuffles and apply
// the inverse diagonal block; y must not be x).  dotwith / part: partial sums of y . dotwith in either mode.
struct SpmvJacobi { const double *r, *minv; double omega; };
template <class TK, bool JAC = false>
__device__ __forceinline__ void spmv_body(int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx, const TK *K,
                                          const double *x, double *y, const double *dotwith, double *part, const int *flag,
                                          const SpmvJacobi J = SpmvJacobi{nullptr, nullptr, 0.0})
{
  // per wave: the partial products (3 per block); the staged path first holds the chunk's values there
  constexpr int BB = std::is_same<TK, double>::value ? 72 : (std::is_same<TK, float>::value ? 36 : 24);   // bytes per block
  static_assert(FEA_CHUNK_ROWS * 3 <= 64, "one lane per (row, component) of a chunk");
  constexpr int NPMAX = (15 + FEA_CHUNK_BLOCKS * BB + 15) >> 4;                                             // 16-byte pieces of a chunk
  constexpr bool STAGED = ((FEA_SPMV_STAGED) & (BB == 72 ? 1 : (BB == 36 ? 2 : 4))) != 0;                    // bit per matrix type
  constexpr int SPD = STAGED ? (2 * NPMAX > FEA_CHUNK_BLOCKS * 3 ? 2 * NPMAX : FEA_CHUNK_BLOCKS * 3) : FEA_CHUNK_BLOCKS * 3;
  __shared__ __attribute__((aligned(16))) double sP[FEA_WAVES_PER_WG][SPD];
  __shared__ double scratch[5];
  if (flag && flag[0] != 0) return;          // solve already converged (uniform)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *tP = sP[wave];
  double dsum = 0;
  for (int ch = chunk0 + blockIdx.x * FEA_WAVES_PER_WG + wave; ch < chunk0 + nchunks; ch += gridDim.x * FEA_WAVES_PER_WG) {
    const int r0 = chunk[ch], r1 = chunk[ch + 1];
    const int b0 = rowptr[r0], nb = rowptr[r1] - b0;
    if (nb > FEA_CHUNK_BLOCKS) {
      // a row with more blocks than the tile: pattern.cpp / amg_setup.cpp give it a chunk of its own.  The lanes
      // stride over its blocks and their partial products meet in a butterfly: no tile, any length, fixed order
      // (the reference's solvers have no row-length limit either, fea_solver.c:300-321)
      double a0 = 0, a1 = 0, a2 = 0;
      for (int k = lane; k < nb; k += 64) {
        double vp[9];
        const int col = load_block9_col<TK>(K, colidx, (size_t)(b0 + k), vp);
        const double x0 = x[(size_t)col * 3], x1 = x[(size_t)col * 3 + 1], x2 = x[(size_t)col * 3 + 2];
        a0 += vp[0] * x0 + vp[1] * x1 + vp[2] * x2;
        a1 += vp[3] * x0 + vp[4] * x1 + vp[5] * x2;
        a2 += vp[6] * x0 + vp[7] * x1 + vp[8] * x2;
      }
      a0 = wave_sum_all(a0); a1 = wave_sum_all(a1); a2 = wave_sum_all(a2);
      if (lane < 3) {
        double acc = lane == 0 ? a0 : lane == 1 ? a1 : a2;
        if constexpr (JAC) {
          const double *rr = J.r + (size_t)r0 * 3;
          const double t0 = rr[0] - a0, t1 = rr[1] - a1, t2 = rr[2] - a2;
          const double *m = J.minv + (size_t)r0 * 9 + 3 * lane;
          acc = x[(size_t)r0 * 3 + lane] + J.omega * (m[0] * t0 + m[1] * t1 + m[2] * t2);
        }
        y[(size_t)r0 * 3 + lane] = acc;
        if (dotwith) dsum += acc * dotwith[(size_t)r0 * 3 + lane];
      }
      continue;
    }
    double v[2][9], xv[2][3];
    // lane t < 3 rows sums component t % 3 of row r0 + t / 3 (a chunk has at most FEA_CHUNK_ROWS = 16 rows: one pass);
    // its block range is loaded here, with everything else, not after the products
    int kb = 0, ke = 0;
    if (lane < (r1 - r0) * 3) { kb = rowptr[r0 + lane / 3] - b0; ke = rowptr[r0 + lane / 3 + 1] - b0; }
    double jm[3] = {0, 0, 0}, jr = 0, jx = 0;              // JAC: the lane's row of D^-1, r and x, likewise
    if constexpr (JAC) {
      if (lane < (r1 - r0) * 3) {
        const double *m = J.minv + (size_t)(r0 + lane / 3) * 9 + 3 * (lane % 3);
        jm[0] = m[0]; jm[1] = m[1]; jm[2] = m[2];
        jr = J.r[(size_t)r0 * 3 + lane]; jx = x[(size_t)r0 * 3 + lane];
      }
    }
    if constexpr (STAGED) {
      // the chunk's values are one contiguous run of nb blocks: the wave reads it as lane-contiguous 16-byte pieces
      // (1 KB per load instruction, every cache line looked up once) and the lanes pick their blocks out of LDS.
      // A lane loading its own 72-byte block touches 36 lines per load instruction, five instructions per block.
      constexpr int NJ = (NPMAX + 63) / 64;
      const size_t s0 = (size_t)b0 * BB, a0 = s0 & ~(size_t)15;      // the arrays are 16-byte aligned and padded by 16 bytes
      const int sh = (int)(s0 - a0), np = (sh + nb * BB + 15) >> 4;
      const spmv_v2d *Kp = reinterpret_cast<const spmv_v2d *>(reinterpret_cast<const char *>(K) + a0);
      spmv_v2d *sW = reinterpret_cast<spmv_v2d *>(tP);
      int col[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) { const int k = lane + 64 * h; col[h] = colidx[k < nb ? b0 + k : b0]; }
      spmv_v2d pc[NJ];
#pragma unroll
      for (int j = 0; j < NJ - 1; ++j) { const int p = lane + 64 * j; pc[j] = Kp[p < np ? p : 0]; }
      if (np > 64 * (NJ - 1)) { const int p = lane + 64 * (NJ - 1); pc[NJ - 1] = Kp[p < np ? p : 0]; }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 3; ++i) xv[h][i] = x[(size_t)col[h] * 3 + i];
#pragma unroll
      for (int j = 0; j < NJ - 1; ++j) { const int p = lane + 64 * j; if (p < np) sW[p] = pc[j]; }
      if (np > 64 * (NJ - 1)) { const int p = lane + 64 * (NJ - 1); if (p < np) sW[p] = pc[NJ - 1]; }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = lane + 64 * h;
        const char *bp = reinterpret_cast<const char *>(tP) + sh + BB * (k < nb ? k : 0);
        if constexpr (BB == 24) {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const uint2 w = reinterpret_cast<const uint2 *>(bp)[i];
            v[h][3 * i] = (double)__uint_as_float(w.x << 16);
            v[h][3 * i + 1] = (double)__uint_as_float(w.x & 0xFFFF0000u);
            v[h][3 * i + 2] = (double)__uint_as_float(w.y << 16);
          }
        } else {
          using TV = typename std::conditional<BB == 72, double, float>::type;
#pragma unroll
          for (int q = 0; q < 9; ++q) v[h][q] = (double)reinterpret_cast<const TV *>(bp)[q];
        }
      }
      // (the products below overwrite the staged values: a wave's LDS operations complete in program order, and every
      // lane has read its blocks by then)
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    } else {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = lane + 64 * h;
      const bool on = k < nb;
      const int kk = on ? b0 + k : b0;
      const int col = load_block9_col<TK>(K, colidx, (size_t)kk, v[h]);
#pragma unroll
      for (int i = 0; i < 3; ++i) xv[h][i] = x[(size_t)col * 3 + i];
    }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = lane + 64 * h;
      if (k < nb) {
        tP[k * 3 + 0] = v[h][0] * xv[h][0] + v[h][1] * xv[h][1] + v[h][2] * xv[h][2];
        tP[k * 3 + 1] = v[h][3] * xv[h][0] + v[h][4] * xv[h][1] + v[h][5] * xv[h][2];
        tP[k * 3 + 2] = v[h][6] * xv[h][0] + v[h][7] * xv[h][1] + v[h][8] * xv[h][2];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if constexpr (JAC) {
      const int t = lane, nt = (r1 - r0) * 3;            // <= 48: one pass
      const int i = t % 3;
      double res = 0.0;
      if (t < nt) {
        double acc = 0;
        for (int k = kb; k < ke; ++k) acc += tP[k * 3 + i];
        res = jr - acc;
      }
      const double t0 = __shfl(res, lane - i), t1 = __shfl(res, lane - i + 1), t2 = __shfl(res, lane - i + 2);
      if (t < nt) {
        const double xn = jx + J.omega * (jm[0] * t0 + jm[1] * t1 + jm[2] * t2);
        y[(size_t)r0 * 3 + t] = xn;
        if (dotwith) dsum += xn * dotwith[(size_t)r0 * 3 + t];
      }
    } else {
    const int t = lane;
    if (t < (r1 - r0) * 3) {
      const int i = t % 3;
      double acc = 0;
      for (int k = kb; k < ke; ++k) acc += tP[k * 3 + i];
      y[(size_t)r0 * 3 + t] = acc;
      if (dotwith) dsum += acc * dotwith[(size_t)r0 * 3 + t];
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }
  if (part) {
    const double s = block_sum(dsum, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(256)
void k_spmv(int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx, const double *K,
            const double *x, double *y, const double *dotwith, double *part, const int *flag)
{
  spmv_body<double>(chunk0, nchunks, chunk, rowptr, colidx, K, x, y, dotwith, part, flag);
}

// the same product with the matrix stored in single precision (coarse levels of the multigrid
// preconditioner: half the bytes; vectors and arithmetic stay double)
__global__ __launch_bounds__(256)
void k_spmv_f32(int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx, const float *K,
                const double *x, double *y)
{
  spmv_body<float>(chunk0, nchunks, chunk, rowptr, colidx, K, x, y, (const double *)nullptr, (double *)nullptr, (const int *)nullptr);
}

// ... and in bfloat16 (the smoother of level 0: the 16 high bits of the float, a third of the double matrix's bytes with
// the index; 10M-tet block: the same 95 CG iterations as with the float copy)
__global__ __launch_bounds__(256)
void k_spmv_bf16(int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx, const bf16_t *K,
                 const double *x, double *y)
{
  spmv_body<bf16_t>(chunk0, nchunks, chunk, rowptr, colidx, K, x, y, (const double *)nullptr, (double *)nullptr, (const int *)nullptr);
}

// One damped block-Jacobi sweep fused with its product: xout = xin + omega D^-1 (r - K xin) (spmv_body<TK, true>: same
// partial products, same order; one launch instead of two, no y vector).  The sweeps on the coarsest level of the
// multigrid cycle and the post-smoothing of every level; `part`: partial sums of xout . r (the CG's r.z after the last
// step of a cycle), or null.
template <class TK>
__global__ __launch_bounds__(256)
void k_spmv_jacobi(int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx, const TK *K,
                   const double *xin, double *xout, const double *r, const double *minv, double omega, double *part)
{
  spmv_body<TK, true>(chunk0, nchunks, chunk, rowptr, colidx, K, xin, xout, part ? r : (const double *)nullptr, part,
                      (const int *)nullptr, SpmvJacobi{r, minv, omega});
}

// exactly one of K (double), K32 (float), K16 (bfloat16 rows of four) is given
void enq_spmv_jacobi(hipStream_t stream, int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx, const double *K,
                     const float *K32, const unsigned short *K16, const double *xin, double *xout, const double *r, const double *minv,
                     double omega, double *part)
{
  int g = (nchunks + FEA_WAVES_PER_WG - 1) / FEA_WAVES_PER_WG;
  g = g < FEA_RED_BLOCKS ? (g > 0 ? g : 1) : FEA_RED_BLOCKS;
#define SJ(TK, ptr) hipLaunchKernelGGL(k_spmv_jacobi<TK>, dim3(g), dim3(256), 0, stream, chunk0, nchunks, chunk, rowptr, colidx, ptr, xin, xout, r, minv, omega, part)
  if (K16) SJ(bf16_t, reinterpret_cast<const bf16_t *>(K16));
  else if (K32) SJ(float, K32);
  else SJ(double, K);
#undef SJ
}

static int spmv_grid(const feahip_ctx *c)
{
  int g = (c->nchunks_local + FEA_WAVES_PER_WG - 1) / FEA_WAVES_PER_WG;
  return g < FEA_RED_BLOCKS ? (g > 0 ? g : 1) : FEA_RED_BLOCKS;
}

int launch_spmv(feahip_ctx *c, const double *d_xv, double *d_yv)
{
  hipLaunchKernelGGL(k_spmv, dim3(spmv_grid(c)), dim3(256), 0, c->stream, c->chunk0, c->nchunks_local, c->d_chunk,
                     c->d_rowptr, c->d_colidx, c->d_K, d_xv, d_yv, (const double *)nullptr,
                     (double *)nullptr, (const int *)nullptr);
  FEA_HIP_CHECK(c, hipGetLastError());
  return FEAHIP_OK;
}

// ------------------------------------------------------------------------
// dot product over the owned rows (two-stage, fixed order)
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_dot_partial(int i0, int i1, const double *a, const double *b, double *part)
{
  __shared__ double scratch[5];
  double v = 0;
  for (int i = i0 + blockIdx.x * 256 + threadIdx.x; i < i1; i += gridDim.x * 256) v += a[i] * b[i];
  v = block_sum(v, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = v;
}

// out[k] = sum of part[k*stride .. k*stride+n)  for k < nsums  (one block)
__global__ __launch_bounds__(256)
void k_reduce_final(int n, int nsums, int stride, const double *part, double *out)
{
  __shared__ double scratch[5];
  for (int k = 0; k < nsums; ++k) {
    const double v = reduce_partials(part + (size_t)k * stride, n, scratch);
    if (threadIdx.x == 0) out[k] = v;
  }
}

// ------------------------------------------------------------------------
// preconditioned conjugate gradients
//
// Device scalars (d_scal): [0],[1] r.z ping-pong, [2] b.b, [3] last r.r,
// [4] tolerance^2, [8..11] reduction results / all-reduce buffer.
// d_flag[0] = iteration at which the stop test fired (0 = still running,
// <0 = breakdown).  Partial-sum arrays in d_part: [0..RB) p.q, [RB..2RB) r.z,
// [2RB..3RB) r.r, [3RB..4RB) b.b.
// A rank of a sharded solve owns the nodes [a0, a1); every kernel below
// touches owned rows only.  gred != nullptr: the sums have already been
// reduced over all ranks (all-reduce) and are read from gred instead of the
// partial arrays.
// ------------------------------------------------------------------------
#define RB FEA_RED_BLOCKS

// multigrid preconditioner (amg.hip); used when the context asks for it and the solve is not plain CG
int amg_prepare(feahip_ctx *c);
double *amg_apply(feahip_ctx *c, const double *r);
double *amg_result(feahip_ctx *c);
static inline bool use_amg(const feahip_ctx *c, int mode) { return mode != 0 && c->precond == 1; }

// 3x3 inverse of the diagonal blocks (block-Jacobi); mode 0 = identity
__global__ void k_precond_build(int a0, int a1, const int *diag, const double *K, int mode, double *minv)
{
  const int a = a0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= a1) return;
  double *o = minv + (size_t)a * 9;
  double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (mode != 0) {
    const double *d = K + (size_t)diag[a] * 9;
    const double det = d[0] * (d[4] * d[8] - d[5] * d[7]) - d[1] * (d[3] * d[8] - d[5] * d[6]) +
                       d[2] * (d[3] * d[7] - d[4] * d[6]);
    if (det != 0.0 && det == det) {
      const double id = 1.0 / det;
      m[0] = (d[4] * d[8] - d[5] * d[7]) * id; m[1] = (d[2] * d[7] - d[1] * d[8]) * id; m[2] = (d[1] * d[5] - d[2] * d[4]) * id;
      m[3] = (d[5] * d[6] - d[3] * d[8]) * id; m[4] = (d[0] * d[8] - d[2] * d[6]) * id; m[5] = (d[2] * d[3] - d[0] * d[5]) * id;
      m[6] = (d[3] * d[7] - d[4] * d[6]) * id; m[7] = (d[1] * d[6] - d[0] * d[7]) * id; m[8] = (d[0] * d[4] - d[1] * d[3]) * id;
    }
  }
  for (int i = 0; i < 9; ++i) o[i] = m[i];
}

// r = b - q ; p = M r ; partial sums r.z, r.r, b.b     (q = A x0, x0 = b)
__global__ __launch_bounds__(256)
void k_cg_init(int a0, int a1, const double *b, const double *q, const double *minv, double *r, double *p,
               double *part)
{
  __shared__ double scratch[5];
  double srz = 0, srr = 0, sbb = 0;
  for (int a = a0 + blockIdx.x * 256 + threadIdx.x; a < a1; a += gridDim.x * 256) {
    const double *m = minv + (size_t)a * 9;
    double rv[3], bv[3];
    for (int i = 0; i < 3; ++i) {
      bv[i] = b[(size_t)a * 3 + i];
      rv[i] = bv[i] - q[(size_t)a * 3 + i];
      r[(size_t)a * 3 + i] = rv[i];
    }
    for (int i = 0; i < 3; ++i) {
      const double z = minv ? m[3 * i] * rv[0] + m[3 * i + 1] * rv[1] + m[3 * i + 2] * rv[2] : rv[i];
      p[(size_t)a * 3 + i] = z;
      srz += rv[i] * z; srr += rv[i] * rv[i]; sbb += bv[i] * bv[i];
    }
  }
  srz = block_sum(srz, scratch); srr = block_sum(srr, scratch); sbb = block_sum(sbb, scratch);
  if (threadIdx.x == 0) { part[RB + blockIdx.x] = srz; part[2 * RB + blockIdx.x] = srr; part[3 * RB + blockIdx.x] = sbb; }
}

// gred (if given) holds {r.z, r.r, b.b} summed over all ranks
__global__ __launch_bounds__(256)
void k_cg_init_scalars(int nparts, const double *part, const double *gred, double *scal, double tol, int *flag)
{
  __shared__ double scratch[5];
  double rz, rr, bb;
  if (gred) { rz = gred[0]; rr = gred[1]; bb = gred[2]; }
  else {
    rz = reduce_partials(part + RB, nparts, scratch);
    rr = reduce_partials(part + 2 * RB, nparts, scratch);
    bb = reduce_partials(part + 3 * RB, nparts, scratch);
  }
  if (threadIdx.x == 0) {
    scal[0] = rz; scal[1] = rz; scal[2] = bb; scal[3] = rr; scal[4] = tol * tol;
    // a zero right-hand side (or an exact start vector) is already solved
    flag[0] = (rr <= tol * tol * bb || rz == 0.0) ? -1000000000 : 0;
  }
}

// alpha = r.z / p.q ; x += alpha p ; r -= alpha q ; partial sums of r.Mr, r.r.
// zq (may alias q, which is dead once read): z = M r is left there for
// k_cg_direction, which then reads one vector instead of r and the 72-byte
// inverse blocks again.
__global__ __launch_bounds__(256)
void k_cg_update(int a0, int a1, int it, int n_pq, const double *p, const double *q, const double *minv,
                 double *x, double *r, double *part, const double *gred, const double *scal, const int *flag,
                 double *zq)
{
  __shared__ double scratch[5];
  if (flag[0] != 0) return;
  const double pq = gred ? gred[0] : reduce_partials(part, n_pq, scratch);
  const double rz = scal[it & 1];
  const double alpha = rz / pq;
  double srz = 0, srr = 0;
  // lane <-> scalar dof, 21 nodes per wave and step: every vector is read as one contiguous run per instruction, and
  // so is D^-1 (row i of node a's block is the three doubles at 3 (3a + i)); the three residual components of a node
  // meet by shuffles.  (A lane per node read its nine inverse values with a 72-byte stride between lanes.)
  const int lane = threadIdx.x & 63, i = lane % 3;
  const bool lane_on = lane < 63;
  for (long long nb = a0 + (long long)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 21; nb < a1; nb += (long long)gridDim.x * 4 * 21) {
    const long long a = nb + lane / 3;
    const bool on = lane_on && a < a1;
    const size_t k = (size_t)nb * 3 + lane;
    double rv = 0;
    if (on) { x[k] += alpha * p[k]; rv = r[k] - alpha * q[k]; r[k] = rv; }
    double z = rv;
    if (minv) {
      double m0 = 0, m1 = 0, m2 = 0;
      if (on) { const double *m = minv + 3 * k; m0 = m[0]; m1 = m[1]; m2 = m[2]; }
      const double r0 = __shfl(rv, lane - i), r1 = __shfl(rv, lane - i + 1), r2 = __shfl(rv, lane - i + 2);
      z = m0 * r0 + m1 * r1 + m2 * r2;
    }
    if (on) { srz += rv * z; srr += rv * rv; if (zq) zq[k] = z; }
  }
  srz = block_sum(srz, scratch); srr = block_sum(srr, scratch);
  if (threadIdx.x == 0) { part[RB + blockIdx.x] = srz; part[2 * RB + blockIdx.x] = srr; }
}

// beta = r.z_new / r.z_old ; p = M r + beta p ; stop test on r.r
// gred (if given) holds {r.z_new, r.r} summed over all ranks
__global__ __launch_bounds__(256)
void k_cg_direction(int a0, int a1, int it, int nparts, const double *r, const double *minv, double *p,
                    const double *part, const double *gred, double *scal, int *flag)
{
  __shared__ double scratch[5];
  if (flag[0] != 0) return;
  const double rz_new = gred ? gred[0] : reduce_partials(part + RB, nparts, scratch);
  const double rr = gred ? gred[1] : reduce_partials(part + 2 * RB, nparts, scratch);
  const double rz_old = scal[it & 1];
  const bool stop = rr <= scal[4] * scal[2];
  const bool broke = !(rz_new == rz_new) || !(rr == rr) || rz_old == 0.0;
  if (!stop && !broke) {
    const double beta = rz_new / rz_old;
    if (!minv) {                                        // z is given: one contiguous run per instruction
      for (size_t k = (size_t)a0 * 3 + blockIdx.x * 256 + threadIdx.x; k < (size_t)a1 * 3; k += (size_t)gridDim.x * 256)
        p[k] = r[k] + beta * p[k];
    } else
    for (int a = a0 + blockIdx.x * 256 + threadIdx.x; a < a1; a += gridDim.x * 256) {
      const double *m = minv + (size_t)a * 9;
      const double r0 = r[(size_t)a * 3], r1 = r[(size_t)a * 3 + 1], r2 = r[(size_t)a * 3 + 2];
      for (int i = 0; i < 3; ++i) {
        const size_t k = (size_t)a * 3 + i;
        const double z = minv ? m[3 * i] * r0 + m[3 * i + 1] * r1 + m[3 * i + 2] * r2 : (i == 0 ? r0 : (i == 1 ? r1 : r2));
        p[k] = z + beta * p[k];
      }
    }
  }
  // The writes below go to the other ping-pong slot of scal; the flag is read
  // only at kernel entry.  A block of THIS launch that starts after the flag
  // was set returns early, which is what stop / broke would have made it do.
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[(it + 1) & 1] = rz_new;
    scal[3] = rr;
    if (broke) flag[0] = -(it + 1);
    else if (stop) flag[0] = it + 1;
  }
}

// ------------------------------------------------------------------------
// single-reduction PCG (Chronopoulos / Gear form of the same recurrence).  Vectors: x, r, z = M r, w = K z,
// p = z + beta p, s = w + beta s (= K p).  Sums of an iteration: gamma = r.z, delta = w.z, rr = r.r -- formed
// together, one all-reduce.  Device scalars: [0],[1] gamma ping-pong, [5],[6] alpha ping-pong, [2] b.b, [3] last r.r,
// [4] tolerance^2; gred = {gamma, delta, rr} of the state the iteration starts from (reduced over all ranks).
// Partial sums in d_part: [0..RB) delta (rows that touch no halo column), [RB..2RB) gamma, [2RB..3RB) rr,
// [3RB..4RB) b.b at the start, then delta of the rows in front of the interior range, [4RB..5RB) delta behind it.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_cgcg_update(int a0, int a1, int it, const double *z, const double *w, const double *minv, double *p, double *s,
                   double *x, double *r, double *znew, double *part, const double *gred, double *scal, int *flag)
{
  __shared__ double scratch[5];
  if (flag[0] != 0) return;
  const double gamma = gred[0], delta = gred[1], rr = gred[2];
  const bool stop = rr <= scal[4] * scal[2];
  double beta = 0.0, alpha = 0.0;
  bool broke = !(gamma == gamma) || !(delta == delta) || !(rr == rr);
  if (!stop && !broke) {
    if (it == 0) alpha = gamma / delta;
    else {
      const double gamma_old = scal[(it + 1) & 1], alpha_old = scal[5 + ((it + 1) & 1)];
      beta = gamma / gamma_old;
      alpha = gamma / (delta - beta * gamma / alpha_old);
    }
    broke = !(alpha == alpha) || !(beta == beta) || alpha == 0.0;
  }
  double sg = 0, srr = 0;
  if (!stop && !broke) {                                // lane <-> scalar dof, 21 nodes per wave and step (k_cg_update)
    const int lane = threadIdx.x & 63, i = lane % 3;
    const bool lane_on = lane < 63;
    for (long long nb = a0 + (long long)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 21; nb < a1; nb += (long long)gridDim.x * 4 * 21) {
      const bool on = lane_on && nb + lane / 3 < a1;
      const size_t k = (size_t)nb * 3 + lane;
      double rv = 0;
      if (on) {
        const double pk = z[k] + beta * p[k], sk = w[k] + beta * s[k];
        p[k] = pk; s[k] = sk;
        x[k] += alpha * pk;
        rv = r[k] - alpha * sk;
        r[k] = rv;
      }
      srr += rv * rv;
      if (minv) {                                       // block-Jacobi: z = M r here; multigrid: the cycle follows
        double m0 = 0, m1 = 0, m2 = 0;
        if (on) { const double *m = minv + 3 * k; m0 = m[0]; m1 = m[1]; m2 = m[2]; }
        const double r0 = __shfl(rv, lane - i), r1 = __shfl(rv, lane - i + 1), r2 = __shfl(rv, lane - i + 2);
        const double zz = m0 * r0 + m1 * r1 + m2 * r2;
        if (on) { znew[k] = zz; sg += rv * zz; }
      }
    }
  }
  sg = block_sum(sg, scratch); srr = block_sum(srr, scratch);
  if (threadIdx.x == 0) { part[RB + blockIdx.x] = sg; part[2 * RB + blockIdx.x] = srr; }
  // the scalars of this iteration go to the other ping-pong slots; the flag is read at kernel entry only
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[it & 1] = gamma; scal[5 + (it & 1)] = alpha; scal[3] = rr;
    if (broke) flag[0] = -(it + 1);
    else if (stop) flag[0] = it > 0 ? it : -1000000000;
  }
}

// znew = z (the multigrid cycle's result) on the owned rows, partial sums of r.z
__global__ __launch_bounds__(256)
void k_copy_dot(int i0, int i1, const double *z, const double *r, double *znew, double *part, const int *flag)
{
  __shared__ double scratch[5];
  if (flag && flag[0] != 0) return;
  double v = 0;
  for (int i = i0 + blockIdx.x * 256 + threadIdx.x; i < i1; i += gridDim.x * 256) { const double zi = z[i]; znew[i] = zi; v += r[i] * zi; }
  v = block_sum(v, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = v;
}

// out = { sum gamma partials, sum of the three delta partial ranges, sum rr partials [, sum bb partials] } (one block, fixed order)
__global__ __launch_bounds__(256)
void k_cgcg_reduce(int gv, int g_int, int g_pre, int g_suf, int with_bb, const double *part, double *out)
{
  __shared__ double scratch[5];
  const double g = reduce_partials(part + RB, gv, scratch);
  double d = reduce_partials(part, g_int, scratch);
  if (g_pre > 0) d += reduce_partials(part + (with_bb ? 5 : 3) * RB, g_pre, scratch);
  if (g_suf > 0) d += reduce_partials(part + 4 * RB, g_suf, scratch);
  const double rr = reduce_partials(part + 2 * RB, gv, scratch);
  const double bb = with_bb ? reduce_partials(part + 3 * RB, gv, scratch) : 0.0;
  if (threadIdx.x == 0) { out[0] = g; out[1] = d; out[2] = rr; if (with_bb) out[3] = bb; }
}

__global__ void k_cgcg_scalars0(const double *gred, double *scal, double tol, int *flag)
{
  if (threadIdx.x || blockIdx.x) return;
  scal[2] = gred[3]; scal[3] = gred[2]; scal[4] = tol * tol;
  flag[0] = (gred[2] <= tol * tol * gred[3] || gred[0] == 0.0) ? -1000000000 : 0;   // a zero right-hand side (or an exact start vector) is already solved
}

// halo rows of a 3N vector: buf[3i+j] = v[3 idx[i] + j] and back
__global__ void k_halo_pack(int n, const int *idx, const double *v, int vstride, double *buf)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 3) return;
  buf[t] = v[(size_t)idx[t / 3] * vstride + t % 3];
}
__global__ void k_halo_unpack(int n, const int *idx, const double *buf, int vstride, double *v)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * 3) return;
  v[(size_t)idx[t / 3] * vstride + t % 3] = buf[t];
}

static int vec_grid_range(int n)
{
  int g = (n + 255) / 256;
  return g < FEA_RED_BLOCKS ? (g > 0 ? g : 1) : FEA_RED_BLOCKS;
}

// ------------------------------------------------------------------------
// launch helpers (one rank)
// ------------------------------------------------------------------------
static inline int own0(const feahip_ctx *c) { return c->row0; }
static inline int own1(const feahip_ctx *c) { return c->row1; }
static inline int vgrid(const feahip_ctx *c) { return vec_grid_range(c->row1 - c->row0); }

static void enq_precond(feahip_ctx *c, int mode)
{
  const int n = own1(c) - own0(c);
  hipLaunchKernelGGL(k_precond_build, dim3((n + 255) / 256 > 0 ? (n + 255) / 256 : 1), dim3(256), 0, c->stream,
                     own0(c), own1(c), c->d_diag, c->d_K, mode, c->d_minv);
}

static void enq_spmv_dot(feahip_ctx *c, const double *xv, double *yv, const double *dotwith, double *part)
{
  hipLaunchKernelGGL(k_spmv, dim3(spmv_grid(c)), dim3(256), 0, c->stream, c->chunk0, c->nchunks_local, c->d_chunk,
                     c->d_rowptr, c->d_colidx, c->d_K, xv, yv, dotwith, part, dotwith ? c->d_flag : (const int *)nullptr);
}

// ------------------------------------------------------------------------
// transports
// ------------------------------------------------------------------------
// test knob of the in-process transport: the halo rows of a vector become NaN (GroupTransport::exchange_begin)
__global__ void k_halo_poison(int n, const int *__restrict__ idx, int stride, double *__restrict__ v)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * 3) v[(size_t)idx[i / 3] * stride + i % 3] = __builtin_nan("");
}

static double *halo_vec(feahip_ctx *c, int which, int &stride)
{
  stride = (which == 2) ? 4 : 3;
  return which == 0 ? c->d_p : (which == 1 ? c->d_u : (which == 2 ? c->d_x : c->d_z));
}

static void enq_pack(feahip_ctx *c, int which)
{
  if (c->nsend <= 0) return;
  int stride; double *v = halo_vec(c, which, stride);
  hipLaunchKernelGGL(k_halo_pack, dim3((c->nsend * 3 + 255) / 256), dim3(256), 0, c->stream, c->nsend,
                     c->d_send_idx, v, stride, c->d_send_buf);
}

static void enq_unpack(feahip_ctx *c, int which)
{
  if (c->nrecv <= 0) return;
  int stride; double *v = halo_vec(c, which, stride);
  hipLaunchKernelGGL(k_halo_unpack, dim3((c->nrecv * 3 + 255) / 256), dim3(256), 0, c->stream, c->nrecv,
                     c->d_recv_idx, c->d_recv_buf, stride, v);
}

void feahip_enq_pack(feahip_ctx *c, double *v, int stride)
{
  if (c->nsend <= 0) return;
  hipLaunchKernelGGL(k_halo_pack, dim3((c->nsend * 3 + 255) / 256), dim3(256), 0, c->stream, c->nsend,
                     c->d_send_idx, v, stride, c->d_send_buf);
}
void feahip_enq_unpack_on(feahip_ctx *c, double *v, int stride, hipStream_t stream)
{
  if (c->nrecv <= 0) return;
  hipLaunchKernelGGL(k_halo_unpack, dim3((c->nrecv * 3 + 255) / 256), dim3(256), 0, stream, c->nrecv,
                     c->d_recv_idx, c->d_recv_buf, stride, v);
}
void feahip_enq_unpack(feahip_ctx *c, double *v, int stride)
{
  if (c->nrecv <= 0) return;
  hipLaunchKernelGGL(k_halo_unpack, dim3((c->nrecv * 3 + 255) / 256), dim3(256), 0, c->stream, c->nrecv,
                     c->d_recv_idx, c->d_recv_buf, stride, v);
}

// All ranks live in this process (any mix of devices): copies between the
// ranks' buffers, sums on the host.  Used by the in-process group API -- the
// way to run and test the sharded solve where one process sees the GPUs.
struct GroupTransport : Transport {
  int exchange(std::vector<feahip_ctx *> &R, int which) override
  {
    for (auto *c : R) { (void)hipSetDevice(c->device); enq_pack(c, which); }
    for (auto *c : R) { (void)hipSetDevice(c->device); FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream)); }
    for (auto *a : R)
      for (size_t k = 0; k < a->peer.size(); ++k) {
        feahip_ctx *b = R[(size_t)a->peer[k]];
        size_t kb = 0;
        while (kb < b->peer.size() && b->peer[kb] != a->rank) ++kb;
        const int n = a->send_off[k + 1] - a->send_off[k];
        if (kb == b->peer.size() || b->recv_off[kb + 1] - b->recv_off[kb] != n) {
          a->err = "halo plans of two ranks disagree"; return FEAHIP_ECOMM;
        }
        // on the RECEIVER's stream: ordered before its unpack kernel (a blocking
        // hipMemcpy on the null stream is not ordered against a non-blocking stream)
        (void)hipSetDevice(b->device);
        FEA_HIP_CHECK(b, hipMemcpyAsync(b->d_recv_buf + (size_t)3 * b->recv_off[kb], a->d_send_buf + (size_t)3 * a->send_off[k],
                                        sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToDevice, b->stream));
      }
    for (auto *c : R) { (void)hipSetDevice(c->device); enq_unpack(c, which); }
    return FEAHIP_OK;
  }
  // The same exchange in two halves, as the RCCL transport runs it (dist.hip): pack on the context's stream, copies and
  // unpack on a communication stream per context, events instead of host synchronisation -- so that the in-process
  // tests execute the choreography the multi-process runs rely on: between begin and end the context's stream
  // multiplies the rows that touch no halo column WHILE the halo rows arrive.
  // FEAHIP_TEST_POISON_HALO=1 (test knob; results must not change): begin() overwrites the halo rows with NaN and the
  // copies are held back until everything enqueued between begin and end has run -- a product that reads a halo row
  // too early then reads NaN every time, not only when a race goes the wrong way.
  bool poison = false, poison_read = false;
  int which_pending = 0;
  std::map<feahip_ctx *, hipEvent_t> ev_interior;      // test knob only: "everything enqueued up to exchange_end has run"
  ~GroupTransport() override { for (auto &kv : ev_interior) (void)hipEventDestroy(kv.second); }
  static int ensure_streams(feahip_ctx *c)
  {
    if (!c->comm_stream) {
      FEA_HIP_CHECK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
      FEA_HIP_CHECK(c, hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
      FEA_HIP_CHECK(c, hipEventCreateWithFlags(&c->ev_unpacked, hipEventDisableTiming));
    }
    return FEAHIP_OK;
  }
  int copies_and_unpack(std::vector<feahip_ctx *> &R, int which)
  {
    for (auto *b : R) {                            // receiver by receiver, on ITS communication stream
      (void)hipSetDevice(b->device);
      for (size_t kb = 0; kb < b->peer.size(); ++kb) {
        feahip_ctx *a = R[(size_t)b->peer[kb]];
        size_t k = 0;
        while (k < a->peer.size() && a->peer[k] != b->rank) ++k;
        const int n = b->recv_off[kb + 1] - b->recv_off[kb];
        if (k == a->peer.size() || a->send_off[k + 1] - a->send_off[k] != n) { b->err = "halo plans of two ranks disagree"; return FEAHIP_ECOMM; }
        FEA_HIP_CHECK(b, hipStreamWaitEvent(b->comm_stream, a->ev_packed, 0));
        if (poison) FEA_HIP_CHECK(b, hipStreamWaitEvent(b->comm_stream, ev_interior[a], 0));
        if (n > 0)
          FEA_HIP_CHECK(b, hipMemcpyAsync(b->d_recv_buf + (size_t)3 * b->recv_off[kb], a->d_send_buf + (size_t)3 * a->send_off[k],
                                          sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToDevice, b->comm_stream));
      }
      // the receiver's own stream has packed (and, under the test knob, poisoned and multiplied) before its halo rows change
      FEA_HIP_CHECK(b, hipStreamWaitEvent(b->comm_stream, poison ? ev_interior[b] : b->ev_packed, 0));
      int stride; double *v = halo_vec(b, which, stride);
      feahip_enq_unpack_on(b, v, stride, b->comm_stream);
      FEA_HIP_CHECK(b, hipEventRecord(b->ev_unpacked, b->comm_stream));
    }
    return FEAHIP_OK;
  }
  int exchange_begin(std::vector<feahip_ctx *> &R, int which) override
  {
    if (!poison_read) { const char *e = getenv("FEAHIP_TEST_POISON_HALO"); poison = e && atoi(e) > 0; poison_read = true; }
    int rc;
    for (auto *c : R) {
      (void)hipSetDevice(c->device);
      if ((rc = ensure_streams(c))) return rc;
      if (poison && !ev_interior.count(c)) { hipEvent_t e; FEA_HIP_CHECK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); ev_interior[c] = e; }
    }
    for (auto *c : R) {
      (void)hipSetDevice(c->device);
      // the send buffer is free once every receiver of the previous exchange has copied it out
      for (int pk : c->peer) FEA_HIP_CHECK(c, hipStreamWaitEvent(c->stream, R[(size_t)pk]->ev_unpacked, 0));
      enq_pack(c, which);
      if (poison && c->nrecv > 0) {
        int stride; double *v = halo_vec(c, which, stride);
        hipLaunchKernelGGL(k_halo_poison, dim3((c->nrecv * 3 + 255) / 256), dim3(256), 0, c->stream, c->nrecv, c->d_recv_idx, stride, v);
      }
      FEA_HIP_CHECK(c, hipEventRecord(c->ev_packed, c->stream));
    }
    which_pending = which;
    return poison ? FEAHIP_OK : copies_and_unpack(R, which);
  }
  int exchange_end(std::vector<feahip_ctx *> &R) override
  {
    if (poison) {                                  // the copies start only now: behind everything enqueued since begin()
      for (auto *c : R) { (void)hipSetDevice(c->device); FEA_HIP_CHECK(c, hipEventRecord(ev_interior[c], c->stream)); }
      const int rc = copies_and_unpack(R, which_pending);
      if (rc) return rc;
    }
    for (auto *c : R) { (void)hipSetDevice(c->device); FEA_HIP_CHECK(c, hipStreamWaitEvent(c->stream, c->ev_unpacked, 0)); }
    return FEAHIP_OK;
  }
  int allreduce(std::vector<feahip_ctx *> &R, int slot, int n) override
  {
    double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tmp[8];
    for (auto *c : R) {                          // fixed rank order: reproducible
      (void)hipSetDevice(c->device);
      FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
      FEA_HIP_CHECK(c, hipMemcpyAsync(tmp, c->d_scal + 8 + slot, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
      FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
      for (int i = 0; i < n; ++i) sum[i] += tmp[i];
    }
    for (auto *c : R) {
      (void)hipSetDevice(c->device);
      FEA_HIP_CHECK(c, hipMemcpyAsync(c->d_scal + 8 + slot, sum, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
      FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    }
    return FEAHIP_OK;
  }
};
Transport *make_group_transport() { return new GroupTransport(); }

// ------------------------------------------------------------------------
// multi-rank PCG.  With one rank and no transport this is the plain solver.
// ------------------------------------------------------------------------
#define FOR_RANKS(c) for (feahip_ctx *c : R) if (hipSetDevice(c->device) == hipSuccess)

static int enq_cg_iteration(std::vector<feahip_ctx *> &R, Transport *T, int it, int mode)
{
  int rc;
  if (T && (rc = T->exchange(R, 0))) return rc;                  // halo rows of p
  FOR_RANKS(c) {
    enq_spmv_dot(c, c->d_p, c->d_q, c->d_p, c->d_part);
    if (T) hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, c->stream, spmv_grid(c), 1, RB, c->d_part, c->d_scal + 8);
  }
  if (T && (rc = T->allreduce(R, 0, 1))) return rc;              // p.q
  FOR_RANKS(c) {
    const int gv = vgrid(c);
    const bool amg = use_amg(c, mode);
    // block-Jacobi: z = M r is formed here and left in q; multigrid: plain update, then the cycle
    hipLaunchKernelGGL(k_cg_update, dim3(gv), dim3(256), 0, c->stream, own0(c), own1(c), it, spmv_grid(c), c->d_p,
                       c->d_q, amg ? (const double *)nullptr : c->d_minv, c->d_u, c->d_r, c->d_part,
                       T ? c->d_scal + 8 : (const double *)nullptr, c->d_scal, c->d_flag, amg ? (double *)nullptr : c->d_q);
    if (amg) {
      const double *z = amg_apply(c, c->d_r);                      // local: block-Jacobi over the ranks, a W-cycle inside
      hipLaunchKernelGGL(k_dot_partial, dim3(gv), dim3(256), 0, c->stream, 3 * own0(c), 3 * own1(c), c->d_r, z, c->d_part + RB);
    }
    if (T) hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, c->stream, gv, 2, RB, c->d_part + RB, c->d_scal + 9);
  }
  if (T && (rc = T->allreduce(R, 1, 2))) return rc;              // r.z, r.r
  FOR_RANKS(c) {
    const int gv = vgrid(c);
    const double *z = use_amg(c, mode) ? amg_result(c) : c->d_q;
    hipLaunchKernelGGL(k_cg_direction, dim3(gv), dim3(256), 0, c->stream, own0(c), own1(c), it, gv, z,
                       (const double *)nullptr, c->d_p, c->d_part, T ? c->d_scal + 9 : (const double *)nullptr, c->d_scal,
                       c->d_flag);
  }
  return FEAHIP_OK;
}

static int enq_cg_start(std::vector<feahip_ctx *> &R, Transport *T, int mode, double tol)
{
  int rc;
  FOR_RANKS(c) {
    if (use_amg(c, mode)) { if ((rc = amg_prepare(c))) return rc; }
    else enq_precond(c, mode);
    FEA_HIP_CHECK(c, hipMemcpyAsync(c->d_u, c->d_f, sizeof(double) * (size_t)c->ndof, hipMemcpyDeviceToDevice, c->stream));
  }
  if (T && (rc = T->exchange(R, 1))) return rc;                  // halo rows of u0 = f
  FOR_RANKS(c) {
    const int gv = vgrid(c);
    const bool amg = use_amg(c, mode);
    enq_spmv_dot(c, c->d_u, c->d_q, nullptr, nullptr);
    hipLaunchKernelGGL(k_cg_init, dim3(gv), dim3(256), 0, c->stream, own0(c), own1(c), c->d_f, c->d_q,
                       amg ? (const double *)nullptr : c->d_minv, c->d_r, c->d_p, c->d_part);
    if (amg) {                                                   // p = z = M^-1 r from the cycle, r.z from it
      const double *z = amg_apply(c, c->d_r);
      hipLaunchKernelGGL(k_dot_partial, dim3(gv), dim3(256), 0, c->stream, 3 * own0(c), 3 * own1(c), c->d_r, z, c->d_part + RB);
      FEA_HIP_CHECK(c, hipMemcpyAsync(c->d_p + (size_t)3 * own0(c), z + (size_t)3 * own0(c),
                                      sizeof(double) * 3 * (size_t)(own1(c) - own0(c)), hipMemcpyDeviceToDevice, c->stream));
    }
    if (T) hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, c->stream, gv, 3, RB, c->d_part + RB, c->d_scal + 8);
  }
  if (T && (rc = T->allreduce(R, 0, 3))) return rc;              // r.z, r.r, b.b
  FOR_RANKS(c) {
    hipLaunchKernelGGL(k_cg_init_scalars, dim3(1), dim3(256), 0, c->stream, vgrid(c), c->d_part,
                       T ? c->d_scal + 8 : (const double *)nullptr, c->d_scal, tol, c->d_flag);
    FEA_HIP_CHECK(c, hipGetLastError());
  }
  return FEAHIP_OK;
}

// ---- single-reduction variant ------------------------------------------------
static int ensure_cgcg(feahip_ctx *c)
{
  if (!c->d_z) {
    FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_z, sizeof(double) * (size_t)c->ndof));
    FEA_HIP_CHECK(c, hipMemsetAsync(c->d_z, 0, sizeof(double) * (size_t)c->ndof, c->stream));
  }
  if (!c->d_w) {
    FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_w, sizeof(double) * (size_t)c->ndof));
    FEA_HIP_CHECK(c, hipMemsetAsync(c->d_w, 0, sizeof(double) * (size_t)c->ndof, c->stream));
  }
  if (!c->d_s) {                                     // s = K p by recurrence (q stays the multigrid cycle's level-0 scratch)
    FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_s, sizeof(double) * (size_t)c->ndof));
    FEA_HIP_CHECK(c, hipMemsetAsync(c->d_s, 0, sizeof(double) * (size_t)c->ndof, c->stream));
  }
  return FEAHIP_OK;
}

static int spmv_grid_n(int nchunks)
{
  int g = (nchunks + FEA_WAVES_PER_WG - 1) / FEA_WAVES_PER_WG;
  return g < FEA_RED_BLOCKS ? (g > 0 ? g : 1) : FEA_RED_BLOCKS;
}

// w = K z with the partial sums of w.z, in three launches: the chunks that touch no halo column (they may run while
// the halo rows of z travel: between exchange_begin and exchange_end), then the chunks in front of and behind them
static void enq_spmv_interior(feahip_ctx *c, const double *zv, double *wv)
{
  const int n = c->ichunk_hi - c->ichunk_lo;
  if (n > 0)
    hipLaunchKernelGGL(k_spmv, dim3(spmv_grid_n(n)), dim3(256), 0, c->stream, c->chunk0 + c->ichunk_lo, n, c->d_chunk,
                       c->d_rowptr, c->d_colidx, c->d_K, zv, wv, zv, c->d_part, (const int *)c->d_flag);
}
static void enq_spmv_boundary(feahip_ctx *c, const double *zv, double *wv, int pre_slot)
{
  const int npre = c->ichunk_lo, nsuf = c->nchunks_local - c->ichunk_hi;
  if (npre > 0)
    hipLaunchKernelGGL(k_spmv, dim3(spmv_grid_n(npre)), dim3(256), 0, c->stream, c->chunk0, npre, c->d_chunk,
                       c->d_rowptr, c->d_colidx, c->d_K, zv, wv, zv, c->d_part + (size_t)pre_slot * RB, (const int *)c->d_flag);
  if (nsuf > 0)
    hipLaunchKernelGGL(k_spmv, dim3(spmv_grid_n(nsuf)), dim3(256), 0, c->stream, c->chunk0 + c->ichunk_hi, nsuf, c->d_chunk,
                       c->d_rowptr, c->d_colidx, c->d_K, zv, wv, zv, c->d_part + (size_t)4 * RB, (const int *)c->d_flag);
}
static void enq_cgcg_reduce(feahip_ctx *c, int with_bb)
{
  const int n_int = c->ichunk_hi - c->ichunk_lo, npre = c->ichunk_lo, nsuf = c->nchunks_local - c->ichunk_hi;
  hipLaunchKernelGGL(k_cgcg_reduce, dim3(1), dim3(256), 0, c->stream, vgrid(c), n_int > 0 ? spmv_grid_n(n_int) : 0,
                     npre > 0 ? spmv_grid_n(npre) : 0, nsuf > 0 ? spmv_grid_n(nsuf) : 0, with_bb, c->d_part, c->d_scal + 8);
}

static int enq_cgcg_start(std::vector<feahip_ctx *> &R, Transport *T, int mode, double tol)
{
  int rc;
  FOR_RANKS(c) {
    if ((rc = ensure_cgcg(c))) return rc;
    if (use_amg(c, mode)) { if ((rc = amg_prepare(c))) return rc; }
    else enq_precond(c, mode);
    FEA_HIP_CHECK(c, hipMemcpyAsync(c->d_u, c->d_f, sizeof(double) * (size_t)c->ndof, hipMemcpyDeviceToDevice, c->stream));
    FEA_HIP_CHECK(c, hipMemsetAsync(c->d_p, 0, sizeof(double) * (size_t)c->ndof, c->stream));     // p = s = 0: the first iteration has beta = 0
    FEA_HIP_CHECK(c, hipMemsetAsync(c->d_s, 0, sizeof(double) * (size_t)c->ndof, c->stream));
  }
  if (T && (rc = T->exchange(R, 1))) return rc;                  // halo rows of u0 = f
  FOR_RANKS(c) {
    const int gv = vgrid(c);
    const bool amg = use_amg(c, mode);
    enq_spmv_dot(c, c->d_u, c->d_w, nullptr, nullptr);           // K x0 (into w, overwritten below)
    // r = b - K x0, z = M r, partial sums r.z, r.r, b.b
    hipLaunchKernelGGL(k_cg_init, dim3(gv), dim3(256), 0, c->stream, own0(c), own1(c), c->d_f, c->d_w,
                       amg ? (const double *)nullptr : c->d_minv, c->d_r, c->d_z, c->d_part);
    if (amg) {
      const double *z = amg_apply(c, c->d_r);
      hipLaunchKernelGGL(k_copy_dot, dim3(gv), dim3(256), 0, c->stream, 3 * own0(c), 3 * own1(c), z, c->d_r, c->d_z, c->d_part + RB, (const int *)nullptr);
    }
  }
  if (T && (rc = T->exchange_begin(R, 3))) return rc;            // halo rows of z
  FOR_RANKS(c) enq_spmv_interior(c, c->d_z, c->d_w);
  if (T && (rc = T->exchange_end(R))) return rc;
  FOR_RANKS(c) {
    enq_spmv_boundary(c, c->d_z, c->d_w, 5);                     // (slot 3 holds b.b at the start)
    enq_cgcg_reduce(c, 1);
  }
  if (T && (rc = T->allreduce(R, 0, 4))) return rc;              // gamma, delta, rr, bb
  FOR_RANKS(c) {
    hipLaunchKernelGGL(k_cgcg_scalars0, dim3(1), dim3(1), 0, c->stream, c->d_scal + 8, c->d_scal, tol, c->d_flag);
    FEA_HIP_CHECK(c, hipGetLastError());
  }
  return FEAHIP_OK;
}

static int enq_cgcg_iteration(std::vector<feahip_ctx *> &R, Transport *T, int it, int mode)
{
  int rc;
  FOR_RANKS(c) {
    const int gv = vgrid(c);
    const bool amg = use_amg(c, mode);
    hipLaunchKernelGGL(k_cgcg_update, dim3(gv), dim3(256), 0, c->stream, own0(c), own1(c), it, c->d_z, c->d_w,
                       amg ? (const double *)nullptr : c->d_minv, c->d_p, c->d_s, c->d_u, c->d_r, c->d_z, c->d_part,
                       c->d_scal + 8, c->d_scal, c->d_flag);
    if (amg) {
      const double *z = amg_apply(c, c->d_r);                      // local: block-Jacobi over the ranks, a W-cycle inside
      hipLaunchKernelGGL(k_copy_dot, dim3(gv), dim3(256), 0, c->stream, 3 * own0(c), 3 * own1(c), z, c->d_r, c->d_z, c->d_part + RB, (const int *)c->d_flag);
    }
  }
  if (T && (rc = T->exchange_begin(R, 3))) return rc;            // halo rows of z on their way ...
  FOR_RANKS(c) enq_spmv_interior(c, c->d_z, c->d_w);             // ... under the rows that do not need them
  if (T && (rc = T->exchange_end(R))) return rc;
  FOR_RANKS(c) {
    enq_spmv_boundary(c, c->d_z, c->d_w, 3);
    enq_cgcg_reduce(c, 0);
  }
  if (T && (rc = T->allreduce(R, 0, 3))) return rc;              // gamma, delta, rr: the one reduction of the iteration
  return FEAHIP_OK;
}

static inline bool use_cgcg(const feahip_ctx *c, Transport *T) { return c->pcg_variant == 1 || (c->pcg_variant < 0 && T != nullptr); }

// Solves K u = f by (preconditioned) CG started from u0 = f, the start vector
// the reference hands to sp_matrix_yale_solve_cg (fea_solver.c:251-256).
int dist_solve_pcg(std::vector<feahip_ctx *> &R, int type, double tol, int max_iter, int *iters, double *resid)
{
  Transport *T = R[0]->tr;
  feahip_ctx *c0 = R[0];
  const int mode = (type == FEAHIP_CG) ? 0 : 1;
  if (type == FEAHIP_CHOLESKY) { tol = 1e-16; if (max_iter < 100000) max_iter = 100000; }
  const bool cgcg = use_cgcg(c0, T);
  int rc = cgcg ? enq_cgcg_start(R, T, mode, tol) : enq_cg_start(R, T, mode, tol);
  if (rc) return rc;
  int flag = 0, it = 0;
  const int batch = use_amg(c0, mode) ? 8 : 32;
  while (it < max_iter) {
    const int n = (max_iter - it < batch) ? (max_iter - it) : batch;
    for (int k = 0; k < n; ++k)
      if ((rc = cgcg ? enq_cgcg_iteration(R, T, it + k, mode) : enq_cg_iteration(R, T, it + k, mode))) return rc;
    it += n;
    (void)hipSetDevice(c0->device);
    FEA_HIP_CHECK(c0, hipGetLastError());
    FEA_HIP_CHECK(c0, hipMemcpyAsync(&flag, c0->d_flag, sizeof(int), hipMemcpyDeviceToHost, c0->stream));
    FEA_HIP_CHECK(c0, hipStreamSynchronize(c0->stream));
    if (flag != 0) break;           // every rank holds the same all-reduced sums, hence the same flag
  }
  double sc[5];
  FEA_HIP_CHECK(c0, hipMemcpyAsync(sc, c0->d_scal, sizeof(sc), hipMemcpyDeviceToHost, c0->stream));
  FEA_HIP_CHECK(c0, hipStreamSynchronize(c0->stream));
  int done_it = it;
  if (flag == -1000000000) done_it = 0;
  else if (flag > 0) done_it = flag;
  else if (flag < 0) done_it = -flag;
  if (iters) *iters = done_it;
  if (resid) *resid = (sc[2] > 0) ? sqrt(sc[3] / sc[2]) : sqrt(sc[3]);
  // leave the flag clear so stand-alone SpMV launches are not skipped
  FOR_RANKS(c) {
    FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    FEA_HIP_CHECK(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
  }
  if (flag < 0 && flag != -1000000000) {
    c0->err = "CG breakdown (NaN or zero curvature) at iteration " + std::to_string(-flag);
    return FEAHIP_ENOTCONVERGED;
  }
  return FEAHIP_OK;
}

int solve_pcg(feahip_ctx *c, int type, double tol, int max_iter, int *iters, double *resid)
{
  std::vector<feahip_ctx *> R(1, c);
  return dist_solve_pcg(R, type, tol, max_iter, iters, resid);
}

// cdot(f, u) over all ranks (fea_solver.c:208-210)
int dist_energy(std::vector<feahip_ctx *> &R, double *out)
{
  Transport *T = R[0]->tr;
  int rc;
  FOR_RANKS(c) {
    const int g = vec_grid_range(3 * (own1(c) - own0(c)));
    hipLaunchKernelGGL(k_dot_partial, dim3(g), dim3(256), 0, c->stream, 3 * own0(c), 3 * own1(c), c->d_f, c->d_u, c->d_part);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, c->stream, g, 1, RB, c->d_part, c->d_scal + 8);
    FEA_HIP_CHECK(c, hipGetLastError());
  }
  if (T && (rc = T->allreduce(R, 0, 1))) return rc;
  feahip_ctx *c0 = R[0];
  (void)hipSetDevice(c0->device);
  FEA_HIP_CHECK(c0, hipMemcpyAsync(out, c0->d_scal + 8, sizeof(double), hipMemcpyDeviceToHost, c0->stream));
  FEA_HIP_CHECK(c0, hipStreamSynchronize(c0->stream));
  return FEAHIP_OK;
}

// solver_update_nodes_with_solution (fea_solver.c:1270-1279) on every rank:
// owners' increments travel to the halo copies first, then x += u everywhere
// (u is zero outside a rank's owned and halo nodes).
int dist_update_nodes_with_solution(std::vector<feahip_ctx *> &R, const double *u_host)
{
  Transport *T = R[0]->tr;
  int rc;
  FOR_RANKS(c) {
    c->state_valid = false;
    if (u_host) FEA_HIP_CHECK(c, hipMemcpyAsync(c->d_u, u_host, sizeof(double) * (size_t)c->ndof, hipMemcpyHostToDevice, c->stream));
  }
  if (T && !u_host && (rc = T->exchange(R, 1))) return rc;
  FOR_RANKS(c) {
    hipLaunchKernelGGL(k_nodes_add, dim3((c->ndof + 255) / 256), dim3(256), 0, c->stream, c->ndof, c->d_u, c->d_x);
    FEA_HIP_CHECK(c, hipGetLastError());
  }
  return FEAHIP_OK;
}

// x += eta u on every rank (line search: trial configurations along the Newton step).
// `exchange`: the owners' u travels to the halo copies first (once per step is enough).
__global__ void k_nodes_axpy(int ndof, double eta, const double *u, double *x)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ndof) return;
  x[(size_t)(t / 3) * 4 + t % 3] += eta * u[t];
}

int dist_nodes_add_scaled(std::vector<feahip_ctx *> &R, double eta, bool exchange)
{
  Transport *T = R[0]->tr;
  int rc;
  FOR_RANKS(c) c->state_valid = false;
  if (T && exchange && (rc = T->exchange(R, 1))) return rc;
  FOR_RANKS(c) {
    hipLaunchKernelGGL(k_nodes_axpy, dim3((c->ndof + 255) / 256), dim3(256), 0, c->stream, c->ndof, eta, c->d_u, c->d_x);
    FEA_HIP_CHECK(c, hipGetLastError());
  }
  return FEAHIP_OK;
}

int time_pcg_iteration(feahip_ctx *c, int warmup, int iters, double *avg_ms)
{
  std::vector<feahip_ctx *> R(1, c);
  Transport *T = c->tr;
  int rc;
  const bool cgcg = use_cgcg(c, T);
  rc = cgcg ? enq_cgcg_start(R, T, 1, 0.0) : enq_cg_start(R, T, 1, 0.0);
  if (rc) return rc;
  FEA_HIP_CHECK(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
  hipEvent_t e0, e1;
  FEA_HIP_CHECK(c, hipEventCreate(&e0));
  FEA_HIP_CHECK(c, hipEventCreate(&e1));
  for (int k = 0; k < warmup; ++k) if ((rc = cgcg ? enq_cgcg_iteration(R, T, k, 1) : enq_cg_iteration(R, T, k, 1))) return rc;
  FEA_HIP_CHECK(c, hipEventRecord(e0, c->stream));
  for (int k = 0; k < iters; ++k) if ((rc = cgcg ? enq_cgcg_iteration(R, T, warmup + k, 1) : enq_cg_iteration(R, T, warmup + k, 1))) return rc;
  FEA_HIP_CHECK(c, hipEventRecord(e1, c->stream));
  FEA_HIP_CHECK(c, hipEventSynchronize(e1));
  float ms = 0;
  FEA_HIP_CHECK(c, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  FEA_HIP_CHECK(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
  *avg_ms = iters > 0 ? (double)ms / iters : 0.0;
  return FEAHIP_OK;
}

// y = K x on any level of a hierarchy (amg.hip): same kernel, explicit arrays
void enq_spmv_arrays_f32(hipStream_t stream, int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx,
                         const float *K, const double *xv, double *yv)
{
  int g = (nchunks + FEA_WAVES_PER_WG - 1) / FEA_WAVES_PER_WG;
  g = g < FEA_RED_BLOCKS ? (g > 0 ? g : 1) : FEA_RED_BLOCKS;
  hipLaunchKernelGGL(k_spmv_f32, dim3(g), dim3(256), 0, stream, chunk0, nchunks, chunk, rowptr, colidx, K, xv, yv);
}

void enq_spmv_arrays_bf16(hipStream_t stream, int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx,
                          const unsigned short *K, const double *xv, double *yv)
{
  int g = (nchunks + FEA_WAVES_PER_WG - 1) / FEA_WAVES_PER_WG;
  g = g < FEA_RED_BLOCKS ? (g > 0 ? g : 1) : FEA_RED_BLOCKS;
  hipLaunchKernelGGL(k_spmv_bf16, dim3(g), dim3(256), 0, stream, chunk0, nchunks, chunk, rowptr, colidx,
                     reinterpret_cast<const bf16_t *>(K), xv, yv);
}

void enq_spmv_arrays(hipStream_t stream, int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx,
                     const double *K, const double *xv, double *yv)
{
  int g = (nchunks + FEA_WAVES_PER_WG - 1) / FEA_WAVES_PER_WG;
  g = g < FEA_RED_BLOCKS ? (g > 0 ? g : 1) : FEA_RED_BLOCKS;
  hipLaunchKernelGGL(k_spmv, dim3(g), dim3(256), 0, stream, chunk0, nchunks, chunk, rowptr, colidx, K, xv, yv,
                     (const double *)nullptr, (double *)nullptr, (const int *)nullptr);
}

