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
#include "feahip_internal.h"

// ------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
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
// f[c] = K[c,c]*p (:1256) whatever the processing order.
__global__ void k_bc_rhs(int n_cdof, const int *cdof, const double *cval, double lambda,
                         const int *rowptr, const int *colidx, const double *K,
                         const uint8_t *mask, double *f)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_cdof) return;
  const double p = cval[t] * lambda;
  if (p == 0.0) return;
  const int c = cdof[t], cn = c / 3, cj = c % 3;
  for (int q = rowptr[cn]; q < rowptr[cn + 1]; ++q) {
    const int b = colidx[q];
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
                            const int *rowptr, const int *colidx, double *K, double *f)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_cdof) return;
  const int c = cdof[t], cn = c / 3, ci = c % 3;
  for (int q = rowptr[cn]; q < rowptr[cn + 1]; ++q) {
    const int b = colidx[q];
    double *blk = K + (size_t)q * 9;             // block (cn, b): row ci
    for (int j = 0; j < 3; ++j)
      if (!(b == cn && j == ci)) blk[3 * ci + j] = 0.0;
    const int tb = bc_find_block(rowptr, colidx, b, cn);   // block (b, cn): column ci
    if (tb >= 0) {
      double *tblk = K + (size_t)tb * 9;
      for (int i = 0; i < 3; ++i)
        if (!(b == cn && i == ci)) tblk[3 * i + ci] = 0.0;
    }
    if (b == cn) f[c] = blk[3 * ci + ci] * (cval[t] * lambda);
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
                       lambda, c->d_rowptr, c->d_colidx, c->d_K, c->d_dofmask, c->d_f);
  hipLaunchKernelGGL(k_bc_cancel, dim3(grid), dim3(256), 0, c->stream, c->n_cdof, c->d_cdof, c->d_cval,
                     lambda, c->d_rowptr, c->d_colidx, c->d_K, c->d_f);
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
__global__ __launch_bounds__(256)
void k_spmv(int chunk0, int nchunks, const int *chunk, const int *rowptr, const int *colidx, const double *K,
            const double *x, double *y, const double *dotwith, double *part, const int *flag)
{
  __shared__ double sP[FEA_WAVES_PER_WG][FEA_CHUNK_BLOCKS * 3];
  __shared__ double scratch[5];
  if (flag && flag[0] != 0) return;          // solve already converged (uniform)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *tP = sP[wave];
  double dsum = 0;
  for (int ch = chunk0 + blockIdx.x * FEA_WAVES_PER_WG + wave; ch < chunk0 + nchunks; ch += gridDim.x * FEA_WAVES_PER_WG) {
    const int r0 = chunk[ch], r1 = chunk[ch + 1];
    const int b0 = rowptr[r0], nb = rowptr[r1] - b0;
    double v[2][9], xv[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = lane + 64 * h;
      const bool on = k < nb;
      const int kk = on ? b0 + k : b0;
      const int col = colidx[kk];
      const double *vp = K + (size_t)kk * 9;
#pragma unroll
      for (int q = 0; q < 9; ++q) v[h][q] = vp[q];
#pragma unroll
      for (int i = 0; i < 3; ++i) xv[h][i] = x[(size_t)col * 3 + i];
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
    for (int t = lane; t < (r1 - r0) * 3; t += 64) {
      const int r = r0 + t / 3, i = t % 3;
      const int kb = rowptr[r] - b0, ke = rowptr[r + 1] - b0;
      double acc = 0;
      for (int k = kb; k < ke; ++k) acc += tP[k * 3 + i];
      y[(size_t)r0 * 3 + t] = acc;
      if (dotwith) dsum += acc * dotwith[(size_t)r0 * 3 + t];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  }
  if (part) {
    const double s = block_sum(dsum, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
  }
}

static int spmv_grid(const feahip_ctx *c)
{
  int g = (c->nchunks_local + FEA_WAVES_PER_WG - 1) / FEA_WAVES_PER_WG;
  return g < FEA_RED_BLOCKS ? (g > 0 ? g : 1) : FEA_RED_BLOCKS;
}
static int vec_grid(const feahip_ctx *c)
{
  int g = (c->ndof + 255) / 256;
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
// dot product (two-stage, fixed order)
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void k_dot_partial(int n, const double *a, const double *b, double *part)
{
  __shared__ double scratch[5];
  double v = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) v += a[i] * b[i];
  v = block_sum(v, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = v;
}

__global__ __launch_bounds__(256)
void k_reduce_final(int n, const double *part, double *out)
{
  __shared__ double scratch[5];
  const double v = reduce_partials(part, n, scratch);
  if (threadIdx.x == 0) out[0] = v;
}

int launch_dot(feahip_ctx *c, const double *a, const double *b, double *out_host)
{
  const int g = vec_grid(c);
  hipLaunchKernelGGL(k_dot_partial, dim3(g), dim3(256), 0, c->stream, c->ndof, a, b, c->d_part);
  hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, c->stream, g, c->d_part, c->d_scal + 8);
  FEA_HIP_CHECK(c, hipGetLastError());
  FEA_HIP_CHECK(c, hipMemcpyAsync(out_host, c->d_scal + 8, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return FEAHIP_OK;
}

// ------------------------------------------------------------------------
// preconditioned conjugate gradients
//
// Device scalars (d_scal): [0],[1] r.z ping-pong, [2] b.b, [3] last r.r,
// [4] tolerance^2.  d_flag[0] = iteration at which the stop test fired
// (0 = still running, <0 = breakdown).  Partial-sum arrays live in d_part:
// [0..RB) p.q, [RB..2RB) r.z, [2RB..3RB) r.r, [3RB..4RB) b.b.
// ------------------------------------------------------------------------
#define RB FEA_RED_BLOCKS

// 3x3 inverse of the diagonal blocks (block-Jacobi); mode 0 = identity
__global__ void k_precond_build(int N, const int *rowptr, const int *colidx, const double *K,
                                int mode, double *minv)
{
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= N) return;
  double *o = minv + (size_t)a * 9;
  double m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (mode != 0) {
    const int q = bc_find_block(rowptr, colidx, a, a);
    if (q >= 0) {
      const double *d = K + (size_t)q * 9;
      const double det = d[0] * (d[4] * d[8] - d[5] * d[7]) - d[1] * (d[3] * d[8] - d[5] * d[6]) +
                         d[2] * (d[3] * d[7] - d[4] * d[6]);
      if (det != 0.0 && det == det) {
        const double id = 1.0 / det;
        m[0] = (d[4] * d[8] - d[5] * d[7]) * id; m[1] = (d[2] * d[7] - d[1] * d[8]) * id; m[2] = (d[1] * d[5] - d[2] * d[4]) * id;
        m[3] = (d[5] * d[6] - d[3] * d[8]) * id; m[4] = (d[0] * d[8] - d[2] * d[6]) * id; m[5] = (d[2] * d[3] - d[0] * d[5]) * id;
        m[6] = (d[3] * d[7] - d[4] * d[6]) * id; m[7] = (d[1] * d[6] - d[0] * d[7]) * id; m[8] = (d[0] * d[4] - d[1] * d[3]) * id;
      }
    }
  }
  for (int i = 0; i < 9; ++i) o[i] = m[i];
}

// r = b - q ; p = M r ; partial sums r.z, r.r, b.b     (q = A x0, x0 = b)
__global__ __launch_bounds__(256)
void k_cg_init(int N, const double *b, const double *q, const double *minv, double *r, double *p,
               double *part)
{
  __shared__ double scratch[5];
  double srz = 0, srr = 0, sbb = 0;
  for (int a = blockIdx.x * 256 + threadIdx.x; a < N; a += gridDim.x * 256) {
    const double *m = minv + (size_t)a * 9;
    double rv[3], bv[3];
    for (int i = 0; i < 3; ++i) {
      bv[i] = b[(size_t)a * 3 + i];
      rv[i] = bv[i] - q[(size_t)a * 3 + i];
      r[(size_t)a * 3 + i] = rv[i];
    }
    for (int i = 0; i < 3; ++i) {
      const double z = m[3 * i] * rv[0] + m[3 * i + 1] * rv[1] + m[3 * i + 2] * rv[2];
      p[(size_t)a * 3 + i] = z;
      srz += rv[i] * z; srr += rv[i] * rv[i]; sbb += bv[i] * bv[i];
    }
  }
  srz = block_sum(srz, scratch); srr = block_sum(srr, scratch); sbb = block_sum(sbb, scratch);
  if (threadIdx.x == 0) { part[RB + blockIdx.x] = srz; part[2 * RB + blockIdx.x] = srr; part[3 * RB + blockIdx.x] = sbb; }
}

__global__ __launch_bounds__(256)
void k_cg_init_scalars(int nparts, const double *part, double *scal, double tol, int *flag)
{
  __shared__ double scratch[5];
  const double rz = reduce_partials(part + RB, nparts, scratch);
  const double rr = reduce_partials(part + 2 * RB, nparts, scratch);
  const double bb = reduce_partials(part + 3 * RB, nparts, scratch);
  if (threadIdx.x == 0) {
    scal[0] = rz; scal[1] = rz; scal[2] = bb; scal[3] = rr; scal[4] = tol * tol;
    // a zero right-hand side (or an exact start vector) is already solved
    flag[0] = (rr <= tol * tol * bb || rz == 0.0) ? -1000000000 : 0;
  }
}

// alpha = r.z / p.q ; x += alpha p ; r -= alpha q ; partial sums of r.Mr, r.r
__global__ __launch_bounds__(256)
void k_cg_update(int N, int it, int n_pq, const double *p, const double *q, const double *minv,
                 double *x, double *r, double *part, const double *scal, const int *flag)
{
  __shared__ double scratch[5];
  if (flag[0] != 0) return;
  const double pq = reduce_partials(part, n_pq, scratch);
  const double rz = scal[it & 1];
  const double alpha = rz / pq;
  double srz = 0, srr = 0;
  for (int a = blockIdx.x * 256 + threadIdx.x; a < N; a += gridDim.x * 256) {
    const double *m = minv + (size_t)a * 9;
    double rv[3];
    for (int i = 0; i < 3; ++i) {
      const size_t k = (size_t)a * 3 + i;
      x[k] += alpha * p[k];
      rv[i] = r[k] - alpha * q[k];
      r[k] = rv[i];
    }
    for (int i = 0; i < 3; ++i) {
      const double z = m[3 * i] * rv[0] + m[3 * i + 1] * rv[1] + m[3 * i + 2] * rv[2];
      srz += rv[i] * z; srr += rv[i] * rv[i];
    }
  }
  srz = block_sum(srz, scratch); srr = block_sum(srr, scratch);
  if (threadIdx.x == 0) { part[RB + blockIdx.x] = srz; part[2 * RB + blockIdx.x] = srr; }
}

// beta = r.z_new / r.z_old ; p = M r + beta p ; stop test on r.r
__global__ __launch_bounds__(256)
void k_cg_direction(int N, int it, int nparts, const double *r, const double *minv, double *p,
                    const double *part, double *scal, int *flag)
{
  __shared__ double scratch[5];
  if (flag[0] != 0) return;
  const double rz_new = reduce_partials(part + RB, nparts, scratch);
  const double rr = reduce_partials(part + 2 * RB, nparts, scratch);
  const double rz_old = scal[it & 1];
  const bool stop = rr <= scal[4] * scal[2];
  const bool broke = !(rz_new == rz_new) || !(rr == rr) || rz_old == 0.0;
  if (!stop && !broke) {
    const double beta = rz_new / rz_old;
    for (int a = blockIdx.x * 256 + threadIdx.x; a < N; a += gridDim.x * 256) {
      const double *m = minv + (size_t)a * 9;
      const double r0 = r[(size_t)a * 3], r1 = r[(size_t)a * 3 + 1], r2 = r[(size_t)a * 3 + 2];
      for (int i = 0; i < 3; ++i) {
        const size_t k = (size_t)a * 3 + i;
        p[k] = m[3 * i] * r0 + m[3 * i + 1] * r1 + m[3 * i + 2] * r2 + beta * p[k];
      }
    }
  }
  // every block has read scal[it&1] and flag[0] before anyone writes: the
  // writes below go to the other ping-pong slot; the flag is only read at
  // kernel entry, and the early return above happens before this point in
  // every block of THIS launch (flag was 0 for all of them).
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[(it + 1) & 1] = rz_new;
    scal[3] = rr;
    if (broke) flag[0] = -(it + 1);
    else if (stop) flag[0] = it + 1;
  }
}

static void enqueue_cg_iteration(feahip_ctx *c, int it)
{
  const int gs = spmv_grid(c), gv = vec_grid(c);
  hipLaunchKernelGGL(k_spmv, dim3(gs), dim3(256), 0, c->stream, c->chunk0, c->nchunks_local, c->d_chunk, c->d_rowptr,
                     c->d_colidx, c->d_K, c->d_p, c->d_q, c->d_p, c->d_part, c->d_flag);
  hipLaunchKernelGGL(k_cg_update, dim3(gv), dim3(256), 0, c->stream, c->N, it, gs, c->d_p, c->d_q,
                     c->d_minv, c->d_u, c->d_r, c->d_part, c->d_scal, c->d_flag);
  hipLaunchKernelGGL(k_cg_direction, dim3(gv), dim3(256), 0, c->stream, c->N, it, gv, c->d_r,
                     c->d_minv, c->d_p, c->d_part, c->d_scal, c->d_flag);
}

// Solves K u = f by (preconditioned) CG started from u0 = f, the start vector
// the reference hands to sp_matrix_yale_solve_cg (fea_solver.c:251-256).
int solve_pcg(feahip_ctx *c, int type, double tol, int max_iter, int *iters, double *resid)
{
  const int gv = vec_grid(c);
  const int mode = (type == FEAHIP_CG) ? 0 : 1;
  if (type == FEAHIP_CHOLESKY) { tol = 1e-16; if (max_iter < 100000) max_iter = 100000; }
  hipLaunchKernelGGL(k_precond_build, dim3((c->N + 255) / 256), dim3(256), 0, c->stream, c->N,
                     c->d_rowptr, c->d_colidx, c->d_K, mode, c->d_minv);
  FEA_HIP_CHECK(c, hipMemcpyAsync(c->d_u, c->d_f, sizeof(double) * (size_t)c->ndof,
                                  hipMemcpyDeviceToDevice, c->stream));
  int rc = launch_spmv(c, c->d_u, c->d_q);
  if (rc) return rc;
  hipLaunchKernelGGL(k_cg_init, dim3(gv), dim3(256), 0, c->stream, c->N, c->d_f, c->d_q, c->d_minv,
                     c->d_r, c->d_p, c->d_part);
  hipLaunchKernelGGL(k_cg_init_scalars, dim3(1), dim3(256), 0, c->stream, gv, c->d_part, c->d_scal,
                     tol, c->d_flag);
  FEA_HIP_CHECK(c, hipGetLastError());

  int flag = 0, it = 0;
  const int batch = 32;
  while (it < max_iter) {
    const int n = (max_iter - it < batch) ? (max_iter - it) : batch;
    for (int k = 0; k < n; ++k) enqueue_cg_iteration(c, it + k);
    it += n;
    FEA_HIP_CHECK(c, hipGetLastError());
    FEA_HIP_CHECK(c, hipMemcpyAsync(&flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    if (flag != 0) break;
  }
  double sc[5];
  FEA_HIP_CHECK(c, hipMemcpy(sc, c->d_scal, sizeof(sc), hipMemcpyDeviceToHost));
  int done_it = it;
  if (flag == -1000000000) done_it = 0;
  else if (flag > 0) done_it = flag;
  else if (flag < 0) done_it = -flag;
  if (iters) *iters = done_it;
  if (resid) *resid = (sc[2] > 0) ? sqrt(sc[3] / sc[2]) : sqrt(sc[3]);
  // leave the flag clear so stand-alone SpMV launches are not skipped
  FEA_HIP_CHECK(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
  if (flag < 0 && flag != -1000000000) {
    c->err = "CG breakdown (NaN or zero curvature) at iteration " + std::to_string(-flag);
    return FEAHIP_ENOTCONVERGED;
  }
  return FEAHIP_OK;
}

int time_pcg_iteration(feahip_ctx *c, int warmup, int iters, double *avg_ms)
{
  // set up a well-defined state: u = f, r = p = f, unit preconditioner blocks
  hipLaunchKernelGGL(k_precond_build, dim3((c->N + 255) / 256), dim3(256), 0, c->stream, c->N,
                     c->d_rowptr, c->d_colidx, c->d_K, 1, c->d_minv);
  FEA_HIP_CHECK(c, hipMemcpyAsync(c->d_u, c->d_f, sizeof(double) * (size_t)c->ndof, hipMemcpyDeviceToDevice, c->stream));
  int rc = launch_spmv(c, c->d_u, c->d_q);
  if (rc) return rc;
  const int gv = vec_grid(c);
  hipLaunchKernelGGL(k_cg_init, dim3(gv), dim3(256), 0, c->stream, c->N, c->d_f, c->d_q, c->d_minv,
                     c->d_r, c->d_p, c->d_part);
  hipLaunchKernelGGL(k_cg_init_scalars, dim3(1), dim3(256), 0, c->stream, gv, c->d_part, c->d_scal,
                     0.0, c->d_flag);
  FEA_HIP_CHECK(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
  hipEvent_t e0, e1;
  FEA_HIP_CHECK(c, hipEventCreate(&e0));
  FEA_HIP_CHECK(c, hipEventCreate(&e1));
  for (int k = 0; k < warmup; ++k) enqueue_cg_iteration(c, k);
  FEA_HIP_CHECK(c, hipEventRecord(e0, c->stream));
  for (int k = 0; k < iters; ++k) enqueue_cg_iteration(c, warmup + k);
  FEA_HIP_CHECK(c, hipEventRecord(e1, c->stream));
  FEA_HIP_CHECK(c, hipEventSynchronize(e1));
  float ms = 0;
  FEA_HIP_CHECK(c, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  FEA_HIP_CHECK(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
  *avg_ms = iters > 0 ? (double)ms / iters : 0.0;
  return FEAHIP_OK;
}
