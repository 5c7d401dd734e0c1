// kernels_assemble.hip -- element-stiffness / residual assembly on gfx950.
//
// Replaces solver_create_stiffness (fea_solver.c:873-883, :887-1068) and
// solver_create_residual_forces (:863-870, :1072-1114).
//
// Two strategies:
//
//  ROW-OWNER (default).  The reference scatters: for every element it adds
//  (3n)^2 values into the global matrix.  On the GPU that is 144 FP64
//  atomics per linear tetrahedron -- an order of magnitude more atomic
//  traffic than HBM bandwidth allows for.  So the sum is turned around: one
//  wavefront OWNS a chunk of consecutive block rows; its lanes walk the
//  (row node, incident element) pairs of those rows (node->element map built
//  once), rebuild the element state in registers, form the 3x3 blocks of
//  that row only, and sum them into the wave's LDS tile with ds_add_f64.
//  The tile is then streamed to HBM: every CSR value is written exactly
//  once, coalesced, no global atomics, no zeroing pass.  Element state is
//  recomputed once per (element, node) visit -- flops are cheap, bytes are
//  not.
//
//  ATOMIC.  One element per lane, global_atomic_add_f64 into the CSR.  Kept
//  as the variant for meshes whose rows are too long for an LDS tile and as
//  an independent cross-check of the row-owner path.
#include "fem_device.h"

// ------------------------------------------------------------------------
// row-owner
// ------------------------------------------------------------------------
// 4 waves per SIMD (<= 128 VGPRs) for 4-node elements; 10-node elements need the whole file
template <int NPE, bool LINTET, bool DOK, bool DOF>
__global__ __launch_bounds__(64 * FEA_WAVES_PER_WG)
void k_assemble_rowowner(AsmArgs A)
{
  __shared__ double sK[FEA_WAVES_PER_WG][DOK ? FEA_CHUNK_BLOCKS * 9 : 1];
  __shared__ double sF[FEA_WAVES_PER_WG][FEA_CHUNK_ROWS * 3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int chunk = A.chunk0 + blockIdx.x * FEA_WAVES_PER_WG + wave;
  if (chunk >= A.chunk0 + A.nchunks) return;     // whole wave leaves; no block barrier below
  double *tK = sK[wave], *tF = sF[wave];

  const int r0 = A.chunk[chunk], r1 = A.chunk[chunk + 1];
  const int b0 = A.rowptr[r0];
  const int nb = A.rowptr[r1] - b0;
  if (DOK)
    for (int t = lane; t < nb * 9; t += 64) tK[t] = 0.0;
  if (DOF)
    for (int t = lane; t < (r1 - r0) * 3; t += 64) tF[t] = 0.0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

  const int p0 = A.incptr[r0], p1 = A.incptr[r1];
  for (int p = p0 + lane; p < p1; p += 64) {
    const uint32_t w = A.inc[p];
    const int e = (int)(w & 0x0FFFFFFFu), la = (int)(w >> 28);
    int nd[NPE];
    double xe[NPE == 4 ? 4 : 1][3], Xe[NPE == 4 ? 4 : 1][3];
    // NPE == 4: the element is presented with its local nodes renumbered
    // k -> k XOR la (an even permutation: orientation and every tensor stay
    // what they are), so the row node is always local node 0 and nothing
    // below needs a per-lane select on doubles.
    // NPE == 10: only the node ids are kept; the coordinates are streamed
    // from L1 inside every Gauss point's sums (gp_state_stream).
    if constexpr (NPE == 4) load_element<4>(A, e, nd, xe, Xe, la);
    else {
#pragma unroll
      for (int k = 0; k < NPE; ++k) nd[k] = A.conn[(size_t)e * NPE + k];
    }
    int a = nd[0];
    if constexpr (NPE != 4) {
#pragma unroll
      for (int k = 1; k < NPE; ++k) a = (la == k) ? nd[k] : a;
    }
    int slot[NPE];
    if (DOK) {
      if constexpr (NPE == 4) {
        const uchar4 s4 = *reinterpret_cast<const uchar4 *>(A.incslot + (size_t)p * 4);
        slot[0] = s4.x; slot[1] = s4.y; slot[2] = s4.z; slot[3] = s4.w;
      } else {
#pragma unroll
        for (int k = 0; k < NPE; ++k) slot[k] = A.incslot[(size_t)p * NPE + k];
      }
    }
    const int rowoff = (A.rowptr[a] - b0) * 9;
    double fa[3] = {0, 0, 0};
    for (int gp = 0; gp < A.G; ++gp) {
      GPState<NPE> s;
      if constexpr (NPE == 4) gp_state<4, LINTET, false>(xe, Xe, A.tab, gp, A.model, A.lambda, A.mu, s);
      else gp_state_stream<NPE>(A.x, A.X0, nd, A.tab, gp, A.model, A.lambda, A.mu, s);
      if (!(s.detJ > 0.0) && DOK && la == 0) atomicAdd(A.bad, 1);
      if (s.detJ == 0.0) continue;      // reference keeps no gradient then (fea_solver.c:697)
      double ga[3] = {s.g[0][0], s.g[0][1], s.g[0][2]};
      if constexpr (NPE != 4) {
#pragma unroll
        for (int k = 1; k < NPE; ++k)
#pragma unroll
          for (int i = 0; i < 3; ++i) ga[i] = (la == k) ? s.g[k][i] : ga[i];
      }
      if (DOF) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
          fa[i] -= s.vol * (s.sig[i][0] * ga[0] + s.sig[i][1] * ga[1] + s.sig[i][2] * ga[2]);
      }
      if (DOK) {
        // the diagonal block (b == la) is skipped: it comes from the row sum below
        if constexpr (NPE == 4) {
          // three off-diagonal blocks: local columns 1..3 of the renumbered element
#pragma unroll
          for (int k = 1; k < 4; ++k) {
            double h[3], m[3], t[3], blk[9];
            col_vectors(s.g[k], s.sig, s.l1, s.m1, s.vol, h, m, t);
            block_ab(ga, h, m, t, blk);
            double *dst = tK + rowoff + slot[k] * 9;
#pragma unroll
            for (int q = 0; q < 9; ++q)
              __hip_atomic_fetch_add(dst + q, blk[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        } else {
#pragma unroll
          for (int b = 0; b < NPE; ++b) {
            if (b == la) continue;
            double h[3], m[3], t[3], blk[9];
            col_vectors(s.g[b], s.sig, s.l1, s.m1, s.vol, h, m, t);
            block_ab(ga, h, m, t, blk);
            double *dst = tK + rowoff + slot[b] * 9;
#pragma unroll
            for (int q = 0; q < 9; ++q)
              __hip_atomic_fetch_add(dst + q, blk[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
    }
    if (DOF) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
        __hip_atomic_fetch_add(tF + (a - r0) * 3 + i, fa[i], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  if (DOK) {
    // Shape functions sum to one, so sum_b grad N_b = 0 and every element's
    // blocks of one row add up to zero: K_aa = -sum_{b != a} K_ab.  The
    // diagonal block -- the one all visits of a row would collide on -- is
    // therefore never added to; it is the negative sum of the finished row.
    for (int t = lane; t < (r1 - r0) * 9; t += 64) {
      const int r = r0 + t / 9, q = t % 9;
      const int kb = A.rowptr[r] - b0, ke = A.rowptr[r + 1] - b0, kd = A.diag[r] - b0;
      double acc = 0;
      for (int k = kb; k < ke; ++k) acc += (k == kd) ? 0.0 : tK[k * 9 + q];
      tK[kd * 9 + q] = -acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    double *Kd = A.K + (size_t)b0 * 9;
    for (int t = lane; t < nb * 9; t += 64) Kd[t] = tK[t];
  }
  if (DOF) {
    double *fd = A.f + (size_t)r0 * 3;
    for (int t = lane; t < (r1 - r0) * 3; t += 64) fd[t] = tF[t];
  }
}

// ------------------------------------------------------------------------
// atomic (element-parallel)
// ------------------------------------------------------------------------
__device__ __forceinline__ int find_block(const AsmArgs &A, int row, int col)
{
  int lo = A.rowptr[row], hi = A.rowptr[row + 1] - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (A.colidx[mid] < col) lo = mid + 1; else hi = mid;
  }
  return lo;
}

template <int NPE, bool LINTET, bool DOK, bool DOF>
__global__ __launch_bounds__(256)
void k_assemble_atomic(AsmArgs A)
{
  // per-lane copy of the gradients so the (a,b) loops can stay rolled
  __shared__ double sg[NPE * 3][256];
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= A.E) return;
  const int tid = threadIdx.x;
  int nd[NPE];
  double xe[NPE][3], Xe[NPE][3];
  load_element<NPE>(A, e, nd, xe, Xe);
  for (int gp = 0; gp < A.G; ++gp) {
    GPState<NPE> s;
    gp_state<NPE, LINTET>(xe, Xe, A.tab, gp, A.model, A.lambda, A.mu, s);
    if (!(s.detJ > 0.0) && DOK) atomicAdd(A.bad, 1);
    if (s.detJ == 0.0) continue;
#pragma unroll
    for (int k = 0; k < NPE; ++k)
#pragma unroll
      for (int i = 0; i < 3; ++i) sg[k * 3 + i][tid] = s.g[k][i];
    if (DOF) {
#pragma unroll
      for (int a = 0; a < NPE; ++a)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double v = -s.vol * (s.sig[i][0] * s.g[a][0] + s.sig[i][1] * s.g[a][1] +
                                     s.sig[i][2] * s.g[a][2]);
          atomicAdd(A.f + (size_t)nd[a] * 3 + i, v);
        }
    }
    if (DOK) {
      for (int b = 0; b < NPE; ++b) {
        const double gb[3] = {sg[b * 3][tid], sg[b * 3 + 1][tid], sg[b * 3 + 2][tid]};
        double h[3], m[3], t[3];
        col_vectors(gb, s.sig, s.l1, s.m1, s.vol, h, m, t);
        int nb = nd[0];
#pragma unroll
        for (int k = 1; k < NPE; ++k) nb = (b == k) ? nd[k] : nb;
        for (int a = 0; a < NPE; ++a) {
          const double ga[3] = {sg[a * 3][tid], sg[a * 3 + 1][tid], sg[a * 3 + 2][tid]};
          int na = nd[0];
#pragma unroll
          for (int k = 1; k < NPE; ++k) na = (a == k) ? nd[k] : na;
          if (na < A.row0 || na >= A.row1) continue;      // another rank's row: not stored here
          double blk[9];
          block_ab(ga, h, m, t, blk);
          double *dst = A.K + (size_t)find_block(A, na, nb) * 9;
#pragma unroll
          for (int q = 0; q < 9; ++q) atomicAdd(dst + q, blk[q]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------
// per-Gauss-point F and sigma in the reference's shapes
// (graddefs[e][g], stresses[e][g]: fea_solver.h:262-269)
// ------------------------------------------------------------------------
template <int NPE, bool LINTET>
__global__ __launch_bounds__(256)
void k_state_export(AsmArgs A)
{
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= A.E) return;
  int nd[NPE];
  double xe[NPE][3], Xe[NPE][3];
  load_element<NPE>(A, e, nd, xe, Xe);
  for (int gp = 0; gp < A.G; ++gp) {
    GPState<NPE> s;
    gp_state<NPE, LINTET>(xe, Xe, A.tab, gp, A.model, A.lambda, A.mu, s);
    double *Fo = A.Fout + ((size_t)e * A.G + gp) * 9;
    double *So = A.Sout + ((size_t)e * A.G + gp) * 9;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        Fo[3 * i + j] = s.F[i][j];
        So[3 * i + j] = s.sig[i][j];
      }
    if (A.Gout) {                 // shape_gradients[e][g]: grads[i][a] = dN_a/dx_i and detJ (fea_solver.h:200-205)
      double *Go = A.Gout + ((size_t)e * A.G + gp) * 3 * NPE;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int a = 0; a < NPE; ++a) Go[i * NPE + a] = s.g[a][i];
      A.Dout[(size_t)e * A.G + gp] = s.detJ;
    }
  }
}

// ------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------
static AsmArgs make_args(feahip_ctx *c)
{
  AsmArgs A;
  A.N = c->N; A.E = c->E; A.G = c->G; A.nchunks = c->nchunks_local; A.chunk0 = c->chunk0; A.model = c->model;
  A.row0 = c->row0; A.row1 = c->row1;
  A.lambda = c->lambda; A.mu = c->mu;
  A.tab = c->d_table; A.conn = c->d_conn; A.X0 = c->d_X0; A.x = c->d_x;
  A.rowptr = c->d_rowptr; A.colidx = c->d_colidx; A.K = c->d_K; A.f = c->d_f;
  A.incptr = c->d_incptr; A.inc = c->d_inc; A.incslot = c->d_incslot;
  A.chunk = c->d_chunk; A.diag = c->d_diag; A.bad = c->d_flag + 1;
  A.Fout = c->d_F; A.Sout = c->d_S; A.Gout = nullptr; A.Dout = nullptr;
  return A;
}

template <int NPE, bool LINTET>
static void launch_rowowner_t(feahip_ctx *c, const AsmArgs &A, bool doK, bool doF)
{
  const int grid = (c->nchunks_local + FEA_WAVES_PER_WG - 1) / FEA_WAVES_PER_WG;
  const dim3 blk(64 * FEA_WAVES_PER_WG);
  if (doK && doF) hipLaunchKernelGGL((k_assemble_rowowner<NPE, LINTET, true, true>), dim3(grid), blk, 0, c->stream, A);
  else if (doK)   hipLaunchKernelGGL((k_assemble_rowowner<NPE, LINTET, true, false>), dim3(grid), blk, 0, c->stream, A);
  else            hipLaunchKernelGGL((k_assemble_rowowner<NPE, LINTET, false, true>), dim3(grid), blk, 0, c->stream, A);
}

template <int NPE, bool LINTET>
static void launch_atomic_t(feahip_ctx *c, const AsmArgs &A, bool doK, bool doF)
{
  const int grid = (c->E + 255) / 256;
  if (doK && doF) hipLaunchKernelGGL((k_assemble_atomic<NPE, LINTET, true, true>), dim3(grid), dim3(256), 0, c->stream, A);
  else if (doK)   hipLaunchKernelGGL((k_assemble_atomic<NPE, LINTET, true, false>), dim3(grid), dim3(256), 0, c->stream, A);
  else            hipLaunchKernelGGL((k_assemble_atomic<NPE, LINTET, false, true>), dim3(grid), dim3(256), 0, c->stream, A);
}

int launch_assemble(feahip_ctx *c, bool doK, bool doF)
{
  { const int rc = ensure_k(c); if (rc) return rc; }
  AsmArgs A = make_args(c);
  const bool rowowner_ok = c->incslot_ok && c->max_rowlen <= FEA_CHUNK_BLOCKS;
  int strat = c->strategy;
  // AUTO: the gather kernels where their maps build and their chunks are compact, the staged visits / the
  // shared-state kernel behind them, the generic row-owner visits or the atomic scatter behind those
  if (strat == FEAHIP_ASM_AUTO) {
    if (c->linear_tet && c->G == 1 && c->h_pat) {
      // GATHER where the chunks of consecutive rows are compact enough that an element is evaluated at most ~2.5
      // times (locality numberings: bricks, space-filling curves); the staged visits otherwise (measured on the
      // 10M-tet block: 0.97 ms against 1.00 with bricks of 4x2x2 nodes, 1.24 against 1.00 with lexicographic ids)
      const bool declined = c->gather_declined_row0 == c->row0 && c->gather_declined_row1 == c->row1;
      if (!declined) { const int rc = ensure_gather(c); if (rc) return rc; }
      if (!declined && c->have_gather && c->gather_evals_per_element <= 2.5) strat = FEAHIP_ASM_GATHER;
      else {
        if (!declined && c->have_gather) {
          // the maps were built to learn what the chunks cost; AUTO does not run them: they do not stay resident
          // (84 B per element), and the sizes reported are those of the kernel that runs
          (void)hipFree(c->d_gmaps); c->d_gmaps = nullptr;
          c->have_gather = false; c->ngchunks = 0; c->gather_row0 = c->gather_row1 = -1; c->gather_bytes = 0;
          c->gather_declined_row0 = c->row0; c->gather_declined_row1 = c->row1;
        }
        { const int rc = ensure_visits(c); if (rc) return rc; }
        if (c->have_visits) strat = FEAHIP_ASM_STAGED;
      }
    }
    if (strat == FEAHIP_ASM_AUTO && (c->npe == 10 || c->npe == 8)) {
      // 10-node tetrahedra, 8-node bricks: gather chunks of up to 64 rows where the numbering keeps them compact (an element's
      // records are expanded in ~3 chunks with a brick numbering; lexicographic ids on the 497 664-element block: ~7
      // chunks, 1.42 ms against the 2.27 of the shared-state kernel, whose 3-row chunks evaluate an element 8 times)
      { const int rc = ensure_gather10(c); if (rc) return rc; }
      if (c->have_gather && c->gather_evals_per_element <= 12.0) strat = FEAHIP_ASM_GATHER;
    }
    if (strat == FEAHIP_ASM_AUTO && c->npe == 10) { const int rc = ensure_quad(c); if (rc) return rc; }
    if (strat == FEAHIP_ASM_AUTO)
      strat = (c->have_quad && doK) ? FEAHIP_ASM_SHARED : (rowowner_ok ? FEAHIP_ASM_ROWOWNER : FEAHIP_ASM_ATOMIC);
  }
  if (strat == FEAHIP_ASM_SHARED) { const int rc = ensure_quad(c); if (rc) return rc; }
  if (strat == FEAHIP_ASM_SHARED && !doK && c->have_quad) strat = rowowner_ok ? FEAHIP_ASM_ROWOWNER : FEAHIP_ASM_ATOMIC;   // residual alone: visit kernel
  c->last_strategy = strat;                          // the kernel that runs, after the residual-only fallbacks
  if (strat == FEAHIP_ASM_SHARED) {
    if (!c->have_quad) {
      c->err = "shared-state assembly needs 10-node elements whose chunks fit the LDS tiles";
      return FEAHIP_EINVAL;
    }
    FEA_HIP_CHECK(c, hipMemsetAsync(c->d_flag + 1, 0, sizeof(int), c->stream));
    return launch_assemble_quad(c, doF);
  }
  if (strat == FEAHIP_ASM_GATHER && (c->npe == 10 || c->npe == 8)) {
    { const int rc = ensure_gather10(c); if (rc) return rc; }
    if (!c->have_gather) {
      c->err = "gather assembly of 10-node tetrahedra / 8-node bricks needs rows that fit the LDS tiles";
      return FEAHIP_EINVAL;
    }
    if (doK) FEA_HIP_CHECK(c, hipMemsetAsync(c->d_flag + 1, 0, sizeof(int), c->stream));
    return launch_assemble_gather10(c, doK, doF);
  }
  if (strat == FEAHIP_ASM_GATHER) {
    { const int rc = ensure_gather(c); if (rc) return rc; }
    if (!c->have_gather) {
      c->err = "gather assembly needs linear tetrahedra (one Gauss point) whose rows fit the LDS tiles";
      return FEAHIP_EINVAL;
    }
    if (doK) FEA_HIP_CHECK(c, hipMemsetAsync(c->d_flag + 1, 0, sizeof(int), c->stream));
    return launch_assemble_gather(c, doK, doF);
  }
  if (strat == FEAHIP_ASM_PAIRED || strat == FEAHIP_ASM_PATCH || strat == FEAHIP_ASM_PIPELINED) {
    c->err = "this assembly strategy was retired (measured slower than the staged visits and the gather kernel: DESIGN.md)";
    return FEAHIP_EINVAL;
  }
  if (strat == FEAHIP_ASM_STAGED) {
    { const int rc = ensure_visits(c); if (rc) return rc; }
    if (!c->have_visits) {
      c->err = "staged assembly needs linear tetrahedra whose chunks fit the LDS tiles";
      return FEAHIP_EINVAL;
    }
    if (doK) FEA_HIP_CHECK(c, hipMemsetAsync(c->d_flag + 1, 0, sizeof(int), c->stream));
    return launch_assemble_visit(c, doK, doF);
  }
  { const int rc = ensure_generic_maps(c); if (rc) return rc; }     // the generic kernels walk the incidence lists
  A.incptr = c->d_incptr; A.inc = c->d_inc; A.incslot = c->d_incslot;
  if (strat == FEAHIP_ASM_ROWOWNER && !rowowner_ok) {
    c->err = "row-owner assembly needs block rows of at most " +
             std::to_string(FEA_CHUNK_BLOCKS) + " blocks (mesh has " +
             std::to_string(c->max_rowlen) + ")";
    return FEAHIP_EINVAL;
  }
  if (doK) FEA_HIP_CHECK(c, hipMemsetAsync(c->d_flag + 1, 0, sizeof(int), c->stream));
  if (strat == FEAHIP_ASM_ATOMIC) {
    if (doK) FEA_HIP_CHECK(c, hipMemsetAsync(c->d_K_base, 0, sizeof(double) * 9 * (size_t)(c->kb1 - c->kb0), c->stream));
    if (doF) FEA_HIP_CHECK(c, hipMemsetAsync(c->d_f, 0, sizeof(double) * (size_t)c->ndof, c->stream));
    if (c->npe == 4) { if (c->linear_tet) launch_atomic_t<4, true>(c, A, doK, doF); else launch_atomic_t<4, false>(c, A, doK, doF); }
    else if (c->npe == 8) launch_atomic_t<8, false>(c, A, doK, doF);
    else launch_atomic_t<10, false>(c, A, doK, doF);
  } else {
    if (c->npe == 4) { if (c->linear_tet) launch_rowowner_t<4, true>(c, A, doK, doF); else launch_rowowner_t<4, false>(c, A, doK, doF); }
    else if (c->npe == 8) launch_rowowner_t<8, false>(c, A, doK, doF);
    else launch_rowowner_t<10, false>(c, A, doK, doF);
  }
  FEA_HIP_CHECK(c, hipGetLastError());
  return FEAHIP_OK;
}

int launch_state_export(feahip_ctx *c, double *d_grads, double *d_detj)
{
  AsmArgs A = make_args(c);
  A.Gout = d_grads; A.Dout = d_detj;
  const int grid = (c->E + 255) / 256;
  if (c->npe == 4) {
    if (c->linear_tet) hipLaunchKernelGGL((k_state_export<4, true>), dim3(grid), dim3(256), 0, c->stream, A);
    else hipLaunchKernelGGL((k_state_export<4, false>), dim3(grid), dim3(256), 0, c->stream, A);
  } else if (c->npe == 8)
    hipLaunchKernelGGL((k_state_export<8, false>), dim3(grid), dim3(256), 0, c->stream, A);
  else
    hipLaunchKernelGGL((k_state_export<10, false>), dim3(grid), dim3(256), 0, c->stream, A);
  FEA_HIP_CHECK(c, hipGetLastError());
  return FEAHIP_OK;
}
