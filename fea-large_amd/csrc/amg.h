// amg.h -- aggregation multigrid preconditioner for the PCG of the Newton
// step ("preconditioners beyond Jacobi").  Not in the
// reference (its PCG_ILU / Cholesky live in libspmatrix); block-Jacobi PCG
// needs ~2 700 iterations on the 10M-tet block, and the linear solve is what
// bounds Newton iterations per second.
//
// Hierarchy: plain (unsmoothed) aggregation of the node graph with the six
// rigid-body modes of every aggregate as coarse unknowns: its translation and
// its rotation about the aggregate's centroid, stored as two 3x3 block rows so
// that every level is again a 3x3 block-CSR matrix for the same SpMV and
// smoother kernels.  (Translations alone leave the bending modes of a slender
// block to the smoother: 335 iterations on the 1M-tet block against 1266 for
// block-Jacobi.)  Coarse matrices are Galerkin products P' K P, with the 3x3
// blocks of P either I, 0 or the cross-product matrix of the offset from the
// centroid.  Everything that depends on topology and geometry only --
// aggregates, centroids, coarse patterns, which fine blocks enter which coarse
// block -- is built once on the host; the numeric part (Galerkin products,
// block-diagonal inverses, damping) is redone on the device whenever K
// changed.  W-cycle (two coarse corrections per level, over-corrected by 2)
// with damped block-Jacobi smoothing before and after, a fixed number of
// sweeps on the coarsest level: a fixed symmetric positive definite operator,
// as CG requires.  A rank of a sharded solve builds the hierarchy of its own
// diagonal block (amg_setup.cpp), so the preconditioner needs no communication.
#pragma once
#include "feahip_internal.h"

struct AmgLevel {
  int N = 0, nnzb = 0, nchunks = 0;
  bool owns_matrix = false;              // level 0 aliases the context's K and pattern
  int *rowptr = nullptr, *colidx = nullptr, *diag = nullptr, *chunk = nullptr;
  double *K = nullptr, *minv = nullptr;
  float *K32 = nullptr;                  // coarse levels may store their matrix in single precision instead (K null)
  unsigned short *K16 = nullptr;         // level 0: bfloat16 copy of the context's K for the smoother's products
  double omega = 0.6;
  uint8_t *type = nullptr;               // [N] 0 = translation row, 1 = rotation row; null on level 0 (all 0)
  // to the next (coarser) level
  int Nc = 0;                            // its block rows (2 per aggregate); 0 on the coarsest level
  int *agg = nullptr;                    // [N] aggregate of every block row
  double *doff = nullptr;                // [N][3] position minus centroid of its aggregate
  int *aptr = nullptr, *anodes = nullptr;      // aggregate -> its block rows
  int *cbptr = nullptr, *cblist = nullptr;     // aggregate pair -> fine blocks entering its 4 coarse blocks
  int *prow = nullptr;                   // aggregate pair -> its row aggregate
  int *cbrow = nullptr;                  // fine block -> its block row
  // work vectors [3N]
  double *r = nullptr, *x = nullptr, *y = nullptr;
};

struct AmgHierarchy {
  std::vector<AmgLevel> lv;
  int coarse_sweeps = 2;                 // damped Jacobi sweeps on the coarsest level (amg_setup.cpp: why so few)
  int tail_from = -1;                    // first level of the subtree the one-workgroup kernel runs (amg.hip: k_amg_tail); -1: none
  int gamma_from = 0;                    // first level whose coarse correction is repeated `gamma` times
  int gamma_until = 1 << 20;             // ... and the first level below them that runs a single correction again (FEAHIP_AMG_GAMMA_UNTIL)
  int gamma = 2;                         // coarse corrections per level below the finest (2 = W-cycle)
  double over = 2.0;                     // over-correction of the prolongated correction (<= 2 keeps the cycle SPD);
                                         // 10M-tet block, PCG to 1e-14: V-cycle 274 iterations (over 1.5), W-cycle below the
                                         // finest level 150, W-cycle on every level 94 (339 ms against 1 704 ms block-Jacobi)
  bool numeric_valid = false;
  int fine_bits = 16;                    // the smoother of level 0 multiplies with a copy of K in bfloat16 (FEAHIP_AMG_FINE_BITS=32:
                                         // float, 64: K itself): 95 iterations each way on the 10M-tet block (float 16 in between
                                         // 88); per iteration 2.7 ms with the float copy, 3.2 with K, +2 bytes per value
  bool coarse_f32 = true;                // coarse matrices stored in single precision (FEAHIP_AMG_F32=0: double): the
                                         // preconditioner stays a fixed linear operator, vectors and arithmetic are double;
                                         // same 94 iterations on the 10M-tet block, 7 % less time, half the memory
  unsigned long long num_epoch = 0; bool num_bc = false;     // the matrix the numeric part was built for
  int row0 = 0, row1 = 0;                // rows of level 0 this hierarchy covers (the rank's own)
  double *d_z = nullptr;                 // level-0 iterate of the cycle (zero outside the rank's rows)
  double *result = nullptr;              // where the last cycle left z: d_z, or the context's q when the last sweep is fused
  // the tail's entry-level matrix re-laid out for its product (k_tail_relayout, amg.hip): groups of 16 rows, per group
  // `slots` blocks per lane, every (slot, 16-byte piece) 64 lanes wide; tail_goff[g] = first slot of group g
  float *d_tail_ell = nullptr; int *d_tail_goff = nullptr; int tail_groups = 0, tail_slots = 0;
  bool tail_ell = true;                  // (FEAHIP_AMG_TAIL_ELL=0: the product gathers 36-byte blocks from L2)
  double *d_tail_cop = nullptr; int tail_cop_n = 0;   // the coarsest level's smoothing procedure as one dense operator (k_tail_coarse_op), n x n, transposed
  bool tail_cop = true;                  // (FEAHIP_AMG_TAIL_COP=0: the sweeps one after the other)
  double *d_tail_blob = nullptr;         // the tail levels' read-only arrays in the tail kernel's LDS layout, repacked at every numeric setup
  bool tail_blob = true;                 // (FEAHIP_AMG_TAIL_BLOB=0: every launch gathers them array by array)
  bool fused_post = false;               // post-smoothing product and update in one launch (FEAHIP_AMG_FUSED_POST=1; measured 1.6-2.5 % slower per CG iteration than the two launches)
  double *d_pw = nullptr;                // scratch for the power iteration
  double *d_lam = nullptr;               // [64] squared norms of the levels' power iterations (read once per numeric setup)
  long long bytes = 0;
};

// host topology (amg_setup.cpp)
struct HostAmgLevel {
  int N = 0, S = 0;                                    // block rows, sites (N on level 0, N/2 below)
  std::vector<int> rowptr, colidx, diag, chunk;        // this level's block pattern
  std::vector<uint8_t> type;                           // empty on level 0
  std::vector<double> pos;                             // [S][3] site positions
  int Sc = 0;                                          // aggregates (sites of the next level); 0 = coarsest
  std::vector<int> agg, aptr, anodes, cbptr, cblist, cbrow, prow;    // to the next level
  std::vector<double> doff;
};
bool build_host_amg(const std::vector<int> &rowptr, const std::vector<int> &colidx, const std::vector<double> &pos,
                    int own0, int own1, std::vector<HostAmgLevel> &out);

// device side (amg.hip)
int amg_create(feahip_ctx *c);
void amg_destroy(feahip_ctx *c);
int amg_prepare(feahip_ctx *c);                      // hierarchy for the current row range, numeric part for the current K
double *amg_apply(feahip_ctx *c, const double *r);    // z = M^-1 r on the rank's rows; returns z
double *amg_result(feahip_ctx *c);                    // the z of the last amg_apply
