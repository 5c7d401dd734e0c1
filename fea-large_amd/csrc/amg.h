// amg.h -- aggregation multigrid preconditioner for the PCG of the Newton
// step ("preconditioners beyond Jacobi").  Not in the
// reference (its PCG_ILU / Cholesky live in libspmatrix); block-Jacobi PCG
// needs ~2 700 iterations on the 10M-tet block, and the linear solve is what
// bounds Newton iterations per second.
//
// Hierarchy: plain (unsmoothed) aggregation of the node graph, one 3x3 block
// of coarse unknowns per aggregate (piecewise-constant translations), coarse
// matrices by Galerkin sums of the fine 3x3 blocks.  Everything that depends
// on topology only -- aggregates, coarse patterns, which fine blocks sum into
// which coarse block -- is built once on the host; the numeric part (sums,
// block-diagonal inverses, damping) is redone on the device whenever K
// changed.  V(1,1) cycle with damped block-Jacobi smoothing, a fixed number
// of sweeps on the coarsest level: a fixed symmetric positive definite
// operator, as CG requires.
#pragma once
#include "feahip_internal.h"

struct AmgLevel {
  int N = 0, nnzb = 0, nchunks = 0;
  bool owns_matrix = false;              // level 0 aliases the context's K and pattern
  int *rowptr = nullptr, *colidx = nullptr, *diag = nullptr, *chunk = nullptr;
  double *K = nullptr, *minv = nullptr;
  double omega = 0.6;
  // to the next (coarser) level
  int Nc = 0;
  int *agg = nullptr;                    // [N] aggregate of every node
  int *aptr = nullptr, *anodes = nullptr;      // aggregate -> its nodes
  int *cbptr = nullptr, *cblist = nullptr;     // coarse block -> fine blocks summing into it
  int *cbrow = nullptr;                  // fine block -> its block row (for the level-0 dof mask)
  // work vectors [3N]
  double *r = nullptr, *x = nullptr, *y = nullptr;
};

struct AmgHierarchy {
  std::vector<AmgLevel> lv;
  int coarse_sweeps = 12;
  int gamma = 1;                         // coarse corrections per level below the finest (2 = W-cycle)
  double over = 1.0;                     // over-correction factor of the prolongated correction
  bool numeric_valid = false;
  double *d_z = nullptr;                 // level-0 output of the V-cycle
  double *d_pw = nullptr;                // scratch for the power iteration
  long long bytes = 0;
};

// host topology (amg_setup.cpp)
struct HostAmgLevel {
  int N = 0;
  std::vector<int> rowptr, colidx, diag, chunk;        // this level's block pattern
  std::vector<int> agg, aptr, anodes, cbptr, cblist, cbrow;   // to the next level
  int Nc = 0;
};
bool build_host_amg(const std::vector<int> &rowptr, const std::vector<int> &colidx, std::vector<HostAmgLevel> &out);

// device side (amg.hip)
int amg_create(feahip_ctx *c);
void amg_destroy(feahip_ctx *c);
int solve_pcg_amg(feahip_ctx *c, double tol, int max_iter, int *iters, double *resid);
