/*
 * fea_hip.h -- C ABI of the MI355X-native assembly + Newton hot path.
 *
 * One opaque context replaces the `fea_solver` object of the reference for
 * the calls that `solve()` makes inside its load-increment / Newton loops
 * (solver-large/fea_solver.c:130-242).  Every entry below names the
 * reference function it stands in for.  Plain pointers and sizes only; all
 * host arrays are copied (caller keeps ownership); getters fill caller-owned
 * host buffers in the reference's own shapes.
 *
 * Every function returns 0 on success or a negative FEAHIP_E* code, never
 * exits and never asserts (the reference's error() calls exit(),
 * fea_solver.c:57-61).  feahip_last_error() gives the message.
 * A context is single-caller (the reference is single-threaded).
 */
#ifndef FEA_HIP_H
#define FEA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct feahip_ctx feahip_ctx;

enum {
  FEAHIP_OK = 0,
  FEAHIP_EINVAL = -1,      /* bad argument / unsupported combination        */
  FEAHIP_ENODEVICE = -2,   /* no usable HIP device: the path has no CPU mode */
  FEAHIP_EHIP = -3,        /* a HIP runtime call failed                      */
  FEAHIP_ENOMEM = -4,
  FEAHIP_ESTATE = -5,      /* call order violated (e.g. restore before stash)*/
  FEAHIP_ENOTCONVERGED = -6,
  FEAHIP_ECOMM = -7        /* RCCL failure                                   */
};

/* material models, numbered as `model_type` (fea_model.h:37-40) */
enum { FEAHIP_MODEL_A5 = 0, FEAHIP_MODEL_COMPRESSIBLE_NEOHOOKEAN = 1 };
/* linear solvers, numbered as `slae_solver_type` (fea_solver.h:62-66).
 * CG = conjugate gradients started from x0 = rhs (fea_solver.c:251-256);
 * PCG_ILU and CHOLESKY have no GPU-idiomatic twin: both run diagonally
 * (3x3 block) preconditioned CG, CHOLESKY iterated to stagnation.           */
enum { FEAHIP_CG = 0, FEAHIP_PCG_ILU = 1, FEAHIP_CHOLESKY = 2 };
/* assembly strategies */
enum {
  FEAHIP_ASM_AUTO = 0,
  FEAHIP_ASM_ROWOWNER = 1, /* node-centric gather, every CSR value written
                              once, deterministic                            */
  FEAHIP_ASM_ATOMIC = 2,   /* element-parallel, FP64 atomics into the CSR    */
  FEAHIP_ASM_PATCH = 3,    /* retired (refused): element states shared through
                              LDS per 16-row chunk; 1.8x slower than STAGED   */
  FEAHIP_ASM_STAGED = 4,   /* linear tets: row-owner visits with node
                              coordinates and connectivity staged in LDS     */
  FEAHIP_ASM_PAIRED = 5,   /* retired (refused): two face-sharing elements per
                              lane; 5 % slower than STAGED                    */
  FEAHIP_ASM_PIPELINED = 6,/* retired (refused): STAGED with the next chunk's
                              loads in flight; equal to STAGED within 2 %     */
  FEAHIP_ASM_SHARED = 7,   /* 10-node tets: Gauss-point states evaluated once
                              per chunk element and shared through LDS, the
                              blocks of a (row, element, column) pair summed
                              over the Gauss points in registers              */
  FEAHIP_ASM_GATHER = 8    /* linear tets: a 1024-thread workgroup owns a run
                              of block rows; every element touching them is
                              evaluated once into an LDS record, one thread
                              per off-diagonal block then sums that block's
                              element contributions in registers.  10-node
                              tets: the same with the Gauss points as an
                              outer loop over per-element state records a
                              first kernel wrote, up to five blocks per
                              thread.  No atomics anywhere, fixed summation
                              order: bitwise reproducible                     */
};

/* ---- lifetime ----------------------------------------------------------- */

/* Stands in for fea_solver_alloc (fea_solver.c:387-456) plus
 * solver_create_element_database (:556-571): the element plug-in arrives as
 * the tables solver_gauss_node_alloc tabulates (:503-535) -- weights[G]
 * (divisor 6 already inside) and dforms[G][3][npe] -- because device code
 * cannot call the host's isoform_t/disoform_t pointers (fea_solver.h:34-40).
 * The material plug-in arrives as (model, parameters[]) of `fea_model`
 * (fea_model.h:46-57): parameters[0] = lambda, parameters[1] = mu.
 * elements: [n_elems][npe] 0-based node ids (elements_array, :132-138)
 * nodes0:   [n_nodes][3]   initial coordinates (nodes_array, :124-128)
 * presc_*:  prescribed_bnd_node fields (:142-146), deck order.              */
int feahip_create(feahip_ctx **out, int device,
                  int n_nodes, int n_elems, int npe, int gauss_count,
                  const double *gauss_weights, const double *dforms,
                  const int *elements, const double *nodes0,
                  int model, const double *model_params, int params_count,
                  int n_presc, const int *presc_node, const int *presc_type,
                  const double *presc_values);

/* fea_solver_free (fea_solver.c:459-501); frees device memory only. */
void feahip_destroy(feahip_ctx *ctx);

const char *feahip_last_error(const feahip_ctx *ctx);
/* message of the last failed feahip_create (no context exists then) */
const char *feahip_create_error(void);

/* ---- the calls of solve() ---------------------------------------------- */

/* solver_update_nodes_with_bc (fea_solver.c:1281-1284): x[c] += lambda*value
 * for every prescribed dof.                                                 */
int feahip_update_nodes_with_bc(feahip_ctx *ctx, double lambda);

/* solver_create_current_shape_gradients + solver_create_stresses
 * (fea_solver.c:831-861).  The assembly kernels recompute J, grad N, F and
 * sigma in registers, so this only invalidates the cached per-Gauss-point
 * F / sigma that feahip_get_graddefs / feahip_get_stresses serve; it also
 * returns, in *n_bad (may be NULL), the count of Gauss points whose current
 * Jacobian determinant is <= 0 as seen by the last assembly.               */
int feahip_update_state(feahip_ctx *ctx, int *n_bad);

/* solver_create_stiffness (fea_solver.c:873-883): K = sum_e (Kc + Ksigma). */
int feahip_create_stiffness(feahip_ctx *ctx);
/* solver_create_residual_forces (fea_solver.c:863-870): f = -T.            */
int feahip_create_residual_forces(feahip_ctx *ctx);
/* both in one pass over the mesh (what a full-Newton iteration needs)      */
int feahip_create_stiffness_and_residual(feahip_ctx *ctx);

/* sp_matrix_copy(global -> stiffness) (fea_solver.c:179) and
 * sp_matrix_free + sp_matrix_copy(stiffness -> global) (:194-195).         */
int feahip_stash_stiffness(feahip_ctx *ctx);
int feahip_restore_stiffness(feahip_ctx *ctx);

/* solver_apply_prescribed_bc (fea_solver.c:1200-1257): for every prescribed
 * dof c with p = lambda*value: f[r] -= K[r,c]*p, row and column c zeroed
 * keeping K[c,c], f[c] = K[c,c]*p.                                          */
int feahip_apply_prescribed_bc(feahip_ctx *ctx, double lambda);

/* solver_solve_slae (fea_solver.c:300-321): solve K u = f.  iters / resid
 * (relative residual ||f-Ku||/||f||) may be NULL.                          */
int feahip_solve_slae(feahip_ctx *ctx, int solver_type, double tolerance,
                      int max_iterations, int *iters, double *resid);

/* cdot(global_forces_vct, global_solution_vct) (fea_solver.c:208-210).     */
int feahip_energy(feahip_ctx *ctx, double *tolerance);

/* solver_update_nodes_with_solution (fea_solver.c:1270-1279): x += u.
 * u == NULL uses the device-resident solution of the last solve.           */
int feahip_update_nodes_with_solution(feahip_ctx *ctx, const double *u);

/* The whole loop of solve() (fea_solver.c:163-236) run by the library.
 * tol_log[tol_log_cap] receives <u,f> of every Newton iteration, its_log
 * [load_increments] the iteration count of every step (either may be NULL).
 * *steps_done = number of completed load steps.                             */
int feahip_solve(feahip_ctx *ctx, int load_increments, int max_newton,
                 int modified_newton, double desired_tolerance,
                 int solver_type, double solver_tolerance, int solver_max_iter,
                 double *tol_log, int tol_log_cap, int *its_log,
                 int *steps_done);

/* ---- multi-GPU: row-sharded operation ---------------------------------- */
/* The reference is one process; sharding is new.  Nodes (block rows of K,
 * entries of f, u, x) are owned by one rank each, in contiguous ranges; a rank
 * assembles every element touching its rows, so assembly needs no exchange.
 * The linear solve exchanges halo rows (ncclSend/ncclRecv between slab
 * neighbours) and all-reduces one to three doubles per CG step; the Newton
 * test <u,f> is all-reduced so every rank takes the same branch.
 *
 * One process per GPU: rank 0 calls feahip_comm_unique_id (128 bytes),
 * broadcasts it by any means (bench.py: torch.distributed), every rank calls
 * feahip_comm_init; after that the ordinary entries above (create_stiffness,
 * apply_prescribed_bc, solve_slae, energy, update_nodes_with_solution, solve)
 * are collective over the ranks.  A rank holds the K rows, the modified-Newton
 * copy, the assembly maps and the multigrid hierarchy of its own slab only;
 * the mesh arrays handed to feahip_create are the whole mesh on every rank
 * (DESIGN.md section 6 says what is sharded and what is not).                */
int feahip_comm_unique_id(void *out, int cap);
int feahip_comm_init(feahip_ctx *ctx, int rank, int nranks, const void *unique_id);
/* nodes with LIBRARY id in [row0, row1) are this rank's; getters are
 * authoritative there only (feahip_node_numbering maps the caller's node ids
 * to library ids: a slab of library ids is a slab of the mesh, not a range of
 * the caller's ids)                                                          */
int feahip_owned_rows(feahip_ctx *ctx, int *row0, int *row1);

/* A rank that holds only its slab.  feahip_create_rank takes the same arguments
 * as feahip_create plus (rank, nranks) and builds the context of ONE rank: the
 * nodes it owns (a slab of library ids), the elements that touch them, their
 * halo nodes -- locally indexed, owned nodes first -- with the block rows, K,
 * maps, vectors and multigrid hierarchy of that sub-mesh only, and the halo plan
 * installed.  Nothing in it is sized by the whole mesh (the reference's single
 * row-wise store, fea_solver.c:444-448, becomes nranks independent ones).  Its
 * node- and element-indexed entries speak LOCAL indices: feahip_rank_maps says
 * which nodes / elements of the caller's mesh they are; feahip_rank_counts:
 * out8 = {local nodes, owned nodes (local ids [0, owned)), local elements,
 * nodes of the whole mesh, blocks of all local rows, blocks of the owned rows,
 * rows sent per exchange, rows received}.  After feahip_comm_init (same rank,
 * nranks) or feahip_group_init the collective entries work as for row shards. */
int feahip_create_rank(feahip_ctx **out, int device, int rank, int nranks,
                       int n_nodes, int n_elems, int npe, int gauss_count,
                       const double *gauss_weights, const double *dforms,
                       const int *elements, const double *nodes0,
                       int model, const double *model_params, int params_count,
                       int n_presc, const int *presc_node, const int *presc_type,
                       const double *presc_values);
int feahip_rank_counts(feahip_ctx *ctx, long long *out8);
int feahip_rank_maps(feahip_ctx *ctx, int *node_global, int *elem_global);
/* Host-only (no device): the same sub-mesh without a context -- counts8 =
 * {local nodes, owned nodes, local elements, blocks of the owned rows, blocks
 * of all local rows, peers, rows sent, rows received}; with non-null arrays
 * (sized by a first call): the caller's ids of the local nodes and elements,
 * and the block rows of the OWNED nodes as the rank builds them from its own
 * elements (rowptr[owned + 1], colidx = the caller's id of the column node).  */
int feahip_host_rank_mesh(int rank, int nranks, int n_nodes, int n_elems, int npe,
                          const int *elements, const double *nodes0, long long *counts8,
                          int *node_global, int *elem_global, long long *rowptr, int *colidx);

/* Host-only: that sub-mesh's halo plan in the caller's node ids -- counts3 =
 * {peers, rows sent, rows received} by a first call with null lists, then
 * peers[npeers], send_off / recv_off[npeers + 1], send_idx / recv_idx in the
 * order the rows travel.                                                      */
int feahip_host_rank_plan(int rank, int nranks, int n_nodes, int n_elems, int npe,
                          const int *elements, const double *nodes0, int *counts3, int *peers,
                          int *send_off, int *recv_off, int *send_idx, int *recv_idx);

/* In-process group: n contexts of the same mesh (on any devices) driven by one
 * host thread; halo rows move by device copies, sums on the host.  Same
 * kernels and halo plan as the RCCL path.                                    */
int feahip_group_init(feahip_ctx **ctxs, int n);
int feahip_group_solve_slae(feahip_ctx **ctxs, int n, int solver_type, double tolerance,
                            int max_iterations, int *iters, double *resid);
int feahip_group_energy(feahip_ctx **ctxs, int n, double *tolerance);
int feahip_group_update_nodes_with_solution(feahip_ctx **ctxs, int n);
int feahip_group_solve(feahip_ctx **ctxs, int n, int load_increments, int max_newton,
                       int modified_newton, double desired_tolerance, int solver_type,
                       double solver_tolerance, int solver_max_iter, double *tol_log,
                       int tol_log_cap, int *its_log, int *steps_done);

/* Host-only (no device): the halo plan of one rank from the element->node
 * map, in the numbering it is given (a context plans in library ids: pass
 * elements translated by feahip_host_numbering to get what it gets).  First call with null lists fills counts[5] = {npeers, nsend, nrecv,
 * row0, row1}; second call fills peers[npeers], send_off/recv_off[npeers+1],
 * send_idx[nsend], recv_idx[nrecv] (node ids, ascending per peer).           */
int feahip_shard_plan(int n_nodes, int n_elems, int npe, const int *elements, int rank,
                      int nranks, int *counts, int *peers, int *send_off, int *recv_off,
                      int *send_idx, int *recv_idx);
/* Host-only (no device is touched): what the assembly maps `rank` of `nranks`
 * builds for ITS block rows say, as one hash per row of the set of (row,
 * column, element, local row node, local column node) contributions they
 * list; rows[0..1] = the rows the rank owns, rowhash[a] = 0 for every other
 * row.  The maps of a shard are cut differently from the unsharded ones, what
 * they say about a row must not be: the hashes of all ranks add up to the
 * unsharded ones (tests/test_host.py).                                      */
int feahip_host_assembly_digest(int n_nodes, int n_elems, int npe, const int *elements, int rank, int nranks,
                                unsigned long long *rowhash, int *rows);

/* Host-only (no device): shape of the GATHER maps (npe = 4, 10 or 8) for a
 * mesh in the numbering it is given (pass library ids to see what a context
 * builds): stats[8] = {chunks, element evaluations, distinct elements, rows,
 * chunks repeating their predecessor's map words, map bytes, chunks with a
 * block list / a diagonal list longer than a thread keeps in registers};
 * rows_hist[65] (may be null) = chunks by row count.  An element is evaluated
 * once per chunk that owns one of its nodes (fea_solver.c:887-1068 visits each
 * element once): evaluations / distinct elements is what a numbering costs.  */
int feahip_host_gather_stats(int n_nodes, int n_elems, int npe, const int *elements,
                             long long *stats, int *rows_hist);

/* ---- node numbering ---------------------------------------------------- */
/* The reference keeps the nodes in deck order (sexp_loader.c:170-215) and its
 * dof index is node * 3 + axis (fea_solver.c:377-384).  The kernels here own
 * runs of consecutive block rows, so feahip_create numbers the nodes itself
 * (compact cells of 4 x 4 x 4 nodes (48 half-grid nodes for 10-node elements), cells in slabs across the longest axis;
 * a mesh that sits on no lattice: recursive coordinate bisection into leaves of up to 64 nodes, the first cuts
 * across the longest axis; csrc/renumber.cpp) and works in that numbering.  Every entry of this header
 * that takes or returns node-indexed data translates: the caller passes and
 * receives its OWN node ids and dof indices, bit-exactly -- elements,
 * prescribed node ids, coordinates, forces, solution, the Yale matrix (rows,
 * columns, sorted as the caller's ids sort), SpMV vectors.  Only the shard
 * ranges (feahip_owned_rows, feahip_shard_plan, feahip_host_assembly_digest)
 * speak of library ids.  library_id_of_node[n_nodes]: library id of the
 * caller's node a (the identity when the caller's numbering was kept).       */
int feahip_node_numbering(feahip_ctx *ctx, int *library_id_of_node);
/* Host-only (no device): the numbering feahip_create would choose for this
 * mesh; returns 1 when it is a renumbering, 0 when the caller's ids are kept
 * (the identity is written then), negative on error.                         */
int feahip_host_numbering(int n_nodes, int n_elems, int npe, const int *elements,
                          const double *nodes0, int *library_id_of_node);

/* ---- reference-shaped views -------------------------------------------- */

int feahip_set_nodes(feahip_ctx *ctx, const double *nodes);   /* nodes_p    */
int feahip_get_nodes(feahip_ctx *ctx, double *nodes);         /* [N][3]     */
int feahip_get_forces(feahip_ctx *ctx, double *f);            /* [3N]       */
int feahip_set_forces(feahip_ctx *ctx, const double *f);
int feahip_get_solution(feahip_ctx *ctx, double *u);          /* [3N]       */
/* graddefs[e][g].components / stresses[e][g].components
 * (fea_solver.h:262-269), layout [E][G][3][3]                               */
int feahip_get_graddefs(feahip_ctx *ctx, double *F);
int feahip_get_stresses(feahip_ctx *ctx, double *S);
/* shape_gradients[e][g] of the CURRENT configuration (fea_solver.h:200-205,
 * filled by solver_create_current_shape_gradients, fea_solver.c:656-722,831):
 * grads[((e*G + g)*3 + i)*npe + a] = dN_a/dx_i, detj[e*G + g] = det J.      */
int feahip_get_shape_gradients(feahip_ctx *ctx, double *grads, double *detj);

/* global_mtx in sp_matrix_yale shape (fea_solver.c:303-304): scalar CSR of
 * the full symmetric pattern, sorted columns.                               */
int feahip_matrix_nnz(feahip_ctx *ctx, long long *nnz);
int feahip_get_matrix_yale(feahip_ctx *ctx, int *offsets, int *indexes,
                           double *values);
/* The same with 64-bit offsets.  feahip_get_matrix_yale REFUSES (FEAHIP_EINVAL, feahip_last_error says why) a matrix
 * of 2^31 or more scalar non-zeros instead of wrapping its int offsets: one rank of eight of BASELINE configs[4]
 * (50M 10-node tetrahedra) already holds 2.2e9.                                */
int feahip_get_matrix_yale64(feahip_ctx *ctx, long long *offsets, int *indexes,
                             double *values);
/* y = K x with host vectors (test hook for the SpMV kernel)                 */
int feahip_spmv(feahip_ctx *ctx, const double *x, double *y);

/* ---- tuning and measurement -------------------------------------------- */

int feahip_set_assembly(feahip_ctx *ctx, int strategy);
/* Preconditioner of the PCG_ILU / CHOLESKY solves: 0 = inverse 3x3 diagonal
 * blocks (default), 1 = aggregation multigrid (rigid-body modes of every
 * aggregate, W-cycle).  In a sharded solve every rank builds the hierarchy of
 * its own diagonal block and the preconditioner is block-Jacobi over the ranks
 * with a W-cycle inside each: no communication beyond the CG's own.  Either
 * way the solve runs to the requested residual, so the solution is the same
 * to that tolerance.                                                          */
int feahip_set_preconditioner(feahip_ctx *ctx, int kind);
/* The CG / PCG loop of feahip_solve_slae.  0: the textbook loop (what the
 * reference's sp_matrix_yale_solve_cg / _pcg_ilu run, fea_solver.c:245-280):
 * two reductions per iteration (p.Kp, then r.z and r.r), halo rows exchanged
 * before the product.  1: the single-reduction form of the same recurrence
 * (Chronopoulos / Gear): p = z + beta p and s = w + beta s with w = K z kept
 * by recurrence, so that r.z, w.z and r.r of an iteration are summed together
 * -- ONE all-reduce of three doubles per iteration -- and the halo rows of z
 * travel while the rows that touch no halo column are multiplied.  Same
 * iterates in exact arithmetic; the solve runs to the same residual.  -1
 * (default): 1 for a sharded context, 0 otherwise.                            */
int feahip_set_pcg_variant(feahip_ctx *ctx, int variant);
/* Line search along every Newton step of feahip_solve / feahip_group_solve:
 * golden-section search, `max_iterations` iterations, for the step length in
 * [1/2, 1] that minimises |eta <u, R(x + eta u)>| -- what the reference's
 * prototype does (solver-prototype/cartesian3d/large/cartesian3d_large.m:
 * 85-119) and what its C solver parses as `line-search :max` and leaves unused
 * (fea_solver.c:1517, sexp_loader.c:153-159).  0 (default) = the reference's
 * solve().  Two residual assemblies per iteration.                         */
int feahip_set_line_search(feahip_ctx *ctx, int max_iterations);
/* Restricts assembly and SpMV to this rank's slab of block rows (rank of
 * nranks, contiguous row ranges of near-equal block count).  Rows are owned
 * by exactly one rank; a rank visits every element that touches its rows, so
 * assembly needs no exchange between ranks.                                  */
int feahip_set_row_shard(feahip_ctx *ctx, int rank, int nranks);
int feahip_sync(feahip_ctx *ctx);
/* Runs `iters` timed launches of one hot-path kernel after `warmup` untimed
 * ones, bracketed by HIP events on the context's own stream; *avg_ms is the
 * mean device time of one launch.  what: 0 stiffness+residual assembly,
 * 1 stiffness only, 2 residual only, 3 SpMV, 4 one PCG iteration.           */
int feahip_time_kernel(feahip_ctx *ctx, int what, int warmup, int iters,
                       double *avg_ms);
/* Streaming copy of `bytes` bytes (16 bytes per lane, read + written counted)
 * on the context's device and stream: the copy bandwidth of this box, to quote
 * roofline fractions against next to the data-sheet peak.     */
int feahip_copy_bandwidth(feahip_ctx *ctx, long long bytes, double *gbytes_per_s);
/* The same measured four ways (the call above reports the best): out4[0] one
 * 16-byte load in flight per lane (grid-stride), out4[1] four independent
 * 16-byte loads in flight per lane, out4[2] hipMemcpyDtoDAsync, out4[3] as
 * [1] with non-temporal loads and stores.                                   */
int feahip_copy_bandwidth_detail(feahip_ctx *ctx, long long bytes, double *out4);
/* device addresses of K, the column indices and the two SpMV vectors (for
 * the alignment column of a bandwidth report): out4                         */
int feahip_device_layout(feahip_ctx *ctx, long long *out4);
/* sizes the roofline model needs: N, E, npe, G, block rows, blocks, and the
 * bytes of the auxiliary maps the kernels read                              */
int feahip_sizes(feahip_ctx *ctx, long long *out8);
/* What the gather strategy's maps look like on this mesh (built on first use; zeros when another strategy runs):
 * out4[0] element evaluations per element the rank touches (1.75 for a 4x4x4 brick of a Kuhn block), out4[1] gather
 * chunks, out4[2] chunks whose map words equal their predecessor's (kept in registers by the kernel), out4[3] map
 * bytes.  For reports (bench.py extras); reference has no counterpart.                                                  */
int feahip_assembly_stats(feahip_ctx *ctx, double *out4);
/* the strategy (FEAHIP_ASM_*) the most recent assembly launch ran -- what
 * FEAHIP_ASM_AUTO resolved to on this mesh; FEAHIP_ASM_AUTO before any launch */
int feahip_assembly_in_use(feahip_ctx *ctx, int *strategy);

#ifdef __cplusplus
}
#endif
#endif
