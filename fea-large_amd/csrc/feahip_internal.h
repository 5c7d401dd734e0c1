// feahip_internal.h -- context layout shared by the translation units of
// libfeahip.so.  Not part of the ABI (include/fea_hip.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "../../include/fea_hip.h"

// Every device allocation of the library goes through here.  FEAHIP_TEST_NAN_ALLOC=1 (test knob: results must not
// change) fills a fresh allocation with 0xFF bytes -- NaN as a double or a float, -1 as an integer -- so that a read of
// memory the library never wrote shows in the results instead of depending on what the allocator handed back
// (GPU AddressSanitizer is not available on the target pool).
hipError_t feahip_device_malloc(void **p, size_t bytes);
#define hipMalloc(p, bytes) feahip_device_malloc((void **)(p), (bytes))

#define FEA_MAX_NPE 10
#define FEA_MAX_GAUSS 27

// one wave owns a run of block rows ("chunk"); its 3x3 blocks are summed in
// LDS and written to HBM once.  CHUNK_BLOCKS bounds the LDS per wave.
#define FEA_CHUNK_BLOCKS 128
#define FEA_CHUNK_ROWS 16
#define FEA_WAVES_PER_WG 4
// grid used by the vector / reduction kernels: their per-block partial sums
// are re-reduced by every block of the consuming kernel.
#define FEA_RED_BLOCKS 2048

struct ElemTable {            // element plug-in, tabulated by the host
  double w[FEA_MAX_GAUSS];
  double dN[FEA_MAX_GAUSS][3][FEA_MAX_NPE];
};

struct feahip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // sizes
  int N = 0, E = 0, npe = 0, G = 0, ndof = 0;
  int nnzb = 0;               // 3x3 blocks in the full symmetric pattern
  int nchunks = 0;
  int chunk0 = 0, nchunks_local = 0;  // this rank's share of the chunks (row shard)
  int nachunks = 0, achunk0 = 0, nachunks_local = 0;   // same for the staged kernel's partition
  std::vector<int> h_super_achunk, h_chunk;
  // row shard of a multi-rank solve: this rank owns nodes [row0, row1)
  int rank = 0, nranks = 1, row0 = 0, row1 = 0;
  struct Transport *tr = nullptr;      // null: single rank, no exchange
  bool owns_tr = false;
  std::vector<int> peer, send_off, recv_off;   // halo plan: per peer, ranges into the index lists
  int nsend = 0, nrecv = 0;
  int *d_send_idx = nullptr, *d_recv_idx = nullptr;
  double *d_send_buf = nullptr, *d_recv_buf = nullptr;
  int max_rowlen = 0;
  bool linear_tet = false;    // npe == 4 and dN is the constant-strain table
  int model = 0;
  double lambda = 0, mu = 0;
  int strategy = FEAHIP_ASM_AUTO;
  int last_strategy = FEAHIP_ASM_AUTO;   // what the most recent assembly launch ran

  ElemTable table;
  ElemTable *d_table = nullptr;

  // mesh (device)
  int *d_conn = nullptr;       // [E][npe]
  double *d_X0 = nullptr;      // [N][4] padded to 32 B (two dwordx4 per node)
  double *d_x = nullptr;       // [N][4] current configuration
  // block-CSR pattern of K (built once; topology never changes)
  int *d_rowptr = nullptr;     // [N+1]
  int *d_colidx = nullptr;     // [nnzb]
  // K holds the block rows this rank owns and nothing else: blocks [kb0, kb1) = rowptr[row0] .. rowptr[row1] (all of
  // them for an unsharded context).  It is allocated on first use, for the shard installed by then, so a rank of a
  // sharded run never holds the other ranks' rows (the reference keeps one row-wise store, fea_solver.c:444-448,
  // and its modified-Newton copy, :179).  Kernels index by GLOBAL block number through d_K = d_K_base - 9 kb0.
  double *d_K = nullptr;       // [nnzb][3][3], valid for blocks [kb0, kb1) only
  double *d_Kstash = nullptr;  // modified-Newton copy (fea_solver.c:179), same window
  double *d_K_base = nullptr, *d_Kstash_base = nullptr;   // first owned value (block kb0) of each
  double *d_K_alloc = nullptr, *d_Kstash_alloc = nullptr; // the allocations: d_K_base = d_K_alloc + (kb0 & 1), so that even GLOBAL value indices are 16-byte aligned on every rank
  long long kb0 = 0, kb1 = 0;
  bool have_stash = false;
  // node -> element incidence (row-owner assembly)
  int *d_incptr = nullptr;     // [N+1]
  uint32_t *d_inc = nullptr;   // [npe*E]  elem | local<<28
  uint8_t *d_incslot = nullptr;// [npe*E][npe] slot of column conn[e][b] in row
  int *d_chunk = nullptr;      // [nchunks+1] first row of every chunk
  int *d_diag = nullptr;       // [N] index of the diagonal block of every row
  // LDS-staged visit assembly maps (linear tetrahedra)
  bool have_visits = false, visits_failed = false;
  struct VisitDesc *d_vdesc = nullptr;
  int *d_vnode = nullptr;
  uint32_t *d_vrec = nullptr;
  long long visit_bytes = 0;
  int nvisit_records = 0;      // length of vrec in records (whole passes per chunk)
  bool have_quad = false, quad_failed = false;
  int quad_a0 = -1, quad_n = 0;        // assembly chunks the quad maps were built for (this rank's)
  struct QuadDesc *d_qdesc = nullptr;
  uint32_t *d_qelem = nullptr, *d_qpair = nullptr;
  int *d_qnode = nullptr;
  long long quad_bytes = 0;
  // GATHER assembly maps (linear tetrahedra, kernels_gather.hip): built for the rows this rank owns
  bool have_gather = false, gather_failed = false;   // gather_failed: the maps did not build for rows [gather_fail_row0, gather_fail_row1)
  int gather_fail_row0 = -1, gather_fail_row1 = -1;
  int gather_declined_row0 = -1, gather_declined_row1 = -1;   // AUTO looked at the gather chunks of these rows and chose another kernel (linear tets)
  int gather_row0 = -1, gather_row1 = -1, ngchunks = 0;
  unsigned char *d_gmaps = nullptr;
  struct GatherLayout *gather_lay = nullptr;
  long long gather_bytes = 0;
  double gather_evals_per_element = 0;   // element evaluations the gather chunks make per element this rank touches
  int gather_same_words = 0;             // gather chunks whose map words equal their predecessor's
  // the same for 10-node tetrahedra (kernels_gather10.hip); shares d_gmaps / ngchunks / gather_row0.. with the above
  struct Gather10Layout *gather10_lay = nullptr;
  int *d_g10_elist = nullptr;            // this rank's elements
  double *d_g10_state = nullptr;         // [G][elements of the rank][18]: Gauss-point state records (kernels_gather10.hip)
  int g10_nloc = 0;
  // vectors (3N doubles)
  double *d_f = nullptr, *d_u = nullptr;
  double *d_r = nullptr, *d_p = nullptr, *d_q = nullptr, *d_minv = nullptr;
  double *d_part = nullptr;    // reduction partials, 6 x FEA_RED_BLOCKS
  // single-reduction PCG (kernels_solve.hip): preconditioned residual z = M r and w = K z (allocated on first use);
  // pcg_variant: -1 = single-reduction when sharded, the reference-shaped two-reduction loop otherwise; 0 / 1 force
  double *d_z = nullptr, *d_w = nullptr, *d_s = nullptr;
  int pcg_variant = -1;
  // SpMV chunks [chunk0 + ichunk_lo, chunk0 + ichunk_hi) of this rank touch no halo column: they run while the halo
  // rows are in flight (everything, for an unsharded context)
  int ichunk_lo = 0, ichunk_hi = 0;
  hipStream_t comm_stream = nullptr;   // RCCL transport: halo exchange beside the interior product
  hipEvent_t ev_packed = nullptr, ev_unpacked = nullptr;
  double *d_scal = nullptr;    // device scalars of the CG recurrences
  int *d_flag = nullptr;       // [0] converged-at iteration, [1] bad Gauss pts
  // prescribed displacements
  int n_presc = 0;             // nodes in the deck
  int n_cdof = 0;              // constrained dofs, deck order x,y,z per node
  int *d_cdof = nullptr;
  double *d_cval = nullptr;    // prescribed value per constrained dof
  uint8_t *d_dofmask = nullptr;// [3N] 1 = constrained
  // cached per-Gauss-point state for the getters
  double *d_F = nullptr, *d_S = nullptr;   // [E][G][9]
  bool state_valid = false;

  // host copies needed by getters / pattern export
  std::vector<int> h_rowptr, h_colidx;
  // The maps of a strategy (gather chunks, staged visits, generic incidence lists) are built the first time it is
  // asked for, from these host copies.
  std::vector<int> h_conn;
  struct HostPattern *h_pat = nullptr;
  bool generic_maps = false;   // incptr / inc / incslot uploaded
  bool incslot_ok = false;     // the mesh has them (row length <= 255)
  long long aux_bytes = 0;

  int last_bad = 0;

  // library-side node numbering (renumber.cpp): everything in the context -- mesh arrays, pattern, K, vectors, shard
  // ranges -- lives in the library's numbering; the ABI translates at its boundary.  Empty = the caller's numbering.
  std::vector<int> perm, iperm;        // perm[caller id] = library id, iperm = its inverse
  // a rank context (feahip_create_rank): this context IS one rank's sub-mesh, locally indexed; its "caller ids" are the
  // local ids, rank_node_global / rank_elem_global say which nodes and elements of the whole mesh they are
  int rank_own = -1;                   // nodes it owns (local ids [0, rank_own)); -1: an ordinary context
  std::vector<int> rank_node_global, rank_elem_global;
  int rank_n_global = 0;

  // preconditioner of PCG_ILU / CHOLESKY solves: 0 = 3x3 block-Jacobi, 1 = aggregation multigrid (amg.h)
  // which matrix d_K holds: bumped by every stiffness assembly, copied by stash / restore; k_bc = prescribed-dof
  // masking applied since.  Only used to skip numeric re-setup of the multigrid hierarchy for an unchanged K
  // (modified Newton restores the same matrix every iteration); a stale hierarchy would cost iterations, not accuracy.
  unsigned long long k_epoch = 0, stash_epoch = 0;
  bool k_bc = false;
  // golden-section line search along the Newton step: iterations (0 = off, the reference's solve())
  int linesearch_max = 0;
  int precond = 0;
  void *amg = nullptr;         // AmgHierarchy, built on first use
};

#define FEA_HIP_CHECK(ctx, call)                                            \
  do {                                                                      \
    hipError_t _e = (call);                                                 \
    if (_e != hipSuccess) {                                                 \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(_e);       \
      return FEAHIP_EHIP;                                                   \
    }                                                                       \
  } while (0)

// pattern.cpp
struct HostPattern {
  std::vector<int> rowptr, colidx;       // block CSR, sorted columns
  std::vector<int> incptr;               // [N+1]
  std::vector<uint32_t> inc;             // [npe*E]
  std::vector<uint8_t> incslot;          // [npe*E*npe]
  std::vector<int> chunk;                // chunk -> first row
  std::vector<int> diag;                 // row -> index of its diagonal block
  std::vector<uint32_t> inc_rows;        // inc before the per-chunk interleave (row-sorted)
  std::vector<int> achunk;               // assembly partition of the staged kernel (4-node elements)
  std::vector<int> super_achunk;         // first achunk of every super, [nsuper+1]
  int max_rowlen = 0;
  int break_chunk = 0, break_super = 0;  // first chunk / super at or behind row_break (their counts when there is no break)
};
int build_host_pattern(int N, int E, int npe, const int *conn, HostPattern &hp,
                       std::string &err, int row_break = -1);

// visits.cpp
// LDS-staged visit assembly (kernels_visit.hip): per chunk, the nodes its
// elements touch (owned rows first) and one 8-byte record per (row, element)
// visit: 4 chunk-local node ids (row node first) + 3 column slots.
// assembly chunks of the staged kernel are smaller than the SpMV chunks (LDS
// per wave sets the occupancy): two passes of 64 visits, 64 nodes, 80 blocks
#define FEA_VISIT_MAX_NODES 64
#define FEA_VISIT_MAX_VISITS 128
#define FEA_ACHUNK_BLOCKS 80
#define FEA_ACHUNK_ROWS 8              // rows of a staged-assembly chunk (sizes its f tile: 16 workgroups per CU fit)
#define FEA_SUPER_CHUNKS 8            // a "super" = 8 SpMV chunks; both partitions break at supers; unit of the row shard
struct VisitDesc {                   // 32 bytes, one per chunk
  int r0, r1, b0, nb;
  int node_off, nnode;               // into vnode
  int visit_off, nvisit;             // into vrec: first record, records (a multiple of 64: whole passes)
};
struct HostVisits {
  std::vector<VisitDesc> desc;
  std::vector<int> vnode;
  std::vector<uint32_t> vrec;        // [2 * visits]: ids (4 x u8), then u8 x 4: parity flag, tile positions of the 3 blocks
  bool ok = false;
};
void build_host_visits(int N, int E, const int *conn, const HostPattern &hp, HostVisits &out);
// shared-state assembly of 10-node elements (kernels_quad.hip): per chunk the
// distinct elements touching its rows and one 4-byte record per
// (row, element, column node) pair
#define FEA_QUAD_BLOCKS 256           // K tile of one workgroup (tile position: 8 bits of the pair word)
#define FEA_QUAD_PAIRS 512            // pairs of a multi-row chunk (a single row may have more)
#define FEA_QUAD_ELEMS 64             // distinct elements of a chunk (6-bit index in a pair record)
#define FEA_QUAD_NODES 126            // coordinate tile: distinct nodes of the chunk's elements
#define FEA_QUAD_VISITS 96            // (row, element) visits of a chunk: one row-vector record per visit and Gauss point of a batch
struct QuadDesc {                    // 40 bytes, one per chunk
  int r0, r1, b0, nb;
  int elem_off, nelem;               // into qelem (3 words per element)
  int pair_off, npair;               // into qpair
  int node_off, nnode;               // into qnode
};
struct HostQuad {
  std::vector<QuadDesc> desc;
  std::vector<int> qnode;            // global ids of the chunk's nodes
  std::vector<uint32_t> qelem;       // per element 12 bytes: 10 chunk-local node ids (u8), flags (bit 0: its local node 0 is a row of this chunk), 0
  std::vector<uint32_t> qpair;       // el(6) | la(4)<<6 | lb(4)<<10 | tile position(8)<<14 | local row(4)<<22 | first pair of its visit<<26
  bool ok = false;
};
void build_host_quad(int N, int E, int npe, const int *conn, const HostPattern &hp, int p_lo, int p_hi, HostQuad &out);
int ensure_quad(feahip_ctx *c);
int launch_assemble_quad(feahip_ctx *c, bool doF);
// GATHER assembly (kernels_gather.hip, gather.cpp): a 256-thread workgroup owns a run of consecutive block rows.
// Per chunk the host prepares one fixed-stride record: header, the chunk's nodes (owned rows first), its distinct
// elements as 4 chunk-local node ids, and per off-diagonal block the list of (element, local row node, local
// column node) contributions that sum to it; per residual thread a slice of one row's (element, local node) visits.
#ifndef FEA_G_BIG
#define FEA_G_BIG 1
#endif
#if FEA_G_BIG == 1
#define FEA_G_THREADS 1024            // one workgroup per CU: sixteen waves share one chunk's records (see kernels_gather.hip)
#define FEA_G_TASK_THREADS 768        // block and residual threads: waves 0-11; waves 12-15 (FEA_G_DIAG_LANES) sum the diagonal blocks
#define FEA_G_MAX_ROWS 64
#define FEA_G_MAX_NODES 256           // 8-bit chunk-local node slots
#define FEA_G_MAX_ELEMS 719           // slots come in sixteens and one stays all-zero; 45 x 16 = 720 records of 208 B fill the LDS next to the coordinates
#define FEA_G_ELEMS_TARGET 672        // a 4 x 4 x 4 brick of nodes of a Kuhn block
#define FEA_G_CELL 4, 4, 4
#elif FEA_G_BIG == 2
#define FEA_G_THREADS 512             // two workgroups per CU, eight waves each
#define FEA_G_TASK_THREADS 384
#define FEA_G_MAX_ROWS 32
#define FEA_G_MAX_NODES 128
#define FEA_G_MAX_ELEMS 351           // 22 x 16 = 352 records of 208 B + the coordinates: 80 KB
#define FEA_G_ELEMS_TARGET 320        // a 4 x 3 x 2 brick of nodes of a Kuhn block touches 300 elements
#define FEA_G_CELL 4, 2, 3
#else
#define FEA_G_THREADS 256             // three workgroups per CU
#define FEA_G_TASK_THREADS 192
#define FEA_G_MAX_ROWS 16
#define FEA_G_MAX_NODES 128
#define FEA_G_MAX_ELEMS 239
#define FEA_G_ELEMS_TARGET 216        // a 4 x 2 x 2 brick
#define FEA_G_CELL 4, 2, 2
#endif
#define FEA_G_DIAG_LANES (FEA_G_THREADS - FEA_G_TASK_THREADS)
#define FEA_G_SLOT_BITS 10            // record slot inside a contribution / visit entry: slot | la << 10 | lb << 12
#define FEA_G_MAX_SLOTS 1024
#define FEA_G_REGW 6                  // contribution words a block thread keeps in registers (2 entries each).  A lattice needs 3 (six
                                      // elements around an edge at most); on the reference's TetGen deck 92 % of the chunks have a list of
                                      // 5 words and 4 % one of 6, and with 4 in registers the rest was fetched INSIDE the gather phase,
                                      // one exposed HBM round trip per word (kernels_gather.hip); the words are loaded only up to the
                                      // mesh's longest list (GatherLayout::max_depth), so a lattice issues no more loads than before
struct GatherHeader {                // 64 bytes, first thing in a chunk record
  int r0, r1, b0, nb;                // rows [r0, r1), blocks [b0, b0+nb) of the CSR
  int nnode, nelem, noffd, depth;    // depth: contribution words per block thread
  int nvthr, vdepth;                 // residual threads, visits per residual thread
  int ddepth;                        // diagonal-block words per lane of the last wave
  unsigned wdepth[3];                // contribution words the block threads of wave slot w walk (the longest list of ITS blocks), one byte per slot
  int flags;                         // bit 0: the map words of the NEXT chunk of the context (everything but header and node list) equal this chunk's
  int pad[1];
};
static_assert(sizeof(GatherHeader) == 64 && FEA_G_TASK_THREADS / 64 <= 12, "GatherHeader: 64 bytes, twelve block-wave slots");
struct GatherLayout {                // the same for every chunk of a context
  int stride;                        // bytes per chunk record
  int o_nodes, o_elems, o_bpos, o_rows, o_vlist, o_dlist, o_clist;   // byte offsets of the sections
  int max_nodes, max_elems, max_tile;                        // LDS tiles: coordinates, element records, K blocks
  int max_tasks, max_depth, max_vthr, max_vdepth, max_ddepth;   // largest chunk: block threads, contribution words, residual threads, visits, diagonal words
};
struct HostGather {
  GatherLayout lay;
  std::vector<unsigned char> blob;   // nchunks records of lay.stride bytes
  std::vector<int> first_row;        // [nchunks+1]
  long long total_evals = 0, distinct_elems = 0;   // element evaluations of all chunks; elements touching the rows
  int nchunks = 0;
  int same_as_previous = 0;          // chunks whose map words equal their predecessor's (GatherHeader::flags)
  bool ok = false;
};
// rows [row_lo, row_hi) only: a rank builds the maps of the rows it owns
void build_host_gather(int N, int E, const int *conn, const HostPattern &hp, int row_lo, int row_hi, HostGather &out);
int ensure_gather(feahip_ctx *c);
int launch_assemble_gather(feahip_ctx *c, bool doK, bool doF);

// GATHER assembly of 10-node tetrahedra (kernels_gather10.hip, gather10.cpp).  Same idea as the 4-node one with the
// Gauss points as an outer loop: a 256-thread workgroup owns up to 64 consecutive block rows, evaluates every
// distinct element touching them once per Gauss point into LDS records (spatial gradient g_k and traction vector
// t_k of its ten nodes), and every thread sums up to five off-diagonal blocks over all Gauss points in registers.
#ifndef FEA_Q_THREADS
#define FEA_Q_THREADS 256             // two workgroups per CU (the accumulators of five blocks per thread take the register file:
                                      // 384 threads x 4 blocks and 512 x 3 spill 240-370 bytes per lane at their register budgets)
#endif
#define FEA_Q_WAVES (FEA_Q_THREADS / 64)
#define FEA_Q_MAX_ROWS 64
#define FEA_Q_MAX_NODES 240           // 8-bit chunk-local node ids
#define FEA_Q_MAX_ELEMS 127           // 7-bit record slot; the slot after the last one in use is the all-zero record
#ifndef FEA_Q_SLOTS
#define FEA_Q_SLOTS 5                 // blocks per thread
#endif
#define FEA_Q_REGW 4                  // contribution words per block a thread keeps in registers (2 entries each)
#define FEA_Q_MAX_PASS 7              // write-out passes of one chunk through the K tile
#define FEA_Q_FLANES 128              // residual lanes (the last two waves)
#define FEA_Q_ROWS_U16 200            // rstart[65] | rdiag[64] at 66 | ffirst[65] at 130
struct Gather10Header {              // 128 bytes
  int r0, r1, b0, nb;
  int nnode, nelem, ntask, npass;
  int nft, fdw;                      // residual lanes, words per residual lane (2 visits each)
  unsigned char prow[8];             // pass p writes the rows [prow[p], prow[p+1]) of the chunk
  unsigned char cnt[40];             // contributions of the longest list among the 64 blocks wave w holds in slot s, at [FEA_Q_WAVES s + w]
  unsigned char sw[8];               // list words stored for slot s (the longest of its waves); rows of the clist section
  int pad[8];
};
static_assert(FEA_Q_SLOTS * FEA_Q_WAVES <= 40 && FEA_Q_SLOTS <= 8, "Gather10Header: cnt / sw too small");
static_assert(sizeof(Gather10Header) == 128, "Gather10Header is 128 bytes");
struct Gather10Layout {
  int stride;
  int o_nodes, o_elems, o_rows, o_tpos, o_flist, o_clist;
  int max_nodes, max_elems, max_cw, max_fdw, tile_blocks;      // max_cw: clist rows (256 words each) of the longest chunk
};
struct HostGather10 {
  Gather10Layout lay;
  std::vector<unsigned char> blob;
  std::vector<int> first_row;
  std::vector<int> elist;            // the rank's elements (touching its rows), ascending: order of the state records
  int npe = 10;
  long long total_evals = 0, distinct_elems = 0;
  int nchunks = 0;
  bool ok = false;
};
void build_host_gather10(int N, int E, int npe, const int *conn, const HostPattern &hp, int row_lo, int row_hi, HostGather10 &out);
int ensure_gather10(feahip_ctx *c);
int launch_assemble_gather10(feahip_ctx *c, bool doK, bool doF);

int launch_assemble_visit(feahip_ctx *c, bool doK, bool doF);

// launchers (kernels_assemble.hip / kernels_patch.hip / kernels_solve.hip)
int launch_assemble(feahip_ctx *c, bool doK, bool doF);
int launch_state_export(feahip_ctx *c, double *d_grads = nullptr, double *d_detj = nullptr);
int launch_apply_bc(feahip_ctx *c, double lambda);
int launch_update_nodes_bc(feahip_ctx *c, double lambda);
int ensure_generic_maps(feahip_ctx *c);
int ensure_k(feahip_ctx *c);
void release_k(feahip_ctx *c);
int ensure_visits(feahip_ctx *c);
int dist_nodes_add_scaled(std::vector<feahip_ctx *> &R, double eta, bool exchange);
int launch_update_nodes_solution(feahip_ctx *c, const double *d_u);
int launch_spmv(feahip_ctx *c, const double *d_xv, double *d_yv);
int solve_pcg(feahip_ctx *c, int type, double tol, int max_iter, int *iters,
              double *resid);
int time_pcg_iteration(feahip_ctx *c, int warmup, int iters, double *avg_ms);

// renumber.cpp -- locality numbering of the nodes; false = no basis for one (identity returned)
bool locality_numbering(int N, int E, int npe, const int *conn, const double *X, std::vector<int> &new_of_old);

// shard.cpp -- host-only plan of a row-sharded solve
struct ShardPlan {
  int rank = 0, nranks = 1, row0 = 0, row1 = 0;
  std::vector<int> peer, send_off, recv_off;   // [npeer], [npeer+1], [npeer+1]
  std::vector<int> send_idx, recv_idx;         // node ids, ascending inside every peer segment
};
// rows of rank k = rows of supers [nsuper*k/n, nsuper*(k+1)/n); chunk = SpMV chunk partition
void shard_row_range(const std::vector<int> &chunk, int rank, int nranks, int &row0, int &row1);
void build_shard_plan(const std::vector<int> &rowptr, const std::vector<int> &colidx,
                      const std::vector<int> &chunk, int rank, int nranks, ShardPlan &plan);

// rankmesh.cpp -- the sub-mesh one rank of a sharded run holds, locally indexed (owned nodes first, then halo)
struct RankMesh {
  int rank = 0, nranks = 1, npe = 0;
  int n_global = 0, n_own = 0;          // nodes of the whole mesh; nodes this rank owns (local ids [0, n_own))
  int lib0 = 0, lib1 = 0;               // the library ids it owns
  std::vector<int> node_global, node_lib;   // per local node: the caller's id, the library id
  std::vector<int> elem_global;         // per local element: the caller's element index
  std::vector<int> elements;            // [local elements][npe] local node ids
  std::vector<double> nodes0;           // [local nodes][3]
  std::vector<int> presc_node, presc_type;
  std::vector<double> presc_values;
  ShardPlan plan;                       // halo plan in local ids
};
void rank_row_range(int N, int npe, int rank, int nranks, int &g0, int &g1);
int build_rank_mesh(int rank, int nranks, int N, int E, int npe, const int *elements, const double *nodes0,
                    int n_presc, const int *presc_node, const int *presc_type, const double *presc_values,
                    RankMesh &out, std::string &err);
int install_plan(feahip_ctx *c, const ShardPlan &plan);      // dist.hip: halo lists to the device, interior chunk range

// multi-rank operations (kernels_solve.hip).  R = the ranks driven by this
// process: one context with the RCCL transport, or all contexts of an
// in-process group.
struct Transport {
  virtual ~Transport() {}
  // halo rows of vector `which` (0 = p, 1 = u, 2 = x, 3 = z) from their owners
  virtual int exchange(std::vector<feahip_ctx *> &R, int which) = 0;
  // the same in two halves: begin() leaves the exchange running (on a stream of its own where the transport has
  // one), end() makes the context's stream wait for the halo rows.  Work enqueued between the two must not read them.
  virtual int exchange_begin(std::vector<feahip_ctx *> &R, int which) { return exchange(R, which); }
  virtual int exchange_end(std::vector<feahip_ctx *> &R) { (void)R; return FEAHIP_OK; }
  // d_scal[8+slot .. 8+slot+n) summed over all ranks, result on every rank
  virtual int allreduce(std::vector<feahip_ctx *> &R, int slot, int n) = 0;
};
Transport *make_group_transport();
Transport *make_rccl_transport(feahip_ctx *c, int rank, int nranks, const void *unique_id, std::string &err);
int rccl_unique_id(void *out, int cap);
int install_shard(feahip_ctx *c, int rank, int nranks);
int dist_solve_pcg(std::vector<feahip_ctx *> &R, int type, double tol, int max_iter, int *iters, double *resid);
int dist_energy(std::vector<feahip_ctx *> &R, double *out);
int dist_update_nodes_with_solution(std::vector<feahip_ctx *> &R, const double *u_host);
int dist_newton(std::vector<feahip_ctx *> &R, int load_increments, int max_newton, int modified_newton,
                double desired_tolerance, int solver_type, double solver_tolerance, int solver_max_iter,
                double *tol_log, int tol_log_cap, int *its_log, int *steps_done);
