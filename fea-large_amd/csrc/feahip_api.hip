// feahip_api.hip -- extern "C" entry points of include/fea_hip.h.
#include "feahip_internal.h"

#undef hipMalloc
hipError_t feahip_device_malloc(void **p, size_t bytes)
{
  static int fill = -1;
  if (fill < 0) { const char *e = getenv("FEAHIP_TEST_NAN_ALLOC"); fill = e && atoi(e) > 0 ? 1 : 0; }
  const hipError_t rc = hipMalloc(p, bytes);
  if (rc == hipSuccess && fill && bytes) { (void)hipMemset(*p, 0xFF, bytes); (void)hipDeviceSynchronize(); }   // (the fill lands before anything a non-blocking stream does)
  return rc;
}
#define hipMalloc(p, bytes) feahip_device_malloc((void **)(p), (bytes))
#include "amg.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <utility>

static std::string g_create_error;

extern "C" const char *feahip_create_error(void) { return g_create_error.c_str(); }
extern "C" const char *feahip_last_error(const feahip_ctx *c) { return c ? c->err.c_str() : "null context"; }

// library id of a caller's node
static inline int lib_id(const feahip_ctx *c, int a) { return c->perm.empty() ? a : c->perm[a]; }

template <class T>
static int dev_upload(feahip_ctx *c, T **dst, const T *src, size_t n)
{
  FEA_HIP_CHECK(c, hipMalloc((void **)dst, sizeof(T) * (n ? n : 1)));
  if (n) FEA_HIP_CHECK(c, hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice));
  return FEAHIP_OK;
}

template <class T>
static int dev_zeros(feahip_ctx *c, T **dst, size_t n)
{
  FEA_HIP_CHECK(c, hipMalloc((void **)dst, sizeof(T) * (n ? n : 1)));
  FEA_HIP_CHECK(c, hipMemset(*dst, 0, sizeof(T) * (n ? n : 1)));
  return FEAHIP_OK;
}

int ensure_generic_maps(feahip_ctx *c)
{
  if (c->generic_maps) return FEAHIP_OK;
  if (!c->h_pat) { c->err = "incidence maps unavailable"; return FEAHIP_ESTATE; }
  const HostPattern &hp = *c->h_pat;
  int rc;
  if ((rc = dev_upload(c, &c->d_incptr, hp.incptr.data(), hp.incptr.size()))) return rc;
  if ((rc = dev_upload(c, &c->d_inc, hp.inc.data(), hp.inc.size()))) return rc;
  if (!hp.incslot.empty() && (rc = dev_upload(c, &c->d_incslot, hp.incslot.data(), hp.incslot.size()))) return rc;
  c->generic_maps = true;
  return FEAHIP_OK;
}

// K for the rows of the shard installed now (see feahip_internal.h)
int ensure_k(feahip_ctx *c)
{
  if (c->d_K_base) return FEAHIP_OK;
  c->kb0 = c->h_rowptr[c->row0]; c->kb1 = c->h_rowptr[c->row1];
  // +2: the SpMV reads aligned 80-byte windows; +1: a shard whose first block is odd starts one double into the
  // allocation, so that EVEN global value indices are 16-byte aligned on every rank (the gather kernels pick the
  // alignment of their 16-byte row stores from the parity of the chunk's first global block)
  const size_t n = (size_t)(c->kb1 - c->kb0) * 9 + 3;
  FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_K_alloc, sizeof(double) * n));
  // on the context's own (non-blocking) stream: a null-stream memset is not ordered against the kernels that follow
  FEA_HIP_CHECK(c, hipMemsetAsync(c->d_K_alloc, 0, sizeof(double) * n, c->stream));
  c->d_K_base = c->d_K_alloc + (c->kb0 & 1);
  c->d_K = c->d_K_base - (size_t)c->kb0 * 9;
  return FEAHIP_OK;
}

void release_k(feahip_ctx *c)
{
  if (c->d_K_alloc) (void)hipFree(c->d_K_alloc);
  if (c->d_Kstash_alloc) (void)hipFree(c->d_Kstash_alloc);
  c->d_K_alloc = c->d_Kstash_alloc = c->d_K_base = c->d_Kstash_base = c->d_K = c->d_Kstash = nullptr;
  c->have_stash = false; c->k_bc = false; ++c->k_epoch;
}

// shared-state maps of 10-node elements for the assembly chunks this rank owns
int ensure_quad(feahip_ctx *c)
{
  if (c->have_quad && c->quad_a0 == c->achunk0 && c->quad_n == c->nachunks_local) return FEAHIP_OK;
  if (c->quad_failed || c->npe != 10 || !c->h_pat || c->h_conn.empty()) return FEAHIP_OK;
  HostQuad hq;
  build_host_quad(c->N, c->E, c->npe, c->h_conn.data(), *c->h_pat, c->achunk0, c->achunk0 + c->nachunks_local, hq);
  for (void *p : {(void *)c->d_qdesc, (void *)c->d_qelem, (void *)c->d_qpair, (void *)c->d_qnode})
    if (p) (void)hipFree(p);
  c->d_qdesc = nullptr; c->d_qelem = c->d_qpair = nullptr; c->d_qnode = nullptr;
  c->have_quad = false;
  if (!hq.ok) { if (c->nranks == 1) c->quad_failed = true; return FEAHIP_OK; }
  int rc;
  if ((rc = dev_upload(c, &c->d_qdesc, hq.desc.data(), hq.desc.size()))) return rc;
  if ((rc = dev_upload(c, &c->d_qelem, hq.qelem.data(), hq.qelem.size()))) return rc;
  if ((rc = dev_upload(c, &c->d_qpair, hq.qpair.data(), hq.qpair.size()))) return rc;
  if ((rc = dev_upload(c, &c->d_qnode, hq.qnode.data(), hq.qnode.size()))) return rc;
  c->have_quad = true;
  c->quad_a0 = c->achunk0; c->quad_n = c->nachunks_local;
  c->quad_bytes = (long long)(hq.desc.size() * sizeof(QuadDesc) + hq.qelem.size() * 4 + hq.qpair.size() * 4 + hq.qnode.size() * 4);
  return FEAHIP_OK;
}

int ensure_visits(feahip_ctx *c)
{
  if (c->have_visits || c->visits_failed || !c->h_pat || c->h_conn.empty()) return FEAHIP_OK;
  HostVisits hv;
  build_host_visits(c->N, c->E, c->h_conn.data(), *c->h_pat, hv);
  if (!hv.ok) { c->visits_failed = true; return FEAHIP_OK; }
  int rc;
  if ((rc = dev_upload(c, &c->d_vdesc, hv.desc.data(), hv.desc.size()))) return rc;
  if ((rc = dev_upload(c, &c->d_vnode, hv.vnode.data(), hv.vnode.size()))) return rc;
  if ((rc = dev_upload(c, &c->d_vrec, hv.vrec.data(), hv.vrec.size()))) return rc;
  c->have_visits = true;
  c->nvisit_records = (int)(hv.vrec.size() / 2);
  c->visit_bytes = (long long)(hv.desc.size() * sizeof(VisitDesc) + hv.vnode.size() * 4 + hv.vrec.size() * 4);
  return FEAHIP_OK;
}

static int create_impl(feahip_ctx *c, int device, int n_nodes, int n_elems, int npe, int gauss_count,
                       const double *gauss_weights, const double *dforms, const int *elements,
                       const double *nodes0, int model, const double *model_params,
                       int params_count, int n_presc, const int *presc_node,
                       const int *presc_type, const double *presc_values)
{
  if (n_nodes <= 0 || n_elems <= 0 || !gauss_weights || !dforms || !elements || !nodes0 || !model_params) {
    c->err = "feahip_create: null or empty input"; return FEAHIP_EINVAL;
  }
  if (npe != 4 && npe != 8 && npe != 10) { c->err = "nodes per element must be 4, 8 or 10"; return FEAHIP_EINVAL; }
  if (gauss_count < 1 || gauss_count > FEA_MAX_GAUSS) { c->err = "gauss_count out of range"; return FEAHIP_EINVAL; }
  if (model != FEAHIP_MODEL_A5 && model != FEAHIP_MODEL_COMPRESSIBLE_NEOHOOKEAN) { c->err = "unknown material model"; return FEAHIP_EINVAL; }
  if (params_count < 2) { c->err = "material needs lambda and mu"; return FEAHIP_EINVAL; }
  if (n_presc < 0 || (n_presc > 0 && (!presc_node || !presc_type || !presc_values))) { c->err = "bad prescribed-displacement arrays"; return FEAHIP_EINVAL; }

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    c->err = "no HIP device visible: this path has no CPU fallback"; return FEAHIP_ENODEVICE;
  }
  if (device < 0 || device >= ndev) { c->err = "device index out of range"; return FEAHIP_EINVAL; }
  c->device = device;
  FEA_HIP_CHECK(c, hipSetDevice(device));
  FEA_HIP_CHECK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));

  c->N = n_nodes; c->E = n_elems; c->npe = npe; c->G = gauss_count; c->ndof = 3 * n_nodes;
  c->model = model; c->lambda = model_params[0]; c->mu = model_params[1];

  // ---- the library's own node numbering (renumber.cpp).  From here on `elements`, `nodes0` and `presc_node` are
  // the permuted copies; the caller's ids come back at the getters.  FEAHIP_RENUMBER=0 keeps the caller's numbering
  // (measurement knob: what the kernels make of the ids as given).
  std::vector<int> elements_p, presc_p;
  std::vector<double> nodes_p;
  {
    for (long long i = 0; i < (long long)n_elems * npe; ++i)
      if (elements[i] < 0 || elements[i] >= n_nodes) {
        c->err = "element " + std::to_string(i / npe) + " refers to node " + std::to_string(elements[i]) + " outside [0," + std::to_string(n_nodes) + ")";
        return FEAHIP_EINVAL;
      }
    const char *e = getenv("FEAHIP_RENUMBER");
    if (c->rank_own < 0 && !(e && atoi(e) == 0) && locality_numbering(n_nodes, n_elems, npe, elements, nodes0, c->perm)) {
      bool identity = true;
      for (int a = 0; a < n_nodes && identity; ++a) identity = c->perm[a] == a;
      if (identity) c->perm.clear();
    } else c->perm.clear();
    if (!c->perm.empty()) {
      c->iperm.resize((size_t)n_nodes);
      for (int a = 0; a < n_nodes; ++a) c->iperm[c->perm[a]] = a;
      elements_p.resize((size_t)n_elems * npe);
      for (size_t i = 0; i < elements_p.size(); ++i) elements_p[i] = c->perm[elements[i]];
      nodes_p.resize((size_t)n_nodes * 3);
      for (int a = 0; a < n_nodes; ++a)
        for (int j = 0; j < 3; ++j) nodes_p[(size_t)c->perm[a] * 3 + j] = nodes0[(size_t)a * 3 + j];
      presc_p.resize((size_t)n_presc);
      for (int i = 0; i < n_presc; ++i) {
        if (presc_node[i] < 0 || presc_node[i] >= n_nodes) { c->err = "prescribed node id out of range"; return FEAHIP_EINVAL; }
        presc_p[i] = c->perm[presc_node[i]];
      }
      elements = elements_p.data(); nodes0 = nodes_p.data(); presc_node = presc_p.data();
    }
  }

  memset(&c->table, 0, sizeof(c->table));
  for (int g = 0; g < gauss_count; ++g) {
    c->table.w[g] = gauss_weights[g];
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < npe; ++k) c->table.dN[g][i][k] = dforms[((size_t)g * 3 + i) * npe + k];
  }
  c->linear_tet = (npe == 4);
  for (int g = 0; g < gauss_count && c->linear_tet; ++g)
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 4; ++k) {
        const double want = (k == 0) ? -1.0 : ((k - 1 == i) ? 1.0 : 0.0);
        if (c->table.dN[g][i][k] != want) c->linear_tet = false;
      }
  int rc;
  if ((rc = dev_upload(c, &c->d_table, &c->table, 1))) return rc;

  // pattern + incidence maps (host, once)
  c->h_pat = new HostPattern();
  HostPattern &hp = *c->h_pat;
  if ((rc = build_host_pattern(n_nodes, n_elems, npe, elements, hp, c->err, c->rank_own))) return rc;    // (a rank context: chunks break at its first halo row)
  c->nnzb = (int)hp.colidx.size();
  c->max_rowlen = hp.max_rowlen;
  c->nchunks = (int)hp.chunk.size() - 1;
  c->chunk0 = 0; c->nchunks_local = c->nchunks;
  c->nachunks = hp.achunk.empty() ? 0 : (int)hp.achunk.size() - 1;
  c->achunk0 = 0; c->nachunks_local = c->nachunks;
  c->h_super_achunk = hp.super_achunk;
  c->h_chunk = hp.chunk;
  if (c->rank_own >= 0) {                                   // a rank context: the chunks of the rows it owns
    c->nchunks_local = hp.break_chunk;
    if (!hp.super_achunk.empty()) c->nachunks_local = hp.super_achunk[hp.break_super];
  }
  c->row0 = 0; c->row1 = n_nodes;
  c->ichunk_lo = 0; c->ichunk_hi = c->nchunks;
  c->h_rowptr = hp.rowptr; c->h_colidx = hp.colidx;
  c->incslot_ok = !hp.incslot.empty();

  if ((rc = dev_upload(c, &c->d_conn, elements, (size_t)n_elems * npe))) return rc;
  {
    std::vector<double> pad((size_t)n_nodes * 4, 0.0);
    for (int a = 0; a < n_nodes; ++a)
      for (int j = 0; j < 3; ++j) pad[(size_t)a * 4 + j] = nodes0[(size_t)a * 3 + j];
    if ((rc = dev_upload(c, &c->d_X0, pad.data(), pad.size()))) return rc;
    if ((rc = dev_upload(c, &c->d_x, pad.data(), pad.size()))) return rc;   // nodes_p = copy of nodes0 (:400)
  }
  if ((rc = dev_upload(c, &c->d_rowptr, hp.rowptr.data(), hp.rowptr.size()))) return rc;
  if ((rc = dev_upload(c, &c->d_colidx, hp.colidx.data(), hp.colidx.size()))) return rc;
  if ((rc = dev_upload(c, &c->d_chunk, hp.chunk.data(), hp.chunk.size()))) return rc;
  if ((rc = dev_upload(c, &c->d_diag, hp.diag.data(), hp.diag.size()))) return rc;
  c->aux_bytes = (long long)(hp.incptr.size() * 4 + hp.inc.size() * 4 + hp.incslot.size() +
                             hp.chunk.size() * 4 + hp.rowptr.size() * 4 + hp.diag.size() * 4);

  const bool lin1 = c->linear_tet && gauss_count == 1;
  if (lin1) {
    // the maps of every linear-tet strategy (gather, staged visits, generic incidence lists) are
    // built the first time a launch asks for them, from these host copies -- for the rows this rank owns where
    // the strategy allows (gather)
    c->h_conn.assign(elements, elements + (size_t)n_elems * npe);
  } else if ((rc = ensure_generic_maps(c))) return rc;
  if (npe == 10 || npe == 8) c->h_conn.assign(elements, elements + (size_t)n_elems * npe);      // gather / shared-state maps: built per shard on first use
  if (!lin1 && npe != 10 && npe != 8) { delete c->h_pat; c->h_pat = nullptr; }                   // nothing is built later for these meshes
  if ((rc = dev_zeros(c, &c->d_f, (size_t)c->ndof))) return rc;
  if ((rc = dev_zeros(c, &c->d_u, (size_t)c->ndof))) return rc;
  if ((rc = dev_zeros(c, &c->d_r, (size_t)c->ndof))) return rc;
  if ((rc = dev_zeros(c, &c->d_p, (size_t)c->ndof))) return rc;
  if ((rc = dev_zeros(c, &c->d_q, (size_t)c->ndof))) return rc;
  if ((rc = dev_zeros(c, &c->d_minv, (size_t)c->N * 9))) return rc;
  if ((rc = dev_zeros(c, &c->d_part, (size_t)6 * FEA_RED_BLOCKS))) return rc;
  if ((rc = dev_zeros(c, &c->d_scal, (size_t)16))) return rc;
  if ((rc = dev_zeros(c, &c->d_flag, (size_t)4))) return rc;

  // prescribed dofs in the order solver_apply_bc_general visits them
  // (fea_solver.c:1210-1240): deck order, x then y then z of a node
  std::vector<int> cdof;
  std::vector<double> cval;
  std::vector<uint8_t> mask((size_t)c->ndof, 0);
  for (int i = 0; i < n_presc; ++i) {
    const int node = presc_node[i], type = presc_type[i];
    if (node < 0 || node >= n_nodes) { c->err = "prescribed node id out of range"; return FEAHIP_EINVAL; }
    if (type < 0 || type > 7) { c->err = "prescribed type must be a 3-bit mask"; return FEAHIP_EINVAL; }
    for (int j = 0; j < 3; ++j)
      if (type & (1 << j)) {
        cdof.push_back(node * 3 + j);
        cval.push_back(presc_values[(size_t)i * 3 + j]);
        mask[(size_t)node * 3 + j] = 1;
      }
  }
  c->n_presc = n_presc;
  c->n_cdof = (int)cdof.size();
  if ((rc = dev_upload(c, &c->d_cdof, cdof.data(), cdof.size()))) return rc;
  if ((rc = dev_upload(c, &c->d_cval, cval.data(), cval.size()))) return rc;
  if ((rc = dev_upload(c, &c->d_dofmask, mask.data(), mask.size()))) return rc;
  return FEAHIP_OK;
}

extern "C" int feahip_create(feahip_ctx **out, int device, int n_nodes, int n_elems, int npe,
                             int gauss_count, const double *gauss_weights, const double *dforms,
                             const int *elements, const double *nodes0, int model,
                             const double *model_params, int params_count, int n_presc,
                             const int *presc_node, const int *presc_type,
                             const double *presc_values)
{
  if (!out) { g_create_error = "null output pointer"; return FEAHIP_EINVAL; }
  *out = nullptr;
  feahip_ctx *c = new (std::nothrow) feahip_ctx();
  if (!c) { g_create_error = "out of host memory"; return FEAHIP_ENOMEM; }
  int rc = create_impl(c, device, n_nodes, n_elems, npe, gauss_count, gauss_weights, dforms, elements,
                       nodes0, model, model_params, params_count, n_presc, presc_node, presc_type,
                       presc_values);
  if (rc != FEAHIP_OK) {
    g_create_error = c->err;
    feahip_destroy(c);
    return rc;
  }
  *out = c;
  return FEAHIP_OK;
}

// One rank's context of a sharded run: the sub-mesh of rankmesh.cpp as an ordinary context -- locally indexed, rows
// [0, n_own) owned, the halo plan installed -- so that nothing on a rank is sized by the whole mesh.
extern "C" int feahip_create_rank(feahip_ctx **out, int device, int rank, int nranks, int n_nodes, int n_elems, int npe,
                                  int gauss_count, const double *gauss_weights, const double *dforms,
                                  const int *elements, const double *nodes0, int model,
                                  const double *model_params, int params_count, int n_presc,
                                  const int *presc_node, const int *presc_type, const double *presc_values)
{
  if (!out) { g_create_error = "null output pointer"; return FEAHIP_EINVAL; }
  *out = nullptr;
  if (n_nodes <= 0 || n_elems <= 0 || !elements || !nodes0 || (npe != 4 && npe != 8 && npe != 10) ||
      n_presc < 0 || (n_presc > 0 && (!presc_node || !presc_type || !presc_values))) {
    g_create_error = "feahip_create_rank: null or empty input"; return FEAHIP_EINVAL;
  }
  RankMesh rm;
  int rc = build_rank_mesh(rank, nranks, n_nodes, n_elems, npe, elements, nodes0, n_presc, presc_node, presc_type, presc_values, rm, g_create_error);
  if (rc) return rc;
  if (rm.elem_global.empty() || rm.n_own <= 0) { g_create_error = "this rank owns no node of the mesh (more ranks than slabs)"; return FEAHIP_EINVAL; }
  feahip_ctx *c = new (std::nothrow) feahip_ctx();
  if (!c) { g_create_error = "out of host memory"; return FEAHIP_ENOMEM; }
  c->rank_own = rm.n_own;
  rc = create_impl(c, device, (int)rm.node_global.size(), (int)rm.elem_global.size(), npe, gauss_count, gauss_weights, dforms,
                   rm.elements.data(), rm.nodes0.data(), model, model_params, params_count, (int)rm.presc_node.size(),
                   rm.presc_node.data(), rm.presc_type.data(), rm.presc_values.data());
  if (rc == FEAHIP_OK) {
    rc = install_plan(c, rm.plan);                         // rows [0, n_own), peers, halo lists, interior chunk range
  }
  if (rc != FEAHIP_OK) { g_create_error = c->err; feahip_destroy(c); return rc; }
  c->rank_node_global = rm.node_global; c->rank_elem_global = rm.elem_global; c->rank_n_global = n_nodes;
  *out = c;
  return FEAHIP_OK;
}

extern "C" int feahip_rank_counts(feahip_ctx *c, long long *o)
{
  if (!c || !o || c->rank_own < 0) return FEAHIP_EINVAL;
  o[0] = c->N; o[1] = c->rank_own; o[2] = c->E; o[3] = c->rank_n_global; o[4] = c->nnzb;
  o[5] = (long long)c->h_rowptr[c->rank_own]; o[6] = c->nsend; o[7] = c->nrecv;
  return FEAHIP_OK;
}

extern "C" int feahip_rank_maps(feahip_ctx *c, int *node_global, int *elem_global)
{
  if (!c || c->rank_own < 0) return FEAHIP_EINVAL;
  if (node_global) std::copy(c->rank_node_global.begin(), c->rank_node_global.end(), node_global);
  if (elem_global) std::copy(c->rank_elem_global.begin(), c->rank_elem_global.end(), elem_global);
  return FEAHIP_OK;
}

// Host-only (no device): what rank `rank` of `nranks` would hold -- counts[8] = {local nodes, owned nodes, local
// elements, blocks of the owned rows, blocks of all local rows, peers, rows sent, rows received}; with non-null arrays
// (sized by a first call) the local nodes' caller ids, the local elements' caller indices, and the block rows of the
// OWNED nodes as built from the rank's own elements (rowptr[owned + 1], column = CALLER id of the column node).
extern "C" int feahip_host_rank_mesh(int rank, int nranks, int n_nodes, int n_elems, int npe, const int *elements,
                                     const double *nodes0, long long *counts, int *node_global, int *elem_global,
                                     long long *rowptr, int *colidx)
{
  if (!elements || !nodes0 || !counts || n_nodes <= 0 || n_elems <= 0) return FEAHIP_EINVAL;
  RankMesh rm;
  std::string err;
  int rc = build_rank_mesh(rank, nranks, n_nodes, n_elems, npe, elements, nodes0, 0, nullptr, nullptr, nullptr, rm, err);
  if (rc) return rc;
  HostPattern hp;
  const int nl = (int)rm.node_global.size();
  if ((rc = build_host_pattern(nl, (int)rm.elem_global.size(), npe, rm.elements.data(), hp, err, rm.n_own))) return rc;
  counts[0] = nl; counts[1] = rm.n_own; counts[2] = (long long)rm.elem_global.size();
  counts[3] = hp.rowptr[rm.n_own]; counts[4] = (long long)hp.colidx.size();
  counts[5] = (long long)rm.plan.peer.size(); counts[6] = (long long)rm.plan.send_idx.size(); counts[7] = (long long)rm.plan.recv_idx.size();
  if (node_global) std::copy(rm.node_global.begin(), rm.node_global.end(), node_global);
  if (elem_global) std::copy(rm.elem_global.begin(), rm.elem_global.end(), elem_global);
  if (rowptr && colidx) {
    for (int a = 0; a <= rm.n_own; ++a) rowptr[a] = hp.rowptr[a];
    for (int q = 0; q < hp.rowptr[rm.n_own]; ++q) colidx[q] = rm.node_global[hp.colidx[q]];
  }
  return FEAHIP_OK;
}

// Host-only: the halo plan of one rank's sub-mesh in the CALLER's node ids (counts[3] = {peers, rows sent, rows received}
// by a first call with null lists): what it sends to and receives from every peer, in the order the rows travel.
extern "C" int feahip_host_rank_plan(int rank, int nranks, int n_nodes, int n_elems, int npe, const int *elements,
                                     const double *nodes0, int *counts, int *peers, int *send_off, int *recv_off,
                                     int *send_idx, int *recv_idx)
{
  if (!elements || !nodes0 || !counts || n_nodes <= 0 || n_elems <= 0) return FEAHIP_EINVAL;
  RankMesh rm;
  std::string err;
  int rc = build_rank_mesh(rank, nranks, n_nodes, n_elems, npe, elements, nodes0, 0, nullptr, nullptr, nullptr, rm, err);
  if (rc) return rc;
  const ShardPlan &pl = rm.plan;
  counts[0] = (int)pl.peer.size(); counts[1] = (int)pl.send_idx.size(); counts[2] = (int)pl.recv_idx.size();
  if (peers) std::copy(pl.peer.begin(), pl.peer.end(), peers);
  if (send_off) std::copy(pl.send_off.begin(), pl.send_off.end(), send_off);
  if (recv_off) std::copy(pl.recv_off.begin(), pl.recv_off.end(), recv_off);
  if (send_idx) for (size_t i = 0; i < pl.send_idx.size(); ++i) send_idx[i] = rm.node_global[pl.send_idx[i]];
  if (recv_idx) for (size_t i = 0; i < pl.recv_idx.size(); ++i) recv_idx[i] = rm.node_global[pl.recv_idx[i]];
  return FEAHIP_OK;
}

extern "C" void feahip_destroy(feahip_ctx *c)
{
  if (!c) return;
  delete c->h_pat; c->h_pat = nullptr;
  delete c->gather_lay; c->gather_lay = nullptr;
  delete c->gather10_lay; c->gather10_lay = nullptr;
  void *ptrs[] = {(void *)c->d_gmaps, (void *)c->d_g10_elist, (void *)c->d_g10_state, c->d_table, c->d_conn, c->d_X0, c->d_x, c->d_rowptr, c->d_colidx, c->d_K_alloc, c->d_Kstash_alloc,
                  c->d_incptr, c->d_inc, c->d_incslot, c->d_chunk, c->d_diag, c->d_vdesc, c->d_vnode, c->d_vrec, c->d_qdesc, c->d_qelem, c->d_qpair, c->d_qnode, c->d_f, c->d_u, c->d_r, c->d_p,
                  c->d_q, c->d_minv, c->d_part, c->d_scal, c->d_flag, c->d_cdof, c->d_cval,
                  c->d_dofmask, c->d_F, c->d_S};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  for (void *p : {(void *)c->d_send_idx, (void *)c->d_recv_idx, (void *)c->d_send_buf, (void *)c->d_recv_buf, (void *)c->d_z, (void *)c->d_w, (void *)c->d_s})
    if (p) (void)hipFree(p);
  if (c->ev_packed) (void)hipEventDestroy(c->ev_packed);
  if (c->ev_unpacked) (void)hipEventDestroy(c->ev_unpacked);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  if (c->tr && c->owns_tr) delete c->tr;
  amg_destroy(c);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

#define CTX_GUARD_NOK(c)                          \
  if (!(c)) return FEAHIP_EINVAL;                 \
  FEA_HIP_CHECK(c, hipSetDevice((c)->device))
// entry points that read or write K make sure it exists for the shard installed now
#define CTX_GUARD(c)                              \
  CTX_GUARD_NOK(c);                               \
  { const int _rk = ensure_k(c); if (_rk) return _rk; }

extern "C" int feahip_sync(feahip_ctx *c)
{
  CTX_GUARD_NOK(c);
  FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return FEAHIP_OK;
}

extern "C" int feahip_set_assembly(feahip_ctx *c, int strategy)
{
  CTX_GUARD_NOK(c);
  if (strategy < FEAHIP_ASM_AUTO || strategy > FEAHIP_ASM_GATHER) { c->err = "unknown assembly strategy"; return FEAHIP_EINVAL; }
  c->strategy = strategy;
  return FEAHIP_OK;
}

extern "C" int feahip_set_preconditioner(feahip_ctx *c, int kind)
{
  CTX_GUARD(c);
  if (kind != 0 && kind != 1) { c->err = "unknown preconditioner"; return FEAHIP_EINVAL; }
  if (kind == 1) { int rc = amg_create(c); if (rc) return rc; }
  c->precond = kind;
  return FEAHIP_OK;
}

extern "C" int feahip_set_pcg_variant(feahip_ctx *c, int variant)
{
  CTX_GUARD_NOK(c);
  if (variant < -1 || variant > 1) { c->err = "unknown PCG variant"; return FEAHIP_EINVAL; }
  c->pcg_variant = variant;
  return FEAHIP_OK;
}

extern "C" int feahip_set_line_search(feahip_ctx *c, int max_iterations)
{
  CTX_GUARD_NOK(c);
  if (max_iterations < 0) { c->err = "line search iterations must be >= 0"; return FEAHIP_EINVAL; }
  c->linesearch_max = max_iterations;
  return FEAHIP_OK;
}

extern "C" int feahip_set_row_shard(feahip_ctx *c, int rank, int nranks)
{
  CTX_GUARD_NOK(c);
  return install_shard(c, rank, nranks);
}

extern "C" int feahip_update_nodes_with_bc(feahip_ctx *c, double lambda)
{
  CTX_GUARD_NOK(c);
  c->state_valid = false;
  return launch_update_nodes_bc(c, lambda);
}

extern "C" int feahip_update_state(feahip_ctx *c, int *n_bad)
{
  CTX_GUARD_NOK(c);
  c->state_valid = false;
  if (n_bad) {
    FEA_HIP_CHECK(c, hipMemcpyAsync(&c->last_bad, c->d_flag + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    *n_bad = c->last_bad;
  }
  return FEAHIP_OK;
}

extern "C" int feahip_create_stiffness(feahip_ctx *c) { CTX_GUARD(c); ++c->k_epoch; c->k_bc = false; return launch_assemble(c, true, false); }
extern "C" int feahip_create_residual_forces(feahip_ctx *c) { CTX_GUARD(c); return launch_assemble(c, false, true); }
extern "C" int feahip_create_stiffness_and_residual(feahip_ctx *c) { CTX_GUARD(c); ++c->k_epoch; c->k_bc = false; return launch_assemble(c, true, true); }

extern "C" int feahip_stash_stiffness(feahip_ctx *c)
{
  CTX_GUARD(c);
  const size_t bytes = sizeof(double) * 9 * (size_t)(c->kb1 - c->kb0);
  if (!c->d_Kstash_base) {
    FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_Kstash_alloc, bytes + 8));
    c->d_Kstash_base = c->d_Kstash_alloc + (c->kb0 & 1);      // same 16-byte phase as K
    c->d_Kstash = c->d_Kstash_base - (size_t)c->kb0 * 9;
  }
  FEA_HIP_CHECK(c, hipMemcpyAsync(c->d_Kstash_base, c->d_K_base, bytes, hipMemcpyDeviceToDevice, c->stream));
  c->have_stash = true;
  c->stash_epoch = c->k_epoch;
  return FEAHIP_OK;
}

extern "C" int feahip_restore_stiffness(feahip_ctx *c)
{
  CTX_GUARD(c);
  if (!c->have_stash) { c->err = "restore_stiffness before stash_stiffness"; return FEAHIP_ESTATE; }
  FEA_HIP_CHECK(c, hipMemcpyAsync(c->d_K_base, c->d_Kstash_base, sizeof(double) * 9 * (size_t)(c->kb1 - c->kb0),
                                  hipMemcpyDeviceToDevice, c->stream));
  c->k_epoch = c->stash_epoch; c->k_bc = false;
  return FEAHIP_OK;
}

extern "C" int feahip_apply_prescribed_bc(feahip_ctx *c, double lambda) { CTX_GUARD(c); c->k_bc = true; return launch_apply_bc(c, lambda); }

extern "C" int feahip_solve_slae(feahip_ctx *c, int type, double tol, int max_iter, int *iters, double *resid)
{
  CTX_GUARD(c);
  if (type < FEAHIP_CG || type > FEAHIP_CHOLESKY) { c->err = "unknown solver type"; return FEAHIP_EINVAL; }
  if (max_iter <= 0) { c->err = "max_iterations must be positive"; return FEAHIP_EINVAL; }
  return solve_pcg(c, type, tol, max_iter, iters, resid);    // block-Jacobi or multigrid by the context's setting
}

extern "C" int feahip_energy(feahip_ctx *c, double *tolerance)
{
  CTX_GUARD(c);
  if (!tolerance) return FEAHIP_EINVAL;
  std::vector<feahip_ctx *> R(1, c);
  return dist_energy(R, tolerance);
}

extern "C" int feahip_update_nodes_with_solution(feahip_ctx *c, const double *u)
{
  CTX_GUARD(c);
  std::vector<feahip_ctx *> R(1, c);
  std::vector<double> tmp;
  if (u && !c->perm.empty()) {
    tmp.resize((size_t)c->ndof);
    for (int a = 0; a < c->N; ++a)
      for (int j = 0; j < 3; ++j) tmp[(size_t)c->perm[a] * 3 + j] = u[(size_t)a * 3 + j];
    u = tmp.data();
  }
  return dist_update_nodes_with_solution(R, u);
}

extern "C" int feahip_solve(feahip_ctx *c, int load_increments, int max_newton, int modified_newton,
                            double desired_tolerance, int solver_type, double solver_tolerance,
                            int solver_max_iter, double *tol_log, int tol_log_cap, int *its_log,
                            int *steps_done)
{
  CTX_GUARD(c);
  std::vector<feahip_ctx *> R(1, c);
  return dist_newton(R, load_increments, max_newton, modified_newton, desired_tolerance, solver_type,
                     solver_tolerance, solver_max_iter, tol_log, tol_log_cap, its_log, steps_done);
}

// ---- sharding ------------------------------------------------------------

static void drop_transport(feahip_ctx *c)
{
  if (c->tr && c->owns_tr) delete c->tr;
  c->tr = nullptr; c->owns_tr = false;
}

extern "C" int feahip_comm_unique_id(void *out, int cap) { return rccl_unique_id(out, cap); }

extern "C" int feahip_comm_init(feahip_ctx *c, int rank, int nranks, const void *unique_id)
{
  CTX_GUARD_NOK(c);
  if (!unique_id) return FEAHIP_EINVAL;
  int rc = install_shard(c, rank, nranks);
  if (rc) return rc;
  drop_transport(c);
  c->tr = make_rccl_transport(c, rank, nranks, unique_id, c->err);
  if (!c->tr) return FEAHIP_ECOMM;
  c->owns_tr = true;
  return FEAHIP_OK;
}

extern "C" int feahip_group_init(feahip_ctx **ctxs, int n)
{
  if (!ctxs || n < 1) return FEAHIP_EINVAL;
  Transport *t = make_group_transport();
  for (int r = 0; r < n; ++r) {
    if (!ctxs[r]) { delete t; return FEAHIP_EINVAL; }
    (void)hipSetDevice(ctxs[r]->device);
    int rc = install_shard(ctxs[r], r, n);
    if (rc) { delete t; return rc; }
    drop_transport(ctxs[r]);
    ctxs[r]->tr = t;
    ctxs[r]->owns_tr = (r == 0);
  }
  return FEAHIP_OK;
}

static int group_vec(feahip_ctx **ctxs, int n, std::vector<feahip_ctx *> &R)
{
  if (!ctxs || n < 1) return FEAHIP_EINVAL;
  R.assign(ctxs, ctxs + n);
  for (int r = 0; r < n; ++r)
    if (!R[r] || R[r]->nranks != n || R[r]->rank != r || !R[r]->tr) {
      if (R[0]) R[0]->err = "not a group: call feahip_group_init on these contexts first";
      return FEAHIP_ESTATE;
    }
  return FEAHIP_OK;
}

extern "C" int feahip_group_solve_slae(feahip_ctx **ctxs, int n, int type, double tol, int max_iter, int *iters, double *resid)
{
  std::vector<feahip_ctx *> R;
  int rc = group_vec(ctxs, n, R);
  if (rc) return rc;
  return dist_solve_pcg(R, type, tol, max_iter, iters, resid);
}

extern "C" int feahip_group_energy(feahip_ctx **ctxs, int n, double *tolerance)
{
  std::vector<feahip_ctx *> R;
  int rc = group_vec(ctxs, n, R);
  if (rc) return rc;
  return dist_energy(R, tolerance);
}

extern "C" int feahip_group_update_nodes_with_solution(feahip_ctx **ctxs, int n)
{
  std::vector<feahip_ctx *> R;
  int rc = group_vec(ctxs, n, R);
  if (rc) return rc;
  return dist_update_nodes_with_solution(R, nullptr);
}

extern "C" int feahip_group_solve(feahip_ctx **ctxs, int n, int load_increments, int max_newton, int modified_newton,
                                  double desired_tolerance, int solver_type, double solver_tolerance,
                                  int solver_max_iter, double *tol_log, int tol_log_cap, int *its_log, int *steps_done)
{
  std::vector<feahip_ctx *> R;
  int rc = group_vec(ctxs, n, R);
  if (rc) return rc;
  return dist_newton(R, load_increments, max_newton, modified_newton, desired_tolerance, solver_type,
                     solver_tolerance, solver_max_iter, tol_log, tol_log_cap, its_log, steps_done);
}

extern "C" int feahip_owned_rows(feahip_ctx *c, int *row0, int *row1)
{
  if (!c || !row0 || !row1) return FEAHIP_EINVAL;
  *row0 = c->row0; *row1 = c->row1;
  return FEAHIP_OK;
}

// Host-only: the halo plan of one rank, from the element->node map alone (no
// device is touched).  Lists are written into caller buffers sized by a first
// call with null lists: counts[0] = npeers, [1] = total send, [2] = total recv,
// [3] = row0, [4] = row1.
extern "C" int feahip_shard_plan(int n_nodes, int n_elems, int npe, const int *elements, int rank, int nranks,
                                 int *counts, int *peers, int *send_off, int *recv_off, int *send_idx, int *recv_idx)
{
  if (!elements || !counts || n_nodes <= 0 || n_elems <= 0 || nranks < 1 || rank < 0 || rank >= nranks) return FEAHIP_EINVAL;
  HostPattern hp;
  std::string err;
  int rc = build_host_pattern(n_nodes, n_elems, npe, elements, hp, err);
  if (rc) return rc;
  ShardPlan plan;
  build_shard_plan(hp.rowptr, hp.colidx, hp.chunk, rank, nranks, plan);
  counts[0] = (int)plan.peer.size(); counts[1] = (int)plan.send_idx.size(); counts[2] = (int)plan.recv_idx.size();
  counts[3] = plan.row0; counts[4] = plan.row1;
  if (peers) std::copy(plan.peer.begin(), plan.peer.end(), peers);
  if (send_off) std::copy(plan.send_off.begin(), plan.send_off.end(), send_off);
  if (recv_off) std::copy(plan.recv_off.begin(), plan.recv_off.end(), recv_off);
  if (send_idx) std::copy(plan.send_idx.begin(), plan.send_idx.end(), send_idx);
  if (recv_idx) std::copy(plan.recv_idx.begin(), plan.recv_idx.end(), recv_idx);
  return FEAHIP_OK;
}

extern "C" int feahip_host_numbering(int n_nodes, int n_elems, int npe, const int *elements, const double *nodes0, int *library_id_of_node)
{
  if (!elements || !nodes0 || !library_id_of_node || n_nodes <= 0 || n_elems <= 0 || (npe != 4 && npe != 8 && npe != 10)) return FEAHIP_EINVAL;
  for (long long i = 0; i < (long long)n_elems * npe; ++i)
    if (elements[i] < 0 || elements[i] >= n_nodes) return FEAHIP_EINVAL;
  std::vector<int> perm;
  const bool any = locality_numbering(n_nodes, n_elems, npe, elements, nodes0, perm);
  bool identity = true;
  for (int a = 0; a < n_nodes; ++a) { library_id_of_node[a] = perm[a]; identity = identity && perm[a] == a; }
  return any && !identity ? 1 : 0;
}

// ---- views ---------------------------------------------------------------

extern "C" int feahip_set_nodes(feahip_ctx *c, const double *nodes)
{
  CTX_GUARD_NOK(c);
  if (!nodes) return FEAHIP_EINVAL;
  std::vector<double> pad((size_t)c->N * 4, 0.0);
  for (int a = 0; a < c->N; ++a)
    for (int j = 0; j < 3; ++j) pad[(size_t)lib_id(c, a) * 4 + j] = nodes[(size_t)a * 3 + j];
  FEA_HIP_CHECK(c, hipMemcpyAsync(c->d_x, pad.data(), sizeof(double) * pad.size(), hipMemcpyHostToDevice, c->stream));
  FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  c->state_valid = false;
  return FEAHIP_OK;
}

extern "C" int feahip_get_nodes(feahip_ctx *c, double *nodes)
{
  CTX_GUARD_NOK(c);
  if (!nodes) return FEAHIP_EINVAL;
  std::vector<double> pad((size_t)c->N * 4);
  FEA_HIP_CHECK(c, hipMemcpyAsync(pad.data(), c->d_x, sizeof(double) * pad.size(), hipMemcpyDeviceToHost, c->stream));
  FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  for (int a = 0; a < c->N; ++a)
    for (int j = 0; j < 3; ++j) nodes[(size_t)a * 3 + j] = pad[(size_t)lib_id(c, a) * 4 + j];
  return FEAHIP_OK;
}

static int get_vec(feahip_ctx *c, const double *d, double *h, size_t n)
{
  if (!h) return FEAHIP_EINVAL;
  FEA_HIP_CHECK(c, hipMemcpyAsync(h, d, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return FEAHIP_OK;
}

// node vectors (3 doubles per node) between the caller's numbering (host) and the library's (device)
static int get_node_vec(feahip_ctx *c, const double *d, double *h)
{
  if (c->perm.empty()) return get_vec(c, d, h, (size_t)c->ndof);
  if (!h) return FEAHIP_EINVAL;
  std::vector<double> tmp((size_t)c->ndof);
  const int rc = get_vec(c, d, tmp.data(), tmp.size());
  if (rc) return rc;
  for (int a = 0; a < c->N; ++a)
    for (int j = 0; j < 3; ++j) h[(size_t)a * 3 + j] = tmp[(size_t)c->perm[a] * 3 + j];
  return FEAHIP_OK;
}

static int set_node_vec(feahip_ctx *c, double *d, const double *h)
{
  if (!h) return FEAHIP_EINVAL;
  std::vector<double> tmp;
  if (!c->perm.empty()) {
    tmp.resize((size_t)c->ndof);
    for (int a = 0; a < c->N; ++a)
      for (int j = 0; j < 3; ++j) tmp[(size_t)c->perm[a] * 3 + j] = h[(size_t)a * 3 + j];
    h = tmp.data();
  }
  FEA_HIP_CHECK(c, hipMemcpyAsync(d, h, sizeof(double) * (size_t)c->ndof, hipMemcpyHostToDevice, c->stream));
  FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return FEAHIP_OK;
}

extern "C" int feahip_get_forces(feahip_ctx *c, double *f) { CTX_GUARD(c); return get_node_vec(c, c->d_f, f); }
extern "C" int feahip_get_solution(feahip_ctx *c, double *u) { CTX_GUARD(c); return get_node_vec(c, c->d_u, u); }
extern "C" int feahip_set_forces(feahip_ctx *c, const double *f) { CTX_GUARD(c); return set_node_vec(c, c->d_f, f); }

extern "C" int feahip_node_numbering(feahip_ctx *c, int *library_id_of_node)
{
  if (!c || !library_id_of_node) return FEAHIP_EINVAL;
  for (int a = 0; a < c->N; ++a) library_id_of_node[a] = lib_id(c, a);
  return FEAHIP_OK;
}

static int ensure_state(feahip_ctx *c)
{
  const size_t n = (size_t)c->E * c->G * 9;
  if (!c->d_F) {
    FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_F, sizeof(double) * n));
    FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_S, sizeof(double) * n));
  }
  if (!c->state_valid) {
    int rc = launch_state_export(c);
    if (rc) return rc;
    c->state_valid = true;
  }
  return FEAHIP_OK;
}

extern "C" int feahip_get_graddefs(feahip_ctx *c, double *F)
{
  CTX_GUARD(c);
  int rc = ensure_state(c);
  if (rc) return rc;
  return get_vec(c, c->d_F, F, (size_t)c->E * c->G * 9);
}

extern "C" int feahip_get_stresses(feahip_ctx *c, double *S)
{
  CTX_GUARD(c);
  int rc = ensure_state(c);
  if (rc) return rc;
  return get_vec(c, c->d_S, S, (size_t)c->E * c->G * 9);
}

extern "C" int feahip_get_shape_gradients(feahip_ctx *c, double *grads, double *detj)
{
  CTX_GUARD(c);
  if (!grads || !detj) return FEAHIP_EINVAL;
  int rc = ensure_state(c);                                  // allocates F / sigma (the kernel writes them too)
  if (rc) return rc;
  const size_t ng = (size_t)c->E * c->G * 3 * c->npe, nd = (size_t)c->E * c->G;
  double *dg = nullptr, *dd = nullptr;
  FEA_HIP_CHECK(c, hipMalloc((void **)&dg, sizeof(double) * ng));
  if (hipMalloc((void **)&dd, sizeof(double) * nd) != hipSuccess) { (void)hipFree(dg); c->err = "out of device memory"; return FEAHIP_ENOMEM; }
  rc = launch_state_export(c, dg, dd);
  if (!rc) rc = get_vec(c, dg, grads, ng);
  if (!rc) rc = get_vec(c, dd, detj, nd);
  (void)hipFree(dg); (void)hipFree(dd);
  return rc;
}

extern "C" int feahip_matrix_nnz(feahip_ctx *c, long long *nnz)
{
  if (!c || !nnz) return FEAHIP_EINVAL;
  *nnz = (long long)c->nnzb * 9;
  return FEAHIP_OK;
}

template <class OFF>
static int matrix_yale(feahip_ctx *c, OFF *offsets, int *indexes, double *values)
{
  std::vector<double> K((size_t)c->nnzb * 9, 0.0);               // rows of other ranks read as zero
  int rc = get_vec(c, c->d_K_base, K.data() + (size_t)c->kb0 * 9, (size_t)(c->kb1 - c->kb0) * 9);
  if (rc) return rc;
  size_t pos = 0;
  offsets[0] = 0;
  std::vector<std::pair<int, int>> row;                           // (caller's column node, block) of one row, sorted by column
  for (int a = 0; a < c->N; ++a) {                                // a: the CALLER's node; its row lives at the library id
    const int la = lib_id(c, a);
    row.clear();
    for (int q = c->h_rowptr[la]; q < c->h_rowptr[la + 1]; ++q)
      row.emplace_back(c->iperm.empty() ? c->h_colidx[q] : c->iperm[c->h_colidx[q]], q);
    if (!c->iperm.empty()) std::sort(row.begin(), row.end());
    for (int i = 0; i < 3; ++i) {
      for (const auto &cb : row)
        for (int j = 0; j < 3; ++j) {
          indexes[pos] = 3 * cb.first + j;
          values[pos] = K[(size_t)cb.second * 9 + 3 * i + j];
          pos++;
        }
      offsets[3 * a + i + 1] = (OFF)pos;
    }
  }
  return FEAHIP_OK;
}

extern "C" int feahip_get_matrix_yale(feahip_ctx *c, int *offsets, int *indexes, double *values)
{
  CTX_GUARD(c);
  if (!offsets || !indexes || !values) return FEAHIP_EINVAL;
  if ((long long)c->nnzb * 9 > 0x7FFFFFFFLL) {                   // sp_matrix_yale keeps int offsets: refused, not wrapped
    c->err = "matrix too large for 32-bit Yale offsets (use feahip_get_matrix_yale64)";
    return FEAHIP_EINVAL;
  }
  return matrix_yale<int>(c, offsets, indexes, values);
}

extern "C" int feahip_get_matrix_yale64(feahip_ctx *c, long long *offsets, int *indexes, double *values)
{
  CTX_GUARD(c);
  if (!offsets || !indexes || !values) return FEAHIP_EINVAL;
  if ((long long)c->N * 3 > 0x7FFFFFFFLL) { c->err = "more than 2^31 dofs: column indexes do not fit"; return FEAHIP_EINVAL; }
  return matrix_yale<long long>(c, offsets, indexes, values);
}

extern "C" int feahip_spmv(feahip_ctx *c, const double *x, double *y)
{
  CTX_GUARD(c);
  if (!x || !y) return FEAHIP_EINVAL;
  int rc = set_node_vec(c, c->d_p, x);
  if (rc) return rc;
  rc = launch_spmv(c, c->d_p, c->d_q);
  if (rc) return rc;
  return get_node_vec(c, c->d_q, y);
}

extern "C" int feahip_time_kernel(feahip_ctx *c, int what, int warmup, int iters, double *avg_ms)
{
  CTX_GUARD(c);
  if (!avg_ms || iters <= 0 || warmup < 0) return FEAHIP_EINVAL;
  if (what == 4) return time_pcg_iteration(c, warmup, iters, avg_ms);
  auto one = [&]() -> int {
    switch (what) {
    case 0: return launch_assemble(c, true, true);
    case 1: return launch_assemble(c, true, false);
    case 2: return launch_assemble(c, false, true);
    case 3: return launch_spmv(c, c->d_p, c->d_q);
    default: c->err = "unknown kernel selector"; return FEAHIP_EINVAL;
    }
  };
  int rc;
  for (int k = 0; k < warmup; ++k) if ((rc = one())) return rc;
  hipEvent_t e0, e1;
  FEA_HIP_CHECK(c, hipEventCreate(&e0));
  FEA_HIP_CHECK(c, hipEventCreate(&e1));
  FEA_HIP_CHECK(c, hipEventRecord(e0, c->stream));
  for (int k = 0; k < iters; ++k) if ((rc = one())) return rc;
  FEA_HIP_CHECK(c, hipEventRecord(e1, c->stream));
  FEA_HIP_CHECK(c, hipEventSynchronize(e1));
  float ms = 0;
  FEA_HIP_CHECK(c, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *avg_ms = (double)ms / iters;
  return FEAHIP_OK;
}

// Streaming copies: the copy bandwidth of THIS box, the figure the roofline fractions can be quoted against next to the
// 8 TB/s of the data sheet (MI355X_MICROARCH.md measures 6.29 TB/s with a float4 copy).  Four ways, the
// best one is reported (feahip_copy_bandwidth) and all four are available (feahip_copy_bandwidth_detail):
//   [0] one 16-byte load in flight per lane, grid-stride (rounds 1-3);
//   [1] FOUR independent 16-byte loads in flight per lane, then their four stores -- each workgroup streams a
//       contiguous 16 KB tile per step, 16 workgroups per CU;
//   [2] hipMemcpyDtoDAsync (the runtime's blit kernel);
//   [3] as [1] with non-temporal loads and stores.
__global__ __launch_bounds__(256)
void k_copy16(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

typedef double copy_v2d __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(256)
void k_copy16x4(const copy_v2d *__restrict__ src, copy_v2d *__restrict__ dst, size_t n)
{
  // tile = 4 x 256 pieces of 16 bytes; tiles dealt round-robin to the workgroups
  const size_t ntiles = n >> 10;
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t i = (tile << 10) + threadIdx.x;
    copy_v2d a, b, c, d;
    if (NT) { a = __builtin_nontemporal_load(src + i); b = __builtin_nontemporal_load(src + i + 256);
              c = __builtin_nontemporal_load(src + i + 512); d = __builtin_nontemporal_load(src + i + 768); }
    else { a = src[i]; b = src[i + 256]; c = src[i + 512]; d = src[i + 768]; }
    if (NT) { __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + 256);
              __builtin_nontemporal_store(c, dst + i + 512); __builtin_nontemporal_store(d, dst + i + 768); }
    else { dst[i] = a; dst[i + 256] = b; dst[i + 512] = c; dst[i + 768] = d; }
  }
  for (size_t i = (ntiles << 10) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

extern "C" int feahip_copy_bandwidth_detail(feahip_ctx *c, long long bytes, double *gbytes_per_s4)
{
  CTX_GUARD_NOK(c);
  if (!gbytes_per_s4 || bytes < (1 << 20)) return FEAHIP_EINVAL;
  const size_t n = (size_t)bytes / 16;
  double2 *a = nullptr, *b = nullptr;
  FEA_HIP_CHECK(c, hipMalloc((void **)&a, n * 16));
  if (hipMalloc((void **)&b, n * 16) != hipSuccess) { (void)hipFree(a); c->err = "out of device memory"; return FEAHIP_ENOMEM; }
  (void)hipMemsetAsync(a, 0x3c, n * 16, c->stream);
  const int grid = 256 * 16;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipError_t err = hipSuccess;
  const int reps = 10;
  for (int m = 0; m < 4; ++m) {
    auto one = [&]() {
      if (m == 0) hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, c->stream, a, b, n);
      else if (m == 1) hipLaunchKernelGGL(k_copy16x4<false>, dim3(grid), dim3(256), 0, c->stream, (const copy_v2d *)a, (copy_v2d *)b, n);
      else if (m == 3) hipLaunchKernelGGL(k_copy16x4<true>, dim3(grid), dim3(256), 0, c->stream, (const copy_v2d *)a, (copy_v2d *)b, n);
      else (void)hipMemcpyDtoDAsync((hipDeviceptr_t)b, (hipDeviceptr_t)a, n * 16, c->stream);
    };
    for (int k = 0; k < 3; ++k) one();
    (void)hipEventRecord(e0, c->stream);
    for (int k = 0; k < reps; ++k) one();
    (void)hipEventRecord(e1, c->stream);
    const hipError_t e = hipEventSynchronize(e1);
    if (e != hipSuccess) err = e;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    gbytes_per_s4[m] = ms > 0 ? 2.0 * (double)(n * 16) * reps / ((double)ms * 1e-3) / 1e9 : 0.0;      // read + written
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b);
  if (err != hipSuccess || !(gbytes_per_s4[0] > 0)) { c->err = "copy kernel failed"; return FEAHIP_EHIP; }
  return FEAHIP_OK;
}

extern "C" int feahip_copy_bandwidth(feahip_ctx *c, long long bytes, double *gbytes_per_s)
{
  double v[4] = {0, 0, 0, 0};
  if (!gbytes_per_s) return FEAHIP_EINVAL;
  const int rc = feahip_copy_bandwidth_detail(c, bytes, v);
  if (rc) return rc;
  *gbytes_per_s = std::max(std::max(v[0], v[1]), std::max(v[2], v[3]));
  return FEAHIP_OK;
}

extern "C" int feahip_assembly_stats(feahip_ctx *c, double *o)
{
  if (!c || !o) return FEAHIP_EINVAL;
  o[0] = c->have_gather ? c->gather_evals_per_element : 0.0;
  o[1] = c->have_gather ? (double)c->ngchunks : 0.0;
  o[2] = c->have_gather ? (double)c->gather_same_words : 0.0;
  o[3] = c->have_gather ? (double)c->gather_bytes : 0.0;
  return FEAHIP_OK;
}

extern "C" int feahip_assembly_in_use(feahip_ctx *c, int *strategy)
{
  if (!c || !strategy) return FEAHIP_EINVAL;
  *strategy = c->last_strategy;
  return FEAHIP_OK;
}

extern "C" int feahip_device_layout(feahip_ctx *c, long long *o)
{
  if (!c || !o) return FEAHIP_EINVAL;
  { const int rc = ensure_k(c); if (rc) return rc; }
  o[0] = (long long)(size_t)c->d_K_base; o[1] = (long long)(size_t)c->d_colidx; o[2] = (long long)(size_t)c->d_p; o[3] = (long long)(size_t)c->d_q;
  return FEAHIP_OK;
}

extern "C" int feahip_sizes(feahip_ctx *c, long long *o)
{
  if (!c || !o) return FEAHIP_EINVAL;
  o[0] = c->N; o[1] = c->E; o[2] = c->npe; o[3] = c->G; o[4] = c->nnzb; o[5] = c->nchunks;
  // bytes of the maps the default assembly kernel reads besides the algorithmic inputs
  o[6] = c->have_gather ? c->gather_bytes
       : c->have_visits ? c->visit_bytes + (long long)(c->N + 1) * 8
       : c->have_quad ? c->quad_bytes + (long long)(c->N + 1) * 8 : c->aux_bytes;
  o[7] = c->max_rowlen;
  return FEAHIP_OK;
}
