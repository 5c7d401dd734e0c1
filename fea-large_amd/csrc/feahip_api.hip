// feahip_api.hip -- extern "C" entry points of include/fea_hip.h.
#include "feahip_internal.h"
#include <cmath>
#include <cstring>

static std::string g_create_error;

extern "C" const char *feahip_create_error(void) { return g_create_error.c_str(); }
extern "C" const char *feahip_last_error(const feahip_ctx *c) { return c ? c->err.c_str() : "null context"; }

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

static int create_impl(feahip_ctx *c, int device, int n_nodes, int n_elems, int npe, int gauss_count,
                       const double *gauss_weights, const double *dforms, const int *elements,
                       const double *nodes0, int model, const double *model_params,
                       int params_count, int n_presc, const int *presc_node,
                       const int *presc_type, const double *presc_values)
{
  if (n_nodes <= 0 || n_elems <= 0 || !gauss_weights || !dforms || !elements || !nodes0 || !model_params) {
    c->err = "feahip_create: null or empty input"; return FEAHIP_EINVAL;
  }
  if (npe != 4 && npe != 10) { c->err = "nodes per element must be 4 or 10"; return FEAHIP_EINVAL; }
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
  HostPattern hp;
  if ((rc = build_host_pattern(n_nodes, n_elems, npe, elements, hp, c->err))) return rc;
  c->nnzb = (int)hp.colidx.size();
  c->max_rowlen = hp.max_rowlen;
  c->nchunks = (int)hp.chunk.size() - 1;
  c->chunk0 = 0; c->nchunks_local = c->nchunks;
  c->nachunks = hp.achunk.empty() ? 0 : (int)hp.achunk.size() - 1;
  c->achunk0 = 0; c->nachunks_local = c->nachunks;
  c->h_super_achunk = hp.super_achunk;
  c->h_rowptr = hp.rowptr; c->h_colidx = hp.colidx;

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
  if ((rc = dev_upload(c, &c->d_incptr, hp.incptr.data(), hp.incptr.size()))) return rc;
  if ((rc = dev_upload(c, &c->d_inc, hp.inc.data(), hp.inc.size()))) return rc;
  if (!hp.incslot.empty()) {
    if ((rc = dev_upload(c, &c->d_incslot, hp.incslot.data(), hp.incslot.size()))) return rc;
  }
  if ((rc = dev_upload(c, &c->d_chunk, hp.chunk.data(), hp.chunk.size()))) return rc;
  if ((rc = dev_upload(c, &c->d_diag, hp.diag.data(), hp.diag.size()))) return rc;
  c->aux_bytes = (long long)(hp.incptr.size() * 4 + hp.inc.size() * 4 + hp.incslot.size() +
                             hp.chunk.size() * 4 + hp.rowptr.size() * 4 + hp.diag.size() * 4);

  if (c->linear_tet && gauss_count == 1) {
    HostPatches pt;
    build_host_patches(n_nodes, n_elems, elements, hp, pt);
    if (pt.ok) {
      if ((rc = dev_upload(c, &c->d_pdesc, pt.desc.data(), pt.desc.size()))) return rc;
      if ((rc = dev_upload(c, &c->d_pnode, pt.pnode.data(), pt.pnode.size()))) return rc;
      if ((rc = dev_upload(c, &c->d_pelem, pt.pelem.data(), pt.pelem.size()))) return rc;
      if ((rc = dev_upload(c, &c->d_pent, pt.pent.data(), pt.pent.size()))) return rc;
      if ((rc = dev_upload(c, &c->d_pbptr, pt.pbptr.data(), pt.pbptr.size()))) return rc;
      c->have_patches = true;
      c->patch_bytes = (long long)(pt.desc.size() * sizeof(PatchDesc) + pt.pnode.size() * 4 +
                                   pt.pelem.size() * 2 + pt.pent.size() * 2 + pt.pbptr.size() * 2);
    }
  }
  if (c->linear_tet && gauss_count == 1) {
    HostVisits hv;
    build_host_visits(n_nodes, n_elems, elements, hp, hv);
    if (hv.ok) {
      if ((rc = dev_upload(c, &c->d_vdesc, hv.desc.data(), hv.desc.size()))) return rc;
      if ((rc = dev_upload(c, &c->d_vnode, hv.vnode.data(), hv.vnode.size()))) return rc;
      if ((rc = dev_upload(c, &c->d_vrec, hv.vrec.data(), hv.vrec.size()))) return rc;
      c->have_visits = true;
      c->visit_bytes = (long long)(hv.desc.size() * sizeof(VisitDesc) + hv.vnode.size() * 4 + hv.vrec.size() * 4);
    }
  }
  if ((rc = dev_zeros(c, &c->d_K, (size_t)c->nnzb * 9))) return rc;
  if ((rc = dev_zeros(c, &c->d_f, (size_t)c->ndof))) return rc;
  if ((rc = dev_zeros(c, &c->d_u, (size_t)c->ndof))) return rc;
  if ((rc = dev_zeros(c, &c->d_r, (size_t)c->ndof))) return rc;
  if ((rc = dev_zeros(c, &c->d_p, (size_t)c->ndof))) return rc;
  if ((rc = dev_zeros(c, &c->d_q, (size_t)c->ndof))) return rc;
  if ((rc = dev_zeros(c, &c->d_minv, (size_t)c->N * 9))) return rc;
  if ((rc = dev_zeros(c, &c->d_part, (size_t)4 * FEA_RED_BLOCKS))) return rc;
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

extern "C" void feahip_destroy(feahip_ctx *c)
{
  if (!c) return;
  void *ptrs[] = {c->d_table, c->d_conn, c->d_X0, c->d_x, c->d_rowptr, c->d_colidx, c->d_K, c->d_Kstash,
                  c->d_incptr, c->d_inc, c->d_incslot, c->d_chunk, c->d_diag, c->d_pdesc, c->d_pnode, c->d_pelem, c->d_pent, c->d_pbptr, c->d_vdesc, c->d_vnode, c->d_vrec, c->d_f, c->d_u, c->d_r, c->d_p,
                  c->d_q, c->d_minv, c->d_part, c->d_scal, c->d_flag, c->d_cdof, c->d_cval,
                  c->d_dofmask, c->d_F, c->d_S};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

#define CTX_GUARD(c)                              \
  if (!(c)) return FEAHIP_EINVAL;                 \
  FEA_HIP_CHECK(c, hipSetDevice((c)->device))

extern "C" int feahip_sync(feahip_ctx *c)
{
  CTX_GUARD(c);
  FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return FEAHIP_OK;
}

extern "C" int feahip_set_assembly(feahip_ctx *c, int strategy)
{
  CTX_GUARD(c);
  if (strategy < FEAHIP_ASM_AUTO || strategy > FEAHIP_ASM_STAGED) { c->err = "unknown assembly strategy"; return FEAHIP_EINVAL; }
  c->strategy = strategy;
  return FEAHIP_OK;
}

extern "C" int feahip_set_row_shard(feahip_ctx *c, int rank, int nranks)
{
  CTX_GUARD(c);
  if (nranks < 1 || rank < 0 || rank >= nranks) { c->err = "bad shard (rank, nranks)"; return FEAHIP_EINVAL; }
  // shards are ranges of "supers" (FEA_SUPER_CHUNKS SpMV chunks each): equal
  // numbers of supers = near-equal numbers of 3x3 blocks per rank
  const int nsuper = (c->nchunks + FEA_SUPER_CHUNKS - 1) / FEA_SUPER_CHUNKS;
  const int s0 = (int)((long long)nsuper * rank / nranks), s1 = (int)((long long)nsuper * (rank + 1) / nranks);
  c->chunk0 = s0 * FEA_SUPER_CHUNKS;
  c->nchunks_local = (s1 * FEA_SUPER_CHUNKS < c->nchunks ? s1 * FEA_SUPER_CHUNKS : c->nchunks) - c->chunk0;
  if (!c->h_super_achunk.empty()) {
    c->achunk0 = c->h_super_achunk[s0];
    c->nachunks_local = c->h_super_achunk[s1] - c->achunk0;
  }
  return FEAHIP_OK;
}

extern "C" int feahip_update_nodes_with_bc(feahip_ctx *c, double lambda)
{
  CTX_GUARD(c);
  c->state_valid = false;
  return launch_update_nodes_bc(c, lambda);
}

extern "C" int feahip_update_state(feahip_ctx *c, int *n_bad)
{
  CTX_GUARD(c);
  c->state_valid = false;
  if (n_bad) {
    FEA_HIP_CHECK(c, hipMemcpyAsync(&c->last_bad, c->d_flag + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
    *n_bad = c->last_bad;
  }
  return FEAHIP_OK;
}

extern "C" int feahip_create_stiffness(feahip_ctx *c) { CTX_GUARD(c); return launch_assemble(c, true, false); }
extern "C" int feahip_create_residual_forces(feahip_ctx *c) { CTX_GUARD(c); return launch_assemble(c, false, true); }
extern "C" int feahip_create_stiffness_and_residual(feahip_ctx *c) { CTX_GUARD(c); return launch_assemble(c, true, true); }

extern "C" int feahip_stash_stiffness(feahip_ctx *c)
{
  CTX_GUARD(c);
  const size_t bytes = sizeof(double) * 9 * (size_t)c->nnzb;
  if (!c->d_Kstash) FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_Kstash, bytes ? bytes : 8));
  FEA_HIP_CHECK(c, hipMemcpyAsync(c->d_Kstash, c->d_K, bytes, hipMemcpyDeviceToDevice, c->stream));
  c->have_stash = true;
  return FEAHIP_OK;
}

extern "C" int feahip_restore_stiffness(feahip_ctx *c)
{
  CTX_GUARD(c);
  if (!c->have_stash) { c->err = "restore_stiffness before stash_stiffness"; return FEAHIP_ESTATE; }
  FEA_HIP_CHECK(c, hipMemcpyAsync(c->d_K, c->d_Kstash, sizeof(double) * 9 * (size_t)c->nnzb,
                                  hipMemcpyDeviceToDevice, c->stream));
  return FEAHIP_OK;
}

extern "C" int feahip_apply_prescribed_bc(feahip_ctx *c, double lambda) { CTX_GUARD(c); return launch_apply_bc(c, lambda); }

extern "C" int feahip_solve_slae(feahip_ctx *c, int type, double tol, int max_iter, int *iters, double *resid)
{
  CTX_GUARD(c);
  if (type < FEAHIP_CG || type > FEAHIP_CHOLESKY) { c->err = "unknown solver type"; return FEAHIP_EINVAL; }
  if (max_iter <= 0) { c->err = "max_iterations must be positive"; return FEAHIP_EINVAL; }
  return solve_pcg(c, type, tol, max_iter, iters, resid);
}

extern "C" int feahip_energy(feahip_ctx *c, double *tolerance)
{
  CTX_GUARD(c);
  if (!tolerance) return FEAHIP_EINVAL;
  return launch_dot(c, c->d_f, c->d_u, tolerance);
}

extern "C" int feahip_update_nodes_with_solution(feahip_ctx *c, const double *u)
{
  CTX_GUARD(c);
  c->state_valid = false;
  if (u) {
    FEA_HIP_CHECK(c, hipMemcpyAsync(c->d_q, u, sizeof(double) * (size_t)c->ndof, hipMemcpyHostToDevice, c->stream));
    return launch_update_nodes_solution(c, c->d_q);
  }
  return launch_update_nodes_solution(c, c->d_u);
}

extern "C" int feahip_solve(feahip_ctx *c, int load_increments, int max_newton, int modified_newton,
                            double desired_tolerance, int solver_type, double solver_tolerance,
                            int solver_max_iter, double *tol_log, int tol_log_cap, int *its_log,
                            int *steps_done)
{
  CTX_GUARD(c);
  int rc, nlog = 0, step = 0;
  for (; step < load_increments; ++step) {                       // fea_solver.c:163
    int it = 0;
    double tolerance = 0;
    if ((rc = feahip_update_nodes_with_bc(c, 1.0))) return rc;   // :168
    if ((rc = feahip_update_state(c, nullptr))) return rc;       // :171-174
    if ((rc = feahip_create_stiffness(c))) return rc;            // :177
    if (modified_newton && (rc = feahip_stash_stiffness(c))) return rc;   // :179
    do {
      it++;
      if (modified_newton) {
        if ((rc = feahip_create_residual_forces(c))) return rc;  // :185
        if ((rc = feahip_restore_stiffness(c))) return rc;       // :194-195
      } else if (it == 1) {
        if ((rc = feahip_create_residual_forces(c))) return rc;  // K of :177 is current
      } else {
        if ((rc = feahip_create_stiffness_and_residual(c))) return rc;   // :185 + :200
      }
      if ((rc = feahip_apply_prescribed_bc(c, 0.0))) return rc;  // :203
      if ((rc = feahip_solve_slae(c, solver_type, solver_tolerance, solver_max_iter, nullptr, nullptr))) return rc;  // :205
      if ((rc = feahip_energy(c, &tolerance))) return rc;        // :208-210
      if (tol_log && nlog < tol_log_cap) tol_log[nlog] = tolerance;
      nlog++;
      if ((rc = feahip_update_nodes_with_solution(c, nullptr))) return rc;   // :216
      if ((rc = feahip_update_state(c, nullptr))) return rc;     // :217-218
    } while (fabs(tolerance) > desired_tolerance && it < max_newton);      // :220-221
    if (its_log) its_log[step] = it;
    if (it == max_newton) break;                                 // :225-231
  }
  if (steps_done) *steps_done = step;
  FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  return FEAHIP_OK;
}

// ---- views ---------------------------------------------------------------

extern "C" int feahip_set_nodes(feahip_ctx *c, const double *nodes)
{
  CTX_GUARD(c);
  if (!nodes) return FEAHIP_EINVAL;
  std::vector<double> pad((size_t)c->N * 4, 0.0);
  for (int a = 0; a < c->N; ++a)
    for (int j = 0; j < 3; ++j) pad[(size_t)a * 4 + j] = nodes[(size_t)a * 3 + j];
  FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  FEA_HIP_CHECK(c, hipMemcpy(c->d_x, pad.data(), sizeof(double) * pad.size(), hipMemcpyHostToDevice));
  c->state_valid = false;
  return FEAHIP_OK;
}

extern "C" int feahip_get_nodes(feahip_ctx *c, double *nodes)
{
  CTX_GUARD(c);
  if (!nodes) return FEAHIP_EINVAL;
  std::vector<double> pad((size_t)c->N * 4);
  FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  FEA_HIP_CHECK(c, hipMemcpy(pad.data(), c->d_x, sizeof(double) * pad.size(), hipMemcpyDeviceToHost));
  for (int a = 0; a < c->N; ++a)
    for (int j = 0; j < 3; ++j) nodes[(size_t)a * 3 + j] = pad[(size_t)a * 4 + j];
  return FEAHIP_OK;
}

static int get_vec(feahip_ctx *c, const double *d, double *h, size_t n)
{
  if (!h) return FEAHIP_EINVAL;
  FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  FEA_HIP_CHECK(c, hipMemcpy(h, d, sizeof(double) * n, hipMemcpyDeviceToHost));
  return FEAHIP_OK;
}

extern "C" int feahip_get_forces(feahip_ctx *c, double *f) { CTX_GUARD(c); return get_vec(c, c->d_f, f, (size_t)c->ndof); }
extern "C" int feahip_get_solution(feahip_ctx *c, double *u) { CTX_GUARD(c); return get_vec(c, c->d_u, u, (size_t)c->ndof); }

extern "C" int feahip_set_forces(feahip_ctx *c, const double *f)
{
  CTX_GUARD(c);
  if (!f) return FEAHIP_EINVAL;
  FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  FEA_HIP_CHECK(c, hipMemcpy(c->d_f, f, sizeof(double) * (size_t)c->ndof, hipMemcpyHostToDevice));
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

extern "C" int feahip_matrix_nnz(feahip_ctx *c, long long *nnz)
{
  if (!c || !nnz) return FEAHIP_EINVAL;
  *nnz = (long long)c->nnzb * 9;
  return FEAHIP_OK;
}

extern "C" int feahip_get_matrix_yale(feahip_ctx *c, int *offsets, int *indexes, double *values)
{
  CTX_GUARD(c);
  if (!offsets || !indexes || !values) return FEAHIP_EINVAL;
  if ((long long)c->nnzb * 9 > 0x7FFFFFFFLL) { c->err = "matrix too large for 32-bit Yale offsets"; return FEAHIP_EINVAL; }
  std::vector<double> K((size_t)c->nnzb * 9);
  int rc = get_vec(c, c->d_K, K.data(), K.size());
  if (rc) return rc;
  int pos = 0;
  offsets[0] = 0;
  for (int a = 0; a < c->N; ++a)
    for (int i = 0; i < 3; ++i) {
      for (int q = c->h_rowptr[a]; q < c->h_rowptr[a + 1]; ++q)
        for (int j = 0; j < 3; ++j) {
          indexes[pos] = 3 * c->h_colidx[q] + j;
          values[pos] = K[(size_t)q * 9 + 3 * i + j];
          pos++;
        }
      offsets[3 * a + i + 1] = pos;
    }
  return FEAHIP_OK;
}

extern "C" int feahip_spmv(feahip_ctx *c, const double *x, double *y)
{
  CTX_GUARD(c);
  if (!x || !y) return FEAHIP_EINVAL;
  FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream));
  FEA_HIP_CHECK(c, hipMemcpy(c->d_p, x, sizeof(double) * (size_t)c->ndof, hipMemcpyHostToDevice));
  int rc = launch_spmv(c, c->d_p, c->d_q);
  if (rc) return rc;
  return get_vec(c, c->d_q, y, (size_t)c->ndof);
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

extern "C" int feahip_sizes(feahip_ctx *c, long long *o)
{
  if (!c || !o) return FEAHIP_EINVAL;
  o[0] = c->N; o[1] = c->E; o[2] = c->npe; o[3] = c->G; o[4] = c->nnzb; o[5] = c->nchunks;
  // bytes of the maps the default assembly kernel reads besides the algorithmic inputs
  o[6] = c->have_visits ? c->visit_bytes + (long long)(c->N + 1) * 8 : c->aux_bytes; o[7] = c->max_rowlen;
  return FEAHIP_OK;
}
