// dist.hip -- row-sharded (multi-GPU) operation: shard installation, the RCCL
// transport, the Newton loop over one or more ranks.
//
// Production shape (bench.py, torch.distributed launch): one process per GPU,
// one context per process, halo rows over ncclSend/ncclRecv (point-to-point
// over xGMI; every rank talks to its slab neighbours only) and one to three
// doubles per ncclAllReduce.  No vector all-reduce, no all-gather on the path.
// The same loop also drives an in-process group of contexts (GroupTransport),
// which is how the sharded path is exercised where one process sees the GPU.
#include "feahip_internal.h"
#include <rccl/rccl.h>
#include <cmath>
#include <cstring>

// halo lists of a plan to the device, and the longest run of this rank's SpMV chunks whose rows touch no halo column:
// they can be multiplied while the halo rows are in flight (a slab has its halo-touching chunks at its two ends)
int install_plan(feahip_ctx *c, const ShardPlan &plan)
{
  c->rank = plan.rank; c->nranks = plan.nranks; c->row0 = plan.row0; c->row1 = plan.row1;
  {
    int best_lo = 0, best_hi = 0, lo = -1;
    for (int k = 0; k <= c->nchunks_local; ++k) {
      bool interior = false;
      if (k < c->nchunks_local) {
        interior = true;
        const int r0 = c->h_chunk[c->chunk0 + k], r1 = c->h_chunk[c->chunk0 + k + 1];
        for (int q = c->h_rowptr[r0]; q < c->h_rowptr[r1] && interior; ++q)
          if (c->h_colidx[q] < plan.row0 || c->h_colidx[q] >= plan.row1) interior = false;
      }
      if (interior) { if (lo < 0) lo = k; }
      else if (lo >= 0) { if (k - lo > best_hi - best_lo) { best_lo = lo; best_hi = k; } lo = -1; }
    }
    c->ichunk_lo = best_lo; c->ichunk_hi = best_hi;
    // what the overlapped exchange relies on, checked outright: no chunk of the interior range reads a halo column
    for (int k = best_lo; k < best_hi; ++k) {
      const int r0 = c->h_chunk[c->chunk0 + k], r1 = c->h_chunk[c->chunk0 + k + 1];
      for (int q = c->h_rowptr[r0]; q < c->h_rowptr[r1]; ++q)
        if (c->h_colidx[q] < plan.row0 || c->h_colidx[q] >= plan.row1) { c->err = "interior chunk range reads a halo column"; return FEAHIP_ESTATE; }
    }
  }
  c->peer = plan.peer; c->send_off = plan.send_off; c->recv_off = plan.recv_off;
  c->nsend = (int)plan.send_idx.size(); c->nrecv = (int)plan.recv_idx.size();
  for (void *p : {(void *)c->d_send_idx, (void *)c->d_recv_idx, (void *)c->d_send_buf, (void *)c->d_recv_buf})
    if (p) (void)hipFree(p);
  c->d_send_idx = c->d_recv_idx = nullptr; c->d_send_buf = c->d_recv_buf = nullptr;
  auto up = [&](int **dst, const std::vector<int> &v) -> int {
    FEA_HIP_CHECK(c, hipMalloc((void **)dst, sizeof(int) * (v.size() ? v.size() : 1)));
    if (!v.empty()) FEA_HIP_CHECK(c, hipMemcpy(*dst, v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice));
    return FEAHIP_OK;
  };
  int rc;
  if ((rc = up(&c->d_send_idx, plan.send_idx))) return rc;
  if ((rc = up(&c->d_recv_idx, plan.recv_idx))) return rc;
  FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_send_buf, sizeof(double) * 3 * (size_t)(c->nsend ? c->nsend : 1)));
  FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_recv_buf, sizeof(double) * 3 * (size_t)(c->nrecv ? c->nrecv : 1)));
  return FEAHIP_OK;
}

int install_shard(feahip_ctx *c, int rank, int nranks)
{
  if (nranks < 1 || rank < 0 || rank >= nranks) { c->err = "bad shard (rank, nranks)"; return FEAHIP_EINVAL; }
  if (c->rank_own >= 0) {                                     // a rank context is its shard: nothing to cut
    if (rank == c->rank && nranks == c->nranks) return FEAHIP_OK;
    c->err = "a rank context (feahip_create_rank) holds one rank's sub-mesh and cannot be re-sharded";
    return FEAHIP_EINVAL;
  }
  const int nsuper = (c->nchunks + FEA_SUPER_CHUNKS - 1) / FEA_SUPER_CHUNKS;
  const int s0 = (int)((long long)nsuper * rank / nranks), s1 = (int)((long long)nsuper * (rank + 1) / nranks);
  c->chunk0 = s0 * FEA_SUPER_CHUNKS < c->nchunks ? s0 * FEA_SUPER_CHUNKS : c->nchunks;
  c->nchunks_local = (s1 * FEA_SUPER_CHUNKS < c->nchunks ? s1 * FEA_SUPER_CHUNKS : c->nchunks) - c->chunk0;
  if (!c->h_super_achunk.empty()) {
    c->achunk0 = c->h_super_achunk[s0];
    c->nachunks_local = c->h_super_achunk[s1] - c->achunk0;
  }
  ShardPlan plan;
  build_shard_plan(c->h_rowptr, c->h_colidx, c->h_chunk, rank, nranks, plan);
  if (plan.row0 != c->row0 || plan.row1 != c->row1) release_k(c);     // K is re-allocated for the new rows on next use
  return install_plan(c, plan);
}

// ---- RCCL ------------------------------------------------------------------
struct RcclTransport : Transport {
  ncclComm_t comm = nullptr;
  ~RcclTransport() override { if (comm) (void)ncclCommDestroy(comm); }
  int fail(feahip_ctx *c, const char *what, ncclResult_t r)
  {
    c->err = std::string(what) + ": " + ncclGetErrorString(r);
    return FEAHIP_ECOMM;
  }
  int exchange(std::vector<feahip_ctx *> &R, int which) override
  {
    feahip_ctx *c = R[0];
    int stride = (which == 2) ? 4 : 3;
    double *v = which == 0 ? c->d_p : (which == 1 ? c->d_u : c->d_x);
    extern void feahip_enq_pack(feahip_ctx *, double *, int);
    extern void feahip_enq_unpack(feahip_ctx *, double *, int);
    feahip_enq_pack(c, v, stride);
    ncclResult_t r = ncclGroupStart();
    if (r != ncclSuccess) return fail(c, "ncclGroupStart", r);
    for (size_t k = 0; k < c->peer.size(); ++k) {
      const size_t ns = (size_t)3 * (c->send_off[k + 1] - c->send_off[k]), nr = (size_t)3 * (c->recv_off[k + 1] - c->recv_off[k]);
      if (ns && (r = ncclSend(c->d_send_buf + (size_t)3 * c->send_off[k], ns, ncclDouble, c->peer[k], comm, c->stream)) != ncclSuccess)
        return fail(c, "ncclSend", r);
      if (nr && (r = ncclRecv(c->d_recv_buf + (size_t)3 * c->recv_off[k], nr, ncclDouble, c->peer[k], comm, c->stream)) != ncclSuccess)
        return fail(c, "ncclRecv", r);
    }
    if ((r = ncclGroupEnd()) != ncclSuccess) return fail(c, "ncclGroupEnd", r);
    feahip_enq_unpack(c, v, stride);
    return FEAHIP_OK;
  }
  // the same exchange on a stream of its own: pack on the context's stream, send / receive / unpack on the
  // communication stream, so that the product of the rows that touch no halo column runs meanwhile
  int exchange_begin(std::vector<feahip_ctx *> &R, int which) override
  {
    feahip_ctx *c = R[0];
    if (!c->comm_stream) {
      FEA_HIP_CHECK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
      FEA_HIP_CHECK(c, hipEventCreateWithFlags(&c->ev_packed, hipEventDisableTiming));
      FEA_HIP_CHECK(c, hipEventCreateWithFlags(&c->ev_unpacked, hipEventDisableTiming));
    }
    const int stride = (which == 2) ? 4 : 3;
    double *v = which == 0 ? c->d_p : (which == 1 ? c->d_u : (which == 2 ? c->d_x : c->d_z));
    extern void feahip_enq_pack(feahip_ctx *, double *, int);
    extern void feahip_enq_unpack_on(feahip_ctx *, double *, int, hipStream_t);
    feahip_enq_pack(c, v, stride);
    FEA_HIP_CHECK(c, hipEventRecord(c->ev_packed, c->stream));
    FEA_HIP_CHECK(c, hipStreamWaitEvent(c->comm_stream, c->ev_packed, 0));
    ncclResult_t r = ncclGroupStart();
    if (r != ncclSuccess) return fail(c, "ncclGroupStart", r);
    for (size_t k = 0; k < c->peer.size(); ++k) {
      const size_t ns = (size_t)3 * (c->send_off[k + 1] - c->send_off[k]), nr = (size_t)3 * (c->recv_off[k + 1] - c->recv_off[k]);
      if (ns && (r = ncclSend(c->d_send_buf + (size_t)3 * c->send_off[k], ns, ncclDouble, c->peer[k], comm, c->comm_stream)) != ncclSuccess)
        return fail(c, "ncclSend", r);
      if (nr && (r = ncclRecv(c->d_recv_buf + (size_t)3 * c->recv_off[k], nr, ncclDouble, c->peer[k], comm, c->comm_stream)) != ncclSuccess)
        return fail(c, "ncclRecv", r);
    }
    if ((r = ncclGroupEnd()) != ncclSuccess) return fail(c, "ncclGroupEnd", r);
    feahip_enq_unpack_on(c, v, stride, c->comm_stream);
    FEA_HIP_CHECK(c, hipEventRecord(c->ev_unpacked, c->comm_stream));
    return FEAHIP_OK;
  }
  int exchange_end(std::vector<feahip_ctx *> &R) override
  {
    feahip_ctx *c = R[0];
    FEA_HIP_CHECK(c, hipStreamWaitEvent(c->stream, c->ev_unpacked, 0));
    return FEAHIP_OK;
  }
  int allreduce(std::vector<feahip_ctx *> &R, int slot, int n) override
  {
    feahip_ctx *c = R[0];
    ncclResult_t r = ncclAllReduce(c->d_scal + 8 + slot, c->d_scal + 8 + slot, (size_t)n, ncclDouble, ncclSum, comm, c->stream);
    if (r != ncclSuccess) return fail(c, "ncclAllReduce", r);
    return FEAHIP_OK;
  }
};

int rccl_unique_id(void *out, int cap)
{
  if (!out || cap < (int)sizeof(ncclUniqueId)) return FEAHIP_EINVAL;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return FEAHIP_ECOMM;
  memcpy(out, &id, sizeof(id));
  return (int)sizeof(id);
}

Transport *make_rccl_transport(feahip_ctx *c, int rank, int nranks, const void *unique_id, std::string &err)
{
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  RcclTransport *t = new RcclTransport();
  (void)hipSetDevice(c->device);
  ncclResult_t r = ncclCommInitRank(&t->comm, nranks, id, rank);
  if (r != ncclSuccess) { err = std::string("ncclCommInitRank: ") + ncclGetErrorString(r); delete t; return nullptr; }
  return t;
}

// ---- the loop of solve() over one or more ranks -------------------------------
int dist_newton(std::vector<feahip_ctx *> &R, int load_increments, int max_newton, int modified_newton,
                double desired_tolerance, int solver_type, double solver_tolerance, int solver_max_iter,
                double *tol_log, int tol_log_cap, int *its_log, int *steps_done)
{
  int rc, nlog = 0, step = 0;
#define EACH(call) for (feahip_ctx *c : R) { if ((rc = (call))) return rc; }
  for (; step < load_increments; ++step) {                           // fea_solver.c:163
    int it = 0;
    double tolerance = 0;
    EACH(feahip_update_nodes_with_bc(c, 1.0));                        // :168 (prescribed values are replicated)
    EACH(feahip_update_state(c, nullptr));                            // :171-174
    EACH(feahip_create_stiffness(c));                                 // :177 (owned rows, ghost elements recomputed)
    if (modified_newton) EACH(feahip_stash_stiffness(c));             // :179
    do {
      it++;
      if (modified_newton) {
        EACH(feahip_create_residual_forces(c));                       // :185
        EACH(feahip_restore_stiffness(c));                            // :194-195
      } else if (it == 1) {
        EACH(feahip_create_residual_forces(c));                       // K of :177 is current
      } else {
        EACH(feahip_create_stiffness_and_residual(c));                // :185 + :200
      }
      EACH(feahip_apply_prescribed_bc(c, 0.0));                       // :203
      if ((rc = dist_solve_pcg(R, solver_type, solver_tolerance, solver_max_iter, nullptr, nullptr))) return rc;  // :205
      if ((rc = dist_energy(R, &tolerance))) return rc;               // :208-210, identical on every rank
      if (tol_log && nlog < tol_log_cap) tol_log[nlog] = tolerance;
      nlog++;
      const int ls_max = R[0]->linesearch_max;
      if (ls_max <= 0) {
        if ((rc = dist_update_nodes_with_solution(R, nullptr))) return rc;   // :216
      } else {
        // Golden-section search for the step length eta in [1/2, 1] that minimises |eta <u, R(x + eta u)>|
        // (solver-prototype/cartesian3d/large/cartesian3d_large.m:85-119; the C solver parses
        // line-search :max and never uses it, fea_solver.c:1517).  Two residual assemblies per iteration.
        const double tau = (sqrt(5.0) - 1.0) / 2.0;
        double a = 0.5, b = 1.0, eta = 1.0, at = 0.0;                 // at: the multiple of u currently added to x
        for (int ls = 0; ls < ls_max; ++ls) {
          const double x1 = b - tau * (b - a), x2 = a + tau * (b - a);
          double f[2];
          for (int k = 0; k < 2; ++k) {
            const double xk = k == 0 ? x1 : x2;
            if ((rc = dist_nodes_add_scaled(R, xk - at, at == 0.0))) return rc;
            at = xk;
            EACH(feahip_create_residual_forces(c));
            double uf = 0;
            if ((rc = dist_energy(R, &uf))) return rc;                // <u, -R> with the sign of the residual vector f
            f[k] = fabs(xk * uf);
          }
          if (f[0] > f[1]) a = x1; else b = x2;
          if (fabs(tolerance) < f[0] && fabs(tolerance) < f[1]) { eta = 1.0; break; }
          eta = 0.5 * (x1 + x2);
        }
        if ((rc = dist_nodes_add_scaled(R, eta - at, at == 0.0))) return rc;
      }
      EACH(feahip_update_state(c, nullptr));                          // :217-218
    } while (fabs(tolerance) > desired_tolerance && it < max_newton); // :220-221
    if (its_log) its_log[step] = it;
    if (it == max_newton) break;                                      // :225-231
  }
#undef EACH
  if (steps_done) *steps_done = step;
  for (feahip_ctx *c : R) { (void)hipSetDevice(c->device); FEA_HIP_CHECK(c, hipStreamSynchronize(c->stream)); }
  return FEAHIP_OK;
}
