/*
 * fea_solve.c -- the load-increment / Newton driver and the Gmsh export,
 * in C on the host, exactly where the reference has them.
 *
 * fea_solve() is solve() of solver-large/fea_solver.c:130-242 with each
 * solver_* call replaced by its C-ABI twin (include/fea_hip.h).  The host
 * never touches element data inside the loops: one scalar (<u,f>) comes back
 * per Newton iteration, which is all the control flow needs.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fea_host.h"

int fea_deck_create_solver(const fea_deck *d, int device, feahip_ctx **ctx, char *errbuf, int errlen)
{
  double w[32], dforms[32 * 3 * 10];
  int npe = fea_element_tables(d->ele_type, d->gauss_nodes_count, w, NULL, dforms);
  int rc;
  if (npe < 0 || npe != d->nodes_per_element) {
    if (errbuf) snprintf(errbuf, (size_t)errlen, "unsupported element type / gauss rule (%d nodes, %d points)",
                         d->nodes_per_element, d->gauss_nodes_count);
    return FEAHIP_EINVAL;
  }
  rc = feahip_create(ctx, device, d->nodes_count, d->elements_count, npe, d->gauss_nodes_count, w, dforms,
                     d->elements, d->nodes, d->model, d->parameters, d->parameters_count,
                     d->prescribed_nodes_count, d->presc_node, d->presc_type, d->presc_values);
  if (rc && errbuf) snprintf(errbuf, (size_t)errlen, "%s", feahip_create_error());
  return rc;
}

#define CALL(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

/* solve() of fea_solver.c:163-236: THE load-increment / Newton loop of the host side (fea_solve and
 * fea_solve_with_snapshots are this loop with two different things done after a converged increment, :233-235).
 * Returns the number of finished increments, or a negative FEAHIP_* code. */
int fea_solve_steps(const fea_deck *d, feahip_ctx *ctx, void *logp, fea_step_fn after_step, void *user)
{
  FILE *log = (FILE *)logp;
  int step, it;
  double tolerance;
  for (step = 0; step < d->load_increments_count; ++step) {                 /* :163 */
    it = 0;
    CALL(feahip_update_nodes_with_bc(ctx, 1));                               /* :168 */
    CALL(feahip_update_state(ctx, NULL));                                    /* :171-174 */
    CALL(feahip_create_stiffness(ctx));                                      /* :177 */
    CALL(feahip_stash_stiffness(ctx));                                       /* :179 */
    do {
      it++;
      CALL(feahip_create_residual_forces(ctx));                              /* :185 */
      if (d->modified_newton) CALL(feahip_restore_stiffness(ctx));           /* :194-195 */
      else CALL(feahip_create_stiffness(ctx));                               /* :200 */
      CALL(feahip_apply_prescribed_bc(ctx, 0));                              /* :203 */
      CALL(feahip_solve_slae(ctx, d->solver_type, d->solver_tolerance,       /* :205 */
                             d->solver_max_iter, NULL, NULL));
      CALL(feahip_energy(ctx, &tolerance));                                  /* :208-210 */
      if (log) {
        fprintf(log, "Tolerance <X,R> = %e\n", tolerance);                   /* :212 */
        fprintf(log, "Newton iteration %d finished\n", it);                  /* :213 */
      }
      CALL(feahip_update_nodes_with_solution(ctx, NULL));                    /* :216 */
      CALL(feahip_update_state(ctx, NULL));                                  /* :217-218 */
    } while (fabs(tolerance) > d->desired_tolerance && it < d->max_newton_count);   /* :220-221 */
    if (log) fprintf(log, "Load increment %d finished\n", step + 1);         /* :224 */
    if (it == d->max_newton_count) {                                         /* :225-231 */
      if (log) fprintf(log, "Unable to finish load step in %d Newton iterations,exit\n", d->max_newton_count);
      break;
    }
    if (after_step) CALL(after_step(d, ctx, step, user));                    /* :233-235 */
  }
  return step;
}

struct node_sink { double *x; int cap; };

static int keep_nodes(const fea_deck *d, feahip_ctx *ctx, int step, void *user)
{
  struct node_sink *k = (struct node_sink *)user;
  if (!k->x || step >= k->cap) return 0;
  return feahip_get_nodes(ctx, k->x + (size_t)step * d->nodes_count * 3);
}

int fea_solve(const fea_deck *d, feahip_ctx *ctx, void *logp, double *x_steps, int x_steps_cap)
{
  struct node_sink k;
  k.x = x_steps; k.cap = x_steps_cap;
  return fea_solve_steps(d, ctx, logp, keep_nodes, &k);
}
