/*
 * feasolver_main.c -- `feasolver_hip deck.sexp`: the reference's command line
 * (fea_solver.c:64-128, 324-333) on top of the HIP path.  Loads the deck,
 * runs the load-increment / Newton loop through the C ABI, writes
 * "<base>.msh" like initial_data_load + solve() do.
 *
 * One option the reference does not have, after the deck name:
 *   --multigrid   PCG_ILU / CHOLESKY solves use the aggregation-multigrid
 *                 preconditioner (feahip_set_preconditioner); an error, not a
 *                 silent return to block-Jacobi, when the mesh is too small
 *                 to coarsen.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fea_host.h"

int main(int argc, char **argv)
{
  fea_deck deck;
  feahip_ctx *ctx = NULL;
  char err[512], *msh;
  fea_step_snapshot *steps;
  int done, rc, cap, status = 0;
  if (argc < 2) {
    printf("Usage: fea_solve input_data.sexp\n");            /* fea_solver.c:328 */
    return 1;
  }
  if (fea_deck_load(argv[1], &deck, err, sizeof err)) {
    fprintf(stderr, "Error. Unable to load %s: %s\n", argv[1], err);
    return 1;
  }
  printf("Initial data loaded\n");
  if ((rc = fea_deck_create_solver(&deck, 0, &ctx, err, sizeof err))) {
    fprintf(stderr, "feasolve error encountered: %s\n", err);
    fea_deck_free(&deck);
    return 1;
  }
  if (argc > 2 && strcmp(argv[2], "--multigrid") == 0 && feahip_set_preconditioner(ctx, 1)) {
    /* asked for and not available: say so and stop, never run another preconditioner in its place */
    fprintf(stderr, "feasolve error encountered: --multigrid: %s\n", feahip_last_error(ctx));
    feahip_destroy(ctx);
    fea_deck_free(&deck);
    return 1;
  }
  cap = deck.load_increments_count > 0 ? deck.load_increments_count : 1;
  steps = (fea_step_snapshot *)calloc((size_t)cap, sizeof *steps);
  done = fea_solve_with_snapshots(&deck, ctx, stdout, steps, deck.load_increments_count);
  if (done < 0) {
    /* a HIP failure or a broken-down linear solve: the reference's error() exits with EXIT_FAILURE
     * (fea_solver.c:57-61); nothing is exported */
    fprintf(stderr, "feasolve error encountered: %s\n", feahip_last_error(ctx));
    status = 1;
  } else {
    printf("Exporting data...\n");
    msh = (char *)malloc(strlen(argv[1]) + 8);
    fea_export_name(argv[1], msh);
    /* a failed increment leaves current_load_step one lower (fea_solver.c:227), so the
     * reference then drops the last completed step from the file: same here */
    if (fea_export_gmsh(msh, &deck, steps, done == deck.load_increments_count ? done : done - 1)) {
      fprintf(stderr, "could not write %s\n", msh);
      status = 1;
    }
    free(msh);
  }
  fea_snapshots_free(steps, cap);                      /* by capacity: an error may leave snapshots behind */
  free(steps);
  feahip_destroy(ctx);
  fea_deck_free(&deck);
  return status;
}
