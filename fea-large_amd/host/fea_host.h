/*
 * fea_host.h -- host side (plain C) above the C ABI of include/fea_hip.h.
 *
 * Keeps what the reference keeps on the host: the S-expression input deck
 * (solver-large/sexp_loader.c), the element plug-in tables
 * (fea_solver.c:32-54, 503-535, 1287-1373), the load-increment / Newton
 * control flow of solve() (fea_solver.c:130-242) and the Gmsh export
 * (fea_solver.c:1375-1488).  Names follow the reference's so that a
 * maintainer can map one onto the other; layouts are flat arrays instead of
 * the reference's pointer-per-row heap objects.
 */
#ifndef FEA_HOST_H
#define FEA_HOST_H

#include "../../include/fea_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define FEA_MAX_MATERIAL_PARAMETERS 10   /* defines.h:22 */

typedef enum { FEA_TETRAHEDRA10 = 0, FEA_TETRAHEDRA4 = 1, FEA_HEXAHEDRA8 = 2 } fea_element_type;

/* fea_task + fea_solution_params + the three input arrays of the reference
 * (fea_solver.h:94-155), as one flat record                                 */
typedef struct fea_deck {
  /* task */
  int model;                    /* FEAHIP_MODEL_*                            */
  double parameters[FEA_MAX_MATERIAL_PARAMETERS]; /* [0]=lambda [1]=mu       */
  int parameters_count;
  int solver_type;              /* FEAHIP_CG / PCG_ILU / CHOLESKY            */
  double solver_tolerance;
  int solver_max_iter;
  int ele_type;                 /* fea_element_type                          */
  int load_increments_count;
  double desired_tolerance;
  int max_newton_count;
  int linesearch_max, arclength_max;   /* parsed, unused (as the reference)  */
  int modified_newton;
  /* solution params */
  int nodes_per_element, gauss_nodes_count;
  /* geometry */
  int nodes_count;
  double *nodes;                /* [nodes_count][3]                          */
  int elements_count;
  int *elements;                /* [elements_count][nodes_per_element]       */
  /* prescribed displacements, deck order */
  int prescribed_nodes_count;
  int *presc_node, *presc_type;
  double *presc_values;         /* [count][3]                                */
} fea_deck;

/* sexp_data_load (sexp_loader.c:275-327).  Returns 0, or -1 with a message
 * in errbuf.  Defaults as fea_task_alloc / fea_solution_params_alloc
 * (fea_solver.c:1509-1549) and process_slae_solver (sexp_loader.c:100-103). */
int fea_deck_load(const char *path, fea_deck *deck, char *errbuf, int errlen);
void fea_deck_free(fea_deck *deck);
/* writes the same grammar (the emitter of utilities/tetgenProcessor)        */
int fea_deck_save(const char *path, const fea_deck *deck);

/* element plug-in: what solver_create_element_params_<type> +
 * solver_gauss_node_alloc produce.  weights[G], forms[G][npe],
 * dforms[G][3][npe].  Returns npe, or -1 for an unsupported pair.           */
int fea_element_tables(int ele_type, int gauss_count, double *weights,
                       double *forms, double *dforms);

/* creates the device context for a deck */
int fea_deck_create_solver(const fea_deck *deck, int device, feahip_ctx **ctx,
                           char *errbuf, int errlen);

/* solve() of the reference, calling the C ABI for every step.  Prints the
 * three quantitative log lines of the reference (fea_solver.c:212-213,224)
 * to `log` when it is not NULL.  x_steps (may be NULL) receives the node
 * coordinates after every completed load step, [steps][N][3].
 * Returns the number of completed load steps, or a negative FEAHIP_E* code. */
/* The load-increment / Newton loop itself (solve(), fea_solver.c:163-236): after every converged increment
 * `after_step(deck, ctx, step, user)` runs where the reference takes its load_step snapshot (:233-235); a
 * non-zero return aborts the loop with that code.  Returns the increments finished, or a negative FEAHIP_* code. */
typedef int (*fea_step_fn)(const fea_deck *deck, feahip_ctx *ctx, int step, void *user);
int fea_solve_steps(const fea_deck *deck, feahip_ctx *ctx, void *log /* FILE* */, fea_step_fn after_step, void *user);
int fea_solve(const fea_deck *deck, feahip_ctx *ctx, void *log /* FILE* */,
              double *x_steps, int x_steps_cap);

/* Per-load-step snapshot (load_step of the reference, fea_solver.h:212-224):
 * node coordinates and the stress of Gauss point 0 of every element.         */
typedef struct fea_step_snapshot {
  double *nodes;      /* [nodes_count][3] */
  double *stress0;    /* [elements_count][9], stresses[e][0] */
} fea_step_snapshot;

/* fea_solve with snapshots for the exporter: steps[cap] are allocated by the
 * callee (free with fea_snapshots_free).  Returns completed steps or <0.     */
int fea_solve_with_snapshots(const fea_deck *deck, feahip_ctx *ctx, void *log,
                             fea_step_snapshot *steps, int cap);
void fea_snapshots_free(fea_step_snapshot *steps, int n);

/* solver_export_tetrahedra10_gmsh (fea_solver.c:1375-1488): Gmsh 2.0 ASCII,
 * nodes with %f, TET10 elements with local nodes 8 and 9 swapped, and per
 * load step (step 0 = zeros) NodeData "Displacements" and ElementData
 * "Stress tensor" (Gauss point 0) tagged load*0.83333333.  4-node elements
 * are written as Gmsh type 4.                                                */
int fea_export_gmsh(const char *filename, const fea_deck *deck,
                    const fea_step_snapshot *steps, int nsteps);

/* "<base>.msh" next to the deck, as initial_data_load builds it
 * (fea_solver.c:1681-1687); out must hold strlen(deck_path)+5 bytes.        */
void fea_export_name(const char *deck_path, char *out);

#ifdef __cplusplus
}
#endif
#endif
