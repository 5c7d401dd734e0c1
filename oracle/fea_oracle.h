/*
 * fea_oracle.h -- CPU restatement of the solver-large hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is product code: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker.  The shipped path is the HIP
 * library behind include/fea_hip.h; it never calls in here.
 *
 * What is restated (reference = /root/reference/solver-large):
 *   fea_solver.c:32-54     Gauss tables (4- and 5-point TET10 rules)
 *   fea_solver.c:503-535   per-Gauss-point shape-function tables
 *   fea_solver.c:656-722   Jacobian, inverse, spatial shape gradients
 *   fea_solver.c:1118-1188 deformation gradient + stress at a Gauss point
 *   fea_solver.c:887-1068  constitutive + initial-stress stiffness, scatter
 *   fea_solver.c:1072-1114 residual forces
 *   fea_solver.c:1200-1284 prescribed-displacement handling, node update
 *   fea_solver.c:130-242   load-increment / Newton loop
 *   fea_model.c:26-148     A5 and compressible Neo-Hookean stress / tangent
 *   dense_matrix.c:16-110  cdot, det3x3, inv3x3, 3x3 products
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - 3x3 algebra and both material models: checked bit-for-bit against the
 *     reference's own dense_matrix.c / fea_model.c compiled into
 *     oracle/_ref/libfearef.so, and against tests.c:17-24.
 *   - element loops / Newton loop: fea_solver.c cannot be compiled here
 *     (it needs liblogger, libsexp and libspmatrix headers that are not in
 *     the tree), so these are pinned by the reference's closed-form
 *     uniaxial solutions on its own *_analytical.sexp decks and by
 *     structural invariants.
 *   - sparse accumulation, cross-cancellation and the linear solvers live in
 *     libspmatrix (github.com/fourier/libspmatrix, no pinned version, not in
 *     the tree): PARITY UNPINNED for their internals; the restatement pins
 *     them mathematically (K u = f).
 */
#ifndef FEA_ORACLE_H
#define FEA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_NPE   10   /* nodes per element  */
#define ORC_MAX_GAUSS 27   /* Gauss points       */

/* element kinds: TET10 is the reference's; the others are build extensions
 * that run through the same generic loops            */
enum { ORC_TET10 = 0, ORC_TET4 = 1, ORC_HEX8 = 2 };
/* material models, numbered as fea_model.h:37-40 */
enum { ORC_MODEL_A5 = 0, ORC_MODEL_NEOHOOKEAN = 1 };
/* linear solvers, numbered as fea_solver.h:62-66 */
enum { ORC_CG = 0, ORC_PCG_ILU = 1, ORC_CHOLESKY = 2 };

typedef struct orc_elem_table {
  int npe, ngauss;
  double weight[ORC_MAX_GAUSS];
  double forms[ORC_MAX_GAUSS][ORC_MAX_NPE];
  double dforms[ORC_MAX_GAUSS][3][ORC_MAX_NPE];
} orc_elem_table;

typedef struct orc_bc_node {
  int node;
  double values[3];
  int type;              /* bitmask 1=x 2=y 4=z, fea_solver.h:74-83 */
} orc_bc_node;

typedef struct orc_solver orc_solver;

/* ---- leaf functions (dense_matrix.c, fea_model.c) ---------------------- */
double orc_cdot(const double *a, const double *b, int n);
double orc_det3x3(double m[3][3]);
int    orc_inv3x3(double m[3][3], double *det);
void   orc_mul3x3(double A[3][3], double B[3][3], double R[3][3]);
void   orc_tmul3x3(double A[3][3], double B[3][3], double R[3][3]);
void   orc_mult3x3(double A[3][3], double B[3][3], double R[3][3]);
void   orc_stress(int model, const double *params, double F[3][3], double S[3][3]);
void   orc_ctensor(int model, const double *params, double F[3][3],
                   double c[3][3][3][3]);

/* ---- element tables ---------------------------------------------------- */
/* returns 0 on success, -1 for an unsupported (kind, ngauss) pair */
int orc_elem_table_init(orc_elem_table *t, int kind, int ngauss);

/* ---- solver object ----------------------------------------------------- */
orc_solver *orc_solver_create(int n_nodes, int n_elems, int kind, int ngauss,
                              const int *conn, const double *X0,
                              int model, const double *params,
                              int n_bc, const orc_bc_node *bc);
void orc_solver_free(orc_solver *s);

/* set current coordinates (N x 3) */
void orc_set_nodes(orc_solver *s, const double *x);
void orc_get_nodes(const orc_solver *s, double *x);

/* fea_solver.c:831-834 + 843-861: gradients, F, sigma for every (e,g).
 * returns the number of (e,g) whose Jacobian was exactly singular          */
int  orc_update_state(orc_solver *s);
/* views into per-(e,g) state, layouts [E][G][3][npe], [E][G], [E][G][9]    */
const double *orc_grads(const orc_solver *s);
const double *orc_detj(const orc_solver *s);
const double *orc_graddefs(const orc_solver *s);
const double *orc_stresses(const orc_solver *s);

/* one element's local matrices, summed over Gauss points in reference order;
 * Kc, Ks are (3*npe)^2 row-major                                           */
void orc_element_stiffness(const orc_solver *s, int e, double *Kc, double *Ks);
void orc_element_residual(const orc_solver *s, int e, double *fe);

/* sparse pattern (full symmetric scalar CSR, sorted columns, Yale shape)   */
int        orc_nnz(const orc_solver *s);
const int *orc_offsets(const orc_solver *s);
const int *orc_indexes(const orc_solver *s);
double    *orc_values(orc_solver *s);
double    *orc_forces(orc_solver *s);
double    *orc_solution(orc_solver *s);

void orc_create_stiffness(orc_solver *s);       /* fea_solver.c:873-883   */
/* state + stiffness + residual of every element with the elements coloured (no two of a colour share a node) and each
 * colour a parallel loop over `nthreads` host threads (<= 0: the OpenMP default): the all-cores CPU baseline of bench.py.
 * Returns the number of colours. */
int  orc_assemble_coloured(orc_solver *s, int nthreads);
void orc_create_residual_forces(orc_solver *s); /* fea_solver.c:863-870   */
void orc_update_nodes_with_bc(orc_solver *s, double lambda);    /* :1281  */
void orc_apply_prescribed_bc(orc_solver *s, double lambda);     /* :1200  */
void orc_update_nodes_with_solution(orc_solver *s, const double *u); /*:1270*/
void orc_stash_stiffness(orc_solver *s);        /* sp_matrix_copy, :179   */
void orc_restore_stiffness(orc_solver *s);      /* :194-195               */

/* linear solve of values*u = forces; returns iterations (0 for Cholesky)   */
int orc_solve_slae(orc_solver *s, int type, double tol, int max_iter,
                   double *residual_out);

/* whole fea_solver.c:130-242 loop.  tol_log (may be NULL) receives <u,f> of
 * every Newton iteration, its_log the iteration count of every load step.
 * returns the number of completed load steps.                              */
int orc_solve(orc_solver *s, int load_increments, int max_newton,
              int modified_newton, double desired_tolerance,
              int slae_type, double slae_tol, int slae_max_iter,
              double *tol_log, int tol_log_cap, int *its_log);

/* y = K x on the current values (for tests) */
void orc_spmv(const orc_solver *s, const double *x, double *y);

#ifdef __cplusplus
}
#endif
#endif
