/*
 * fea_oracle.c -- CPU restatement of the solver-large hot path.
 * TEST INFRASTRUCTURE ONLY (see fea_oracle.h).  Build with
 *   gcc -O2 -std=c99 -ffp-contract=off
 * so that the arithmetic contract of the reference build (Makefile:11,
 * x86-64, no FMA contraction) is kept: every expression below is written in
 * the evaluation order of the reference line it cites.
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>
#include "fea_oracle.h"

#define ORC_EPS 2.2204460492503131e-16            /* defines.h:12-14        */
#define ORC_EQUAL(x, y) \
  ((fabs((x) - (y)) <= fmax(fabs((x)), fabs((y))) * ORC_EPS) ? 1 : 0) /* :50 */
#define ORC_DELTA(i, j) ((i) == (j) ? 1 : 0)      /* defines.h:53           */

/* ======================================================================== */
/* dense_matrix.c                                                           */

/* dense_matrix.c:16-23 */
double orc_cdot(const double *a, const double *b, int n)
{
  double r = 0;
  int i;
  for (i = 0; i < n; ++i)
    r += a[i] * b[i];
  return r;
}

/* dense_matrix.c:25-32 */
double orc_det3x3(double m[3][3])
{
  double r;
  r = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) -
      m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
      m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
  return r;
}

/* dense_matrix.c:34-60: adjugate over det, nine separate divisions */
int orc_inv3x3(double m[3][3], double *det)
{
  double a00, a01, a02, a10, a11, a12, a20, a21, a22;
  *det = orc_det3x3(m);
  if (ORC_EQUAL(*det, 0.0))
    return 0;
  a00 = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / (*det);
  a01 = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / (*det);
  a02 = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / (*det);
  a10 = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) / (*det);
  a11 = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / (*det);
  a12 = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / (*det);
  a20 = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / (*det);
  a21 = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / (*det);
  a22 = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / (*det);
  m[0][0] = a00; m[0][1] = a01; m[0][2] = a02;
  m[1][0] = a10; m[1][1] = a11; m[1][2] = a12;
  m[2][0] = a20; m[2][1] = a21; m[2][2] = a22;
  return 1;
}

/* dense_matrix.c:62-76  R = A B */
void orc_mul3x3(double A[3][3], double B[3][3], double R[3][3])
{
  int i, j, k;
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      double sum = 0.0;
      for (k = 0; k < 3; ++k)
        sum += A[i][k] * B[k][j];
      R[i][j] = sum;
    }
}

/* dense_matrix.c:79-93  R = A' B */
void orc_tmul3x3(double A[3][3], double B[3][3], double R[3][3])
{
  int i, j, k;
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      double sum = 0.0;
      for (k = 0; k < 3; ++k)
        sum += A[k][i] * B[k][j];
      R[i][j] = sum;
    }
}

/* dense_matrix.c:96-110  R = A B' */
void orc_mult3x3(double A[3][3], double B[3][3], double R[3][3])
{
  int i, j, k;
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      double sum = 0.0;
      for (k = 0; k < 3; ++k)
        sum += A[i][k] * B[j][k];
      R[i][j] = sum;
    }
}

/* ======================================================================== */
/* fea_model.c                                                              */

/* fea_model.c:26-77 */
static void stress_A5(const double *par, double F[3][3], double S[3][3])
{
  int i, j, k;
  double C[3][3], G[3][3], Sn[3][3];
  double lambda = par[0], mu = par[1];
  double detF, I1 = 0;
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      G[i][j] = 0;
      for (k = 0; k < 3; ++k)
        G[i][j] += F[k][i] * F[k][j];
    }
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j)
      C[i][j] = 0.5 * (G[i][j] - ORC_DELTA(i, j));
  for (i = 0; i < 3; ++i)
    I1 += C[i][i];
  detF = orc_det3x3(F);
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j)
      Sn[i][j] = (lambda * I1 * ORC_DELTA(i, j) + 2 * mu * C[i][j]) / detF;
  orc_mul3x3(F, Sn, C);
  orc_mult3x3(C, F, S);
}

/* fea_model.c:79-107 */
static void stress_neohookean(const double *par, double F[3][3], double S[3][3])
{
  int i, j, k;
  double B[3][3];
  double J = orc_det3x3(F);
  double lambda = par[0], mu = par[1];
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      B[i][j] = 0;
      for (k = 0; k < 3; ++k)
        B[i][j] += F[i][k] * F[j][k];
    }
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j)
      S[i][j] = mu * (B[i][j] - ORC_DELTA(i, j)) / J +
                lambda * log(J) * ORC_DELTA(i, j) / J;
}

/* fea_model.c:110-127 */
static void ctensor_A5(const double *par, double F[3][3], double c[3][3][3][3])
{
  int i, j, k, l;
  double detF = orc_det3x3(F);
  double lambda = par[0], mu = par[1];
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j)
      for (k = 0; k < 3; ++k)
        for (l = 0; l < 3; ++l)
          c[i][j][k][l] = (lambda * ORC_DELTA(i, j) * ORC_DELTA(k, l) +
                           mu * ORC_DELTA(i, k) * ORC_DELTA(j, l) +
                           mu * ORC_DELTA(i, l) * ORC_DELTA(j, k)) / detF;
}

/* fea_model.c:129-148 */
static void ctensor_neohookean(const double *par, double F[3][3],
                               double c[3][3][3][3])
{
  int i, j, k, l;
  double J = orc_det3x3(F);
  double lambda = par[0], mu = par[1];
  double lambda1 = lambda / J;
  double mu1 = (mu - lambda * log(J)) / J;
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j)
      for (k = 0; k < 3; ++k)
        for (l = 0; l < 3; ++l)
          c[i][j][k][l] = lambda1 * ORC_DELTA(i, j) * ORC_DELTA(k, l) +
                          2 * mu1 * ORC_DELTA(i, k) * ORC_DELTA(j, l);
}

/* fea_model.c:7-23 dispatch */
void orc_stress(int model, const double *par, double F[3][3], double S[3][3])
{
  if (model == ORC_MODEL_A5) stress_A5(par, F, S);
  else stress_neohookean(par, F, S);
}

void orc_ctensor(int model, const double *par, double F[3][3],
                 double c[3][3][3][3])
{
  if (model == ORC_MODEL_A5) ctensor_A5(par, F, c);
  else ctensor_neohookean(par, F, c);
}

/* ======================================================================== */
/* element tables                                                           */

/* fea_solver.c:1287-1304 */
static double tet10_form(int i, double r, double s, double t)
{
  switch (i) {
  case 0: return (2 * (1 - r - s - t) - 1) * (1 - r - s - t);
  case 1: return (2 * r - 1) * r;
  case 2: return (2 * s - 1) * s;
  case 3: return (2 * t - 1) * t;
  case 4: return 4 * r * (1 - r - s - t);
  case 5: return 4 * r * s;
  case 6: return 4 * s * (1 - r - s - t);
  case 7: return 4 * t * (1 - r - s - t);
  case 8: return 4 * r * t;
  case 9: return 4 * s * t;
  }
  return 0;
}

/* fea_solver.c:1306-1373 */
static double tet10_dform(int i, int d, double r, double s, double t)
{
  if (d == 0) {
    switch (i) {
    case 0: return 4 * t + 4 * s + 4 * r - 3;
    case 1: return 4 * r - 1;
    case 4: return -4 * t - 4 * s - 8 * r + 4;
    case 5: return 4 * s;
    case 6: return -4 * s;
    case 7: return -4 * t;
    case 8: return 4 * t;
    default: return 0;
    }
  } else if (d == 1) {
    switch (i) {
    case 0: return 4 * t + 4 * s + 4 * r - 3;
    case 2: return 4 * s - 1;
    case 4: return -4 * r;
    case 5: return 4 * r;
    case 6: return -4 * t - 8 * s - 4 * r + 4;
    case 7: return -4 * t;
    case 9: return 4 * t;
    default: return 0;
    }
  } else {
    switch (i) {
    case 0: return 4 * t + 4 * s + 4 * r - 3;
    case 3: return 4 * t - 1;
    case 4: return -4 * r;
    case 6: return -4 * s;
    case 7: return -8 * t - 4 * s - 4 * r + 4;
    case 8: return 4 * r;
    case 9: return 4 * s;
    default: return 0;
    }
  }
}

/* linear tetrahedron: build extension, not in the reference (the enum name
 * is commented out at fea_solver.h:69); same node order as TET10 corners   */
static double tet4_form(int i, double r, double s, double t)
{
  switch (i) {
  case 0: return 1 - r - s - t;
  case 1: return r;
  case 2: return s;
  case 3: return t;
  }
  return 0;
}

static double tet4_dform(int i, int d, double r, double s, double t)
{
  (void)r; (void)s; (void)t;
  if (i == 0) return -1;
  return (i - 1 == d) ? 1 : 0;
}


/* trilinear 8-node hexahedron on the unit cube (r,s,t) in [0,1]^3: build
 * extension (BASELINE.json's "synthetic hex/tet meshes"; the reference has
 * tetrahedra only, fea_solver.h:68-71).  Node k sits at corner
 * (cx,cy,cz)[k], the usual counter-clockwise bottom face then top face; it
 * goes through the same generic loops (isoform / disoform tables,
 * fea_solver.c:503-535), with the 2 x 2 x 2 Gauss rule (weights 1/8: the
 * reference's rules carry the volume of the parent element too, :26-28).    */
static const int hex8_corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};

static double hex8_form(int i, double r, double s, double t)
{
  const double c[3] = {r, s, t};
  double v = 1;
  int d;
  for (d = 0; d < 3; ++d) v *= hex8_corner[i][d] ? c[d] : 1 - c[d];
  return v;
}

static double hex8_dform(int i, int dd, double r, double s, double t)
{
  const double c[3] = {r, s, t};
  double v = 1;
  int d;
  for (d = 0; d < 3; ++d)
    v *= d == dd ? (hex8_corner[i][d] ? 1.0 : -1.0) : (hex8_corner[i][d] ? c[d] : 1 - c[d]);
  return v;
}

static void rule8_orc(double rule[27][4])
{
  const double a = 0.5 - 0.28867513459481287, b = 0.5 + 0.28867513459481287;      /* (1 -+ 1/sqrt 3)/2 */
  int n;
  for (n = 0; n < 8; ++n) {
    rule[n][0] = 1. / 8.;
    rule[n][1] = (n & 1) ? b : a; rule[n][2] = (n & 2) ? b : a; rule[n][3] = (n & 4) ? b : a;
  }
}

/* 27-point rule (3 x 3 x 3 Gauss-Legendre on the unit cube collapsed onto the
 * tetrahedron: r = u, s = v(1-u), t = w(1-u)(1-v), Jacobian (1-u)^2 (1-v)).
 * Not in the reference (it stops at 5 points, fea_solver.c:1495-1504);
 * BASELINE.json config 5 asks for it.  Weights sum to 1/6 like the
 * reference's rules ("divisor 6 already taken into account", :26-28).       */
static void rule27_orc(double rule[27][4])
{
  static const double gx[3] = {0.5 - 0.38729833462074170, 0.5, 0.5 + 0.38729833462074170};   /* (1 -+ sqrt(3/5))/2 */
  static const double gw[3] = {5. / 18., 8. / 18., 5. / 18.};
  int a, b, c, n = 0;
  for (a = 0; a < 3; ++a)
    for (b = 0; b < 3; ++b)
      for (c = 0; c < 3; ++c, ++n) {
        const double u = gx[a], v = gx[b], w = gx[c];
        rule[n][0] = gw[a] * gw[b] * gw[c] * (1 - u) * (1 - u) * (1 - v);
        rule[n][1] = u;
        rule[n][2] = v * (1 - u);
        rule[n][3] = w * (1 - u) * (1 - v);
      }
}

int orc_elem_table_init(orc_elem_table *tb, int kind, int ngauss)
{
  /* {weight, r, s, t}; the 8-digit literals are the reference's own
   * (fea_solver.c:32-54), not (5+-sqrt5)/20                                */
  double g4[4][4] = {{(1 / 4.) / 6., 0.58541020, 0.13819660, 0.13819660},
                     {(1 / 4.) / 6., 0.13819660, 0.58541020, 0.13819660},
                     {(1 / 4.) / 6., 0.13819660, 0.13819660, 0.58541020},
                     {(1 / 4.) / 6., 0.13819660, 0.13819660, 0.13819660}};
  double g5[5][4] = {{(-4 / 5.) / 6., 1 / 4., 1 / 4., 1 / 4.},
                     {(9 / 20.) / 6., 1 / 2., 1 / 6., 1 / 6.},
                     {(9 / 20.) / 6., 1 / 6., 1 / 2., 1 / 6.},
                     {(9 / 20.) / 6., 1 / 6., 1 / 6., 1 / 2.},
                     {(9 / 20.) / 6., 1 / 6., 1 / 6., 1 / 6.}};
  double g1[1][4] = {{1 / 6., 1 / 4., 1 / 4., 1 / 4.}};
  double g27[27][4];
  double (*gd)[4];
  int g, i, j;
  memset(tb, 0, sizeof(*tb));
  if (kind == ORC_TET10) {
    tb->npe = 10;
    if (ngauss == 4) gd = g4;
    else if (ngauss == 5) gd = g5;
    else if (ngauss == 27) { rule27_orc(g27); gd = g27; }   /* extension */
    else return -1;           /* fea_solver.c:1495-1504 */
  } else if (kind == ORC_TET4) {
    tb->npe = 4;
    if (ngauss == 1) gd = g1;
    else if (ngauss == 4) gd = g4;
    else if (ngauss == 5) gd = g5;
    else return -1;
  } else if (kind == ORC_HEX8) {
    tb->npe = 8;
    if (ngauss == 8) { rule8_orc(g27); gd = g27; }
    else return -1;
  } else
    return -1;
  tb->ngauss = ngauss;
  /* fea_solver.c:515-531 */
  for (g = 0; g < ngauss; ++g) {
    double r = gd[g][1], s = gd[g][2], t = gd[g][3];
    tb->weight[g] = gd[g][0];
    for (i = 0; i < tb->npe; ++i) {
      tb->forms[g][i] = kind == ORC_TET10 ? tet10_form(i, r, s, t)
                        : kind == ORC_HEX8 ? hex8_form(i, r, s, t) : tet4_form(i, r, s, t);
      for (j = 0; j < 3; ++j)
        tb->dforms[g][j][i] = kind == ORC_TET10 ? tet10_dform(i, j, r, s, t)
                              : kind == ORC_HEX8 ? hex8_dform(i, j, r, s, t) : tet4_dform(i, j, r, s, t);
    }
  }
  return 0;
}

/* ======================================================================== */
/* solver object                                                            */

struct orc_solver {
  int N, E, npe, G, ndof;
  orc_elem_table tb;
  int *conn;          /* [E][npe]                                           */
  double *X0, *x;     /* [N][3]  nodes0_p / nodes_p (fea_solver.c:399-400)  */
  int model;
  double params[10];
  int n_bc;
  orc_bc_node *bc;
  /* per (e,g) state: shape_gradients / graddefs / stresses                 */
  double *grads;      /* [E][G][3][npe]                                     */
  double *detJ;       /* [E][G]                                             */
  char *have_grads;   /* [E][G]  the reference's "pointer is non-null"      */
  double *graddefs;   /* [E][G][9]                                          */
  double *stresses;   /* [E][G][9]                                          */
  /* global_mtx (full symmetric pattern, sorted columns)                    */
  int nnz;
  int *offsets, *indexes;
  double *values, *stash;
  double *forces, *solution;
  /* skyline factor cache for the direct solver                             */
  int *sky_first;
  long *sky_ptr;
  double *sky;
};

static int cmp_int(const void *a, const void *b)
{
  int x = *(const int *)a, y = *(const int *)b;
  return (x > y) - (x < y);
}

static void build_pattern(orc_solver *s)
{
  int N = s->N, E = s->E, npe = s->npe;
  int *cnt = (int *)calloc((size_t)N + 1, sizeof(int));
  int *inc, *fill, *nb_ptr, *nb, *tmp;
  int e, k, a, i, j;
  long tot;
  for (e = 0; e < E; ++e)
    for (k = 0; k < npe; ++k)
      cnt[s->conn[(size_t)e * npe + k] + 1]++;
  for (a = 0; a < N; ++a) cnt[a + 1] += cnt[a];
  inc = (int *)malloc(sizeof(int) * (size_t)(cnt[N] > 0 ? cnt[N] : 1));
  fill = (int *)calloc((size_t)N, sizeof(int));
  for (e = 0; e < E; ++e)
    for (k = 0; k < npe; ++k) {
      a = s->conn[(size_t)e * npe + k];
      inc[cnt[a] + fill[a]++] = e;
    }
  nb_ptr = (int *)calloc((size_t)N + 1, sizeof(int));
  tmp = (int *)malloc(sizeof(int) * (size_t)npe * 4096);
  /* two passes: count unique neighbours, then store */
  nb = NULL;
  for (i = 0; i < 2; ++i) {
    for (a = 0; a < N; ++a) {
      int m = 0, u = 0, ne = cnt[a + 1] - cnt[a];
      if (ne > 4096) ne = 4096;
      for (j = 0; j < ne; ++j) {
        e = inc[cnt[a] + j];
        for (k = 0; k < npe; ++k) tmp[m++] = s->conn[(size_t)e * npe + k];
      }
      if (m == 0) tmp[m++] = a;   /* isolated node keeps a diagonal */
      qsort(tmp, (size_t)m, sizeof(int), cmp_int);
      for (j = 0; j < m; ++j)
        if (j == 0 || tmp[j] != tmp[j - 1]) {
          if (i == 1) nb[nb_ptr[a] + u] = tmp[j];
          u++;
        }
      if (i == 0) nb_ptr[a + 1] = u;
    }
    if (i == 0) {
      for (a = 0; a < N; ++a) nb_ptr[a + 1] += nb_ptr[a];
      nb = (int *)malloc(sizeof(int) * (size_t)nb_ptr[N]);
    }
  }
  tot = (long)nb_ptr[N] * 9;
  s->nnz = (int)tot;
  s->offsets = (int *)malloc(sizeof(int) * ((size_t)s->ndof + 1));
  s->indexes = (int *)malloc(sizeof(int) * (size_t)tot);
  s->offsets[0] = 0;
  for (a = 0; a < N; ++a) {
    int len = nb_ptr[a + 1] - nb_ptr[a];
    for (i = 0; i < 3; ++i) {
      int row = a * 3 + i;
      int *dst = s->indexes + s->offsets[row];
      for (j = 0; j < len; ++j) {
        int b = nb[nb_ptr[a] + j];
        dst[3 * j + 0] = 3 * b + 0;
        dst[3 * j + 1] = 3 * b + 1;
        dst[3 * j + 2] = 3 * b + 2;
      }
      s->offsets[row + 1] = s->offsets[row] + 3 * len;
    }
  }
  free(cnt); free(inc); free(fill); free(nb_ptr); free(nb); free(tmp);
}

orc_solver *orc_solver_create(int n_nodes, int n_elems, int kind, int ngauss,
                              const int *conn, const double *X0,
                              int model, const double *params,
                              int n_bc, const orc_bc_node *bc)
{
  orc_solver *s = (orc_solver *)calloc(1, sizeof(orc_solver));
  size_t eg;
  if (orc_elem_table_init(&s->tb, kind, ngauss) != 0) { free(s); return NULL; }
  s->N = n_nodes; s->E = n_elems; s->npe = s->tb.npe; s->G = ngauss;
  s->ndof = 3 * n_nodes;
  s->conn = (int *)malloc(sizeof(int) * (size_t)n_elems * s->npe);
  memcpy(s->conn, conn, sizeof(int) * (size_t)n_elems * s->npe);
  s->X0 = (double *)malloc(sizeof(double) * 3 * (size_t)n_nodes);
  s->x = (double *)malloc(sizeof(double) * 3 * (size_t)n_nodes);
  memcpy(s->X0, X0, sizeof(double) * 3 * (size_t)n_nodes);
  memcpy(s->x, X0, sizeof(double) * 3 * (size_t)n_nodes); /* :400 copy */
  s->model = model;
  memset(s->params, 0, sizeof(s->params));
  s->params[0] = params[0]; s->params[1] = params[1];
  s->n_bc = n_bc;
  s->bc = (orc_bc_node *)malloc(sizeof(orc_bc_node) * (size_t)(n_bc > 0 ? n_bc : 1));
  if (n_bc > 0) memcpy(s->bc, bc, sizeof(orc_bc_node) * (size_t)n_bc);
  eg = (size_t)n_elems * ngauss;
  s->grads = (double *)calloc(eg * 3 * s->npe, sizeof(double));
  s->detJ = (double *)calloc(eg, sizeof(double));
  s->have_grads = (char *)calloc(eg, 1);
  s->graddefs = (double *)calloc(eg * 9, sizeof(double));   /* :431-436 zero */
  s->stresses = (double *)calloc(eg * 9, sizeof(double));
  build_pattern(s);
  s->values = (double *)calloc((size_t)s->nnz, sizeof(double));
  s->stash = NULL;
  s->forces = (double *)calloc((size_t)s->ndof, sizeof(double));   /* :453 */
  s->solution = (double *)calloc((size_t)s->ndof, sizeof(double)); /* :454 */
  return s;
}

void orc_solver_free(orc_solver *s)
{
  if (!s) return;
  free(s->conn); free(s->X0); free(s->x); free(s->bc);
  free(s->grads); free(s->detJ); free(s->have_grads);
  free(s->graddefs); free(s->stresses);
  free(s->offsets); free(s->indexes); free(s->values); free(s->stash);
  free(s->forces); free(s->solution);
  free(s->sky_first); free(s->sky_ptr); free(s->sky);
  free(s);
}

void orc_set_nodes(orc_solver *s, const double *x)
{ memcpy(s->x, x, sizeof(double) * 3 * (size_t)s->N); }
void orc_get_nodes(const orc_solver *s, double *x)
{ memcpy(x, s->x, sizeof(double) * 3 * (size_t)s->N); }
const double *orc_grads(const orc_solver *s) { return s->grads; }
const double *orc_detj(const orc_solver *s) { return s->detJ; }
const double *orc_graddefs(const orc_solver *s) { return s->graddefs; }
const double *orc_stresses(const orc_solver *s) { return s->stresses; }
int orc_nnz(const orc_solver *s) { return s->nnz; }
const int *orc_offsets(const orc_solver *s) { return s->offsets; }
const int *orc_indexes(const orc_solver *s) { return s->indexes; }
double *orc_values(orc_solver *s) { return s->values; }
double *orc_forces(orc_solver *s) { return s->forces; }
double *orc_solution(orc_solver *s) { return s->solution; }

/* ======================================================================== */
/* per-(e,g) state                                                          */

/* fea_solver.c:656-722 for one (e,g); returns 0 when det J == 0 exactly
 * (then the stored gradient is left as it was, :808-825)                   */
static int shape_gradients_eg(orc_solver *s, const double *nodes, int e, int g)
{
  int npe = s->npe, i, j, k;
  double J[3][3], detJ;
  const int *c = s->conn + (size_t)e * npe;
  double *gr = s->grads + ((size_t)e * s->G + g) * 3 * npe;
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      J[i][j] = 0;
      for (k = 0; k < npe; ++k)                                 /* :693-695 */
        J[i][j] += s->tb.dforms[g][i][k] * nodes[3 * (size_t)c[k] + j];
    }
  if (!orc_inv3x3(J, &detJ))                                    /* :697     */
    return 0;
  s->detJ[(size_t)e * s->G + g] = detJ;                         /* :709     */
  for (i = 0; i < 3; ++i)
    for (j = 0; j < npe; ++j) {
      double acc = 0;                                           /* :706     */
      for (k = 0; k < 3; ++k)                                   /* :714-718 */
        acc += J[i][k] * s->tb.dforms[g][k][j];
      gr[i * npe + j] = acc;
    }
  s->have_grads[(size_t)e * s->G + g] = 1;
  return 1;
}

/* fea_solver.c:1131-1152 (the CURRENT_SHAPE_GRADIENTS branch the Makefile
 * builds) followed by model.stress (:1187)                                 */
static void graddef_stress_eg(orc_solver *s, int e, int g)
{
  int npe = s->npe, i, j, k;
  const int *c = s->conn + (size_t)e * npe;
  const double *gr = s->grads + ((size_t)e * s->G + g) * 3 * npe;
  double (*F)[3] = (double (*)[3])(s->graddefs + ((size_t)e * s->G + g) * 9);
  double (*S)[3] = (double (*)[3])(s->stresses + ((size_t)e * s->G + g) * 9);
  double detF = 0;
  for (i = 0; i < 3; ++i)
    for (j = 0; j < 3; ++j) {
      F[i][j] = 0;
      for (k = 0; k < npe; ++k)
        F[i][j] += gr[j * npe + k] * s->X0[3 * (size_t)c[k] + i];
    }
  orc_inv3x3(F, &detF);
  orc_stress(s->model, s->params, F, S);
}

int orc_update_state(orc_solver *s)
{
  int e, g, singular = 0;
  for (e = 0; e < s->E; ++e)                                    /* :794-827 */
    for (g = 0; g < s->G; ++g)
      if (!shape_gradients_eg(s, s->x, e, g)) singular++;
  for (e = 0; e < s->E; ++e)                                    /* :847-860 */
    for (g = 0; g < s->G; ++g)
      graddef_stress_eg(s, e, g);
  return singular;
}

/* ======================================================================== */
/* element integrals                                                        */

/* value of one (g,a,b,i,j) constitutive term, fea_solver.c:942-960 */
static double kc_term(const double c4[3][3][3][3], const double *gr, int npe,
                      int a, int b, int i, int j, double detJ, double w)
{
  int k, l;
  double sum = 0.0;
  for (k = 0; k < 3; ++k)
    for (l = 0; l < 3; ++l) {
      double cikjl = (c4[i][k][j][l] + c4[i][k][l][j] +
                      c4[k][i][j][l] + c4[k][i][l][j]) / 4.;
      sum += gr[k * npe + a] * cikjl * gr[l * npe + b];
    }
  sum *= fabs(detJ);
  sum *= w;
  return sum;
}

/* one (g,a,b,i,j) initial-stress term, fea_solver.c:1033-1049 */
static double ks_term(const double *sig, const double *gr, int npe,
                      int a, int b, int i, int j, double detJ, double w)
{
  int k, l;
  double sum = 0.0;
  for (k = 0; k < 3; ++k)
    for (l = 0; l < 3; ++l)
      sum += gr[k * npe + a] * sig[3 * k + l] * gr[l * npe + b] *
             ORC_DELTA(i, j);
  sum *= fabs(detJ);
  sum *= w;
  return sum;
}

void orc_element_stiffness(const orc_solver *s, int e, double *Kc, double *Ks)
{
  int npe = s->npe, n3 = 3 * npe, g, a, b, i, j;
  double c4[3][3][3][3];
  memset(Kc, 0, sizeof(double) * (size_t)n3 * n3);
  memset(Ks, 0, sizeof(double) * (size_t)n3 * n3);
  for (g = 0; g < s->G; ++g) {
    size_t eg = (size_t)e * s->G + g;
    const double *gr = s->grads + eg * 3 * npe;
    double (*F)[3] = (double (*)[3])(s->graddefs + eg * 9);
    if (!s->have_grads[eg]) continue;
    orc_ctensor(s->model, s->params, F, c4);
    for (a = 0; a < npe; ++a)
      for (b = 0; b < npe; ++b)
        for (i = 0; i < 3; ++i)
          for (j = 0; j < 3; ++j) {
            Kc[(a * 3 + i) * n3 + b * 3 + j] +=
                kc_term((const double (*)[3][3][3])c4, gr, npe, a, b, i, j,
                        s->detJ[eg], s->tb.weight[g]);
            Ks[(a * 3 + i) * n3 + b * 3 + j] +=
                ks_term(s->stresses + eg * 9, gr, npe, a, b, i, j,
                        s->detJ[eg], s->tb.weight[g]);
          }
  }
}

void orc_element_residual(const orc_solver *s, int e, double *fe)
{
  int npe = s->npe, g, a, i, j;
  memset(fe, 0, sizeof(double) * 3 * (size_t)npe);
  for (g = 0; g < s->G; ++g) {
    size_t eg = (size_t)e * s->G + g;
    const double *gr = s->grads + eg * 3 * npe;
    const double *sig = s->stresses + eg * 9;
    if (!s->have_grads[eg]) continue;
    for (a = 0; a < npe; ++a)
      for (i = 0; i < 3; ++i) {
        double sum = 0.0;
        for (j = 0; j < 3; ++j)
          sum += sig[3 * i + j] * gr[j * npe + a];
        sum *= fabs(s->detJ[eg]);
        sum *= s->tb.weight[g];
        fe[a * 3 + i] += -sum;
      }
  }
}

/* ======================================================================== */
/* sparse accumulation: libspmatrix sp_matrix_element_add, restated as a    */
/* plain += in call order (PARITY UNPINNED: library not in the tree)        */

static int find_entry(const orc_solver *s, int row, int col)
{
  int lo = s->offsets[row], hi = s->offsets[row + 1] - 1;
  while (lo <= hi) {
    int mid = (lo + hi) >> 1;
    int c = s->indexes[mid];
    if (c == col) return mid;
    if (c < col) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

static void element_add(orc_solver *s, int row, int col, double v)
{
  int p = find_entry(s, row, col);
  if (p >= 0) s->values[p] += v;
}

/* fea_solver.c:887-983 */
static void local_constitutive_part(orc_solver *s, int e)
{
  int npe = s->npe, g, a, b, i, j;
  const int *c = s->conn + (size_t)e * npe;
  double c4[3][3][3][3];
  for (g = 0; g < s->G; ++g) {
    size_t eg = (size_t)e * s->G + g;
    const double *gr = s->grads + eg * 3 * npe;
    double (*F)[3] = (double (*)[3])(s->graddefs + eg * 9);
    orc_ctensor(s->model, s->params, F, c4);                    /* :921-923 */
    if (!s->have_grads[eg]) continue;                           /* :926     */
    for (a = 0; a < npe; ++a)
      for (b = 0; b < npe; ++b)
        for (i = 0; i < 3; ++i)
          for (j = 0; j < 3; ++j)
            element_add(s, c[a] * 3 + i, c[b] * 3 + j,          /* :964-969 */
                        kc_term((const double (*)[3][3][3])c4, gr, npe,
                                a, b, i, j, s->detJ[eg], s->tb.weight[g]));
  }
}

/* fea_solver.c:986-1068 */
static void local_initial_stress_part(orc_solver *s, int e)
{
  int npe = s->npe, g, a, b, i, j;
  const int *c = s->conn + (size_t)e * npe;
  for (g = 0; g < s->G; ++g) {
    size_t eg = (size_t)e * s->G + g;
    const double *gr = s->grads + eg * 3 * npe;
    if (!s->have_grads[eg]) continue;                           /* :1017    */
    for (a = 0; a < npe; ++a)
      for (b = 0; b < npe; ++b)
        for (i = 0; i < 3; ++i)
          for (j = 0; j < 3; ++j)
            element_add(s, c[a] * 3 + i, c[b] * 3 + j,
                        ks_term(s->stresses + eg * 9, gr, npe, a, b, i, j,
                                s->detJ[eg], s->tb.weight[g]));
  }
}

/* fea_solver.c:873-883 */
void orc_create_stiffness(orc_solver *s)
{
  int e;
  memset(s->values, 0, sizeof(double) * (size_t)s->nnz);  /* sp_matrix_clear */
  for (e = 0; e < s->E; ++e) {
    local_constitutive_part(s, e);
    local_initial_stress_part(s, e);
  }
}

/* fea_solver.c:863-870 + 1072-1114 */
void orc_create_residual_forces(orc_solver *s)
{
  int npe = s->npe, e, g, a, i, j;
  memset(s->forces, 0, sizeof(double) * (size_t)s->ndof);
  for (e = 0; e < s->E; ++e) {
    const int *c = s->conn + (size_t)e * npe;
    for (g = 0; g < s->G; ++g) {
      size_t eg = (size_t)e * s->G + g;
      const double *gr = s->grads + eg * 3 * npe;
      const double *sig = s->stresses + eg * 9;
      if (!s->have_grads[eg]) continue;
      for (a = 0; a < npe; ++a)
        for (i = 0; i < 3; ++i) {
          double sum = 0.0;
          for (j = 0; j < 3; ++j)
            sum += sig[3 * i + j] * gr[j * npe + a];
          sum *= fabs(s->detJ[eg]);
          sum *= s->tb.weight[g];
          s->forces[c[a] * 3 + i] += -sum;                      /* :1108-9  */
        }
    }
  }
}

/* ------------------------------------------------------------------------ */
/* All host cores on ONE problem ("OpenMP over coloured         */
/* elements"): the reference is single-threaded by construction; this is what  */
/* its loops give when the elements are coloured so that no two elements of a  */
/* colour share a node (no two threads then touch the same matrix row or force */
/* entry) and every colour is a parallel loop.  State, stiffness and residual  */
/* of an element in one visit.  Summation order inside an entry follows the    */
/* colours, not the element numbers: equal to orc_update_state +               */
/* orc_create_stiffness + orc_create_residual_forces to rounding.  A baseline  */
/* for bench.py, not a checker.  Returns the number of colours (<= 0: failed). */
int orc_assemble_coloured(orc_solver *s, int nthreads)
{
  const int npe = s->npe, E = s->E, N = s->ndof / 3;
  int e, k, ncol = 0;
  int *colour = (int *)malloc(sizeof(int) * (size_t)E);
  /* greedy colouring: the smallest colour none of the element's nodes has seen; per node a bit set of seen colours */
  unsigned long long *seen = (unsigned long long *)calloc((size_t)N, sizeof(unsigned long long));
  if (!colour || !seen) { free(colour); free(seen); return -1; }
  for (e = 0; e < E; ++e) {
    const int *c = s->conn + (size_t)e * npe;
    unsigned long long used = 0;
    int col = 0;
    for (k = 0; k < npe; ++k) used |= seen[c[k]];
    while (col < 64 && ((used >> col) & 1ull)) ++col;
    if (col >= 64) { free(colour); free(seen); return -2; }      /* more than 64 colours: not a mesh this is meant for */
    colour[e] = col;
    for (k = 0; k < npe; ++k) seen[c[k]] |= 1ull << col;
    if (col + 1 > ncol) ncol = col + 1;
  }
  free(seen);
  /* elements grouped by colour */
  int *first = (int *)calloc((size_t)ncol + 1, sizeof(int)), *list = (int *)malloc(sizeof(int) * (size_t)E);
  if (!first || !list) { free(colour); free(first); free(list); return -1; }
  for (e = 0; e < E; ++e) first[colour[e] + 1]++;
  for (k = 0; k < ncol; ++k) first[k + 1] += first[k];
  {
    int *fill = (int *)malloc(sizeof(int) * (size_t)ncol);
    for (k = 0; k < ncol; ++k) fill[k] = first[k];
    for (e = 0; e < E; ++e) list[fill[colour[e]]++] = e;
    free(fill);
  }
  memset(s->values, 0, sizeof(double) * (size_t)s->nnz);
  memset(s->forces, 0, sizeof(double) * (size_t)s->ndof);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
  for (k = 0; k < ncol; ++k) {
    int q;
#pragma omp parallel for schedule(static)
    for (q = first[k]; q < first[k + 1]; ++q) {
      const int el = list[q];
      const int *c = s->conn + (size_t)el * npe;
      int g, a, i, j;
      for (g = 0; g < s->G; ++g) shape_gradients_eg(s, s->x, el, g);
      for (g = 0; g < s->G; ++g) graddef_stress_eg(s, el, g);
      local_constitutive_part(s, el);
      local_initial_stress_part(s, el);
      for (g = 0; g < s->G; ++g) {                              /* the residual loop of orc_create_residual_forces */
        size_t eg = (size_t)el * s->G + g;
        const double *gr = s->grads + eg * 3 * npe;
        const double *sig = s->stresses + eg * 9;
        if (!s->have_grads[eg]) continue;
        for (a = 0; a < npe; ++a)
          for (i = 0; i < 3; ++i) {
            double sum = 0.0;
            for (j = 0; j < 3; ++j) sum += sig[3 * i + j] * gr[j * npe + a];
            sum *= fabs(s->detJ[eg]);
            sum *= s->tb.weight[g];
            s->forces[c[a] * 3 + i] += -sum;
          }
      }
    }
  }
  free(colour); free(first); free(list);
  return ncol;
}

/* ======================================================================== */
/* boundary conditions and node updates                                     */

typedef void (*apply_fn)(orc_solver *s, int index, double arg);

/* fea_solver.c:1205-1242: deck order, x then y then z of each node */
static void apply_bc_general(orc_solver *s, apply_fn apply, double lambda)
{
  int i, j;
  for (i = 0; i < s->n_bc; ++i) {
    double presc[3];
    int type = s->bc[i].type, node = s->bc[i].node;
    for (j = 0; j < 3; ++j)
      presc[j] = s->bc[i].values[j] * lambda;
    if (type == 1 || type == 3 || type == 5 || type == 7)
      apply(s, node * 3 + 0, presc[0]);
    if (type == 2 || type == 3 || type == 6 || type == 7)
      apply(s, node * 3 + 1, presc[1]);
    if (type == 4 || type == 5 || type == 6 || type == 7)
      apply(s, node * 3 + 2, presc[2]);
  }
}

/* fea_solver.c:1244-1257.  The reference walks column `index` of its CCS
 * store; with the full symmetric pattern the stored rows of that column are
 * the stored columns of row `index`, and entry (r,index) is looked up as
 * such (it may differ from (index,r) in the last bit).
 * sp_matrix_cross_cancellation: zero row and column, keep and return the
 * diagonal (PARITY UNPINNED, semantics taken from its use at :1254-1256).  */
static void apply_single_bc(orc_solver *s, int index, double presc)
{
  int p, q;
  double diag = 0;
  for (p = s->offsets[index]; p < s->offsets[index + 1]; ++p) {
    int r = s->indexes[p];
    q = find_entry(s, r, index);
    if (q >= 0) s->forces[r] -= s->values[q] * presc;
  }
  for (p = s->offsets[index]; p < s->offsets[index + 1]; ++p) {
    int r = s->indexes[p];
    if (r == index) { diag = s->values[p]; continue; }
    s->values[p] = 0;
    q = find_entry(s, r, index);
    if (q >= 0) s->values[q] = 0;
  }
  s->forces[index] = diag * presc;
}

/* fea_solver.c:1259-1266 */
static void update_node_with_bc(orc_solver *s, int index, double value)
{
  s->x[(size_t)(index / 3) * 3 + index % 3] += value;
}

void orc_apply_prescribed_bc(orc_solver *s, double lambda)
{ apply_bc_general(s, apply_single_bc, lambda); }

void orc_update_nodes_with_bc(orc_solver *s, double lambda)
{ apply_bc_general(s, update_node_with_bc, lambda); }

/* fea_solver.c:1270-1279 */
void orc_update_nodes_with_solution(orc_solver *s, const double *u)
{
  int i, j;
  for (i = 0; i < s->N; ++i)
    for (j = 0; j < 3; ++j)
      s->x[(size_t)i * 3 + j] += u[i * 3 + j];
}

void orc_stash_stiffness(orc_solver *s)
{
  if (!s->stash) s->stash = (double *)malloc(sizeof(double) * (size_t)s->nnz);
  memcpy(s->stash, s->values, sizeof(double) * (size_t)s->nnz);
}

void orc_restore_stiffness(orc_solver *s)
{
  if (s->stash) memcpy(s->values, s->stash, sizeof(double) * (size_t)s->nnz);
}

/* ======================================================================== */
/* linear solvers (libspmatrix stand-ins; PARITY UNPINNED for their         */
/* internals, pinned by K u = f)                                            */

void orc_spmv(const orc_solver *s, const double *x, double *y)
{
  int r, p;
  for (r = 0; r < s->ndof; ++r) {
    double acc = 0;
    for (p = s->offsets[r]; p < s->offsets[r + 1]; ++p)
      acc += s->values[p] * x[s->indexes[p]];
    y[r] = acc;
  }
}

/* Hestenes-Stiefel CG started from x0 = b (fea_solver.c:251-256 passes the
 * rhs as the start vector); loop of solver-prototype/solvers/cg.m:1-18 with
 * the stop test on ||r||_2 / ||b||_2.  jacobi != 0 adds diagonal scaling,
 * standing in for PCG_ILU.                                                 */
static int cg_solve(orc_solver *s, int jacobi, double tol, int max_iter,
                    double *res_out)
{
  int n = s->ndof, i, it = 0;
  double *b = s->forces, *x = s->solution;
  double *r = (double *)malloc(sizeof(double) * (size_t)n * 4);
  double *p = r + n, *q = p + n, *d = q + n;
  double rz, bnorm, rnorm = 0;
  for (i = 0; i < n; ++i) {
    int e = find_entry(s, i, i);
    d[i] = (jacobi && e >= 0 && s->values[e] != 0) ? 1.0 / s->values[e] : 1.0;
    x[i] = b[i];
  }
  orc_spmv(s, x, q);
  for (i = 0; i < n; ++i) r[i] = b[i] - q[i];
  for (i = 0; i < n; ++i) p[i] = d[i] * r[i];
  rz = 0; for (i = 0; i < n; ++i) rz += r[i] * p[i];
  bnorm = sqrt(orc_cdot(b, b, n));
  if (bnorm == 0) bnorm = 1;
  for (it = 0; it < max_iter; ++it) {
    double pq, alpha, rz_new, beta;
    rnorm = sqrt(orc_cdot(r, r, n));
    if (rnorm / bnorm < tol) break;
    orc_spmv(s, p, q);
    pq = orc_cdot(p, q, n);
    if (pq == 0) break;
    alpha = rz / pq;
    for (i = 0; i < n; ++i) { x[i] += alpha * p[i]; r[i] -= alpha * q[i]; }
    rz_new = 0; for (i = 0; i < n; ++i) rz_new += r[i] * d[i] * r[i];
    beta = rz_new / rz;
    for (i = 0; i < n; ++i) p[i] = d[i] * r[i] + beta * p[i];
    rz = rz_new;
  }
  if (res_out) *res_out = rnorm / bnorm;
  free(r);
  return it;
}

/* envelope (skyline) Cholesky, K = L L', rows stored from their first
 * non-zero column.  Plays the role of the CHOLESKY solver of the decks.    */
static int skyline_factor(orc_solver *s)
{
  int n = s->ndof, i, j, k;
  long tot = 0;
  free(s->sky_first); free(s->sky_ptr); free(s->sky);
  s->sky_first = (int *)malloc(sizeof(int) * (size_t)n);
  s->sky_ptr = (long *)malloc(sizeof(long) * ((size_t)n + 1));
  for (i = 0; i < n; ++i) {
    int first = i, p;
    for (p = s->offsets[i]; p < s->offsets[i + 1]; ++p)
      if (s->indexes[p] < first) { first = s->indexes[p]; break; }
    s->sky_first[i] = first;
    s->sky_ptr[i] = tot;
    tot += i - first + 1;
  }
  s->sky_ptr[n] = tot;
  s->sky = (double *)calloc((size_t)tot, sizeof(double));
  for (i = 0; i < n; ++i) {
    int p;
    double *Li = s->sky + s->sky_ptr[i] - s->sky_first[i];
    for (p = s->offsets[i]; p < s->offsets[i + 1]; ++p)
      if (s->indexes[p] <= i) Li[s->indexes[p]] = s->values[p];
  }
  for (i = 0; i < n; ++i) {
    double *Li = s->sky + s->sky_ptr[i] - s->sky_first[i];
    int fi = s->sky_first[i];
    for (j = fi; j <= i; ++j) {
      double *Lj = s->sky + s->sky_ptr[j] - s->sky_first[j];
      int fj = s->sky_first[j];
      int k0 = fi > fj ? fi : fj;
      double acc = Li[j];
      for (k = k0; k < j; ++k) acc -= Li[k] * Lj[k];
      if (j < i) Li[j] = acc / Lj[j];
      else {
        if (!(acc > 0)) return -1;
        Li[i] = sqrt(acc);
      }
    }
  }
  return 0;
}

static void skyline_solve(orc_solver *s)
{
  int n = s->ndof, i, k;
  double *x = s->solution;
  memcpy(x, s->forces, sizeof(double) * (size_t)n);
  for (i = 0; i < n; ++i) {
    double *Li = s->sky + s->sky_ptr[i] - s->sky_first[i];
    double acc = x[i];
    for (k = s->sky_first[i]; k < i; ++k) acc -= Li[k] * x[k];
    x[i] = acc / Li[i];
  }
  for (i = n - 1; i >= 0; --i) {
    double *Li = s->sky + s->sky_ptr[i] - s->sky_first[i];
    x[i] /= Li[i];
    for (k = s->sky_first[i]; k < i; ++k) x[k] -= Li[k] * x[i];
  }
}

int orc_solve_slae(orc_solver *s, int type, double tol, int max_iter,
                   double *residual_out)
{
  if (type == ORC_CHOLESKY) {
    if (skyline_factor(s) != 0) {
      if (residual_out) *residual_out = NAN;
      return -1;
    }
    skyline_solve(s);
    if (residual_out) {
      double *q = (double *)malloc(sizeof(double) * (size_t)s->ndof);
      double rn = 0, bn = 0;
      int i;
      orc_spmv(s, s->solution, q);
      for (i = 0; i < s->ndof; ++i) {
        double d = s->forces[i] - q[i];
        rn += d * d; bn += s->forces[i] * s->forces[i];
      }
      *residual_out = bn > 0 ? sqrt(rn / bn) : sqrt(rn);
      free(q);
    }
    return 0;
  }
  return cg_solve(s, type == ORC_PCG_ILU, tol, max_iter, residual_out);
}

/* ======================================================================== */
/* fea_solver.c:130-242                                                     */

int orc_solve(orc_solver *s, int load_increments, int max_newton,
              int modified_newton, double desired_tolerance,
              int slae_type, double slae_tol, int slae_max_iter,
              double *tol_log, int tol_log_cap, int *its_log)
{
  int step, nlog = 0;
  for (step = 0; step < load_increments; ++step) {
    int it = 0;
    double tolerance;
    orc_update_nodes_with_bc(s, 1);                             /* :168 */
    orc_update_state(s);                                        /* :171-174 */
    orc_create_stiffness(s);                                    /* :177 */
    orc_stash_stiffness(s);                                     /* :179 */
    do {
      it++;
      orc_create_residual_forces(s);                            /* :185 */
      if (modified_newton) orc_restore_stiffness(s);            /* :188-196 */
      else orc_create_stiffness(s);                             /* :200 */
      orc_apply_prescribed_bc(s, 0);                            /* :203 */
      orc_solve_slae(s, slae_type, slae_tol, slae_max_iter, NULL); /* :205 */
      tolerance = orc_cdot(s->forces, s->solution, s->ndof);    /* :208-210 */
      if (tol_log && nlog < tol_log_cap) tol_log[nlog] = tolerance;
      nlog++;
      orc_update_nodes_with_solution(s, s->solution);           /* :216 */
      orc_update_state(s);                                      /* :217-218 */
    } while (fabs(tolerance) > desired_tolerance && it < max_newton); /* :220 */
    if (its_log) its_log[step] = it;
    if (it == max_newton)                                       /* :225-231 */
      return step;
  }
  return step;
}
