/*
 * fea_elements.c -- host-side element plug-in: Gauss rules and shape-function
 * tables handed to the device through feahip_create().
 *
 * The reference binds isoform_t / disoform_t function pointers and a Gauss
 * table per element type (fea_solver.c:1491-1506) and tabulates them per
 * Gauss point in solver_gauss_node_alloc (:503-535).  Device code cannot
 * call host pointers, so the tabulation is the interface: weights[G],
 * forms[G][npe], dforms[G][3][npe].
 *
 * TETRAHEDRA10 with 4 or 5 points is the reference's element; its 4-point
 * rule uses the 8-digit literals of fea_solver.c:33-47 on purpose.
 * TETRAHEDRA4 (1 point) is a build extension for the BASELINE.json
 * linear-tet configurations: corner shape functions, centroid rule.
 * The 27-point rule is a build extension for BASELINE.json config 5.
 * HEXAHEDRA8 (trilinear brick on the unit cube, 2 x 2 x 2 Gauss points) is a
 * build extension for BASELINE.json's "synthetic hex/tet meshes".
 */
#include <string.h>
#include "fea_host.h"

/* local node order: corners 0-3, mid-sides 4:(0,1) 5:(1,2) 6:(0,2) 7:(0,3)
 * 8:(1,3) 9:(2,3) -- fea_solver.c:1287-1304 */
static double t10_N(int i, double r, double s, double t)
{
  const double u = 1 - r - s - t;
  switch (i) {
  case 0: return (2 * u - 1) * u;
  case 1: return (2 * r - 1) * r;
  case 2: return (2 * s - 1) * s;
  case 3: return (2 * t - 1) * t;
  case 4: return 4 * r * u;
  case 5: return 4 * r * s;
  case 6: return 4 * s * u;
  case 7: return 4 * t * u;
  case 8: return 4 * r * t;
  default: return 4 * s * t;
  }
}

/* d N_i / d(r,s,t)[d] -- fea_solver.c:1306-1373 */
static double t10_dN(int i, int d, double r, double s, double t)
{
  const double c0 = 4 * t + 4 * s + 4 * r - 3;
  static const int corner_axis[4] = {-1, 0, 1, 2};
  double v[3] = {r, s, t};
  if (i == 0) return c0;
  if (i < 4) return corner_axis[i] == d ? 4 * v[d] - 1 : 0;
  switch (i) {
  case 4: return d == 0 ? -4 * t - 4 * s - 8 * r + 4 : -4 * r;
  case 5: return d == 0 ? 4 * s : (d == 1 ? 4 * r : 0);
  case 6: return d == 1 ? -4 * t - 8 * s - 4 * r + 4 : -4 * s;
  case 7: return d == 2 ? -8 * t - 4 * s - 4 * r + 4 : -4 * t;
  case 8: return d == 0 ? 4 * t : (d == 2 ? 4 * r : 0);
  default: return d == 1 ? 4 * t : (d == 2 ? 4 * s : 0);
  }
}

static double t4_N(int i, double r, double s, double t)
{
  switch (i) {
  case 0: return 1 - r - s - t;
  case 1: return r;
  case 2: return s;
  default: return t;
  }
}

static double t4_dN(int i, int d) { return i == 0 ? -1 : (i - 1 == d ? 1 : 0); }

/* 8-node brick on (r,s,t) in [0,1]^3, node k at corner h8_corner[k] (bottom face counter-clockwise, then top) */
static const int h8_corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};

static double h8_N(int i, double r, double s, double t)
{
  const double c[3] = {r, s, t};
  double v = 1;
  int d;
  for (d = 0; d < 3; ++d) v *= h8_corner[i][d] ? c[d] : 1 - c[d];
  return v;
}

static double h8_dN(int i, int dd, double r, double s, double t)
{
  const double c[3] = {r, s, t};
  double v = 1;
  int d;
  for (d = 0; d < 3; ++d)
    v *= d == dd ? (h8_corner[i][d] ? 1.0 : -1.0) : (h8_corner[i][d] ? c[d] : 1 - c[d]);
  return v;
}


/* 27-point rule (3 x 3 x 3 Gauss-Legendre on the unit cube collapsed onto the
 * tetrahedron: r = u, s = v(1-u), t = w(1-u)(1-v), Jacobian (1-u)^2 (1-v)).
 * Not in the reference (it stops at 5 points, fea_solver.c:1495-1504);
 * BASELINE.json config 5 asks for it.  Weights sum to 1/6 like the
 * reference's rules ("divisor 6 already taken into account", :26-28).       */
static void rule27_host(double rule[27][4])
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

int fea_element_tables(int ele_type, int G, double *weights, double *forms, double *dforms)
{
  /* {weight (divisor 6 inside), r, s, t} */
  static const double a = 0.58541020, b = 0.13819660;
  double rule[27][4];
  int npe, g, i, d;
  if (ele_type == FEA_TETRAHEDRA10) npe = 10;
  else if (ele_type == FEA_TETRAHEDRA4) npe = 4;
  else if (ele_type == FEA_HEXAHEDRA8) npe = 8;
  else return -1;
  memset(rule, 0, sizeof rule);
  if (ele_type == FEA_HEXAHEDRA8) {
    const double ga = 0.5 - 0.28867513459481287, gb = 0.5 + 0.28867513459481287;   /* (1 -+ 1/sqrt 3)/2 */
    if (G != 8) return -1;
    for (g = 0; g < 8; ++g) {
      rule[g][0] = 1 / 8.;
      rule[g][1] = (g & 1) ? gb : ga; rule[g][2] = (g & 2) ? gb : ga; rule[g][3] = (g & 4) ? gb : ga;
    }
  } else if (G == 4) {
    for (g = 0; g < 4; ++g) {
      rule[g][0] = (1 / 4.) / 6.;
      rule[g][1] = g == 0 ? a : b; rule[g][2] = g == 1 ? a : b; rule[g][3] = g == 2 ? a : b;
    }
  } else if (G == 5) {
    rule[0][0] = (-4 / 5.) / 6.; rule[0][1] = rule[0][2] = rule[0][3] = 1 / 4.;
    for (g = 1; g < 5; ++g) {
      rule[g][0] = (9 / 20.) / 6.;
      rule[g][1] = g == 1 ? 1 / 2. : 1 / 6.; rule[g][2] = g == 2 ? 1 / 2. : 1 / 6.; rule[g][3] = g == 3 ? 1 / 2. : 1 / 6.;
    }
  } else if (G == 27) {
    rule27_host(rule);
  } else if (G == 1 && ele_type == FEA_TETRAHEDRA4) {
    rule[0][0] = 1 / 6.; rule[0][1] = rule[0][2] = rule[0][3] = 1 / 4.;
  } else
    return -1;                      /* fea_solver.c:1495-1504 rejects others */
  for (g = 0; g < G; ++g) {
    const double r = rule[g][1], s = rule[g][2], t = rule[g][3];
    weights[g] = rule[g][0];
    for (i = 0; i < npe; ++i) {
      if (forms) forms[g * npe + i] = npe == 10 ? t10_N(i, r, s, t) : npe == 8 ? h8_N(i, r, s, t) : t4_N(i, r, s, t);
      for (d = 0; d < 3; ++d)
        dforms[(g * 3 + d) * npe + i] = npe == 10 ? t10_dN(i, d, r, s, t) : npe == 8 ? h8_dN(i, d, r, s, t) : t4_dN(i, d);
    }
  }
  return npe;
}
