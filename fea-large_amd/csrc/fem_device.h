// fem_device.h -- per-Gauss-point device math of the hot path.
//
// What the reference evaluates in four separate sweeps over heap objects --
//   J, J^-1, grad N      solver_shape_gradients_alloc   fea_solver.c:656-722
//   F^-1 -> F            solver_element_gauss_graddef   fea_solver.c:1131-1152
//   sigma(F)             fea_model_stress_*             fea_model.c:26-107
//   c(F)                 fea_model_ctensor_*            fea_model.c:110-148
// is evaluated here in registers, once per (element, Gauss point) visit.
//
// The spatial tangent enters the stiffness only through its minor-symmetrised
// form (fea_solver.c:948-949), which for both models collapses to
//   c~_ikjl = l1 d_ik d_jl + m1 (d_ij d_kl + d_il d_kj)
// with  l1 = lambda/J, m1 = (mu - lambda ln J)/J   (Neo-Hookean, :138-146)
//       l1 = lambda/J, m1 = mu/J                   (A5, :116-126)
// so one 3x3 block of the element stiffness is
//   K_ab = w|detJ| [ l1 g_a (x) g_b + m1 g_b (x) g_a
//                    + (m1 g_a.g_b + g_a.sigma.g_b) I ]          (:944-1049)
// about 25 FMAs instead of the reference's 2 x 81-term loops.
#pragma once
#include "feahip_internal.h"

struct AsmArgs {
  int N, E, G, nchunks, chunk0, model;   // chunks [chunk0, chunk0+nchunks) are this launch's
  int row0, row1;                        // block rows this rank owns (K holds no others)
  double lambda, mu;
  const ElemTable *tab;
  const int *conn;
  const double *X0, *x;          // [N][4]
  const int *rowptr, *colidx;
  double *K;
  double *f;
  const int *incptr;
  const uint32_t *inc;
  const uint8_t *incslot;
  const int *chunk;
  const int *diag;               // [N] diagonal block of every row
  int *bad;                      // counter of Gauss points with det J <= 0
  double *Fout, *Sout;           // state export
  double *Gout, *Dout;           // shape gradients [E][G][3][npe] and det J [E][G] (null: not exported)
};

template <int NPE>
struct GPState {
  double g[NPE][3];    // spatial shape gradients  g[a][i] = dN_a/dx_i
  double sig[3][3];    // Cauchy stress
  double l1, m1;       // tangent coefficients (see header)
  double vol;          // w_g * |det J|
  double detJ;
  double F[3][3];
};

__device__ __forceinline__ double fd_det3(const double m[3][3])
{
  return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) -
         m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
         m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

// 1/x by v_rcp_f64 and two Newton steps (~1 ulp); the IEEE division sequence
// hipcc emits for 1.0/x is three times as long
__device__ __forceinline__ double fd_rcp(double x)
{
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

__device__ __forceinline__ double fd_log(double x);     // below

// adjugate / det (dense_matrix.c:34-60 divides nine times; one reciprocal
// here, the difference is 1 ulp per entry)
__device__ __forceinline__ void fd_inv3(const double m[3][3], double r[3][3], double &det)
{
  det = fd_det3(m);
  const double id = fd_rcp(det);
  r[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) * id;
  r[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
  r[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
  r[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * id;
  r[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id;
  r[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
  r[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * id;
  r[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
  r[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
}

// State of one Gauss point from the element's current (xe) and initial (Xe)
// node coordinates.  LINTET: constant-strain tetrahedron, dN/dxi is the
// fixed table {-1,1,0,0; -1,0,1,0; -1,0,0,1} and is folded into the algebra.
// NEEDF = false lets the Neo-Hookean branch skip F itself (see below).
template <int NPE, bool LINTET, bool NEEDF = true>
__device__ __forceinline__ void gp_state(const double (&xe)[NPE][3], const double (&Xe)[NPE][3],
                                         const ElemTable *tab, int gp, int model,
                                         double lambda, double mu, GPState<NPE> &s)
{
  double J[3][3], Ji[3][3];
  if constexpr (LINTET) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) J[i][j] = xe[i + 1][j] - xe[0][j];
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < NPE; ++k) acc += tab->dN[gp][i][k] * xe[k][j];
        J[i][j] = acc;
      }
  }
  fd_inv3(J, Ji, s.detJ);
  if constexpr (LINTET) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      s.g[1][i] = Ji[i][0]; s.g[2][i] = Ji[i][1]; s.g[3][i] = Ji[i][2];
      s.g[0][i] = -(Ji[i][0] + Ji[i][1] + Ji[i][2]);
    }
  } else {
#pragma unroll
    for (int a = 0; a < NPE; ++a)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc += Ji[i][k] * tab->dN[gp][k][a];
        s.g[a][i] = acc;
      }
  }
  // F^-1_ij = sum_k dN_k/dx_j X_k,i  (fea_solver.c:1141-1151), then invert
  double Fi[3][3], detFi;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double acc = 0;
      if constexpr (LINTET) {
#pragma unroll
        for (int k = 1; k < 4; ++k) acc += s.g[k][j] * (Xe[k][i] - Xe[0][i]);
      } else {
#pragma unroll
        for (int k = 0; k < NPE; ++k) acc += s.g[k][j] * Xe[k][i];
      }
      Fi[i][j] = acc;
    }
  if constexpr (!NEEDF) {
    if (model == FEAHIP_MODEL_COMPRESSIBLE_NEOHOOKEAN) {
      // sigma needs only B = F F' = (Fi' Fi)^-1 and J = 1/det Fi: invert the
      // symmetric Fi'Fi by its adjugate (det = det(Fi)^2) instead of
      // inverting Fi and multiplying out -- about half the flops, no F.
      detFi = fd_det3(Fi);
      const double Jd = fd_rcp(detFi);           // J = det F
      const double lnJ = -fd_log(detFi);
      double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        c00 += Fi[i][0] * Fi[i][0]; c01 += Fi[i][0] * Fi[i][1]; c02 += Fi[i][0] * Fi[i][2];
        c11 += Fi[i][1] * Fi[i][1]; c12 += Fi[i][1] * Fi[i][2]; c22 += Fi[i][2] * Fi[i][2];
      }
      // sigma = mu/J (B - I) + lambda lnJ/J I,  B = adj(C) J^2,  1/J = det Fi
      const double mJ = mu * Jd;                  // mu J = (mu/J) J^2
      const double dg = (mu - lambda * lnJ) * detFi;
      s.sig[0][0] = mJ * (c11 * c22 - c12 * c12) - dg;
      s.sig[1][1] = mJ * (c00 * c22 - c02 * c02) - dg;
      s.sig[2][2] = mJ * (c00 * c11 - c01 * c01) - dg;
      s.sig[0][1] = s.sig[1][0] = mJ * (c02 * c12 - c01 * c22);
      s.sig[0][2] = s.sig[2][0] = mJ * (c01 * c12 - c02 * c11);
      s.sig[1][2] = s.sig[2][1] = mJ * (c01 * c02 - c00 * c12);
      s.l1 = lambda * detFi;
      s.m1 = dg;
      s.vol = tab->w[gp] * fabs(s.detJ);
      return;
    }
  }
  fd_inv3(Fi, s.F, detFi);
  const double Jd = fd_det3(s.F);
  const double iJ = fd_rcp(Jd);
  if (model == FEAHIP_MODEL_COMPRESSIBLE_NEOHOOKEAN) {
    const double lnJ = log(Jd);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double b = s.F[i][0] * s.F[j][0] + s.F[i][1] * s.F[j][1] + s.F[i][2] * s.F[j][2];
        double d = (i == j) ? 1.0 : 0.0;
        s.sig[i][j] = (mu * (b - d) + lambda * lnJ * d) * iJ;
      }
    s.l1 = lambda * iJ;
    s.m1 = (mu - lambda * lnJ) * iJ;
  } else {
    // A5: S = (lambda tr(C) I + 2 mu C)/J, C = (F'F - I)/2, sigma = F S F'
    double Sn[3][3], T[3][3], I1 = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double gij = s.F[0][i] * s.F[0][j] + s.F[1][i] * s.F[1][j] + s.F[2][i] * s.F[2][j];
        Sn[i][j] = 0.5 * (gij - ((i == j) ? 1.0 : 0.0));
      }
    I1 = Sn[0][0] + Sn[1][1] + Sn[2][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        Sn[i][j] = (lambda * I1 * ((i == j) ? 1.0 : 0.0) + 2 * mu * Sn[i][j]) * iJ;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        T[i][j] = s.F[i][0] * Sn[0][j] + s.F[i][1] * Sn[1][j] + s.F[i][2] * Sn[2][j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        s.sig[i][j] = T[i][0] * s.F[j][0] + T[i][1] * s.F[j][1] + T[i][2] * s.F[j][2];
    s.l1 = lambda * iJ;
    s.m1 = mu * iJ;
  }
  s.vol = tab->w[gp] * fabs(s.detJ);
}

// Cauchy stress and tangent coefficients l1, m1 of the two material models from
// the deformation gradient (fea_model.c:97-150; compact forms: file header).
__device__ __forceinline__ void fd_constitutive(const double (&F)[3][3], int model, double lambda, double mu,
                                                double (&sig)[3][3], double &l1, double &m1)
{
  const double Jd = fd_det3(F);
  const double iJ = fd_rcp(Jd);
  if (model == FEAHIP_MODEL_COMPRESSIBLE_NEOHOOKEAN) {
    const double lnJ = log(Jd);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double b = F[i][0] * F[j][0] + F[i][1] * F[j][1] + F[i][2] * F[j][2];
        const double d = (i == j) ? 1.0 : 0.0;
        sig[i][j] = (mu * (b - d) + lambda * lnJ * d) * iJ;
      }
    l1 = lambda * iJ;
    m1 = (mu - lambda * lnJ) * iJ;
  } else {
    double Sn[3][3], T[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double gij = F[0][i] * F[0][j] + F[1][i] * F[1][j] + F[2][i] * F[2][j];
        Sn[i][j] = 0.5 * (gij - ((i == j) ? 1.0 : 0.0));
      }
    const double I1 = Sn[0][0] + Sn[1][1] + Sn[2][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        Sn[i][j] = (lambda * I1 * ((i == j) ? 1.0 : 0.0) + 2 * mu * Sn[i][j]) * iJ;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        T[i][j] = F[i][0] * Sn[0][j] + F[i][1] * Sn[1][j] + F[i][2] * Sn[2][j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        sig[i][j] = T[i][0] * F[j][0] + T[i][1] * F[j][1] + T[i][2] * F[j][2];
    l1 = lambda * iJ;
    m1 = mu * iJ;
  }
}

// Same state for elements with many nodes (TET10), streaming the node
// coordinates from memory inside the sums instead of holding x and X0 of all
// nodes in registers: 120 fewer VGPRs per lane (2-3x the occupancy); the
// re-reads of every Gauss point hit L1.
template <int NPE, class TAB = ElemTable>
__device__ __forceinline__ void gp_state_stream(const double *xg, const double *X0g, const int (&nd)[NPE],
                                                const TAB *tab, int gp, int model,
                                                double lambda, double mu, GPState<NPE> &s)
{
  double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Ji[3][3];
#pragma unroll
  for (int k = 0; k < NPE; ++k) {
    const double2 a0 = *reinterpret_cast<const double2 *>(xg + (size_t)nd[k] * 4);
    const double a2 = xg[(size_t)nd[k] * 4 + 2];
    const double c[3] = {a0.x, a0.y, a2};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) J[i][j] += tab->dN[gp][i][k] * c[j];
  }
  fd_inv3(J, Ji, s.detJ);
#pragma unroll
  for (int a = 0; a < NPE; ++a)
#pragma unroll
    for (int i = 0; i < 3; ++i)
      s.g[a][i] = Ji[i][0] * tab->dN[gp][0][a] + Ji[i][1] * tab->dN[gp][1][a] + Ji[i][2] * tab->dN[gp][2][a];
  double Fi[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, detFi;
#pragma unroll
  for (int k = 0; k < NPE; ++k) {
    const double2 a0 = *reinterpret_cast<const double2 *>(X0g + (size_t)nd[k] * 4);
    const double a2 = X0g[(size_t)nd[k] * 4 + 2];
    const double c[3] = {a0.x, a0.y, a2};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Fi[i][j] += s.g[k][j] * c[i];
  }
  fd_inv3(Fi, s.F, detFi);
  fd_constitutive(s.F, model, lambda, mu, s.sig, s.l1, s.m1);
  s.vol = tab->w[gp] * fabs(s.detJ);
}

// Row-node form of the same block, for kernels that hold ONE row node a and
// walk several column nodes b: with A = vol l1 g_a, B = vol m1 g_a and
// c = B + vol sigma g_a (all per visit; vol sigma g_a is also minus the
// residual contribution), K_ab[i][j] = A_i g_b[j] + g_b[i] B_j + d_ij (c . g_b):
// 21 FMAs per block instead of 42 flops.
struct RowVecs { double A[3], B[3], c[3], s[3]; };

__device__ __forceinline__ void row_vectors(const double ga[3], const double sig[3][3], double l1, double m1,
                                            double vol, RowVecs &r)
{
  const double vl = vol * l1, vm = vol * m1;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    r.A[i] = vl * ga[i];
    r.B[i] = vm * ga[i];
    // sigma is symmetric to rounding; (sigma g_a)_i as the reference's residual sums it (fea_solver.c:1096-1098)
    r.s[i] = vol * (sig[i][0] * ga[0] + sig[i][1] * ga[1] + sig[i][2] * ga[2]);
    r.c[i] = r.B[i] + r.s[i];
  }
}

__device__ __forceinline__ void block_row(const RowVecs &r, const double gb[3], double out[9])
{
  const double d = r.c[0] * gb[0] + r.c[1] * gb[1] + r.c[2] * gb[2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      out[3 * i + j] = r.A[i] * gb[j] + (gb[i] * r.B[j] + ((i == j) ? d : 0.0));
}

// ln x for positive normal x: x = m 2^e with m in [sqrt(1/2), sqrt 2),
// ln m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716, odd series to s^21
// (truncation < 1e-17 relative).  About half the instructions of the library
// log (no special-case handling: det F of a valid element is positive and
// finite; 0, negatives and NaN propagate as NaN / -inf like log would).
__device__ __forceinline__ double fd_log(double x)
{
  int e = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x);       // [0.5, 1)
  if (m < 0.70710678118654752) { m += m; e -= 1; }
  const double s = (m - 1.0) * fd_rcp(m + 1.0);
  const double z = s * s;
  // The coefficients are pinned to scalar registers: left to itself the compiler turns p = fma(p, z, c) into
  // v_fmac with c copied to a vector register, hoists the ten copies out of the caller's loop and, in a kernel
  // short of vector registers, spills them -- ten scratch reloads inside the state evaluation, each waiting for
  // every global load in flight (in-order counter).  v_fma_f64 takes a scalar pair as its addend.
#define FD_SC(name, v) double name = (v); asm("" : "+s"(name))
  FD_SC(c21, 1.0 / 21.0); FD_SC(c19, 1.0 / 19.0); FD_SC(c17, 1.0 / 17.0); FD_SC(c15, 1.0 / 15.0); FD_SC(c13, 1.0 / 13.0);
  FD_SC(c11, 1.0 / 11.0); FD_SC(c9, 1.0 / 9.0); FD_SC(c7, 1.0 / 7.0); FD_SC(c5, 1.0 / 5.0); FD_SC(c3, 1.0 / 3.0);
#undef FD_SC
  double p = c21;
  p = fma(p, z, c19); p = fma(p, z, c17); p = fma(p, z, c15);
  p = fma(p, z, c13); p = fma(p, z, c11); p = fma(p, z, c9);
  p = fma(p, z, c7);  p = fma(p, z, c5);  p = fma(p, z, c3);
  p = fma(p, z, 1.0);
  const double lnm = 2.0 * s * p;
  const double ed = (double)e;
  // e ln2 with ln2 split so the product is exact in the high part
  return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, lnm));
}

// column-node vectors of a block: h = vol l1 g_b, m = vol m1 g_b,
// t = m + vol sigma g_b
__device__ __forceinline__ void col_vectors(const double gb[3], const double sig[3][3],
                                            double l1, double m1, double vol,
                                            double h[3], double m[3], double t[3])
{
  const double vl = vol * l1, vm = vol * m1;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    h[k] = vl * gb[k];
    m[k] = vm * gb[k];
    t[k] = m[k] + vol * (sig[k][0] * gb[0] + sig[k][1] * gb[1] + sig[k][2] * gb[2]);
  }
}

// K_ab (row-major 3x3) from the row-node gradient and the column vectors
__device__ __forceinline__ void block_ab(const double ga[3], const double h[3], const double m[3],
                                         const double t[3], double out[9])
{
  const double d = ga[0] * t[0] + ga[1] * t[1] + ga[2] * t[2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      out[3 * i + j] = ga[i] * h[j] + ga[j] * m[i] + ((i == j) ? d : 0.0);
}

// rot (NPE == 4 only): local node k of the result is local node k XOR rot of
// the stored element
template <int NPE>
__device__ __forceinline__ void load_element(const AsmArgs &A, int e, int (&nd)[NPE],
                                             double (&xe)[NPE][3], double (&Xe)[NPE][3], int rot = 0)
{
  if constexpr (NPE == 4) {
    const int4 c = *reinterpret_cast<const int4 *>(A.conn + (size_t)e * 4);
    const int m[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int src = k ^ rot;
      nd[k] = (src == 0) ? m[0] : (src == 1) ? m[1] : (src == 2) ? m[2] : m[3];
    }
  } else {
#pragma unroll
    for (int k = 0; k < NPE; ++k) nd[k] = A.conn[(size_t)e * NPE + k];
  }
#pragma unroll
  for (int k = 0; k < NPE; ++k) {
    const double2 a0 = *reinterpret_cast<const double2 *>(A.x + (size_t)nd[k] * 4);
    const double2 a1 = *reinterpret_cast<const double2 *>(A.x + (size_t)nd[k] * 4 + 2);
    const double2 b0 = *reinterpret_cast<const double2 *>(A.X0 + (size_t)nd[k] * 4);
    const double2 b1 = *reinterpret_cast<const double2 *>(A.X0 + (size_t)nd[k] * 4 + 2);
    xe[k][0] = a0.x; xe[k][1] = a0.y; xe[k][2] = a1.x;
    Xe[k][0] = b0.x; Xe[k][1] = b0.y; Xe[k][2] = b1.x;
  }
}
