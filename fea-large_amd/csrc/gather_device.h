// gather_device.h -- device helpers shared by the GATHER assembly kernels of linear tetrahedra
// (kernels_gather.hip): the element record and its LDS layout, one contribution to a block,
// the map words a thread holds.  Algebra of fem_device.h; replaces fea_solver.c:873-883, :887-1068, :1072-1114
// for TETRAHEDRA4 meshes.
#pragma once
#include "fem_device.h"
// Wave priorities by phase: a workgroup closer to the end of its chunk goes first.  Three workgroups share a CU and are
// in different phases at any time; with equal priorities the latency-bound state phase of one took issue slots from the
// tile writes and row stores of another, which is what releases LDS and the barrier for the next chunk: 3 % faster.
#define PRIO_STATE 0
#define PRIO_GATHER 1
#define PRIO_OUT 3
#include <algorithm>
#include <cstdlib>
#include <cstring>

struct GatherArgs {
  int chunk0, nchunks, model;
  double lambda, mu;
  const ElemTable *tab;
  const unsigned char *maps;
  GatherLayout lay;
  const double *X0, *x;          // [N][4]
  double *K, *f;
  int *bad;
  unsigned long long *stamps;    // diagnostic build only: [chunk][8] s_memtime deltas
  int ablate;                    // diagnostic build only: timing experiments (results meaningless)
};
#if defined(FEAHIP_DEBUG) && !defined(FEAHIP_NOABL)       // FEAHIP_NOABL: stamps only, the code of the shipped kernel otherwise
#define G_ABL(bit) (A.ablate & (bit))
#else
#define G_ABL(bit) 0
#endif

#ifdef FEAHIP_DEBUG
#define G_STAMP(i) do { if (A.stamps) st[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G_STAMP(i) do { } while (0)
#endif

#define G_RS 0                        // u16 offsets inside the "rows" section: rstart[MAX_ROWS + 1]
#define G_RD 66                       // rdiag[MAX_ROWS]
#define G_VF 130                      // vfirst[MAX_ROWS + 1]
#define G_SLOT(w) ((w) & 1023u)       // contribution / visit entry: record slot | local row node << 10 | local column node << 12
#define G_LA(w) (((w) >> 10) & 3)
#define G_LB(w) (((w) >> 12) & 3)

#define GREC 26                  // doubles per element record (stiffness): g[4][3], t[4][3], vl, vm
#define GREC_F 12                // residual only: s[4][3] = vol sigma g

// Record layout (13 pieces of 16 bytes; every read of a record is one aligned ds_read_b128, and which LDS bank
// slot a piece falls into is decided by the record's slot mod 16 -- gather.cpp places the elements accordingly):
//   pieces 0-3 P_k = (g_k.x, g_k.y)   4-7 Q_k = (t_k.x, t_k.y)   8-11 Z_k = (g_k.z, t_k.z)   12 (vl, vm)
// residual-only record (6 pieces): 0-3 (s_k.x, s_k.y), 4-5 (s_0.z .. s_3.z), s = vol sigma g
//
// one contribution (element slot, local row node la, local column node lb) to the thread's block
// (an empty slot of the list points at an all-zero record: no branch)
struct GRead { double2 pa, za, pb, qb, zb, vv; };
__device__ __forceinline__ GRead g_fetch(const double *sT, unsigned w)
{
  const double2 *T = reinterpret_cast<const double2 *>(sT + G_SLOT(w) * GREC);
  const int la = G_LA(w), lb = G_LB(w);
  GRead r;
  r.pa = T[la]; r.za = T[8 + la]; r.pb = T[lb]; r.qb = T[4 + lb]; r.zb = T[8 + lb]; r.vv = T[12];
  return r;
}
// acc[0..8]: the block without the isotropic term; accd: the isotropic term sum_e g_a . t_b, kept apart (one fused chain,
// no additions onto the three diagonal entries per contribution) and added to them once, by g_finish
__device__ __forceinline__ void g_apply(const GRead &r, double (&acc)[9], double &accd)
{
  const double ga0 = r.pa.x, ga1 = r.pa.y, ga2 = r.za.x;
  const double gb0 = r.pb.x, gb1 = r.pb.y, gb2 = r.zb.x;
  const double tb0 = r.qb.x, tb1 = r.qb.y, tb2 = r.zb.y;
  const double vl = r.vv.x, vm = r.vv.y;
  const double h0 = vl * gb0, h1 = vl * gb1, h2 = vl * gb2;
  const double m0 = vm * gb0, m1 = vm * gb1, m2 = vm * gb2;
  accd = fma(ga0, tb0, fma(ga1, tb1, fma(ga2, tb2, accd)));
  acc[0] = fma(ga0, h0, fma(ga0, m0, acc[0])); acc[1] = fma(ga0, h1, fma(ga1, m0, acc[1])); acc[2] = fma(ga0, h2, fma(ga2, m0, acc[2]));
  acc[3] = fma(ga1, h0, fma(ga0, m1, acc[3])); acc[4] = fma(ga1, h1, fma(ga1, m1, acc[4])); acc[5] = fma(ga1, h2, fma(ga2, m1, acc[5]));
  acc[6] = fma(ga2, h0, fma(ga0, m2, acc[6])); acc[7] = fma(ga2, h1, fma(ga1, m2, acc[7])); acc[8] = fma(ga2, h2, fma(ga2, m2, acc[8]));
}
__device__ __forceinline__ void g_finish(double (&acc)[9], double accd)
{
  acc[0] += accd; acc[4] += accd; acc[8] += accd;
}
__device__ __forceinline__ void g_consume(const double *sT, unsigned w, double (&acc)[9], double &accd)
{
  g_apply(g_fetch(sT, w), acc, accd);
}
__device__ __forceinline__ void g_apply_cheap(const GRead &r, double (&acc)[9])
{
  acc[0] += r.pa.x; acc[1] += r.za.x; acc[2] += r.pb.x; acc[3] += r.qb.x; acc[4] += r.zb.x; acc[5] += r.vv.x;
}
__device__ __forceinline__ GRead g_fetch_cheap(const double *sT, unsigned w)
{
  GRead r; const double v = (double)w;
  r.pa = make_double2(v, v); r.za = r.pa; r.pb = r.pa; r.qb = r.pa; r.zb = r.pa; r.vv = r.pa;
  return r;
}

// one visit (element slot, local node la) to a row's diagonal block: K_aa^e = (vl + vm) g_a (x) g_a + (g_a . t_a) I,
// symmetric: a = { 00, 01, 02, 11, 12, 22 }
template <bool DOF>
__device__ __forceinline__ void g_consume_diag(const double *sT, unsigned w, double (&a)[6], double (&fa)[3])
{
  const double2 *T = reinterpret_cast<const double2 *>(sT + G_SLOT(w) * GREC);
  const int la = G_LA(w);
  const double2 pa = T[la], qa = T[4 + la], za = T[8 + la], vv = T[12];
  const double s = vv.x + vv.y;
  const double d = pa.x * qa.x + pa.y * qa.y + za.x * za.y;
  const double h0 = s * pa.x, h1 = s * pa.y, h2 = s * za.x;
  a[0] += fma(h0, pa.x, d); a[1] = fma(h0, pa.y, a[1]); a[2] = fma(h0, za.x, a[2]);
  a[3] += fma(h1, pa.y, d); a[4] = fma(h1, za.x, a[4]); a[5] += fma(h2, za.x, d);
  if (DOF) {
    // the same visits carry the row's residual: -vol sigma g_a = -(t_a - vm g_a), with the product rounded exactly
    // as it was when t_a was formed (no fused multiply-add on either side): a stress-free state gives f = 0 to the bit
    fa[0] -= __dsub_rn(qa.x, __dmul_rn(vv.y, pa.x)); fa[1] -= __dsub_rn(qa.y, __dmul_rn(vv.y, pa.y)); fa[2] -= __dsub_rn(za.y, __dmul_rn(vv.y, za.x));
  }
}

// residual contribution of one (element, local node) visit: -vol sigma g_a (fea_solver.c:1096-1109)
template <bool DOK>
__device__ __forceinline__ void g_visit(const double *sT, unsigned w, double (&fa)[3])
{
  const int la = G_LA(w);
  if (DOK) {
    const double2 *T = reinterpret_cast<const double2 *>(sT + G_SLOT(w) * GREC);
    const double2 pa = T[la], qa = T[4 + la], za = T[8 + la];
    const double vm = T[12].y;
    fa[0] -= qa.x - vm * pa.x; fa[1] -= qa.y - vm * pa.y; fa[2] -= za.y - vm * za.x;
  } else {
    const double *T = sT + G_SLOT(w) * GREC_F;
    const double2 pa = *reinterpret_cast<const double2 *>(T + 2 * la);
    fa[0] -= pa.x; fa[1] -= pa.y; fa[2] -= T[8 + la];
  }
}

// logical record R = { g[4][3], t[4][3], vl, vm } (or { s[4][3] }) -> the piece layout above
// two doubles of a record -> LDS with one ds_write2_b64: the instruction takes its two operands from ANY two register
// pairs, whereas the ds_write_b128 the compiler forms out of adjacent stores needs four consecutive registers and cost
// 47 v_mov_b64 per record to pack them.  Inline asm: the compiler does not see these stores, so the caller waits for
// them itself (G_LDS_DRAIN) before the barrier.
#define G_W2(addr, a, b, o) asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" : : "v"(addr), "v"(a), "v"(b), "n"(o), "n"((o) + 1) : "memory")
#define G_LDS_DRAIN() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#ifndef G_W128
#define G_W128 1
#endif
typedef double g_v2d __attribute__((ext_vector_type(2)));
#if G_W128
// one piece = one ds_write_b128: eight consecutive lanes cover the 32 write banks once at the 208-byte record stride
// (conflict-free, where the two halves of ds_write2_b64 at a 16-byte-aligned stride meet two ways); the pair has to sit
// in four consecutive registers, which is the register allocator's business here
#undef G_W2
#define G_W2(addr, a, b, o) do { const g_v2d pr_ = {a, b}; asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(pr_), "n"((o) * 8) : "memory"); } while (0)
#endif

template <bool DOK>
__device__ __forceinline__ void g_store_record(double *dst, const double *R)
{
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const void *)dst;
  G_W2(a, R[0], R[1], 0); G_W2(a, R[3], R[4], 2); G_W2(a, R[6], R[7], 4); G_W2(a, R[9], R[10], 6);
  if (DOK) {
    G_W2(a, R[12], R[13], 8); G_W2(a, R[15], R[16], 10); G_W2(a, R[18], R[19], 12); G_W2(a, R[21], R[22], 14);
    G_W2(a, R[2], R[14], 16); G_W2(a, R[5], R[17], 18); G_W2(a, R[8], R[20], 20); G_W2(a, R[11], R[23], 22);
    G_W2(a, R[24], R[25], 24);
  } else {
    G_W2(a, R[2], R[5], 8); G_W2(a, R[8], R[11], 10);
  }
}

// Element record of a constant-strain tetrahedron straight from its node coordinates, in the fewest operations:
//   J = [x_k - x_0], g_k = rows of adj(J)/det J (k = 1..3), g_0 = -(g_1 + g_2 + g_3)      fea_solver.c:690-718
//   F^-1 = sum_k (X_k - X_0) (x) g_k                                                     fea_solver.c:1141-1151
//   Neo-Hookean: B = F F' = (F^-T F^-1)^-1 by the adjugate of the symmetric C = Fi'Fi, J = 1/det Fi:
//     vol sigma = vol mu J adj(C) - vol (mu - lambda ln J)/J I,  l1 = lambda/J, m1 = (mu - lambda ln J)/J
//                                                                                        fea_model.c:79-107,129-148
//   t_k = vol (m1 g_k + sigma g_k)
// R = { g[4][3], t[4][3], vol l1, vol m1 } (DOK) or { vol sigma g [4][3] } (residual only).
// Returns det J (its sign and zero test are the caller's business).
template <bool DOK>
__device__ __forceinline__ double lintet_record_nh(const double (&x)[4][3], const double (&X)[4][3], double w,
                                                   double lambda, double mu, double *R)
{
  double J[3][3], D[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { J[i][j] = x[i + 1][j] - x[0][j]; D[i][j] = X[i + 1][j] - X[0][j]; }
  // cofactors: c[k][i] = d detJ / d J[k][i]  ->  g_{k+1}[i] = c[k][i] / det... with J[k][.] = x_{k+1} - x_0 the
  // inverse Ji[i][k] = cof(J)[k][i]/det, and g_{k+1}[i] = Ji[i][k]
  double c[3][3];
  c[0][0] = J[1][1] * J[2][2] - J[1][2] * J[2][1]; c[0][1] = J[1][2] * J[2][0] - J[1][0] * J[2][2]; c[0][2] = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  c[1][0] = J[0][2] * J[2][1] - J[0][1] * J[2][2]; c[1][1] = J[0][0] * J[2][2] - J[0][2] * J[2][0]; c[1][2] = J[0][1] * J[2][0] - J[0][0] * J[2][1];
  c[2][0] = J[0][1] * J[1][2] - J[0][2] * J[1][1]; c[2][1] = J[0][2] * J[1][0] - J[0][0] * J[1][2]; c[2][2] = J[0][0] * J[1][1] - J[0][1] * J[1][0];
  const double detJ = J[0][0] * c[0][0] + J[0][1] * c[0][1] + J[0][2] * c[0][2];
  const double id = fd_rcp(detJ);
  double g[4][3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 3; ++i) g[k + 1][i] = c[k][i] * id;
#pragma unroll
  for (int i = 0; i < 3; ++i) g[0][i] = -((g[1][i] + g[2][i]) + g[3][i]);
  // Fi[i][j] = sum_k g_{k+1}[j] D[k][i]
  double Fi[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Fi[i][j] = g[1][j] * D[0][i] + g[2][j] * D[1][i] + g[3][j] * D[2][i];
  const double detFi = fd_det3(Fi);
  const double Jd = fd_rcp(detFi);                    // J = det F
  const double lnJ = -fd_log(detFi);
  double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    c00 += Fi[i][0] * Fi[i][0]; c01 += Fi[i][0] * Fi[i][1]; c02 += Fi[i][0] * Fi[i][2];
    c11 += Fi[i][1] * Fi[i][1]; c12 += Fi[i][1] * Fi[i][2]; c22 += Fi[i][2] * Fi[i][2];
  }
  const double vol = w * fabs(detJ);
  const double m1 = (mu - lambda * lnJ) * detFi;      // (mu - lambda ln J)/J
  const double vm = vol * m1;
  const double mJ = (vol * mu) * Jd;                  // vol mu J
  // T = vol (sigma + m1 I) = vol mu J adj(C): the isotropic part -vm of vol sigma cancels against vm I, so
  //   t_b = vol (m1 g_b + sigma g_b) = T g_b
  // is nine operations per node instead of fifteen (round 4; the evaluation is the FP64-bound third of a chunk).  The
  // diagonal product goes LAST into each fused chain: in a stress-free state the off-diagonal entries are exact zeros and
  // T_ii = vol mu J = vm to the bit, so t_b = round(vm g_b) exactly and the residual -(t_a - round(vm g_a)) of
  // g_consume_diag is still zero to the bit there.
  const double a00 = c11 * c22 - c12 * c12, a11 = c00 * c22 - c02 * c02, a22 = c00 * c11 - c01 * c01;
  const double a01 = c02 * c12 - c01 * c22, a02 = c01 * c12 - c02 * c11, a12 = c01 * c02 - c00 * c12;
  if (DOK) {
    const double T00 = mJ * a00, T11 = mJ * a11, T22 = mJ * a22, T01 = mJ * a01, T02 = mJ * a02, T12 = mJ * a12;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 3; ++i) R[b * 3 + i] = g[b][i];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      R[12 + b * 3 + 0] = fma(T00, g[b][0], fma(T01, g[b][1], T02 * g[b][2]));
      R[12 + b * 3 + 1] = fma(T11, g[b][1], fma(T01, g[b][0], T12 * g[b][2]));
      R[12 + b * 3 + 2] = fma(T22, g[b][2], fma(T02, g[b][0], T12 * g[b][1]));
    }
    R[24] = (vol * lambda) * detFi;                   // vol lambda/J
    R[25] = vm;
  } else {
    double S[3][3];                                   // vol sigma
    S[0][0] = mJ * a00 - vm; S[1][1] = mJ * a11 - vm; S[2][2] = mJ * a22 - vm;
    S[0][1] = S[1][0] = mJ * a01; S[0][2] = S[2][0] = mJ * a02; S[1][2] = S[2][1] = mJ * a12;
#pragma unroll
    for (int b = 1; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 3; ++i) R[b * 3 + i] = S[i][0] * g[b][0] + S[i][1] * g[b][1] + S[i][2] * g[b][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) R[i] = -((R[3 + i] + R[6 + i]) + R[9 + i]);
  }
  return detJ;
}

// The same record for the A5 model (fea_model.c:26-77, 110-127): S = (lambda tr(C) I + 2 mu C)/J with C = (F'F - I)/2,
// sigma = F S F'.  With B = F F' = J^2 adj(Fi) adj(Fi)' (Fi = F^-1 as above, J = 1/det Fi) this is
//   sigma = (lambda I1 B + mu (B B - B))/J,  I1 = (tr B - 3)/2,   l1 = lambda/J, m1 = mu/J
// -- symmetric 3x3 products only, no F, no general inverse: what keeps the record inside the register budget of a
// 1024-thread workgroup (the general state of fem_device.h spilled 76 bytes per lane there).
template <bool DOK>
__device__ __forceinline__ double lintet_record_a5(const double (&x)[4][3], const double (&X)[4][3], double w,
                                                   double lambda, double mu, double *R)
{
  double J[3][3], D[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { J[i][j] = x[i + 1][j] - x[0][j]; D[i][j] = X[i + 1][j] - X[0][j]; }
  double c[3][3];
  c[0][0] = J[1][1] * J[2][2] - J[1][2] * J[2][1]; c[0][1] = J[1][2] * J[2][0] - J[1][0] * J[2][2]; c[0][2] = J[1][0] * J[2][1] - J[1][1] * J[2][0];
  c[1][0] = J[0][2] * J[2][1] - J[0][1] * J[2][2]; c[1][1] = J[0][0] * J[2][2] - J[0][2] * J[2][0]; c[1][2] = J[0][1] * J[2][0] - J[0][0] * J[2][1];
  c[2][0] = J[0][1] * J[1][2] - J[0][2] * J[1][1]; c[2][1] = J[0][2] * J[1][0] - J[0][0] * J[1][2]; c[2][2] = J[0][0] * J[1][1] - J[0][1] * J[1][0];
  const double detJ = J[0][0] * c[0][0] + J[0][1] * c[0][1] + J[0][2] * c[0][2];
  const double id = fd_rcp(detJ);
  double g[4][3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 3; ++i) g[k + 1][i] = c[k][i] * id;
#pragma unroll
  for (int i = 0; i < 3; ++i) g[0][i] = -((g[1][i] + g[2][i]) + g[3][i]);
  double Fi[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Fi[i][j] = g[1][j] * D[0][i] + g[2][j] * D[1][i] + g[3][j] * D[2][i];
  // adj(Fi): F = adj(Fi) / det(Fi)
  double A[3][3];
  A[0][0] = Fi[1][1] * Fi[2][2] - Fi[1][2] * Fi[2][1]; A[0][1] = Fi[0][2] * Fi[2][1] - Fi[0][1] * Fi[2][2]; A[0][2] = Fi[0][1] * Fi[1][2] - Fi[0][2] * Fi[1][1];
  A[1][0] = Fi[1][2] * Fi[2][0] - Fi[1][0] * Fi[2][2]; A[1][1] = Fi[0][0] * Fi[2][2] - Fi[0][2] * Fi[2][0]; A[1][2] = Fi[0][2] * Fi[1][0] - Fi[0][0] * Fi[1][2];
  A[2][0] = Fi[1][0] * Fi[2][1] - Fi[1][1] * Fi[2][0]; A[2][1] = Fi[0][1] * Fi[2][0] - Fi[0][0] * Fi[2][1]; A[2][2] = Fi[0][0] * Fi[1][1] - Fi[0][1] * Fi[1][0];
  const double detFi = Fi[0][0] * A[0][0] + Fi[0][1] * A[1][0] + Fi[0][2] * A[2][0];
  const double Jd = fd_rcp(detFi), J2 = Jd * Jd;
  // B = F F' (symmetric: 00 01 02 11 12 22)
  const double b00 = J2 * (A[0][0] * A[0][0] + A[0][1] * A[0][1] + A[0][2] * A[0][2]);
  const double b01 = J2 * (A[0][0] * A[1][0] + A[0][1] * A[1][1] + A[0][2] * A[1][2]);
  const double b02 = J2 * (A[0][0] * A[2][0] + A[0][1] * A[2][1] + A[0][2] * A[2][2]);
  const double b11 = J2 * (A[1][0] * A[1][0] + A[1][1] * A[1][1] + A[1][2] * A[1][2]);
  const double b12 = J2 * (A[1][0] * A[2][0] + A[1][1] * A[2][1] + A[1][2] * A[2][2]);
  const double b22 = J2 * (A[2][0] * A[2][0] + A[2][1] * A[2][1] + A[2][2] * A[2][2]);
  const double I1 = 0.5 * ((b00 + b11 + b22) - 3.0);
  const double vol = w * fabs(detJ);
  const double vd = vol * detFi;                          // vol / J
  const double a_ = vd * (lambda * I1 - mu), m_ = vd * mu; // vol sigma = a_ B + m_ B B
  double S[3][3];
  S[0][0] = a_ * b00 + m_ * (b00 * b00 + b01 * b01 + b02 * b02);
  S[1][1] = a_ * b11 + m_ * (b01 * b01 + b11 * b11 + b12 * b12);
  S[2][2] = a_ * b22 + m_ * (b02 * b02 + b12 * b12 + b22 * b22);
  S[0][1] = S[1][0] = a_ * b01 + m_ * (b00 * b01 + b01 * b11 + b02 * b12);
  S[0][2] = S[2][0] = a_ * b02 + m_ * (b00 * b02 + b01 * b12 + b02 * b22);
  S[1][2] = S[2][1] = a_ * b12 + m_ * (b01 * b02 + b11 * b12 + b12 * b22);
  const double vm = vd * mu;                              // vol m1, m1 = mu / J
  if (DOK) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 3; ++i) R[b * 3 + i] = g[b][i];
    // t_b = (vol sigma + vm I) g_b, the diagonal product last (lintet_record_nh: nine operations per node, and
    // t_b = round(vm g_b) to the bit where vol sigma is an exact zero)
    const double T00 = S[0][0] + vm, T11 = S[1][1] + vm, T22 = S[2][2] + vm;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      R[12 + b * 3 + 0] = fma(T00, g[b][0], fma(S[0][1], g[b][1], S[0][2] * g[b][2]));
      R[12 + b * 3 + 1] = fma(T11, g[b][1], fma(S[0][1], g[b][0], S[1][2] * g[b][2]));
      R[12 + b * 3 + 2] = fma(T22, g[b][2], fma(S[0][2], g[b][0], S[1][2] * g[b][1]));
    }
    R[24] = vd * lambda;                                  // vol lambda / J
    R[25] = vm;
  } else {
#pragma unroll
    for (int b = 1; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 3; ++i) R[b * 3 + i] = S[i][0] * g[b][0] + S[i][1] * g[b][1] + S[i][2] * g[b][2];
#pragma unroll
    for (int i = 0; i < 3; ++i) R[i] = -((R[3 + i] + R[6 + i]) + R[9 + i]);
  }
  return detJ;
}

// v + (v of the lane whose id differs in bit 0), then the same over bit 1: the sum of a quad, in every lane of it, by
// DPP quad permutes (VALU only).  __shfl_xor compiles to ds_bpermute_b32, an LDS-pipe instruction per 32-bit half and
// step: the 72 of them a diagonal wave issued for its nine sums queued behind the block waves' tile writes and made
// the four diagonal waves the last to finish the tile phase (2 600 of its 2 950 cycles, by the in-kernel stamps).
template <int CTRL>
__device__ __forceinline__ double g_quad_perm(double x)
{
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double g_quad_sum(double v)
{
  v += g_quad_perm<0xB1>(v);            // quad_perm [1,0,3,2]
  v += g_quad_perm<0x4E>(v);            // quad_perm [2,3,0,1]
  return v;
}

// workgroup barrier that orders LDS only: __syncthreads() would also drain every global load and store in flight
// (s_waitcnt vmcnt(0)), i.e. the prefetches of the next chunk and the row stores of the previous one
#define G_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)

typedef int g_v8i __attribute__((ext_vector_type(8)));
typedef int g_v4i __attribute__((ext_vector_type(4)));

// what a thread holds of one chunk's map record
struct GMaps {
  unsigned eids, tpos, cw[FEA_G_REGW], vw[2];
  int kd, vb, ve;
};

#define G_TASK_THREADS FEA_G_TASK_THREADS

template <bool DOK, bool DOF>
__device__ __forceinline__ void g_load_maps(const GatherLayout &lay, const unsigned char *rec, int t, GMaps &m)
{
  // Loads from inside the record (always in bounds), masked by what the LARGEST chunk of the mesh needs (known
  // without the header of this chunk, so none of them waits for it); a thread never uses a word it does not own.
  const unsigned short *rows = reinterpret_cast<const unsigned short *>(rec + lay.o_rows);
  m.eids = 0xFFFFFFFFu; m.tpos = 0; m.kd = m.vb = m.ve = 0;
#pragma unroll
  for (int k = 0; k < FEA_G_REGW; ++k) m.cw[k] = 0;
  m.vw[0] = m.vw[1] = 0;
  if (t < lay.max_elems) m.eids = reinterpret_cast<const unsigned *>(rec + lay.o_elems)[t];
  if (DOK && t < ((lay.max_tasks + 63) & ~63)) {
    m.tpos = reinterpret_cast<const unsigned *>(rec + lay.o_bpos)[t];
#pragma unroll
    for (int k = 0; k < FEA_G_REGW; ++k)
      if (k < lay.max_depth) m.cw[k] = reinterpret_cast<const unsigned *>(rec + lay.o_clist)[k * FEA_G_THREADS + t];
  }
  if (DOK && t >= G_TASK_THREADS) {                   // the last four waves: four lanes per row, the visits of its diagonal block
    const int l = t - G_TASK_THREADS;
    m.kd = rows[G_RD + (l >> 2)];
#pragma unroll
    for (int k = 0; k < FEA_G_REGW; ++k)
      if (k < lay.max_ddepth) m.cw[k] = reinterpret_cast<const unsigned *>(rec + lay.o_dlist)[k * FEA_G_DIAG_LANES + l];
  }
  if (!DOK && t < ((lay.max_vthr + 63) & ~63)) {
#pragma unroll
    for (int v = 0; v < 2; ++v)
      if (v < lay.max_vdepth) m.vw[v] = reinterpret_cast<const unsigned short *>(rec + lay.o_vlist)[v * FEA_G_THREADS + t];
  }
  if (!DOK) {
    const int fr = min(t / 3, FEA_G_MAX_ROWS - 1);
    m.vb = rows[G_VF + fr]; m.ve = rows[G_VF + fr + 1];
  }
}

