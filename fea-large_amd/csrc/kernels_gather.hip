// kernels_gather.hip -- GATHER assembly of linear tetrahedra: stiffness and
// residual with no atomics anywhere (replaces fea_solver.c:873-883 and
// :863-870 for TETRAHEDRA4 meshes; algebra of fem_device.h).
//
// The reference scatters (3n)^2 values per element into a growing sparse matrix.
// The row-owner kernels of kernels_visit.hip turned that into LDS atomics: one
// lane per (row node, element) visit, the element state re-evaluated by each of
// its four visits, 27 + 3 ds_add_f64 per visit -- and the LDS pipe (~8 clk per
// f64 atomic wave-instruction, more with bank conflicts) ended up bounding the
// kernel at 27 % of the HBM roofline.  Here the scatter is inverted on the host
// once (gather.cpp) and a 256-thread workgroup owns a chunk of consecutive
// block rows:
//   0. the chunk's map record and the coordinates of the nodes its elements
//      touch are loaded (two dependent latencies), coordinates into LDS;
//   1. thread <-> element: every element that touches the chunk's rows is
//      evaluated ONCE per chunk (J^-1, F^-1, sigma, tangent coefficients) and
//      parked in LDS as { g_b, t_b = vol (m1 g_b + sigma g_b), vol l1, vol m1 };
//   2. thread <-> off-diagonal block (a, b): walks the block's contribution list
//      (element, local a, local b), reads g_a, g_b, t_b, vl, vm from the record
//      and sums K_ab = sum_e [ vl g_a (x) g_b + vm g_b (x) g_a + (g_a . t_b) I ]
//      in registers, in ascending element order;
//      thread <-> slice of a row's visits: partial sums of the residual
//      -(t_a - vm g_a) = -vol sigma g_a (fea_solver.c:1096-1109);
//   3. finished blocks and residual partials go to an LDS tile (aliasing the
//      records); the diagonal block of every row is minus the sum of the row's
//      other blocks (shape functions sum to one), the residual of a row the sum
//      of its partials;
//   4. the rows leave LDS as one contiguous, coalesced stream of 16-byte
//      stores: every CSR value is written exactly once.
// Every sum has a fixed order: the assembly is bitwise reproducible.
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
#ifdef FEAHIP_DEBUG
#define G_ABL(bit) (A.ablate & (bit))
#else
#define G_ABL(bit) 0
#endif

#ifdef FEAHIP_DEBUG
#define G_STAMP(i) do { if (A.stamps) st[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define G_STAMP(i) do { } while (0)
#endif

#define G_RS 0
#define G_RD 20
#define G_VF 40

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
  const double2 *T = reinterpret_cast<const double2 *>(sT + (w & 255u) * GREC);
  const int la = (w >> 8) & 3, lb = (w >> 10) & 3;
  GRead r;
  r.pa = T[la]; r.za = T[8 + la]; r.pb = T[lb]; r.qb = T[4 + lb]; r.zb = T[8 + lb]; r.vv = T[12];
  return r;
}
__device__ __forceinline__ void g_apply(const GRead &r, double (&acc)[9])
{
  const double ga0 = r.pa.x, ga1 = r.pa.y, ga2 = r.za.x;
  const double gb0 = r.pb.x, gb1 = r.pb.y, gb2 = r.zb.x;
  const double tb0 = r.qb.x, tb1 = r.qb.y, tb2 = r.zb.y;
  const double vl = r.vv.x, vm = r.vv.y;
  const double h0 = vl * gb0, h1 = vl * gb1, h2 = vl * gb2;
  const double m0 = vm * gb0, m1 = vm * gb1, m2 = vm * gb2;
  const double d = ga0 * tb0 + ga1 * tb1 + ga2 * tb2;
  acc[0] += d; acc[4] += d; acc[8] += d;
  acc[0] = fma(ga0, h0, fma(ga0, m0, acc[0])); acc[1] = fma(ga0, h1, fma(ga1, m0, acc[1])); acc[2] = fma(ga0, h2, fma(ga2, m0, acc[2]));
  acc[3] = fma(ga1, h0, fma(ga0, m1, acc[3])); acc[4] = fma(ga1, h1, fma(ga1, m1, acc[4])); acc[5] = fma(ga1, h2, fma(ga2, m1, acc[5]));
  acc[6] = fma(ga2, h0, fma(ga0, m2, acc[6])); acc[7] = fma(ga2, h1, fma(ga1, m2, acc[7])); acc[8] = fma(ga2, h2, fma(ga2, m2, acc[8]));
}
__device__ __forceinline__ void g_consume(const double *sT, unsigned w, double (&acc)[9])
{
  g_apply(g_fetch(sT, w), acc);
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
  const double2 *T = reinterpret_cast<const double2 *>(sT + (w & 255u) * GREC);
  const int la = (w >> 8) & 3;
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
  const int la = (w >> 8) & 3;
  if (DOK) {
    const double2 *T = reinterpret_cast<const double2 *>(sT + (w & 255u) * GREC);
    const double2 pa = T[la], qa = T[4 + la], za = T[8 + la];
    const double vm = T[12].y;
    fa[0] -= qa.x - vm * pa.x; fa[1] -= qa.y - vm * pa.y; fa[2] -= za.y - vm * za.x;
  } else {
    const double *T = sT + (w & 255u) * GREC_F;
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
  double S[3][3];                                     // vol sigma
  S[0][0] = mJ * (c11 * c22 - c12 * c12) - vm;
  S[1][1] = mJ * (c00 * c22 - c02 * c02) - vm;
  S[2][2] = mJ * (c00 * c11 - c01 * c01) - vm;
  S[0][1] = S[1][0] = mJ * (c02 * c12 - c01 * c22);
  S[0][2] = S[2][0] = mJ * (c01 * c12 - c02 * c11);
  S[1][2] = S[2][1] = mJ * (c01 * c02 - c00 * c12);
  if (DOK) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 3; ++i) R[b * 3 + i] = g[b][i];
    // t_b = round(vm g_b) + vol sigma g_b, every node alike and without fused multiply-add, so that the residual
    // -(t_a - round(vm g_a)) of a stress-free state is zero to the bit (g_consume_diag)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        R[12 + b * 3 + i] = __dadd_rn(__dmul_rn(vm, g[b][i]), S[i][0] * g[b][0] + S[i][1] * g[b][1] + S[i][2] * g[b][2]);
    R[24] = (vol * lambda) * detFi;                   // vol lambda/J
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

// any model through the general state of fem_device.h (A5)
template <bool DOK>
__device__ __forceinline__ double lintet_record_any(const double (&xe)[4][3], const double (&Xe)[4][3], const ElemTable *tab,
                                                    int model, double lambda, double mu, double *R)
{
  GPState<4> s;
  gp_state<4, true, false>(xe, Xe, tab, 0, model, lambda, mu, s);
  const double vm = s.vol * s.m1;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double sg = s.vol * (s.sig[i][0] * s.g[b][0] + s.sig[i][1] * s.g[b][1] + s.sig[i][2] * s.g[b][2]);
      if (DOK) { R[b * 3 + i] = s.g[b][i]; R[12 + b * 3 + i] = __dadd_rn(__dmul_rn(vm, s.g[b][i]), sg); }
      else R[b * 3 + i] = sg;
    }
  if (DOK) { R[24] = s.vol * s.l1; R[25] = vm; }
  return s.detJ;
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

#define G_TASK_THREADS 192           // block and residual threads: waves 0-2; wave 3 sums the diagonal blocks

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
  if (DOK && t >= G_TASK_THREADS) {                   // wave 3: four lanes per row, the visits of its diagonal block
    const int l = t - G_TASK_THREADS;
    m.kd = rows[G_RD + (l >> 2)];
#pragma unroll
    for (int k = 0; k < FEA_G_REGW; ++k)
      if (k < lay.max_ddepth) m.cw[k] = reinterpret_cast<const unsigned *>(rec + lay.o_dlist)[k * 64 + l];
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

// Persistent form: a workgroup walks a run of consecutive chunks and keeps the next chunk's loads in flight under
// the current chunk's arithmetic --
//   * the map words and the node coordinates of chunk i+1 (and the node ids of chunk i+2) are requested before the
//     state phase of chunk i and sit in registers; the coordinates move into the LDS tile after chunk i's gather
//     phase (the tile is dead from the end of the state phase), a whole state + gather phase after their request;
//   * the finished rows of chunk i are stored last and nobody waits for them.
template <bool DOK, bool DOF, bool NH>
__global__ __launch_bounds__(FEA_G_THREADS, 3)
void k_assemble_gather(GatherArgs A, int run_len)
{
  extern __shared__ double2 g_smem[];
  double2 *sC = g_smem;                                       // per node slot 3 x 16 bytes: (x0,x1) (x2,X0) (X1,X2)
  double *sT = reinterpret_cast<double *>(g_smem + A.lay.max_nodes * 3);   // element records; later the K tile and the residual partials
  const int t = threadIdx.x;
#ifdef FEAHIP_DEBUG
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), real0 = __builtin_amdgcn_s_memrealtime();
#endif
  // XCD-aware order: workgroups b and b+8 share an L2; each XCD gets a contiguous eighth of the runs so that
  // neighbouring chunks re-read each other's halo coordinates from the same L2 (speed only)
  const int nruns = (A.nchunks + run_len - 1) / run_len;
  const int per = ((int)gridDim.x + 7) >> 3;
  const int ridx = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (ridx >= nruns) return;
  int chunk = ridx * run_len;
  const int cend = min(A.nchunks, chunk + run_len);
  constexpr int REC = DOK ? GREC : GREC_F;
  const size_t stride = (size_t)A.lay.stride;
  const unsigned char *rec = A.maps + (size_t)(A.chunk0 + chunk) * stride;
  const bool node_lane = t < A.lay.max_nodes;

  // the Gauss weight is read ONCE: a load inside the loop would be the youngest memory operation when the state
  // phase needs it, and waiting for it waits for every prefetch issued before it (in-order counter)
  const double gauss_w = A.tab->w[0];
  // ---- prologue: maps and coordinates of the first chunk, nothing to hide behind
  GMaps mn;                                                   // "next": the chunk about to be worked on
  g_load_maps<DOK, DOF>(A.lay, rec, t, mn);
  GatherHeader hn = *reinterpret_cast<const GatherHeader *>(rec);
  int node1 = reinterpret_cast<const int *>(rec + A.lay.o_nodes)[t & (FEA_G_MAX_NODES - 1)];       // node slot t of the chunk whose coordinates are loaded next
  if (node_lane) {
    const double *gx = A.x + (size_t)node1 * 4, *gX = A.X0 + (size_t)node1 * 4;
    const double2 a0 = *reinterpret_cast<const double2 *>(gx), a1 = *reinterpret_cast<const double2 *>(gx + 2);
    const double2 c0 = *reinterpret_cast<const double2 *>(gX), c1 = *reinterpret_cast<const double2 *>(gX + 2);
    sC[t * 3] = a0; sC[t * 3 + 1] = make_double2(a1.x, c0.x); sC[t * 3 + 2] = make_double2(c0.y, c1.x);
  }
  {
    const int cn = min(chunk + 1, cend - 1);
    node1 = reinterpret_cast<const int *>(A.maps + (size_t)(A.chunk0 + cn) * stride + A.lay.o_nodes)[t & (FEA_G_MAX_NODES - 1)];
  }
  // nothing is pending when the loop is entered: otherwise the loop header inherits "node1 may still be in
  // flight" and the compiler waits for everything (s_waitcnt vmcnt(0)) at the top of EVERY iteration
  asm volatile("" : : "v"(mn.eids), "v"(mn.tpos), "v"(mn.cw[0]), "v"(mn.cw[1]), "v"(mn.cw[2]), "v"(mn.cw[3]),
               "v"(mn.vw[0]), "v"(mn.vw[1]), "v"(mn.kd), "v"(mn.vb), "v"(mn.ve), "v"(node1), "v"(gauss_w));
  G_BARRIER();

  for (;;) {
    G_STAMP(0);
    const bool more = chunk + 1 < cend;
    const GatherHeader h = hn;
    int hword;
    const GMaps m = mn;
    const int nrows = h.r1 - h.r0;
    rec = A.maps + (size_t)(A.chunk0 + chunk) * stride;

    // ---- next chunk's loads, a whole state + gather phase ahead of their first use: header, map words, node
    // coordinates (by the node ids requested one chunk earlier), and the node ids of the chunk after it.
    // Clamped indices instead of branches (a load under a branch is waited for at the join).
    double2 ca0, ca1, cc0, cc1;
    {
      const int c1 = min(chunk + 1, cend - 1), c2 = min(chunk + 2, cend - 1);
      const unsigned char *rec1 = A.maps + (size_t)(A.chunk0 + c1) * stride;
      // the header of the next chunk as a VECTOR load (lane l holds word l, read back with v_readlane): a scalar
      // load shares its counter with the LDS, and every barrier's wait for the LDS would wait for it as well
      hword = reinterpret_cast<const int *>(rec1)[t & 15];
      g_load_maps<DOK, DOF>(A.lay, rec1, t, mn);
      const size_t n1 = (size_t)(node_lane ? node1 : 0);
      ca0 = *reinterpret_cast<const double2 *>(A.x + n1 * 4); ca1.x = A.x[n1 * 4 + 2];
      cc0 = *reinterpret_cast<const double2 *>(A.X0 + n1 * 4); cc1.x = A.X0[n1 * 4 + 2];
      node1 = reinterpret_cast<const int *>(A.maps + (size_t)(A.chunk0 + c2) * stride + A.lay.o_nodes)[t & (FEA_G_MAX_NODES - 1)];
    }

    __builtin_amdgcn_s_setprio(PRIO_STATE);
    // ---- phase 1: one state evaluation per element of the chunk
    if (t < h.nelem && m.eids != 0xFFFFFFFFu) {
      const unsigned eids = m.eids;
      const int nd[4] = {(int)(eids & 255u), (int)((eids >> 8) & 255u), (int)((eids >> 16) & 255u), (int)(eids >> 24)};
      double xe[4][3], Xe[4][3];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double2 *cc = sC + nd[k] * 3;                 // 16-byte reads: the bank slot follows the node slot mod 16
        const double2 p0 = cc[0], p1 = cc[1], p2 = cc[2];
        xe[k][0] = p0.x; xe[k][1] = p0.y; xe[k][2] = p1.x;
        Xe[k][0] = p1.y; Xe[k][1] = p2.x; Xe[k][2] = p2.y;
      }
      double R[REC];
      double detJ;
      if (G_ABL(8)) { detJ = 1.0; for (int q = 0; q < REC; ++q) R[q] = xe[q & 3][q % 3] + Xe[(q >> 2) & 3][q % 3]; }
      else detJ = NH ? lintet_record_nh<DOK>(xe, Xe, gauss_w, A.lambda, A.mu, R)
                     : lintet_record_any<DOK>(xe, Xe, A.tab, A.model, A.lambda, A.mu, R);
      if (!(detJ > 0.0)) {                             // rare, kept off the fast path
        if (DOK) {                                     // counted by the chunk that owns its lowest-numbered node
          const int *gn = reinterpret_cast<const int *>(rec + A.lay.o_nodes);
          const int g0 = min(min(gn[nd[0]], gn[nd[1]]), min(gn[nd[2]], gn[nd[3]]));
          if (g0 >= h.r0 && g0 < h.r1) atomicAdd(A.bad, 1);
        }
      }
      if (detJ == 0.0) {                               // fea_solver.c:697: no gradient, no contribution: an all-zero record
        double2 *o = reinterpret_cast<double2 *>(sT + t * REC);
#pragma unroll
        for (int q = 0; q < REC / 2; ++q) o[q] = make_double2(0.0, 0.0);
      } else if (!G_ABL(1)) g_store_record<DOK>(sT + t * REC, R);
      else sT[t * REC] = R[0] + R[25 % REC];
    } else if (t < h.nelem) {                          // unused slot: the all-zero record empty list slots point at
      double2 *o = reinterpret_cast<double2 *>(sT + t * REC);
#pragma unroll
      for (int q = 0; q < REC / 2; ++q) o[q] = make_double2(0.0, 0.0);
    }
    G_LDS_DRAIN();                                     // the asm record stores
    G_BARRIER();                                       // records visible; the coordinate tile is dead
    G_STAMP(1);

    __builtin_amdgcn_s_setprio(PRIO_GATHER);
    // ---- phase 2: block sums and residual partials, out of the records
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (DOK && t < h.noffd) {
      // the reads of the next contribution are in flight while the current one is summed; a list of n words is
      // walked as 2n entries whatever it holds (empty entries read the all-zero record)
#define G_FETCH(w) (G_ABL(4) ? g_fetch_cheap(sT, w) : g_fetch(sT, w))
#define G_APPLY(r) do { if (G_ABL(2)) g_apply_cheap(r, acc); else g_apply(r, acc); } while (0)
#pragma unroll
      for (int k = 0; k < FEA_G_REGW; ++k)
        if (k < h.depth) { { const GRead r = G_FETCH(m.cw[k] & 0xFFFFu); G_APPLY(r); } { const GRead r = G_FETCH(m.cw[k] >> 16); G_APPLY(r); } }
      for (int k = FEA_G_REGW; k < h.depth; ++k) {          // blocks with more than 8 contributions (unstructured meshes)
        const unsigned w = reinterpret_cast<const unsigned *>(rec + A.lay.o_clist)[k * FEA_G_THREADS + t];
        g_consume(sT, w & 0xFFFFu, acc); g_consume(sT, w >> 16, acc);
      }
    }
    double dg[6] = {0, 0, 0, 0, 0, 0};
    double fa[3] = {0, 0, 0};
    if (DOK && t >= G_TASK_THREADS && t - G_TASK_THREADS < 4 * nrows) {
#pragma unroll
      for (int k = 0; k < FEA_G_REGW; ++k)
        if (k < h.ddepth) { g_consume_diag<DOF>(sT, m.cw[k] & 0xFFFFu, dg, fa); g_consume_diag<DOF>(sT, m.cw[k] >> 16, dg, fa); }
      for (int k = FEA_G_REGW; k < h.ddepth; ++k) {        // nodes with more than 32 elements around them
        const unsigned w = reinterpret_cast<const unsigned *>(rec + A.lay.o_dlist)[k * 64 + t - G_TASK_THREADS];
        g_consume_diag<DOF>(sT, w & 0xFFFFu, dg, fa); g_consume_diag<DOF>(sT, w >> 16, dg, fa);
      }
    }
    if (!DOK && t < h.nvthr) {                         // residual alone: slices of a row's visits on every wave
      g_visit<DOK>(sT, m.vw[0], fa);
      if (h.vdepth > 1) g_visit<DOK>(sT, m.vw[1], fa);
      for (int v = 2; v < h.vdepth; ++v)
        g_visit<DOK>(sT, reinterpret_cast<const unsigned short *>(rec + A.lay.o_vlist)[v * FEA_G_THREADS + t], fa);
    }
    if (more && node_lane) {                           // next chunk's coordinates: the tile has been dead since the state phase
      sC[t * 3] = ca0; sC[t * 3 + 1] = make_double2(ca1.x, cc0.x); sC[t * 3 + 2] = make_double2(cc0.y, cc1.x);
    }
    G_STAMP(2);
    __builtin_amdgcn_s_setprio(PRIO_OUT);
    G_BARRIER();                                       // the records are dead: their space becomes the tile
    G_STAMP(3);

    // the tile sits at an LDS offset with the parity of the chunk's first global value, so LDS and HBM agree on
    // 16-byte alignment in the write-out
    const int odd = h.b0 & 1;
    double *sK = sT + odd;
    double *sF = DOK ? sT + ((A.lay.max_tile * 9 + 3) & ~1) : sT;
    if (DOK && t < h.noffd && !G_ABL(16)) {
      const int bpos = (int)(m.tpos & 0xFFFFu), mpos = (int)(m.tpos >> 16);
#pragma unroll
      for (int q = 0; q < 9; ++q) sK[bpos * 9 + q] = acc[q];
      if (mpos != 0xFFFF) {                            // K_ba = K_ab' (fea_solver.c:1249 relies on the same symmetry)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) sK[mpos * 9 + 3 * j + i] = acc[3 * i + j];
      }
    }
    if (DOK && t >= G_TASK_THREADS) {                  // the four partial sums of a row's diagonal block meet in its first lane
#pragma unroll
      for (int q = 0; q < 6; ++q) { dg[q] += __shfl_xor(dg[q], 1); dg[q] += __shfl_xor(dg[q], 2); }
      if (DOF) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { fa[q] += __shfl_xor(fa[q], 1); fa[q] += __shfl_xor(fa[q], 2); }
      }
      if ((t & 3) == 0 && t - G_TASK_THREADS < 4 * nrows) {
        double *o = sK + m.kd * 9;
        o[0] = dg[0]; o[1] = dg[1]; o[2] = dg[2]; o[3] = dg[1]; o[4] = dg[3]; o[5] = dg[4]; o[6] = dg[2]; o[7] = dg[4]; o[8] = dg[5];
      }
    }
    if (!DOK && t < h.nvthr) { sF[t * 3] = fa[0]; sF[t * 3 + 1] = fa[1]; sF[t * 3 + 2] = fa[2]; }
    // The prefetched words of the next chunk are "used" HERE, before this chunk's row stores are issued: the
    // hardware counts loads and stores in one in-order counter and the number of stores is not a compile-time
    // constant, so a first use after the stores would make the compiler wait for everything in flight, the
    // stores just issued included (s_waitcnt vmcnt(0): measured, a third of the chunk time).  Here only the
    // prefetches themselves are younger, they were requested a state + gather phase ago, and nothing ever waits
    // for a store.
    asm volatile("" : : "v"(mn.eids), "v"(mn.tpos), "v"(mn.cw[0]), "v"(mn.cw[1]), "v"(mn.cw[2]), "v"(mn.cw[3]),
                 "v"(mn.vw[0]), "v"(mn.vw[1]), "v"(mn.kd), "v"(mn.vb), "v"(mn.ve), "v"(node1), "v"(hword));
    hn.r0 = __builtin_amdgcn_readlane(hword, 0); hn.r1 = __builtin_amdgcn_readlane(hword, 1);
    hn.b0 = __builtin_amdgcn_readlane(hword, 2); hn.nb = __builtin_amdgcn_readlane(hword, 3);
    hn.nnode = __builtin_amdgcn_readlane(hword, 4); hn.nelem = __builtin_amdgcn_readlane(hword, 5);
    hn.noffd = __builtin_amdgcn_readlane(hword, 6); hn.depth = __builtin_amdgcn_readlane(hword, 7);
    hn.nvthr = __builtin_amdgcn_readlane(hword, 8); hn.vdepth = __builtin_amdgcn_readlane(hword, 9);
    hn.ddepth = __builtin_amdgcn_readlane(hword, 10);
    G_BARRIER();
    G_STAMP(4);
    if (!DOK) {                                        // f_a = sum of the row's partials
      const int ft = t;
      const int fr = ft / 3, fi = ft - 3 * fr;
      if (fr < nrows) {
        double a = 0;
        for (int k = m.vb; k < m.ve; ++k) a += sF[k * 3 + fi];
        A.f[(size_t)(h.r0 + fr) * 3 + fi] = a;
      }
    }
    G_STAMP(5);
    if (DOK && DOF && t >= G_TASK_THREADS && (t & 3) == 0 && t - G_TASK_THREADS < 4 * nrows) {
      // the row's residual, stored with the rows (after the wait for the prefetches above: no store before it)
      double *fo = A.f + (size_t)(h.r0 + ((t - G_TASK_THREADS) >> 2)) * 3;
      fo[0] = fa[0]; fo[1] = fa[1]; fo[2] = fa[2];
    }
    if (DOK) {
      // ---- phase 3: stream the finished rows out, 16-byte LDS reads and HBM stores
      double *Kd = A.K + (size_t)h.b0 * 9;
      const int total = h.nb * 9;
      if (odd && t == 0) Kd[0] = sK[0];
      const int npair = (total - odd) >> 1;
      for (int j = t; j < npair; j += FEA_G_THREADS) {
        const int p = odd + 2 * j;
        if (G_ABL(32) && j >= 64) break;               // timing experiment: one 1 KB store per chunk instead of all rows
        *reinterpret_cast<double2 *>(Kd + p) = *reinterpret_cast<const double2 *>(sK + p);
      }
      if (((total - odd) & 1) && t == 0) Kd[total - 1] = sK[total - 1];
    }
    G_BARRIER();                                       // the tile is free again (its reads are done)
    G_STAMP(6);
#ifdef FEAHIP_DEBUG
    if (A.stamps) for (int i = 0; i < 6; ++i) sa[i] += st[i + 1] - st[i];
#endif
    if (!more) break;
    ++chunk;
  }
#ifdef FEAHIP_DEBUG
  if (A.stamps && (t & 63) == 0) {                     // one line per wave: [run][wave][8]
    unsigned long long *o = A.stamps + ((size_t)ridx * 4 + (t >> 6)) * 8;
    for (int i = 0; i < 6; ++i) o[i] = sa[i];
    o[6] = __builtin_amdgcn_s_memtime() - clk0;          // shader cycles of this run ...
    o[7] = __builtin_amdgcn_s_memrealtime() - real0;     // ... and 100 MHz ticks: their ratio is the in-kernel clock
  }
#endif
}

int ensure_gather(feahip_ctx *c)
{
  if (c->have_gather && c->gather_row0 == c->row0 && c->gather_row1 == c->row1) return FEAHIP_OK;
  if (c->gather_failed || !c->h_pat || c->h_conn.empty() || !c->linear_tet || c->G != 1) return FEAHIP_OK;
  HostGather hg;
  build_host_gather(c->N, c->E, c->h_conn.data(), *c->h_pat, c->row0, c->row1, hg);
  if (!hg.ok) { c->gather_failed = true; return FEAHIP_OK; }
  if (c->d_gmaps) { (void)hipFree(c->d_gmaps); c->d_gmaps = nullptr; }
  FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_gmaps, hg.blob.size() ? hg.blob.size() : 1));
  FEA_HIP_CHECK(c, hipMemcpy(c->d_gmaps, hg.blob.data(), hg.blob.size(), hipMemcpyHostToDevice));
  if (!c->gather_lay) c->gather_lay = new GatherLayout();
  *c->gather_lay = hg.lay;
  c->ngchunks = hg.nchunks;
  c->gather_row0 = c->row0; c->gather_row1 = c->row1;
  c->gather_bytes = (long long)hg.blob.size();
  c->gather_evals_per_element = hg.distinct_elems ? (double)hg.total_evals / (double)hg.distinct_elems : 0.0;
  c->have_gather = true;
  return FEAHIP_OK;
}

#ifdef FEAHIP_DEBUG
// diagnostic build only: one chunk's map record and the layout, for the host-side LDS bank model (dbg/lds_model.py)
extern "C" int feahip_debug_gather_record(feahip_ctx *c, int chunk, int *layout_ints, unsigned char *record)
{
  int rc = ensure_gather(c);
  if (rc || !c->have_gather) return FEAHIP_ESTATE;
  if (chunk < 0) chunk = c->ngchunks / 2;
  memcpy(layout_ints, c->gather_lay, sizeof(GatherLayout));
  if (record) (void)hipMemcpy(record, c->d_gmaps + (size_t)chunk * c->gather_lay->stride, c->gather_lay->stride, hipMemcpyDeviceToHost);
  return (int)(sizeof(GatherLayout) / sizeof(int));
}
#endif

int launch_assemble_gather(feahip_ctx *c, bool doK, bool doF)
{
  GatherArgs A;
  A.chunk0 = 0; A.nchunks = c->ngchunks; A.model = c->model; A.lambda = c->lambda; A.mu = c->mu;
  A.tab = c->d_table; A.maps = c->d_gmaps; A.lay = *c->gather_lay; A.X0 = c->d_X0; A.x = c->d_x;
  A.K = c->d_K; A.f = c->d_f; A.bad = c->d_flag + 1; A.stamps = nullptr; A.ablate = 0;
  if (c->ngchunks <= 0) return FEAHIP_OK;
#ifdef FEAHIP_DEBUG
  static unsigned long long *d_stamps = nullptr;
  static int stamps_cap = 0;
  { const char *e = getenv("FEAHIP_GATHER_ABLATE"); A.ablate = e ? atoi(e) : 0; }
  const char *dbg = getenv("FEAHIP_GATHER_STAMPS");
  if (dbg && atoi(dbg)) {
    if (!d_stamps || stamps_cap < c->ngchunks) {
      if (d_stamps) (void)hipFree(d_stamps);
      (void)hipMalloc((void **)&d_stamps, sizeof(unsigned long long) * 32 * (size_t)c->ngchunks);
      (void)hipMemset(d_stamps, 0, sizeof(unsigned long long) * 32 * (size_t)c->ngchunks);
      stamps_cap = c->ngchunks;
    }
    A.stamps = d_stamps;
  }
#endif
  static int run_len = -1;           // chunks per workgroup run (FEAHIP_GATHER_RUN: tuning only, results unchanged)
  if (run_len < 0) { const char *e = getenv("FEAHIP_GATHER_RUN"); run_len = e && atoi(e) > 0 ? atoi(e) : 16; }
  const int nruns = (c->ngchunks + run_len - 1) / run_len;
  const dim3 grid((nruns + 7) & ~7), blk(FEA_G_THREADS);
  // LDS: coordinates (48 bytes per node slot) | element records, later the K tile (+1 double of alignment slack) and the residual partials
  const int regK = std::max(A.lay.max_elems * GREC, ((A.lay.max_tile * 9 + 3) & ~1) + 3 * FEA_G_THREADS);
  const int regF = std::max(A.lay.max_elems * GREC_F, 3 * FEA_G_THREADS);
  const int ldsK = A.lay.max_nodes * 48 + ((regK + 1) & ~1) * 8, ldsF = A.lay.max_nodes * 48 + ((regF + 1) & ~1) * 8;
  const bool nh = c->model == FEAHIP_MODEL_COMPRESSIBLE_NEOHOOKEAN;
#define G_LAUNCH(K, F, M, LDS)                                                                                         \
  do {                                                                                                               \
    FEA_HIP_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_assemble_gather<K, F, M>),                \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS));                         \
    hipLaunchKernelGGL((k_assemble_gather<K, F, M>), grid, blk, LDS, c->stream, A, run_len);                                  \
  } while (0)
  if (doK && doF) { if (nh) G_LAUNCH(true, true, true, ldsK); else G_LAUNCH(true, true, false, ldsK); }
  else if (doK)   { if (nh) G_LAUNCH(true, false, true, ldsK); else G_LAUNCH(true, false, false, ldsK); }
  else            { if (nh) G_LAUNCH(false, true, true, ldsF); else G_LAUNCH(false, true, false, ldsF); }
#undef G_LAUNCH
  FEA_HIP_CHECK(c, hipGetLastError());
#ifdef FEAHIP_DEBUG
  if (A.stamps) {
    static int calls = 0;
    if (++calls == 50) {
      (void)hipStreamSynchronize(c->stream);
      std::vector<unsigned long long> hst((size_t)c->ngchunks * 32);
      (void)hipMemcpy(hst.data(), A.stamps, hst.size() * 8, hipMemcpyDeviceToHost);
      double sum[4][8] = {};
      for (int i = 0; i < nruns; ++i)
        for (int w = 0; w < 4; ++w)
          for (int q = 0; q < 8; ++q) sum[w][q] += (double)hst[((size_t)i * 4 + w) * 8 + q];
      fprintf(stderr, "[gather stamps] in-kernel clock %.0f MHz (s_memtime / s_memrealtime x 100 MHz over a run), run = %.0f shader cycles for %d chunks\n",
              sum[0][7] > 0 ? 100.0 * sum[0][6] / sum[0][7] : 0.0, sum[0][6] / nruns, run_len);
      for (int w = 0; w < 4; ++w)
        fprintf(stderr, "[gather stamps K=%d F=%d wave %d, per chunk] state %.0f  prefetch+gather %.0f  barrier %.0f  tile %.0f  diag+drain %.0f  writeout %.0f cycles\n",
                (int)doK, (int)doF, w, sum[w][0] / c->ngchunks, sum[w][1] / c->ngchunks, sum[w][2] / c->ngchunks, sum[w][3] / c->ngchunks,
                sum[w][4] / c->ngchunks, sum[w][5] / c->ngchunks);
    }
  }
#endif
  return FEAHIP_OK;
}
