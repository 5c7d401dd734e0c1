// kernels_gather10.hip -- stiffness + residual assembly of 10-node tetrahedra
// (the reference's element: fea_solver.c:873-883 shape gradients, :887-1068
// element stiffness, :1072-1114 residual) in two kernels.
//
// k_state10     thread <-> element, per Gauss point: inverse Jacobian of the current configuration, volume-weighted
//               stress S = w|J| sigma and tangent coefficients vl = w|J| l1, vm = w|J| m1 -- 17 doubles, stored as one
//               144-byte record.  Every element is evaluated ONCE, at full occupancy (inside the assembly chunks this
//               stage ran on a quarter of the lanes behind two dependent 3x3 inversions and a logarithm, and an
//               element was evaluated once per chunk touching it).
// k_assemble_gather10
//               the 4-node kernel of kernels_gather.hip with the Gauss points as an outer loop.  A 256-thread
//               workgroup owns a chunk of up to 64 consecutive block rows (gather10.cpp).  Per Gauss point:
//     expand    two threads per element of the chunk read its state record (prefetched one Gauss point ahead) and
//               write g_k = J^-T dN_k, t_k = vm g_k + S g_k of its ten nodes into the element's LDS record (31
//               pieces of 16 bytes: P_k = (g_kx, g_ky) at k, Q_k = (t_kx, t_ky) at 10+k, Z_k = (g_kz, t_kz) at
//               20+k, (vl, vm) at 30)
//     gather    thread <-> up to five off-diagonal blocks: K_ab += vl g_a (x) g_b + vm g_b (x) g_a + (g_a . t_b) I
//               over the block's contribution list, accumulators in registers across ALL Gauss points; the last
//               two waves also sum, per (row, element) visit, the diagonal block K_aa and the residual
//               f_a -= t_a - vm g_a
//   and at the end the records' LDS becomes the K tile, rows go through it in passes: blocks (and their
//   transposes, the mirror blocks) from the registers, then streamed to the CSR values -- every value written
//   exactly once, no atomics.
#include "fem_device.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

// doubles per element record: 3 NPE + 1 pieces of 16 bytes (31 for ten nodes, 25 for eight)
#define T_RECD(NPE) (6 * (NPE) + 2)
#define T_HDR 18                      // doubles per state record (Ji 9, S 6, vl, vm, pad)
#define T_GMAX 32                     // Gauss points of a rule at most
#define T_GLDS 8                      // rules up to this many points keep their shape-gradient table in LDS

// record (element le, Gauss point g) of a rank with nloc elements: state[g][le][18], so that the 64 records a wave of
// k_state10 produces per Gauss point are 9 216 contiguous bytes (and the records an expand wave reads are neighbours)
__device__ __forceinline__ size_t t_state_index(int le, int g, int nloc)
{
  return ((size_t)g * nloc + le) * T_HDR;
}

struct S10Args {
  int nloc, G, model, row0, row1;
  double lambda, mu;
  const ElemTable *tab;
  const int *elist, *conn;
  const double *X0, *x;
  double *state;
  int *bad;
};

// thread <-> element, its Gauss points one after the other with the twenty coordinate triples held in registers (one
// thread per (element, Gauss point) gathered them G times: 0.18 ms of the 1.11 ms assembly at G = 5)
template <int NPE>
__global__ __launch_bounds__(256)
void k_state10(S10Args A)
{
  __shared__ double sTd[T_GMAX * 3 * NPE];
  __shared__ double sTw[T_GMAX];
  for (int i = threadIdx.x; i < A.G; i += 256) sTw[i] = A.tab->w[i];
  for (int i = threadIdx.x; i < A.G * 3 * NPE; i += 256) sTd[i] = A.tab->dN[i / (3 * NPE)][(i / NPE) % 3][i % NPE];
  __syncthreads();
  __shared__ __attribute__((aligned(16))) double sOut[4][64 * T_HDR];       // a wave's records of one Gauss point
  const int le_raw = blockIdx.x * 256 + threadIdx.x;
  const bool act = le_raw < A.nloc;
  const int le = act ? le_raw : A.nloc - 1;                                 // the last wave: spare lanes repeat the last element, write nothing
  const int e = A.elist[le];
  const int *cn = A.conn + (size_t)e * NPE;
  double xc[NPE][3], Xc[NPE][3];
  int n0 = 0;
#pragma unroll
  for (int k = 0; k < NPE; ++k) {
    const size_t n = (size_t)cn[k];
    if (k == 0) n0 = (int)n;
    const double2 a0 = *reinterpret_cast<const double2 *>(A.x + n * 4), c0 = *reinterpret_cast<const double2 *>(A.X0 + n * 4);
    xc[k][0] = a0.x; xc[k][1] = a0.y; xc[k][2] = A.x[n * 4 + 2];
    Xc[k][0] = c0.x; Xc[k][1] = c0.y; Xc[k][2] = A.X0[n * 4 + 2];
  }
  const bool mine = n0 >= A.row0 && n0 < A.row1;           // inverted points are counted once per mesh: by the rank that owns the first node
  int nbad = 0;
  for (int g = 0; g < A.G; ++g) {
    double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    const double *td = sTd + g * 3 * NPE;
#pragma unroll
    for (int k = 0; k < NPE; ++k)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double dn_ = td[i * NPE + k];
#pragma unroll
        for (int j = 0; j < 3; ++j) { J[i][j] = fma(dn_, xc[k][j], J[i][j]); M[i][j] = fma(dn_, Xc[k][j], M[i][j]); }
      }
    double Ji[3][3], detJ;
    fd_inv3(J, Ji, detJ);
    // F^-1 = sum_k X_k (x) g_k with g_k = Ji dN_k  =>  Finv[i][j] = sum_m M[m][i] Ji[j][m]
    double Fi[3][3], F[3][3], detFi;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Fi[i][j] = M[0][i] * Ji[j][0] + M[1][i] * Ji[j][1] + M[2][i] * Ji[j][2];
    fd_inv3(Fi, F, detFi);
    double sig[3][3], l1, m1;
    fd_constitutive(F, A.model, A.lambda, A.mu, sig, l1, m1);
    nbad += act && !(detJ > 0.0);
    double st[T_HDR];
#pragma unroll
    for (int i = 0; i < T_HDR; ++i) st[i] = 0.0;
    if (detJ != 0.0) {                                       // fea_solver.c:697: no gradient otherwise
      const double vol = sTw[g] * fabs(detJ);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int m = 0; m < 3; ++m) st[3 * i + m] = Ji[i][m];
      st[9] = vol * sig[0][0]; st[10] = vol * sig[0][1]; st[11] = vol * sig[0][2];
      st[12] = vol * sig[1][1]; st[13] = vol * sig[1][2]; st[14] = vol * sig[2][2];
      st[15] = vol * l1; st[16] = vol * m1;
    }
    {
      // a lane storing its own 144-byte record touches a cache line of its own with every store instruction (64 lines
      // per instruction, nine instructions per point); through LDS the wave writes 1 KB per instruction
      const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
      double2 *so = reinterpret_cast<double2 *>(sOut[wave] + lane * T_HDR);
#pragma unroll
      for (int i = 0; i < T_HDR / 2; ++i) so[i] = make_double2(st[2 * i], st[2 * i + 1]);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      const int le0 = blockIdx.x * 256 + wave * 64;
      const int np = (A.nloc - le0 < 64 ? A.nloc - le0 : 64) * (T_HDR / 2);
      double2 *o = reinterpret_cast<double2 *>(A.state + t_state_index(le0, g, A.nloc));
      const double2 *si = reinterpret_cast<const double2 *>(sOut[wave]);
#pragma unroll
      for (int j = 0; j < T_HDR / 2; ++j) { const int p = lane + 64 * j; if (p < np) o[p] = si[p]; }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
  }
  if (A.bad && mine && nbad) atomicAdd(A.bad, nbad);
}

struct G10Args {
  int nchunks, G;
  const ElemTable *tab;
  const unsigned char *maps;
  Gather10Layout lay;
  const double *state;
  int nloc;                     // elements of the rank (state records per Gauss point)
  double *K, *f;
  unsigned long long *stamps;   // diagnostic build only
};
#ifdef FEAHIP_DEBUG
#define T_STAMP(i) do { if (A.stamps) { const unsigned long long _t = __builtin_amdgcn_s_memtime(); qa[i] += _t - qt; qt = _t; } } while (0)
#else
#define T_STAMP(i) do { } while (0)
#endif

typedef double t_v2d __attribute__((ext_vector_type(2)));
#ifndef T_NT_ROWS
#define T_NT_ROWS 1
#endif
#if T_NT_ROWS
#define T_NT_STORE(ptr, v) do { const t_v2d nt_ = {(v).x, (v).y}; __builtin_nontemporal_store(nt_, reinterpret_cast<t_v2d *>(ptr)); } while (0)
#else
#define T_NT_STORE(ptr, v) do { *reinterpret_cast<double2 *>(ptr) = (v); } while (0)
#endif
#define T_BARRIER() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); } while (0)

// one contribution (element le, local row node la, local column node lb) to a block
template <int NPE>
__device__ __forceinline__ void t_entry(const unsigned char *sRb, uint32_t w, double (&acc)[9])
{
  const uint32_t le = w & 127u, la = (w >> 7) & 15u, lb = (w >> 11) & 15u;
  const unsigned char *base = sRb + le * (T_RECD(NPE) * 8u);
  const double2 Pa = *reinterpret_cast<const double2 *>(base + la * 16u);
  const double Zax = *reinterpret_cast<const double *>(base + 32u * NPE + la * 16u);
  const double2 Pb = *reinterpret_cast<const double2 *>(base + lb * 16u);
  const double2 Qb = *reinterpret_cast<const double2 *>(base + 16u * NPE + lb * 16u);
  const double2 Zb = *reinterpret_cast<const double2 *>(base + 32u * NPE + lb * 16u);
  const double2 VV = *reinterpret_cast<const double2 *>(base + 48u * NPE);
  const double ga[3] = {Pa.x, Pa.y, Zax}, gb[3] = {Pb.x, Pb.y, Zb.x};
  const double d = ga[0] * Qb.x + ga[1] * Qb.y + ga[2] * Zb.y;
  const double A_[3] = {VV.x * ga[0], VV.x * ga[1], VV.x * ga[2]};
  const double B_[3] = {VV.y * ga[0], VV.y * ga[1], VV.y * ga[2]};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[3 * i + j] = fma(A_[i], gb[j], fma(gb[i], B_[j], acc[3 * i + j]));
  acc[0] += d; acc[4] += d; acc[8] += d;
  __builtin_amdgcn_sched_barrier(0);      // one entry's loads in flight at a time: hoisting the next ones costs registers the accumulators need
}

// the contributions of one block slot: `cnt` entries (wave-uniform), the first 2*FEA_Q_REGW from registers
template <int NPE>
__device__ __forceinline__ void t_slot(const unsigned char *sRb, const uint32_t (&cw)[FEA_Q_REGW], const uint32_t *more, int cnt, double (&acc)[9])
{
#pragma unroll
  for (int k = 0; k < FEA_Q_REGW; ++k) {
    if (2 * k < cnt) t_entry<NPE>(sRb, cw[k] & 0xFFFFu, acc);
    if (2 * k + 1 < cnt) t_entry<NPE>(sRb, cw[k] >> 16, acc);
  }
  for (int k = FEA_Q_REGW; 2 * k < cnt; ++k) {          // lists longer than the registers hold: the rest from memory
    const uint32_t w = more[(size_t)k * FEA_Q_THREADS];
    t_entry<NPE>(sRb, w & 0xFFFFu, acc);
    if (2 * k + 1 < cnt) t_entry<NPE>(sRb, w >> 16, acc);
  }
}

// one (row node, element) visit: diagonal block K_aa = (vl + vm) g_a (x) g_a + (g_a . t_a) I (symmetric: 00 01 02 11 12
// 22) and the residual f_a -= S g_a.  S g_a is formed from S here, not as t_a - vm g_a: that difference carries the
// rounding of vm g_a (the stiffness scale) into a quantity of the stress scale, which showed as a convergence floor
// of <u, R> ~ 1e-16 where the other kernels reach 1e-26 on the reference's analytical decks.
template <int NPE, bool DOF>
__device__ __forceinline__ void t_visit(const unsigned char *sRb, const double *sS, uint32_t w, double (&kd)[6], double (&fa)[3])
{
  const uint32_t le = w & 127u, la = (w >> 7) & 15u;
  const unsigned char *base = sRb + le * (T_RECD(NPE) * 8u);
  const double2 Pa = *reinterpret_cast<const double2 *>(base + la * 16u);
  const double gz = *reinterpret_cast<const double *>(base + 32u * NPE + la * 16u);
  const double2 VV = *reinterpret_cast<const double2 *>(base + 48u * NPE);
  const double2 s0 = *reinterpret_cast<const double2 *>(sS + le * 6), s1 = *reinterpret_cast<const double2 *>(sS + le * 6 + 2),
                s2 = *reinterpret_cast<const double2 *>(sS + le * 6 + 4);      // 00 01 | 02 11 | 12 22
  const double sx = s0.x * Pa.x + s0.y * Pa.y + s1.x * gz;
  const double sy = s0.y * Pa.x + s1.y * Pa.y + s2.x * gz;
  const double sz = s1.x * Pa.x + s2.x * Pa.y + s2.y * gz;
  const double c = VV.x + VV.y;
  const double d = VV.y * (Pa.x * Pa.x + Pa.y * Pa.y + gz * gz) + (Pa.x * sx + Pa.y * sy + gz * sz);
  const double cx = c * Pa.x, cy = c * Pa.y, cz = c * gz;
  kd[0] += fma(cx, Pa.x, d); kd[1] = fma(cx, Pa.y, kd[1]); kd[2] = fma(cx, gz, kd[2]);
  kd[3] += fma(cy, Pa.y, d); kd[4] = fma(cy, gz, kd[4]); kd[5] += fma(cz, gz, d);
  if (DOF) { fa[0] -= sx; fa[1] -= sy; fa[2] -= sz; }
}

// DOK = false: the residual alone -- expand and the visit lanes only, no blocks, no tile
// TLDS: the shape-gradient table in LDS (rules of up to T_GLDS points); the 27-point rule reads it through the cache -- 8.6 KB
// more LDS would cost the second workgroup of a CU
template <int NPE, bool DOK, bool DOF, bool TLDS>
__global__ __launch_bounds__(FEA_Q_THREADS, FEA_Q_THREADS / 128)
void k_assemble_gather10(G10Args A, int run_len)
{
  constexpr int T_REC = T_RECD(NPE);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int mxe = A.lay.max_elems, G = A.G;
  double *sTd = reinterpret_cast<double *>(smem);                // [G][10][4]: dN/dxi of node k at Gauss point g, pad (TLDS)
  double *sFp = sTd + (TLDS ? G * NPE * 4 : 0);                       // [FLANES][9] diagonal block (6) + residual (3) partials
  double *sS = sFp + FEA_Q_FLANES * 9;                           // [mxe + 1][6] volume-weighted stress 00 01 02 11 12 22
  uint16_t *sRows = reinterpret_cast<uint16_t *>(sS + (mxe + 1) * 6);        // [208]
  double *sR = reinterpret_cast<double *>(sRows + 208);          // [mxe + 1][62]; record mxe stays all-zero
  const unsigned char *sRb = reinterpret_cast<const unsigned char *>(sR);

  const int nruns = (A.nchunks + run_len - 1) / run_len;
  const int per = ((int)gridDim.x + 7) >> 3;
  const int ridx = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);     // XCD-contiguous run order
  if (ridx >= nruns) return;
  int chunk = ridx * run_len;
  const int cend = min(A.nchunks, chunk + run_len);
#ifdef FEAHIP_DEBUG
  unsigned long long qa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, qt = __builtin_amdgcn_s_memtime();
#endif
  for (int i = t; i < (TLDS ? G * NPE * 4 : 0); i += FEA_Q_THREADS) {
    const int g = i / (NPE * 4), k = (i % (NPE * 4)) >> 2, c = i & 3;
    sTd[i] = c < 3 ? A.tab->dN[g][c][k] : 0.0;
  }
  if (t < T_REC) sR[mxe * T_REC + t] = 0.0;
  if (t < 6) sS[mxe * 6 + t] = 0.0;
  const uint32_t ZZ = (uint32_t)mxe | ((uint32_t)mxe << 16);
  const bool vlane = t >= FEA_Q_THREADS - FEA_Q_FLANES;
  const int fl = t - (FEA_Q_THREADS - FEA_Q_FLANES);

  for (; chunk < cend; ++chunk) {
    const unsigned char *rec = A.maps + (size_t)chunk * A.lay.stride;
    const Gather10Header *hp = reinterpret_cast<const Gather10Header *>(rec);
    const int r0 = hp->r0, r1 = hp->r1, b0 = hp->b0, nelem = hp->nelem, npass = hp->npass;
    const int fdw = hp->fdw;
    const int nrows = r1 - r0;
    const uint32_t prow_lo = reinterpret_cast<const uint32_t *>(hp->prow)[0], prow_hi = reinterpret_cast<const uint32_t *>(hp->prow)[1];
    int cnt[FEA_Q_SLOTS], srow[FEA_Q_SLOTS];             // this wave's list lengths; first clist row of every slot
    {
      int row = 0;
#pragma unroll
      for (int s = 0; s < FEA_Q_SLOTS; ++s) { cnt[s] = hp->cnt[FEA_Q_WAVES * s + wv]; srow[s] = row; row += hp->sw[s]; }
    }
    // ---- stage the chunk: row table; this thread's element, block positions and lists
    // expand: `parts` threads per element, thread xp of them takes the nodes xp, xp + parts, ...
    const int parts = 4 * nelem <= FEA_Q_THREADS ? 4 : (3 * nelem <= FEA_Q_THREADS ? 3 : 2);
    const int xe = parts == 4 ? t >> 2 : (parts == 3 ? (int)(((unsigned)t * 21846u) >> 16) : t >> 1), xp = t - xe * parts;
    const bool xact = xe < nelem;
    const int xle = (int)reinterpret_cast<const uint32_t *>(rec + A.lay.o_elems)[xact ? xe : 0];
    const double2 *srec = reinterpret_cast<const double2 *>(A.state + t_state_index(xle, 0, A.nloc));
    const size_t gstride = (size_t)A.nloc * (T_HDR / 2);   // double2 between the Gauss points of an element
    if (t < FEA_Q_ROWS_U16 / 2) reinterpret_cast<uint32_t *>(sRows)[t] = reinterpret_cast<const uint32_t *>(rec + A.lay.o_rows)[t];
    uint32_t tp[FEA_Q_SLOTS], cw[FEA_Q_SLOTS][FEA_Q_REGW], fw[FEA_Q_REGW];
    const uint32_t *cl = reinterpret_cast<const uint32_t *>(rec + A.lay.o_clist) + t;
#pragma unroll
    for (int s = 0; s < FEA_Q_SLOTS; ++s) {
      tp[s] = DOK ? reinterpret_cast<const uint32_t *>(rec + A.lay.o_tpos)[s * FEA_Q_THREADS + t] : 0xFFFFFFFFu;
#pragma unroll
      for (int k = 0; k < FEA_Q_REGW; ++k) cw[s][k] = (DOK && 2 * k < cnt[s]) ? cl[(size_t)(srow[s] + k) * FEA_Q_THREADS] : ZZ;
    }
#pragma unroll
    for (int k = 0; k < FEA_Q_REGW; ++k)
      fw[k] = (vlane && k < fdw) ? reinterpret_cast<const uint32_t *>(rec + A.lay.o_flist)[k * FEA_Q_FLANES + fl] : ZZ;
    double acc[FEA_Q_SLOTS][9], kd[6] = {0, 0, 0, 0, 0, 0}, fa[3] = {0, 0, 0};
#pragma unroll
    for (int s = 0; s < FEA_Q_SLOTS; ++s)
#pragma unroll
      for (int q = 0; q < 9; ++q) acc[s][q] = 0.0;
    double2 hn[T_HDR / 2];                               // state record of the Gauss point to expand next
#pragma unroll
    for (int i = 0; i < T_HDR / 2; ++i) hn[i] = srec[i];
    T_STAMP(0);

    for (int g = 0; g < G; ++g) {
      // ---- expand: this thread's element, five of its nodes -> record pieces
#ifdef FEAHIP_DEBUG
      if (A.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); T_STAMP(7); }
#endif
      __builtin_amdgcn_s_setprio(2);                     // wave priorities by phase: the short LDS-bound expand ahead of the other
      if (xact) {                                        // workgroup's long gather (0), the write-out ahead of both (3): 1 %
        const double2 (&h)[T_HDR / 2] = hn;
        const double vl = h[7].y, vm = h[8].x;
        double *r = sR + xe * T_REC;
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
          const int k = xp + kk * parts;
          if (k >= NPE) break;
          double2 d01; double d2;
          if (TLDS) {
            const double *td = sTd + (g * NPE + k) * 4;
            d01 = *reinterpret_cast<const double2 *>(td); d2 = td[2];
          } else {
            d01 = make_double2(A.tab->dN[g][0][k], A.tab->dN[g][1][k]); d2 = A.tab->dN[g][2][k];
          }
          // g_i = sum_m Ji[i][m] dN[m];  S = (h4.y h5.x h5.y; . h6.x h6.y; . . h7.x)
          const double gx = h[0].x * d01.x + h[0].y * d01.y + h[1].x * d2;
          const double gy = h[1].y * d01.x + h[2].x * d01.y + h[2].y * d2;
          const double gz = h[3].x * d01.x + h[3].y * d01.y + h[4].x * d2;
          const double sx = h[4].y * gx + h[5].x * gy + h[5].y * gz;
          const double sy = h[5].x * gx + h[6].x * gy + h[6].y * gz;
          const double sz = h[5].y * gx + h[6].y * gy + h[7].x * gz;
          const double tx = __dadd_rn(__dmul_rn(vm, gx), sx), ty = __dadd_rn(__dmul_rn(vm, gy), sy), tz = __dadd_rn(__dmul_rn(vm, gz), sz);
          *reinterpret_cast<double2 *>(r + 2 * k) = make_double2(gx, gy);
          *reinterpret_cast<double2 *>(r + 2 * NPE + 2 * k) = make_double2(tx, ty);
          *reinterpret_cast<double2 *>(r + 4 * NPE + 2 * k) = make_double2(gz, tz);
        }
        if (xp == 0) {
          *reinterpret_cast<double2 *>(r + 6 * NPE) = make_double2(vl, vm);
          double *ss = sS + xe * 6;
          *reinterpret_cast<double2 *>(ss) = make_double2(h[4].y, h[5].x);
          *reinterpret_cast<double2 *>(ss + 2) = make_double2(h[5].y, h[6].x);
          *reinterpret_cast<double2 *>(ss + 4) = make_double2(h[6].y, h[7].x);
        }
      }
      {                                                  // the next Gauss point's record: in flight under the gather
        const double2 *nx = srec + (size_t)(g + 1 < G ? g + 1 : g) * gstride;
#pragma unroll
        for (int i = 0; i < T_HDR / 2; ++i) hn[i] = nx[i];
      }
      T_STAMP(1);
      T_BARRIER();
      T_STAMP(2);
      __builtin_amdgcn_s_setprio(0);
      // ---- gather: this thread's blocks; the visit lanes' diagonal blocks and residuals
      // (the list words are made opaque per Gauss point: hoisted out of this loop, the LDS addresses decoded from them
      // -- some 200 registers' worth -- were spilled to scratch and reloaded here)
#pragma unroll
      for (int s = 0; s < FEA_Q_SLOTS; ++s)
#pragma unroll
        for (int k = 0; k < FEA_Q_REGW; ++k) asm volatile("" : "+v"(cw[s][k]));
#pragma unroll
      for (int k = 0; k < FEA_Q_REGW; ++k) asm volatile("" : "+v"(fw[k]));
      if (DOK) {
#pragma unroll
        for (int s = 0; s < FEA_Q_SLOTS; ++s) t_slot<NPE>(sRb, cw[s], cl + (size_t)srow[s] * FEA_Q_THREADS, cnt[s], acc[s]);
      }
      if (vlane) {
#pragma unroll
        for (int k = 0; k < FEA_Q_REGW; ++k)
          if (k < fdw) { t_visit<NPE, DOF>(sRb, sS, fw[k] & 0xFFFFu, kd, fa); t_visit<NPE, DOF>(sRb, sS, fw[k] >> 16, kd, fa); }
      }
      T_STAMP(3);
      if (g + 1 < G) T_BARRIER();                        // the records have been read
      T_STAMP(4);
    }
    // ---- write-out
    __builtin_amdgcn_s_setprio(3);
    if (vlane) {
      double *o = sFp + fl * 9;
      o[0] = kd[0]; o[1] = kd[1]; o[2] = kd[2]; o[3] = kd[3]; o[4] = kd[4]; o[5] = kd[5];
      o[6] = fa[0]; o[7] = fa[1]; o[8] = fa[2];
    }
    T_BARRIER();                                         // the records have been read: their LDS becomes the K tile
    T_STAMP(5);
    // per (row, component): the sum over the row's visit lanes; components 0-5 the diagonal block (00 01 02 11 12 22),
    // 6-8 the residual
    double dsum[3] = {0, 0, 0};
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      const int it = t + rr * FEA_Q_THREADS;
      if (it < nrows * 9) {
        const int row = it / 9, c = it - row * 9;
        double s = 0.0;
        for (int l = sRows[130 + row]; l < sRows[130 + row + 1]; ++l) s += sFp[l * 9 + c];
        dsum[rr] = s;
        if (DOF && c >= 6) A.f[(size_t)(r0 + row) * 3 + (c - 6)] = s;
      }
    }
    for (int p = 0; p < (DOK ? npass : 0); ++p) {
      const int rlo = (int)(((p < 4 ? prow_lo >> (8 * p) : prow_hi >> (8 * (p - 4)))) & 255u);
      const int rhi = (int)(((p + 1 < 4 ? prow_lo >> (8 * (p + 1)) : prow_hi >> (8 * (p - 3)))) & 255u);
      const int pb0 = sRows[rlo], pb1 = sRows[rhi];
      const int odd = (b0 + pb0) & 1;
      double *sT = sR + odd;
#pragma unroll
      for (int s = 0; s < FEA_Q_SLOTS; ++s) {
        const int bpos = (int)(tp[s] & 0xFFFFu), mpos = (int)(tp[s] >> 16);
        if (bpos >= pb0 && bpos < pb1) {
          double *d = sT + (bpos - pb0) * 9;
#pragma unroll
          for (int q = 0; q < 9; ++q) d[q] = acc[s][q];
        }
        if (mpos >= pb0 && mpos < pb1) {
          double *d = sT + (mpos - pb0) * 9;
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) d[3 * j + i] = acc[s][3 * i + j];
        }
      }
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        const int it = t + rr * FEA_Q_THREADS;
        const int row = it / 9, c = it - row * 9;
        if (it < nrows * 9 && c < 6 && row >= rlo && row < rhi) {
          // 00 01 02 11 12 22 -> positions (0) (1,3) (2,6) (4) (5,7) (8) of the 3x3 block
          double *d = sT + (sRows[66 + row] - pb0) * 9;
          const int q0 = c < 3 ? c : (c < 5 ? c + 1 : 8), q1 = c == 1 ? 3 : (c == 2 ? 6 : (c == 4 ? 7 : q0));
          d[q0] = dsum[rr]; d[q1] = dsum[rr];
        }
      }
      T_BARRIER();
      {
        double *Kd = A.K + (size_t)(b0 + pb0) * 9;
        const int total = (pb1 - pb0) * 9;
        if (odd && t == 0) Kd[0] = sT[0];
        const int npair2 = (total - odd) >> 1;
        int i = t;
        for (; i + 3 * FEA_Q_THREADS < npair2; i += 4 * FEA_Q_THREADS) {      // four tile reads in flight
          const int j = odd + 2 * i;
          const double2 v0 = *reinterpret_cast<const double2 *>(sT + j), v1 = *reinterpret_cast<const double2 *>(sT + j + 2 * FEA_Q_THREADS),
                        v2 = *reinterpret_cast<const double2 *>(sT + j + 4 * FEA_Q_THREADS), v3 = *reinterpret_cast<const double2 *>(sT + j + 6 * FEA_Q_THREADS);
          T_NT_STORE(Kd + j, v0); T_NT_STORE(Kd + j + 2 * FEA_Q_THREADS, v1);      // written once, read by other kernels only: non-temporal
          T_NT_STORE(Kd + j + 4 * FEA_Q_THREADS, v2); T_NT_STORE(Kd + j + 6 * FEA_Q_THREADS, v3);
        }
        for (; i < npair2; i += FEA_Q_THREADS) {
          const int j = odd + 2 * i;
          const double2 v = *reinterpret_cast<const double2 *>(sT + j);
          T_NT_STORE(Kd + j, v);
        }
        if (((total - odd) & 1) && t == 0) Kd[total - 1] = sT[total - 1];
      }
      T_BARRIER();                                       // the tile has been read
    }
    if (!DOK) T_BARRIER();                               // the partials have been read
    T_STAMP(6);
  }
#ifdef FEAHIP_DEBUG
  if (A.stamps && (t & 63) == 0) {
    unsigned long long *o = A.stamps + ((size_t)ridx * FEA_Q_WAVES + (t >> 6)) * 8;
    for (int i = 0; i < 8; ++i) o[i] = qa[i];
  }
#endif
}

int ensure_gather10(feahip_ctx *c)
{
  if (c->have_gather && c->gather10_lay && c->gather_row0 == c->row0 && c->gather_row1 == c->row1) return FEAHIP_OK;
  if (!c->h_pat || c->h_conn.empty() || (c->npe != 10 && c->npe != 8) || c->G > T_GMAX) return FEAHIP_OK;
  // a new row range (re-shard): the old maps describe rows K no longer holds (release_k re-allocates the window), so
  // they go before anything else can launch them -- also when the new range is known not to fit, or turns out not to
  for (void *p : {(void *)c->d_gmaps, (void *)c->d_g10_elist, (void *)c->d_g10_state})
    if (p) (void)hipFree(p);
  c->d_gmaps = nullptr; c->d_g10_elist = nullptr; c->d_g10_state = nullptr;
  c->have_gather = false; c->ngchunks = 0; c->g10_nloc = 0; c->gather_row0 = c->gather_row1 = -1;
  if (c->gather_failed && c->gather_fail_row0 == c->row0 && c->gather_fail_row1 == c->row1) return FEAHIP_OK;
  HostGather10 hg;
  build_host_gather10(c->N, c->E, c->npe, c->h_conn.data(), *c->h_pat, c->row0, c->row1, hg);
  if (!hg.ok) { c->gather_failed = true; c->gather_fail_row0 = c->row0; c->gather_fail_row1 = c->row1; return FEAHIP_OK; }   // this row range only: another shard of the same context may fit
  c->g10_nloc = (int)hg.elist.size();
  const size_t state_bytes = sizeof(double) * T_HDR * (size_t)c->G * (size_t)std::max(c->g10_nloc, 1);
  FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_gmaps, hg.blob.size() ? hg.blob.size() : 1));
  FEA_HIP_CHECK(c, hipMemcpy(c->d_gmaps, hg.blob.data(), hg.blob.size(), hipMemcpyHostToDevice));
  FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_g10_elist, sizeof(int) * (size_t)std::max(c->g10_nloc, 1)));
  FEA_HIP_CHECK(c, hipMemcpy(c->d_g10_elist, hg.elist.data(), sizeof(int) * hg.elist.size(), hipMemcpyHostToDevice));
  FEA_HIP_CHECK(c, hipMalloc((void **)&c->d_g10_state, state_bytes));
  FEA_HIP_CHECK(c, hipMemset(c->d_g10_state, 0, state_bytes));
  if (!c->gather10_lay) c->gather10_lay = new Gather10Layout();
  *c->gather10_lay = hg.lay;
  c->ngchunks = hg.nchunks;
  c->gather_row0 = c->row0; c->gather_row1 = c->row1;
  c->gather_bytes = (long long)hg.blob.size() + (long long)state_bytes + 4LL * c->g10_nloc;
  c->gather_evals_per_element = hg.distinct_elems ? (double)hg.total_evals / (double)hg.distinct_elems : 0.0;
  c->have_gather = true;
#ifdef FEAHIP_DEBUG
  fprintf(stderr, "[gather10] %d chunks over %d rows, an element in %.2f chunks, map record %d bytes, max elems %d list rows %d visit words %d tile %d blocks, %d elements\n",
          hg.nchunks, c->row1 - c->row0, c->gather_evals_per_element, hg.lay.stride, hg.lay.max_elems, hg.lay.max_cw, hg.lay.max_fdw, hg.lay.tile_blocks, c->g10_nloc);
#endif
  return FEAHIP_OK;
}

static int gather10_lds_bytes(const Gather10Layout &lay, int G, int npe)
{
  return (G <= T_GLDS ? G * npe * 4 * 8 : 0) + FEA_Q_FLANES * 9 * 8 + (lay.max_elems + 1) * 6 * 8 + 208 * 2 + (lay.max_elems + 1) * T_RECD(npe) * 8;
}

int launch_assemble_gather10(feahip_ctx *c, bool doK, bool doF)
{
  if (c->ngchunks <= 0) return FEAHIP_OK;
  {
    S10Args S;
    S.nloc = c->g10_nloc; S.G = c->G; S.model = c->model; S.row0 = c->row0; S.row1 = c->row1; S.lambda = c->lambda; S.mu = c->mu;
    S.tab = c->d_table; S.elist = c->d_g10_elist; S.conn = c->d_conn; S.X0 = c->d_X0; S.x = c->d_x;
    S.state = c->d_g10_state; S.bad = doK ? c->d_flag + 1 : nullptr;      // the counter is reset by stiffness assemblies only
    if (c->g10_nloc > 0) {
      if (c->npe == 10) hipLaunchKernelGGL(k_state10<10>, dim3((unsigned)((c->g10_nloc + 255) / 256)), dim3(256), 0, c->stream, S);
      else              hipLaunchKernelGGL(k_state10<8>, dim3((unsigned)((c->g10_nloc + 255) / 256)), dim3(256), 0, c->stream, S);
    }
  }
  G10Args A;
  A.nchunks = c->ngchunks; A.G = c->G; A.tab = c->d_table; A.maps = c->d_gmaps; A.lay = *c->gather10_lay;
  A.state = c->d_g10_state; A.nloc = c->g10_nloc; A.K = c->d_K; A.f = c->d_f; A.stamps = nullptr;
#ifdef FEAHIP_DEBUG
  static unsigned long long *d_stamps = nullptr;
  static int cap = 0;
  if (getenv("FEAHIP_GATHER10_STAMPS")) {
    if (!d_stamps || cap < c->ngchunks) { if (d_stamps) (void)hipFree(d_stamps); (void)hipMalloc((void **)&d_stamps, 8 * 32 * (size_t)c->ngchunks); cap = c->ngchunks; }
    (void)hipMemset(d_stamps, 0, 8 * 32 * (size_t)c->ngchunks);
    A.stamps = d_stamps;
  }
#endif
  static int run_len = -1;           // chunks per workgroup run (FEAHIP_GATHER10_RUN: tuning only, results unchanged)
  if (run_len < 0) { const char *e = getenv("FEAHIP_GATHER10_RUN"); run_len = e && atoi(e) > 0 ? atoi(e) : 2; }
  const int nruns = (c->ngchunks + run_len - 1) / run_len;
  const dim3 grid((nruns + 7) & ~7), blk(FEA_Q_THREADS);
  const int lds = gather10_lds_bytes(A.lay, c->G, c->npe);
  const bool tl = c->G <= T_GLDS;
  const int variant = ((c->npe == 10 ? 0 : 1) * 3 + (!doK ? 2 : (doF ? 1 : 0))) * 2 + (tl ? 1 : 0);
  typedef void (*g10_fn)(G10Args, int);
#define G10_ROW(N_) k_assemble_gather10<N_, true, false, false>, k_assemble_gather10<N_, true, false, true>, k_assemble_gather10<N_, true, true, false>, \
                    k_assemble_gather10<N_, true, true, true>, k_assemble_gather10<N_, false, true, false>, k_assemble_gather10<N_, false, true, true>
  static const g10_fn fns[12] = {G10_ROW(10), G10_ROW(8)};
#undef G10_ROW
  // on every launch: the attribute belongs to the current device and to the size asked for (another device of an
  // in-process group, or a later context with a longer Gauss rule, must not inherit the first caller's value)
  FEA_HIP_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(fns[variant]), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(fns[variant], grid, blk, lds, c->stream, A, run_len);
  FEA_HIP_CHECK(c, hipGetLastError());
#ifdef FEAHIP_DEBUG
  if (A.stamps) {
    static int calls = 0;
    if (++calls == 8) {
      (void)hipStreamSynchronize(c->stream);
      std::vector<unsigned long long> h((size_t)c->ngchunks * 8 * FEA_Q_WAVES);
      (void)hipMemcpy(h.data(), A.stamps, h.size() * 8, hipMemcpyDeviceToHost);
      for (int w = 0; w < 4; ++w) {
        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < nruns; ++i) for (int q = 0; q < 8; ++q) sum[q] += (double)h[((size_t)i * FEA_Q_WAVES + w) * 8 + q];
        fprintf(stderr, "[gather10 stamps wave %d, per chunk] stage %.0f  state wait %.0f  expand %.0f  barrier %.0f  gather %.0f  barrier %.0f  partials+barrier %.0f  write-out %.0f\n",
                w, sum[0] / c->ngchunks, sum[7] / c->ngchunks, sum[1] / c->ngchunks, sum[2] / c->ngchunks, sum[3] / c->ngchunks, sum[4] / c->ngchunks, sum[5] / c->ngchunks, sum[6] / c->ngchunks);
      }
    }
  }
#endif
  return FEAHIP_OK;
}
