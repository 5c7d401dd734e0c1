// kernels_gather_pc.hip -- GATHER assembly of linear tetrahedra with producer and consumer waves
// (stiffness and stiffness + residual; replaces fea_solver.c:873-883, :887-1068 and :1072-1114 for TETRAHEDRA4
// meshes; same host maps as kernels_gather.hip, gather.cpp; records and algebra of gather_device.h).
//
// kernels_gather.hip walks a chunk in four barrier-separated phases (state, gather, tile, write-out), every wave
// taking part in every phase; PMC showed the LDS pipe and the FP64 pipe each about half busy, alternating, with
// 42 % of the wave-cycles parked at the phase barriers.  Here the two halves of a 512-thread workgroup do
// different things between two barriers:
//   waves 0-3 (producers)  thread <-> element of chunk i+1: state evaluation from the coordinate tile, the
//                          13-piece record into record buffer (i+1) mod 2; the coordinates of chunk i+2 from HBM
//                          into the coordinate tile that chunk i used;
//   waves 4-7 (consumers)  the finished rows of chunk i-1 from tile (i-1) mod 2 to HBM (16-byte stores); thread
//                          <-> off-diagonal block of chunk i: its contribution list summed in registers out of
//                          record buffer i mod 2, block and mirror block into tile i mod 2; the last wave sums the
//                          diagonal blocks and the residual of the rows.
// One barrier per chunk; a workgroup per CU (records, tiles and coordinates are double-buffered: ~140 KB of LDS);
// waves w and w+4 share a SIMD, so every SIMD has one FP64-bound and one LDS-bound instruction stream to pick from.
// No atomics, fixed summation order: bitwise reproducible, and the same bits as kernels_gather.hip (same records,
// same lists, same order).
#include "gather_device.h"

#define PC_THREADS 768

#ifdef FEAHIP_DEBUG
#define PC_STAMP(v) do { if (A.stamps) (v) = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PC_STAMP(v) do { } while (0)
#endif

struct PcHeader { int r0, r1, b0, nb, nelem, noffd, depth, ddepth; };

__device__ __forceinline__ PcHeader pc_decode(int hword)
{
  PcHeader h;
  h.r0 = __builtin_amdgcn_readlane(hword, 0); h.r1 = __builtin_amdgcn_readlane(hword, 1);
  h.b0 = __builtin_amdgcn_readlane(hword, 2); h.nb = __builtin_amdgcn_readlane(hword, 3);
  h.nelem = __builtin_amdgcn_readlane(hword, 5); h.noffd = __builtin_amdgcn_readlane(hword, 6);
  h.depth = __builtin_amdgcn_readlane(hword, 7); h.ddepth = __builtin_amdgcn_readlane(hword, 10);
  return h;
}

// one state evaluation per element slot of a chunk: coordinates from the tile sC, record into sT (slot p)
template <bool NH>
__device__ __forceinline__ void pc_state(const GatherArgs &A, const unsigned char *rec, unsigned eids, const PcHeader &h,
                                         const double2 *sC, double *sT, int p, double gauss_w, int mode = 0)
{
  if (p < h.nelem && eids != 0xFFFFFFFFu) {
    const int nd[4] = {(int)(eids & 255u), (int)((eids >> 8) & 255u), (int)((eids >> 16) & 255u), (int)(eids >> 24)};
    double xe[4][3], Xe[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double2 *cc = sC + nd[k] * 3;
      const double2 p0 = cc[0], p1 = cc[1], p2 = cc[2];
      xe[k][0] = p0.x; xe[k][1] = p0.y; xe[k][2] = p1.x;
      Xe[k][0] = p1.y; Xe[k][1] = p2.x; Xe[k][2] = p2.y;
    }
    double R[GREC];
    double detJ;
    if (mode & 8) { detJ = 1.0; for (int q = 0; q < GREC; ++q) R[q] = xe[q & 3][q % 3] + Xe[(q >> 2) & 3][q % 3]; }   // timing experiment
    else detJ = NH ? lintet_record_nh<true>(xe, Xe, gauss_w, A.lambda, A.mu, R)
                   : lintet_record_any<true>(xe, Xe, A.tab, A.model, A.lambda, A.mu, R);
    if (!(detJ > 0.0)) {                               // rare: counted by the chunk that owns its lowest-numbered node
      const int *gn = reinterpret_cast<const int *>(rec + A.lay.o_nodes);
      const int g0 = min(min(gn[nd[0]], gn[nd[1]]), min(gn[nd[2]], gn[nd[3]]));
      if (g0 >= h.r0 && g0 < h.r1) atomicAdd(A.bad, 1);
    }
    if (detJ == 0.0) {                                 // fea_solver.c:697: no gradient, no contribution
      double2 *o = reinterpret_cast<double2 *>(sT + p * GREC);
#pragma unroll
      for (int q = 0; q < GREC / 2; ++q) o[q] = make_double2(0.0, 0.0);
    } else if (!(mode & 16)) g_store_record<true>(sT + p * GREC, R);
    else sT[p * GREC] = R[0] + R[25];
  } else if (p < h.nelem) {                            // unused slot: the all-zero record empty list slots point at
    double2 *o = reinterpret_cast<double2 *>(sT + p * GREC);
#pragma unroll
    for (int q = 0; q < GREC / 2; ++q) o[q] = make_double2(0.0, 0.0);
  }
  G_LDS_DRAIN();                                       // the asm record stores
}

// rows [hp.r0, hp.r1) of a finished chunk, tile -> HBM: 16-byte LDS reads and stores by the NT threads of the caller's role
template <bool DOF, int NT>
__device__ __forceinline__ void pc_write_out(const GatherArgs &A, const PcHeader &hp, const double *sKbuf, int ftile, int c)
{
  const int odd = hp.b0 & 1;
  const double *sK = sKbuf + odd;
  double *Kd = A.K + (size_t)hp.b0 * 9;
  const int total = hp.nb * 9;
  if (odd && c == 0) Kd[0] = sK[0];
  const int npair = (total - odd) >> 1;
  int j = c;
  for (; j + 3 * NT < npair; j += 4 * NT) {                   // four LDS reads in flight, then their four stores
    const int p = odd + 2 * j;
    const double2 v0 = *reinterpret_cast<const double2 *>(sK + p), v1 = *reinterpret_cast<const double2 *>(sK + p + 2 * NT);
    const double2 v2 = *reinterpret_cast<const double2 *>(sK + p + 4 * NT), v3 = *reinterpret_cast<const double2 *>(sK + p + 6 * NT);
    *reinterpret_cast<double2 *>(Kd + p) = v0; *reinterpret_cast<double2 *>(Kd + p + 2 * NT) = v1;
    *reinterpret_cast<double2 *>(Kd + p + 4 * NT) = v2; *reinterpret_cast<double2 *>(Kd + p + 6 * NT) = v3;
  }
  for (; j < npair; j += NT) {
    const int p = odd + 2 * j;
    *reinterpret_cast<double2 *>(Kd + p) = *reinterpret_cast<const double2 *>(sK + p);
  }
  if (((total - odd) & 1) && c == 0) Kd[total - 1] = sK[total - 1];
  if (DOF && c < 3 * (hp.r1 - hp.r0)) A.f[(size_t)hp.r0 * 3 + c] = sKbuf[ftile + c];
}

// half of a block's contribution list (the 16-bit entries at bit `sh` of the list words): all reads, then all sums
template <int D>
__device__ __forceinline__ void pc_gather_half(const double *sT, const unsigned (&cw)[FEA_G_REGW], int sh, double (&acc)[9])
{
  GRead r[D];
#pragma unroll
  for (int k = 0; k < D; ++k) r[k] = g_fetch(sT, (cw[k] >> sh) & 0xFFFFu);
#pragma unroll
  for (int k = 0; k < D; ++k) g_apply(r[k], acc);
}

// Roles of the twelve waves of a workgroup (wave-uniform; waves w, w+4, w+8 share a SIMD):
//   0-3   producers: state of chunk i+1, coordinates of chunk i+2
//   4-9   gather:    two lanes per off-diagonal block of chunk i, each sums every other contribution of its list;
//                    the pair adds up by one cross-lane step, one lane stores the block, the other its mirror
//   10    diagonal:  four lanes per row: diagonal block and residual of the rows of chunk i
//   11    writer:    finished rows of chunk i-1 from its tile to HBM
#define PC_W_GATHER 4
#define PC_W_DIAG 10
#define PC_W_WRITER 11

template <bool DOF, bool NH>
__global__ __launch_bounds__(PC_THREADS, 3)
void k_assemble_gather_pc(GatherArgs A, int nruns, int mode)
{
  extern __shared__ double2 g_smem[];
  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);    // wave-uniform: scalar branches, headers stay in SGPRs
  const int role = wave < PC_W_GATHER ? 0 : wave < PC_W_DIAG ? 1 : wave == PC_W_DIAG ? 2 : 3;
  const int c = role == 0 ? t : role == 1 ? t - 64 * PC_W_GATHER : t & 63;   // index inside the role
  // LDS: two coordinate tiles | two record buffers | two K tiles (each followed by the residual rows of its chunk)
  const int ncoord = A.lay.max_nodes * 3;                     // double2 per coordinate tile
  const int nrec = A.lay.max_elems * GREC;                    // doubles per record buffer (GREC is even)
  const int ftile = (A.lay.max_tile * 9 + 3) & ~1;            // doubles of K (+ alignment slack) before the residual rows
  const int ntile = ftile + 3 * FEA_G_MAX_ROWS;
  double2 *sCb = g_smem;
  double *sRb = reinterpret_cast<double *>(g_smem + 2 * ncoord);
  double *sKb = sRb + 2 * nrec;

  // XCD-aware order (speed only): workgroups b and b+8 share an L2; every XCD gets a contiguous eighth of the runs
  const int per = ((int)gridDim.x + 7) >> 3;
  const int ridx = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (ridx >= nruns) return;
  const int first = (int)((long long)A.nchunks * ridx / nruns);
  const int cend = (int)((long long)A.nchunks * (ridx + 1) / nruns);
  const int n = cend - first;
  if (n <= 0) return;
  const size_t stride = (size_t)A.lay.stride;
  const unsigned char *maps = A.maps + (size_t)A.chunk0 * stride;
#define PC_REC(k) (maps + (size_t)((mode & 64) ? first : min(first + (k), cend - 1)) * stride)     /* (mode 64: timing experiment, every chunk reads the first one's maps) */
  const double gauss_w = A.tab->w[0];
#ifdef FEAHIP_DEBUG
  unsigned long long s0 = 0, s1 = 0, s2 = 0, busy = 0, wait = 0;
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), real0 = __builtin_amdgcn_s_memrealtime();
#endif

  // ---- what a thread carries from one chunk to the next.  HBM answers in 1-2 us under this kernel's own store
  // traffic, as long as a whole chunk takes: every load is therefore requested TWO chunks before its first use and
  // rides through one iteration in registers (measured with all arithmetic removed: a skeleton with one chunk of
  // distance ran at 0.5 ms, the latency of one dependent load per chunk).  All the loads of a role are issued
  // unconditionally from clamped, in-bounds addresses -- no branch around a load -- so that the compiler can count
  // exactly how many younger loads may still be in flight when an older one is used (s_waitcnt vmcnt(N), N > 0).
  struct PMaps { unsigned eids; int hw; };                     // producers: element word and header of a chunk
  struct PCoord { double2 a0, c0; double a2, c2; };            // producers: x and X0 of node slot c of a chunk
  struct CMaps { unsigned tpos, cw[FEA_G_REGW]; int kd, hw; }; // gather / diagonal / writer lanes
  const int blk = c >> 1, half_sh = (c & 1) * 16;             // (gather lanes) block thread of gather.cpp, which half of its list
  auto load_pmaps = [&](int k, PMaps &m) {
    const unsigned char *rec = PC_REC(k);
    m.eids = reinterpret_cast<const unsigned *>(rec + A.lay.o_elems)[min(c, A.lay.max_elems - 1)];
    m.hw = reinterpret_cast<const int *>(rec)[c & 15];
  };
  auto load_node = [&](int k) { return reinterpret_cast<const int *>(PC_REC(k) + A.lay.o_nodes)[c & (FEA_G_MAX_NODES - 1)]; };
  auto load_coord = [&](int node, PCoord &p) {
    const size_t nn = (size_t)node;
    p.a0 = *reinterpret_cast<const double2 *>(A.x + nn * 4); p.a2 = A.x[nn * 4 + 2];
    p.c0 = *reinterpret_cast<const double2 *>(A.X0 + nn * 4); p.c2 = A.X0[nn * 4 + 2];
  };
  auto store_coord = [&](const PCoord &p, double2 *tile) {
    if (c < A.lay.max_nodes) { double2 *o = tile + c * 3; o[0] = p.a0; o[1] = make_double2(p.a2, p.c0.x); o[2] = make_double2(p.c0.y, p.c2); }
  };
  const size_t o_list = role == 2 ? (size_t)(A.lay.max_ddepth > 0 ? A.lay.o_dlist : A.lay.o_bpos) : (size_t)(A.lay.max_depth > 0 ? A.lay.o_clist : A.lay.o_bpos);
  const int l_stride = role == 2 ? 64 : FEA_G_THREADS, l_last = max((role == 2 ? A.lay.max_ddepth : A.lay.max_depth) - 1, 0);
  const int l_lane = role == 2 ? c : (role == 1 ? blk : 0);
  auto load_cmaps = [&](int k, CMaps &m) {
    const unsigned char *rec = PC_REC(k);
    m.hw = reinterpret_cast<const int *>(rec)[c & 15];
    m.tpos = reinterpret_cast<const unsigned *>(rec + A.lay.o_bpos)[role == 1 ? blk : 0];
    const unsigned *lw = reinterpret_cast<const unsigned *>(rec + o_list) + l_lane;
#pragma unroll
    for (int q = 0; q < FEA_G_REGW; ++q) m.cw[q] = lw[min(q, l_last) * l_stride];
    // (the 32-bit word that holds the row's 16-bit entry: a 16-bit load is zero-extended where the value is copied,
    // i.e. used, one iteration early)
    m.kd = reinterpret_cast<const int *>(rec + A.lay.o_rows)[(G_RD + (role == 2 ? (c >> 2) : 0)) >> 1];
  };
#define PC_TOUCH_C(m) asm volatile("" : : "v"((m).tpos), "v"((m).cw[0]), "v"((m).cw[1]), "v"((m).cw[2]), "v"((m).cw[3]), "v"((m).kd), "v"((m).hw))
#define PC_TOUCH_XY(p) asm volatile("" : : "v"((p).a0.x), "v"((p).a0.y), "v"((p).a2), "v"((p).c0.x), "v"((p).c0.y), "v"((p).c2))

  // Every role runs its own loop (same trip count, one barrier per iteration each): the registers a role carries
  // from chunk to chunk are live in its own loop only, not in everybody's.  The loops are unrolled three times by
  // hand with the three register sets of a pipeline (in use / arrived / just requested) renamed instead of copied:
  // a copy of a just-requested register is a use, and the compiler waits for the load in front of it -- one chunk
  // of distance again.
#define PC_LOOP_HEAD()                                                                              \
    double *sR_cur = sRb + (i & 1) * nrec, *sR_nxt = sRb + ((i + 1) & 1) * nrec;                    \
    double *sK_cur = sKb + (i & 1) * ntile, *sK_prv = sKb + ((i + 1) & 1) * ntile;                  \
    (void)sR_cur; (void)sR_nxt; (void)sK_cur; (void)sK_prv;                                         \
    PC_STAMP(s0)
#ifdef FEAHIP_DEBUG
#define PC_LOOP_TAIL() do { PC_STAMP(s1); G_BARRIER(); PC_STAMP(s2); busy += s1 - s0; wait += s2 - s1; } while (0)
#else
#define PC_LOOP_TAIL() do { if (!(mode & 512)) G_BARRIER(); } while (0)
#endif
#define PC_UNROLL3(body, X, Y, Z)                                                                   \
    for (int i = 0;;) {                                                                             \
      body(i, X##0, X##1, X##2, Y##0, Y##1, Y##2, Z##0, Z##1, Z##2); if (++i >= n) break;           \
      body(i, X##1, X##2, X##0, Y##1, Y##2, Y##0, Z##1, Z##2, Z##0); if (++i >= n) break;           \
      body(i, X##2, X##0, X##1, Y##2, Y##0, Y##1, Z##2, Z##0, Z##1); if (++i >= n) break;           \
    }

  if (role == 0) {
    // ================= producers: state of chunk i+1, coordinates of chunk i+2 into their tile
    // in iteration i:  e?0 = element word / header of chunk i+1 (evaluated now), e?1 of chunk i+2, e?2 <- chunk i+3
    //                  c?0 = coordinates of chunk i+2 (into their tile at the end), c?1 of chunk i+3, c?2 <- chunk i+4
    //                  n?0 = node ids of chunk i+4 (their coordinates are requested now), n?1 of chunk i+5, n?2 <- chunk i+6
    PMaps ef, e0, e1, e2;
    PCoord cf, cg, c0, c1, c2;
    load_pmaps(0, ef); load_pmaps(1, e0); load_pmaps(2, e1);
    const int nd0 = load_node(0), nd1 = load_node(1), nd2 = load_node(2), nd3 = load_node(3);
    int n0 = load_node(4), n1 = load_node(5), n2 = 0;
    load_coord(nd0, cf); load_coord(nd1, cg); load_coord(nd2, c0); load_coord(nd3, c1);
    store_coord(cf, sCb); store_coord(cg, sCb + ncoord);      // (prologue: nothing to hide behind, waited for in place)
    G_BARRIER();
    pc_state<NH>(A, PC_REC(0), ef.eids, pc_decode(ef.hw), sCb, sRb, c, gauss_w);
    PC_TOUCH_XY(c0); PC_TOUCH_XY(c1);
    asm volatile("" : : "v"(e0.eids), "v"(e0.hw), "v"(e1.eids), "v"(e1.hw), "v"(n0), "v"(n1));   // nothing pending when the loop is entered
    PcHeader h = pc_decode(e0.hw);                            // chunk 1
    e2 = e1; c2 = c1;                                         // (defined values; overwritten by the first requests)
    G_BARRIER();
#define PC_PRODUCER_BODY(i, eU, eM, eL, cW, cM, cL, nU, nM, nL)                                                        \
    {                                                                                                                  \
      PC_LOOP_HEAD();                                                                                                  \
      if (!(mode & 128)) { load_coord(nU, cL); load_pmaps(i + 3, eL); nL = load_node(i + 6); }  /* coordinates of chunk i+4, ... */ \
      /* state of chunk i+1 (h is its header), coordinates from tile (i+1) mod 2 */                                    \
      if (i + 1 < n) pc_state<NH>(A, PC_REC(i + 1), eU.eids, h, sCb + ((i + 1) & 1) * ncoord, sR_nxt, c, gauss_w, mode); \
      /* coordinates of chunk i+2 (requested two iterations ago) into the tile chunk i used: its readers finished      \
         before the last barrier */                                                                                    \
      PC_TOUCH_XY(cW);                                                                                                 \
      store_coord(cW, sCb + (i & 1) * ncoord);                                                                         \
      asm volatile("" : : "v"(eM.eids), "v"(eM.hw), "v"(nM));                                                          \
      h = pc_decode(eM.hw);                                                                                            \
      PC_LOOP_TAIL();                                                                                                  \
    }
    PC_UNROLL3(PC_PRODUCER_BODY, e, c, n)
#undef PC_PRODUCER_BODY
  } else if (role == 3) {
    // ================= writer: finished rows of chunk i-1 from their tile to HBM
    CMaps m0, m1, m2;
    load_cmaps(0, m0); load_cmaps(1, m1);
    G_BARRIER();
    PC_TOUCH_C(m0); PC_TOUCH_C(m1);
    m2 = m1;
    PcHeader h = pc_decode(m0.hw), hp = h;                    // h: the chunk the gather waves work on, hp: the one before it
    int u0 = 0, u1 = 0, u2 = 0, v0 = 0, v1 = 0, v2 = 0;       // (unused pipelines of the unroll macro)
    G_BARRIER();
#define PC_WRITER_BODY(i, mA, mB, mC, ua, ub, uc, va, vb, vc)                                                          \
    {                                                                                                                  \
      PC_LOOP_HEAD();                                                                                                  \
      /* stores first: the loads below stay the youngest memory operations */                                          \
      if (i > 0 && !(mode & 32)) pc_write_out<DOF, 64>(A, hp, sK_prv, ftile, c);                                       \
      if (!(mode & 128)) load_cmaps(i + 2, mC);                                                                       \
      hp = h;                                                                                                          \
      PC_TOUCH_C(mB);                                                                                                  \
      h = pc_decode(mB.hw);                                                                                            \
      PC_LOOP_TAIL();                                                                                                  \
    }
    PC_UNROLL3(PC_WRITER_BODY, m, u, v)
#undef PC_WRITER_BODY
    (void)u0; (void)u1; (void)u2; (void)v0; (void)v1; (void)v2;
    pc_write_out<DOF, 64>(A, hp, sKb + ((n - 1) & 1) * ntile, ftile, c);       // rows of the last chunk
  } else if (role == 1) {
    // ================= gather lanes: two per off-diagonal block of chunk i
    CMaps m0, m1, m2;                                         // in iteration i: chunk i (in use), chunk i+1, <- chunk i+2
    load_cmaps(0, m0); load_cmaps(1, m1);
    G_BARRIER();
    PC_TOUCH_C(m0); PC_TOUCH_C(m1);
    m2 = m1;
    PcHeader h = pc_decode(m0.hw);
    int u0 = 0, u1 = 0, u2 = 0, v0 = 0, v1 = 0, v2 = 0;
    G_BARRIER();
#define PC_GATHER_BODY(i, mA, mB, mC, ua, ub, uc, va, vb, vc)                                                          \
    {                                                                                                                  \
      PC_LOOP_HEAD();                                                                                                  \
      if (!(mode & 128)) load_cmaps(i + 2, mC);                                                                       \
      double *sK = sK_cur + (h.b0 & 1);                       /* LDS and HBM agree on 16-byte alignment in the write-out */ \
      /* block sums of chunk i out of its records: this lane's half of the list, the pair's sum by one cross-lane step */ \
      double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};                                                                     \
      if (blk < h.noffd && !(mode & 2)) {                                                                              \
        switch (min(h.depth, FEA_G_REGW)) {                    /* all reads of the registers' list words in flight before the first sum */ \
        case 1: pc_gather_half<1>(sR_cur, mA.cw, half_sh, acc); break;                                                 \
        case 2: pc_gather_half<2>(sR_cur, mA.cw, half_sh, acc); break;                                                 \
        case 3: pc_gather_half<3>(sR_cur, mA.cw, half_sh, acc); break;                                                 \
        case 4: pc_gather_half<4>(sR_cur, mA.cw, half_sh, acc); break;                                                 \
        default: break;                                                                                                \
        }                                                                                                              \
        for (int k = FEA_G_REGW; k < h.depth; ++k) {          /* blocks with more than 8 contributions (unstructured meshes) */ \
          const unsigned w = reinterpret_cast<const unsigned *>(PC_REC(i) + A.lay.o_clist)[k * FEA_G_THREADS + blk];   \
          g_consume(sR_cur, (w >> half_sh) & 0xFFFFu, acc);                                                            \
        }                                                                                                              \
      }                                                                                                                \
      _Pragma("unroll")                                                                                                \
      for (int q = 0; q < 9; ++q) if (!(mode & 256)) acc[q] += __shfl_xor(acc[q], 1);     /* a + b = b + a to the bit */ \
      /* block (even lane) and mirror block (odd lane) into tile i mod 2 (its last readers, the row stores of          \
         chunk i-2, finished before the last barrier) */                                                               \
      if (blk < h.noffd && !(mode & 4)) {                                                                              \
        const int bpos = (int)(mA.tpos & 0xFFFFu), mpos = (int)(mA.tpos >> 16);                                        \
        if (half_sh == 0) {                                                                                            \
          _Pragma("unroll")                                                                                            \
          for (int q = 0; q < 9; ++q) sK[bpos * 9 + q] = acc[q];                                                       \
        } else if (mpos != 0xFFFF) {                          /* K_ba = K_ab' (fea_solver.c:1249 relies on the same symmetry) */ \
          _Pragma("unroll")                                                                                            \
          for (int a = 0; a < 3; ++a)                                                                                  \
            _Pragma("unroll")                                                                                          \
            for (int b = 0; b < 3; ++b) sK[mpos * 9 + 3 * b + a] = acc[3 * a + b];                                     \
        }                                                                                                              \
      }                                                                                                                \
      PC_TOUCH_C(mB);                                         /* requested an iteration ago; this iteration's request stays in flight */ \
      h = pc_decode(mB.hw);                                                                                            \
      PC_LOOP_TAIL();                                                                                                  \
    }
    PC_UNROLL3(PC_GATHER_BODY, m, u, v)
#undef PC_GATHER_BODY
    (void)u0; (void)u1; (void)u2; (void)v0; (void)v1; (void)v2;
  } else {
    // ================= diagonal wave: four lanes per row, diagonal block and residual of the rows of chunk i
    CMaps m0, m1, m2;
    load_cmaps(0, m0); load_cmaps(1, m1);
    G_BARRIER();
    PC_TOUCH_C(m0); PC_TOUCH_C(m1);
    m2 = m1;
    PcHeader h = pc_decode(m0.hw);
    int u0 = 0, u1 = 0, u2 = 0, v0 = 0, v1 = 0, v2 = 0;
    G_BARRIER();
#define PC_DIAG_BODY(i, mA, mB, mC, ua, ub, uc, va, vb, vc)                                                            \
    {                                                                                                                  \
      PC_LOOP_HEAD();                                                                                                  \
      if (!(mode & 128)) load_cmaps(i + 2, mC);                                                                       \
      const int nrows = h.r1 - h.r0;                                                                                   \
      double *sK = sK_cur + (h.b0 & 1);                                                                                \
      double dg[6] = {0, 0, 0, 0, 0, 0};                                                                               \
      double fa[3] = {0, 0, 0};                                                                                        \
      if (c < 4 * nrows && !(mode & 2)) {                                                                              \
        switch (min(h.ddepth, FEA_G_REGW)) {                  /* four visits (sixteen reads) in flight at a time */    \
        case 1: g_diag_batch<0, 1, DOF>(sR_cur, mA.cw, dg, fa); break;                                                 \
        case 2: g_diag_batch<0, 2, DOF>(sR_cur, mA.cw, dg, fa); break;                                                 \
        case 3: g_diag_batch<0, 2, DOF>(sR_cur, mA.cw, dg, fa); g_diag_batch<2, 1, DOF>(sR_cur, mA.cw, dg, fa); break; \
        case 4: g_diag_batch<0, 2, DOF>(sR_cur, mA.cw, dg, fa); g_diag_batch<2, 2, DOF>(sR_cur, mA.cw, dg, fa); break; \
        default: break;                                                                                                \
        }                                                                                                              \
        for (int k = FEA_G_REGW; k < h.ddepth; ++k) {         /* nodes with more than 32 elements around them */       \
          const unsigned w = reinterpret_cast<const unsigned *>(PC_REC(i) + A.lay.o_dlist)[k * 64 + c];                \
          g_consume_diag<DOF>(sR_cur, w & 0xFFFFu, dg, fa); g_consume_diag<DOF>(sR_cur, w >> 16, dg, fa);              \
        }                                                                                                              \
      }                                                                                                                \
      /* the four partial sums of a row meet in its first lane */                                                      \
      _Pragma("unroll")                                                                                                \
      for (int q = 0; q < 6; ++q) if (!(mode & 256)) { dg[q] += __shfl_xor(dg[q], 1); dg[q] += __shfl_xor(dg[q], 2); } \
      if (DOF) {                                                                                                       \
        _Pragma("unroll")                                                                                              \
        for (int q = 0; q < 3; ++q) { fa[q] += __shfl_xor(fa[q], 1); fa[q] += __shfl_xor(fa[q], 2); }                  \
      }                                                                                                                \
      if ((c & 3) == 0 && c < 4 * nrows) {                                                                             \
        double *o = sK + ((mA.kd >> (((c >> 2) & 1) * 16)) & 0xFFFF) * 9;      /* G_RD is even: the row's parity picks the half */ \
        o[0] = dg[0]; o[1] = dg[1]; o[2] = dg[2]; o[3] = dg[1]; o[4] = dg[3]; o[5] = dg[4]; o[6] = dg[2]; o[7] = dg[4]; o[8] = dg[5]; \
        if (DOF) { double *fo = sK_cur + ftile + 3 * (c >> 2); fo[0] = fa[0]; fo[1] = fa[1]; fo[2] = fa[2]; }         \
      }                                                                                                                \
      PC_TOUCH_C(mB);                                                                                                  \
      h = pc_decode(mB.hw);                                                                                            \
      PC_LOOP_TAIL();                                                                                                  \
    }
    PC_UNROLL3(PC_DIAG_BODY, m, u, v)
#undef PC_DIAG_BODY
    (void)u0; (void)u1; (void)u2; (void)v0; (void)v1; (void)v2;
  }
#undef PC_UNROLL3
#undef PC_LOOP_HEAD
#undef PC_LOOP_TAIL
#ifdef FEAHIP_DEBUG
  if (A.stamps && (t & 63) == 0) {                            // one line per wave: [run][wave][4]
    unsigned long long *o = A.stamps + ((size_t)ridx * 12 + (t >> 6)) * 4;
    o[0] = busy; o[1] = wait;
    o[2] = __builtin_amdgcn_s_memtime() - clk0; o[3] = __builtin_amdgcn_s_memrealtime() - real0;
  }
#endif
#undef PC_REC
}

// LDS one workgroup needs: two coordinate tiles, two record buffers, two K tiles with their residual rows
static size_t pc_lds_bytes(const GatherLayout &lay)
{
  const size_t ncoord = (size_t)lay.max_nodes * 3 * 16;
  const size_t nrec = (size_t)lay.max_elems * GREC * 8;
  const size_t ntile = ((size_t)((lay.max_tile * 9 + 3) & ~1) + 3 * FEA_G_MAX_ROWS) * 8;
  return 2 * (ncoord + nrec + ntile);
}

bool gather_pc_fits(const feahip_ctx *c)
{
  return c->have_gather && c->gather_lay && pc_lds_bytes(*c->gather_lay) <= 160u * 1024u;
}

int launch_assemble_gather_pc(feahip_ctx *c, bool doF)
{
  GatherArgs A;
  A.chunk0 = 0; A.nchunks = c->ngchunks; A.model = c->model; A.lambda = c->lambda; A.mu = c->mu;
  A.tab = c->d_table; A.maps = c->d_gmaps; A.lay = *c->gather_lay; A.X0 = c->d_X0; A.x = c->d_x;
  A.K = c->d_K; A.f = c->d_f; A.bad = c->d_flag + 1; A.stamps = nullptr; A.ablate = 0;
  if (c->ngchunks <= 0) return FEAHIP_OK;
  // runs: equal shares of the chunks, a multiple of the CU count so that every round of resident workgroups (one per
  // CU) ends together (FEAHIP_GATHER_PC_RUNS: tuning only, results unchanged)
  static int runs_per_cu = -1;
  if (runs_per_cu < 0) { const char *e = getenv("FEAHIP_GATHER_PC_RUNS"); runs_per_cu = e && atoi(e) > 0 ? atoi(e) : 2; }
  static int ncu_of[64];                                      // compute units per device, asked once
  int &ncu = ncu_of[c->device & 63];
  if (ncu <= 0) { hipDeviceProp_t p; ncu = (hipGetDeviceProperties(&p, c->device) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256; }
  static int mode = -1;                                       // timing experiments (FEAHIP_GATHER_PC_MODE; non-zero: results meaningless): 2 no gather, 4 no tile writes, 8 no state arithmetic, 16 no record stores, 32 no row stores
  if (mode < 0) { const char *e = getenv("FEAHIP_GATHER_PC_MODE"); mode = e ? atoi(e) : 0; }
  int nruns = ncu * runs_per_cu;
  if (nruns > c->ngchunks) nruns = c->ngchunks;
  const dim3 grid((nruns + 7) & ~7), blk(PC_THREADS);
  const int lds = (int)pc_lds_bytes(A.lay);
#ifdef FEAHIP_DEBUG
  static unsigned long long *d_stamps = nullptr;
  static int stamps_cap = 0;
  const char *dbg = getenv("FEAHIP_GATHER_STAMPS");
  if (dbg && atoi(dbg)) {
    if (!d_stamps || stamps_cap < nruns) {
      if (d_stamps) (void)hipFree(d_stamps);
      (void)hipMalloc((void **)&d_stamps, sizeof(unsigned long long) * 48 * (size_t)nruns);
      (void)hipMemset(d_stamps, 0, sizeof(unsigned long long) * 48 * (size_t)nruns);
      stamps_cap = nruns;
    }
    A.stamps = d_stamps;
  }
#endif
  const bool nh = c->model == FEAHIP_MODEL_COMPRESSIBLE_NEOHOOKEAN;
#define PC_LAUNCH(F, M)                                                                                              \
  do {                                                                                                               \
    FEA_HIP_CHECK(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_assemble_gather_pc<F, M>),                \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds));                         \
    hipLaunchKernelGGL((k_assemble_gather_pc<F, M>), grid, blk, lds, c->stream, A, nruns, mode);                     \
  } while (0)
  if (doF) { if (nh) PC_LAUNCH(true, true); else PC_LAUNCH(true, false); }
  else     { if (nh) PC_LAUNCH(false, true); else PC_LAUNCH(false, false); }
#undef PC_LAUNCH
  FEA_HIP_CHECK(c, hipGetLastError());
#ifdef FEAHIP_DEBUG
  if (A.stamps) {
    static int calls = 0;
    if (++calls == 50) {
      (void)hipStreamSynchronize(c->stream);
      std::vector<unsigned long long> hst((size_t)nruns * 48);
      (void)hipMemcpy(hst.data(), A.stamps, hst.size() * 8, hipMemcpyDeviceToHost);
      double sum[12][4] = {};
      for (int i = 0; i < nruns; ++i)
        for (int w = 0; w < 12; ++w)
          for (int q = 0; q < 4; ++q) sum[w][q] += (double)hst[((size_t)i * 12 + w) * 4 + q];
      fprintf(stderr, "[gather pc stamps] in-kernel clock %.0f MHz, %d runs of %.1f chunks, %.0f shader cycles per chunk\n",
              sum[0][3] > 0 ? 100.0 * sum[0][2] / sum[0][3] : 0.0, nruns, (double)c->ngchunks / nruns, sum[0][2] / c->ngchunks);
      for (int w = 0; w < 12; ++w)
        fprintf(stderr, "[gather pc stamps F=%d wave %d (%s), per chunk] busy %.0f  barrier wait %.0f cycles\n", (int)doF, w,
                w < 4 ? "producer" : w < 10 ? "gather" : w == 10 ? "diagonal" : "writer", sum[w][0] / c->ngchunks, sum[w][1] / c->ngchunks);
    }
  }
#endif
  return FEAHIP_OK;
}
