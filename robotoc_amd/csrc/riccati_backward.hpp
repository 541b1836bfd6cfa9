// riccati_backward.hpp -- batched backward Riccati recursion for gfx950 (CDNA4).
//
// Replaces RiccatiRecursion::backwardRiccatiRecursion (reference
// src/riccati/riccati_recursion.cpp:32-80) and everything it calls
// (src/riccati/riccati_factorizer.cpp:44-197,
//  src/riccati/backward_riccati_recursion_factorizer.cpp:31-174).
//
// Mapping: one workgroup of NW wavefronts per OCP instance, serial over the
// stages (the recursion is a chain), the next-stage factorisation P+ / s+ stays
// resident in LDS for the whole sweep.  Per stage (A = Fxx, Bv = Fvu):
//
//   z    = s+ - P+ Fx                                   VALU mat-vec
//   PB   = P+[:,v] Bv                                   MFMA  (NX x NU, K = NV)
//   G    = Quu + Bv^T PB[v,:]                           MFMA  (NU x NU, K = NV)
//   lu   = lu - Bv^T z[v]
//   PAa  = [P+ ; PB^T] A                                MFMA  ((NX+NU) x NX, K = NX)
//          rows <  NX : (A^T P+)^T, kept in REGISTERS: the f64 MFMA result layout
//                       (row = q+4r, col = lane&15) is exactly the A-operand layout of the
//                       next product, so AtP never touches LDS ("chained MFMA");
//          rows >= NX : (A^T PB)^T = (H - Qxu)^T  -> H
//   F    = Qxx + AtP A                                  MFMA, chained from the PAa registers
//   LLT(G) (wave shuffles), K = -G^-1 H^T, k = -G^-1 lu   VALU, overlaps the F product
//   GK   = G K ;  F -= K^T GK                           MFMA
//   P    = (F + F^T)/2 ;  s = A^T z - lx - H k
//
// which is the algebra of the reference (brrf.cpp:31-45,78-91) re-associated so that
// the only dense NX^3 products are two MFMA GEMMs: H = Qxu + A^T (P+[:,v] Bv) instead
// of (A^T P+)[:,v] Bv, and A^T P+ Fx = A^T (P+ Fx).  Results agree with the reference
// order to fp64 round-off (tests/test_gpu_parity.py states the tolerance).
//
// Switching-constraint (Schur complement, riccati_factorizer.cpp:58-89) and
// switching-time (STO, :93-175) terms are computed with VALU loops on LDS data:
// they occur on O(#events) stages per horizon.
#pragma once
#include "device_utils.hpp"
#include "../../include/rtoc.h"
#include "riccati_scan_sto.hpp"   // sto_prep_body: rides in the one-stage launch of riccati_backward_kernel

namespace rtoc {

// Phase time-stamps (s_memtime) of block 0, written only when a profiling buffer is attached.
// s_memtime stamps for tools/phase_profile*.py: compiled in only with -DRTOC_ENABLE_PROF (make PROF=1);
// thirty scalar tests per stage and two live SGPR pairs are not free in the production build
#ifdef RTOC_ENABLE_PROF
#define RTOC_PROF(k)                                                             \
  do {                                                                          \
    if (a.prof && b == 0 && tid0 == 0) a.prof[st * 32 + (k)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
// same, stamped by the first lane of the second wave (the vector wave of the role-split kernel)
#define RTOC_PROFV(k)                                                            \
  do {                                                                          \
    if (a.prof && b == 0 && tid0 == 64) a.prof[st * 32 + (k)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#else
#define RTOC_PROF(k) do { } while (0)
#define RTOC_PROFV(k) do { } while (0)
#endif

struct BwdArgs {
  const double* kkt;       // [batch][nstages][kkt stride]
  double* kkt_rw;          // same buffer, writable (writeback of F,H,G,lu)
  double* ric;             // [batch][nstages][ric stride]
  const rtoc_grid* grid;   // [nstages] (device)
  uint32_t* status;        // [batch]
  long long* prof;         // optional [nstages][16] cycle stamps of block 0 (tuning aid), or nullptr
  int nstages;
  int batch;  // instances [first, batch) are processed by this launch
  int first;
  int writeback;
  double max_dts0;
  // horizon-scan mode of riccati_backward_kernel (riccati_scan.hpp): value records [P | s] of every
  // grid point, [batch][nstages][scan_ps_stride]; workgroup (b, st) then does the ONE stage st from
  // P_{st+1}, s_{st+1} read there.  nullptr = serial recursion.
  const double* scan_ps;
  int scan_ps_stride;
  int scan_ps_soff;  // offset of s inside a value record
  // Scan on a grid with switching-time optimisation (riccati_scan_sto.hpp): workgroups (b, nstages + st) of the same launch prepare
  // the bundle of grid point st for the serial vector pass -- they read what the policy workgroups read (the scan's value records,
  // the KKT records), none of their output, so they need neither a launch nor an event of their own.  nullptr: no such workgroups.
  double* sto_scr;   // [batch][nstages][scan::StoScratch::STRIDE]
  // Segment of the horizon: the register kernels (riccati_backward_rv.hpp, _rw.hpp) walk the grid points seg_hi .. seg_lo and take
  // P+ / s+ of grid point seg_hi + 1 from the Riccati records unless that is the terminal one -- the quadruped kernel is launched
  // once per horizon (seg_hi = N - 1, seg_lo = 0), the iCub one between its switching-constraint grid points; in the one-stage
  // mode above the tile-split kernel does grid point blockIdx.y + seg_lo.  Zero in every other launch.
  int seg_hi, seg_lo;
  // Structured-Fxx forms on records the runtime cannot vouch for (a bound buffer the caller may have rewritten since the last
  // device check, RTOC_OPT_FXX_STRUCTURE = 0): the kernel verifies the rows it does NOT multiply -- it has them in LDS anyway --
  // and raises RTOC_STAT_FXX_UNSTRUCTURED on the instance instead of returning a silently wrong factorisation.
  int check_fxx;
};

template <int NV, int NU, int NS, int NW>
struct BwdCfg {
  static constexpr int NX = 2 * NV;
  static constexpr int NT = 64 * NW;
  static constexpr int LDP = lds_ld(NX);
  static constexpr int TNX = (NX + 15) / 16;
  static constexpr int MA = NX + NU;
  static constexpr int TMA = (MA + 15) / 16;
  static constexpr int TNU = (NU + 15) / 16;
  static constexpr int CNT = (TNX + NW - 1) / NW;  // owned 16-tiles of the state dim per wave
  static constexpr int NSP = NS > 0 ? NS : 1;
  static constexpr int pad8(int n) { return (n + 7) & ~7; }
  // ---- LDS carve (doubles) ----
  static constexpr int OFF_P = 0;
  static constexpr int OFF_A = OFF_P + NX * LDP;
  // the switching-constraint scratch (S_* below) is aliased onto A after the F product: for small robots with contacts
  // (iiwa14 + 1 point contact: nx = 14, ns = 3) it is the larger of the two
  static constexpr int SC_DOUBLES = NS > 0 ? (pad8(NSP * NX) + 3 * pad8(NSP * NU) + pad8(NU * NU) + pad8(NSP * NSP) + 6 * pad8(NSP)) : 0;
  static constexpr int A_DOUBLES = (NX * LDP > SC_DOUBLES) ? NX * LDP : ((SC_DOUBLES + 1) & ~1);
  static constexpr int OFF_PB = OFF_A + A_DOUBLES;   // PB = P+[:,v] Bv; dead after PAa -> reused for K^T
  static constexpr int OFF_KT = OFF_PB;
  static constexpr int OFF_H = OFF_PB + NU * LDP;   // H = Qxu'; dead after the K solve -> reused for GK
  static constexpr int OFF_GK = OFF_H;
  static constexpr int OFF_BV = OFF_H + NU * LDP;
  // leading dimension of Bv in LDS: the MFMA operand reads walk it as q + li * LDB (16 rows x 4 k of a tile); with
  // NV = 32 a stride of NV doubles puts all 16 rows of a half-wave in ONE bank pair (53 % of the LDS cycles of the
  // nv = 32 kernel were bank conflicts, profiles/r02_rocprof_summary.txt); even and LDB/2 odd spreads them
  static constexpr int LDB = (NV % 2 == 0) ? lds_ld(NV) : NV;
  static constexpr int OFF_G = OFF_BV + pad8(LDB * NU);
  static constexpr int OFF_L = OFF_G + pad8(NU * NU);
  static constexpr int OFF_VEC = OFF_L + pad8(NU * NU);
  // ---- switching-constraint scratch, aliased on A after the F product ----
  static constexpr int S_M = OFF_A;                       // NS x NX (ld NS)
  static constexpr int S_PHIU = S_M + pad8(NSP * NX);     // NS x NU (ld NS)
  static constexpr int S_DGINV = S_PHIU + pad8(NSP * NU); // NS x NU
  static constexpr int S_SDG = S_DGINV + pad8(NSP * NU);  // SinvDGinv NS x NU
  static constexpr int S_GINV = S_SDG + pad8(NSP * NU);   // NU x NU
  static constexpr int S_LS = S_GINV + pad8(NU * NU);     // NS x NS: S, factorised in place
  static constexpr int S_PHIT = S_LS + pad8(NSP * NSP);
  static constexpr int S_PRES = S_PHIT + pad8(NSP);
  static constexpr int S_MV = S_PRES + pad8(NSP);
  static constexpr int S_MT = S_MV + pad8(NSP);
  static constexpr int S_MTN = S_MT + pad8(NSP);
  static constexpr int S_LSINV = S_MTN + pad8(NSP);
  static constexpr int S_END = S_LSINV + pad8(NSP);
  static constexpr int VX = pad8(NX), VU = pad8(NU);
  // vectors
  static constexpr int V_SN = OFF_VEC;          // s+
  static constexpr int V_FX = V_SN + VX;        // Fx
  static constexpr int V_LX = V_FX + VX;        // lx
  static constexpr int V_Z = V_LX + VX;         // z = s+ - P+ Fx
  static constexpr int V_SNEW = V_Z + VX;       // new s
  static constexpr int V_PSIN = V_SNEW + VX;    // Psi+
  static constexpr int V_PHIN = V_PSIN + VX;    // Phi+
  static constexpr int V_FFX = V_PHIN + VX;     // fx
  static constexpr int V_HX = V_FFX + VX;       // hx
  static constexpr int V_Y = V_HX + VX;         // y = P+ fx + Psi+
  static constexpr int V_PSIX = V_Y + VX;       // psi_x -> Psi
  static constexpr int V_PHIX = V_PSIX + VX;    // phi_x -> Phi
  static constexpr int V_PSI = V_PHIX + VX;     // Psi (new)
  static constexpr int V_PHI = V_PSI + VX;      // Phi (new)
  static constexpr int V_LU = V_PHI + VX;       // lu'
  static constexpr int V_KV = V_LU + VU;        // k
  static constexpr int V_HU = V_KV + VU;
  static constexpr int V_PSIU = V_HU + VU;
  static constexpr int V_PHIU = V_PSIU + VU;
  static constexpr int V_TV = V_PHIU + VU;
  static constexpr int V_WV = V_TV + VU;
  static constexpr int V_LINV = V_WV + VU;      // 1/diag(L)
  static constexpr int V_SCN = V_LINV + VU;     // next scalars [xi,chi,rho,eta,iota]
  static constexpr int V_SC = V_SCN + 8;        // new scalars + policy
  static constexpr int V_KSC = V_SC + 8;        // kkt scalars [Qtt,Qtt_prev,h]
  static constexpr int V_FLAG = V_KSC + 8;      // status accumulation (as double bits)
  // role-split kernel: Bv^T s+_v, Bv^T Psi+_v, Bv^T Phi+_v -- free-rider columns of the G product (matrix wave)
  static constexpr int V_BTS = V_FLAG + 8;
  static constexpr int V_BTPSI = V_BTS + VU;
  static constexpr int V_BTPHI = V_BTPSI + VU;
  // Z^T = L^-1 H^T (nu x nx, ld NU) of the tile-split shapes (nx > 63: one instance per CU, LDS to spare): F -= Z Z^T instead of
  // G K and F -= K^T G K on grid points without a switching constraint
  static constexpr int OFF_ZT = V_BTPHI + VU;
  static constexpr int LDS_DOUBLES = OFF_ZT + (NX > 63 ? pad8(NU * NX) : 0);
  static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
  static_assert(NS == 0 || S_END <= OFF_PB, "switching-constraint scratch must fit in A");
  static_assert(NX + 1 <= NT, "need one thread per state entry plus one");
};

// ---- register prefetch of the next stage's record (HBM latency hidden behind the current stage) ----
template <int NT, int N2>
struct PreCnt { static constexpr int value = (N2 + NT - 1) / NT; };
template <int CNT>
struct PreBuf { d2 v[CNT]; };

template <int NT, int N2>
__device__ __forceinline__ void pre_load(PreBuf<PreCnt<NT, N2>::value>& buf, const double* __restrict__ src, int tid) {
  d2* v = buf.v;
  const d2* s2 = reinterpret_cast<const d2*>(src);
#pragma unroll
  for (int k = 0; k < PreCnt<NT, N2>::value; ++k) {
    const int e = tid + k * NT;
    v[k] = s2[e < N2 ? e : N2 - 1];   // unconditional (clamped): a branch around a load makes the compiler wait for ALL outstanding
  }                                   // loads at the next use of any loaded register (pre_store never stores the clamped ones)
}

template <int NT, int ROWS, int COLS>
__device__ __forceinline__ void pre_load_mat(PreBuf<MatMap<NT, ROWS>::passes(COLS)>& buf,
                                             const double* __restrict__ src, int tid) {
  using M = MatMap<NT, ROWS>;
  const int r2 = tid % M::CPL, c0 = tid / M::CPL;
  const d2* s2 = reinterpret_cast<const d2*>(src);
  const bool act = c0 < M::CPP;
#pragma unroll
  for (int k = 0; k < M::passes(COLS); ++k) {   // unconditional, clamped to the first entry (see pre_load)
    const bool ok = act && (k * M::CPP + c0 < COLS);
    buf.v[k] = s2[ok ? r2 + c0 * M::CPL + k * M::CPP * M::CPL : 0];
  }
}

template <int NT, int ROWS, int COLS, int LD>
__device__ __forceinline__ void pre_store_mat(double* __restrict__ dst,
                                              const PreBuf<MatMap<NT, ROWS>::passes(COLS)>& buf, int tid) {
  using M = MatMap<NT, ROWS>;
  static_assert(LD % 2 == 0, "even ld");
  const int r2 = tid % M::CPL, c0 = tid / M::CPL;
  double* d = dst + 2 * r2 + c0 * LD;
  const bool act = c0 < M::CPP;
#pragma unroll
  for (int k = 0; k < M::passes(COLS); ++k)
    if (act && (k * M::CPP + c0 < COLS)) *reinterpret_cast<d2*>(d + k * M::CPP * LD) = buf.v[k];
}

// flat prefetch registers of a ROWS x cols column-major block -> LDS with leading dimension LD (ROWS even: a d2
// never straddles two columns)
template <int NT, int N2, int ROWS, int LD>
__device__ __forceinline__ void pre_store_ld(double* __restrict__ dst, const PreBuf<PreCnt<NT, N2>::value>& buf, int tid) {
  if constexpr (ROWS == LD) {
#pragma unroll
    for (int k = 0; k < PreCnt<NT, N2>::value; ++k) {
      const int e = tid + k * NT;
      if (e < N2) reinterpret_cast<d2*>(dst)[e] = buf.v[k];
    }
  } else {
    static_assert(ROWS % 2 == 0 && LD % 2 == 0, "pairs must not straddle columns");
#pragma unroll
    for (int k = 0; k < PreCnt<NT, N2>::value; ++k) {
      const int e = tid + k * NT;
      if (e < N2) *reinterpret_cast<d2*>(dst + (2 * e) % ROWS + ((2 * e) / ROWS) * LD) = buf.v[k];
    }
  }
}

template <int NT, int N2>
__device__ __forceinline__ void pre_store_flat(double* __restrict__ dst, const PreBuf<PreCnt<NT, N2>::value>& buf, int tid) {
  const d2* v = buf.v;
#pragma unroll
  for (int k = 0; k < PreCnt<NT, N2>::value; ++k) {
    const int e = tid + k * NT;
    if (e < N2) reinterpret_cast<d2*>(dst)[e] = v[k];
  }
}

// In-wave Cholesky of an n x n SPD matrix held in LDS (column-major, ld = LD).
// Lane i owns row i in registers; pivots / columns travel by wave shuffles.
// Writes the lower factor back to Ldst (ld LD) and 1/diag to linv.  Returns true on failure.
template <int NMAX, int LD>
__device__ __forceinline__ bool wave_llt(const double* __restrict__ A, double* __restrict__ Ldst,
                                         double* __restrict__ linv, int n, int lane) {
  double g[NMAX];
  const int li = lane < n ? lane : 0;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) g[k] = (k < n) ? A[li + k * LD] : 0.0;
  bool bad = false;
#pragma unroll
  for (int j = 0; j < NMAX; ++j) {
    if (j < n) {
      // j, k are compile-time constants (fully unrolled): pivots and column entries travel through
      // the scalar unit (readlane), not the LDS crossbar
      const double d = readlane_d(g[j], j);
      if (!(d > 0.0)) bad = true;
      const double inv = rsqrt_d(d);
      const double lij = g[j] * inv;  // lane j: d / sqrt(d) = sqrt(d)
      g[j] = lij;
      if (lane == j) linv[j] = inv;
#pragma unroll
      for (int k = j + 1; k < NMAX; ++k) {
        if (k < n) {
          const double lkj = readlane_d(lij, k);
          g[k] -= lij * lkj;
        }
      }
    }
  }
  if (lane < n) {
#pragma unroll
    for (int k = 0; k < NMAX; ++k)
      if (k < n) Ldst[lane + k * LD] = (k <= lane) ? g[k] : 0.0;
  }
  return bad;
}

// In-wave Cholesky as wave_llt, and in the same sweep Y = L^-1 by forward substitution on the
// identity.  Lanes 0..15 carry the rows of G / L, lanes 16..31 the columns of Y (lane 16+j = column
// j): the column-j step of both is "scale entry j by 1/l_jj, subtract (entry j) x L[k][j] from
// entry k", with the same broadcast scalars L[k][j], so ONE instruction stream serves both -- the
// inverse factor costs no instructions beyond the Cholesky's own.
// Writes L / 1/diag like wave_llt and Y column-major (ld NMAX) to Ydst.  Returns true on failure.
// The same in three parts, so that a wave can run the column steps in the gaps its other work leaves (the tile-split kernel gives
// the first columns to a wave that would otherwise wait at the barrier behind P+ A): the state -- row lane's entries of G / L, or
// one column of Y -- stays in registers between the calls.
template <int NMAX>
struct LltInvState {
  double g[NMAX];
  bool bad;
};
template <int NMAX, int LD, int YOFF>
__device__ __forceinline__ void llt_inv_load(LltInvState<NMAX>& s, const double* __restrict__ A, int n, int lane) {
  // rows in lanes [0, YOFF), columns of Y in lanes [YOFF, 2*YOFF): YOFF = 16 (n <= 16) or 32 (n <= 32)
  static_assert(NMAX <= YOFF && 2 * YOFF <= 64, "rows of G and columns of Y share one wave");
  const int li = lane < n ? lane : 0;
  const bool ylane = lane >= YOFF;
#pragma unroll
  for (int k = 0; k < NMAX; ++k) {
    const double a = (k < n) ? A[li + k * LD] : 0.0;
    s.g[k] = ylane ? ((k == lane - YOFF) ? 1.0 : 0.0) : a;
  }
  s.bad = false;
}
template <int NMAX, int J0, int J1>
__device__ __forceinline__ void llt_inv_steps(LltInvState<NMAX>& s, double* __restrict__ linv, int n, int lane) {
#pragma unroll
  for (int j = J0; j < J1; ++j) {
    if (j < n) {
      const double d = readlane_d(s.g[j], j);
      if (!(d > 0.0)) s.bad = true;
      const double inv = rsqrt_d(d);
      const double xj = s.g[j] * inv;  // L[lane][j] | Y[j][lane-YOFF]
      s.g[j] = xj;
      if (lane == j) linv[j] = inv;
#pragma unroll
      for (int k = j + 1; k < NMAX; ++k) {
        if (k < n) {
          const double lkj = readlane_d(xj, k);
          s.g[k] -= xj * lkj;
        }
      }
    }
  }
}
template <int NMAX, int LD, int YOFF>
__device__ __forceinline__ bool llt_inv_store(const LltInvState<NMAX>& s, double* __restrict__ Ldst, double* __restrict__ Ydst, int n,
                                              int lane) {
  const bool ylane = lane >= YOFF;
  if (lane < n) {
#pragma unroll
    for (int k = 0; k < NMAX; ++k)
      if (k < n) Ldst[lane + k * LD] = (k <= lane) ? s.g[k] : 0.0;
  } else if (ylane && lane < YOFF + NMAX) {
    const int c = lane - YOFF;
#pragma unroll
    for (int k = 0; k < NMAX; ++k) Ydst[k + c * NMAX] = (c < n && k < n) ? s.g[k] : 0.0;
  }
  return s.bad;
}
template <int NMAX, int LD, int YOFF = 16>
__device__ __forceinline__ bool wave_llt_inv(const double* __restrict__ A, double* __restrict__ Ldst,
                                             double* __restrict__ linv, double* __restrict__ Ydst,
                                             int n, int lane) {
  LltInvState<NMAX> s;
  llt_inv_load<NMAX, LD, YOFF>(s, A, n, lane);
  llt_inv_steps<NMAX, 0, NMAX>(s, linv, n, lane);
  return llt_inv_store<NMAX, LD, YOFF>(s, Ldst, Ydst, n, lane);
}

// x <- (L L^T)^-1 x for one right-hand side held in registers (x[NMAX]); L in LDS.
template <int NMAX, int LD>
__device__ __forceinline__ void llt_solve_reg(const double* __restrict__ L,
                                              const double* __restrict__ linv, double (&x)[NMAX],
                                              int n) {
#pragma unroll
  for (int i = 0; i < NMAX; ++i) {
    if (i < n) {
      double v = x[i];
#pragma unroll
      for (int k = 0; k < i; ++k) v -= L[i + k * LD] * x[k];
      x[i] = v * linv[i];
    }
    // keep the scheduler from hoisting the (uniform) loads of the whole factor ahead of the chain:
    // the rows are serially dependent anyway, and ~80 hoisted doubles cost 160 VGPRs
    if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = NMAX - 1; i >= 0; --i) {
    if (i < n) {
      double v = x[i];
#pragma unroll
      for (int k = i + 1; k < NMAX; ++k)
        if (k < n) v -= L[k + i * LD] * x[k];
      x[i] = v * linv[i];
    }
    if ((i & 1) == 0) __builtin_amdgcn_sched_barrier(0);
  }
}

// The same factorisation and inverse factor for 16 < n <= 32, blocked 16 + (n - 16) with the trailing update and the off-diagonal
// block of the inverse on the matrix cores.  The column steps of wave_llt_inv cost (n - j) broadcasts each -- n^2 / 2 readlane
// pairs, every one of them spilled through a VGPR lane in the register-starved tile-split kernels (19k cycles at n = 29).  Here:
//   panel     columns 0..15 of ALL n rows (lanes = rows) by the column steps, 16 - j broadcasts each: L11, L21; Y11 = L11^-1 on
//             lanes 32..47 in the same stream
//   trailing  S = G22 - L21 L21^T: 4 MFMAs, both operands the same fragment of L21
//   block 2   S = L22 L22^T by the column steps (n - 16 columns), Y22 = L22^-1 beside it
//   inverse   Y21 = -Y22 (L21 Y11): two MFMA products chained through the C layout
// Same outputs as wave_llt_inv: L (lower, zeros above) at Ldst (ld LD), 1/diag(L) at linv, Y column-major (ld NMAX) at Ydst; scr: 256
// doubles of LDS scratch.  One wave; LDS hand-offs inside it are ordered by wave_lds_fence.
__device__ __forceinline__ void wave_lds_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}
template <int NMAX, int LD>
__device__ __forceinline__ bool wave_llt_inv_blocked(const double* __restrict__ A, double* __restrict__ Ldst, double* __restrict__ linv,
                                                     double* __restrict__ Ydst, double* __restrict__ scr, int lane) {
  static_assert(NMAX > 16 && NMAX <= 32, "two blocks");
  constexpr int n = NMAX, N2 = NMAX - 16;
  const int li = lane & 15, q = lane >> 4;
  const bool ylane = lane >= 32;
  const int yc = lane - 32;            // column of Y11 / Y22 this lane carries
  const int row = lane < n ? lane : 0;
  bool bad = false;
  // ---- panel: columns 0..15 ----
  double g[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) g[k] = ylane ? ((k == yc) ? 1.0 : 0.0) : A[row + k * LD];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const double d = readlane_d(g[j], j);
    if (!(d > 0.0)) bad = true;
    const double inv = rsqrt_d(d);
    const double xj = g[j] * inv;   // L[lane][j] | Y11[j][yc]
    g[j] = xj;
    if (lane == j) linv[j] = inv;
#pragma unroll
    for (int k = j + 1; k < 16; ++k) {
      const double lkj = readlane_d(xj, k);
      g[k] -= xj * lkj;
    }
  }
  if (lane < n) {
#pragma unroll
    for (int k = 0; k < 16; ++k) Ldst[lane + k * LD] = (k <= lane) ? g[k] : 0.0;
#pragma unroll
    for (int k = 16; k < n; ++k)
      if (lane < 16) Ldst[lane + k * LD] = 0.0;   // the block above the diagonal
  } else if (ylane && yc < 16) {
#pragma unroll
    for (int k = 0; k < 16; ++k) Ydst[k + yc * NMAX] = g[k];
#pragma unroll
    for (int k = 16; k < n; ++k) Ydst[k + yc * NMAX] = 0.0;   // placeholder of Y21 (overwritten below)
  }
  wave_lds_fence();
  // ---- trailing update on the matrix cores: S = G22 - L21 L21^T (C layout: row q + 4r, column li) ----
  {
    d4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int u0 = drow(q, r), u1 = li;
      acc[r] = (u0 < N2 && u1 < N2) ? A[(16 + u0) + (16 + u1) * LD] : ((u0 == u1) ? 1.0 : 0.0);   // padding: identity
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const double v = (li < N2) ? Ldst[(16 + li) + (ks * 4 + q) * LD] : 0.0;   // L21[li][4 ks + q]: A and B fragment alike
      acc = mfma16(-v, v, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) scr[drow(q, r) + li * 16] = acc[r];
  }
  wave_lds_fence();
  // ---- block 2 ----
  double h[16];
  const int row2 = lane < N2 ? lane : 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) h[k] = ylane ? ((k == yc) ? 1.0 : 0.0) : scr[row2 + k * 16];
#pragma unroll
  for (int j = 0; j < N2; ++j) {
    const double d = readlane_d(h[j], j);
    if (!(d > 0.0)) bad = true;
    const double inv = rsqrt_d(d);
    const double xj = h[j] * inv;   // L22[lane][j] | Y22[j][yc]
    h[j] = xj;
    if (lane == j) linv[16 + j] = inv;
#pragma unroll
    for (int k = j + 1; k < N2; ++k) {
      const double lkj = readlane_d(xj, k);
      h[k] -= xj * lkj;
    }
  }
  if (lane < N2) {
#pragma unroll
    for (int k = 0; k < N2; ++k) Ldst[(16 + lane) + (16 + k) * LD] = (k <= lane) ? h[k] : 0.0;
  } else if (ylane && yc < N2) {
#pragma unroll
    for (int k = 0; k < 16; ++k) Ydst[k + (16 + yc) * NMAX] = 0.0;            // Y12 = 0
#pragma unroll
    for (int k = 0; k < N2; ++k) Ydst[(16 + k) + (16 + yc) * NMAX] = h[k];   // Y22
  }
  wave_lds_fence();
  // ---- Y21 = -Y22 (L21 Y11) ----
  {
    d4 t = zero4(), y = zero4();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const double a_ = (li < N2) ? Ldst[(16 + li) + (ks * 4 + q) * LD] : 0.0;   // L21[i = li][k]
      const double b_ = Ydst[(ks * 4 + q) + li * NMAX];                            // Y11[k][n = li]
      t = mfma16(a_, b_, t);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {   // k = 4 ks + q < N2 (Y22 is zero-padded by the masks)
      const int k = ks * 4 + q;
      const double a_ = (li < N2 && k < N2) ? Ydst[(16 + li) + (16 + k) * NMAX] : 0.0;   // Y22[i = li][k]
      y = mfma16(-a_, t[ks], y);   // the C layout of T (row q + 4 ks, column li) is the B fragment of k-step ks
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = drow(q, r);
      if (i < N2) Ydst[(16 + i) + li * NMAX] = y[r];
    }
  }
  wave_lds_fence();
  return bad;
}

// the shared stage fragments (*.inc) synchronise the threads of ONE instance through this macro
#define RTOC_BLOCK_SYNC() __syncthreads()
template <int NV, int NU, int NS, int NW>
__global__ __launch_bounds__(64 * NW) void riccati_backward_kernel(BwdArgs a) {
  using C = BwdCfg<NV, NU, NS, NW>;
  constexpr int NX = C::NX, NT = C::NT, LDP = C::LDP, TNX = C::TNX, TMA = C::TMA, TNU = C::TNU,
                CNT = C::CNT;
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout KL = SL.kkt, RL = SL.ric;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* const sP = smem + C::OFF_P;
  double* const sA = smem + C::OFF_A;
  double* const sPB = smem + C::OFF_PB;
  double* const sH = smem + C::OFF_H;
  double* const sKt = smem + C::OFF_KT;
  double* const sGK = smem + C::OFF_GK;
  double* const sBv = smem + C::OFF_BV;
  double* const sG = smem + C::OFF_G;
  double* const sL = smem + C::OFF_L;

  const int tid0 = threadIdx.x;
  const int b = a.first + blockIdx.x;
  if (b >= a.batch) return;
  int tid = tid0, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
  const int N = a.nstages - 1;
  const size_t kinst = (size_t)b * a.nstages * KL.stride;
  const size_t rinst = (size_t)b * a.nstages * RL.stride;
  unsigned stat = 0;

  // horizon-scan mode: this workgroup handles grid point blockIdx.y only (no sto grids in this mode)
  const bool one_stage = a.scan_ps != nullptr;
  const int my_stage = one_stage ? (int)blockIdx.y + a.seg_lo : N;
  if (one_stage && my_stage > N) {
    const int st = my_stage - a.nstages;   // grid point whose bundle this workgroup prepares
    if (a.sto_scr && st < N) {
      const size_t rec = (size_t)b * a.nstages + st;
      const unsigned ps = scan::sto_prep_body<NV, NU, NS, NT>(a.grid[st], a.kkt + rec * KL.stride, a.scan_ps + (rec + 1) * (size_t)a.scan_ps_stride,
                                                            a.sto_scr + rec * scan::StoScratch<NV, NU, NS>::STRIDE, smem, tid0);
      if (ps && tid0 == 0) atomicOr(&a.status[b], ps);
    }
    return;
  }
  int st_first = N - 1, st_last = 0;
  if (one_stage && my_stage < N) {
    st_first = st_last = my_stage;
  } else
  // ---- terminal stage: P_N = Qxx_N, s_N = -lx_N (riccati_recursion.cpp:37-38) ----
  {
    const double* kr = a.kkt + kinst + (size_t)N * KL.stride;
    double* rr = a.ric + rinst + (size_t)N * RL.stride;
    copy_g2s_mat<NT, NX, NX, LDP>(sP, kr + KL.off[RTOC_KKT_QXX], tid);
    if (tid < NX) {
      const double v = -kr[KL.off[RTOC_KKT_LX] + tid];
      smem[C::V_SN + tid] = v;
      smem[C::V_PSIN + tid] = 0.0;
      smem[C::V_PHIN + tid] = 0.0;
      rr[RL.off[RTOC_RIC_S] + tid] = v;
    }
    if (tid < 8) smem[C::V_SCN + tid] = 0.0;
    __syncthreads();
    copy_s2g_mat<NT, NX, NX, LDP>(rr + RL.off[RTOC_RIC_P], sP, tid);
  }

  // prefetch registers (next stage's record, loaded one stage ahead)
  constexpr int N2B = (NV * NU + 1) / 2, N2G = (NU * NU + 1) / 2;
  // Who prefetches (PW0 = first prefetching wave, PT threads, ptid = index among them): every wave.  The 84 KB of an iCub record
  // arrive at the ~11 B/clk a CU gets when all CUs burst at once, and the issuing waves stall for those ~6k cycles (the queues
  // fill).  Leaving wave 0 -- which owns two of the five column tiles of nv = 35 and is the critical wave of every interval -- out
  // of it (PW0 = 1) was measured: 5.82 -> 6.19 ms per 1024 instances; three waves take longer over the same bytes than wave 0 saves.
  constexpr int PW0 = 0, PT = NT - 64 * PW0;
  const int ptid = tid0 - 64 * PW0;
  PreBuf<MatMap<PT, NX>::passes(NX)> preA;
  PreBuf<MatMap<PT, NX>::passes(NU)> preH;
  PreBuf<PreCnt<PT, N2B>::value> preB;
  PreBuf<PreCnt<PT, N2G>::value> preG;
  double preFx = 0.0, preLx = 0.0, preLu = 0.0;
  auto issue_loads = [&](int stage) {
    const double* kp = a.kkt + kinst + (size_t)stage * KL.stride;
    const bool imp = a.grid[stage].type == RTOC_GRID_IMPACT;
    (void)imp;   // every load is issued on every grid point (an impact record has the fields too; they are not stored to LDS there):
    if (ptid < 0) return;   // wave-uniform
    pre_load_mat<PT, NX, NX>(preA, kp + KL.off[RTOC_KKT_FXX], ptid);   // no branch between the loads, so the compiler counts them
    pre_load_mat<PT, NX, NU>(preH, kp + KL.off[RTOC_KKT_QXU], ptid);
    pre_load<PT, N2B>(preB, kp + KL.off[RTOC_KKT_FVU], ptid);
    pre_load<PT, N2G>(preG, kp + KL.off[RTOC_KKT_QUU], ptid);
    preFx = kp[KL.off[RTOC_KKT_FX] + (ptid < NX ? ptid : 0)];
    preLx = kp[KL.off[RTOC_KKT_LX] + (ptid < NX ? ptid : 0)];
    preLu = kp[KL.off[RTOC_KKT_LU] + (ptid < NU ? ptid : 0)];
  };
  if (one_stage && my_stage == N) return;  // the terminal record is written, nothing else to do
  if (N >= 1) issue_loads(st_first);
  if (one_stage) {  // P+ / s+ of this grid point from the scan's value records (the record's loads are in flight)
    const double* pn = a.scan_ps + ((size_t)b * a.nstages + my_stage + 1) * a.scan_ps_stride;
    copy_g2s_mat<NT, NX, NX, LDP>(sP, pn, tid);
    if (tid < NX) {
      smem[C::V_SN + tid] = pn[a.scan_ps_soff + tid];
      smem[C::V_PSIN + tid] = 0.0;
      smem[C::V_PHIN + tid] = 0.0;
    }
    if (tid < 8) smem[C::V_SCN + tid] = 0.0;
    __syncthreads();
  }

  for (int st = st_first; st >= st_last; --st) {
    // Opaque re-definition of the thread index per stage: keeps LLVM's LICM from hoisting the
    // (hundreds of) per-lane LDS/HBM address computations of the unrolled copy and MFMA loops
    // out of the stage loop, where they would pin > 256 VGPRs and spill.
    tid = tid0;
    asm volatile("" : "+v"(tid));
    lane = tid & 63;
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    li = lane & 15;
    q = lane >> 4;
    const rtoc_grid g = a.grid[st];
    const rtoc_grid gn = a.grid[st + 1];
    const bool impact = (g.type == RTOC_GRID_IMPACT);
    const bool next_lift = (gn.type == RTOC_GRID_LIFT);
    const int ns = impact ? 0 : g.dims;
    const bool sto = g.sto != 0, sto_next = g.sto_next != 0;
    const double* kr = a.kkt + kinst + (size_t)st * KL.stride;
    double* rr = a.ric + rinst + (size_t)st * RL.stride;

    RTOC_PROF(0);
#define RTOC_GRID_PREV_STO (a.grid[st - 1].sto != 0)
#define RTOC_PT_TOP_SYNC() RTOC_BLOCK_SYNC()
#define RTOC_PT_SKIP one_stage
#include "riccati_pt_block.inc"
#undef RTOC_PT_SKIP
#undef RTOC_PT_TOP_SYNC
#undef RTOC_GRID_PREV_STO
    RTOC_PROF(1);
    // ---- stage data: prefetched registers -> LDS (the HBM loads were issued one stage ahead) ----
    if (ptid >= 0) {
      pre_store_mat<PT, NX, NX, LDP>(sA, preA, ptid);
      if (!impact) {
        pre_store_ld<PT, N2B, NV, C::LDB>(sBv, preB, ptid);
        pre_store_mat<PT, NX, NU, LDP>(sH, preH, ptid);
        pre_store_flat<PT, N2G>(sG, preG, ptid);
      }
      if (ptid < NX) smem[C::V_FX + ptid] = preFx, smem[C::V_LX + ptid] = preLx;
      if (!impact && ptid < NU) smem[C::V_LU + ptid] = preLu;
    }
    if (tid < NX) {
      if (sto) {
        smem[C::V_FFX + tid] = kr[KL.off[RTOC_KKT_FFX] + tid];
        smem[C::V_HX + tid] = kr[KL.off[RTOC_KKT_HX] + tid];
      }
    }
    if (!impact && tid < NU) {
      if (sto) smem[C::V_HU + tid] = kr[KL.off[RTOC_KKT_HU] + tid];
    }
    if (sto && tid < 8) smem[C::V_KSC + tid] = kr[KL.off[RTOC_KKT_SCAL] + tid];
    __syncthreads();

    RTOC_PROF(2);
    // ---- z = s+ - P+ Fx ;  y = P+ fx + Psi+ (STO) ----
    if (tid < NX) {
      // four accumulators, uniform STO test hoisted out of the loop: a single dependent chain with a
      // branch per iteration cost ~8k cycles at nx = 64
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      static_assert(NX % 4 == 0 || NX % 2 == 0, "state dimension is even");
#pragma unroll 2  // partial: full unrolling hoists all 2*NX loads (register explosion at nx = 64, 70)
      for (int k = 0; k + 3 < NX; k += 4) {
        a0 += sP[tid + k * LDP] * smem[C::V_FX + k];
        a1 += sP[tid + (k + 1) * LDP] * smem[C::V_FX + k + 1];
        a2 += sP[tid + (k + 2) * LDP] * smem[C::V_FX + k + 2];
        a3 += sP[tid + (k + 3) * LDP] * smem[C::V_FX + k + 3];
      }
#pragma unroll
      for (int k = NX - NX % 4; k < NX; ++k) a0 += sP[tid + k * LDP] * smem[C::V_FX + k];
      smem[C::V_Z + tid] = smem[C::V_SN + tid] - ((a0 + a1) + (a2 + a3));
      if (sto) {
        double y0 = 0.0, y1 = 0.0;
#pragma unroll 4
        for (int k = 0; k + 1 < NX; k += 2) {
          y0 += sP[tid + k * LDP] * smem[C::V_FFX + k];
          y1 += sP[tid + (k + 1) * LDP] * smem[C::V_FFX + k + 1];
        }
        smem[C::V_Y + tid] = (y0 + y1) + smem[C::V_PSIN + tid];
      }
    }

    if (!impact) {
      // ---- PB = P+[:,v] Bv  (NX x NU, K = NV) ----
      {
        d4 acc[CNT][TNU];
#pragma unroll
        for (int c = 0; c < CNT; ++c)
#pragma unroll
          for (int t = 0; t < TNU; ++t) acc[c][t] = zero4();
        const double* pa_ = sP + (NV + q) + (wave * 16 + li) * LDP;  // P+[NV+k][i] = P+[i][NV+k]: k contiguous, conflict-free
        const double* pb_ = sBv + q + li * C::LDB;                    // Bv[k][u]
#pragma unroll
        for (int ks = 0; ks < (NV + 3) / 4; ++ks) {
          const bool kok = (ks * 4 + 3 < NV) || (ks * 4 + q < NV);
          double bv[TNU];
#pragma unroll
          for (int t = 0; t < TNU; ++t) {
            const double v = pb_[ks * 4 + t * 16 * C::LDB];
            bv[t] = (kok && (t * 16 + li < NU)) ? v : 0.0;
          }
#pragma unroll
          for (int c = 0; c < CNT; ++c) {
            const double v = pa_[c * NW * 16 * LDP + ks * 4];
            const double av = (kok && ((wave + c * NW) * 16 + li < NX)) ? v : 0.0;
#pragma unroll
            for (int t = 0; t < TNU; ++t) acc[c][t] = mfma16(av, bv[t], acc[c][t]);
          }
        }
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
          const int tm = wave + c * NW;
#pragma unroll
          for (int t = 0; t < TNU; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = tm * 16 + drow(q, r), u = t * 16 + li;
              double* const d = (i < NX && u < NU) ? sPB + i + u * LDP : smem + C::V_BTS + (lane & 7);
              *d = acc[c][t][r];
            }
        }
      }
      __syncthreads();
      // ---- G = Quu + Bv^T PB[v,:] (its TNU x TNU tiles dealt to the waves) ; lu' = lu - Bv^T z[v] ----
      {
        const double* pa_ = sBv + q + li * C::LDB;    // Bv^T[u][k] = Bv[k][u]
        const double* pb_ = sPB + NV + q + li * LDP;  // PB[NV+k][u]
#pragma unroll
        for (int t = 0; t < TNU * TNU; ++t) {
          if ((t % NW) != wave) continue;  // wave-uniform
          const int t0 = t / TNU, t1 = t % TNU;
          d4 acc = zero4();
#pragma unroll
          for (int ks = 0; ks < (NV + 3) / 4; ++ks) {
            const bool kok = (ks * 4 + 3 < NV) || (ks * 4 + q < NV);
            const double va = pa_[ks * 4 + t0 * 16 * C::LDB];
            const double vb = pb_[ks * 4 + t1 * 16 * LDP];
            acc = mfma16((kok && t0 * 16 + li < NU) ? va : 0.0, (kok && t1 * 16 + li < NU) ? vb : 0.0, acc);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int u0 = t0 * 16 + drow(q, r), u1 = t1 * 16 + li;
            double* const d = (u0 < NU && u1 < NU) ? sG + u0 + u1 * NU : smem + C::V_BTS + (lane & 7);
            *d += acc[r];
          }
        }
      }
      if (tid < NU) {
        double acc = 0.0, ap = 0.0, aph = 0.0;
        {
          double b0 = 0.0, b1 = 0.0;
#pragma unroll
          for (int k = 0; k + 1 < NV; k += 2) {
            b0 += sBv[k + tid * C::LDB] * smem[C::V_Z + NV + k];
            b1 += sBv[k + 1 + tid * C::LDB] * smem[C::V_Z + NV + k + 1];
          }
          if (NV & 1) b0 += sBv[NV - 1 + tid * C::LDB] * smem[C::V_Z + 2 * NV - 1];
          acc = b0 + b1;
        }
        if (sto) {
#pragma unroll
          for (int k = 0; k < NV; ++k) {
            const double bv = sBv[k + tid * C::LDB];
            ap += bv * smem[C::V_Y + NV + k];
            aph += bv * (sto_next ? smem[C::V_PHIN + NV + k] : 0.0);
          }
        }
        smem[C::V_LU + tid] -= acc;
        if (sto) {
          smem[C::V_PSIU + tid] = ap + smem[C::V_HU + tid];  // psi_u (brrf.cpp:53-57)
          smem[C::V_PHIU + tid] = sto_next ? aph : 0.0;      // phi_u (:58-65)
        }
      }
    } else {
      // impact grid: no control; zero the augmented rows so that they add nothing
      for (int e = tid; e < NU * LDP; e += NT) sPB[e] = 0.0;
    }
    __syncthreads();

    RTOC_PROF(3);
    // ---- PAa = [P+ ; PB^T] A, this wave owns column tiles tn = wave + c*NW ----
    d4 pa[TMA][CNT];
#pragma unroll
    for (int tm = 0; tm < TMA; ++tm)
#pragma unroll
      for (int c = 0; c < CNT; ++c) pa[tm][c] = zero4();
    {
      // per-lane A-operand addressing: row i of [P+ ; PB^T].  Tiles that lie entirely in P+ or
      // entirely in PB^T use compile-time strides; only the (at most one) mixed tile selects per lane.
      const double* pb_ = sA + q + (wave * 16 + li) * LDP;  // A[k][j], j in the owned column tile
#pragma unroll
      for (int ks = 0; ks < (NX + 3) / 4; ++ks) {
        const bool kok = (ks * 4 + 3 < NX) || (ks * 4 + q < NX);
        double av[TMA], bv[CNT];
#pragma unroll
        for (int tm = 0; tm < TMA; ++tm) {
          const int i = tm * 16 + li;
          double v;
          if (tm * 16 + 15 < NX) {
            v = sP[q + ks * 4 + (li + tm * 16) * LDP];  // P+[k][i] = P+[i][k]: k contiguous, conflict-free (riccati_backward_rs.hpp)
          } else if (tm * 16 >= NX) {
            v = sPB[q + li * LDP + (tm * 16 - NX) * LDP + ks * 4];  // PB[k][i-NX]
            if (tm * 16 + 15 >= NX + NU) v = (i < NX + NU) ? v : 0.0;
          } else {
            const double vp = sP[q + ks * 4 + (li + tm * 16) * LDP];
            const double vb = sPB[q + li * LDP + (tm * 16 - NX) * LDP + ks * 4];
            v = (i < NX) ? vp : vb;
            if (tm * 16 + 15 >= NX + NU) v = (i < NX + NU) ? v : 0.0;
          }
          av[tm] = kok ? v : 0.0;
        }
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
          const double v = pb_[ks * 4 + c * NW * 16 * LDP];
          bv[c] = (kok && ((wave + c * NW) * 16 + li < NX)) ? v : 0.0;
        }
        // a wave whose second column tile lies beyond the matrix (TNX tiles over NW waves: iCub nv = 35 has 5 over 4) skips its
        // products instead of multiplying zeros -- it is the wave that takes the Cholesky below in the time this leaves
#pragma unroll
        for (int c = 0; c < CNT; ++c)
          if (c == 0 || wave + c * NW < TNX) {
#pragma unroll
            for (int tm = 0; tm < TMA; ++tm) pa[tm][c] = mfma16(av[tm], bv[c], pa[tm][c]);
          }
      }
    }
    // H += (A^T PB)  : rows >= NX of PAa hold (H - Qxu)^T
    if (!impact) {
#pragma unroll
      for (int tm = 0; tm < TMA; ++tm)
#pragma unroll
        for (int c = 0; c < CNT; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = tm * 16 + drow(q, r);
            const int j = (wave + c * NW) * 16 + li;
            if (tm * 16 + 4 * r + 3 >= NX) {  // compile-time prune of pure-P register groups
              const bool ok = row >= NX && row < NX + NU && j < NX;   // branch-free: the others add into a dummy slot
              double* const d = ok ? sH + j + (row - NX) * LDP : smem + C::V_BTS + (lane & 7);
              *d += pa[tm][c][r];
            }
          }
    }

    RTOC_PROF(4);
    // Qxx of THIS stage straight from HBM into the accumulators of the F product, in the MFMA C layout (row i = q + 4r of the
    // owned row tile, column j = lane & 15 of tile t): no prefetch registers, no LDS staging, no barriers -- the loads fly while
    // w = A^T z and the Cholesky run.  The four r of a lane read the same 128-byte line of column j.
    d4 f[CNT][TNX];
    {
      const double* qx_ = kr + KL.off[RTOC_KKT_QXX] + (wave * 16 + q) + li * NX;   // Qxx[i][j]
#pragma unroll
      for (int c = 0; c < CNT; ++c)
#pragma unroll
        for (int t = 0; t < TNX; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = (wave + c * NW) * 16 + drow(q, r), j = t * 16 + li;
            const bool ok = i < NX && j < NX;
            const double v = qx_[ok ? (c * NW * 16 + 4 * r + t * 16 * NX) : 0];
            f[c][t][r] = ok ? v : 0.0;
          }
    }
    // ---- LLT(G) (riccati_factorizer.cpp:49) by the LAST wave, right behind its P+ A tiles: there is no barrier between P+ A and
    //      the end of the F chain, and the last wave owns the fewest column tiles (iCub nv = 35: one of five, wave 0 two) and no
    //      rows of w = A^T z -- 19k cycles of dependent VALU / readlane work that used to sit on wave 0 on top of its two tiles ----
    RTOC_PROF(13);
    if (!impact && wave == NW - 1) {
      if constexpr (NU <= 32) {
        // with the inverse factor Y = L^-1 in the same instruction stream (dead Bv buffer): the
        // triangular solves of the policy become MFMA products below
        if constexpr (NU > 16 && NX > 63) {   // blocked, trailing update and Y21 on the matrix cores (scratch: the Z^T slot, free now)
          if (wave_llt_inv_blocked<NU, NU>(sG, sL, smem + C::V_LINV, sBv, smem + C::OFF_ZT, lane)) stat |= RTOC_STAT_QUU_NOT_SPD;
        } else {
          if (wave_llt_inv<NU, NU, 32>(sG, sL, smem + C::V_LINV, sBv, NU, lane)) stat |= RTOC_STAT_QUU_NOT_SPD;
        }
      } else {
        if (wave_llt<NU, NU>(sG, sL, smem + C::V_LINV, NU, lane)) stat |= RTOC_STAT_QUU_NOT_SPD;
      }
    }
    RTOC_PROF(14);
    // ---- s-vector part that needs A: w = A^T z  (and STO: psi_x, phi_x) ----
    if (tid < NX) {
      double acc = 0.0, ap = 0.0, aph = 0.0;
      {
        typedef double dbl2 __attribute__((ext_vector_type(2)));
        static_assert((LDP & 1) == 0 && (C::OFF_A & 1) == 0 && (C::V_Z & 1) == 0 && (NX & 1) == 0, "128-bit LDS reads");
        const dbl2* pa2 = reinterpret_cast<const dbl2*>(sA + tid * LDP);
        const dbl2* pz2 = reinterpret_cast<const dbl2*>(smem + C::V_Z);
        double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
#pragma unroll 2
        for (int k2 = 0; k2 + 1 < NX / 2; k2 += 2) {
          const dbl2 av0 = pa2[k2], zv0 = pz2[k2], av1 = pa2[k2 + 1], zv1 = pz2[k2 + 1];
          w0 += av0.x * zv0.x;
          w1 += av0.y * zv0.y;
          w2 += av1.x * zv1.x;
          w3 += av1.y * zv1.y;
        }
        if ((NX / 2) & 1) {
          const dbl2 av0 = pa2[NX / 2 - 1], zv0 = pz2[NX / 2 - 1];
          w0 += av0.x * zv0.x;
          w1 += av0.y * zv0.y;
        }
        acc = (w0 + w1) + (w2 + w3);
      }
      if (sto) {
#pragma unroll 4
        for (int k = 0; k < NX; ++k) {
          const double av = sA[k + tid * LDP];
          ap += av * (impact ? 0.0 : smem[C::V_Y + k]);
          aph += av * smem[C::V_PHIN + k];
        }
      }
      smem[C::V_SNEW + tid] = acc - smem[C::V_LX + tid];
      if (sto) {
        if (!impact) {
          smem[C::V_PSIX + tid] = ap + smem[C::V_HX + tid];  // psi_x
          smem[C::V_PHIX + tid] = sto_next ? aph : 0.0;      // phi_x
        } else {
          smem[C::V_PHIX + tid] = aph;  // impact: Phi = A^T Phi+ (brrf.cpp:166)
        }
      }
    }

    RTOC_PROF(5);
    // ---- F = Qxx + AtP A, chained: A-operand = PAa registers (row-tile = owned column tile) ----
    {
      // (f holds Qxx of this stage since the top of interval 4: straight from HBM in the MFMA C layout)
      const double* pbf_ = sA + q + li * LDP;  // A[k][j]
#pragma unroll
      for (int tm = 0; tm < TMA; ++tm)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (tm * 16 + 4 * r < NX) {  // register group holds at least one P row (k index)
            const bool kok = (tm * 16 + 4 * r + 3 < NX) || (tm * 16 + 4 * r + q < NX);
            double bv[TNX];
#pragma unroll
            for (int t = 0; t < TNX; ++t) {
              const double v = pbf_[tm * 16 + 4 * r + t * 16 * LDP];
              bv[t] = (kok && (t * 16 + li < NX)) ? v : 0.0;
            }
#pragma unroll
            for (int c = 0; c < CNT; ++c)
              if (c == 0 || wave + c * NW < TNX) {
                const double av = kok ? pa[tm][c][r] : 0.0;
#pragma unroll
                for (int t = 0; t < TNX; ++t) f[c][t] = mfma16(av, bv[t], f[c][t]);
              }
          }
        }
    }
    RTOC_PROF(6);
    __syncthreads();  // L published; sA / sPB no longer read by MFMA after this point
    RTOC_PROF(15);
    RTOC_PROF(16);

    // Without a switching constraint K^T G K = H G^-1 H^T = Z Z^T with Z^T = L^-1 H^T, which the policy products form anyway: the
    // tile-split shapes (one instance per CU, LDS to spare) park Z^T in LDS and run F -= Z Z^T from there -- no G K product and
    // one barrier less.  (riccati_backward_rs.hpp does the same from registers; here the row blocks of F belong to different waves.)
    constexpr bool ZZ_SHAPE = NX > 63 && NU <= 32;
    double* const sZt = smem + C::OFF_ZT;
    const bool zz = ZZ_SHAPE && !impact && ns == 0;
    if (impact) {
      // riccati_factorizer.cpp:178-197 -- no policy
    } else {
      if (ns == 0 && NU <= 32) {
        // K = -G^-1 H^T, k = -G^-1 lu', T = -G^-1 psi_u, W = -G^-1 phi_u (riccati_factorizer.cpp:55-56,
        // :125-130) for all right-hand sides at once, as the two triangular solves written as products
        // with Y = L^-1:  Z^T = Y [H^T | lu' | psi_u | phi_u],  [K | k | T | W] = -Y^T Z^T.
        // Column tiles of the right-hand side are dealt to the waves; the C layout of Z^T (row
        // i = 16 ti + q + 4r, column = lane&15) is the B-operand layout of the second product
        // (k-step 4 ti + r), so Z^T stays in registers.
        constexpr int TKT = (NX + 3 + 15) / 16, TU = (NU + 15) / 16, KSU = (NU + 3) / 4;
        const double* sY = sBv;
        double chk = 0.0;
        for (int c = wave; c < TKT; c += NW) {
          const int x = c * 16 + li;
          const double* bsrc = (x < NX) ? (sH + x + q * LDP)
                                        : (smem + (x == NX ? C::V_LU : (x == NX + 1 ? C::V_PSIU : C::V_PHIU)) + q);
          const int bstr = (x < NX) ? 4 * LDP : 4;
          const bool bok = (x < NX) || x == NX || (sto && (x == NX + 1 || (x == NX + 2 && sto_next)));
          d4 zt[TU], kk[TU];
#pragma unroll
          for (int t = 0; t < TU; ++t) {
            zt[t] = zero4();
            kk[t] = zero4();
          }
#pragma unroll
          for (int ks = 0; ks < KSU; ++ks) {
            const bool kok = (ks * 4 + 3 < NU) || (ks * 4 + q < NU);
            const double bv0 = bsrc[(kok ? ks : 0) * bstr];
            const double bv = (kok && bok) ? bv0 : 0.0;
#pragma unroll
            for (int t = 0; t < TU; ++t) {
              const int i = t * 16 + li;
              const double yv = sY[(i < NU ? i : 0) + ((kok ? ks * 4 + q : 0)) * NU];  // Y[i][u]
              zt[t] = mfma16((kok && i < NU) ? yv : 0.0, bv, zt[t]);
            }
          }
          // results leave branch-free: every lane stores every value, the ones that have no home go to a dummy slot (the per-value
          // "if (u < NU) if (x < NX) ... else if ..." nests compiled to ~150 exec-mask branches per tile: 10k cycles for 32 MFMAs)
          double* const dummy = smem + C::V_BTS + (lane & 7);   // unused by this kernel (the role-split kernel's rider slots)
          if constexpr (ZZ_SHAPE) {
#pragma unroll
            for (int t = 0; t < TU; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int u = t * 16 + drow(q, r);
                double* const dz = (u < NU && x < NX) ? sZt + u + x * NU : dummy;
                *dz = zt[t][r];
              }
          }
#pragma unroll
          for (int ks = 0; ks < KSU; ++ks) {  // k = i = 4 ks + q
            const bool kok = (ks * 4 + 3 < NU) || (ks * 4 + q < NU);
#pragma unroll
            for (int t = 0; t < TU; ++t) {
              const int u = t * 16 + li;
              const double yv = sY[(kok ? ks * 4 + q : 0) + (u < NU ? u : 0) * NU];  // Y[i][u]
              kk[t] = mfma16((kok && u < NU) ? -yv : 0.0, zt[ks / 4][ks % 4], kk[t]);
            }
          }
          {
            // column x of the right-hand side: K^T (x < NX), k (NX), T (NX + 1), W (NX + 2; zero without sto_next)
            double* const dbase = x < NX ? sKt + x : smem + (x == NX ? C::V_KV : (x == NX + 1 ? C::V_TV : C::V_WV));
            const int dstr = x < NX ? LDP : 1;
            const bool okx = x <= NX || (sto && x <= NX + 2);
            const bool zero = x == NX + 2 && !sto_next;
#pragma unroll
            for (int t = 0; t < TU; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int u = t * 16 + drow(q, r);
                const double v = kk[t][r];
                const bool ok = okx && u < NU;
                double* const d = ok ? dbase + u * dstr : dummy;
                *d = zero ? 0.0 : v;
                chk = __builtin_fma((ok && x <= NX) ? v : 0.0, 0.0, chk);
              }
          }
        }
        if (is_bad(chk)) stat |= RTOC_STAT_NAN;
        RTOC_PROF(17);
      } else if (ns == 0) {
        // K = -G^-1 H^T, k = -G^-1 lu   (:55-56); thread t < NX owns column t, thread NX owns k
        if (tid <= NX) {
          double x[NU];
#pragma unroll
          for (int u = 0; u < NU; ++u)
            x[u] = (tid < NX) ? sH[tid + u * LDP] : smem[C::V_LU + u];
          llt_solve_reg<NU, NU>(sL, smem + C::V_LINV, x, NU);
          bool bad = false;
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            bad = bad || is_bad(x[u]);
            if (tid < NX)
              sKt[tid + u * LDP] = -x[u];
            else
              smem[C::V_KV + u] = -x[u];
          }
          if (bad) stat |= RTOC_STAT_NAN;
          if (sto && tid == NX) {
            // T = -G^-1 psi_u ; W = -G^-1 phi_u  (:125-130)
            double t[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) t[u] = smem[C::V_PSIU + u];
            llt_solve_reg<NU, NU>(sL, smem + C::V_LINV, t, NU);
#pragma unroll
            for (int u = 0; u < NU; ++u) smem[C::V_TV + u] = -t[u];
            if (sto_next) {
#pragma unroll
              for (int u = 0; u < NU; ++u) t[u] = smem[C::V_PHIU + u];
              llt_solve_reg<NU, NU>(sL, smem + C::V_LINV, t, NU);
#pragma unroll
              for (int u = 0; u < NU; ++u) smem[C::V_WV + u] = -t[u];
            } else {
#pragma unroll
              for (int u = 0; u < NU; ++u) smem[C::V_WV + u] = 0.0;
            }
          }
        }
      } else if (NS > 0) {
#include "riccati_sc_block.inc"
      }
      __syncthreads();

      RTOC_PROF(7);
      // (the record prefetch of the next stage is issued below, behind the policy products: in front of them the compiler made the
      //  first instruction of this path wait for ALL of it -- s_waitcnt vmcnt(0), ~8k cycles of HBM time on every wave)
      if (a.writeback) {  // mutated Qxu, Quu, lu (reference in-place semantics), before H is reused
        double* kw = a.kkt_rw + kinst + (size_t)st * KL.stride;
        copy_s2g_mat<NT, NX, NU, LDP>(kw + KL.off[RTOC_KKT_QXU], sH, tid);
        copy_s2g_flat<NT>(kw + KL.off[RTOC_KKT_QUU], sG, NU * NU, tid);
        if (tid < NU) kw[KL.off[RTOC_KKT_LU] + tid] = smem[C::V_LU + tid];
        __syncthreads();
      }
      if (!zz)
      // ---- GK = G K (+ 2 Phiu^T M on switching-constraint grids, which folds
      //      P -= KtDtM + KtDtM^T (:84-87) into the symmetrised F - K^T GK) ----
      {
        d4 acc[TNU][CNT];
#pragma unroll
        for (int t = 0; t < TNU; ++t)
#pragma unroll
          for (int c = 0; c < CNT; ++c) acc[t][c] = zero4();
        // original Quu + BtP Fvu is in sG (wave_llt wrote the factor to sL, sG untouched)
        const double* pa_ = sG + li + q * NU;                     // G[u][k]
        const double* pb_ = sKt + (wave * 16 + li) + q * LDP;     // K[k][j] = Kt[j][k]
#pragma unroll
        for (int ks = 0; ks < (NU + 3) / 4; ++ks) {
          const bool kok = (ks * 4 + 3 < NU) || (ks * 4 + q < NU);
          double av[TNU], bv[CNT];
#pragma unroll
          for (int t = 0; t < TNU; ++t) {
            const double v = pa_[t * 16 + ks * 4 * NU];
            av[t] = (kok && (t * 16 + li < NU)) ? v : 0.0;
          }
#pragma unroll
          for (int c = 0; c < CNT; ++c) {
            const double v = pb_[c * NW * 16 + ks * 4 * LDP];
            bv[c] = (kok && ((wave + c * NW) * 16 + li < NX)) ? v : 0.0;
          }
#pragma unroll
          for (int t = 0; t < TNU; ++t)
#pragma unroll
            for (int c = 0; c < CNT; ++c) acc[t][c] = mfma16(av[t], bv[c], acc[t][c]);
        }
#pragma unroll
        for (int t = 0; t < TNU; ++t)
#pragma unroll
          for (int c = 0; c < CNT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int u = t * 16 + drow(q, r), j = (wave + c * NW) * 16 + li;
              if (u < NU && j < NX) {
                double v = acc[t][c][r];
                if (NS > 0 && ns > 0) {
                  double dtm = 0.0;
                  for (int l = 0; l < ns; ++l)
                    dtm += smem[C::S_PHIU + l + u * C::NSP] * smem[C::S_M + l + j * C::NSP];
                  v += 2.0 * dtm;
                }
                sGK[u + j * NU] = v;
              }
            }
      }
      if (!zz) __syncthreads();   // (wave-uniform)
      RTOC_PROF(8);
      // ---- F -= K^T GK  (zz: F -= Z Z^T, A operand Z[i][k] = Z^T[k][i] from the same buffer) ----
      {
        const double* pa_ = zz ? sZt + q + (wave * 16 + li) * NU : sKt + (wave * 16 + li) + q * LDP;  // Z^T[k][i] | K^T[i][k]
        const int pa_c = zz ? NW * 16 * NU : NW * 16, pa_k = zz ? 4 : 4 * LDP;
        const double* pb_ = (zz ? sZt : sGK) + q + li * NU;    // Z^T[k][j] | GK[k][j]
#pragma unroll
        for (int ks = 0; ks < (NU + 3) / 4; ++ks) {
          const bool kok = (ks * 4 + 3 < NU) || (ks * 4 + q < NU);
          double av[CNT], bv[TNX];
#pragma unroll
          for (int c = 0; c < CNT; ++c) {
            const double v = pa_[c * pa_c + ks * pa_k];
            av[c] = (kok && ((wave + c * NW) * 16 + li < NX)) ? -v : 0.0;
          }
#pragma unroll
          for (int t = 0; t < TNX; ++t) {
            const double v = pb_[ks * 4 + t * 16 * NU];
            bv[t] = (kok && (t * 16 + li < NX)) ? v : 0.0;
          }
#pragma unroll
          for (int c = 0; c < CNT; ++c)
#pragma unroll
            for (int t = 0; t < TNX; ++t) f[c][t] = mfma16(av[c], bv[t], f[c][t]);
        }
      }
      // s -= H k  (brrf.cpp:90).  Without switching constraint K^T = -H G^-1, hence H k = K^T lu'
      // (H itself has been overwritten by GK); the constrained path subtracted H k above.
      if (ns == 0 && tid < NX) {
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < NU; ++u) acc += sKt[tid + u * LDP] * smem[C::V_LU + u];
        smem[C::V_SNEW + tid] -= acc;
      }
    }

    RTOC_PROF(9);
    if (st > st_last) issue_loads(st - 1);   // HBM -> registers, in flight during the symmetrisation, the copy-out and the next stage's top
    // ---- optional write-back of the mutated KKT blocks (reference in-place semantics) ----
    if (a.writeback) {
      double* kw = a.kkt_rw + kinst + (size_t)st * KL.stride;
#pragma unroll
      for (int c = 0; c < CNT; ++c)
#pragma unroll
        for (int t = 0; t < TNX; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = (wave + c * NW) * 16 + drow(q, r), j = t * 16 + li;
            if (i < NX && j < NX) kw[KL.off[RTOC_KKT_QXX] + i + j * NX] = f[c][t][r];
          }
    }

    // ---- P = (F + F^T)/2 (brrf.cpp:85): F -> sP, read back the transposed element in the MFMA
    //      register layout (affine LDS addresses), average, store.  0.5*(a+b) is commutative, so
    //      P is exactly symmetric. ----
    {
      double* pw_ = sP + (wave * 16 + q) + li * LDP;        // F[i][j] at i + j*LDP
      double* const pdummy = smem + C::V_BTS + (lane & 7);   // home of the entries beyond the matrix (unused slot of this kernel)
      const double* pr_ = sP + li + (wave * 16 + q) * LDP;  // F[j][i]
#pragma unroll
      for (int c = 0; c < CNT; ++c)
#pragma unroll
        for (int t = 0; t < TNX; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = (wave + c * NW) * 16 + drow(q, r), j = t * 16 + li;
            double* const d = (i < NX && j < NX) ? pw_ + (c * NW * 16 + 4 * r + t * 16 * LDP) : pdummy;   // branch-free (see the policy products)
            *d = f[c][t][r];
          }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < CNT; ++c)
#pragma unroll
        for (int t = 0; t < TNX; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double v = pr_[t * 16 + (c * NW * 16 + 4 * r) * LDP];
            f[c][t][r] = 0.5 * (f[c][t][r] + v);
          }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < CNT; ++c)
#pragma unroll
        for (int t = 0; t < TNX; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = (wave + c * NW) * 16 + drow(q, r), j = t * 16 + li;
            double* const d = (i < NX && j < NX) ? pw_ + (c * NW * 16 + 4 * r + t * 16 * LDP) : pdummy;
            *d = f[c][t][r];
          }
    }

    RTOC_PROF(10);
#include "riccati_sto_block.inc"
    __syncthreads();

    RTOC_PROF(11);
    // ---- results -> HBM; roll the LDS "next" state ----
    copy_s2g_mat<NT, NX, NX, LDP>(rr + RL.off[RTOC_RIC_P], sP, tid);
    if (!impact) {
      // K row-major nu x nx == Kt column-major nx x nu
      copy_s2g_mat<NT, NX, NU, LDP>(rr + RL.off[RTOC_RIC_K], sKt, tid);
      if (tid < NU) {
        rr[RL.off[RTOC_RIC_KV] + tid] = smem[C::V_KV + tid];
        if (sto) {
          rr[RL.off[RTOC_RIC_T] + tid] = smem[C::V_TV + tid];
          rr[RL.off[RTOC_RIC_W] + tid] = smem[C::V_WV + tid];
          rr[RL.off[RTOC_RIC_PSIU] + tid] = smem[C::V_PSIU + tid];
          rr[RL.off[RTOC_RIC_PHIU] + tid] = smem[C::V_PHIU + tid];
        }
      }
    }
    if (tid < NX) {
      const double sv = smem[C::V_SNEW + tid];
      const double psi = smem[C::V_PSI + tid], phi = smem[C::V_PHI + tid];
      rr[RL.off[RTOC_RIC_S] + tid] = sv;
      rr[RL.off[RTOC_RIC_PSI] + tid] = psi;
      rr[RL.off[RTOC_RIC_PHI] + tid] = phi;
      if (sto && !impact) {
        rr[RL.off[RTOC_RIC_PSIX] + tid] = smem[C::V_PSIX + tid];
        rr[RL.off[RTOC_RIC_PHIX] + tid] = smem[C::V_PHIX + tid];
      }
      smem[C::V_SN + tid] = sv;
      smem[C::V_PSIN + tid] = psi;
      smem[C::V_PHIN + tid] = phi;
    }
    if (tid < 5) {
      const double v = smem[C::V_SC + tid];
      rr[RL.off[RTOC_RIC_SCAL] + tid] = v;
      smem[C::V_SCN + tid] = v;
    }
    RTOC_PROF(12);
  }

  // ---- grid[0].sto: trailing phase transition writes sto_policy_[0] (riccati_recursion.cpp:75-79) ----
  __syncthreads();
  {
    const rtoc_grid g0 = a.grid[0];
    if (g0.sto && g0.sto_next) {
      double* pr = a.ric + rinst;
      const double xi = smem[C::V_SCN + 0], chi = smem[C::V_SCN + 1], rho = smem[C::V_SCN + 2],
                   eta = smem[C::V_SCN + 3], iota = smem[C::V_SCN + 4];
      double sgm = xi - 2.0 * chi + rho;
      const double eps = 1.4901161193847656e-08;
      if ((sgm * a.max_dts0) < fabs(eta - iota) || sgm < eps)
        sgm = fabs(sgm) + fabs(eta - iota) / a.max_dts0;
      const double isg = 1.0 / sgm;
      if (tid < NX)
        pr[RL.off[RTOC_RIC_DTSDX] + tid] = -isg * (smem[C::V_PSIN + tid] - smem[C::V_PHIN + tid]);
      if (tid == 0) {
        pr[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTSDTS] = isg * (xi - chi);
        pr[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTS0] = -isg * (eta - iota);
      }
    }
  }
  if (stat) atomicOr(&a.status[b], stat);
}

#undef RTOC_BLOCK_SYNC

}  // namespace rtoc
