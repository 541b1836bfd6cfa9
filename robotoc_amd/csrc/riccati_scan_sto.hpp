// riccati_scan_sto.hpp -- the horizon scan on grids WITH switching-time optimisation: "matrix scan + serial vector pass".
//
// The phase transitions of the backward recursion (reference src/riccati/riccati_factorizer.cpp:145-175) leave P untouched
// (riccati_m.P = riccati.P), and no STO term enters P, K or M (riccati_factorizer.cpp:44-90, brrf.cpp:31-91): the MATRIX half
// of the recursion is the same associative scan as without STO (riccati_scan_core.hpp) -- value matrices P_i of all grid points
// in ceil(log2 n) levels, then K_i, M_i, P_i of all grid points at once (one-stage mode of riccati_backward_kernel).  The VECTOR
// half -- s, k, m and the STO quantities Psi, Phi, psi_x, psi_u, phi_x, phi_u, T, W, mt, mt_next, xi, chi, rho, eta, iota, the
// STOPolicy of every transition -- is NOT a composition of fixed affine maps: sgm = xi - 2 chi + rho is regularised by a
// data-dependent branch (:159-162) and divides the update of s and Phi (:168-175).  It is a chain of O(nx^2) mat-vecs per grid
// point once the matrices are there, run serially by ONE workgroup per instance (sto_vector_body), after a stage-parallel
// preparation (sto_prep_body: extra workgroups of the policy kernel's launch) has taken everything off the chain that does not
// depend on it:
//   P+ Fx, P+ fx, and the inverse of G = Quu + Bv^T P+[v,v] Bv:
//     no switching constraint:  Ginv = Y^T Y, Y = L^-1 (L L^T = G);  k = -Ginv lu' is ONE product on the chain (the reference's
//     llt.solve is two dependent triangular ones; same conditioning)
//     switching constraint (riccati_factorizer.cpp:58-77): Ginv' = Ginv - SinvDGinv^T DGinv, SinvDGinv, Sinv explicitly,
//     as the reference forms them.
// The chain itself, per control grid point (z = P+ Fx - s+, y = P+ fx + Psi+; K, M from the matrix half):
//   lu' = lu + Bv^T z[v]                      k = -Ginv lu' (- SinvDGinv^T P)        m = Sinv P - SinvDGinv lu'
//   s   = -A^T z - lx - K^T lu' (- M^T P)     [H k = K^T lu' - Phix^T m + M^T P, so the reference's s - Phix^T m needs no H]
//   psi_x = A^T y + hx   psi_u = Bv^T y[v] + hu   phi_x = A^T Phi+   phi_u = Bv^T Phi+[v]
//   T = -Ginv psi_u (- SinvDGinv^T Phit)   W = -Ginv phi_u   mt = Sinv Phit - SinvDGinv psi_u   mt_next = -SinvDGinv phi_u
//   Psi = psi_x + K^T psi_u (+ M^T Phit)   Phi = phi_x + K^T phi_u   scalars as brrf.cpp:110-142, riccati_factorizer.cpp:136-141
// impact grid points (:178-197): s = -A^T z - lx, Phi = A^T Phi+, iota += Phi+ . Fx.
// Same barrier-separated "for (i = tid; i < n; i += NT)" style as riccati_scan_core.hpp: the bodies compile for the host with
// NT = 1 (tests/cpp/scan_emulation.cpp checks them on the CPU against the oracle's serial recursion).
#pragma once
#include "riccati_scan_core.hpp"

namespace rtoc {
namespace scan {

// The BUNDLE of one grid point: everything its step of the chain reads that does not depend on the chain, except K and M (which the
// policy kernel writes to the Riccati record next to this preparation).  One contiguous block per grid point in the scratch buffer,
// in the order the vector pass keeps it in LDS, padded to whole 16-byte loads of a 256-thread workgroup: the pass moves it with
// unguarded dwordx4 loads and ds_write_b128s and nothing else.
template <int NV, int NU, int NS>
struct StoScratch {
  static constexpr int NX = 2 * NV, NSP = NS > 0 ? NS : 1, VX = pad8(NX), VU = pad8(NU), VS = pad8(NSP), CHUNK = 512;
  static constexpr int O_AT = 0, O_BV = O_AT + pad8(NX * NX), O_GI = O_BV + pad8(NV * NU), O_SDG = O_GI + pad8(NU * NU);
  static constexpr int O_SIN = O_SDG + pad8(NSP * NU), O_PFX = O_SIN + pad8(NSP * NSP), O_PFF = O_PFX + VX, O_FX = O_PFF + VX;
  static constexpr int O_FFX = O_FX + VX, O_LX = O_FFX + VX, O_HX = O_LX + VX, O_LU = O_HX + VX, O_HU = O_LU + VU, O_PRES = O_HU + VU;
  static constexpr int O_PHIT = O_PRES + VS, O_SCAL = O_PHIT + VS, END = O_SCAL + 8, STRIDE = (END + CHUNK - 1) / CHUNK * CHUNK;
};

template <int NV, int NU, int NS>
struct StoPrepCfg {
  static constexpr int NX = 2 * NV, NSP = NS > 0 ? NS : 1;
  // LDS: P+ (NX^2), PB (NV x NU), G / L (NU^2), Y (NU^2), Ginv (NU^2), DGinv (NSP x NU), S / Ls (NSP^2), Ys (NSP^2), vectors
  static constexpr int S_P = 0, S_PB = S_P + pad8(NX * NX), S_G = S_PB + pad8(NV * NU), S_Y = S_G + pad8(NU * NU);
  static constexpr int S_GI = S_Y + pad8(NU * NU), S_DG = S_GI + pad8(NU * NU), S_S = S_DG + pad8(NSP * NU);
  static constexpr int S_YS = S_S + pad8(NSP * NSP), S_FLAG = S_YS + pad8(NSP * NSP), LDS_DOUBLES = S_FLAG + 8;
};

// One grid point st < N: everything of its vector recursion that does not depend on the chain.  ps_next = the scan's value
// record of grid point st + 1 (P+ first).  Returns RTOC_STAT_* bits (thread 0's return value counts).
template <int NV, int NU, int NS, int NT>
RTOC_SCAN_DEV unsigned sto_prep_body(const rtoc_grid g, const double* kr, const double* ps_next, double* out, double* smem, int tid) {
  using C = StoPrepCfg<NV, NU, NS>;
  using W = StoScratch<NV, NU, NS>;
  constexpr int NX = 2 * NV, NSP = C::NSP;
  constexpr rtoc_layout SL = ScanLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout KL = SL.kkt;
  const bool impact = g.type == RTOC_GRID_IMPACT;
  const int ns = impact ? 0 : g.dims;
  double* sP = smem + C::S_P;
  const double* A = kr + KL.off[RTOC_KKT_FXX];
  const double* Fx = kr + KL.off[RTOC_KKT_FX];
  const double* fx = kr + KL.off[RTOC_KKT_FFX];
  for (int e = tid; e < NX * NX; e += NT) {
    sP[e] = ps_next[e];
    out[W::O_AT + e] = A[e];   // column j of A = row j of A^T, contiguous: the lanes that share a row read neighbouring words
  }
  RTOC_SCAN_SYNC();
  for (int e = tid; e < NX; e += NT) {
    double a = 0.0, f = 0.0;
    for (int c = 0; c < NX; ++c) {
      const double p = sP[e + c * NX];
      a += p * Fx[c];
      f += p * fx[c];
    }
    out[W::O_PFX + e] = a;
    out[W::O_PFF + e] = impact ? 0.0 : f;
    out[W::O_FX + e] = Fx[e];
    out[W::O_FFX + e] = impact ? 0.0 : fx[e];
    out[W::O_LX + e] = kr[KL.off[RTOC_KKT_LX] + e];
    out[W::O_HX + e] = impact ? 0.0 : kr[KL.off[RTOC_KKT_HX] + e];
  }
  if (impact) return 0u;
  const double* Bv = kr + KL.off[RTOC_KKT_FVU];   // NV x NU
  for (int e = tid; e < NV * NU; e += NT) out[W::O_BV + e] = Bv[e];
  for (int e = tid; e < NU; e += NT) out[W::O_LU + e] = kr[KL.off[RTOC_KKT_LU] + e], out[W::O_HU + e] = kr[KL.off[RTOC_KKT_HU] + e];
  for (int e = tid; e < NSP; e += NT)
    out[W::O_PRES + e] = e < ns ? kr[KL.off[RTOC_KKT_PRES] + e] : 0.0, out[W::O_PHIT + e] = e < ns ? kr[KL.off[RTOC_KKT_PHIT] + e] : 0.0;
  for (int e = tid; e < 8; e += NT) out[W::O_SCAL + e] = e < 3 ? kr[KL.off[RTOC_KKT_SCAL] + e] : 0.0;   // [Qtt, Qtt_prev, h, 0 ...]
  double* PB = smem + C::S_PB;
  double* G = smem + C::S_G;
  double* Y = smem + C::S_Y;
  for (int e = tid; e < NV * NU; e += NT) {   // PB = P+[v, v] Bv
    const int r = e % NV, u = e / NV;
    double acc = 0.0;
    for (int c = 0; c < NV; ++c) acc += sP[(NV + r) + (NV + c) * NX] * Bv[c + u * NV];
    PB[e] = acc;
  }
  RTOC_SCAN_SYNC();
  for (int e = tid; e < NU * NU; e += NT) {   // G = Quu + Bv^T PB
    const int r = e % NU, c = e / NU;
    double acc = kr[KL.off[RTOC_KKT_QUU] + e];
    for (int k = 0; k < NV; ++k) acc += Bv[k + r * NV] * PB[k + c * NV];
    G[e] = acc;
    Y[e] = r == c ? 1.0 : 0.0;
  }
  if (tid == 0) smem[C::S_FLAG] = 0.0;
  RTOC_SCAN_SYNC();
  // L L^T = G in place (lower), right-looking, one barrier pair per column
  for (int k = 0; k < NU; ++k) {
    if (tid == 0) {
      const double d = G[k + k * NU];
      if (!(d > 0.0)) smem[C::S_FLAG] = 1.0;
      G[k + k * NU] = sqrt(d > 0.0 ? d : 1.0);
    }
    RTOC_SCAN_SYNC();
    const double piv = G[k + k * NU];
    for (int r = k + 1 + tid; r < NU; r += NT) G[r + k * NU] /= piv;
    RTOC_SCAN_SYNC();
    for (int e = tid; e < (NU - k - 1) * (NU - k - 1); e += NT) {
      const int r = k + 1 + e % (NU - k - 1), c = k + 1 + e / (NU - k - 1);
      if (r >= c) G[r + c * NU] -= G[r + k * NU] * G[c + k * NU];
    }
    RTOC_SCAN_SYNC();
  }
  // Y = L^-1: column c by forward substitution on e_c (columns are independent)
  for (int c = tid; c < NU; c += NT) {
    for (int r = c; r < NU; ++r) {
      double acc = r == c ? 1.0 : 0.0;
      for (int k = c; k < r; ++k) acc -= G[r + k * NU] * Y[k + c * NU];
      Y[r + c * NU] = acc / G[r + r * NU];
    }
  }
  RTOC_SCAN_SYNC();
  unsigned stat = smem[C::S_FLAG] != 0.0 ? RTOC_STAT_QUU_NOT_SPD : 0u;
  if (ns == 0) {   // Ginv = Y^T Y
    for (int e = tid; e < NU * NU; e += NT) {
      const int r = e % NU, c = e / NU;
      double acc = 0.0;
      for (int k = (r > c ? r : c); k < NU; ++k) acc += Y[k + r * NU] * Y[k + c * NU];
      out[W::O_GI + e] = acc;
    }
    return stat;
  }
  if (NS > 0) {
    double* Gi = smem + C::S_GI;
    double* DG = smem + C::S_DG;
    double* S = smem + C::S_S;
    double* Ys = smem + C::S_YS;
    const double* Phiu = kr + KL.off[RTOC_KKT_PHIU];   // ns x NU, ld NS
    for (int e = tid; e < NU * NU; e += NT) {   // Ginv = Y^T Y  (= llt.solve(I), :60)
      const int r = e % NU, c = e / NU;
      double acc = 0.0;
      for (int k = (r > c ? r : c); k < NU; ++k) acc += Y[k + r * NU] * Y[k + c * NU];
      Gi[e] = acc;
    }
    RTOC_SCAN_SYNC();
    for (int e = tid; e < ns * NU; e += NT) {   // DGinv = Phiu Ginv (:61)
      const int l = e % ns, c = e / ns;
      double acc = 0.0;
      for (int k = 0; k < NU; ++k) acc += Phiu[l + k * NS] * Gi[k + c * NU];
      DG[l + c * NSP] = acc;
    }
    RTOC_SCAN_SYNC();
    for (int e = tid; e < ns * ns; e += NT) {   // S = DGinv Phiu^T (:62)
      const int r = e % ns, c = e / ns;
      double acc = 0.0;
      for (int k = 0; k < NU; ++k) acc += DG[r + k * NSP] * Phiu[c + k * NS];
      S[r + c * NSP] = acc;
      Ys[r + c * NSP] = 0.0;
    }
    if (tid == 0) smem[C::S_FLAG] = 0.0;
    RTOC_SCAN_SYNC();
    for (int k = 0; k < ns; ++k) {   // Ls Ls^T = S (:63)
      if (tid == 0) {
        const double d = S[k + k * NSP];
        if (!(d > 0.0)) smem[C::S_FLAG] = 1.0;
        S[k + k * NSP] = sqrt(d > 0.0 ? d : 1.0);
      }
      RTOC_SCAN_SYNC();
      const double piv = S[k + k * NSP];
      for (int r = k + 1 + tid; r < ns; r += NT) S[r + k * NSP] /= piv;
      RTOC_SCAN_SYNC();
      for (int e = tid; e < (ns - k - 1) * (ns - k - 1); e += NT) {
        const int r = k + 1 + e % (ns - k - 1), c = k + 1 + e / (ns - k - 1);
        if (r >= c) S[r + c * NSP] -= S[r + k * NSP] * S[c + k * NSP];
      }
      RTOC_SCAN_SYNC();
    }
    for (int c = tid; c < ns; c += NT)   // Ys = Ls^-1
      for (int r = c; r < ns; ++r) {
        double acc = r == c ? 1.0 : 0.0;
        for (int k = c; k < r; ++k) acc -= S[r + k * NSP] * Ys[k + c * NSP];
        Ys[r + c * NSP] = acc / S[r + r * NSP];
      }
    RTOC_SCAN_SYNC();
    if (smem[C::S_FLAG] != 0.0) stat |= RTOC_STAT_S_NOT_SPD;
    for (int e = tid; e < ns * ns; e += NT) {   // Sinv = Ys^T Ys, into S's upper-left (S is dead below)
      const int r = e % ns, c = e / ns;
      double acc = 0.0;
      for (int k = (r > c ? r : c); k < ns; ++k) acc += Ys[k + r * NSP] * Ys[k + c * NSP];
      out[W::O_SIN + r + c * NSP] = acc;
    }
    RTOC_SCAN_SYNC();
    // SinvDGinv = Sinv DGinv (:65) ; Ginv' = Ginv - SinvDGinv^T DGinv (:66)
    for (int e = tid; e < ns * NU; e += NT) {
      const int l = e % ns, c = e / ns;
      double acc = 0.0;
      for (int k = 0; k < ns; ++k) acc += out[W::O_SIN + l + k * NSP] * DG[k + c * NSP];
      out[W::O_SDG + l + c * NSP] = acc;
    }
    RTOC_SCAN_SYNC();
    for (int e = tid; e < NU * NU; e += NT) {
      const int r = e % NU, c = e / NU;
      double acc = Gi[e];
      for (int k = 0; k < ns; ++k) acc -= out[W::O_SDG + k + r * NSP] * DG[k + c * NSP];
      out[W::O_GI + e] = acc;
    }
  }
  return stat;
}

// The barrier of the vector pass orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, for which hipcc drains
// vmcnt(0): the bundle prefetched for the next grid point and the stores of the previous one would be waited for at the first
// barrier of every step -- a full HBM / L2 round trip on the chain per grid point.
#if defined(__HIPCC__)
#define RTOC_STO_SYNC()                                         \
  do {                                                          \
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    if (NT > 64) __builtin_amdgcn_s_barrier();                  \
    __asm__ volatile("" ::: "memory");                          \
  } while (0)
#else
#define RTOC_STO_SYNC() \
  do {                  \
  } while (0)
#endif
// s_memtime stamps of one step of the chain (make PROF=1, tools/phase_profile_sto.py): prof[st * 32 + k]
#if defined(__HIPCC__) && defined(RTOC_ENABLE_PROF)
#define RTOC_STO_STAMP_L(k, lane)                                                                \
  do {                                                                                           \
    if (prof && tid == (lane)) prof[st * 32 + (k)] = (long long)__builtin_readcyclecounter();    \
  } while (0)
#else
#define RTOC_STO_STAMP_L(k, lane) \
  do {                            \
  } while (0)
#endif
#define RTOC_STO_STAMP(k) RTOC_STO_STAMP_L(k, 0)

// ---- the serial vector pass ----
// LDS: the bundle of the current grid point (StoScratch block, K^T, M), the chain's vectors, the grid.
template <int NV, int NU, int NS>
struct StoVecCfg {
  using W = StoScratch<NV, NU, NS>;
  static constexpr int NX = 2 * NV, NSP = NS > 0 ? NS : 1, VX = pad8(NX), VU = pad8(NU), VS = pad8(NSP), CHUNK = W::CHUNK;
  static constexpr int B_KT = W::STRIDE, B_M = B_KT + (NX * NU + CHUNK - 1) / CHUNK * CHUNK, B_END = B_M + (NSP * NX + CHUNK - 1) / CHUNK * CHUNK;
  static constexpr int V_SN = B_END, V_PSIN = V_SN + VX, V_PHIN = V_PSIN + VX, V_WZ = V_PHIN + VX, V_WY = V_WZ + VX, V_WP = V_WY + VX;
  static constexpr int V_LU = V_WP + VX, V_PSU = V_LU + VU, V_PHU = V_PSU + 2 * VU, V_K = V_PHU + 2 * VU, V_T = V_K + VU, V_W = V_T + VU;
  static constexpr int V_M = V_W + VU, V_MT = V_M + VS, V_MTN = V_MT + VS, V_SCN = V_MTN + VS, V_DOT = V_SCN + 8, V_PSX = V_DOT + 16;
  static constexpr int V_PHX = V_PSX + VX, G_TAB = V_PHX + VX;
  static constexpr int MAX_STAGES = 512, LDS_DOUBLES = G_TAB + MAX_STAGES / 2;   // G_TAB: the grid, one packed int per grid point
  static constexpr int LDS_BYTES = LDS_DOUBLES * (int)sizeof(double);
};

struct alignas(16) sto_d2 {
  double x, y;
};
// LEN2 16-byte pieces by NT threads: ceil(LEN2 / NT) registers each (clamped loads, guarded stores -- no branch around a load: it
// would hide the count of outstanding loads from the compiler, which then drains vmcnt(0) at the next use of any loaded register)
template <int NT, int LEN2>
struct StoSeg {
  static constexpr int CNT = (LEN2 + NT - 1) / NT;
};
template <int NT, int LEN2>
RTOC_SCAN_DEV void sto_seg_load(double* regs, const double* src, int tid) {
  const sto_d2* s2 = reinterpret_cast<const sto_d2*>(src);
#pragma unroll
  for (int q = 0; q < StoSeg<NT, LEN2>::CNT; ++q) {
    const int e = tid + q * NT;
    const sto_d2 v = s2[(LEN2 % NT == 0 || e < LEN2) ? e : LEN2 - 1];
    regs[2 * q] = v.x, regs[2 * q + 1] = v.y;
  }
}
template <int NT, int LEN2>
RTOC_SCAN_DEV void sto_seg_store(const double* regs, double* dst, int tid) {
  sto_d2* d2 = reinterpret_cast<sto_d2*>(dst);
#pragma unroll
  for (int q = 0; q < StoSeg<NT, LEN2>::CNT; ++q) {
    const int e = tid + q * NT;
    sto_d2 v;
    v.x = regs[2 * q], v.y = regs[2 * q + 1];
    if (LEN2 % NT == 0 || e < LEN2) d2[e] = v;
  }
}

// sum over the PARTS neighbouring lanes that share a row (quad permutes: no LDS, no barrier)
template <int PARTS>
RTOC_SCAN_DEV double sto_parts_sum(double x) {
#if defined(__HIPCC__)
  if (PARTS >= 2) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0xB1, 0xf, 0xf, true);
    x += __hiloint2double(hi, lo);
  }
  if (PARTS >= 4) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x4E, 0xf, 0xf, true);
    x += __hiloint2double(hi, lo);
  }
#endif
  return x;
}

// lanes the widest phase needs when PARTS lanes share a row (row kinds start at wavefront boundaries)
constexpr int sto_lanes_needed(int nx, int nu, int nsp, int parts) {
  const int p1 = ((nx + 2) * parts + 63) / 64 * 64 + nu * parts, p2 = (nu * parts + 63) / 64 * 64 + nsp * parts, p3 = (nx + 5) * parts;
  return p1 > p2 ? (p1 > p3 ? p1 : p3) : (p2 > p3 ? p2 : p3);
}

// The serial vector pass of ONE instance over the whole horizon.  ric: the instance's records (K, M, the terminal s are there:
// matrix half); scr: the instance's bundles.  Returns RTOC_STAT_NAN if k / m went bad.
//
// One step = four barrier-separated phases; in each, a ROW (a dot product of the step) is shared by PARTS neighbouring lanes:
//   P1  rows of A^T (against z, y, Phi+, Psi+), of fx and Fx (the NX-long dot products of the scalars ride along), of Bv^T
//       -> A^T z, A^T y, A^T Phi+, lu', psi_u, phi_u           (z = P+ Fx - s+, y = P+ fx + Psi+ formed on the fly)
//   P2  rows of [Ginv' | SinvDGinv^T] and [Sinv | -SinvDGinv]  -> k, T, W, m, mt, mt_next
//   P3  rows of K^T | M^T, and T, W, k | mt, mt_next, m like rows (the NU- and ns-long dot products of the scalars)  -> s, Psi, Phi
//   P4  lane 0: the scalars; everyone: the next grid point's bundle, registers -> LDS
// The bundle of grid point st - 1 is fetched into registers at the top of step st; the results of step st + 1 are stored to its
// record by the last wave during P1 of step st (from LDS, where they stay untouched until the barrier that ends P1).
template <int NV, int NU, int NS, int NT>
RTOC_SCAN_DEV unsigned sto_vector_body(const rtoc_grid* grid, int nstages, const double* kkt, double* ric, const double* scr, double max_dts0,
                                       double* smem, int tid, long long* prof = nullptr) {
  using C = StoVecCfg<NV, NU, NS>;
  using W = StoScratch<NV, NU, NS>;
  constexpr int NX = 2 * NV, NSP = C::NSP;
  constexpr int R3 = NX + 5;                                           // rows of P3
  constexpr int FL0 = NT == 1 ? 0 : NT - 64, FLN = NT == 1 ? 1 : 64;   // the lanes that store the results of the previous step
  constexpr int MT = NT == 1 ? 1 : (NT > 256 ? 256 : NT);   // the lanes that move the bundles and form the products (StoScratch::CHUNK = 2 * 256)
  // rows of different kinds (different loops) start at a wavefront boundary: a wave that holds both runs both loops in turn
  constexpr int WV = NT == 1 ? 1 : 64;
  constexpr int PARTS = NT == 1 ? 1 : (sto_lanes_needed(NX, NU, NSP, 4) <= MT ? 4 : (sto_lanes_needed(NX, NU, NSP, 2) <= MT ? 2 : 1));
  constexpr int BV0 = ((NX + 2) * PARTS + WV - 1) / WV * WV, P1_END = BV0 + NU * PARTS;   // P1: Bv^T rows behind the A^T, fx, Fx rows
  constexpr int MR0 = (NU * PARTS + WV - 1) / WV * WV, P2_END = MR0 + NSP * PARTS;          // P2: m rows behind the k rows
  static_assert(NT == 1 || (MT >= P1_END && MT >= P2_END && MT >= PARTS * R3 && NT % 64 == 0), "one lane per row part");
  // a lane's share of a row: every PARTS-th entry (the lanes of a row read neighbouring LDS words; a blocked share needs a guard
  // in every iteration and measured 30 % slower)
  constexpr int CHX = (NX + PARTS - 1) / PARTS, CHV = (NV + PARTS - 1) / PARTS, CHU = (NU + PARTS - 1) / PARTS, CHS = (NSP + PARTS - 1) / PARTS;
  constexpr rtoc_layout SL = ScanLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout RL = SL.ric;
  (void)kkt, (void)prof;
  const int N = nstages - 1;
  unsigned stat = 0;
  double* sN = smem + C::V_SN;
  double* psiN = smem + C::V_PSIN;
  double* phiN = smem + C::V_PHIN;
  double* scN = smem + C::V_SCN;
  constexpr int QB = StoSeg<MT, W::STRIDE / 2>::CNT, QK = StoSeg<MT, NV * NU>::CNT, QM = StoSeg<MT, NSP * NV>::CNT;
  double regs[2 * (QB + QK + QM)];   // plain doubles: an array of 16-byte structs stays in scratch memory
  auto fetch = [&](int st) {   // the bundle of grid point st: global -> registers (no waiting here)
    if (tid >= MT) return;
    const double* rr = ric + (size_t)st * RL.stride;
    sto_seg_load<MT, W::STRIDE / 2>(regs, scr + (size_t)st * W::STRIDE, tid);
    sto_seg_load<MT, NV * NU>(regs + 2 * QB, rr + RL.off[RTOC_RIC_K], tid);
    sto_seg_load<MT, NSP * NV>(regs + 2 * (QB + QK), rr + RL.off[RTOC_RIC_M], tid);
  };
  auto drop = [&]() {   // registers -> LDS
    if (tid >= MT) return;
    sto_seg_store<MT, W::STRIDE / 2>(regs, smem, tid);
    sto_seg_store<MT, NV * NU>(regs + 2 * QB, smem + C::B_KT, tid);
    sto_seg_store<MT, NSP * NV>(regs + 2 * (QB + QK), smem + C::B_M, tid);
  };
  // The grid goes to LDS first: a global load issued behind the prefetch can only be waited for together with the prefetch.
  int* gtab = reinterpret_cast<int*>(smem + C::G_TAB);
  for (int e = tid; e < nstages; e += NT) {
    const rtoc_grid g = grid[e];
    gtab[e] = g.type | (g.dims << 4) | ((g.sto != 0) << 12) | ((g.sto_next != 0) << 13);
  }
  RTOC_STO_SYNC();
  if (N >= 1) fetch(N - 1);
  {
    const double* rN = ric + (size_t)N * RL.stride;
    for (int e = tid; e < NX; e += NT) sN[e] = rN[RL.off[RTOC_RIC_S] + e], psiN[e] = 0.0, phiN[e] = 0.0;
    for (int e = tid; e < 8; e += NT) scN[e] = 0.0;
    for (int e = tid; e < 16; e += NT) smem[C::V_DOT + e] = 0.0;
  }
  if (N >= 1) drop();
  RTOC_STO_SYNC();
  // the phase transition (riccati_factorizer.cpp:145-175) in place on the "next" quantities; pol: the record of the STOPolicy
  auto phase_transition = [&](double* pol, bool sto_next) {
    const double xi = scN[0], chi = scN[1], rho = scN[2], eta = scN[3], iota = scN[4];
    double isg = 0.0;
    if (sto_next) {
      double sgm = xi - 2.0 * chi + rho;
      const double eps = 1.4901161193847656e-08;  // sqrt(DBL_EPSILON)
      if ((sgm * max_dts0) < fabs(eta - iota) || sgm < eps) sgm = fabs(sgm) + fabs(eta - iota) / max_dts0;
      isg = 1.0 / sgm;
    }
    RTOC_STO_SYNC();   // every thread has read the scalars
    for (int e = tid; e < NX; e += NT) {
      const double psi = psiN[e], d = psi - phiN[e];
      double phim = psi;   // Phi_m = Psi
      if (sto_next) {
        pol[RL.off[RTOC_RIC_DTSDX] + e] = -isg * d;
        sN[e] += isg * d * (eta - iota);
        phim -= isg * d * (xi - chi);
      }
      psiN[e] = 0.0;
      phiN[e] = phim;
    }
    if (tid == 0) {
      scN[0] = 0.0, scN[1] = 0.0, scN[3] = 0.0;
      if (sto_next) {
        pol[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTSDTS] = isg * (xi - chi);
        pol[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTS0] = -isg * (eta - iota);
        scN[2] = xi - isg * (xi - chi) * (xi - chi);
        scN[4] = eta - isg * (xi - chi) * (eta - iota);
      } else {
        scN[2] = xi;
        scN[4] = eta;
      }
    }
    RTOC_STO_SYNC();
  };
  const double* Am = smem + W::O_AT;   // A column-major: column j = row j of A^T
  const double* Bv = smem + W::O_BV;
  const double* Gi = smem + W::O_GI;
  const double* SDG = smem + W::O_SDG;
  const double* Sin = smem + W::O_SIN;
  const double* PFx = smem + W::O_PFX;
  const double* PFf = smem + W::O_PFF;
  const double* Fx = smem + W::O_FX;
  const double* fx = smem + W::O_FFX;
  const double* lx = smem + W::O_LX;
  const double* hx = smem + W::O_HX;
  const double* Pres = smem + W::O_PRES;
  const double* Phit = smem + W::O_PHIT;
  const double* Kr = smem + C::B_KT;   // K row-major (nu x nx), as the record holds it
  const double* Mm = smem + C::B_M;    // ns x nx, ld NS
  double* wz = smem + C::V_WZ;
  double* wy = smem + C::V_WY;
  double* wp = smem + C::V_WP;
  double* lup = smem + C::V_LU;
  double* kv = smem + C::V_K;
  double* Tv = smem + C::V_T;
  double* Wv = smem + C::V_W;
  double* mv = smem + C::V_M;
  double* mt = smem + C::V_MT;
  double* mtn = smem + C::V_MTN;
  double* dots = smem + C::V_DOT;
  double* psx = smem + C::V_PSX;
  double* phx = smem + C::V_PHX;
  // The results of grid point st go to its record during the NEXT step: vmcnt counts stores too, so stores issued just ahead of
  // drop()'s wait for the prefetched bundle would put a write round trip on the chain.  By the lanes FL0 .. FL0 + FLN - 1.
  auto flush = [&](int st) {
    const int ft = tid - FL0;
    if (ft < 0) return;
    const int g = gtab[st];
    const bool impact = (g & 15) == RTOC_GRID_IMPACT, sto = (g >> 12) & 1;
    const int ns = impact ? 0 : (g >> 4) & 255;
    const double* psu = smem + C::V_PSU + (st & 1) * C::VU;
    const double* phu = smem + C::V_PHU + (st & 1) * C::VU;
    double* rr = ric + (size_t)st * RL.stride;
    for (int e = ft; e < NX; e += FLN) {
      rr[RL.off[RTOC_RIC_S] + e] = sN[e];
      rr[RL.off[RTOC_RIC_PSI] + e] = psiN[e];
      rr[RL.off[RTOC_RIC_PHI] + e] = phiN[e];
      if (sto && !impact) rr[RL.off[RTOC_RIC_PSIX] + e] = psx[e], rr[RL.off[RTOC_RIC_PHIX] + e] = phx[e];
    }
    for (int e = ft; e < 5; e += FLN) rr[RL.off[RTOC_RIC_SCAL] + e] = scN[e];
    if (impact) return;
    for (int u = ft; u < NU; u += FLN) {
      rr[RL.off[RTOC_RIC_KV] + u] = kv[u];
      if (sto) {
        rr[RL.off[RTOC_RIC_PSIU] + u] = psu[u];
        rr[RL.off[RTOC_RIC_PHIU] + u] = phu[u];
        rr[RL.off[RTOC_RIC_T] + u] = Tv[u];
        rr[RL.off[RTOC_RIC_W] + u] = Wv[u];
      }
    }
    if (NS > 0 && ns > 0)
      for (int l = ft; l < ns; l += FLN) {
        rr[RL.off[RTOC_RIC_MV] + l] = mv[l];
        if (sto) rr[RL.off[RTOC_RIC_MT] + l] = mt[l], rr[RL.off[RTOC_RIC_MTN] + l] = mtn[l];
      }
  };
  for (int st = N - 1; st >= 0; --st) {
    const int g = gtab[st];
    const bool impact = (g & 15) == RTOC_GRID_IMPACT, next_lift = (gtab[st + 1] & 15) == RTOC_GRID_LIFT;
    const int ns = impact ? 0 : (g >> 4) & 255;
    const bool sto = (g >> 12) & 1, sto_next = (g >> 13) & 1;
    double* psu = smem + C::V_PSU + (st & 1) * C::VU;   // by grid-point parity: the previous step's are still being stored
    double* phu = smem + C::V_PHU + (st & 1) * C::VU;
    double* rr = ric + (size_t)st * RL.stride;
    (void)ns;
    RTOC_STO_STAMP(0);
    if (st > 0) fetch(st - 1);   // lands while this grid point is being processed
    RTOC_STO_STAMP(1);
    // dispatch of riccati_recursion.cpp:41-70
    const bool pt_here = impact ? (((gtab[st - 1] >> 12) & 1) || sto) : (next_lift && (sto || sto_next));
    if (pt_here) {   // a transition moves s+, Psi+, Phi+, the scalars: the previous grid point's results leave first
      if (st < N - 1) flush(st + 1);
      RTOC_STO_SYNC();
      phase_transition(impact ? rr : rr + RL.stride, sto_next);
    } else if (st < N - 1) {
      flush(st + 1);
    }
    RTOC_STO_STAMP(2);
    RTOC_STO_STAMP_L(10, FL0);
    // ---- P1
    for (int t = tid; t < P1_END; t += NT) {
      const int part = t % PARTS;
      if (t < (NX + 2) * PARTS) {
        const int row = t / PARTS;
        const double* rp = row < NX ? Am + row * NX : (row == NX ? fx : Fx);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int k = 0; k < CHX; ++k) {
          const int i = part + k * PARTS;
          if ((k + 1) * PARTS <= NX || i < NX) {
            const double m = rp[i], ps = psiN[i], ph = phiN[i];
            a0 += m * (PFx[i] - sN[i]);
            a1 += m * (PFf[i] + ps);
            a2 += m * ph;
            a3 += m * ps;
          }
        }
        a0 = sto_parts_sum<PARTS>(a0), a1 = sto_parts_sum<PARTS>(a1), a2 = sto_parts_sum<PARTS>(a2), a3 = sto_parts_sum<PARTS>(a3);
        if (part == 0) {
          if (row < NX) wz[row] = a0, wy[row] = a1, wp[row] = a2;
          else if (row == NX) dots[0] = a0, dots[1] = a1, dots[2] = a2, dots[3] = a3;   // fx.z, fx.y, fx.Phi+, fx.Psi+
          else dots[4] = a2, dots[5] = a3;                                               // Fx.Phi+, Fx.Psi+
        }
      } else if (!impact && t >= BV0) {   // Bv^T of the velocity halves
        const int u = (t - BV0) / PARTS;
        const double* rp = Bv + u * NV;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int k = 0; k < CHV; ++k) {
          const int i = part + k * PARTS;
          if ((k + 1) * PARTS <= NV || i < NV) {
            const double m = rp[i], ps = psiN[NV + i];
            a0 += m * (PFx[NV + i] - sN[NV + i]);
            a1 += m * (PFf[NV + i] + ps);
            a2 += m * phiN[NV + i];
          }
        }
        a0 = sto_parts_sum<PARTS>(a0), a1 = sto_parts_sum<PARTS>(a1), a2 = sto_parts_sum<PARTS>(a2);
        if (part == 0) {
          lup[u] = smem[W::O_LU + u] + a0;
          psu[u] = sto ? smem[W::O_HU + u] + a1 : 0.0;
          phu[u] = (sto && sto_next) ? a2 : 0.0;
        }
      }
    }
    RTOC_STO_STAMP_L(11, FL0);
    RTOC_STO_STAMP_L(12, 0);
    RTOC_STO_STAMP_L(13, 64);
    RTOC_STO_SYNC();
    RTOC_STO_STAMP(3);
    if (impact) {   // riccati_factorizer.cpp:178-197
      for (int e = tid; e < NX; e += NT) {
        const double s = -wz[e] - lx[e];
        sN[e] = s, psiN[e] = 0.0, phiN[e] = sto ? wp[e] : 0.0;
      }
      if (tid == 0) {
        const double rho = sto ? scN[2] : 0.0, iota = sto ? scN[4] + dots[4] : 0.0;
        scN[0] = 0.0, scN[1] = 0.0, scN[2] = rho, scN[3] = 0.0, scN[4] = iota;
      }
      RTOC_STO_SYNC();   // lx has been read
      if (st > 0) drop();
      RTOC_STO_SYNC();
      continue;
    }
    // ---- P2
    for (int t = tid; t < P2_END; t += NT) {
      const int part = t % PARTS;
      if (t < NU * PARTS) {
        const int row = t / PARTS;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int k = 0; k < CHU; ++k) {
          const int c = part + k * PARTS;
          if ((k + 1) * PARTS <= NU || c < NU) {
            const double gv = Gi[c + row * NU];   // Ginv is symmetric
            a0 += gv * lup[c], a1 += gv * psu[c], a2 += gv * phu[c];
          }
        }
        if (NS > 0 && ns > 0) {
#pragma unroll
          for (int k = 0; k < CHS; ++k) {
            const int l = part + k * PARTS;
            if (l < ns) {
              const double sd = SDG[l + row * NSP];
              a0 += sd * Pres[l], a1 += sd * Phit[l];
            }
          }
        }
        a0 = sto_parts_sum<PARTS>(a0), a1 = sto_parts_sum<PARTS>(a1), a2 = sto_parts_sum<PARTS>(a2);
        if (part == 0) {
          kv[row] = -a0, Tv[row] = sto ? -a1 : 0.0, Wv[row] = (sto && sto_next) ? -a2 : 0.0;
          if (!(fabs(a0) <= 1.79769313486231570815e308)) stat |= RTOC_STAT_NAN;
        }
      } else if (NS > 0 && t >= MR0 && (t - MR0) / PARTS < ns) {
        const int l = (t - MR0) / PARTS;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int k = 0; k < CHS; ++k) {
          const int c = part + k * PARTS;
          if (c < ns) {
            const double si = Sin[c + l * NSP];   // Sinv is symmetric
            a0 += si * Pres[c], a1 += si * Phit[c];
          }
        }
#pragma unroll
        for (int k = 0; k < CHU; ++k) {
          const int u = part + k * PARTS;
          if ((k + 1) * PARTS <= NU || u < NU) {
            const double sd = SDG[l + u * NSP];
            a0 -= sd * lup[u], a1 -= sd * psu[u], a2 -= sd * phu[u];
          }
        }
        a0 = sto_parts_sum<PARTS>(a0), a1 = sto_parts_sum<PARTS>(a1), a2 = sto_parts_sum<PARTS>(a2);
        if (part == 0) {
          mv[l] = a0, mt[l] = sto ? a1 : 0.0, mtn[l] = (sto && sto_next) ? a2 : 0.0;
          if (!(fabs(a0) <= 1.79769313486231570815e308)) stat |= RTOC_STAT_NAN;
        }
      }
    }
    RTOC_STO_SYNC();
    RTOC_STO_STAMP(4);
    // ---- P3: s, Psi, Phi of this grid point, straight into the "next" slots (nobody reads s+, Psi+, Phi+ any more).  Five more rows
    // walk T, W, k (and mt, mt_next, m) like rows of K^T (of M^T): the NU- and ns-long dot products of the scalars (brrf.cpp:110-142,
    // riccati_factorizer.cpp:136-141), each row finishing ITS scalar:  fx^T P+ fx + 2 Psi+ . fx = fx . y + fx . Psi+
    for (int t = tid; t < R3 * PARTS; t += NT) {
      const int row = t / PARTS, part = t % PARTS, x = row - NX;
      const double* rk = row < NX ? Kr + row : (x <= 1 ? Tv : (x == 2 ? Wv : kv));
      const int ldk = row < NX ? NX : 1;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, mp = 0.0, mph = 0.0;
#pragma unroll
      for (int k = 0; k < CHU; ++k) {
        const int u = part + k * PARTS;
        if ((k + 1) * PARTS <= NU || u < NU) {
          const double kt = rk[u * ldk];
          a0 += kt * lup[u], a1 += kt * psu[u], a2 += kt * phu[u];
        }
      }
      if (NS > 0 && ns > 0) {
        const double* rm = row < NX ? Mm + row * NSP : (x == 0 ? mt : (x == 1 ? mtn : mv));
#pragma unroll
        for (int k = 0; k < CHS; ++k) {
          const int l = part + k * PARTS;
          if (l < ns) mp += rm[l] * Pres[l], mph += rm[l] * Phit[l];
        }
      }
      a0 = sto_parts_sum<PARTS>(a0), a1 = sto_parts_sum<PARTS>(a1), a2 = sto_parts_sum<PARTS>(a2);
      mp = sto_parts_sum<PARTS>(mp), mph = sto_parts_sum<PARTS>(mph);
      if (part == 0) {
        if (row < NX) {
          const double s = -wz[row] - lx[row] - a0 - mp;
          double psi = 0.0, phi = 0.0;
          if (sto) {
            const double psix = wy[row] + hx[row], phix = sto_next ? wp[row] : 0.0;
            psx[row] = psix, phx[row] = phix;
            psi = psix + a1 + mph;
            phi = sto_next ? phix + a2 : 0.0;
          }
          sN[row] = s, psiN[row] = psi, phiN[row] = phi;
        } else {
          // x:    0 xi                       1 chi                  2 rho      3 eta                       4 iota
          // v = dots[1] + dots[3] + Qtt     dots[2] + Qtt_prev     .          dots[0] + dots[5] + h       dots[4]
          //     + T.psi_u + mt.Phit         + T.phi_u + mtn.Phit   + W.phi_u  + k.psi_u + m.Phit          + k.phi_u      (+ its own last value)
          // one set of loads for all five (dots[15] and the record's scalar 3 are zero)
          const int i0 = x == 0 ? 1 : (x == 1 ? 2 : (x == 2 ? 15 : (x == 3 ? 0 : 4)));
          const int i1 = x == 0 ? 3 : (x == 3 ? 5 : 15);
          const int ik = x == 0 ? RTOC_KKT_SCAL_QTT : (x == 1 ? RTOC_KKT_SCAL_QTT_PREV : (x == 3 ? RTOC_KKT_SCAL_H : 3));
          const double d0 = dots[i0], d1 = dots[i1], kk = smem[W::O_SCAL + ik], prev = scN[x];
          const bool gated = x == 1 || x == 2 || x == 4;   // exist only with sto_next
          double v = d0 + d1 + kk + ((x == 0 || x == 3) ? a1 : a2) + prev + ((x == 2 || x == 4) ? 0.0 : mph);
          if (!sto || (gated && !sto_next)) v = 0.0;
          scN[x] = v;
        }
      }
    }
    RTOC_STO_SYNC();
    RTOC_STO_STAMP(5);
    RTOC_STO_STAMP(6);
    if (st > 0) drop();   // the next grid point's bundle takes the place of this one's
    RTOC_STO_SYNC();
    RTOC_STO_STAMP(7);
  }
  if (N >= 1) flush(0);
  RTOC_STO_SYNC();
  // riccati_recursion.cpp:72-79: the transition ahead of grid point 0 (its STOPolicy goes to record 0; the modified copy is dropped)
  if ((gtab[0] >> 12) & 1) phase_transition(ric, (gtab[0] >> 13) & 1);
  return stat;
}

}  // namespace scan
}  // namespace rtoc
