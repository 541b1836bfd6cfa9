// riccati_scan_core.hpp -- the horizon scan of the backward Riccati recursion: workgroup bodies.
//
// RiccatiRecursion::backwardRiccatiRecursion (reference src/riccati/riccati_recursion.cpp:32-80) is a
// chain over the grid points.  For few OCP instances (one MPC problem, BASELINE configs 2-4) the chain
// leaves the chip idle, so the value functions P_i, s_i of ALL grid points are computed here by an
// associative suffix scan over *interval elements* (A, b, C, eta, J) -- the conditional value function
// of an interval [i, j) of the horizon in its dual form
//
//   V(x_i, x_j) = max_lam  1/2 x_i^T J x_i - eta^T x_i - 1/2 lam^T C lam - lam^T (x_j - A x_i - b)
//
// (Saerkkae & Garcia-Fernandez, "Temporal parallelization of dynamic programming and linear quadratic
// control", IEEE TAC 2023) -- in ceil(log2(#grid points)) combination levels, every level two
// workgroups per grid point.  Then the policies K, k (M, m) of all grid points are computed at once by
// the one-stage mode of riccati_backward_kernel from P_{i+1}, s_{i+1}, i.e. with the reference's own
// per-stage algebra (riccati_factorizer.cpp:44-90).  The forward recursion is a prefix scan of the
// closed-loop maps (end of this file).
//
//   element of an intermediate / lift grid point (brrf.cpp:31-45 with P+ = 0, eliminated control):
//     L L^T = Quu,  Z = L^-1 [Qxu^T | lu | Fvu^T | Phiu^T]  =: [Zs | zl | Zb | Zd]
//     no switching constraint:  Ys = Zs, yl = zl, Zb' = Zb
//     switching constraint Phix dx + Phiu du + P = 0 (riccati_factorizer.cpp:58-77), eliminated inside
//     the element:  Ls Ls^T = Zd^T Zd (= Phiu Quu^-1 Phiu^T, the reference's S),  Qd = Zd Ls^-T,
//       Ys = (I - Qd Qd^T) Zs + Qd Ls^-1 Phix,  yl = (I - Qd Qd^T) zl + Qd Ls^-1 P,  Zb' = (I - Qd Qd^T) Zb
//     A = Fxx - [0; Zb^T Ys],  b = Fx - [0; Zb^T yl],  C = [0 0; 0 Zb'^T Zb'],
//     J = Qxx - Zs^T Ys - Ys^T Zs + Ys^T Ys,  eta = -lx + Zs^T yl + Ys^T zl - Ys^T yl
//   impact grid point (riccati_factorizer.cpp:178-197): A = Fxx, b = Fx, C = 0, J = Qxx, eta = -lx
//   terminal grid point (riccati_recursion.cpp:37-38): closed element J = Qxx, eta = -lx
//
//   combination of [i,j) and [j,k):  M = I + C1 J2,  [Ta | tb | Tc] = M^-1 [A1 | b1 + C1 eta2 | C1]
//     J = J1 + A1^T J2 Ta,  eta = eta1 + A1^T (eta2 - J2 tb),
//     A = A2 Ta,  b = b2 + A2 tb,  C = C2 + A2 Tc A2^T        (skipped when [j,k) contains the terminal)
//
// tests/scan_reference.py states the same formulas in numpy.  Grids with switching-time optimisation
// (sto flags) are not covered: rtoc_riccati_backward falls back to the serial HIP kernel for them.
//
// The bodies are written as barrier-separated phases of "for (i = tid; i < n; i += NT)" loops; the few wave-level
// pieces (DPP / ds_bpermute exchange between the lanes of a column, MFMA products, hardware rcp / rsq) sit behind
// small functions with a plain host version, so that the very same code also compiles for the host with NT = 1
// (tests/cpp/scan_emulation.cpp checks the algebra and the indexing on the CPU against numpy and the oracle).
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/rtoc.h"
#if defined(__HIPCC__)
#include "lds_gemm.hpp"
#endif

// tools/probes/scan_probe.hip compiles parts of combine_body out (timing only)
#ifndef RTOC_SCAN_PROBE
#define RTOC_SCAN_PROBE 0
#endif
#if defined(__HIPCC__)
#define RTOC_SCAN_DEV __device__ __forceinline__
#define RTOC_SCAN_SYNC() __syncthreads()
#else
#define RTOC_SCAN_DEV inline
#define RTOC_SCAN_SYNC() \
  do {                   \
  } while (0)
#endif

namespace rtoc {
namespace scan {

constexpr int pad8(int n) { return (n + 7) & ~7; }

// interval element [A | C | J | b | eta] (matrices NX x NX column-major, ld NX) and the closed value
// record [P | s] of a grid point
template <int NV>
struct EltLayout {
  static constexpr int NX = 2 * NV;
  static constexpr int MAT = pad8(NX * NX);
  static constexpr int VEC = pad8(NX);
  static constexpr int OFF_A = 0, OFF_C = MAT, OFF_J = 2 * MAT, OFF_B = 3 * MAT, OFF_ETA = 3 * MAT + VEC;
  static constexpr int STRIDE = 3 * MAT + 2 * VEC;
  static constexpr int PS_P = 0, PS_S = MAT, PS_STRIDE = MAT + VEC;
};

// Hillis-Steele suffix scan over n grid points (index n-1 = terminal).  Before the level with distance
// d the element of grid point i covers [i, i+d) (closed, i.e. a value record, once i+d >= n).  In that
// level grid point i is combined with j = i+d iff it is still open.  Returns false if i has nothing to
// do; otherwise *j and whether the right operand is closed (then the result is closed, too).
RTOC_SCAN_DEV bool level_plan(int n, int d, int i, int* j, bool* closed2) {
  if (i + d >= n) return false;
  *j = i + d;
  *closed2 = (i + 2 * d >= n);
  return true;
}
inline int num_levels(int n) {
  int lv = 0;
  for (int d = 1; d < n; d *= 2) ++lv;
  return lv;
}

// record offsets as compile-time constants (the kkt / ric records do not depend on np / nc_max)
template <int NV, int NU, int NS>
struct ScanLayout {
  static constexpr rtoc_layout make() {
    rtoc_dims d = {NV, NU, 0, NS, NS, 0};
    rtoc_layout L = {};
    rtoc_compute_layout(&d, &L);
    return L;
  }
};

// C(i,j) = sum_k A(i,k) B(k,j) between LDS-resident operands with compile-time shapes and strides,
// A(i,k) = A[i*ARS + k*ACS], B(k,j) = B[k*BRS + j*BCS]; epilogue(row, col, value).  On the GPU the
// 16x16 tiles are dealt to the NT/64 waves and computed with v_mfma_f64_16x16x4_f64 (lds_gemm.hpp); the
// host emulation runs plain loops.
#if defined(__HIPCC__)
template <int NT, int M, int N, int K, int ARS, int ACS, int BRS, int BCS, class E>
RTOC_SCAN_DEV void scan_gemm(const double* A, const double* B, int tid, E&& epilogue) {
  rtoc::lds_gemm<NT / 64, M, N, K, ARS, ACS, BRS, BCS>(
      A, B, tid, [&](int row, int col, double v, int, int) { epilogue(row, col, v); });
}
#else
template <int NT, int M, int N, int K, int ARS, int ACS, int BRS, int BCS, class E>
RTOC_SCAN_DEV void scan_gemm(const double* A, const double* B, int tid, E&& epilogue) {
  for (int idx = tid; idx < M * N; idx += NT) {
    const int i = idx % M, j = idx / M;
    double acc = 0.0;
    for (int k = 0; k < K; ++k) acc += A[i * ARS + k * ACS] * B[k * BRS + j * BCS];
    epilogue(i, j, acc);
  }
}
#endif

// 1/sqrt(d) to fp64 accuracy: hardware estimate + two Newton steps (no sqrt, no divide)
RTOC_SCAN_DEV double fast_rsqrt(double d) {
#if defined(__HIPCC__)
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  y = y * __builtin_fma(-h * y, y, 1.5);
  y = y * __builtin_fma(-h * y, y, 1.5);
  return y;
#else
  return 1.0 / sqrt(d);
#endif
}

// ---- cooperative Cholesky with ONE barrier per column: the trailing update works on the unscaled column
//      (A[i][k] -= A[i][j] A[k][j] / d), the scaled column goes to a separate output, so that nothing a
//      thread reads in a step is written in the same step.  Lout: lower factor (ld as A), linv: 1/diag. ----
template <int NT>
RTOC_SCAN_DEV void lds_cholesky_1b(double* A, double* Lout, double* linv, int n, int ld, int tid, double* flag) {
  for (int j = 0; j < n; ++j) {
    double d = A[j + j * ld];
    if (!(d > 0.0)) {
      if (tid == 0) *flag = 1.0;
      d = 1.0;
    }
    const double rs = fast_rsqrt(d), id = rs * rs;
    for (int i = j + tid; i < n; i += NT) Lout[i + j * ld] = A[i + j * ld] * rs;
    if (tid == 0) linv[j] = rs;
    const int m = n - j - 1;
    for (int idx = tid; idx < m * m; idx += NT) {
      const int ii = idx % m, kk = idx / m;
      if (ii >= kk) {
        const int i = j + 1 + ii, k = j + 1 + kk;
        A[i + k * ld] -= A[i + j * ld] * A[k + j * ld] * id;
      }
    }
    RTOC_SCAN_SYNC();
  }
}

// x <- L^-1 x for one column, solved in registers (N compile-time: no dependent LDS round trips)
template <int N>
RTOC_SCAN_DEV void fwd_subst_reg(const double* L, const double* linv, int ld, double* xlds) {
  double x[N];
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] = xlds[k];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double acc = x[k];
#pragma unroll
    for (int m = 0; m < k; ++m) acc -= L[k + m * ld] * x[m];
    x[k] = acc * linv[k];
  }
#pragma unroll
  for (int k = 0; k < N; ++k) xlds[k] = x[k];
}

template <int NV, int NU, int NS>
struct ElementCfg {
  static constexpr int NX = 2 * NV;
  static constexpr int NSP = NS > 0 ? NS : 1;
  static constexpr int LDZ = NU | 1;   // odd leading dimensions: conflict-free column walks
  static constexpr int LDS_ = NSP | 1;
  static constexpr int C_S = 0, C_L = NX, C_B = NX + 1, C_D = NX + 1 + NV;  // columns of Z
  static constexpr int NCOL = C_D;                                          // [Zs | zl | Zb]
  static constexpr int NZ = NX + 1 + NV + NS;
  static constexpr int OFF_G = 0;                                 // Quu, consumed by the factorisation
  static constexpr int OFF_L = OFF_G + pad8(LDZ * NU);            // L         NU x NU
  static constexpr int OFF_Z = OFF_L + pad8(LDZ * NU);            // Z         NU x NZ
  static constexpr int OFF_T = OFF_Z + pad8(LDZ * NZ);            // [T|tl|Tb] NS x NCOL
  static constexpr int OFF_QD = OFF_T + pad8(LDS_ * NCOL);        // Qd        NU x NS
  static constexpr int OFF_SC = OFF_QD + pad8(LDZ * NSP);         // S = Zd^T Zd, consumed
  static constexpr int OFF_LS = OFF_SC + pad8(LDS_ * NSP);        // Ls        NS x NS
  static constexpr int OFF_RX = OFF_LS + pad8(LDS_ * NSP);        // Ls^-1 [Phix | P]  NS x (NX+1)
  static constexpr int OFF_LINV = OFF_RX + pad8(LDS_ * (NX + 1)); // 1/diag(L), 1/diag(Ls)
  static constexpr int OFF_FLAG = OFF_LINV + pad8(NU) + pad8(NSP);
  static constexpr int LDG = NCOL | 1;
  static constexpr int OFF_GRAM = OFF_FLAG + 8;                   // G1 - G2, NCOL x NCOL
  static constexpr int LDS_DOUBLES = OFF_GRAM + pad8(LDG * NCOL);
  static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
  static_assert(LDS_BYTES <= 160 * 1024, "element scratch must fit the LDS of a CU");
};

// One grid point -> its element.  kr: the KKT record; elt: the element record (open grid points);
// ps: the closed value record (terminal only).  Returns RTOC_STAT_* bits.
//
// With Zall = [Zs | zl | Zb] = L^-1 [Qxu^T | lu | Fvu^T] and, on a switching-constraint grid point,
// Tall = [T | tl | Tb] = [Ls^-1 Phix | Ls^-1 P | 0] - Qd^T Zall  (Ys = Zs + Qd T, yl = zl + Qd tl, Zb' = Zb + Qd Tb
// in the notation of the file header; Qd has orthonormal columns) every block of the element is a block of
// the two Gram matrices G1 = Zall^T Zall, G2 = Tall^T Tall:
//   J = Qxx - (G1 - G2)[s,s]   eta = -lx + (G1 - G2)[l,s]   A[v,:] = Fxx[v,:] - (G1 - G2)[b,s]
//   b[v] = Fx[v] - (G1 - G2)[l,b]   C[v,v] = (G1 - G2)[b,b]
// computed on the matrix cores; the blocks are picked so that every store runs along a column of the output.
template <int NV, int NU, int NS, int NT>
RTOC_SCAN_DEV unsigned element_body(const rtoc_grid& g, const double* kr, double* elt, double* ps,
                                    double* smem, int tid) {
  constexpr rtoc_layout SL = ScanLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout KL = SL.kkt;
  using C = ElementCfg<NV, NU, NS>;
  using E = EltLayout<NV>;
  constexpr int NX = C::NX, LDZ = C::LDZ, LDSS = C::LDS_, NCOL = C::NCOL;
  const double* Qxx = kr + KL.off[RTOC_KKT_QXX];
  const double* lx = kr + KL.off[RTOC_KKT_LX];
  if (g.type == RTOC_GRID_TERMINAL) {
    for (int i = tid; i < NX * NX; i += NT) ps[E::PS_P + i] = Qxx[i];
    for (int i = tid; i < NX; i += NT) ps[E::PS_S + i] = -lx[i];
    return 0;
  }
  const double* Fxx = kr + KL.off[RTOC_KKT_FXX];
  const double* Fx = kr + KL.off[RTOC_KKT_FX];
  if (g.type == RTOC_GRID_IMPACT) {
    for (int i = tid; i < NX * NX; i += NT) {
      elt[E::OFF_A + i] = Fxx[i];
      elt[E::OFF_C + i] = 0.0;
      elt[E::OFF_J + i] = Qxx[i];
    }
    for (int i = tid; i < NX; i += NT) {
      elt[E::OFF_B + i] = Fx[i];
      elt[E::OFF_ETA + i] = -lx[i];
    }
    return 0;
  }
  const double* Qxu = kr + KL.off[RTOC_KKT_QXU];
  const double* Quu = kr + KL.off[RTOC_KKT_QUU];
  const double* Fvu = kr + KL.off[RTOC_KKT_FVU];
  const double* lu = kr + KL.off[RTOC_KKT_LU];
  const double* Phix = kr + KL.off[RTOC_KKT_PHIX];
  const double* Phiu = kr + KL.off[RTOC_KKT_PHIU];
  const double* Pres = kr + KL.off[RTOC_KKT_PRES];
  const int ns = NS > 0 ? g.dims : 0;
  constexpr int ldn = NS;  // leading dimension of Phix / Phiu (ns_max)
  double* sG = smem + C::OFF_G;
  double* sL = smem + C::OFF_L;
  double* sZ = smem + C::OFF_Z;
  double* sT = smem + C::OFF_T;
  double* sQd = smem + C::OFF_QD;
  double* sSc = smem + C::OFF_SC;
  double* sLs = smem + C::OFF_LS;
  double* sRx = smem + C::OFF_RX;
  double* linv = smem + C::OFF_LINV;
  double* linvs = linv + pad8(NU);
  double* flag = smem + C::OFF_FLAG;
  if (tid == 0) {
    flag[0] = 0.0;
    flag[1] = 0.0;
  }
  // ---- plain parts of the element: rows q of A and b, the zero blocks of C ----
  for (int idx = tid; idx < NX * NX; idx += NT) {
    const int r = idx % NX, c = idx / NX;
    if (r < NV) elt[E::OFF_A + idx] = Fxx[idx];
    if (r < NV || c < NV) elt[E::OFF_C + idx] = 0.0;
  }
  for (int r = tid; r < NV; r += NT) elt[E::OFF_B + r] = Fx[r];
  // ---- Quu -> sG ; [Qxu^T | lu | Fvu^T | Phiu^T] -> sZ ----
  for (int i = tid; i < NU * NU; i += NT) sG[(i % NU) + (i / NU) * LDZ] = Quu[i];
  for (int i = tid; i < NX * NU; i += NT) {  // Qxu(j,k), j = i % NX fastest (coalesced)
    const int j = i % NX, k = i / NX;
    sZ[k + (C::C_S + j) * LDZ] = Qxu[i];
  }
  for (int k = tid; k < NU; k += NT) sZ[k + C::C_L * LDZ] = lu[k];
  for (int i = tid; i < NV * NU; i += NT) {  // Fvu(j,k)
    const int j = i % NV, k = i / NV;
    sZ[k + (C::C_B + j) * LDZ] = Fvu[i];
  }
  const int nsd = ns > 0 ? ns : 1;
  for (int i = tid; i < ns * NU; i += NT) {  // Phiu(j,k), ld ns_max
    const int j = i % nsd, k = i / nsd;
    sZ[k + (C::C_D + j) * LDZ] = Phiu[j + k * ldn];
  }
  RTOC_SCAN_SYNC();
  lds_cholesky_1b<NT>(sG, sL, linv, NU, LDZ, tid, flag);
  // ---- Z <- L^-1 Z, one column per thread, in registers ----
  {
    const int ncol = C::C_D + ns;
    for (int c = tid; c < ncol; c += NT) fwd_subst_reg<NU>(sL, linv, LDZ, sZ + c * LDZ);
  }
  RTOC_SCAN_SYNC();
  if (ns > 0) {
    const double* Zd = sZ + C::C_D * LDZ;
    for (int idx = tid; idx < ns * ns; idx += NT) {
      const int i = idx % ns, j = idx / ns;
      double acc = 0.0;
      for (int k = 0; k < NU; ++k) acc += Zd[k + i * LDZ] * Zd[k + j * LDZ];
      sSc[i + j * LDSS] = acc;
    }
    RTOC_SCAN_SYNC();
    lds_cholesky_1b<NT>(sSc, sLs, linvs, ns, LDSS, tid, flag + 1);
    // Qd = Zd Ls^-T (row k of Zd per thread); Rx = Ls^-1 [Phix | P] (one column per thread)
    for (int w = tid; w < NU + NX + 1; w += NT) {
      if (w < NU) {
        const int k = w;
        for (int j = 0; j < ns; ++j) {
          double acc = Zd[k + j * LDZ];
          for (int m = 0; m < j; ++m) acc -= sQd[k + m * LDZ] * sLs[j + m * LDSS];
          sQd[k + j * LDZ] = acc * linvs[j];
        }
      } else {
        const int c = w - NU;
        double* x = sRx + c * LDSS;
        for (int j = 0; j < ns; ++j) {
          double acc = (c < NX) ? Phix[j + c * ldn] : Pres[j];
          for (int m = 0; m < j; ++m) acc -= sLs[j + m * LDSS] * x[m];
          x[j] = acc * linvs[j];
        }
      }
    }
    RTOC_SCAN_SYNC();
    // Tall = [Rx | 0] - Qd^T Zall, rows >= ns zero (the Gram product runs over all NS rows)
    for (int c = tid; c < NCOL; c += NT) {
      const double* z = sZ + c * LDZ;
      for (int j = 0; j < NS; ++j) {
        double acc = 0.0;
        if (j < ns) {
          acc = (c <= NX) ? sRx[j + c * LDSS] : 0.0;
          for (int k = 0; k < NU; ++k) acc -= sQd[k + j * LDZ] * z[k];
        }
        sT[j + c * LDSS] = acc;
      }
    }
    RTOC_SCAN_SYNC();
  }
  // ---- G = G1 - G2 -> LDS (symmetric), then the element in one pass of independent, coalesced HBM accesses
  //      (an epilogue that fetched its Qxx / Fxx entries tile by tile paid one HBM round trip per tile) ----
  double* sGr = smem + C::OFF_GRAM;
  constexpr int LDG = C::LDG;
  scan_gemm<NT, NCOL, NCOL, NU, LDZ, 1, 1, LDZ>(sZ, sZ, tid,
                                                [&](int row, int col, double v) { sGr[row + col * LDG] = v; });
  if (NS > 0 && ns > 0)
    scan_gemm<NT, NCOL, NCOL, (NS > 0 ? NS : 1), LDSS, 1, 1, LDSS>(
        sT, sT, tid, [&](int row, int col, double v) { sGr[row + col * LDG] -= v; });  // same lane as above
  RTOC_SCAN_SYNC();
  for (int idx = tid; idx < NX * NX; idx += NT) {
    const int r = idx % NX, c = idx / NX;
    elt[E::OFF_J + idx] = Qxx[idx] - sGr[r + c * LDG];
    if (r >= NV) {
      elt[E::OFF_A + idx] = Fxx[idx] - sGr[(NX + 1 + r - NV) + c * LDG];
      if (c >= NV) elt[E::OFF_C + idx] = sGr[(NX + 1 + r - NV) + (NX + 1 + c - NV) * LDG];
    }
  }
  for (int r = tid; r < NX; r += NT) {
    elt[E::OFF_ETA + r] = -lx[r] + sGr[r + NX * LDG];
    if (r >= NV) elt[E::OFF_B + r] = Fx[r] - sGr[(NX + 1 + r - NV) + NX * LDG];
  }
  unsigned stat = 0;
  if (flag[0] != 0.0) stat |= RTOC_STAT_QUU_NOT_SPD;
  if (flag[1] != 0.0) stat |= RTOC_STAT_S_NOT_SPD;
  return stat;
}

// ---- combination ---------------------------------------------------------------------------------
// exchange between the LPC (<= 8) adjacent lanes that share one column of the elimination (DPP)
#if defined(__HIPCC__)
template <int CTRL>
RTOC_SCAN_DEV int quad_perm_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
RTOC_SCAN_DEV double quad_perm_d(double v) {
  const int lo = quad_perm_i<CTRL>(__double2loint(v));
  const int hi = quad_perm_i<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
#else
template <int CTRL>
RTOC_SCAN_DEV int quad_perm_i(int v) { return v; }
template <int CTRL>
RTOC_SCAN_DEV double quad_perm_d(double v) { return v; }
#endif
// 1/x to fp64 accuracy: hardware estimate + two Newton steps (off the divide's long dependent chain)
RTOC_SCAN_DEV double fast_rcp(double x) {
#if defined(__HIPCC__)
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}
// |v| for the pivot search; a used ("dead") row gets its sign and exponent cleared, i.e. at most a
// denormal that loses against every real candidate (one v_cndmask on the high dword)
RTOC_SCAN_DEV double masked_abs(double v, bool dead) {
#if defined(__HIPCC__)
  const int hi = dead ? 0 : (__double2hiint(v) & 0x7fffffff);
  return __hiloint2double(hi, __double2loint(v));
#else
  return dead ? 0.0 : fabs(v);
#endif
}
// A column slice in registers.  On the GPU a clang vector: indexing it with a wave-uniform (scalar-register)
// index is a register-relative move (s_set_gpr_idx), not a chain of selects.
#if defined(__HIPCC__)
template <int N>
struct ColVec {
  typedef double type __attribute__((ext_vector_type(N)));
};
RTOC_SCAN_DEV int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }
#else
template <int N>
struct ColVec {
  struct type {
    double v[N];
    double& operator[](int i) { return v[i]; }
    const double& operator[](int i) const { return v[i]; }
  };
};
RTOC_SCAN_DEV int uniform_int(int v) { return v; }
#endif
// value of lane `src_h` of every group of LPC adjacent lanes, for all lanes of the group (LDS crossbar, no LDS memory)
template <int LPC>
RTOC_SCAN_DEV double group_bcast_d(double v, int lane, int src_h) {
#if defined(__HIPCC__)
  if (LPC == 1) return v;
  const int addr = ((lane & ~(LPC - 1)) | src_h) << 2;
  const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
#else
  return v;
#endif
}
constexpr int QUAD_XOR1 = 0xB1;  // quad_perm [1,0,3,2]
constexpr int QUAD_XOR2 = 0x4E;  // quad_perm [2,3,0,1]
constexpr int HALF_MIRROR = 0x141;  // row_half_mirror: lane i <-> 7 - i of every 8 lanes (the other quad)

constexpr int scan_lds_ld(int n) { return (n % 4 == 2) ? n : n + 2; }  // as lds_ld (device_utils.hpp)

template <int NV, int NT>
struct CombineCfg {
  static constexpr int NX = 2 * NV;
  static constexpr int LDW = 2 * NX + 1;  // columns of one workgroup's tableau: [M | A1 | t] or [M | C1]
  // lanes per column (up to 8 adjacent lanes), rows per lane, column slots per thread.
  // Big robots take two column slots per thread rather than fewer lanes per column: the pivot columns (< NX)
  // all sit in slot 0, and the rows per lane set the length of the serial search / elimination chain.
#ifdef RTOC_SCAN_FORCE_LPC
  static constexpr int LPC = RTOC_SCAN_FORCE_LPC;  // tuning probes only
#else
  static constexpr int LPC = (2 * NT >= 8 * LDW) ? 8 : (2 * NT >= 4 * LDW) ? 4 : (2 * NT >= 2 * LDW) ? 2 : 1;
#endif
  static constexpr int RPL = (NX + LPC - 1) / LPC;
  static constexpr int GPS = NT / LPC;                    // column groups per slot
  static constexpr int CPT = (LDW + GPS - 1) / GPS;       // column slots per thread: 1 or 2 on the GPU
  static_assert(NT < 64 || GPS >= NX, "the pivot columns must sit in slot 0");
  // Two slots per thread (16 waves, all busy): the step is bound by VALU issue -> the owner publishes 1/pivot and
  // the pivot-row entry travels by ds_bpermute (fewest instructions).  One slot (ANYmal, 10 waves): the step is
  // bound by its dependent chain -> every thread takes the reciprocal itself and the entry travels by DPP adds
  // (shortest latency).  Measured both ways on both robots (tools/probes/scan_probe.hip).
  static constexpr bool PUBLISH_RCP = CPT > 1;
  static constexpr int LDM = scan_lds_ld(NX);            // column-major NX x NX operands
  static constexpr int LDT = NX | 1;                     // row-major Ta / Tc
  static constexpr int REG = pad8(NX * (LDM > LDT ? LDM : LDT));
  // small robots: A1 and A2 get LDS regions of their own and are fetched up front, under the M product and
  // the elimination, instead of taking over R1 in the middle of the kernel
  static constexpr bool PRE = (5 * REG + 16 * pad8(NX + 8)) * 8 <= 150 * 1024;
  static constexpr int NREG = PRE ? 5 : 3;
  static constexpr int OFF_R0 = 0, OFF_R1 = REG, OFF_R2 = 2 * REG, OFF_R3 = PRE ? 3 * REG : REG,
                       OFF_R4 = PRE ? 4 * REG : REG;
  static constexpr int MPAD = pad8(LPC * RPL);              // rows incl. the padding of the last lane
  static constexpr int OFF_MULT = NREG * REG;               // published pivot column, double-buffered
  static constexpr int OFF_VT = OFF_MULT + 2 * MPAD;        // t, then tb
  static constexpr int OFF_VW = OFF_VT + pad8(NX);          // eta2 - J2 tb
  static constexpr int OFF_ETA2 = OFF_VW + pad8(NX);
  static constexpr int OFF_PIV = OFF_ETA2 + pad8(NX);       // per step: pivot row, 1/pivot (double-buffered)
  static constexpr int OFF_KOF = OFF_PIV + 8;               // kof[r] = elimination step whose pivot row is r
  static constexpr int OFF_FLAG = OFF_KOF + MPAD;
  static constexpr int LDS_DOUBLES = OFF_FLAG + 8;
  static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
  static_assert(LPC * RPL >= NX && (64 % LPC) == 0, "row split");
};

// column handled by slot s of a thread: slot 0 in group order, the further slots in reverse group order so that
// the waves that own the pivot columns carry none of them
template <class C>
RTOC_SCAN_DEV int slot_col(int tid, int s) {
  const int q = tid / C::LPC;
  return s == 0 ? q : s * C::GPS + (C::GPS - 1 - q);
}

// column-major NX x NX matrix, HBM (ld NX) -> LDS (ld LDM)
template <int NX, int LDM, int NT>
RTOC_SCAN_DEV void load_mat(double* dst, const double* src, int tid) {
  for (int e = tid; e < NX * NX; e += NT) dst[(e % NX) + (e / NX) * LDM] = src[e];
}

// e1 = element of [i, j), (J2, eta2, A2, b2, C2) = element of [j, k).  closed2: [j,k) contains the
// terminal grid point, then J2 / eta2 are its value record, A2 / b2 / C2 are not read and the result
// is the closed value record `ps_out`; otherwise the result is the element `out`.
// The work of one combination is split over TWO workgroups that share nothing but their inputs (the
// elimination is bound by the VALU throughput of a CU, and the pivots depend on M alone, so both run it on M):
//   part 0: [M | A1 | t]  ->  Ta, tb  ->  J, eta (and A, b)        part 1 (open right operand only): [M | C1] -> Tc -> C
// Three NX x NX LDS regions are time-shared:  R0: C1 -> Ta (Tc)   R1: J2 -> A1 -> A2   R2: M -> U (V)
// (CombineCfg::PRE: A1 and A2 live in R3 / R4 from the start)
template <int NV, int NT>
RTOC_SCAN_DEV unsigned combine_body(const double* e1, const double* J2, const double* eta2,
                                    const double* A2, const double* b2, const double* C2, bool closed2, int part,
                                    double* out, double* ps_out, double* smem, int tid) {
  if (part == 1 && closed2) return 0;
  using C = CombineCfg<NV, NT>;
  using E = EltLayout<NV>;
  constexpr int NX = C::NX, LDW = C::LDW, CPT = C::CPT, LPC = C::LPC, RPL = C::RPL, LDM = C::LDM,
                LDT = C::LDT;
  const double* A1 = e1 + E::OFF_A;
  const double* C1 = e1 + E::OFF_C;
  const double* J1 = e1 + E::OFF_J;
  const double* b1 = e1 + E::OFF_B;
  const double* eta1 = e1 + E::OFF_ETA;
  double* R0 = smem + C::OFF_R0;
  double* R1 = smem + C::OFF_R1;
  double* R2 = smem + C::OFF_R2;
  double* RA1 = smem + C::OFF_R3;  // == R1 unless PRE
  double* RA2 = smem + C::OFF_R4;
  constexpr bool PRE = C::PRE;
  double* mult = smem + C::OFF_MULT;
  double* vt = smem + C::OFF_VT;
  double* vw = smem + C::OFF_VW;
  double* seta2 = smem + C::OFF_ETA2;
  double* piv = smem + C::OFF_PIV;
  int* kof = reinterpret_cast<int*>(smem + C::OFF_KOF);
  double* flag = smem + C::OFF_FLAG;
  if (tid == 0) flag[0] = 0.0;
  for (int r = tid; r < C::MPAD; r += NT) kof[r] = -1;
  // ---- R0 = C1, R1 = J2 ; M = I + C1 J2 -> R2 ; t = b1 + C1 eta2 ----
  load_mat<NX, LDM, NT>(R0, C1, tid);
  load_mat<NX, LDM, NT>(R1, J2, tid);
  if (PRE) {
    if (part == 0) load_mat<NX, LDM, NT>(RA1, A1, tid);
    if (!closed2) load_mat<NX, LDM, NT>(RA2, A2, tid);
  }
  for (int i = tid; i < NX; i += NT) seta2[i] = eta2[i];
  RTOC_SCAN_SYNC();
  scan_gemm<NT, NX, NX, NX, 1, LDM, 1, LDM>(R0, R1, tid, [&](int row, int col, double v) {
    R2[row + col * LDM] = v + (row == col ? 1.0 : 0.0);
  });
  for (int r = tid; r < NX; r += NT) {
    double acc = b1[r];
    for (int k = 0; k < NX; ++k) acc += R0[r + k * LDM] * seta2[k];
    vt[r] = acc;
  }
  RTOC_SCAN_SYNC();
  // ---- Gauss-Jordan elimination of [M | A1 | t] / [M | C1] with implicit partial pivoting.  Column c lives
  //      in the registers of LPC adjacent lanes (lane h owns the rows h, h+LPC, ...).  Step k: the
  //      lanes of column k pick the pivot among the unused rows and publish the column, then all
  //      columns > k eliminate.  One barrier per step (the published column is double-buffered). ----
  const int ncol = part == 0 ? 2 * NX + 1 : 2 * NX;
  // (the compiler turns a length of 8 into register-relative moves and expands odd lengths into selects;
  //  padding 5 to 8 was measured slower than the selects)
  constexpr int VL = RPL;
  typedef typename ColVec<VL>::type colvec;
  colvec col[CPT];
#pragma unroll
  for (int s = 0; s < CPT; ++s) {
    const int h = tid % LPC, c = slot_col<C>(tid, s);
    if (c < ncol) {
#pragma unroll
      for (int t = 0; t < RPL; ++t) {
        const int r = h + LPC * t;
        double v = 0.0;
        if (r < NX) {
          if (c < NX)
            v = R2[r + c * LDM];
          else if (part == 1)
            v = R0[r + (c - NX) * LDM];
          else if (c < 2 * NX)
            v = PRE ? RA1[r + (c - NX) * LDM] : A1[r + (c - NX) * NX];
          else
            v = vt[r];
        }
        col[s][t] = v;
      }
    }
  }
  // The loop body is branch-free apart from the column tests: the rows r >= NX that pad the last lane
  // hold zeros throughout (0 - mk[r] * x with mk[r] = 0) and can never become pivots.  Pivot search of
  // the owner lanes: |.| of the unused rows (used rows: exponent cleared), maximum by v_max trees and
  // quad DPP, then the smallest row that attains it; the pivot VALUE is not extracted -- every thread
  // reads it back as mk[p].
  constexpr int UW = (RPL + 31) / 32;
  unsigned used[UW];  // bit t: owned row h + LPC*t has been a pivot row (same for all columns of a lane)
  for (int w = 0; w < UW; ++w) used[w] = 0u;
  int* pki = reinterpret_cast<int*>(piv);
  // search + publish of pivot column kk by its LPC owner lanes (slot s)
  auto publish = [&](const colvec& cs, int kk, int h) {
    double* mk = mult + (kk & 1) * C::MPAD;
    double a[RPL];
    double m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int t = 0; t < RPL; ++t) {
      const double cv = cs[t];
      mk[h + LPC * t] = cv;
      a[t] = masked_abs(cv, (used[t / 32] >> (t % 32)) & 1u);
      if (t & 1)
        m1 = fmax(m1, a[t]);
      else
        m0 = fmax(m0, a[t]);
    }
    double m = fmax(m0, m1);
    if (LPC > 1) m = fmax(m, quad_perm_d<QUAD_XOR1>(m));
    if (LPC > 2) m = fmax(m, quad_perm_d<QUAD_XOR2>(m));
    if (LPC > 4) m = fmax(m, quad_perm_d<HALF_MIRROR>(m));
    int p = 1 << 20;
    double pvl = 1.0;
#pragma unroll
    for (int t = RPL - 1; t >= 0; --t) {  // descending: the smallest row of this lane that attains m wins
      const bool hit = a[t] == m;
      p = hit ? h + LPC * t : p;
      pvl = hit ? cs[t] : pvl;
    }
    const int pl = p;
    if (LPC > 1) {
      const int o = quad_perm_i<QUAD_XOR1>(p);
      p = o < p ? o : p;
    }
    if (LPC > 2) {
      const int o = quad_perm_i<QUAD_XOR2>(p);
      p = o < p ? o : p;
    }
    if (LPC > 4) {
      const int o = quad_perm_i<HALF_MIRROR>(p);
      p = o < p ? o : p;
    }
    if (C::PUBLISH_RCP && pl == p && p < (1 << 20)) piv[2 + (kk & 1)] = fast_rcp(pvl);  // the lane that holds the pivot
    if (h == 0) {
      if (!(m >= 2.3e-308)) {  // no usable pivot: flag, go on with the first unused row
        flag[0] = 1.0;
        p = 0;
        for (int r = NX - 1; r >= 0; --r)
          if (kof[r] < 0) p = r;
        piv[2 + (kk & 1)] = 1.0;
      }
      pki[kk & 1] = p;
      kof[p] = kk;
    }
  };
  if (!(RTOC_SCAN_PROBE & 1)) {
#pragma unroll
    for (int s = 0; s < CPT; ++s)
      if (slot_col<C>(tid, s) == 0) publish(col[s], 0, tid % LPC);
  }
  for (int k = 0; k < ((RTOC_SCAN_PROBE & 1) ? 0 : NX); ++k) {
    const double* mk = mult + (k & 1) * C::MPAD;
    RTOC_SCAN_SYNC();
    const int p = pki[k & 1];
    const double ipv = C::PUBLISH_RCP ? piv[2 + (k & 1)] : fast_rcp(mk[p]);
    const int hp = uniform_int(p % LPC);
    const int h = tid % LPC;
    const bool mine = (p % LPC) == h;  // this lane owns the pivot row
    const int tp = uniform_int(p / LPC);
#pragma unroll
    for (int w = 0; w < UW; ++w) used[w] |= (mine && (tp / 32) == w) ? (1u << (tp % 32)) : 0u;
#pragma unroll
    for (int s = 0; s < CPT; ++s) {
      const int c = slot_col<C>(tid, s);
      if (c > k && c < ncol && !(RTOC_SCAN_PROBE & 8)) {
        // entry of the pivot row: tp is wave-uniform -> register-relative read / write instead of RPL selects
        double wp = col[s][tp];  // the lane that owns row p has the entry of the pivot row
        if (C::PUBLISH_RCP) {
          wp = group_bcast_d<LPC>(wp, tid & 63, hp);
        } else {
          wp = mine ? wp : 0.0;
          if (LPC > 1) wp += quad_perm_d<QUAD_XOR1>(wp);  // the other lanes contribute exact zeros
          if (LPC > 2) wp += quad_perm_d<QUAD_XOR2>(wp);
          if (LPC > 4) wp += quad_perm_d<HALF_MIRROR>(wp);
        }
        const double x = wp * ipv;
#pragma unroll
        for (int t = 0; t < RPL; ++t) col[s][t] -= mk[h + LPC * t] * x;
        if (mine) col[s][tp] = x;
        // the next pivot column goes out as soon as it is up to date (before this thread's other slots)
        if (c == k + 1 && k + 1 < NX && !(RTOC_SCAN_PROBE & 16)) publish(col[s], k + 1, h);
      }
      if ((RTOC_SCAN_PROBE & 24) && c == k + 1 && k + 1 < NX && h == 0) {  // timing probes: keep the hand-off alive
        pki[(k + 1) & 1] = k + 1;
        kof[k + 1] = k + 1;
      }
    }
  }
  RTOC_SCAN_SYNC();
  // the solution row k sits in pivot row p_k.  Ta (part 1: Tc) -> R0 (row-major, ld LDT), tb -> vt
#pragma unroll
  for (int s = 0; s < CPT; ++s) {
    const int h = tid % LPC, c = slot_col<C>(tid, s);
    if (c >= NX && c < ncol) {
#pragma unroll
      for (int t = 0; t < RPL; ++t) {
        const int r = h + LPC * t;
        if (r < NX) {
          if (c < 2 * NX)
            R0[kof[r] * LDT + (c - NX)] = col[s][t];
          else
            vt[kof[r]] = col[s][t];
        }
      }
    }
  }
  RTOC_SCAN_SYNC();
  if (RTOC_SCAN_PROBE & 2) return 0;
  const unsigned stat = flag[0] != 0.0 ? RTOC_STAT_NAN : 0u;
  if (part == 1) {
    // ---- V = A2 Tc -> R2 ; C = C2 + V A2^T (symmetric, stored transposed) ----
    if (!PRE) {
      load_mat<NX, LDM, NT>(RA2, A2, tid);  // R1: J2 is dead after the M product
      RTOC_SCAN_SYNC();
    }
    scan_gemm<NT, NX, NX, NX, 1, LDM, LDT, 1>(RA2, R0, tid,
                                              [&](int row, int col_, double v) { R2[row + col_ * LDM] = v; });
    RTOC_SCAN_SYNC();
    scan_gemm<NT, NX, NX, NX, 1, LDM, LDM, 1>(R2, RA2, tid, [&](int row, int col_, double v) {
      out[E::OFF_C + col_ + row * NX] = v + C2[col_ + row * NX];
    });
    return stat;
  }
  // ---- U = J2 Ta -> R2 ; w = eta2 - J2 tb ----
  scan_gemm<NT, NX, NX, NX, 1, LDM, LDT, 1>(R1, R0, tid,
                                            [&](int row, int col_, double v) { R2[row + col_ * LDM] = v; });
  for (int i = tid; i < NX; i += NT) {
    double acc = 0.0;
    for (int k = 0; k < NX; ++k) acc += R1[i + k * LDM] * vt[k];
    vw[i] = seta2[i] - acc;
  }
  RTOC_SCAN_SYNC();
  // ---- R1 = A1 ; J = J1 + A1^T U (stored transposed: J is symmetric) ; eta = eta1 + A1^T w ----
  if (!PRE) {
    load_mat<NX, LDM, NT>(RA1, A1, tid);
    RTOC_SCAN_SYNC();
  }
  double* Jout = closed2 ? ps_out + E::PS_P : out + E::OFF_J;
  double* eout = closed2 ? ps_out + E::PS_S : out + E::OFF_ETA;
  scan_gemm<NT, NX, NX, NX, LDM, 1, 1, LDM>(RA1, R2, tid, [&](int row, int col_, double v) {
    Jout[col_ + row * NX] = v + J1[col_ + row * NX];
  });
  for (int i = tid; i < NX; i += NT) {
    double acc = eta1[i];
    for (int k = 0; k < NX; ++k) acc += RA1[k + i * LDM] * vw[k];
    eout[i] = acc;
  }
  if (closed2) return stat;
  // ---- A2 in LDS ; A = A2 Ta (computed as Ta^T A2^T: coalesced stores) ; b = b2 + A2 tb ----
  if (!PRE) {
    RTOC_SCAN_SYNC();
    load_mat<NX, LDM, NT>(RA2, A2, tid);
    RTOC_SCAN_SYNC();
  }
  scan_gemm<NT, NX, NX, NX, 1, LDT, LDM, 1>(R0, RA2, tid, [&](int row, int col_, double v) {
    out[E::OFF_A + col_ + row * NX] = v;
  });
  for (int i = tid; i < NX; i += NT) {
    double acc = b2[i];
    for (int k = 0; k < NX; ++k) acc += RA2[i + k * LDM] * vt[k];
    out[E::OFF_B + i] = acc;
  }
  return stat;
}

// ---- forward recursion as a prefix scan -------------------------------------------------------------
// RiccatiRecursion::forwardRiccatiRecursion (reference src/riccati/riccati_recursion.cpp:83-131) without
// switching-time terms is the chain  dx_{i+1} = Phi_i dx_i + phi_i  of the closed-loop maps
//   Phi_i = Fxx + [0; Fvu K_i],  phi_i = Fx + [0; Fvu k_i]       (riccati_factorizer.cpp:200-216)
//   Phi_i = Fxx,                 phi_i = Fx  on impact grids      (:219-231)
// A Hillis-Steele prefix scan composes the maps, (Phi_i, phi_i) <- (Phi_i Phi_j, Phi_i phi_j + phi_i) with
// j = i - d; a map whose prefix reaches grid point 0 collapses to the vector dx_{i+1} (written straight into
// the direction record), and composing with such a partner is a mat-vec.  du, dlmdgmm, dxi of all grid points
// follow at once from dx_i (:200-277).  The element records of the backward scan are reused: [Phi | phi].
template <int NV>
struct FwdEltLayout {
  static constexpr int NX = 2 * NV;
  static constexpr int OFF_PHI = 0, OFF_VEC = pad8(NX * NX);
  static_assert(OFF_VEC + pad8(NX) <= EltLayout<NV>::STRIDE, "forward elements fit the backward element records");
};

template <int NV, int NU>
struct FwdCfg {
  static constexpr int NX = 2 * NV;
  static constexpr int LDM = scan_lds_ld(NX);
  static constexpr int REG = pad8(NX * LDM);
  static constexpr int OFF_A = 0, OFF_B = REG;          // two NX x NX operands (element: Phi, [Fvu | K])
  static constexpr int OFF_X = 2 * REG;                 // a vector
  static constexpr int LDS_DOUBLES = OFF_X + 2 * pad8(NX + NU);
  static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
};

// Grid point st < N -> (Phi, phi) into `elt`; st == 0 also writes dx_0 and dx_1 into the direction records.
template <int NV, int NU, int NS, int NT>
RTOC_SCAN_DEV void fwd_element_body(const rtoc_grid& g, int st, const double* kr, const double* rr,
                                    const double* dx0, double* elt, double* dir0, double* smem, int tid) {
  constexpr rtoc_layout SL = ScanLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout KL = SL.kkt, RL = SL.ric, DL = SL.dir;
  using C = FwdCfg<NV, NU>;
  using F = FwdEltLayout<NV>;
  constexpr int NX = C::NX, LDM = C::LDM;
  const bool impact = g.type == RTOC_GRID_IMPACT;
  const double* Fxx = kr + KL.off[RTOC_KKT_FXX];
  const double* Fvu = kr + KL.off[RTOC_KKT_FVU];
  const double* Fx = kr + KL.off[RTOC_KKT_FX];
  const double* K = rr + RL.off[RTOC_RIC_K];  // row-major NU x NX
  const double* kv = rr + RL.off[RTOC_RIC_KV];
  double* sPhi = smem + C::OFF_A;
  double* sB = smem + C::OFF_B;       // Fvu (NV x NU, ld NV) then K (NU x NX row-major)
  double* sK = sB + pad8(NV * NU);
  double* sx = smem + C::OFF_X;       // k (NU), then x0
  if (!impact) {
    for (int e = tid; e < NV * NU; e += NT) sB[e] = Fvu[e];
    for (int e = tid; e < NU * NX; e += NT) sK[e] = K[e];
    for (int e = tid; e < NU; e += NT) sx[e] = kv[e];
  }
  if (st == 0)
    for (int e = tid; e < NX; e += NT) sx[pad8(NU) + e] = dx0[e];
  RTOC_SCAN_SYNC();
  for (int idx = tid; idx < NX * NX; idx += NT) {
    const int r = idx % NX, c = idx / NX;
    double acc = Fxx[idx];
    if (!impact && r >= NV)
      for (int u = 0; u < NU; ++u) acc += sB[(r - NV) + u * NV] * sK[u * NX + c];
    elt[F::OFF_PHI + idx] = acc;
    sPhi[r + c * LDM] = acc;
  }
  for (int r = tid; r < NX; r += NT) {
    double acc = Fx[r];
    if (!impact && r >= NV)
      for (int u = 0; u < NU; ++u) acc += sB[(r - NV) + u * NV] * sx[u];
    elt[F::OFF_VEC + r] = acc;
  }
  if (st != 0) return;
  RTOC_SCAN_SYNC();
  for (int r = tid; r < NX; r += NT) {
    double acc = elt[F::OFF_VEC + r];  // written by this very thread
    for (int k = 0; k < NX; ++k) acc += sPhi[r + k * LDM] * sx[pad8(NU) + k];
    dir0[DL.off[RTOC_DIR_DX] + r] = sx[pad8(NU) + r];
    dir0[DL.stride + DL.off[RTOC_DIR_DX] + r] = acc;
  }
}

// Map of grid point i composed with its partner j = i - d.  closed2: the partner is the vector xj = dx_{j+1};
// the result is dx_{i+1} -> xout.  Otherwise (Phi_i Phi_j, Phi_i phi_j + phi_i) -> out.
template <int NV, int NU, int NT>
RTOC_SCAN_DEV void fwd_combine_body(const double* ei, const double* ej, const double* xj, bool closed2,
                                    double* out, double* xout, double* smem, int tid) {
  using C = FwdCfg<NV, NU>;
  using F = FwdEltLayout<NV>;
  constexpr int NX = C::NX, LDM = C::LDM;
  double* sI = smem + C::OFF_A;
  double* sJ = smem + C::OFF_B;
  double* sx = smem + C::OFF_X;
  load_mat<NX, LDM, NT>(sI, ei + F::OFF_PHI, tid);
  if (!closed2) load_mat<NX, LDM, NT>(sJ, ej + F::OFF_PHI, tid);
  for (int e = tid; e < NX; e += NT) sx[e] = closed2 ? xj[e] : ej[F::OFF_VEC + e];
  RTOC_SCAN_SYNC();
  for (int r = tid; r < NX; r += NT) {
    double acc = ei[F::OFF_VEC + r];
    for (int k = 0; k < NX; ++k) acc += sI[r + k * LDM] * sx[k];
    if (closed2)
      xout[r] = acc;
    else
      out[F::OFF_VEC + r] = acc;
  }
  if (closed2) return;
  // (Phi_i Phi_j)^T = Phi_j^T Phi_i^T: the transposed product makes the stores coalesced
  scan_gemm<NT, NX, NX, NX, LDM, 1, LDM, 1>(sJ, sI, tid, [&](int row, int col_, double v) {
    out[F::OFF_PHI + col_ + row * NX] = v;
  });
}

// du, dlmdgmm, dxi (and the zero switching-time entries) of grid point st from its dx (riccati_factorizer.cpp:200-277)
template <int NV, int NU, int NS, int NT>
RTOC_SCAN_DEV void fwd_finish_body(const rtoc_grid& g, bool terminal, const double* rr, double* dr, double* smem,
                                   int tid) {
  constexpr rtoc_layout SL = ScanLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout RL = SL.ric, DL = SL.dir;
  constexpr int NX = 2 * NV;
  double* sx = smem;
  for (int e = tid; e < NX; e += NT) sx[e] = dr[DL.off[RTOC_DIR_DX] + e];
  RTOC_SCAN_SYNC();
  const bool impact = g.type == RTOC_GRID_IMPACT;
  const int ns = (NS > 0 && !terminal && !impact && g.switching_constraint) ? g.dims : 0;
  for (int w = tid; w < NX + NU + ns; w += NT) {
    if (w < NX) {
      const double* P = rr + RL.off[RTOC_RIC_P] + w;
      double acc = -rr[RL.off[RTOC_RIC_S] + w];
      for (int j = 0; j < NX; ++j) acc += P[j * NX] * sx[j];
      dr[DL.off[RTOC_DIR_DLMDGMM] + w] = acc;
    } else if (w < NX + NU) {
      if (!terminal && !impact) {
        const int u = w - NX;
        const double* K = rr + RL.off[RTOC_RIC_K] + u * NX;
        double acc = rr[RL.off[RTOC_RIC_KV] + u];
        for (int j = 0; j < NX; ++j) acc += K[j] * sx[j];
        dr[DL.off[RTOC_DIR_DU] + u] = acc;
      }
    } else {
      const int q = w - NX - NU;
      const double* M = rr + RL.off[RTOC_RIC_M] + q;
      double acc = rr[RL.off[RTOC_RIC_MV] + q];
      for (int j = 0; j < NX; ++j) acc += M[j * (NS > 0 ? NS : 1)] * sx[j];
      dr[DL.off[RTOC_DIR_DXI] + q] = acc;
    }
  }
  if (tid == 0) {
    dr[DL.off[RTOC_DIR_DTS] + 0] = 0.0;
    dr[DL.off[RTOC_DIR_DTS] + 1] = 0.0;
  }
}

}  // namespace scan
}  // namespace rtoc
