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
// control", IEEE TAC 2023) -- in ceil(log2(#grid points)) combination levels, every level one
// workgroup per grid point.  Then the policies K, k (M, m) of all grid points are computed at once by
// the one-stage mode of riccati_backward_kernel from P_{i+1}, s_{i+1}, i.e. with the reference's own
// per-stage algebra (riccati_factorizer.cpp:44-90).
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
// The bodies are written as barrier-separated phases of "for (i = tid; i < n; i += NT)" loops without
// wave intrinsics, so that the very same code also compiles for the host with NT = 1
// (tests/cpp/scan_emulation.cpp checks the algebra and the indexing on the CPU against numpy).
#pragma once
#include <math.h>
#include <stdint.h>

#include "../../include/rtoc.h"

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

// ---- cooperative Cholesky of an n x n matrix in LDS (lower factor in place; the strict upper
//      triangle is left untouched).  Ends with a barrier.  *flag is set on a non-positive pivot. ----
template <int NT>
RTOC_SCAN_DEV void lds_cholesky(double* A, int n, int ld, int tid, double* flag) {
  for (int j = 0; j < n; ++j) {
    if (tid == 0) {
      double d = A[j + j * ld];
      if (!(d > 0.0)) {
        *flag = 1.0;
        d = 1.0;
      }
      A[j + j * ld] = sqrt(d);
    }
    RTOC_SCAN_SYNC();
    const double inv = 1.0 / A[j + j * ld];
    for (int i = j + 1 + tid; i < n; i += NT) A[i + j * ld] *= inv;
    RTOC_SCAN_SYNC();
    const int m = n - j - 1;
    for (int idx = tid; idx < m * m; idx += NT) {
      const int ii = idx % m, kk = idx / m;
      if (ii >= kk) {
        const int i = j + 1 + ii, k = j + 1 + kk;
        A[i + k * ld] -= A[i + j * ld] * A[k + j * ld];
      }
    }
    RTOC_SCAN_SYNC();
  }
}

// x <- L^-1 x for one column held in LDS
RTOC_SCAN_DEV void fwd_subst(const double* L, int n, int ld, double* x) {
  for (int k = 0; k < n; ++k) {
    double acc = x[k];
    for (int m = 0; m < k; ++m) acc -= L[k + m * ld] * x[m];
    x[k] = acc / L[k + k * ld];
  }
}

template <int NV, int NU, int NS>
struct ElementCfg {
  static constexpr int NX = 2 * NV;
  static constexpr int NSP = NS > 0 ? NS : 1;
  static constexpr int LDZ = NU | 1;   // odd leading dimensions: conflict-free column walks
  static constexpr int LDS_ = NSP | 1;
  static constexpr int C_S = 0, C_L = NX, C_B = NX + 1, C_D = NX + 1 + NV;  // columns of Z
  static constexpr int NZ = NX + 1 + NV + NS;
  static constexpr int OFF_L = 0;                                 // NU x NU
  static constexpr int OFF_Z = OFF_L + pad8(LDZ * NU);            // NU x NZ
  static constexpr int OFF_Y = OFF_Z + pad8(LDZ * NZ);            // [Ys | yl]  NU x (NX+1)
  static constexpr int OFF_ZBP = OFF_Y + pad8(LDZ * (NX + 1));    // Zb'        NU x NV
  static constexpr int OFF_QD = OFF_ZBP + pad8(LDZ * NV);         // Qd         NU x NS
  static constexpr int OFF_LS = OFF_QD + pad8(LDZ * NSP);         // Ls         NS x NS
  static constexpr int OFF_RX = OFF_LS + pad8(LDS_ * NSP);        // Ls^-1 [Phix | P]  NS x (NX+1)
  static constexpr int OFF_FLAG = OFF_RX + pad8(LDS_ * (NX + 1));
  static constexpr int LDS_DOUBLES = OFF_FLAG + 8;
  static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
};

// One grid point -> its element.  kr: the KKT record; elt: the element record (open grid points);
// ps: the closed value record (terminal only).  Returns RTOC_STAT_* bits (valid for tid == 0).
template <int NV, int NU, int NS, int NT>
RTOC_SCAN_DEV unsigned element_body(const rtoc_grid& g, const double* kr, double* elt, double* ps,
                                    double* smem, int tid) {
  constexpr rtoc_layout SL = ScanLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout KL = SL.kkt;
  using C = ElementCfg<NV, NU, NS>;
  using E = EltLayout<NV>;
  constexpr int NX = C::NX, LDZ = C::LDZ, LDSS = C::LDS_;
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
  double* sL = smem + C::OFF_L;
  double* sZ = smem + C::OFF_Z;
  double* sY = smem + C::OFF_Y;
  double* sZbp = smem + C::OFF_ZBP;
  double* sQd = smem + C::OFF_QD;
  double* sLs = smem + C::OFF_LS;
  double* sRx = smem + C::OFF_RX;
  double* flag = smem + C::OFF_FLAG;
  if (tid == 0) {
    flag[0] = 0.0;
    flag[1] = 0.0;
  }
  // ---- Quu -> sL ; [Qxu^T | lu | Fvu^T | Phiu^T] -> sZ ----
  for (int i = tid; i < NU * NU; i += NT) sL[(i % NU) + (i / NU) * LDZ] = Quu[i];
  for (int i = tid; i < NX * NU; i += NT) {  // Qxu(j,k), j = i % NX fastest (coalesced)
    const int j = i % NX, k = i / NX;
    sZ[k + (C::C_S + j) * LDZ] = Qxu[i];
  }
  for (int k = tid; k < NU; k += NT) sZ[k + C::C_L * LDZ] = lu[k];
  for (int i = tid; i < NV * NU; i += NT) {  // Fvu(j,k)
    const int j = i % NV, k = i / NV;
    sZ[k + (C::C_B + j) * LDZ] = Fvu[i];
  }
  for (int i = tid; i < ns * NU; i += NT) {  // Phiu(j,k), ld ns_max
    const int j = i % ns, k = i / ns;
    sZ[k + (C::C_D + j) * LDZ] = Phiu[j + k * ldn];
  }
  RTOC_SCAN_SYNC();
  lds_cholesky<NT>(sL, NU, LDZ, tid, flag);
  // ---- Z <- L^-1 Z, one column per thread ----
  {
    const int ncol = C::C_D + ns;
    for (int c = tid; c < ncol; c += NT) fwd_subst(sL, NU, LDZ, sZ + c * LDZ);
  }
  RTOC_SCAN_SYNC();
  const double* Zs = sZ + C::C_S * LDZ;
  const double* zl = sZ + C::C_L * LDZ;
  const double* Zb = sZ + C::C_B * LDZ;
  const double* Ys = Zs;
  const double* yl = zl;
  const double* Zbp = Zb;
  if (ns > 0) {
    const double* Zd = sZ + C::C_D * LDZ;
    for (int idx = tid; idx < ns * ns; idx += NT) {
      const int i = idx % ns, j = idx / ns;
      double acc = 0.0;
      for (int k = 0; k < NU; ++k) acc += Zd[k + i * LDZ] * Zd[k + j * LDZ];
      sLs[i + j * LDSS] = acc;
    }
    RTOC_SCAN_SYNC();
    lds_cholesky<NT>(sLs, ns, LDSS, tid, flag + 1);
    // Qd = Zd Ls^-T (row k of Zd per thread); Rx = Ls^-1 [Phix | P] (one column per thread)
    for (int w = tid; w < NU + NX + 1; w += NT) {
      if (w < NU) {
        const int k = w;
        for (int j = 0; j < ns; ++j) {
          double acc = Zd[k + j * LDZ];
          for (int m = 0; m < j; ++m) acc -= sQd[k + m * LDZ] * sLs[j + m * LDSS];
          sQd[k + j * LDZ] = acc / sLs[j + j * LDSS];
        }
      } else {
        const int c = w - NU;
        double* x = sRx + c * LDSS;
        for (int j = 0; j < ns; ++j) x[j] = (c < NX) ? Phix[j + c * ldn] : Pres[j];
        fwd_subst(sLs, ns, LDSS, x);
      }
    }
    RTOC_SCAN_SYNC();
    // [Ys | yl] = z + Qd (Rx - Qd^T z) ;  Zb' = z - Qd Qd^T z
    for (int w = tid; w < NX + 1 + NV; w += NT) {
      const bool isb = w > NX;
      const double* z = isb ? Zb + (w - NX - 1) * LDZ : sZ + w * LDZ;
      double* out = isb ? sZbp + (w - NX - 1) * LDZ : sY + w * LDZ;
      double t[C::NSP];
      for (int j = 0; j < ns; ++j) {
        double acc = isb ? 0.0 : sRx[j + w * LDSS];
        for (int k = 0; k < NU; ++k) acc -= sQd[k + j * LDZ] * z[k];
        t[j] = acc;
      }
      for (int k = 0; k < NU; ++k) {
        double acc = z[k];
        for (int j = 0; j < ns; ++j) acc += sQd[k + j * LDZ] * t[j];
        out[k] = acc;
      }
    }
    RTOC_SCAN_SYNC();
    Ys = sY;
    yl = sY + NX * LDZ;
    Zbp = sZbp;
  }
  // ---- element ----
  for (int idx = tid; idx < NX * NX; idx += NT) {
    const int i = idx % NX, j = idx / NX;
    double aj = 0.0, aa = 0.0, ac = 0.0;
    for (int k = 0; k < NU; ++k) {
      const double zsi = Zs[k + i * LDZ], ysi = Ys[k + i * LDZ];
      const double zsj = Zs[k + j * LDZ], ysj = Ys[k + j * LDZ];
      aj += zsi * ysj + ysi * (zsj - ysj);
    }
    if (i >= NV) {
      for (int k = 0; k < NU; ++k) aa += Zb[k + (i - NV) * LDZ] * Ys[k + j * LDZ];
      if (j >= NV)
        for (int k = 0; k < NU; ++k) ac += Zbp[k + (i - NV) * LDZ] * Zbp[k + (j - NV) * LDZ];
    }
    elt[E::OFF_J + idx] = Qxx[idx] - aj;
    elt[E::OFF_A + idx] = Fxx[idx] - aa;
    elt[E::OFF_C + idx] = ac;
  }
  for (int i = tid; i < NX; i += NT) {
    double ae = 0.0, ab = 0.0;
    for (int k = 0; k < NU; ++k) {
      const double zsi = Zs[k + i * LDZ], ysi = Ys[k + i * LDZ];
      ae += zsi * yl[k] + ysi * (zl[k] - yl[k]);
    }
    if (i >= NV)
      for (int k = 0; k < NU; ++k) ab += Zb[k + (i - NV) * LDZ] * yl[k];
    elt[E::OFF_ETA + i] = -lx[i] + ae;
    elt[E::OFF_B + i] = Fx[i] - ab;
  }
  unsigned stat = 0;
  if (flag[0] != 0.0) stat |= RTOC_STAT_QUU_NOT_SPD;
  if (flag[1] != 0.0) stat |= RTOC_STAT_S_NOT_SPD;
  return stat;
}

// ---- combination ---------------------------------------------------------------------------------
template <int NV, int NT>
struct CombineCfg {
  static constexpr int NX = 2 * NV;
  static constexpr int LDW = 3 * NX + 1;                 // [M | A1 | t | C1], row-major, odd
  static constexpr int CPT = (LDW + NT - 1) / NT;        // columns per thread (1 on the GPU)
  static constexpr int LDU = NX | 1;
  static constexpr int OFF_W = 0;                        // NX x LDW
  static constexpr int OFF_U = OFF_W + pad8(NX * LDW);   // NX x NX scratch (J2 Ta, then A2 Tc)
  static constexpr int OFF_MULT = OFF_U + pad8(NX * LDU);  // 2 x NX multipliers (double-buffered)
  static constexpr int OFF_VEC = OFF_MULT + 2 * pad8(NX);  // u = J2 tb ; w = eta2 - u
  static constexpr int OFF_PIV = OFF_VEC + 2 * pad8(NX);   // per step: pivot row (as double), 1/pivot
  static constexpr int OFF_KOF = OFF_PIV + 8;              // kof[r] = elimination step whose pivot row is r
  static constexpr int OFF_FLAG = OFF_KOF + pad8(NX);
  static constexpr int LDS_DOUBLES = OFF_FLAG + 8;
  static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
};

// e1 = element of [i, j), (J2, eta2, A2, b2, C2) = element of [j, k).  closed2: [j,k) contains the
// terminal grid point, then J2 / eta2 are its value record, A2 / b2 / C2 are not read and the result
// is the closed value record `ps_out`; otherwise the result is the element `out`.
template <int NV, int NT>
RTOC_SCAN_DEV unsigned combine_body(const double* e1, const double* J2, const double* eta2,
                                    const double* A2, const double* b2, const double* C2, bool closed2,
                                    double* out, double* ps_out, double* smem, int tid) {
  using C = CombineCfg<NV, NT>;
  using E = EltLayout<NV>;
  constexpr int NX = C::NX, LDW = C::LDW, CPT = C::CPT, LDU = C::LDU;
  const double* A1 = e1 + E::OFF_A;
  const double* C1 = e1 + E::OFF_C;
  const double* J1 = e1 + E::OFF_J;
  const double* b1 = e1 + E::OFF_B;
  const double* eta1 = e1 + E::OFF_ETA;
  double* W = smem + C::OFF_W;
  double* U = smem + C::OFF_U;
  double* mult = smem + C::OFF_MULT;
  double* vu = smem + C::OFF_VEC;
  double* vw = vu + pad8(NX);
  double* piv = smem + C::OFF_PIV;
  double* kof = smem + C::OFF_KOF;
  double* flag = smem + C::OFF_FLAG;
  if (tid == 0) flag[0] = 0.0;
  // ---- W = [I + C1 J2 | A1 | b1 + C1 eta2 | C1] (row-major) ----
  for (int idx = tid; idx < NX * NX; idx += NT) {
    const int r = idx % NX, c = idx / NX;
    double acc = (r == c) ? 1.0 : 0.0;
    for (int k = 0; k < NX; ++k) acc += C1[r + k * NX] * J2[k + c * NX];
    W[r * LDW + c] = acc;
    W[r * LDW + NX + c] = A1[idx];
    W[r * LDW + 2 * NX + 1 + c] = C1[idx];
  }
  for (int r = tid; r < NX; r += NT) {
    double acc = b1[r];
    for (int k = 0; k < NX; ++k) acc += C1[r + k * NX] * eta2[k];
    W[r * LDW + 2 * NX] = acc;
  }
  RTOC_SCAN_SYNC();
  // ---- Gauss-Jordan with implicit partial pivoting; thread c keeps column c in registers.  Step k:
  //      the owner of column k picks the pivot among the unused rows and publishes the column, all
  //      threads eliminate.  One barrier per step (the published column is double-buffered). ----
  double col[CPT][NX];
  for (int s = 0; s < CPT; ++s) {
    const int c = tid + s * NT;
    if (c < LDW) {
#pragma unroll
      for (int r = 0; r < NX; ++r) col[s][r] = W[r * LDW + c];
    }
  }
  uint64_t used_lo = 0, used_hi = 0;  // rows already taken as pivots (NX <= 128)
  for (int k = 0; k < NX; ++k) {
    double* mk = mult + (k & 1) * pad8(NX);
    double* pk = piv + (k & 1) * 2;
    for (int s = 0; s < CPT; ++s) {
      const int c = tid + s * NT;
      if (c == k) {
        int p = -1;
        double best = -1.0;
#pragma unroll
        for (int r = 0; r < NX; ++r) {
          const bool used = r < 64 ? ((used_lo >> r) & 1) : ((used_hi >> (r - 64)) & 1);
          const double a = fabs(col[s][r]);
          if (!used && a > best) {
            best = a;
            p = r;
          }
          mk[r] = col[s][r];
        }
        if (!(best > 0.0)) {
          flag[0] = 1.0;
          if (p < 0) p = 0;
        }
        double pv = 0.0;
#pragma unroll
        for (int r = 0; r < NX; ++r)
          if (r == p) pv = col[s][r];
        pk[0] = (double)p;
        pk[1] = 1.0 / pv;
        kof[p] = (double)k;
      }
    }
    RTOC_SCAN_SYNC();
    const int p = (int)pk[0];
    const double ipv = pk[1];
    if (p < 64)
      used_lo |= (uint64_t)1 << p;
    else
      used_hi |= (uint64_t)1 << (p - 64);
    for (int s = 0; s < CPT; ++s) {
      const int c = tid + s * NT;
      if (c > k && c < LDW) {
        double wp = 0.0;
#pragma unroll
        for (int r = 0; r < NX; ++r)
          if (r == p) wp = col[s][r];
        const double x = wp * ipv;
#pragma unroll
        for (int r = 0; r < NX; ++r) col[s][r] = (r == p) ? x : col[s][r] - mk[r] * x;
      }
    }
  }
  RTOC_SCAN_SYNC();
  // solution row k sits in pivot row p_k: T[k][c] = col[p_k] -> back into W (row-major, rows = k)
  for (int s = 0; s < CPT; ++s) {
    const int c = tid + s * NT;
    if (c >= NX && c < LDW) {
#pragma unroll
      for (int r = 0; r < NX; ++r) W[(int)kof[r] * LDW + c] = col[s][r];
    }
  }
  RTOC_SCAN_SYNC();
  const double* Ta = W + NX;          // Ta[k][j] = W[k*LDW + NX + j]
  const double* tb = W + 2 * NX;      // tb[k]    = W[k*LDW + 2NX]
  const double* Tc = W + 2 * NX + 1;  // Tc[k][j]
  // ---- U = J2 Ta (row-major, ld LDU) ; u = J2 tb ; w = eta2 - u ----
  for (int idx = tid; idx < NX * NX; idx += NT) {
    const int i = idx % NX, j = idx / NX;
    double acc = 0.0;
    for (int k = 0; k < NX; ++k) acc += J2[i + k * NX] * Ta[k * LDW + j];
    U[i * LDU + j] = acc;
  }
  for (int i = tid; i < NX; i += NT) {
    double acc = 0.0;
    for (int k = 0; k < NX; ++k) acc += J2[i + k * NX] * tb[k * LDW];
    vw[i] = eta2[i] - acc;
  }
  RTOC_SCAN_SYNC();
  // ---- J = J1 + A1^T U (stored transposed: J is symmetric) ; eta = eta1 + A1^T w ----
  double* Jout = closed2 ? ps_out + E::PS_P : out + E::OFF_J;
  double* eout = closed2 ? ps_out + E::PS_S : out + E::OFF_ETA;
  for (int idx = tid; idx < NX * NX; idx += NT) {
    const int j = idx % NX, i = idx / NX;
    double acc = J1[idx];
    for (int k = 0; k < NX; ++k) acc += A1[k + i * NX] * U[k * LDU + j];
    Jout[idx] = acc;
  }
  for (int i = tid; i < NX; i += NT) {
    double acc = eta1[i];
    for (int k = 0; k < NX; ++k) acc += A1[k + i * NX] * vw[k];
    eout[i] = acc;
  }
  const unsigned stat = flag[0] != 0.0 ? RTOC_STAT_NAN : 0u;
  if (closed2) return stat;
  RTOC_SCAN_SYNC();
  // ---- A = A2 Ta ; b = b2 + A2 tb ; V = A2 Tc -> U (column-major, ld LDU) ----
  for (int idx = tid; idx < NX * NX; idx += NT) {
    const int i = idx % NX, j = idx / NX;
    double acc = 0.0, acv = 0.0;
    for (int k = 0; k < NX; ++k) {
      const double a2 = A2[i + k * NX];
      acc += a2 * Ta[k * LDW + j];
      acv += a2 * Tc[k * LDW + j];
    }
    out[E::OFF_A + idx] = acc;
    U[i + j * LDU] = acv;
  }
  for (int i = tid; i < NX; i += NT) {
    double acc = b2[i];
    for (int k = 0; k < NX; ++k) acc += A2[i + k * NX] * tb[k * LDW];
    out[E::OFF_B + i] = acc;
  }
  RTOC_SCAN_SYNC();
  // ---- C = C2 + V A2^T ----
  for (int idx = tid; idx < NX * NX; idx += NT) {
    const int i = idx % NX, j = idx / NX;
    double acc = C2[idx];
    for (int k = 0; k < NX; ++k) acc += U[i + k * LDU] * A2[j + k * NX];
    out[E::OFF_C + idx] = acc;
  }
  return stat;
}

}  // namespace scan
}  // namespace rtoc
