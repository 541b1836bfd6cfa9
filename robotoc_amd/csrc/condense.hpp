// condense.hpp -- batched KKT condensation / expansion kernels for gfx950.
//
// condense_kernel replaces, per (OCP instance, grid point):
//   Robot::computeMJtJinv            include/robotoc/robot/robot.hxx:642-684 (dense LLT instead of
//                                    Pinocchio's sparse Cholesky: same matrix, other elimination order)
//   condenseContactDynamics          src/dynamics/contact_dynamics.cpp:55-164
//   condenseImpactDynamics           src/dynamics/impact_dynamics.cpp:38-80
//   the STO scalings at the end of IntermediateStage::evalKKT (src/ocp/intermediate_stage.cpp:140-148)
// expand_kernel replaces expandContactDynamicsPrimal/Dual (contact_dynamics.cpp:167-202) and the
// impact forms (impact_dynamics.cpp:83-96).
//
// Mapping.  Unlike the Riccati recursion this work has no chain over the horizon: every
// (instance, grid point) pair is independent, i.e. batch x stages ~ 1.9e5 work items for the
// headline workload.  One wavefront per work item; the dense products run on the f64 matrix
// cores straight from the (L2-resident, 64-byte aligned) records -- there is no reuse across
// work items to stage in LDS, and 8 resident waves per SIMD hide the operand latency, which a
// LDS-staged, 40 KB-per-item design (3 items per CU) could not.  The intermediate blocks
// (MJtJinv, MJtJinv_dIDCdqv, Qafqv, Qafu_full) are outputs of the reference as well
// (ContactDynamicsData keeps them for the expansion), so they are produced in place in the
// contact-dynamics record and re-read from there.
#pragma once
#include "device_utils.hpp"
#include "riccati_backward.hpp"  // wave_llt, llt_solve_reg
#include "../../include/rtoc.h"

namespace rtoc {

struct CondArgs {
  double* kkt;
  double* cdd;
  const rtoc_grid* grid;
  uint32_t* status;
  int nstages, batch;
  double damping;  // RobotModelInfo::contact_inv_damping (robot_model_info.hpp:95)
  rtoc_record_layout kl, cl;
};

struct ExpArgs {
  double* cdd;
  double* dir;
  const rtoc_grid* grid;
  int nstages, batch;
  rtoc_record_layout cl, dl;
  double tau;
};

template <int NV, int NU, int NF, int NS>
struct CondCfg {
  static constexpr int NT = 64;
  static constexpr int NFP = NF > 0 ? NF : 1;
  static constexpr int pad8(int n) { return (n + 7) & ~7; }
  static constexpr int O_L = 0;                          // Cholesky factor of M      NV x NV
  static constexpr int O_JM = O_L + pad8(NV * NV);       // J Minv                    NF x NV (ld NFP)
  static constexpr int O_S = O_JM + pad8(NFP * NV);      // J Minv J^T and its factor NF x NF
  static constexpr int O_BR = O_S + pad8(NFP * NFP);     // -(J Minv J^T)^-1          NF x NF
  static constexpr int O_LINV = O_BR + pad8(NFP * NFP);  // 1/diag
  static constexpr int O_SINV = O_LINV + 64;
  static constexpr int LDS_DOUBLES = O_SINV + 64;
  static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
};

// One-wave dense product on the f64 matrix cores with generic (row, column) strides:
//   C(i,j) = beta * C(i,j) + alpha * sum_k A(i,k) B(k,j),  A(i,k) = A[i*ars + k*acs], ...
// Operands may live in HBM/L2 or LDS (flat addressing).  Runtime dimensions; tiles are
// processed two column tiles at a time to keep two MFMA chains in flight.
__device__ __forceinline__ void wave_gemm(int M, int N, int K, double alpha, const double* A, int ars,
                                          int acs, const double* B, int brs, int bcs, double beta,
                                          double* C, int crs, int ccs, int lane) {
  const int li = lane & 15, q = lane >> 4;
  const int tmn = (M + 15) >> 4, tnn = (N + 15) >> 4, ksn = (K + 3) >> 2;
  for (int tm = 0; tm < tmn; ++tm) {
    const int i = tm * 16 + li;
    const bool iok = i < M;
    const double* ap = A + (size_t)(iok ? i : 0) * ars;
    for (int tn = 0; tn < tnn; tn += 2) {
      const int j0 = tn * 16 + li, j1 = j0 + 16;
      const bool j0ok = j0 < N, j1ok = j1 < N;
      const double* bp0 = B + (size_t)(j0ok ? j0 : 0) * bcs;
      const double* bp1 = B + (size_t)(j1ok ? j1 : 0) * bcs;
      d4 acc0 = zero4(), acc1 = zero4();
      for (int ks = 0; ks < ksn; ++ks) {
        const int k = ks * 4 + q;
        const bool kok = k < K;
        const int kc = kok ? k : 0;
        const double av = ap[(size_t)kc * acs];
        const double b0 = bp0[(size_t)kc * brs];
        const double b1 = bp1[(size_t)kc * brs];
        const double a_ = (iok && kok) ? av : 0.0;
        acc0 = mfma16(a_, (j0ok && kok) ? b0 : 0.0, acc0);
        acc1 = mfma16(a_, (j1ok && kok) ? b1 : 0.0, acc1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = tm * 16 + drow(q, r);
        if (row < M) {
          if (j0ok) {
            double* c = C + (size_t)row * crs + (size_t)j0 * ccs;
            *c = (beta == 0.0 ? 0.0 : beta * *c) + alpha * acc0[r];
          }
          if (j1ok) {
            double* c = C + (size_t)row * crs + (size_t)j1 * ccs;
            *c = (beta == 0.0 ? 0.0 : beta * *c) + alpha * acc1[r];
          }
        }
      }
    }
  }
}

// y(i) = beta*y(i) + alpha * sum_k A(i,k) x(k): one lane per row, operands anywhere.
__device__ __forceinline__ void wave_gemv(int M, int K, double alpha, const double* A, int ars, int acs,
                                          const double* x, double beta, double* y, int lane) {
  for (int i = lane; i < M; i += 64) {
    double acc = 0.0;
    for (int k = 0; k < K; ++k) acc += A[(size_t)i * ars + (size_t)k * acs] * x[k];
    y[i] = (beta == 0.0 ? 0.0 : beta * y[i]) + alpha * acc;
  }
}

template <int NV, int NU, int NF, int NS>
__global__ __launch_bounds__(64) void condense_kernel(CondArgs a) {
  using C = CondCfg<NV, NU, NF, NS>;
  constexpr int NX = 2 * NV, NP = NV - NU, LDV = NV + NF, LDF = C::NFP, LDS_ = NS > 0 ? NS : 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* const sL = smem + C::O_L;
  double* const sJM = smem + C::O_JM;
  double* const sS = smem + C::O_S;
  double* const sBR = smem + C::O_BR;
  double* const sLinv = smem + C::O_LINV;
  double* const sSinv = smem + C::O_SINV;
  const int lane = threadIdx.x;
  const int item = blockIdx.x;  // instance * (nstages-1) + stage
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  const bool impact = g.type == RTOC_GRID_IMPACT;
  const int nf = g.dimf, nvf = NV + nf, ns = impact ? 0 : g.dims;
  const double dt = g.dt;
  double* kr = a.kkt + ((size_t)b * a.nstages + st) * a.kl.stride;
  double* cr = a.cdd + ((size_t)b * a.nstages + st) * a.cl.stride;
  const int* ko = a.kl.off;
  const int* co = a.cl.off;
  double* const M = cr + co[RTOC_CDD_DIDDA];
  double* const D = cr + co[RTOC_CDD_DIDCDQV];
  // J: dCda (ld NF) on contact grids, dCdv = D[nv:, nv:] (ld LDV) on impact grids
  const double* const J = impact ? D + NV + (size_t)NV * LDV : cr + co[RTOC_CDD_DCDA];
  const int ldj = impact ? LDV : LDF;
  double* const IDC = cr + co[RTOC_CDD_IDC];
  double* const Qaa = cr + co[RTOC_CDD_QAA];
  double* const Qff = cr + co[RTOC_CDD_QFF];
  double* const Qqf = cr + co[RTOC_CDD_QQF];
  double* const la = cr + co[RTOC_CDD_LA];
  double* const lf = cr + co[RTOC_CDD_LF];
  double* const ha = cr + co[RTOC_CDD_HA];
  double* const hf = cr + co[RTOC_CDD_HF];
  double* const Phia = cr + co[RTOC_CDD_PHIA];
  double* const lup = cr + co[RTOC_CDD_LUP];
  double* const Lam = cr + co[RTOC_CDD_MJTJINV];
  double* const LD = cr + co[RTOC_CDD_MJD];
  double* const Lr = cr + co[RTOC_CDD_MJIDC];
  double* const Qafqv = cr + co[RTOC_CDD_QAFQV];
  double* const Qafu = cr + co[RTOC_CDD_QAFU];
  double* const laf = cr + co[RTOC_CDD_LAF];
  double* const Qxup = cr + co[RTOC_CDD_QXUP];
  double* const Quuptr = cr + co[RTOC_CDD_QUUPTR];
  double* const haf = cr + co[RTOC_CDD_HAF];
  double* const Fxx = kr + ko[RTOC_KKT_FXX];
  double* const Fvu = kr + ko[RTOC_KKT_FVU];
  double* const Qxx = kr + ko[RTOC_KKT_QXX];
  double* const Qxu = kr + ko[RTOC_KKT_QXU];
  double* const Quu = kr + ko[RTOC_KKT_QUU];
  double* const Fx = kr + ko[RTOC_KKT_FX];
  double* const lx = kr + ko[RTOC_KKT_LX];
  double* const lu = kr + ko[RTOC_KKT_LU];
  double* const fx = kr + ko[RTOC_KKT_FFX];
  double* const hx = kr + ko[RTOC_KKT_HX];
  double* const hu = kr + ko[RTOC_KKT_HU];
  double* const scal = kr + ko[RTOC_KKT_SCAL];
  double* const Phix = kr + ko[RTOC_KKT_PHIX];
  double* const Phiu = kr + ko[RTOC_KKT_PHIU];
  double* const Phit = kr + ko[RTOC_KKT_PHIT];
  double* const Pres = kr + ko[RTOC_KKT_PRES];
  unsigned stat = 0;

  // ================= computeMJtJinv (robot.hxx:642-684) =================
  if (wave_llt<NV, NV>(M, sL, sLinv, NV, lane)) stat |= RTOC_STAT_M_NOT_SPD;
  __syncthreads();
  // topLeft = M^-1: lane t < NV solves column t
  if (lane < NV) {
    double x[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] = (i == lane) ? 1.0 : 0.0;
    llt_solve_reg<NV, NV>(sL, sLinv, x, NV);
#pragma unroll
    for (int i = 0; i < NV; ++i) Lam[i + (size_t)lane * LDV] = x[i];
  }
  __syncthreads();
  if (nf > 0) {
    // bottomLeft = J M^-1 (:677) -> sJM ; JMinvJt (:660-661) -> sS
    wave_gemm(nf, NV, NV, 1.0, J, 1, ldj, Lam, 1, LDV, 0.0, sJM, 1, LDF, lane);
    __syncthreads();
    wave_gemm(nf, nf, NV, 1.0, sJM, 1, LDF, J, ldj, 1, 0.0, sS, 1, LDF, lane);
    __syncthreads();
    if (lane < nf) sS[lane + lane * LDF] += a.damping;  // (:662-664)
    __syncthreads();
    if (wave_llt<C::NFP, C::NFP>(sS, sS, sSinv, nf, lane)) stat |= RTOC_STAT_M_NOT_SPD;  // (:665)
    __syncthreads();
    // bottomRight = -(JMinvJt)^-1 (:673-675): lane t < nf solves column t of -I
    if (lane < nf) {
      double x[C::NFP];
#pragma unroll
      for (int i = 0; i < C::NFP; ++i) x[i] = (i == lane) ? -1.0 : 0.0;
      llt_solve_reg<C::NFP, C::NFP>(sS, sSinv, x, nf);
#pragma unroll
      for (int i = 0; i < C::NFP; ++i)
        if (i < nf) {
          sBR[i + lane * LDF] = x[i];
          Lam[(NV + i) + (size_t)(NV + lane) * LDV] = x[i];
        }
    }
    __syncthreads();
    // topRight = bottomLeft^T * (-bottomRight) (:678)
    wave_gemm(NV, nf, nf, -1.0, sJM, LDF, 1, sBR, 1, LDF, 0.0, Lam + (size_t)NV * LDV, 1, LDV, lane);
    __syncthreads();
    // topLeft -= topRight * bottomLeft (:679) ; bottomLeft = topRight^T (:680)
    wave_gemm(NV, NV, nf, -1.0, Lam + (size_t)NV * LDV, 1, LDV, sJM, 1, LDF, 1.0, Lam, 1, LDV, lane);
    for (int e = lane; e < nf * NV; e += 64) {
      const int i = e % nf, j = e / nf;
      Lam[(NV + i) + (size_t)j * LDV] = Lam[j + (size_t)(NV + i) * LDV];
    }
    __syncthreads();
  }

  // ================= MJtJinv_dIDCdqv, MJtJinv_IDC (contact_dynamics.cpp:64-65 / impact :44-50) ===
  if (!impact) {
    wave_gemm(nvf, NX, nvf, 1.0, Lam, 1, LDV, D, 1, LDV, 0.0, LD, 1, LDV, lane);
  } else {
    wave_gemm(nvf, NV, nvf, 1.0, Lam, 1, LDV, D, 1, LDV, 0.0, LD, 1, LDV, lane);
    // right half only through dCdv: [Lam[:, f]] * dCdv
    wave_gemm(nvf, NV, nf, 1.0, Lam + (size_t)NV * LDV, 1, LDV, J, 1, ldj, 0.0, LD + (size_t)NV * LDV, 1, LDV,
              lane);
  }
  wave_gemv(nvf, nvf, 1.0, Lam, 1, LDV, IDC, 0.0, Lr, lane);
  __syncthreads();

  // ================= Qafqv, Qafu_full, laf (:67-88) =================
  for (int e = lane; e < NV * NX; e += 64) {
    const int i = e % NV, j = e / NV;
    Qafqv[i + (size_t)j * LDV] = -Qaa[i] * LD[i + (size_t)j * LDV];
  }
  if (!impact)
    for (int e = lane; e < NV * NV; e += 64) {
      const int i = e % NV, j = e / NV;
      Qafu[i + (size_t)j * LDV] = Qaa[i] * Lam[i + (size_t)j * LDV];
    }
  if (lane < NV) laf[lane] = la[lane] - Qaa[lane] * Lr[lane];
  if (!impact && lane < NV) haf[lane] = ha[lane];
  if (nf > 0) {
    wave_gemm(nf, NX, nf, -1.0, Qff, 1, LDF, LD + NV, 1, LDV, 0.0, Qafqv + NV, 1, LDV, lane);
    if (!impact) wave_gemm(nf, NV, nf, 1.0, Qff, 1, LDF, Lam + NV, 1, LDV, 0.0, Qafu + NV, 1, LDV, lane);
    if (lane < nf) {
      double acc = 0.0;
      for (int k = 0; k < nf; ++k) acc += Qff[lane + k * LDF] * Lr[NV + k];
      laf[NV + lane] = -lf[lane] - acc;
      if (!impact) haf[NV + lane] = -hf[lane];
    }
    __syncthreads();
    for (int e = lane; e < nf * NV; e += 64) {
      const int i = e % nf, j = e / nf;
      Qafqv[(NV + i) + (size_t)j * LDV] -= Qqf[j + (size_t)i * NV];
    }
  }
  __syncthreads();

  // ================= Schur updates of the Hessian blocks and gradients (:90-130) =================
  wave_gemm(NX, NX, nvf, -1.0, LD, LDV, 1, Qafqv, 1, LDV, 1.0, Qxx, 1, NX, lane);
  wave_gemv(NX, nvf, -1.0, LD, LDV, 1, laf, 1.0, lx, lane);
  if (!impact) {
    if (NP > 0) {
      wave_gemm(NX, NP, nvf, -1.0, LD, LDV, 1, Qafu, 1, LDV, 0.0, Qxup, 1, NX, lane);
      wave_gemm(NP, NU, nvf, 1.0, Lam, 1, LDV, Qafu + (size_t)NP * LDV, 1, LDV, 0.0, Quuptr, 1, NP, lane);
      wave_gemv(NP, nvf, 1.0, Lam, 1, LDV, laf, 1.0, lup, lane);
    }
    wave_gemm(NX, NU, nvf, -1.0, LD, LDV, 1, Qafu + (size_t)NP * LDV, 1, LDV, 1.0, Qxu, 1, NX, lane);
    wave_gemm(NU, NU, nvf, 1.0, Lam + NP, 1, LDV, Qafu + (size_t)NP * LDV, 1, LDV, 1.0, Quu, 1, NU, lane);
    wave_gemv(NU, nvf, 1.0, Lam + NP, 1, LDV, laf, 1.0, lu, lane);
    // STO sensitivities (:156-163)
    wave_gemv(NX, nvf, -1.0, LD, LDV, 1, haf, 1.0, hx, lane);
    wave_gemv(NU, nvf, 1.0, Lam + NP, 1, LDV, haf, 1.0, hu, lane);
  }
  __syncthreads();
  if (nf > 0) {
    // the Qqf corrections touch rows < NV of blocks updated above
    wave_gemm(NV, NX, nf, 1.0, Qqf, 1, NV, LD + NV, 1, LDV, 1.0, Qxx, 1, NX, lane);
    wave_gemv(NV, nf, 1.0, Qqf, 1, NV, Lr + NV, 1.0, lx, lane);
    if (!impact) {
      if (NP > 0) wave_gemm(NV, NP, nf, -1.0, Qqf, 1, NV, Lam + NV, 1, LDV, 1.0, Qxup, 1, NX, lane);
      wave_gemm(NV, NU, nf, -1.0, Qqf, 1, NV, Lam + NV + (size_t)NP * LDV, 1, LDV, 1.0, Qxu, 1, NX, lane);
      wave_gemv(NV, nf, 1.0 / dt, Qqf, 1, NV, Lr + NV, 1.0, hx, lane);
    }
  }

  // ================= condensed dynamics (:132-136 / impact :74-77) =================
  const double sdt = impact ? 1.0 : dt;
  for (int e = lane; e < NV * NV; e += 64) {
    const int i = e % NV, j = e / NV;
    Fxx[(NV + i) + (size_t)j * NX] = -sdt * LD[i + (size_t)j * LDV];
    Fxx[(NV + i) + (size_t)(NV + j) * NX] = -sdt * LD[i + (size_t)(NV + j) * LDV] + (i == j ? 1.0 : 0.0);
  }
  if (!impact)
    for (int e = lane; e < NV * NU; e += 64) {
      const int i = e % NV, j = e / NV;
      Fvu[i + (size_t)j * NV] = dt * Lam[i + (size_t)(NP + j) * LDV];
    }
  if (lane < NV) Fx[NV + lane] -= sdt * Lr[lane];

  if (!impact) {
    // ================= switching constraint (:138-153) =================
    if (NS > 0 && ns > 0) {
      wave_gemm(ns, NX, NV, -1.0, Phia, 1, LDS_, LD, 1, LDV, 1.0, Phix, 1, LDS_, lane);
      wave_gemm(ns, NU, NV, 1.0, Phia, 1, LDS_, Lam + (size_t)NP * LDV, 1, LDV, 0.0, Phiu, 1, LDS_, lane);
      if (lane < ns) {
        double acc = 0.0;
        for (int k = 0; k < NV; ++k) acc += Phia[lane + k * LDS_] * Lr[k];
        Phit[lane] = (Phit[lane] - acc) * (1.0 / (double)g.num_grids_in_phase);
        Pres[lane] -= acc;
      }
    }
    __syncthreads();
    // ================= h, and the evalKKT tail scalings (intermediate_stage.cpp:140-148) ========
    const double inv = 1.0 / (double)g.num_grids_in_phase;
    if (lane == 0) {
      double acc = 0.0;
      for (int k = 0; k < nvf; ++k) acc += Lr[k] * haf[k];
      scal[RTOC_KKT_SCAL_H] = (scal[RTOC_KKT_SCAL_H] - acc) * inv;
      const double qtt = scal[RTOC_KKT_SCAL_QTT] * inv * inv;
      scal[RTOC_KKT_SCAL_QTT] = qtt;
      scal[RTOC_KKT_SCAL_QTT_PREV] = -qtt;
    }
    for (int i = lane; i < NX; i += 64) {
      hx[i] *= inv;
      fx[i] *= inv;
    }
    if (lane < NU) hu[lane] *= inv;
  }
  if (stat) atomicOr(&a.status[b], stat);
}

template <int NV, int NU, int NF, int NS>
__global__ __launch_bounds__(64) void expand_kernel(ExpArgs a) {
  constexpr int NX = 2 * NV, NP = NV - NU, LDV = NV + NF, LDS_ = NS > 0 ? NS : 1;
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  const bool impact = g.type == RTOC_GRID_IMPACT;
  const int nf = g.dimf, nvf = NV + nf, ns = impact ? 0 : g.dims;
  const double dt = g.dt;
  double* cr = a.cdd + ((size_t)b * a.nstages + st) * a.cl.stride;
  double* dr = a.dir + ((size_t)b * a.nstages + st) * a.dl.stride;
  const double* dn = dr + a.dl.stride;
  const int* co = a.cl.off;
  const int* dof = a.dl.off;
  const double* Lam = cr + co[RTOC_CDD_MJTJINV];
  const double* LD = cr + co[RTOC_CDD_MJD];
  const double* Lr = cr + co[RTOC_CDD_MJIDC];
  const double* Qafqv = cr + co[RTOC_CDD_QAFQV];
  const double* Qafu = cr + co[RTOC_CDD_QAFU];
  double* laf = cr + co[RTOC_CDD_LAF];
  const double* haf = cr + co[RTOC_CDD_HAF];
  const double* Qxup = cr + co[RTOC_CDD_QXUP];
  const double* Quuptr = cr + co[RTOC_CDD_QUUPTR];
  const double* lup = cr + co[RTOC_CDD_LUP];
  const double* Phia = cr + co[RTOC_CDD_PHIA];
  __shared__ double sdx[NX + 8], sdu[NU + 8], sg[NV + 8], sxi[LDS_ + 8], slaf[LDV + 8];
  for (int i = lane; i < NX; i += 64) sdx[i] = dr[dof[RTOC_DIR_DX] + i];
  if (lane < NU) sdu[lane] = impact ? 0.0 : dr[dof[RTOC_DIR_DU] + lane];
  for (int i = lane; i < NV; i += 64) sg[i] = dn[dof[RTOC_DIR_DLMDGMM] + NV + i];
  if (lane < ns) sxi[lane] = dr[dof[RTOC_DIR_DXI] + lane];
  __syncthreads();
  double dtsv = 0.0;
  if (!impact && g.num_grids_in_phase > 0)
    dtsv = (dr[dof[RTOC_DIR_DTS] + 1] - dr[dof[RTOC_DIR_DTS] + 0]) / (double)g.num_grids_in_phase;
  const bool use_dts = (dtsv < -2.220446049250313e-16 || dtsv > 2.220446049250313e-16);
  // primal (:167-174, impact :83-88) and the laf accumulation of the dual (:190-198, impact :91-95)
  for (int i = lane; i < nvf; i += 64) {
    double acc = 0.0, accl = laf[i];
    for (int j = 0; j < NX; ++j) {
      const double x = sdx[j];
      acc -= LD[i + (size_t)j * LDV] * x;
      accl += Qafqv[i + (size_t)j * LDV] * x;
    }
    if (!impact) {
      for (int j = 0; j < NU; ++j) {
        const double u = sdu[j];
        acc += Lam[i + (size_t)(NP + j) * LDV] * u;
        accl += Qafu[i + (size_t)(NP + j) * LDV] * u;
      }
    }
    acc -= Lr[i];
    if (i >= NV) acc = -acc;
    dr[dof[RTOC_DIR_DAF] + i] = acc;
    if (i < NV) {
      accl += (impact ? 1.0 : dt) * sg[i];
      if (NS > 0 && ns > 0)
        for (int l = 0; l < ns; ++l) accl += Phia[l + (size_t)i * LDS_] * sxi[l];
    }
    if (use_dts) accl += dtsv * haf[i];
    laf[i] = accl;  // the reference updates data.laf() in place as well
    slaf[i] = accl;
  }
  // dnu_passive (:178-188)
  if (!impact && NP > 0 && lane < NP) {
    double acc = -lup[lane];
    for (int j = 0; j < NU; ++j) acc -= Quuptr[lane + j * NP] * sdu[j];
    for (int j = 0; j < NX; ++j) acc -= Qxup[j + (size_t)lane * NX] * sdx[j];
    for (int j = 0; j < NV; ++j) acc -= dt * Lam[lane + (size_t)j * LDV] * sg[j];
    dr[dof[RTOC_DIR_DNUP] + lane] = acc;
  }
  __syncthreads();
  // dbetamu = -MJtJinv * laf (:201, impact :95)
  for (int i = lane; i < nvf; i += 64) {
    double acc = 0.0;
    for (int j = 0; j < nvf; ++j) acc -= Lam[i + (size_t)j * LDV] * slaf[j];
    dr[dof[RTOC_DIR_DBETAMU] + i] = acc;
  }
}

}  // namespace rtoc

static inline int rtoc_con_stride(const rtoc_dims* d) { return 8 * ((d->nc_max * 7 + 7) / 8); }
