// unconstr_dynamics.hpp -- condensation / expansion of the inverse-dynamics constraint of the
// unconstrained (fixed-base, contact-free) OCP.
//
// Replaces UnconstrDynamics::condenseUnconstrDynamics / expandPrimal / expandDual
// (reference src/dynamics/unconstr_dynamics.cpp:67-104).  The unconstrained Riccati recursion uses
// the acceleration as the control (unconstr_riccati_factorizer.cpp:26-59), so in the packed records
//   KKT.Quu := Qaa, KKT.lu := la, KKT.Qxu := [Qqa; Qva]  ("this is actually Qqa and Qqv", :86-88),
// and the torque-level quantities ride in the ContactDynamicsData record:
//   CDD.dIDCdqv = [dID_dq | dID_dv], CDD.dIDda = dID_da, CDD.IDC = ID,
//   CDD.Qaa = diag(Quu) of the torque cost (only the diagonal enters the condensation, :71,:76-78),
//   CDD.MJtJinv = Quu itself (nv x nv; only its off-diagonal entries are read, by expandDual :99-104; zero = diagonal cost),
//   CDD.la  = lu of the torque cost.
// After rtoc_unconstr_expand the direction record reads like the reference's SplitDirection:
//   DIR.daf = da (the Riccati control), DIR.du = torque direction, DIR.dbetamu = dbeta.
// O(nv^3) flops on 7x7 blocks: one wave per grid point, VALU only.
#pragma once
#include "device_utils.hpp"
#include "../../include/rtoc.h"

namespace rtoc {

struct UdArgs {
  double* kkt;
  double* cdd;
  double* dir;
  int nstages, batch;
  double dt;
  rtoc_record_layout kl, cl, dl;
};

template <int NV>
__global__ __launch_bounds__(64) void unconstr_condense_kernel(UdArgs a) {
  constexpr int NX = 2 * NV;
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;  // the terminal stage has no dynamics
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  double* kr = a.kkt + ((size_t)b * a.nstages + st) * a.kl.stride;
  const double* cr = a.cdd + ((size_t)b * a.nstages + st) * a.cl.stride;
  __shared__ double dq[NV * NV], dv[NV * NV], da[NV * NV], w[NV], luc[NV];
  for (int e = lane; e < NV * NV; e += 64) {
    dq[e] = cr[a.cl.off[RTOC_CDD_DIDCDQV] + e];
    dv[e] = cr[a.cl.off[RTOC_CDD_DIDCDQV] + NV * NV + e];
    da[e] = cr[a.cl.off[RTOC_CDD_DIDDA] + e];
  }
  if (lane < NV) {
    const double quu = cr[a.cl.off[RTOC_CDD_QAA] + lane];
    w[lane] = quu;
    // lu_condensed = lu + diag(Quu) ID (:70-71)
    luc[lane] = cr[a.cl.off[RTOC_CDD_LA] + lane] + quu * cr[a.cl.off[RTOC_CDD_IDC] + lane];
  }
  __syncthreads();
  double* Qxx = kr + a.kl.off[RTOC_KKT_QXX];
  double* Qxu = kr + a.kl.off[RTOC_KKT_QXU];
  double* Quu = kr + a.kl.off[RTOC_KKT_QUU];
  double* lx = kr + a.kl.off[RTOC_KKT_LX];
  double* lu = kr + a.kl.off[RTOC_KKT_LU];
  // gradients (:72-74)
  for (int j = lane; j < 3 * NV; j += 64) {
    const double* m = j < NV ? dq : (j < 2 * NV ? dv : da);
    const int c = j % NV;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < NV; ++i) acc += m[i + c * NV] * luc[i];
    if (j < 2 * NV)
      lx[j] += acc;
    else
      lu[c] += acc;
  }
  // Hessian blocks (:76-88): X^T (diag(Quu) Y)
  auto xtwy = [&](const double* X, int r, const double* Y, int c) {
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < NV; ++i) acc += X[i + r * NV] * (w[i] * Y[i + c * NV]);
    return acc;
  };
  for (int e = lane; e < NV * NV; e += 64) {
    const int r = e % NV, c = e / NV;
    const double qqv = Qxx[r + (size_t)(NV + c) * NX] + xtwy(dq, r, dv, c);
    Qxx[r + (size_t)c * NX] += xtwy(dq, r, dq, c);                  // Qqq (:79)
    Qxx[r + (size_t)(NV + c) * NX] = qqv;                            // Qqv (:80)
    Qxx[(NV + c) + (size_t)r * NX] = qqv;                            // Qvq = Qqv^T (:81)
    Qxx[(NV + r) + (size_t)(NV + c) * NX] += xtwy(dv, r, dv, c);    // Qvv (:82)
    Quu[r + (size_t)c * NV] += xtwy(da, r, da, c);                  // Qaa (:83)
    // (diag(Quu) dID_dq)^T dID_da: same products, the weight sits on the left factor (:86-87)
    double aq = 0.0, av = 0.0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      aq += (w[i] * dq[i + r * NV]) * da[i + c * NV];
      av += (w[i] * dv[i + r * NV]) * da[i + c * NV];
    }
    Qxu[r + (size_t)c * NX] = aq;         // Qqu (:86)
    Qxu[(NV + r) + (size_t)c * NX] = av;  // Qvu (:87)
  }
}

template <int NV>
__global__ __launch_bounds__(64) void unconstr_expand_kernel(UdArgs a) {
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  const double* cr = a.cdd + ((size_t)b * a.nstages + st) * a.cl.stride;
  double* dr = a.dir + ((size_t)b * a.nstages + st) * a.dl.stride;
  __shared__ double x[3 * NV];
  if (lane < 2 * NV) x[lane] = dr[a.dl.off[RTOC_DIR_DX] + lane];
  if (lane < NV) {
    const double dacc = dr[a.dl.off[RTOC_DIR_DU] + lane];  // the Riccati control is da
    x[2 * NV + lane] = dacc;
    dr[a.dl.off[RTOC_DIR_DAF] + lane] = dacc;
  }
  __syncthreads();
  if (lane < NV) {
    // expandPrimal (:91-96): du = ID + dID_dq dq + dID_dv dv + dID_da da
    double du = cr[a.cl.off[RTOC_CDD_IDC] + lane];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) acc += cr[a.cl.off[RTOC_CDD_DIDCDQV] + lane + k * NV] * x[k];
    du += acc;
    acc = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) acc += cr[a.cl.off[RTOC_CDD_DIDCDQV] + NV * NV + lane + k * NV] * x[NV + k];
    du += acc;
    acc = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) acc += cr[a.cl.off[RTOC_CDD_DIDDA] + lane + k * NV] * x[2 * NV + k];
    du += acc;
    dr[a.dl.off[RTOC_DIR_DU] + lane] = du;
    x[lane] = du;  // dx is consumed: du of all joints for the off-diagonal part below
  }
  __syncthreads();
  if (lane < NV) {
    // expandDual (:99-104): dbeta = (lu + Quu du) / dt with the FULL torque-cost Hessian: its diagonal is CDD.Qaa (all
    // the condensation uses, :71,:76-78), its strictly off-diagonal part is read from the MJTJINV field (nv x nv,
    // column-major), which the unconstrained path does not use otherwise and which is zero unless the caller fills
    // it -- robotoc's own costs have a diagonal Quu
    const double du = x[lane];
    double off = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k)
      off += (k == lane) ? 0.0 : cr[a.cl.off[RTOC_CDD_MJTJINV] + lane + k * NV] * x[k];
    dr[a.dl.off[RTOC_DIR_DBETAMU] + lane] =
        (cr[a.cl.off[RTOC_CDD_LA] + lane] + (cr[a.cl.off[RTOC_CDD_QAA] + lane] * du + off)) / a.dt;
  }
}

}  // namespace rtoc
