// unconstr_riccati.hpp -- the unconstrained (fixed base, no contacts) Riccati recursion in its structured form.
//
// Reference: UnconstrBackwardRiccatiRecursionFactorizer (src/riccati/unconstr_backward_riccati_recursion_factorizer.cpp:
// 27-70), UnconstrRiccatiFactorizer (unconstr_riccati_factorizer.cpp:26-59), UnconstrRiccatiRecursion
// (unconstr_riccati_recursion.cpp:26-48).  With A = [[I, dt I], [0, I]] and B = [0; dt I] the products A^T P+ A, A^T P+ B,
// B^T P+ B are BLOCK ADDS of P+ (the reference never forms A or B); only the policy (LLT of the nv x nv Qaa, K, k) and
// K^T Qaa K are dense, all of size nv.  The general kernels (riccati_backward*.hpp) on materialised A, B -- what
// rtoc_unconstr_backward ran before -- read 2 nx^2 + nv^2 doubles of constants per stage and pad nx = 14 to MFMA tiles.
// Here: one wave per instance, everything of a stage in LDS (nv = 7: 8 KB), elementwise loops over the blocks, one lane
// per right-hand side of the two triangular solves.  Records and results as the general path (KKT record: the Quu slot
// holds Qaa, the lu slot holds la).
#pragma once
#include <cstdint>

#include "device_utils.hpp"
#include "../../include/rtoc.h"

namespace rtoc {

struct UrArgs {
  const double* kkt;
  double* kkt_rw;      // writeback of the mutated Qxx, Qxu, Qaa, la (RTOC_OPT_WRITEBACK_KKT), else unused
  double* ric;
  double* dir;
  const double* dx0;   // [batch][nx] or nullptr
  uint32_t* status;
  int nstages, batch, writeback;
  double dt;
  rtoc_record_layout kl, rl, dl;
};

template <int NV>
__global__ __launch_bounds__(64) void unconstr_riccati_backward_lds_kernel(UrArgs a) {
  constexpr int NX = 2 * NV;
  static_assert(NX + 1 <= 64, "one lane per right-hand side of the policy solve");
  __shared__ double sP[NX * NX], sQ[NX * NX], sH[NX * NV], sG[NV * NV], sL[NV * NV], sK[NV * NX], sGK[NV * NX];
  __shared__ double sS[NX], sFx[NX], sLx[NX], sLa[NV], sKv[NV], sPF[NX];
  __shared__ int sBad;
  const int lane = threadIdx.x, b = blockIdx.x;
  if (b >= a.batch) return;
  const int N = a.nstages - 1;
  const double dt = a.dt, dt2 = a.dt * a.dt;
  const size_t ks = a.kl.stride, rs = a.rl.stride;
  const double* const kb = a.kkt + (size_t)b * a.nstages * ks;
  double* const kw = a.writeback ? a.kkt_rw + (size_t)b * a.nstages * ks : nullptr;
  double* const rb = a.ric + (size_t)b * a.nstages * rs;
  const int oQxx = a.kl.off[RTOC_KKT_QXX], oQxu = a.kl.off[RTOC_KKT_QXU], oQaa = a.kl.off[RTOC_KKT_QUU];
  const int oFx = a.kl.off[RTOC_KKT_FX], oLx = a.kl.off[RTOC_KKT_LX], oLa = a.kl.off[RTOC_KKT_LU];
  const int oP = a.rl.off[RTOC_RIC_P], oS = a.rl.off[RTOC_RIC_S], oK = a.rl.off[RTOC_RIC_K], oKv = a.rl.off[RTOC_RIC_KV];
  uint32_t stat = 0;
  if (lane == 0) sBad = 0;
  // terminal grid point: P = Qxx, s = -lx (unconstr_riccati_recursion.cpp:29-30)
  {
    const double* const q = kb + (size_t)N * ks;
    double* const r = rb + (size_t)N * rs;
    for (int e = lane; e < NX * NX; e += 64) sP[e] = q[oQxx + e], r[oP + e] = sP[e];
    for (int e = lane; e < NX; e += 64) sS[e] = -q[oLx + e], r[oS + e] = sS[e];
  }
  __syncthreads();
  for (int st = N - 1; st >= 0; --st) {
    const double* const q = kb + (size_t)st * ks;
    double* const r = rb + (size_t)st * rs;
    // ---- factorizeKKTMatrix (:27-50): block adds of P+ ----
    for (int e = lane; e < NX * NX; e += 64) {
      const int i = e % NX, j = e / NX;
      double v = q[oQxx + e] + sP[e];
      if (i >= NV) v += dt * sP[(i - NV) + j * NX];
      if (j >= NV) v += dt * sP[i + (j - NV) * NX];
      if (i >= NV && j >= NV) v += dt2 * sP[(i - NV) + (j - NV) * NX];
      sQ[e] = v;
    }
    for (int e = lane; e < NX * NV; e += 64) {
      const int i = e % NX, j = e / NX;
      double v = q[oQxu + e] + dt * sP[i + (NV + j) * NX];
      if (i >= NV) v += dt2 * sP[(i - NV) + (NV + j) * NX];
      sH[e] = v;
    }
    for (int e = lane; e < NV * NV; e += 64) {
      const int i = e % NV, j = e / NV;
      const double v = q[oQaa + e] + dt2 * sP[(NV + i) + (NV + j) * NX];
      sG[e] = v, sL[e] = v;
    }
    for (int e = lane; e < NX; e += 64) sFx[e] = q[oFx + e], sLx[e] = q[oLx + e];
    __syncthreads();
    if (lane < NX) {   // P+ Fx, shared by la and s
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) t += sP[lane + j * NX] * sFx[j];
      sPF[lane] = t;
    }
    __syncthreads();
    if (lane < NV) sLa[lane] = (q[oLa + lane] + dt * sPF[NV + lane]) - dt * sS[NV + lane];
    // ---- LLT(Qaa) (unconstr_riccati_factorizer.cpp:31): right-looking, one column per step ----
    for (int c = 0; c < NV; ++c) {
      __syncthreads();
      const double d = sL[c + c * NV];
      if (!(d > 0.0)) stat |= RTOC_STAT_QUU_NOT_SPD;
      const double piv = sqrt(d > 0.0 ? d : 1.0);
      __syncthreads();
      if (lane >= c && lane < NV) sL[lane + c * NV] = lane == c ? piv : sL[lane + c * NV] / piv;
      __syncthreads();
      for (int e = lane; e < NV * NV; e += 64) {
        const int i = e % NV, j = e / NV;
        if (j > c && i >= j) sL[e] -= sL[i + c * NV] * sL[j + c * NV];
      }
    }
    __syncthreads();
    // ---- K = -Qaa^-1 Qxu^T, k = -Qaa^-1 la (:32-33): lane j solves the right-hand side Qxu(j, :), lane NX solves la ----
    if (lane <= NX) {
      double x[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) x[i] = lane < NX ? sH[lane + i * NX] : sLa[i];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double t = x[i];
#pragma unroll
        for (int l = 0; l < i; ++l) t -= sL[i + l * NV] * x[l];
        x[i] = t / sL[i + i * NV];
      }
#pragma unroll
      for (int i = NV - 1; i >= 0; --i) {
        double t = x[i];
#pragma unroll
        for (int l = i + 1; l < NV; ++l) t -= sL[l + i * NV] * x[l];
        x[i] = t / sL[i + i * NV];
      }
      bool bad = false;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        bad = bad || !(x[i] == x[i]);
        if (lane < NX) sK[i * NX + lane] = -x[i];
        else sKv[i] = -x[i];
      }
      if (bad) sBad = 1;
    }
    __syncthreads();
    // ---- factorizeRiccatiFactorization (:53-70) ----
    for (int e = lane; e < NV * NX; e += 64) {   // GK = Qaa K
      const int i = e % NV, j = e / NV;
      double t = 0.0;
#pragma unroll
      for (int l = 0; l < NV; ++l) t += sG[i + l * NV] * sK[l * NX + j];
      sGK[e] = t;
    }
    __syncthreads();
    for (int e = lane; e < NX * NX; e += 64) {   // Qxx -= K^T GK
      const int i = e % NX, j = e / NX;
      double t = 0.0;
#pragma unroll
      for (int l = 0; l < NV; ++l) t += sK[l * NX + i] * sGK[l + j * NV];
      sQ[e] -= t;
    }
    double snew = 0.0;
    if (lane < NX) {
      snew = sS[lane];
      if (lane >= NV) snew += dt * sS[lane - NV];
      snew -= sPF[lane];
      if (lane >= NV) snew -= dt * sPF[lane - NV];
      snew -= sLx[lane];
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < NV; ++j) t += sH[lane + j * NX] * sKv[j];
      snew -= t;
    }
    __syncthreads();
    if (sBad) stat |= RTOC_STAT_NAN;
    // P = (Qxx + Qxx^T) / 2; records out; the new P+ and s+ replace the old ones
    for (int e = lane; e < NX * NX; e += 64) {
      const int i = e % NX, j = e / NX;
      const double p = 0.5 * (sQ[e] + sQ[j + i * NX]);
      sP[e] = p;
      r[oP + e] = p;
      if (kw) kw[(size_t)st * ks + oQxx + e] = sQ[e];
    }
    if (lane < NX) sS[lane] = snew, r[oS + lane] = snew;
    for (int e = lane; e < NV * NX; e += 64) r[oK + e] = sK[e];
    if (lane < NV) r[oKv + lane] = sKv[lane];
    if (kw) {
      for (int e = lane; e < NX * NV; e += 64) kw[(size_t)st * ks + oQxu + e] = sH[e];
      for (int e = lane; e < NV * NV; e += 64) kw[(size_t)st * ks + oQaa + e] = sG[e];
      if (lane < NV) kw[(size_t)st * ks + oLa + lane] = sLa[lane];
    }
    __syncthreads();
  }
  if (lane == 0 && stat) atomicOr(&a.status[b], stat);
}

// The same recursion with the stage in REGISTERS (small arms, nv <= 8): lane j < nx owns column j of Qxx / P and the
// right-hand side Qxu(j, :) of the policy solve, lane nx owns la.  Because P+ is symmetric, everything a lane needs of P+
// are the two columns j and j - nv, read once from LDS; the block adds, P+ Fx, the policy and K^T (Qaa K) are then
// unrolled register arithmetic.  Every solving lane factorises the nv x nv Qaa redundantly (no hand-offs, no divisions:
// rsqrt + two Newton steps, the inverse diagonal multiplies) -- four block-wide syncs per stage instead of thirty.
template <int NV>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8))) void unconstr_riccati_backward_kernel(UrArgs a) {
  constexpr int NX = 2 * NV;
  static_assert(NX + 1 <= 64 && NV <= 8, "register variant: small arms");
  __shared__ __attribute__((aligned(16))) double sP[NX * NX], sQ[NX * NX], sK[NV * NX], sG[NV * NV];
  __shared__ double sS[NX], sFx[NX], sPF[NX], sKv[NV];
  __shared__ int sBad;
  const int lane = threadIdx.x, b = blockIdx.x;
  if (b >= a.batch) return;
  const int N = a.nstages - 1;
  const double dt = a.dt, dt2 = a.dt * a.dt;
  const size_t ks = a.kl.stride, rs = a.rl.stride;
  const double* const kb = a.kkt + (size_t)b * a.nstages * ks;
  double* const kw = a.writeback ? a.kkt_rw + (size_t)b * a.nstages * ks : nullptr;
  double* const rb = a.ric + (size_t)b * a.nstages * rs;
  const int oQxx = a.kl.off[RTOC_KKT_QXX], oQxu = a.kl.off[RTOC_KKT_QXU], oQaa = a.kl.off[RTOC_KKT_QUU];
  const int oFx = a.kl.off[RTOC_KKT_FX], oLx = a.kl.off[RTOC_KKT_LX], oLa = a.kl.off[RTOC_KKT_LU];
  const int oP = a.rl.off[RTOC_RIC_P], oS = a.rl.off[RTOC_RIC_S], oK = a.rl.off[RTOC_RIC_K], oKv = a.rl.off[RTOC_RIC_KV];
  uint32_t stat = 0;
  if (lane == 0) sBad = 0;
  {
    const double* const q = kb + (size_t)N * ks;
    double* const r = rb + (size_t)N * rs;
    for (int e = lane; e < NX * NX; e += 64) sP[e] = q[oQxx + e], r[oP + e] = sP[e];
    for (int e = lane; e < NX; e += 64) sS[e] = -q[oLx + e], r[oS + e] = sS[e];
  }
  const int j = lane < NX ? lane : 0;        // column owned (lanes >= nx compute along, results unused)
  const int jm = j >= NV ? j - NV : j;       // the column dt-coupled to it
  const bool hi = lane < NX && j >= NV;
  __syncthreads();
  for (int st = N - 1; st >= 0; --st) {
    const double* const q = kb + (size_t)st * ks;
    double* const r = rb + (size_t)st * rs;
    // ---- loads of the stage record (in flight while P+ is read) ----
    double qc[NX], hr[NV];
#pragma unroll
    for (int i = 0; i < NX; ++i) qc[i] = q[oQxx + i + j * NX];
#pragma unroll
    for (int c = 0; c < NV; ++c) hr[c] = lane < NX ? q[oQxu + j + c * NX] : q[oLa + c];
    const double gin = lane < NV * NV ? q[oQaa + lane] : 0.0;
    const double fx = lane < NX ? q[oFx + lane] : 0.0, lx = lane < NX ? q[oLx + lane] : 0.0;
    double pc[NX], pm[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) pc[i] = sP[i + j * NX], pm[i] = sP[i + jm * NX];
    if (lane < NX) sFx[lane] = fx;
    if (lane < NV * NV) sG[lane] = gin + dt2 * sP[(NV + lane % NV) + (NV + lane / NV) * NX];   // Qaa += dt^2 Pvv (:44)
    __syncthreads();
    // ---- factorizeKKTMatrix (:27-50) on the owned column; P+ Fx by symmetry: (P+ Fx)_j = column j . Fx ----
    double pf = 0.0;
#pragma unroll
    for (int i = 0; i < NX; ++i) pf += pc[i] * sFx[i];
    if (lane < NX) sPF[lane] = pf;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double v = qc[i] + pc[i];
      if (i >= NV) v += dt * pc[i - NV];
      if (hi) v += dt * pm[i];
      if (i >= NV && hi) v += dt2 * pm[i - NV];
      if (lane < NX) sQ[i + j * NX] = v;     // parked in LDS while the policy is computed (registers: 4 waves per SIMD)
    }
    // row j of Qxu: + dt P+(j, nv + c) [+ dt^2 P+(j - nv, nv + c)], both read down the owned columns
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      double v = hr[c] + dt * pc[NV + c];
      if (hi) v += dt2 * pm[NV + c];
      if (lane < NX) hr[c] = v;
    }
    __syncthreads();   // sPF complete
    if (lane == NX) {  // la += dt (P+ Fx)_v - dt s+_v (:46-49)
#pragma unroll
      for (int c = 0; c < NV; ++c) hr[c] = (hr[c] + dt * sPF[NV + c]) - dt * sS[NV + c];
    }
    // ---- LLT(Qaa), every lane for itself (lower triangle, the diagonal holds 1 / L_cc); the solves
    //      (unconstr_riccati_factorizer.cpp:31-33) ----
    double Lf[NV * (NV + 1) / 2];
    auto LI = [](int i, int l) constexpr { return i * (i + 1) / 2 + l; };
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int l = 0; l <= i; ++l) Lf[LI(i, l)] = sG[i + l * NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const double d = Lf[LI(c, c)];
      if (!(d > 0.0)) stat |= RTOC_STAT_QUU_NOT_SPD;
      const double ri = rsqrt_d(d > 0.0 ? d : 1.0);
      Lf[LI(c, c)] = ri;
#pragma unroll
      for (int i = c + 1; i < NV; ++i) Lf[LI(i, c)] *= ri;
#pragma unroll
      for (int l = c + 1; l < NV; ++l)
#pragma unroll
        for (int i = l; i < NV; ++i) Lf[LI(i, l)] -= Lf[LI(i, c)] * Lf[LI(l, c)];
    }
    double x[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double t = hr[i];
#pragma unroll
      for (int l = 0; l < i; ++l) t -= Lf[LI(i, l)] * x[l];
      x[i] = t * Lf[LI(i, i)];
    }
#pragma unroll
    for (int i = NV - 1; i >= 0; --i) {
      double t = x[i];
#pragma unroll
      for (int l = i + 1; l < NV; ++l) t -= Lf[LI(l, i)] * x[l];
      x[i] = t * Lf[LI(i, i)];
    }
    bool bad = false;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      x[i] = -x[i];   // K(:, j) resp. k
      bad = bad || !(x[i] == x[i]);
      if (lane < NX) sK[i * NX + lane] = x[i];
      else if (lane == NX) sKv[i] = x[i];
    }
    if (bad && lane <= NX) sBad = 1;
    // GK(:, j) = Qaa K(:, j), Qaa from LDS (the same for every lane: broadcast reads)
    double gk[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      double t = 0.0;
#pragma unroll
      for (int l = 0; l < NV; ++l) t += sG[i + l * NV] * x[l];
      gk[i] = t;
    }
    __syncthreads();   // sK, sKv complete
    // ---- factorizeRiccatiFactorization (:53-70): Qxx(:, j) -= K^T GK(:, j) ----
    double qc2[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      double t = 0.0;
#pragma unroll
      for (int l = 0; l < NV; ++l) t += sK[l * NX + i] * gk[l];
      qc2[i] = sQ[i + j * NX] - t;
    }
#pragma unroll
    for (int i = 0; i < NX; ++i)
      if (lane < NX) sQ[i + j * NX] = qc2[i];
    double snew = 0.0;
    if (lane < NX) {
      snew = sS[lane];
      if (hi) snew += dt * sS[lane - NV];
      snew -= pf;
      if (hi) snew -= dt * sPF[lane - NV];
      snew -= lx;
      double t = 0.0;
#pragma unroll
      for (int c = 0; c < NV; ++c) t += hr[c] * sKv[c];
      snew -= t;
    }
    if (sBad) stat |= RTOC_STAT_NAN;
    __syncthreads();   // sQ complete; every read of sP, sS, sPF of this stage done
    // P = (Qxx + Qxx^T) / 2 (:58), column j against row j
#pragma unroll
    for (int i = 0; i < NX; ++i)
      if (lane < NX) sP[i + j * NX] = 0.5 * (qc2[i] + sQ[j + i * NX]);
    if (lane < NX) sS[lane] = snew, r[oS + lane] = snew;
    if (lane < NV) r[oKv + lane] = sKv[lane];
    for (int e = lane; e < NV * NX; e += 64) r[oK + e] = sK[e];
    if (kw) {
      double* const w = kw + (size_t)st * ks;
      for (int e = lane; e < NX * NX; e += 64) w[oQxx + e] = sQ[e];
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        if (lane < NX) w[oQxu + j + c * NX] = hr[c];
        else if (lane == NX) w[oLa + c] = hr[c];
      }
      if (lane < NV * NV) w[oQaa + lane] = sG[lane];
    }
    __syncthreads();   // new P+ complete
    for (int e = lane; e < NX * NX; e += 64) r[oP + e] = sP[e];
  }
  if (lane == 0 && stat) atomicOr(&a.status[b], stat);
}

// forwardRiccatiRecursion + computeCostateDirection (unconstr_riccati_factorizer.cpp:37-57, unconstr_riccati_recursion.cpp:
// 37-48): da = K dx + k, dx+ = Fx + dx, dq+ += dt dv, dv+ += dt da, dlmdgmm = P dx - s
template <int NV>
__global__ __launch_bounds__(64) void unconstr_riccati_forward_kernel(UrArgs a) {
  constexpr int NX = 2 * NV;
  __shared__ double sP[NX * NX], sK[NV * NX], sDx[NX], sDa[NV];
  const int lane = threadIdx.x, b = blockIdx.x;
  if (b >= a.batch) return;
  const int N = a.nstages - 1;
  const double dt = a.dt;
  const size_t ks = a.kl.stride, rs = a.rl.stride, ds = a.dl.stride;
  const double* const kb = a.kkt + (size_t)b * a.nstages * ks;
  const double* const rb = a.ric + (size_t)b * a.nstages * rs;
  double* const db = a.dir + (size_t)b * a.nstages * ds;
  const int oFx = a.kl.off[RTOC_KKT_FX];
  const int oP = a.rl.off[RTOC_RIC_P], oS = a.rl.off[RTOC_RIC_S], oK = a.rl.off[RTOC_RIC_K], oKv = a.rl.off[RTOC_RIC_KV];
  const int oDx = a.dl.off[RTOC_DIR_DX], oDu = a.dl.off[RTOC_DIR_DU], oDl = a.dl.off[RTOC_DIR_DLMDGMM];
  if (lane < NX) sDx[lane] = a.dx0 ? a.dx0[(size_t)b * NX + lane] : db[oDx + lane];
  for (int st = 0; st <= N; ++st) {
    const double* const r = rb + (size_t)st * rs;
    double* const d = db + (size_t)st * ds;
    for (int e = lane; e < NX * NX; e += 64) sP[e] = r[oP + e];
    if (st < N)
      for (int e = lane; e < NV * NX; e += 64) sK[e] = r[oK + e];
    __syncthreads();
    if (lane < NX) {
      d[oDx + lane] = sDx[lane];
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) t += sP[lane + j * NX] * sDx[j];
      d[oDl + lane] = t - r[oS + lane];
    }
    if (st < N && lane < NV) {
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < NX; ++j) t += sK[lane * NX + j] * sDx[j];
      t += r[oKv + lane];
      sDa[lane] = t;
      d[oDu + lane] = t;
    }
    __syncthreads();
    double x = 0.0;
    if (st < N && lane < NX) {
      const double* const q = kb + (size_t)st * ks;
      x = q[oFx + lane] + sDx[lane];
      x += dt * (lane < NV ? sDx[NV + lane] : sDa[lane - NV]);
    }
    __syncthreads();
    if (st < N && lane < NX) sDx[lane] = x;
    __syncthreads();
  }
}

}  // namespace rtoc
