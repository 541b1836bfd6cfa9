// riccati_forward.hpp -- batched forward Riccati recursion for gfx950.
//
// Replaces RiccatiRecursion::forwardRiccatiRecursion (reference
// src/riccati/riccati_recursion.cpp:83-131) and the free functions it calls
// (src/riccati/riccati_factorizer.cpp:200-277): forwardRiccatiRecursion (x2),
// computeSwitchingTimeDirection, computeCostateDirection (x2),
// computeLagrangeMultiplierDirection.
//
// The forward pass is a chain of mat-vecs (about 0.24 flop/byte): HBM-bound.  One
// workgroup of NWF wavefronts per OCP instance; thread t < NX owns row t of Fxx and P,
// threads NX..NX+NU-1 own a row of K.  For a fixed column j consecutive threads read
// consecutive doubles of Fxx / P (coalesced); the per-instance chain dependency is only
// through the NX-vector dx kept in LDS, many instances per CU hide the load latency.
#pragma once
#include "device_utils.hpp"
#include "../../include/rtoc.h"

namespace rtoc {

#ifndef FWD_UNROLL
#define FWD_UNROLL 18
#endif

struct FwdArgs {
  const double* kkt;
  const double* ric;
  double* dir;
  const double* dx0;  // [batch][nx] or nullptr (then dir[...][0].dx is used as given)
  const rtoc_grid* grid;
  int nstages;
  int batch;  // instances [first, batch) are processed by this launch
  int first;
};

template <int NV, int NU, int NS, int NWF>
__global__ __launch_bounds__(64 * NWF) void riccati_forward_kernel(FwdArgs a) {
  constexpr int NX = 2 * NV, NT = 64 * NWF;
  static_assert(NX + NU <= NT, "one thread per row of [Fxx;K]");
  __shared__ double sDx[2][NX + 8];
  __shared__ double sDu[NU + 8];
  __shared__ double sRed[8];
  const int tid = threadIdx.x;
  const int b = a.first + blockIdx.x;
  if (b >= a.batch) return;
  const int N = a.nstages - 1;
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout KL = SL.kkt, RL = SL.ric, DL = SL.dir;
  const double* kb = a.kkt + (size_t)b * a.nstages * KL.stride;
  const double* rb = a.ric + (size_t)b * a.nstages * RL.stride;
  double* db = a.dir + (size_t)b * a.nstages * DL.stride;

  if (tid < NX) {
    const double v = a.dx0 ? a.dx0[(size_t)b * NX + tid] : db[DL.off[RTOC_DIR_DX] + tid];
    sDx[0][tid] = v;
    if (a.dx0) db[DL.off[RTOC_DIR_DX] + tid] = v;
  }
  __syncthreads();
  double dts = 0.0, dtsn = 0.0;  // d[i].dts, d[i].dts_next carried along (uniform)
  {
    const rtoc_grid g0 = a.grid[0];
    if (g0.sto) {
      // computeSwitchingTimeDirection(sto_policy_[0], d[0], false)  (riccati_recursion.cpp:91-94)
      if (tid == 0) {
        double acc = 0.0;
        for (int k = 0; k < NX; ++k) acc += rb[RL.off[RTOC_RIC_DTSDX] + k] * sDx[0][k];
        sRed[0] = acc + rb[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTS0];
      }
      __syncthreads();
      dtsn = sRed[0];
      __syncthreads();
    }
  }
  int cur = 0;
  for (int st = 0; st < N; ++st) {
    const rtoc_grid g = a.grid[st];
    const bool impact = g.type == RTOC_GRID_IMPACT, lift = g.type == RTOC_GRID_LIFT;
    const bool sto = g.sto != 0, sto_next = g.sto_next != 0;
    const double* kr = kb + (size_t)st * KL.stride;
    const double* rr = rb + (size_t)st * RL.stride;
    double* dr = db + (size_t)st * DL.stride;
    const double* dx = sDx[cur];
    double* dxn = sDx[cur ^ 1];

    if (impact || lift) {
      dts = dtsn;  // d[i].dts = d[i-1].dts_next
      dtsn = 0.0;
      if (lift && sto_next) {
        if (tid == 0) {
          double acc = 0.0;
          for (int k = 0; k < NX; ++k) acc += rr[RL.off[RTOC_RIC_DTSDX] + k] * dx[k];
          acc += rr[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTS0];
          if (sto) acc += rr[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTSDTS] * dts;
          sRed[0] = acc;
        }
        __syncthreads();
        dtsn = sRed[0];
        __syncthreads();
      }
    }

    // ---- row products: threads < NX: Fxx dx and P dx ; threads NX..NX+NU-1: K dx ----
    double acc_a = 0.0, acc_p = 0.0;
    if (tid < NX) {
      const double* A = kr + KL.off[RTOC_KKT_FXX] + tid;
      const double* P = rr + RL.off[RTOC_RIC_P] + tid;
#pragma unroll FWD_UNROLL
      for (int j = 0; j < NX; ++j) {
        const double x = dx[j];
        acc_a += A[j * NX] * x;
        acc_p += P[j * NX] * x;
      }
    } else if (!impact && tid < NX + NU) {
      const int u = tid - NX;
      const double* K = rr + RL.off[RTOC_RIC_K] + (size_t)u * NX;  // row u of row-major K
#pragma unroll 6
      for (int j = 0; j < NX; ++j) acc_a += K[j] * dx[j];
      double du = acc_a + rr[RL.off[RTOC_RIC_KV] + u];
      if (sto) {
        du += rr[RL.off[RTOC_RIC_T] + u] * (dtsn - dts);
        if (sto_next) du -= rr[RL.off[RTOC_RIC_W] + u] * dtsn;
      }
      sDu[u] = du;
      dr[DL.off[RTOC_DIR_DU] + u] = du;
    }
    __syncthreads();
    if (tid < NX) {
      double v = kr[KL.off[RTOC_KKT_FX] + tid] + acc_a;
      if (!impact) {
        if (tid >= NV) {
          const double* Bv = kr + KL.off[RTOC_KKT_FVU] + (tid - NV);
#pragma unroll 4
          for (int u = 0; u < NU; ++u) v += Bv[u * NV] * sDu[u];
        }
        if (sto) v += kr[KL.off[RTOC_KKT_FFX] + tid] * (dtsn - dts);
      }
      dxn[tid] = v;
      (dr + DL.stride)[DL.off[RTOC_DIR_DX] + tid] = v;
    }
    if (impact && sto_next) {
      // riccati_recursion.cpp:101-107: dts_next of d[i+1] from sto_policy_[i] and dx[i+1]
      __syncthreads();
      if (tid == 0) {
        double acc = 0.0;
        for (int k = 0; k < NX; ++k) acc += rr[RL.off[RTOC_RIC_DTSDX] + k] * dxn[k];
        acc += rr[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTS0];
        if (sto) acc += rr[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTSDTS] * dts;
        sRed[0] = acc;
      }
      __syncthreads();
      dtsn = sRed[0];
    }
    // ---- costate (riccati_factorizer.cpp:243-262) ----
    if (tid < NX) {
      double lam = acc_p - rr[RL.off[RTOC_RIC_S] + tid];
      if (sto) {
        if (impact) {
          lam -= rr[RL.off[RTOC_RIC_PHI] + tid] * dtsn;
        } else {
          lam += rr[RL.off[RTOC_RIC_PSI] + tid] * (dtsn - dts);
          if (sto_next) lam -= rr[RL.off[RTOC_RIC_PHI] + tid] * dtsn;
        }
      }
      dr[DL.off[RTOC_DIR_DLMDGMM] + tid] = lam;
    }
    // ---- switching-constraint multiplier (:265-277) ----
    if (NS > 0 && g.switching_constraint && tid < g.dims) {
      const double* M = rr + RL.off[RTOC_RIC_M] + tid;
      double acc = 0.0;
      for (int j = 0; j < NX; ++j) acc += M[j * NS] * dx[j];
      acc += rr[RL.off[RTOC_RIC_MV] + tid];
      if (sto) {
        acc += rr[RL.off[RTOC_RIC_MT] + tid] * (dtsn - dts);
        if (sto_next) acc -= rr[RL.off[RTOC_RIC_MTN] + tid] * dtsn;
      }
      dr[DL.off[RTOC_DIR_DXI] + tid] = acc;
    }
    if (tid == 0) {
      dr[DL.off[RTOC_DIR_DTS] + 0] = dts;
      dr[DL.off[RTOC_DIR_DTS] + 1] = dtsn;
    }
    __syncthreads();
    cur ^= 1;
  }
  // terminal costate (riccati_recursion.cpp:128-130)
  {
    const double* rr = rb + (size_t)N * RL.stride;
    double* dr = db + (size_t)N * DL.stride;
    const double* dx = sDx[cur];
    if (tid < NX) {
      const double* P = rr + RL.off[RTOC_RIC_P] + tid;
      double acc = 0.0;
#pragma unroll 6
      for (int j = 0; j < NX; ++j) acc += P[j * NX] * dx[j];
      dr[DL.off[RTOC_DIR_DLMDGMM] + tid] = acc - rr[RL.off[RTOC_RIC_S] + tid];
    }
    if (tid == 0) {
      dr[DL.off[RTOC_DIR_DTS] + 0] = dts;
      dr[DL.off[RTOC_DIR_DTS] + 1] = dtsn;
    }
  }
}

// Unconstrained (fixed-base, no contact) OCPs reuse the general kernels: the structured
// A = [[I, dt I],[0, I]], Bv = dt I of unconstr_backward_riccati_recursion_factorizer.cpp:27-50
// are materialised once into the Fxx / Fvu slots of every record.
struct FillArgs {
  double* kkt;
  int nstages, batch;
  double dt;
  rtoc_record_layout kl;
};

template <int NV>
__global__ void unconstr_fill_kernel(FillArgs a) {
  constexpr int NX = 2 * NV;
  const int rec = blockIdx.x;  // instance*nstages + stage
  if (rec >= a.batch * a.nstages) return;
  double* r = a.kkt + (size_t)rec * a.kl.stride;
  double* A = r + a.kl.off[RTOC_KKT_FXX];
  double* Bv = r + a.kl.off[RTOC_KKT_FVU];
  for (int e = threadIdx.x; e < NX * NX; e += blockDim.x) {
    const int i = e % NX, j = e / NX;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < NV && j == i + NV) v = a.dt;
    A[e] = v;
  }
  for (int e = threadIdx.x; e < NV * NV; e += blockDim.x) {
    const int i = e % NV, j = e / NV;
    Bv[e] = (i == j) ? a.dt : 0.0;
  }
}

}  // namespace rtoc
