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
#ifndef FWD_REST_PARTS
#define FWD_REST_PARTS 2  // the remaining columns are walked in this many load batches
#endif
#ifndef FWD_HEAD
#define FWD_HEAD 8  // columns of the next stage requested ahead of the tail of this one
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

// Hand-over of dx / du / dts between the threads of an instance goes through LDS only: one wavefront needs no barrier
// at all (its LDS operations complete in order), several need s_barrier behind their own LDS traffic -- and neither
// needs the s_waitcnt vmcnt(0) that __syncthreads() puts in front of it, which would drain the loads in flight.
template <int NWF>
__device__ __forceinline__ void fwd_sync() {
  if constexpr (NWF == 1)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NV, int NU, int NS, int NWF>
__global__ __launch_bounds__(64 * NWF, 4) void riccati_forward_kernel(FwdArgs a) {
  constexpr int NX = 2 * NV, NT = 64 * NWF;
  constexpr bool MFOLD = NS > 0 && NX + NU + NS <= NT;  // the rows of M have threads of their own
  static_assert(NX + NU <= NT, "one thread per row of [Fxx;K]");
  __shared__ double sDx[2][NX + 8];
  __shared__ double sDu[NU + 8];
  __shared__ double sRed[8];
  extern __shared__ int sGridTab[];  // [nstages] (dynamic): type | sto << 4 | sto_next << 5 | switching_constraint << 6 | dims << 8
  const int tid = threadIdx.x;
  const int b = a.first + blockIdx.x;
  if (b >= a.batch) return;
  for (int st = tid; st < a.nstages; st += NT) {
    const rtoc_grid* gp = a.grid + st;
    sGridTab[st] = (gp->type & 15) | ((gp->sto != 0) << 4) | ((gp->sto_next != 0) << 5) |
                   ((gp->switching_constraint != 0) << 6) | (gp->dims << 8);
  }
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
  struct GridBits { int type, sto, sto_next, switching_constraint, dims; };
  // ---- Row walks.  Every HBM read of a stage that does not depend on dx: the rows of Fxx / P (threads < NX), of K
  //      (threads NX..NX+NU-1) and of M (the NS threads behind them, switching-constraint stages) share the SAME two load
  //      instructions per column -- a divergent branch per role used to serialise them into ~13 dependent round trips per
  //      stage --, Fx / k / m and s ride along.  Threads without a role read element 0 of a live field; nothing consumes
  //      what they load.  The first FWD_HEAD columns (and the two vectors) of stage st + 1 are requested BEFORE the tail
  //      of stage st (du -> LDS -> dx+ -> LDS, stores): all 4096 instances run in step, so without that the memory system
  //      idles during the tail of every stage (tools/probes/read_bw_probe.hip: the walks alone stream at 5.6 TB/s). ----
  struct StageRows {
    const double *pa, *pp, *pv, *ps;
    int sa;
  };
  auto stage_rows = [&](int st) {
    const int gb = __builtin_amdgcn_readfirstlane(sGridTab[st]);
    const bool imp = (gb & 15) == RTOC_GRID_IMPACT;
    const bool xrow = tid < NX;
    const bool krow = !imp && tid >= NX && tid < NX + NU;
    const bool mrow = MFOLD && ((gb >> 6) & 1) && tid >= NX + NU && tid < NX + NU + (gb >> 8);
    const int u = krow ? tid - NX : 0;
    const int m = mrow ? tid - NX - NU : 0;
    const int t = xrow ? tid : 0;
    const double* kr = kb + (size_t)st * KL.stride;
    const double* rr = rb + (size_t)st * RL.stride;
    StageRows r;
    r.pa = xrow ? kr + KL.off[RTOC_KKT_FXX] + tid
                : (mrow ? rr + RL.off[RTOC_RIC_M] + m : rr + RL.off[RTOC_RIC_K] + (size_t)u * NX);
    r.sa = xrow ? NX : (mrow ? NS : 1);
    r.pp = rr + RL.off[RTOC_RIC_P] + t;
    r.pv = xrow ? kr + KL.off[RTOC_KKT_FX] + tid : (mrow ? rr + RL.off[RTOC_RIC_MV] + m : rr + RL.off[RTOC_RIC_KV] + u);
    r.ps = rr + RL.off[RTOC_RIC_S] + t;
    return r;
  };
  constexpr int H = NX < FWD_HEAD ? NX : FWD_HEAD;
  double ha[H], hp[H], vec, sv;
  {
    const StageRows r0 = stage_rows(0);
#pragma unroll
    for (int j = 0; j < H; ++j) {
      ha[j] = r0.pa[j * r0.sa];
      hp[j] = r0.pp[j * NX];
    }
    vec = *r0.pv;  // Fx[t] | k[u] | m[m]
    sv = *r0.ps;
  }
  for (int st = 0; st < N; ++st) {
    // grid descriptor from the LDS table: a global read here waits, on gfx9's single vector-memory counter, for the
    // acknowledgement of the previous stage's stores as well -- a write round trip per stage with nothing else in flight
    const int gb = __builtin_amdgcn_readfirstlane(sGridTab[st]);
    const GridBits g = {gb & 15, (gb >> 4) & 1, (gb >> 5) & 1, (gb >> 6) & 1, gb >> 8};
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
        fwd_sync<NWF>();
        dtsn = sRed[0];
        fwd_sync<NWF>();
      }
    }

    const bool xrow = tid < NX;
    const bool krow = !impact && tid >= NX && tid < NX + NU;
    const bool mrow = MFOLD && g.switching_constraint && tid >= NX + NU && tid < NX + NU + g.dims;
    const int u = krow ? tid - NX : 0;
    const int m = mrow ? tid - NX - NU : 0;
    const StageRows r = stage_rows(st);
    double bv[NU];
    double acc_a = 0.0, acc_p = 0.0;
#pragma unroll
    for (int j = 0; j < H; ++j) {  // the head, in flight since the previous stage
      const double x = dx[j];
      acc_a += ha[j] * x;
      acc_p += hp[j] * x;
    }
#pragma unroll((NX - H + FWD_REST_PARTS - 1) / FWD_REST_PARTS > 0 ? (NX - H + FWD_REST_PARTS - 1) / FWD_REST_PARTS : 1)
    for (int j = H; j < NX; ++j) {
      const double x = dx[j];
      acc_a += r.pa[j * r.sa] * x;
      acc_p += r.pp[j * NX] * x;
    }
    {  // the row of Fvu, needed by the tail: behind the last column, in front of the next head
      const double* Bv = kr + KL.off[RTOC_KKT_FVU] + ((tid >= NV && tid < NX) ? tid - NV : 0);
#pragma unroll
      for (int c = 0; c < NU; ++c) bv[c] = Bv[c * NV];
    }
    double vec_n, sv_n;
    {
      const StageRows rn = stage_rows(st + 1);  // st + 1 <= N: the terminal records exist, what is read there is dropped
#pragma unroll
      for (int j = 0; j < H; ++j) {
        ha[j] = rn.pa[j * rn.sa];
        hp[j] = rn.pp[j * NX];
      }
      vec_n = *rn.pv;
      sv_n = *rn.ps;
    }
    if (krow) {
      double du = acc_a + vec;
      if (sto) {
        du += rr[RL.off[RTOC_RIC_T] + u] * (dtsn - dts);
        if (sto_next) du -= rr[RL.off[RTOC_RIC_W] + u] * dtsn;
      }
      sDu[u] = du;
      dr[DL.off[RTOC_DIR_DU] + u] = du;
    }
    fwd_sync<NWF>();
    if (xrow) {
      double v = vec + acc_a;
      if (!impact) {
        if (tid >= NV) {
#pragma unroll
          for (int c = 0; c < NU; ++c) v += bv[c] * sDu[c];
        }
        if (sto) v += kr[KL.off[RTOC_KKT_FFX] + tid] * (dtsn - dts);
      }
      dxn[tid] = v;
      (dr + DL.stride)[DL.off[RTOC_DIR_DX] + tid] = v;
    }
    if (impact && sto_next) {
      // riccati_recursion.cpp:101-107: dts_next of d[i+1] from sto_policy_[i] and dx[i+1]
      fwd_sync<NWF>();
      if (tid == 0) {
        double acc = 0.0;
        for (int k = 0; k < NX; ++k) acc += rr[RL.off[RTOC_RIC_DTSDX] + k] * dxn[k];
        acc += rr[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTS0];
        if (sto) acc += rr[RL.off[RTOC_RIC_SCAL] + RTOC_RIC_SCAL_DTSDTS] * dts;
        sRed[0] = acc;
      }
      fwd_sync<NWF>();
      dtsn = sRed[0];
    }
    // ---- costate (riccati_factorizer.cpp:243-262) ----
    if (tid < NX) {
      double lam = acc_p - sv;
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
    if constexpr (MFOLD) {
      if (mrow) {
        double acc = acc_a + vec;
        if (sto) {
          acc += rr[RL.off[RTOC_RIC_MT] + m] * (dtsn - dts);
          if (sto_next) acc -= rr[RL.off[RTOC_RIC_MTN] + m] * dtsn;
        }
        dr[DL.off[RTOC_DIR_DXI] + m] = acc;
      }
    } else if (NS > 0 && g.switching_constraint && tid < g.dims) {
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
    fwd_sync<NWF>();
    cur ^= 1;
    vec = vec_n;
    sv = sv_n;
  }
  // terminal costate (riccati_recursion.cpp:128-130)
  {
    const double* rr = rb + (size_t)N * RL.stride;
    double* dr = db + (size_t)N * DL.stride;
    const double* dx = sDx[cur];
    if (tid < NX) {
      const double* P = rr + RL.off[RTOC_RIC_P] + tid;
      double acc = 0.0;
#pragma unroll FWD_UNROLL
      for (int j = 0; j < NX; ++j) acc += P[j * NX] * dx[j];
      dr[DL.off[RTOC_DIR_DLMDGMM] + tid] = acc - rr[RL.off[RTOC_RIC_S] + tid];
    }
    if (tid == 0) {
      dr[DL.off[RTOC_DIR_DTS] + 0] = dts;
      dr[DL.off[RTOC_DIR_DTS] + 1] = dtsn;
    }
  }
}

// Instances whose rows need more than one wavefront (NX + NU > 64: iCub): the plain form -- thread t < NX owns row t of
// Fxx and P, threads NX..NX+NU-1 a row of K, hardware barriers between the phases.  The single-wave kernel above relies on
// the in-order LDS of ONE wave for its hand-overs and on its roles sharing a wave's load instructions; with two waves the
// same structure measured slower than this one (iCub nv = 32: 1.51 vs 0.77 ms per 1024 x 35 stages), so it is kept.
template <int NV, int NU, int NS, int NWF>
__global__ __launch_bounds__(64 * NWF) void riccati_forward_mw_kernel(FwdArgs a) {
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
