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
// headline workload.  Split pipeline (default): mjtjinv_kernel -- one wave per work item, the saddle-matrix
// factorisations and the cone rows, MJtJinv into the ContactDynamicsData record -- then condense_kernel<SPLIT> -- three
// waves per work item, every input field once into LDS, ~15 barrier-separated phases of compile-time-shaped f64-MFMA
// products (lds_gemm.hpp), the Hessian blocks read-modify-written once in HBM.  Both kernels are occupancy x chain
// bound, so their LDS carves and register budgets are sized by work items per CU (160 KB LDS in granules of 1280 B;
// ANYmal: 10 and 5 work items per CU, CondCfg / MjCfg below).  The intermediate blocks the reference keeps for the
// expansion (MJtJinv, MJtJinv_dIDCdqv; Qafqv / Qafu_full on request) are produced in the contact-dynamics record.
#pragma once
#include "device_utils.hpp"
#include "riccati_backward.hpp"  // wave_llt, llt_solve_reg
#include "lds_gemm.hpp"
#include "../../include/rtoc.h"

// Wavefronts per (instance, grid point) of the condensation kernel.  Measured on MI355X (4096 ANYmal x 46 grid
// points / 512 iCub): 2 waves 6.12 / 3.50 / 5.24 ms, 3 waves 5.90 / 3.20 / 4.54 ms (the 3 x 3 tile grids of the
// Schur updates deal evenly to three waves), 4 waves 7.09 ms.
#ifndef RTOC_COND_NW
#define RTOC_COND_NW 3
#endif

#include "friction_cone.hpp"

namespace rtoc {

__device__ __forceinline__ void wave_lds_sync_() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

#ifdef RTOC_ENABLE_PROF
#define RTOC_CPROF(k)                                                                   \
  do {                                                                                  \
    if (a.prof && blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0) a.prof[(k)] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#else
#define RTOC_CPROF(k) do { } while (0)
#endif

struct CondArgs {
  double* kkt;
  double* cdd;
  const rtoc_grid* grid;
  uint32_t* status;
  int nstages, batch;
  double damping;  // RobotModelInfo::contact_inv_damping (robot_model_info.hpp:95)
  long long* prof;  // optional cycle stamps of work item 0 (tuning aid)
  double* con;               // constraint records or nullptr
  const rtoc_box_row* rows;  // [nrows] joint-limit rows (device)
  const int* entry;          // CSR of the rows per primal entry: [2nv+nu+1] offsets, then row ids
  const int4* pair;          // per primal entry: {row0, row1, sign0 | level0 << 8, sign1 | level1 << 8}, row = -1: none
  int nrows;
  rtoc_record_layout nl;
  rtoc_record_layout kl, cl;
  // friction / wrench cone rows condensed by mjtjinv_kernel (split condensation): 0 = none (or done by their
  // own kernel), RTOC_FRICTION_ROWS, RTOC_WRENCH_ROWS
  int cone_rows;
  const double* cone;
  double* cone_con;  // constraint records (the box rows' `con` may be null when only cones are set)
  int cone_contacts, cone_dim, cone_row0, cone_stride, cone_dgdf_off, cone_impact;
  int keep_qaf;  // RTOC_OPT_CONDENSE_KEEP_QAF: also store Qafqv / Qafu_full in the ContactDynamicsData record
  const double* dt_inst;  // [batch][nstages] per-instance time steps (switching-time optimisation) or nullptr (device_utils.hpp: grid_dt)
  // work items = batch x these grid points (nullptr: all of 0 .. nstages - 2): the impact grid points behind condense_rv_kernel
  const int* stage_list;
  int nlist;
};

struct ExpArgs {
  double* cdd;
  double* dir;
  double* con;               // constraint records or nullptr
  const rtoc_box_row* rows;
  int nrows;
  rtoc_record_layout nl;
  unsigned long long* steps; // [batch][2] max primal / dual step (bit patterns of positive doubles)
  const rtoc_grid* grid;
  int nstages, batch;
  rtoc_record_layout cl, dl;
  double tau;
  const double* dt_inst;  // per-instance time steps or nullptr (grid_dt)
  long long* prof;        // optional cycle stamps (slots 32..39) of the work item in the middle of the launch (tuning aid)
};

template <int NV, int NU, int NF, int NS, bool SPLIT = false>
struct CondCfg {
  static constexpr int NW = RTOC_COND_NW;  // wavefronts per work item (tiles dealt round-robin)
  static constexpr int NT = 64 * NW;
  static constexpr int NX = 2 * NV;
  static constexpr int NFP = NF > 0 ? NF : 1;
  static constexpr int LDV = NV + NF;
  static constexpr int pad8(int n) { return (n + 7) & ~7; }
  // LDS carve (doubles): one work item = one (instance, grid point); everything the stage
  // touches more than once lives here, HBM sees each record field once in and once out.
  static constexpr int O_LAM = 0;                              // MJtJinv            LDV x LDV
  static constexpr int O_D = O_LAM + pad8(LDV * LDV);          // dIDCdqv, later Qafqv  LDV x NX
  static constexpr int O_LD = O_D + pad8(LDV * NX);            // MJtJinv_dIDCdqv    LDV x NX
  // region X: {M -> L, J, J Minv, S, -(S)^-1} while MJtJinv is built, then Qafu_full
  static constexpr int O_X = O_LD + pad8(LDV * NX);
  static constexpr int X_A = pad8(NV * NV) + 2 * pad8(NFP * NV) + 2 * pad8(NFP * NFP);
  // FUSE (the one-kernel condensation on contact shapes): the factorisation scratch is dead before MJtJinv_dIDCdqv is born
  // and lives in ITS region; wave 0 assembles MJtJinv ALONE (the serial factorisations dominate it) while wave 1 condenses the
  // cone rows (scratch: the dIDCdqv region, not staged yet) and wave 2 fetches dIDCdqv and IDC.  ANYmal: 37.2 -> 31.1 KB = 25
  // LDS granules, five work items per CU like the split kernel.
  static constexpr bool FUSE = !SPLIT && NF > 0 && X_A <= pad8(LDV * NX) && RTOC_COND_NW >= 3 && ConeScratch<NV, NF>::DOUBLES <= LDV * NX;
  static constexpr int O_L = FUSE ? O_LD : O_X;                // NV x NV
  static constexpr int O_J = O_L + pad8(NV * NV);              // NF x NV (ld NFP)
  static constexpr int O_JM = O_J + pad8(NFP * NV);            // NF x NV
  static constexpr int O_S = O_JM + pad8(NFP * NV);            // NF x NF
  static constexpr int O_BR = O_S + pad8(NFP * NFP);           // NF x NF
  static constexpr int Y_ROOM = O_BR + pad8(NFP * NFP) - O_JM;  // J M^-1, S, bottomRight: free while M is factorised
  static constexpr int X_B = pad8(LDV * NV);                   // Qafu_full LDV x NV
  // The split kernel gets MJtJinv from mjtjinv_kernel: no factorisation scratch (X_A).  And once MJtJinv_dIDCdqv and
  // MJtJinv_IDC exist, the columns NV.. of MJtJinv are only read through their exact mirror images in the rows NV..
  // (bottomLeft is stored as a copy of topRight^T, condense_mjtjinv.inc): with nu <= nf_max the actuated columns of
  // Qafu_full live there (TAIL) and region X holds its passive columns only.  ANYmal: 37.2 -> 29.9 KB = 24 of the 128
  // LDS granules (1280 B) of a CU, five work items per CU instead of four.
  static constexpr bool TAIL = (SPLIT || FUSE) && NF > 0 && NU <= NF;
  static constexpr int X_T = pad8(LDV * (NV - NU) > 8 ? LDV * (NV - NU) : 8);
  static constexpr int X_SZ = TAIL ? X_T : ((SPLIT || FUSE) ? X_B : (X_A > X_B ? X_A : X_B));
  static constexpr int O_QAFU = O_X;
  static constexpr int O_QFF = O_X + X_SZ;                     // NF x NF
  static constexpr int O_QQF = O_QFF + pad8(NFP * NFP);        // NV x NF
  static constexpr int O_VEC = O_QQF + pad8(NV * NFP);
  static constexpr int V_IDC = O_VEC, V_LR = V_IDC + pad8(LDV), V_LAF = V_LR + pad8(LDV),
                       V_HAF = V_LAF + pad8(LDV), V_QAA = V_HAF + pad8(LDV), V_LINV = V_QAA + pad8(NV),
                       V_SINV = V_LINV + pad8(NV);
  // PDIPM box rows: what they add to diag(Qqq), diag(Qvv), diag(Quu) and to lq, lv, lu, per primal entry
  static constexpr int V_PH = V_SINV + pad8(NFP), V_PG = V_PH + pad8(2 * NV + NU);
  static constexpr int LDS_DOUBLES = V_PG + pad8(2 * NV + NU);
  static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
  // work items per CU by LDS (160 KB in granules of 1280 B) -> waves per SIMD the register budget has to allow
  static constexpr int ITEMS = 128 / ((LDS_BYTES + 1279) / 1280);
  static constexpr int MIN_WAVES = TAIL && (ITEMS * NW + 3) / 4 >= 4 ? 4 : ((FUSE && (ITEMS * NW + 3) / 4 >= 2) ? ((ITEMS * NW + 3) / 4 >= 3 ? 3 : 2) : 1);
};

// One-wave dense product on the f64 matrix cores with generic (row, column) strides:
//   C(i,j) = beta * C(i,j) + alpha * sum_k A(i,k) B(k,j),  A(i,k) = A[i*ars + k*acs], ...
// Operands may live in HBM/L2 or LDS (flat addressing).  Runtime dimensions; tiles are
// processed two column tiles at a time to keep two MFMA chains in flight.
template <int NW, int MAXP = 0>
__device__ __forceinline__ void wave_gemm(int M, int N, int K, double alpha, const double* A, int ars,
                                          int acs, const double* B, int brs, int bcs, double beta,
                                          double* C, int crs, int ccs, int tid,
                                          // optional second product accumulated into the rows < M2 of
                                          // the same tiles (one read-modify-write of C instead of two):
                                          int M2 = 0, int K2 = 0, double alpha2 = 0.0,
                                          const double* A2 = nullptr, int a2rs = 0, int a2cs = 0,
                                          const double* B2 = nullptr, int b2rs = 0, int b2cs = 0,
                                          // optional: C values of this wave's tile pairs, fetched
                                          // earlier with prefetch_c (same M, N, strides)
                                          const double (*cpre)[8] = nullptr) {
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, q = lane >> 4;
  const int tmn = (M + 15) >> 4, tnp = (((N + 15) >> 4) + 1) >> 1, ksn = (K + 3) >> 2;
  // tile pairs (16 rows x 32 columns) are dealt round-robin to the NW waves of the work item
  // one tile pair; `cp` = its prefetched C values (or nullptr)
  auto do_pair = [&](const int tp, const double* cp) {
    const int tm = tp / tnp, tn = (tp - tm * tnp) * 2;
    const int i = tm * 16 + li;
    const bool iok = i < M;
    const double* ap = A + (size_t)(iok ? i : 0) * ars;
    const int j0 = tn * 16 + li, j1 = j0 + 16;
    const bool j0ok = j0 < N, j1ok = j1 < N;
    const double* bp0 = B + (size_t)(j0ok ? j0 : 0) * bcs;
    const double* bp1 = B + (size_t)(j1ok ? j1 : 0) * bcs;
    // read-modify-write targets are fetched BEFORE the k-loop: when C lives in HBM (in-place
    // update of the KKT record) the load latency overlaps the MFMA work instead of trailing it
    double c0[4], c1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = tm * 16 + drow(q, r);
      if (cp) {
        c0[r] = cp[r];
        c1[r] = cp[4 + r];
      } else {
        const bool ok0 = beta != 0.0 && row < M && j0ok, ok1 = beta != 0.0 && row < M && j1ok;
        const double v0 = C[ok0 ? (size_t)row * crs + (size_t)j0 * ccs : 0];
        const double v1 = C[ok1 ? (size_t)row * crs + (size_t)j1 * ccs : 0];
        c0[r] = ok0 ? v0 : 0.0;
        c1[r] = ok1 ? v1 : 0.0;
      }
    }
    d4 acc0 = zero4(), acc1 = zero4();
    // four k-steps per trip: 12 operand loads in flight before the 8 MFMAs that consume them
    for (int ks = 0; ks < ksn; ks += 4) {
      double av[4], b0[4], b1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = (ks + u) * 4 + q;
        const bool kok = k < K;
        const int kc = kok ? k : 0;
        const double a_ = ap[(size_t)kc * acs];
        const double x0 = bp0[(size_t)kc * brs];
        const double x1 = bp1[(size_t)kc * brs];
        av[u] = (iok && kok) ? a_ : 0.0;
        b0[u] = (j0ok && kok) ? x0 : 0.0;
        b1[u] = (j1ok && kok) ? x1 : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc0 = mfma16(av[u], b0[u], acc0);
        acc1 = mfma16(av[u], b1[u], acc1);
      }
    }
    d4 acd0 = zero4(), acd1 = zero4();
    if (M2 > 0 && tm * 16 < M2) {
      const bool i2ok = i < M2;
      const double* ap2 = A2 + (size_t)(i2ok ? i : 0) * a2rs;
      const double* bq0 = B2 + (size_t)(j0ok ? j0 : 0) * b2cs;
      const double* bq1 = B2 + (size_t)(j1ok ? j1 : 0) * b2cs;
      for (int ks = 0; ks < ((K2 + 3) >> 2); ks += 4) {
        double av[4], b0[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = (ks + u) * 4 + q;
          const bool kok = k < K2;
          const int kc = kok ? k : 0;
          const double a_ = ap2[(size_t)kc * a2cs];
          const double x0 = bq0[(size_t)kc * b2rs];
          const double x1 = bq1[(size_t)kc * b2rs];
          av[u] = (i2ok && kok) ? a_ : 0.0;
          b0[u] = (j0ok && kok) ? x0 : 0.0;
          b1[u] = (j1ok && kok) ? x1 : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acd0 = mfma16(av[u], b0[u], acd0);
          acd1 = mfma16(av[u], b1[u], acd1);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = tm * 16 + drow(q, r);
      if (row < M) {
        if (j0ok) C[(size_t)row * crs + (size_t)j0 * ccs] = beta * c0[r] + alpha * acc0[r] + alpha2 * acd0[r];
        if (j1ok) C[(size_t)row * crs + (size_t)j1 * ccs] = beta * c1[r] + alpha * acc1[r] + alpha2 * acd1[r];
      }
    }
  };
  if constexpr (MAXP > 0) {
    // compile-time pair slots: the prefetched C values are indexed statically and stay in registers
    // (a run-time index would put the array into scratch memory -- an HBM round trip per tile pair)
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int tp = wave + p * NW;
      if (tp < tmn * tnp) do_pair(tp, cpre ? cpre[p] : nullptr);
    }
  } else {
    for (int tp = wave; tp < tmn * tnp; tp += NW) do_pair(tp, nullptr);
  }
}

// The C values wave_gemm<NW> would read-modify-write for (M, N, C, crs, ccs), fetched ahead of time
// into registers: out[p] holds the p-th tile pair of this wave (MAXP must cover them, compile-time).
template <int NW, int MAXP>
__device__ __forceinline__ void prefetch_c(int M, int N, const double* C, int crs, int ccs, int tid,
                                           double (&out)[MAXP][8]) {
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, q = lane >> 4;
  const int tmn = (M + 15) >> 4, tnp = (((N + 15) >> 4) + 1) >> 1;
#pragma unroll
  for (int p = 0; p < MAXP; ++p) {
    const int tp = wave + p * NW;
    const bool tok = tp < tmn * tnp;
    const int tm = tp / tnp, tn = (tp - tm * tnp) * 2;
    const int j0 = tn * 16 + li, j1 = j0 + 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = tm * 16 + drow(q, r);
      // unconditional loads from clamped addresses + select: a predicated load would be compiled
      // into a branch with its own s_waitcnt, i.e. one HBM round trip per element
      const bool ok0 = tok && row < M && j0 < N, ok1 = tok && row < M && j1 < N;
      const double v0 = C[ok0 ? (size_t)row * crs + (size_t)j0 * ccs : 0];
      const double v1 = C[ok1 ? (size_t)row * crs + (size_t)j1 * ccs : 0];
      out[p][r] = ok0 ? v0 : 0.0;
      out[p][4 + r] = ok1 ? v1 : 0.0;
    }
  }
}

// The C values of the 16x16 tiles lds_gemm<NW, M, N, ...> deals to this wave (tile t -> wave t % NW,
// slot t / NW), fetched from HBM ahead of time; C(i,j) = C[i + j*LDC].
template <int NW, int M, int N, int LDC, int SLOTS>
__device__ __forceinline__ void prefetch_tiles(const double* C, int tid, double (&out)[SLOTS][4]) {
  constexpr int TM = (M + 15) / 16, TN = (N + 15) / 16;
  static_assert(SLOTS * NW >= TM * TN, "not enough slots");
  const int lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;
  // branch-free over the slots (tile index computed from the wave id) and no select on the loaded
  // values: out-of-range lanes read element 0 and are never consumed (the epilogue masks them), so
  // nothing here forces an s_waitcnt before the loads of the other fields are in flight
#pragma unroll
  for (int p = 0; p < SLOTS; ++p) {
    const int t = p * NW + wave;
    const int tm = t / TN, tn = t - tm * TN;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = tm * 16 + drow(q, r), col = tn * 16 + li;
      const bool ok = t < TM * TN && row < M && col < N;
      out[p][r] = C[ok ? row + col * LDC : 0];
    }
  }
}
template <int NW, int M, int N>
struct TileSlots {
  static constexpr int value = (((M + 15) / 16) * ((N + 15) / 16) + NW - 1) / NW;
};

// y(i) = beta*y(i) + alpha * sum_k A(i,k) x(k): one lane per row, operands anywhere.
template <int NT>
__device__ __forceinline__ void wave_gemv(int M, int K, double alpha, const double* A, int ars, int acs,
                                          const double* x, double beta, double* y, int lane) {
  for (int i = lane; i < M; i += NT) {
    double acc = 0.0;
#pragma unroll 8
    for (int k = 0; k < K; ++k) acc += A[(size_t)i * ars + (size_t)k * acs] * x[k];
    y[i] = (beta == 0.0 ? 0.0 : beta * y[i]) + alpha * acc;
  }
}

// SPLIT = false: everything in one kernel.  SPLIT = true: MJtJinv was computed by mjtjinv_kernel (below) and
// is read back from the ContactDynamicsData record -- the serial factorisations then run at four times the
// occupancy (16 KB instead of 38 KB of LDS per grid point) and off this kernel's critical path.
template <int NV, int NU, int NF, int NS, bool SPLIT = false>
__global__ __launch_bounds__(64 * RTOC_COND_NW, (CondCfg<NV, NU, NF, NS, SPLIT>::MIN_WAVES)) void condense_kernel(CondArgs a) {
  static_assert(CondCfg<NV, NU, NF, NS>::NT == 64 * RTOC_COND_NW, "launch bounds must match CondCfg::NT");
  using C = CondCfg<NV, NU, NF, NS, SPLIT>;
  constexpr bool TAIL = C::TAIL;
  constexpr int NT = C::NT, NW = C::NW;
  constexpr int NX = 2 * NV, NP = NV - NU, LDV = C::LDV, LDF = C::NFP, LDS_ = NS > 0 ? NS : 1;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* const Lam = smem + C::O_LAM;
  double* const D = smem + C::O_D;
  double* const Qafqv = smem + C::O_D;  // overwrites dIDCdqv once MJtJinv_dIDCdqv exists
  double* const LD = smem + C::O_LD;
  double* const sL = smem + C::O_L;
  double* const sJ = smem + C::O_J;
  double* const sJM = smem + C::O_JM;
  double* const sS = smem + C::O_S;
  double* const sBR = smem + C::O_BR;
  double* const Qafu = smem + C::O_QAFU;                               // passive columns first (all of it unless TAIL)
  double* const QafuU = TAIL ? Lam + NV * LDV : Qafu + (NV - NU) * LDV;  // actuated columns
  double* const Qff = smem + C::O_QFF;
  double* const Qqf = smem + C::O_QQF;
  double* const IDC = smem + C::V_IDC;
  double* const Lr = smem + C::V_LR;
  double* const laf = smem + C::V_LAF;
  double* const haf = smem + C::V_HAF;
  double* const Qaa = smem + C::V_QAA;
  double* const sLinv = smem + C::V_LINV;
  double* const sSinv = smem + C::V_SINV;
  const int lane = threadIdx.x;  // thread index within the work item (NT threads)
  const int wv = lane >> 6, wl = lane & 63;
  const int item = blockIdx.x;  // instance * (nstages-1) + stage
  const int nst1 = a.stage_list ? a.nlist : a.nstages - 1;
  const int b = item / nst1;
  const int st = a.stage_list ? a.stage_list[item % nst1] : item % nst1;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  const bool impact = g.type == RTOC_GRID_IMPACT;
  const int nf = g.dimf, nvf = NV + nf, ns = impact ? 0 : g.dims;
  const double dt = grid_dt(a.grid, a.dt_inst, b, a.nstages, st);
  // record offsets as immediates (StaticLayout: the host checks them against the run-time layout)
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout KL = SL.kkt, CL = SL.cdd;
  static_assert(NF == NS, "kernel sets are built with nf_max == ns_max");
  double* kr = a.kkt + ((size_t)b * a.nstages + st) * KL.stride;
  double* cr = a.cdd + ((size_t)b * a.nstages + st) * CL.stride;
  double* const Fxx = kr + KL.off[RTOC_KKT_FXX];
  double* const Fvu = kr + KL.off[RTOC_KKT_FVU];
  double* const Qxx = kr + KL.off[RTOC_KKT_QXX];
  double* const Qxu = kr + KL.off[RTOC_KKT_QXU];
  double* const Quu = kr + KL.off[RTOC_KKT_QUU];
  double* const Fx = kr + KL.off[RTOC_KKT_FX];
  double* const lx = kr + KL.off[RTOC_KKT_LX];
  double* const lu = kr + KL.off[RTOC_KKT_LU];
  double* const fx = kr + KL.off[RTOC_KKT_FFX];
  double* const hx = kr + KL.off[RTOC_KKT_HX];
  double* const hu = kr + KL.off[RTOC_KKT_HU];
  double* const scal = kr + KL.off[RTOC_KKT_SCAL];
  double* const Phix = kr + KL.off[RTOC_KKT_PHIX];
  double* const Phiu = kr + KL.off[RTOC_KKT_PHIU];
  double* const Phit = kr + KL.off[RTOC_KKT_PHIT];
  double* const Pres = kr + KL.off[RTOC_KKT_PRES];
  const double* const Phia = cr + CL.off[RTOC_CDD_PHIA];
  double* const lup = cr + CL.off[RTOC_CDD_LUP];
  double* const Qxup = cr + CL.off[RTOC_CDD_QXUP];
  double* const Quuptr = cr + CL.off[RTOC_CDD_QUUPTR];
  unsigned stat = 0;

  RTOC_CPROF(0);
  // ================= HBM -> registers: every input field once, all loads in flight together ======
  // 16 B per lane; odd-sized fields read/write one double of their 64-B padding
  constexpr bool FUSE = C::FUSE;
  constexpr int H_L = (NV * NV + 1) / 2, H_D = (LDV * NX + 1) / 2, H_J = (C::NFP * NV + 1) / 2,
                H_F = (C::NFP * C::NFP + 1) / 2;
  constexpr int N_L = (H_L + NT - 1) / NT, N_D = (H_D + NT - 1) / NT, N_J = (H_J + NT - 1) / NT,
                N_F = (H_F + NT - 1) / NT;
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  constexpr int H_LAM = (LDV * LDV + 1) / 2, N_LAM = SPLIT ? (H_LAM + NT - 1) / NT : 1;
  dbl2 gL[N_L], gD[N_D], gJ[N_J], gF[N_F], gQ[N_J], gLam[N_LAM];
  const dbl2 zero2 = {0.0, 0.0};
  // compile-time trip counts (arrays stay in registers) and unconditional loads from clamped
  // addresses: every field exists in the max-size record whatever dimf is, and a load under a branch
  // would get its own s_waitcnt -- one HBM round trip per field.  (Lanes past the end hold a copy of element 0 and
  // never store it: no select.)
#define RTOC_LD2(dst, CNT, src, n2)                                              \
  _Pragma("unroll") for (int k = 0; k < (CNT); ++k) {                            \
    const int e = lane + k * NT;                                                 \
    dst[k] = reinterpret_cast<const dbl2*>(src)[e < (n2) ? e : 0];               \
  }
#define RTOC_ST2(dst, src, CNT, n2)                                              \
  _Pragma("unroll") for (int k = 0; k < (CNT); ++k) {                            \
    const int e = lane + k * NT;                                                 \
    if (e < (n2)) reinterpret_cast<dbl2*>(dst)[e] = src[k];                      \
  }
  // FUSE: wave 0 alone fetches what it factorises -- M and both candidates of J (dCda on contact grids, dCdv inside
  // dIDCdqv on impact grids), from addresses that do not depend on the grid descriptor -- and starts on them while the
  // other waves are still waiting for their loads
  constexpr int N_L0 = (H_L + 63) / 64, N_J0 = (C::NFP * NV + 63) / 64;
  dbl2 fL[FUSE ? N_L0 : 1];
  double fJ[FUSE ? N_J0 : 1], fJi[FUSE ? N_J0 : 1];
  if constexpr (FUSE) {
    if (wv == 0) {
#pragma unroll
      for (int k = 0; k < N_L0; ++k) {
        const int e = wl + k * 64;
        fL[k] = reinterpret_cast<const dbl2*>(cr + CL.off[RTOC_CDD_DIDDA])[e < H_L ? e : 0];
      }
#pragma unroll
      for (int k = 0; k < N_J0; ++k) {
        const int e = wl + k * 64, r = e % LDF, c = e / LDF;
        const bool ok = c < NV;
        fJ[k] = cr[CL.off[RTOC_CDD_DCDA] + (ok ? r + c * LDF : 0)];
        fJi[k] = cr[CL.off[RTOC_CDD_DIDCDQV] + NV + NV * LDV + (ok ? r + c * LDV : 0)];
      }
    }
  } else if constexpr (!SPLIT) {
    RTOC_LD2(gL, N_L, cr + CL.off[RTOC_CDD_DIDDA], H_L)
    RTOC_LD2(gJ, N_J, cr + CL.off[RTOC_CDD_DCDA], H_J)
  } else {
    RTOC_LD2(gLam, N_LAM, cr + CL.off[RTOC_CDD_MJTJINV], H_LAM)
  }
  // Everything else the work item reads.  !FUSE: requested here, with the first round trip.  FUSE: behind the barrier
  // that ends the factorisation of M (RTOC_MJ_AFTER_B1 in the fragment below) -- wave 1 of this very work item condenses the
  // cone rows meanwhile, which add to Qqq (inside Qxx), Qqf, Qff, lq (inside lx) and lf, and the factorisation / the cone
  // Gram product want the registers these values would occupy; they arrive while MJtJinv is assembled.
  const int lv_ = lane < NV ? lane : 0, lf_ = lane < nf ? lane : 0, lvf_ = lane < nvf ? lane : 0;
  const int ix_ = lane < NX ? lane : 0, iv_ = lane < NV ? lane : 0;
  double vQaa, vLa, vHa, vHf, vIdc, vLf;
  // gradient / sensitivity entries that get one read-modify-write at the end
  double pLx, pHx, pFf, pFxv, pLup, pLu, pHu, pSh, pSq;
  // the Hessian blocks the Schur updates read-modify-write: fetched early, consumed ~40k cycles later -- their HBM latency
  // is off the chain
  double cQxx[TileSlots<NW, NX, NX>::value][4];
  double cQxu[TileSlots<NW, NX, NU>::value][4];
  double cQuu[TileSlots<NW, NU, NU>::value][4];
  // PDIPM box rows: the sums stay in LDS and enter the Hessian / gradient entries where those get their one
  // read-modify-write below.  Descriptor -> row data is a dependent pair of HBM round trips: the descriptor
  // load rides with the field loads, the row data is requested as soon as it is back (after the
  // registers -> LDS stage) and consumed ahead of the Schur updates, ~15k cycles of products later.
  double* const sPH = smem + C::V_PH;
  double* const sPG = smem + C::V_PG;
  constexpr int NE = 3 * NV + NU;  // primal entries with box rows: q, v, u, a
  static_assert(NE <= NT, "one lane per primal entry");
  const bool box_on = a.con != nullptr && !impact;
  const bool box_lane = box_on && lane < NE;
  int4 pd = make_int4(-1, -1, 0, 0);
  int ent0 = 0, ent1 = 0;  // rows of the entry beyond the first two: CSR range
  // the packed row descriptor of the lane's primal entry: the same for every work item (an L2 hit), requested first -- in the
  // fused kernel at the very top, so that the row data can be requested the moment the cone rows are done
#define RTOC_BOX_DESCRIPTOR                                  \
  if (box_on) {                                              \
    const int t = lane < NE ? lane : 0;                      \
    pd = a.pair[t];                                          \
    ent0 = a.entry[t] + 2;                                   \
    ent1 = a.entry[t + 1];                                   \
  }
  // the gradient / sensitivity entries that get one read-modify-write at the very end: with the first round trip -- or, in the
  // fused kernel (whose register budget they would overrun for 40k cycles: they were spilled, 8 KB of scratch traffic per work
  // item), ahead of the Schur updates, one phase before they are consumed
#define RTOC_GRADIENT_LOADS                                                                                    \
  pLx = lx[ix_], pHx = hx[ix_], pFf = fx[ix_], pFxv = Fx[NV + iv_];                                            \
  pLup = lup[iv_ < NP ? iv_ : 0], pLu = lu[iv_ >= NP ? iv_ - NP : 0], pHu = hu[iv_ >= NP ? iv_ - NP : 0];      \
  pSh = scal[RTOC_KKT_SCAL_H], pSq = scal[RTOC_KKT_SCAL_QTT];
#define RTOC_FIELD_LOADS                                                                                     \
  if constexpr (!FUSE) {                                                                                      \
    RTOC_BOX_DESCRIPTOR                                                                                       \
  }                                                                                                           \
  if constexpr (!FUSE) {                                                                                      \
    RTOC_LD2(gD, N_D, cr + CL.off[RTOC_CDD_DIDCDQV], H_D)                                                     \
    vIdc = cr[CL.off[RTOC_CDD_IDC] + lvf_];                                                                   \
  }                                                                                                           \
  RTOC_LD2(gF, N_F, cr + CL.off[RTOC_CDD_QFF], H_F)                                                           \
  RTOC_LD2(gQ, N_J, cr + CL.off[RTOC_CDD_QQF], H_J)                                                           \
  vQaa = cr[CL.off[RTOC_CDD_QAA] + lv_], vLa = cr[CL.off[RTOC_CDD_LA] + lv_], vHa = cr[CL.off[RTOC_CDD_HA] + lv_]; \
  vLf = cr[CL.off[RTOC_CDD_LF] + lf_], vHf = cr[CL.off[RTOC_CDD_HF] + lf_];                                    \
  if constexpr (!FUSE) {                                                                                       \
    RTOC_GRADIENT_LOADS                                                                                        \
  }                                                                                                            \
  prefetch_tiles<NW, NX, NX, NX>(Qxx, lane, cQxx);                                                             \
  prefetch_tiles<NW, NX, NU, NX>(Qxu, lane, cQxu);                                                             \
  prefetch_tiles<NW, NU, NU, NU>(Quu, lane, cQuu);
  if constexpr (!FUSE) {
    RTOC_FIELD_LOADS
  } else {
    RTOC_BOX_DESCRIPTOR
  }
  RTOC_CPROF(21);
  // one lane per primal entry (q_k, v_k, u_k): the first two rows of the entry (a lower and an upper limit:
  // all there is for joint limits) come from the packed descriptor; their data is fetched unconditionally
  double* const nr = box_on ? a.con + ((size_t)b * a.nstages + st) * a.nl.stride : nullptr;
  double bs0 = 1.0, bd0 = 0.0, bq0 = 0.0, bc0 = 0.0, bs1 = 1.0, bd1 = 0.0, bq1 = 0.0, bc1 = 0.0;
#define RTOC_BOX_LOADS                                                                                              \
  if (box_on) {                                                                                                     \
    const int* no = a.nl.off;                                                                                       \
    const int r0 = pd.x >= 0 ? pd.x : 0, r1 = pd.y >= 0 ? pd.y : 0;                                                \
    bs0 = nr[no[RTOC_CON_SLACK] + r0], bd0 = nr[no[RTOC_CON_DUAL] + r0], bq0 = nr[no[RTOC_CON_RESIDUAL] + r0],      \
    bc0 = nr[no[RTOC_CON_CMPL] + r0];                                                                               \
    bs1 = nr[no[RTOC_CON_SLACK] + r1], bd1 = nr[no[RTOC_CON_DUAL] + r1], bq1 = nr[no[RTOC_CON_RESIDUAL] + r1],      \
    bc1 = nr[no[RTOC_CON_CMPL] + r1];                                                                               \
  }
  // registers -> LDS; vectors zero beyond the active contact dimension (the products run over the full extents)
#define RTOC_FIELD_STORES                            \
  if constexpr (!FUSE) {                             \
    RTOC_ST2(D, gD, N_D, H_D)                        \
    if (lane < LDV) IDC[lane] = lane < nvf ? vIdc : 0.0; \
  }                                                  \
  RTOC_ST2(Qff, gF, N_F, H_F)                        \
  RTOC_ST2(Qqf, gQ, N_J, H_J)                        \
  if constexpr (!FUSE) {                             \
    RTOC_BOX_LOADS                                   \
  }                                                  \
  RTOC_CPROF(23);                                    \
  if (lane < NV) {                                   \
    Qaa[lane] = vQaa;                                \
    laf[lane] = vLa;                                 \
    haf[lane] = vHa;                                 \
  }                                                  \
  if (lane < NF) {                                   \
    laf[NV + lane] = lane < nf ? -vLf : 0.0;         \
    haf[NV + lane] = lane < nf ? -vHf : 0.0;         \
  }
  if constexpr (!FUSE) {
    // inactive rows / columns (dimf < max_dimf) of the stored blocks stay zero, like the reference's
    // max-size backing matrices
    if constexpr (!SPLIT)
      for (int e = lane; e < LDV * LDV; e += NT) Lam[e] = 0.0;
    for (int e = lane; e < LDV * NX; e += NT) LD[e] = 0.0;
    RTOC_CPROF(22);
    // ================= registers -> LDS =================
    if constexpr (!SPLIT) {
      RTOC_ST2(sL, gL, N_L, H_L)
      RTOC_ST2(sJ, gJ, N_J, H_J)
    } else {
      RTOC_ST2(Lam, gLam, N_LAM, H_LAM)
    }
    RTOC_FIELD_STORES
    __syncthreads();
  }
  // inactive rows / columns of the max-size blocks are unspecified in the record: zero them in
  // LDS so that every product below can use its compile-time extents
#define RTOC_ZERO_PADDING                                                                                           \
  if (nf < NF) {                                                                                                    \
    if constexpr (!FUSE)                                                                                            \
      for (int e = lane; e < (LDV - nvf) * NX; e += NT) D[nvf + e % (LDV - nvf) + (e / (LDV - nvf)) * LDV] = 0.0;  \
    if constexpr (!SPLIT && !FUSE)                                                                                  \
      if (!impact) /* (impact: J = dCdv lives inside D, its inactive rows were zeroed with D's) */                  \
        for (int e = lane; e < (NF - nf) * NV; e += NT) sJ[nf + e % (NF - nf) + (e / (NF - nf)) * LDF] = 0.0;       \
    for (int e = lane; e < LDF * LDF; e += NT)                                                                      \
      if (e % LDF >= nf || e / LDF >= nf) Qff[e] = 0.0;                                                             \
    for (int e = lane; e < NV * (NF - nf); e += NT) Qqf[nf * NV + e] = 0.0;                                         \
    __syncthreads();                                                                                                \
  }
  if constexpr (!FUSE) {
    RTOC_ZERO_PADDING
  }
  // J: dCda (ld NF) on contact grids, dCdv = D[nv:, nv:] (ld LDV) on impact grids
  const double* const Jd = impact ? D + NV + (size_t)NV * LDV : sJ;
  const double* const J = FUSE ? sJ : Jd;
  const int ldj = impact ? LDV : LDF;

  // FUSE: wave 2's early reads -- dIDCdqv (all of it, over 64 lanes) and IDC, which no cone row touches and the first products need
  constexpr int N_D2 = (H_D + 63) / 64;
  dbl2 gD2[FUSE ? N_D2 : 1];
  if constexpr (FUSE) {
    // ---- wave roles until MJtJinv exists: 0 assembles it alone, 1 condenses the cone rows, 2 fetches dIDCdqv / IDC ----
    if (wv == 0) {
      for (int e = wl; e < LDV * LDV; e += 64) Lam[e] = 0.0;   // inactive rows / columns stay zero
#pragma unroll
      for (int k = 0; k < N_L0; ++k) {
        const int e = wl + k * 64;
        if (e < H_L) reinterpret_cast<dbl2*>(sL)[e] = fL[k];
      }
#pragma unroll
      for (int k = 0; k < N_J0; ++k) {  // rows >= dimf are zero; J in sJ with ld NF on contact and impact grids alike
        const int e = wl + k * 64;
        if (e < C::NFP * NV) sJ[e] = (e % LDF < nf) ? (impact ? fJi[k] : fJ[k]) : 0.0;
      }
      wave_lds_sync_();
      {
        // computeMJtJinv by this wave alone: the fragment with one wave's worth of "work item" (names shadowed on purpose)
        const int lane = wl;
        constexpr int NT = 64, NW = 1;
#define RTOC_MJ_SYNC() wave_lds_sync_()
#define RTOC_J_IN_D false
#include "condense_mjtjinv.inc"
#undef RTOC_J_IN_D
#undef RTOC_MJ_SYNC
        // out to HBM now: its columns nv.. are recycled once MJtJinv_dIDCdqv exists (TAIL)
        copy_s2g_flat16<NT>(cr + CL.off[RTOC_CDD_MJTJINV], Lam, LDV * LDV, lane);
      }
    } else if (wv == 1) {
      // Constraints::condenseSlackAndDual of the cone rows (intermediate_stage.cpp:134-135) ahead of everything that reads
      // Qqq, Qqf, Qff, lq, lf; scratch: the dIDCdqv region, which is staged behind the barrier
      if (a.cone_rows != 0) {
        ConeArgs ca;
        ca.kkt = a.kkt;
        ca.cdd = a.cdd;
        ca.con = a.cone_con;
        ca.cone = a.cone;
        ca.dir = nullptr;
        ca.grid = a.grid;
        ca.steps = nullptr;
        ca.nstages = a.nstages;
        ca.batch = a.batch;
        ca.max_contacts = a.cone_contacts;
        ca.contact_dim = a.cone_dim;
        ca.row0 = a.cone_row0;
        ca.rows_per_contact = a.cone_rows;
        ca.cone_stride = a.cone_stride;
        ca.dgdf_off = a.cone_dgdf_off;
        ca.impact_cones = a.cone_impact;
        ca.tau = 0.0;
        ca.kl = a.kl;
        ca.cl = a.cl;
        ca.nl = a.nl;
        ca.dl = a.cl;  // unused by the condensation
        ca.prof = nullptr;
        if (a.cone_rows == RTOC_WRENCH_ROWS)
          wrench_condense_body<NV, NF>(ca, b, st, wl, D);
        else
          cone_condense_body<NV, NF>(ca, b, st, wl, D);
        cone_wave_sync();
      }
    } else if (wv == 2) {
#pragma unroll
      for (int k = 0; k < N_D2; ++k) {
        const int e = wl + k * 64;
        gD2[k] = reinterpret_cast<const dbl2*>(cr + CL.off[RTOC_CDD_DIDCDQV])[e < H_D ? e : 0];
      }
      vIdc = cr[CL.off[RTOC_CDD_IDC] + (wl < nvf ? wl : 0)];
    }
    __syncthreads();   // MJtJinv assembled (and in flight to HBM), cone rows condensed (their writes visible), scratch free
    RTOC_CPROF(3);
    if (wv == 2) {   // dIDCdqv and IDC into LDS, zero beyond the active contact dimension
#pragma unroll
      for (int k = 0; k < N_D2; ++k) {
        const int e = wl + k * 64;
        if (e < H_D) {
          dbl2 v = gD2[k];
          if (nf < NF) {
            if ((2 * e) % LDV >= nvf) v.x = 0.0;
            if ((2 * e + 1) % LDV >= nvf) v.y = 0.0;
          }
          reinterpret_cast<dbl2*>(D)[e] = v;
        }
      }
      if (wl < LDV) IDC[wl] = wl < nvf ? vIdc : 0.0;
    }
    RTOC_BOX_LOADS     // the joint-limit rows' data (descriptor requested at the top) ...
    RTOC_FIELD_LOADS   // ... and everything else: it arrives during MJtJinv_dIDCdqv
    __syncthreads();
  } else if constexpr (!SPLIT) {
#define RTOC_J_IN_D impact
#include "condense_mjtjinv.inc"
#undef RTOC_J_IN_D
  }

  RTOC_CPROF(4);
  // ================= MJtJinv_dIDCdqv, MJtJinv_IDC (contact_dynamics.cpp:64-65 / impact :44-50) ===
  if (!impact) {
    lds_gemm<NW, LDV, NX, LDV, 1, LDV, 1, LDV>(Lam, D, lane, [&](int r, int c, double v, int, int) { LD[r + c * LDV] = v; });
  } else {
    lds_gemm<NW, LDV, NV, LDV, 1, LDV, 1, LDV>(Lam, D, lane, [&](int r, int c, double v, int, int) { LD[r + c * LDV] = v; });
    lds_gemm<NW, LDV, NV, C::NFP, 1, LDV, 1, LDV>(Lam + NV * LDV, Jd, lane,
                                                  [&](int r, int c, double v, int, int) { LD[r + (NV + c) * LDV] = v; });
  }
  wave_gemv<NT>(LDV, LDV, 1.0, Lam, 1, LDV, IDC, 0.0, Lr, lane);  // full extents: zero rows give Lr = 0 there
  if constexpr (FUSE) {
    // the fields requested behind the assembly of MJtJinv have arrived during the product above: registers -> LDS
    RTOC_FIELD_STORES
    __syncthreads();
    RTOC_ZERO_PADDING
  }
#undef RTOC_LD2
#undef RTOC_ST2
#undef RTOC_FIELD_LOADS
#undef RTOC_BOX_DESCRIPTOR
#undef RTOC_FIELD_STORES
#undef RTOC_ZERO_PADDING
#undef RTOC_BOX_LOADS
  // ================= PDIPM slack/dual elimination of the joint-limit rows =================
  // Constraints::condenseSlackAndDual (constraints.cpp:322-357, joint_*_limit.cpp, pdipm.hxx:66-69):
  // purely additive on diag(Qqq), diag(Qvv), diag(Quu), lq, lv, lu, so it commutes with the
  // contact-dynamics condensation -- except the acceleration limits, whose Qaa / la the condensation consumes:
  // hence here, between MJtJinv_dIDCdqv and Qafqv; every entry is accumulated by a single lane in row order
  // (deterministic, no atomics).
  if (lane < NE) {
    double hess = 0.0, grad = 0.0;
    if (box_lane) {
      const int* no = a.nl.off;
      if (pd.x >= 0 && g.time_stage >= (pd.z >> 8)) {
        const double cond = (bd0 * bq0 - bc0) / bs0;
        nr[no[RTOC_CON_COND] + pd.x] = cond;
        hess += bd0 / bs0;
        grad += (double)(signed char)(pd.z & 0xff) * cond;
      }
      if (pd.y >= 0 && g.time_stage >= (pd.w >> 8)) {
        const double cond = (bd1 * bq1 - bc1) / bs1;
        nr[no[RTOC_CON_COND] + pd.y] = cond;
        hess += bd1 / bs1;
        grad += (double)(signed char)(pd.w & 0xff) * cond;
      }
      const int* rowid = a.entry + (NE + 1);
      for (int e = ent0; e < ent1; ++e) {  // further rows on the same entry (none for joint limits)
        const int r = rowid[e];
        const rtoc_box_row row = a.rows[r];
        if (g.time_stage >= row.level) {
          const double slack = nr[no[RTOC_CON_SLACK] + r], dual = nr[no[RTOC_CON_DUAL] + r];
          const double cond = (dual * nr[no[RTOC_CON_RESIDUAL] + r] - nr[no[RTOC_CON_CMPL] + r]) / slack;
          nr[no[RTOC_CON_COND] + r] = cond;
          hess += dual / slack;
          grad += row.sign * cond;
        }
      }
    }
    if (lane < 2 * NV + NU) {
      sPH[lane] = hess;
      sPG[lane] = grad;
    } else if (box_lane && pd.x >= 0) {
      // JointAcceleration{Lower,Upper}Limit (joint_acceleration_lower_limit.cpp:69-77): Qaa.diagonal() and la, ahead of the
      // contact-dynamics condensation that reads them (contact_dynamics.cpp:68-86); the updated diagonal stays in the record
      // (the expansion rebuilds Qafqv dx + Qafu du from it), like the reference leaves it in kkt_matrix.Qaa
      const int i = lane - (2 * NV + NU);
      const double qn = Qaa[i] + hess;
      Qaa[i] = qn;
      laf[i] += grad;
      cr[CL.off[RTOC_CDD_QAA] + i] = qn;
    }
  }
  __syncthreads();  // D (and J inside it) is dead from here on: its space becomes Qafqv; region X becomes Qafu
  // (no zero fill + barrier ahead of the writes below: together they cover every entry of Qafqv / Qafu_full -- the rows of
  // the inactive contact dimensions come out as exact zeros from the zero-padded Qff / Qqf, and a grid point without contacts
  // zeroes its force rows itself)

  RTOC_CPROF(5);
  // ================= Qafqv, Qafu_full, laf (:67-88) =================
  for (int e = lane; e < NV * NX; e += NT) {
    const int i = e % NV, j = e / NV;
    Qafqv[i + j * LDV] = -Qaa[i] * LD[i + j * LDV];
  }
  if (!impact)
    for (int e = lane; e < NV * NV; e += NT) {
      const int i = e % NV, j = e / NV;
      (j < NP ? Qafu + j * LDV : QafuU + (j - NP) * LDV)[i] = Qaa[i] * Lam[i + j * LDV];
    }
  if (lane < NV) laf[lane] -= Qaa[lane] * Lr[lane];
  if (nf > 0) {
    // Qafqv[f rows] = -Qff (MJtJinv_dIDCdqv)[f rows] - [Qqf^T | 0]  (:76-80)
    lds_gemm<NW, C::NFP, NX, C::NFP, 1, LDF, 1, LDV>(Qff, LD + NV, lane, [&](int r, int c, double v, int, int) {
      Qafqv[(NV + r) + c * LDV] = -v - (c < NV ? Qqf[c + r * NV] : 0.0);
    });
    if (!impact)
      lds_gemm<NW, C::NFP, NV, C::NFP, 1, LDF, 1, LDV>(Qff, Lam + NV, lane,
                                                       [&](int r, int c, double v, int, int) {
                                                         (c < NP ? Qafu + c * LDV : QafuU + (c - NP) * LDV)[NV + r] = v;
                                                       });
    if (lane < nf) {
      double acc = 0.0;
      for (int k = 0; k < nf; ++k) acc += Qff[lane + k * LDF] * Lr[NV + k];
      laf[NV + lane] -= acc;
    }
  } else if constexpr (NF > 0) {
    for (int e = lane; e < NF * NX; e += NT) Qafqv[NV + e % NF + (e / NF) * LDV] = 0.0;
    if (!impact)
      for (int e = lane; e < NF * NV; e += NT) {
        const int c = e / NF;
        (c < NP ? Qafu + c * LDV : QafuU + (c - NP) * LDV)[NV + e % NF] = 0.0;
      }
  }
  __syncthreads();

  RTOC_CPROF(6);
  if constexpr (FUSE) {
    RTOC_GRADIENT_LOADS
  }
#undef RTOC_GRADIENT_LOADS
  // ================= Schur updates of the Hessian blocks and gradients (:90-130), in place in HBM ==
  // Each block gets ONE read-modify-write: the Qqf corrections (:92-93,:99-100,:106-107), which touch
  // the rows < NV only, ride along as a second product in the same tiles.
  lds_gemm2<NW, NX, NX, LDV, LDV, 1, 1, LDV, NV, C::NFP, 1, NV, 1, LDV>(
      LD, Qafqv, Qqf, LD + NV, nf > 0, lane, [&](int r, int c, double v1, double v2, int slot, int reg) {
        Qxx[r + (size_t)c * NX] = (cQxx[slot][reg] + (r == c ? sPH[r] : 0.0)) - v1 + v2;
      });
  RTOC_CPROF(10);
  if (!impact) {
    if constexpr (NP > 0) {
      lds_gemm2<NW, NX, NP, LDV, LDV, 1, 1, LDV, NV, C::NFP, 1, NV, 1, LDV>(
          LD, Qafu, Qqf, Lam + NV, nf > 0, lane,
          [&](int r, int c, double v1, double v2, int, int) { Qxup[r + (size_t)c * NX] = -v1 - v2; });
      if constexpr (!TAIL) {
        lds_gemm<NW, NP, NU, LDV, 1, LDV, 1, LDV>(Lam, QafuU, lane, [&](int r, int c, double v, int, int) {
          Quuptr[r + (size_t)c * NP] = v;
        });
      } else {  // MJtJinv[p, nv:] through its mirror image MJtJinv[nv:, p]
        lds_gemm2<NW, NP, NU, NV, 1, LDV, 1, LDV, NP, C::NFP, LDV, 1, 1, LDV>(
            Lam, QafuU, Lam + NV, QafuU + NV, nf > 0, lane,
            [&](int r, int c, double v1, double v2, int, int) { Quuptr[r + (size_t)c * NP] = v1 + v2; });
      }
    }
    RTOC_CPROF(11);
    lds_gemm2<NW, NX, NU, LDV, LDV, 1, 1, LDV, NV, C::NFP, 1, NV, 1, LDV>(
        LD, QafuU, Qqf, Lam + NV + NP * LDV, nf > 0, lane,
        [&](int r, int c, double v1, double v2, int slot, int reg) {
          Qxu[r + (size_t)c * NX] = cQxu[slot][reg] - v1 - v2;
        });
    if constexpr (!TAIL) {
      lds_gemm<NW, NU, NU, LDV, 1, LDV, 1, LDV>(Lam + NP, QafuU, lane, [&](int r, int c, double v, int slot, int reg) {
        Quu[r + (size_t)c * NU] = (cQuu[slot][reg] + (r == c ? sPH[2 * NV + r] : 0.0)) + v;
      });
    } else {
      lds_gemm2<NW, NU, NU, NV, 1, LDV, 1, LDV, NU, C::NFP, LDV, 1, 1, LDV>(
          Lam + NP, QafuU, Lam + NV + NP * LDV, QafuU + NV, nf > 0, lane,
          [&](int r, int c, double v1, double v2, int slot, int reg) {
            Quu[r + (size_t)c * NU] = (cQuu[slot][reg] + (r == c ? sPH[2 * NV + r] : 0.0)) + (v1 + v2);
          });
    }
  }
  RTOC_CPROF(12);
  // gradients: one lane per entry, every term of an entry in the same lane (:110-113,:123-130,:156-163);
  // sums run over the full zero-padded extents with four accumulators, and the evalKKT tail scalings
  // (intermediate_stage.cpp:140-148) are applied here so that every entry is written exactly once
  const double inv = impact ? 1.0 : 1.0 / (double)g.num_grids_in_phase;
  if (lane < NX) {
    const int i = lane;
    double al[4] = {0.0, 0.0, 0.0, 0.0}, ah[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < LDV; ++k) {
      const double ld = LD[k + i * LDV];
      al[k & 3] += ld * laf[k];
      ah[k & 3] += ld * haf[k];
    }
    double l = (pLx + sPG[i]) - ((al[0] + al[1]) + (al[2] + al[3])), h = pHx - ((ah[0] + ah[1]) + (ah[2] + ah[3]));
    if (i < NV && nf > 0) {
      double aq0 = 0.0, aq1 = 0.0;
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        if (k & 1)
          aq1 += Qqf[i + k * NV] * Lr[NV + k];
        else
          aq0 += Qqf[i + k * NV] * Lr[NV + k];
      }
      l += aq0 + aq1;
      h += (aq0 + aq1) / dt;
    }
    lx[i] = l;
    if (!impact) {
      hx[i] = h * inv;
      fx[i] = pFf * inv;
    }
  }
  RTOC_CPROF(13);
  if (!impact && lane < NV) {
    const int i = lane;
    double al[4] = {0.0, 0.0, 0.0, 0.0}, ah[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < LDV; ++k) {
      const double lm = (TAIL && k >= NV) ? Lam[k + i * LDV] : Lam[i + k * LDV];  // (TAIL: the exact mirror image)
      al[k & 3] += lm * laf[k];
      ah[k & 3] += lm * haf[k];
    }
    const double sl = (al[0] + al[1]) + (al[2] + al[3]), sh = (ah[0] + ah[1]) + (ah[2] + ah[3]);
    if (i < NP) {
      lup[i] = pLup + sl;
    } else {
      lu[i - NP] = (pLu + sPG[2 * NV + i - NP]) + sl;
      hu[i - NP] = (pHu + sh) * inv;
    }
  }
  if (!impact && lane == NT - 1) {
    // h -= MJtJinv_IDC . haf, then the 1/num_grids_in_phase scalings of h and Qtt (:144-147)
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < LDV; ++k) acc[k & 3] += Lr[k] * haf[k];
    scal[RTOC_KKT_SCAL_H] = (pSh - ((acc[0] + acc[1]) + (acc[2] + acc[3]))) * inv;
    const double qtt = pSq * inv * inv;
    scal[RTOC_KKT_SCAL_QTT] = qtt;
    scal[RTOC_KKT_SCAL_QTT_PREV] = -qtt;
  }

  RTOC_CPROF(7);
  // ================= condensed dynamics (:132-136 / impact :74-77) =================
  const double sdt = impact ? 1.0 : dt;
  for (int e = lane; e < NV * NV; e += NT) {
    const int i = e % NV, j = e / NV;
    Fxx[(NV + i) + (size_t)j * NX] = -sdt * LD[i + j * LDV];
    Fxx[(NV + i) + (size_t)(NV + j) * NX] = -sdt * LD[i + (NV + j) * LDV] + (i == j ? 1.0 : 0.0);
  }
  if (!impact)
    for (int e = lane; e < NV * NU; e += NT) {
      const int i = e % NV, j = e / NV;
      Fvu[i + (size_t)j * NV] = dt * Lam[i + (NP + j) * LDV];
    }
  if (lane < NV) Fx[NV + lane] = pFxv - sdt * Lr[lane];

  if (!impact) {
    // ================= switching constraint (:138-153) =================
    if (NS > 0 && ns > 0) {
      wave_gemm<NW>(ns, NX, NV, -1.0, Phia, 1, LDS_, LD, 1, LDV, 1.0, Phix, 1, LDS_, lane);
      wave_gemm<NW>(ns, NU, NV, 1.0, Phia, 1, LDS_, Lam + NP * LDV, 1, LDV, 0.0, Phiu, 1, LDS_, lane);
      if (lane < ns) {
        double acc = 0.0;
        for (int k = 0; k < NV; ++k) acc += Phia[lane + k * LDS_] * Lr[k];
        Phit[lane] = (Phit[lane] - acc) * (1.0 / (double)g.num_grids_in_phase);
        Pres[lane] -= acc;
      }
    }
  }

  RTOC_CPROF(8);
  // ================= LDS -> HBM: the ContactDynamicsData the expansion needs, each field once ====
  if constexpr (!SPLIT && !C::FUSE) copy_s2g_flat16<NT>(cr + CL.off[RTOC_CDD_MJTJINV], Lam, LDV * LDV, lane);
  copy_s2g_flat16<NT>(cr + CL.off[RTOC_CDD_MJD], LD, LDV * NX, lane);
  if (a.keep_qaf) {  // scratch of the reference's expandContactDynamicsDual; expand_kernel rebuilds what it needs of them
    copy_s2g_flat16<NT>(cr + CL.off[RTOC_CDD_QAFQV], Qafqv, LDV * NX, lane);
    if (!impact) {
      if constexpr (!TAIL) {
        copy_s2g_flat16<NT>(cr + CL.off[RTOC_CDD_QAFU], Qafu, LDV * NV, lane);
      } else {
        for (int e = lane; e < LDV * NV; e += NT) cr[CL.off[RTOC_CDD_QAFU] + e] = e < LDV * NP ? Qafu[e] : QafuU[e - LDV * NP];
      }
    }
  }
  if (lane < nvf) {
    cr[CL.off[RTOC_CDD_MJIDC] + lane] = Lr[lane];
    cr[CL.off[RTOC_CDD_LAF] + lane] = laf[lane];
    if (!impact) cr[CL.off[RTOC_CDD_HAF] + lane] = haf[lane];
  }
  RTOC_CPROF(9);
  if (stat) atomicOr(&a.status[b], stat);
}

// ---------------------------------------------------------------------------------------------
// First kernel of the split condensation: MJtJinv of every (instance, grid point) on ONE wave, into the
// ContactDynamicsData record.  4.3 KB in (dIDda, dCda | dCdv), 7.2 KB out at ANYmal size; the two serial
// Cholesky factorisations that dominate it only need the saddle-matrix scratch in LDS.
// ---------------------------------------------------------------------------------------------
template <int NV, int NF>
struct MjCfg {
  static constexpr int NFP = NF > 0 ? NF : 1, LDV = NV + NF, NX = 2 * NV;
  static constexpr int pad8(int n) { return (n + 7) & ~7; }
  static constexpr int O_LAM = 0;
  static constexpr int O_L = O_LAM + pad8(LDV * LDV);
  static constexpr int O_J = O_L + pad8(NV * NV);
  static constexpr int O_JM = O_J + pad8(NFP * NV);
  static constexpr int O_S = O_JM + pad8(NFP * NV);
  // bottomRight is born when the factor of M is dead and the inverse factor of S (NFP x NFP) sits at the head of that
  // region: it takes the doubles behind it when they are there (ANYmal: 15.6 -> 14.4 KB = 12 LDS granules, ten work
  // items per CU)
  static constexpr bool BR_IN_L = NF > 0 && NFP <= 16 && 2 * pad8(NFP * NFP) <= pad8(NV * NV);
  static constexpr int O_BR = BR_IN_L ? O_L + pad8(NFP * NFP) : O_S + pad8(NFP * NFP);
  static constexpr int Y_ROOM = (BR_IN_L ? O_S : O_BR) + pad8(NFP * NFP) - O_JM;
  static constexpr int O_LD = O_S + pad8(NFP * NFP) + (BR_IN_L ? 0 : pad8(NFP * NFP));  // scratch the fragment borrows on fixed-base sets
  static constexpr int V_LINV = O_LD + (NF == 0 ? pad8(LDV * NX > NV * NV ? LDV * NX : NV * NV) : 0);
  static constexpr int V_SINV = V_LINV + pad8(NV);
  static constexpr int LDS_DOUBLES = V_SINV + pad8(NFP);
  static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
  // one wave per work item: work items per CU by LDS -> waves per SIMD the register budget should allow (at most 3:
  // the factorisations want their ~170 registers)
  static constexpr int ITEMS = 128 / ((LDS_BYTES + 1279) / 1280);
  static constexpr int MIN_WAVES = (ITEMS + 3) / 4 >= 3 ? 3 : ((ITEMS + 3) / 4 >= 2 ? 2 : 1);
};

template <int NV, int NU, int NF, int NS>
__global__ __launch_bounds__(64, (MjCfg<NV, NF>::MIN_WAVES)) void mjtjinv_kernel(CondArgs a) {
  using C = MjCfg<NV, NF>;
  constexpr int NT = 64, NW = 1, NX = 2 * NV, LDV = C::LDV, LDF = C::NFP;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* const Lam = smem + C::O_LAM;
  double* const sL = smem + C::O_L;
  double* const sJ = smem + C::O_J;
  double* const sJM = smem + C::O_JM;
  double* const sS = smem + C::O_S;
  double* const sBR = smem + C::O_BR;
  double* const LD = smem + C::O_LD;
  double* const sLinv = smem + C::V_LINV;
  double* const sSinv = smem + C::V_SINV;
  const int lane = threadIdx.x, wv = 0, wl = lane;
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  const bool impact = g.type == RTOC_GRID_IMPACT;
  const int nf = g.dimf;
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout CL = SL.cdd;
  double* cr = a.cdd + ((size_t)b * a.nstages + st) * CL.stride;
  unsigned stat = 0;
  RTOC_CPROF(14);
  // the factorisation inputs are requested first: they arrive with the cone data, one HBM round trip for both
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  constexpr int H_L = (NV * NV + 1) / 2, N_L = (H_L + NT - 1) / NT, N_J = (C::NFP * NV + NT - 1) / NT;
  dbl2 gL[N_L];
  double gJ[N_J];
  const dbl2 zero2 = {0.0, 0.0};
#pragma unroll
  for (int k = 0; k < N_L; ++k) {
    const int e = lane + k * NT;
    const dbl2 v = reinterpret_cast<const dbl2*>(cr + CL.off[RTOC_CDD_DIDDA])[e < H_L ? e : 0];
    gL[k] = (e < H_L) ? v : zero2;
  }
  // J = dCda (NF x NV, ld NF) on contact grids, dCdv = dIDCdqv[nv:, nv:] (ld LDV) on impact grids
  // (impact_dynamics.cpp:44-50): both land in sJ with ld NF; rows >= dimf are zero
  // Both candidates are requested, from addresses that do not depend on the grid descriptor: nothing here
  // waits for the descriptor's own fetch.
  double gJi[N_J];
#pragma unroll
  for (int k = 0; k < N_J; ++k) {
    const int e = lane + k * NT, r = e % LDF, c = e / LDF;
    const bool ok = NF > 0 && c < NV;
    gJ[k] = cr[CL.off[RTOC_CDD_DCDA] + (ok ? r + c * LDF : 0)];
    gJi[k] = cr[CL.off[RTOC_CDD_DIDCDQV] + NV + NV * LDV + (ok ? r + c * LDV : 0)];
  }
  // Constraints::condenseSlackAndDual of the cone rows (intermediate_stage.cpp:134-135): they only touch
  // Qqq, Qqf, Qff, lq, lf, which this kernel does not read -- they ride here, in the shadow of the loads
  // below, and are in HBM before the second kernel starts.  Scratch: the not yet initialised Lam.
  if constexpr (NF > 0) if (a.cone_rows != 0) {
    static_assert(ConeScratch<NV, NF>::DOUBLES <= C::LDS_DOUBLES, "cone scratch is aliased onto the (not yet initialised) LDS carve");
    ConeArgs ca;
    ca.kkt = a.kkt;
    ca.cdd = a.cdd;
    ca.con = a.cone_con;
    ca.cone = a.cone;
    ca.dir = nullptr;
    ca.grid = a.grid;
    ca.steps = nullptr;
    ca.nstages = a.nstages;
    ca.batch = a.batch;
    ca.max_contacts = a.cone_contacts;
    ca.contact_dim = a.cone_dim;
    ca.row0 = a.cone_row0;
    ca.rows_per_contact = a.cone_rows;
    ca.cone_stride = a.cone_stride;
    ca.dgdf_off = a.cone_dgdf_off;
    ca.impact_cones = a.cone_impact;
    ca.tau = 0.0;
    ca.kl = a.kl;
    ca.cl = a.cl;
    ca.nl = a.nl;
    ca.dl = a.cl;  // unused by the condensation
    ca.prof = a.prof;
    if (a.cone_rows == RTOC_WRENCH_ROWS)
      wrench_condense_body<NV, NF>(ca, b, st, lane, smem);
    else
      cone_condense_body<NV, NF>(ca, b, st, lane, smem);
    cone_wave_sync();  // scratch is reused below
  }
  RTOC_CPROF(15);
  for (int e = lane; e < LDV * LDV; e += NT) Lam[e] = 0.0;
#pragma unroll
  for (int k = 0; k < N_L; ++k) {
    const int e = lane + k * NT;
    if (e < H_L) reinterpret_cast<dbl2*>(sL)[e] = gL[k];
  }
#pragma unroll
  for (int k = 0; k < N_J; ++k) {
    const int e = lane + k * NT;
    if (e < C::NFP * NV) sJ[e] = (NF > 0 && e % LDF < nf) ? (impact ? gJi[k] : gJ[k]) : 0.0;
  }
  const double* const J = sJ;
  __syncthreads();
#define RTOC_J_IN_D false
#include "condense_mjtjinv.inc"
#undef RTOC_J_IN_D
  copy_s2g_flat16<NT>(cr + CL.off[RTOC_CDD_MJTJINV], Lam, LDV * LDV, lane);
  RTOC_CPROF(24);
  if (stat) atomicOr(&a.status[b], stat);
}

// LDS carve of expand_kernel: the blocks of the ContactDynamicsData record its mat-vecs walk, staged once with coalesced
// 16-B loads that are all in flight together with every other read of the work item (the vectors, the joint-limit rows' data):
// ONE round trip to HBM per work item.  (The walks used to fetch the blocks column by column, 288 B per load instruction in
// dependent batches, MJtJinv twice, and the row data in two further dependent round trips.)  Cycle stamps of one work item
// (tools/expand_profile.py): 15k of its 28k cycles are that one round trip under load -- the kernel is bound by the bytes it can
// keep in flight (six work items per CU by LDS); a persistent, software-pipelined form (next item's reads in flight during this
// item's walks) was tried and is slower: its staging registers allow four work items per CU only.
template <int NV, int NU, int NF>
struct ExpCfg {
  static constexpr int NX = 2 * NV, NP = NV - NU, LDV = NV + NF, NFP = NF > 0 ? NF : 1, NPP = NP > 0 ? NP : 1;
  static constexpr int pad8(int n) { return (n + 7) & ~7; }
  // quadruped-size shapes (nvf <= 32: the two half-waves split the columns, 18 loads per lane) read MJtJinv_dIDCdqv straight from HBM
  // in the one walk that uses it -- 8.6 of 21.8 KB of LDS per work item less: eleven work items per CU instead of seven
  static constexpr bool STREAM_LD = LDV <= 32;
  static constexpr int O_LD = 0, O_LAM = O_LD + (STREAM_LD ? 0 : pad8(LDV * NX)), O_QFF = O_LAM + pad8(LDV * LDV), O_QQF = O_QFF + pad8(NFP * NFP),
                       O_QXUP = O_QQF + pad8(NV * NFP), O_QUUP = O_QXUP + pad8(NX * NPP), LDS_DOUBLES = O_QUUP + pad8(NPP * NU);
  static constexpr int LDS_BYTES = LDS_DOUBLES * 8;
};

template <int NV, int NU, int NF, int NS>
__global__ __launch_bounds__(64) void expand_kernel(ExpArgs a) {
  using E = ExpCfg<NV, NU, NF>;
  constexpr int NX = 2 * NV, NP = NV - NU, LDV = NV + NF, LDS_ = NS > 0 ? NS : 1, NT = 64;
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  constexpr rtoc_layout SL = StaticLayout<NV, NU, NS>::make();
  constexpr rtoc_record_layout CL = SL.cdd, DL = SL.dir;
  double* cr = a.cdd + ((size_t)b * a.nstages + st) * CL.stride;
  double* dr = a.dir + ((size_t)b * a.nstages + st) * DL.stride;
  const double* dn = dr + DL.stride;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* const LD = smem + E::O_LD;      // MJtJinv_dIDCdqv
  double* const Lam = smem + E::O_LAM;    // MJtJinv
  double* const Qff = smem + E::O_QFF;
  double* const Qqf = smem + E::O_QQF;
  double* const Qxup = smem + E::O_QXUP;
  double* const Quuptr = smem + E::O_QUUP;
  RTOC_CPROF(32);
  // ---- every read of the work item, requested up front (max-size records: no address depends on the grid descriptor) ----
  typedef double dbl2 __attribute__((ext_vector_type(2)));
  constexpr int LDF = NF > 0 ? NF : 1;
  constexpr int H_LD = (LDV * NX + 1) / 2, H_LAM = (LDV * LDV + 1) / 2, H_FF = (LDF * LDF + 1) / 2, H_QF = (NV * LDF + 1) / 2,
                H_XP = (NX * E::NPP + 1) / 2, H_UP = (E::NPP * NU + 1) / 2;
  constexpr int N_LD = (H_LD + NT - 1) / NT, N_LAM = (H_LAM + NT - 1) / NT, N_FF = (H_FF + NT - 1) / NT, N_QF = (H_QF + NT - 1) / NT,
                N_XP = (H_XP + NT - 1) / NT, N_UP = (H_UP + NT - 1) / NT;
  dbl2 gLD[E::STREAM_LD ? 1 : N_LD], gLam[N_LAM], gFF[N_FF], gQF[N_QF], gXP[N_XP], gUP[N_UP];
  // STREAM_LD: row i0 of MJtJinv_dIDCdqv, the columns of this half-wave (rows beyond the active contact dimension are stored as zeros)
  constexpr int HW = (LDV <= 32) ? 2 : 1, NLD = E::STREAM_LD ? (NX + HW - 1) / HW : 1;
  double gl[NLD];
#define RTOC_XLD(dst, CNT, src, n2)                                                  \
  _Pragma("unroll") for (int k = 0; k < (CNT); ++k) {                                \
    const int e = lane + k * NT;                                                     \
    dst[k] = reinterpret_cast<const dbl2*>(src)[e < (n2) ? e : 0];                   \
  }
#define RTOC_XST(dst, src, CNT, n2)                                                  \
  _Pragma("unroll") for (int k = 0; k < (CNT); ++k) {                                \
    const int e = lane + k * NT;                                                     \
    if (e < (n2)) reinterpret_cast<dbl2*>(dst)[e] = src[k];                          \
  }
  if constexpr (!E::STREAM_LD) {
    RTOC_XLD(gLD, N_LD, cr + CL.off[RTOC_CDD_MJD], H_LD)
  } else {
    const int hw = lane >> 5, iw = (lane & 31) < LDV ? (lane & 31) : 0;
    const double* const ldg = cr + CL.off[RTOC_CDD_MJD] + iw;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = hw + HW * k;
      gl[k] = ldg[(j < NX ? j : 0) * LDV];
    }
  }
  RTOC_XLD(gLam, N_LAM, cr + CL.off[RTOC_CDD_MJTJINV], H_LAM)
  RTOC_XLD(gFF, N_FF, cr + CL.off[RTOC_CDD_QFF], H_FF)
  RTOC_XLD(gQF, N_QF, cr + CL.off[RTOC_CDD_QQF], H_QF)
  if constexpr (NP > 0) {
    RTOC_XLD(gXP, N_XP, cr + CL.off[RTOC_CDD_QXUP], H_XP)
    RTOC_XLD(gUP, N_UP, cr + CL.off[RTOC_CDD_QUUPTR], H_UP)
  }
  // the joint-limit rows' data too: rows r = lane and lane + 64 cover the 72 rows of a quadruped, further rows in the loop below
  double* const nr = a.con ? a.con + ((size_t)b * a.nstages + st) * a.nl.stride : nullptr;
  constexpr int PRE = 2;
  rtoc_box_row prow[PRE];
  double pslack[PRE], pdual[PRE], pres[PRE], pcmpl[PRE];
  if (a.con) {
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
      const int r = lane + 64 * k, rc = r < a.nrows ? r : 0;
      prow[k] = a.rows[rc];
      pslack[k] = nr[a.nl.off[RTOC_CON_SLACK] + rc], pdual[k] = nr[a.nl.off[RTOC_CON_DUAL] + rc];
      pres[k] = nr[a.nl.off[RTOC_CON_RESIDUAL] + rc], pcmpl[k] = nr[a.nl.off[RTOC_CON_CMPL] + rc];
    }
  }
  const rtoc_grid g = a.grid[st];
  const bool impact = g.type == RTOC_GRID_IMPACT;
  const int nf = g.dimf, nvf = NV + nf, ns = impact ? 0 : g.dims;
  const double dt = grid_dt(a.grid, a.dt_inst, b, a.nstages, st);
  const double* Lr = cr + CL.off[RTOC_CDD_MJIDC];
  // Qafqv / Qafu_full are NOT read back: Qafqv dx + Qafu du is rebuilt from t = MJtJinv[:, u] du - MJtJinv_dIDCdqv dx,
  // which the primal expansion needs anyway, and the small input blocks Qaa (diagonal), Qff, Qqf
  //   rows < nv :  Qaa_i t_i                         (contact_dynamics.cpp:68-70,76-78)
  //   rows >= nv:  Qff t_f - Qqf^T dx_q              (:71-75,79-80)
  // -- 1.6k doubles per grid point that the condensation then need not store (RTOC_OPT_CONDENSE_KEEP_QAF)
  const double* Qaa = cr + CL.off[RTOC_CDD_QAA];
  double* laf = cr + CL.off[RTOC_CDD_LAF];
  const double* haf = cr + CL.off[RTOC_CDD_HAF];
  const double* lup = cr + CL.off[RTOC_CDD_LUP];
  const double* Phia = cr + CL.off[RTOC_CDD_PHIA];
  __shared__ double sdx[NX + 8], sdu[NU + 8], sg[NV + 8], sxi[LDS_ + 8], slaf[LDV + 8], st_[LDV + 8], sda[NV + 8];
  for (int i = lane; i < NX; i += 64) sdx[i] = dr[DL.off[RTOC_DIR_DX] + i];
  if (lane < NU) sdu[lane] = impact ? 0.0 : dr[DL.off[RTOC_DIR_DU] + lane];
  for (int i = lane; i < NV; i += 64) sg[i] = dn[DL.off[RTOC_DIR_DLMDGMM] + NV + i];
  if (lane < ns) sxi[lane] = dr[DL.off[RTOC_DIR_DXI] + lane];
  // the row-wise vectors of the walks, one per lane
  static_assert(LDV <= 64, "one lane per row of the nvf-row products");
  constexpr int H = (LDV <= 32) ? 2 : 1;  // with nvf <= 32 the two half-waves split the columns
  const int hh = (H == 2) ? (lane >> 5) : 0, i0 = (H == 2) ? (lane & 31) : lane;
  const int ir0 = i0 < nvf ? i0 : 0;
  const double vLr = Lr[ir0], vLaf = laf[ir0], vHaf = haf[ir0], vQaa = Qaa[ir0 < NV ? ir0 : 0];
  double dtsv = 0.0;
  if (!impact && g.num_grids_in_phase > 0)
    dtsv = (dr[DL.off[RTOC_DIR_DTS] + 1] - dr[DL.off[RTOC_DIR_DTS] + 0]) / (double)g.num_grids_in_phase;
  const bool use_dts = (dtsv < -2.220446049250313e-16 || dtsv > 2.220446049250313e-16);
  RTOC_CPROF(33);
  if constexpr (!E::STREAM_LD) {
    RTOC_XST(LD, gLD, N_LD, H_LD)
  }
  RTOC_XST(Lam, gLam, N_LAM, H_LAM)
  RTOC_XST(Qff, gFF, N_FF, H_FF)
  RTOC_XST(Qqf, gQF, N_QF, H_QF)
  if constexpr (NP > 0) {
    RTOC_XST(Qxup, gXP, N_XP, H_XP)
    RTOC_XST(Quuptr, gUP, N_UP, H_UP)
  }
#undef RTOC_XLD
#undef RTOC_XST
  RTOC_CPROF(34);
  __syncthreads();
  RTOC_CPROF(35);
  // primal (:167-174, impact :83-88) and the laf accumulation of the dual (:190-198, impact :91-95)
  {
    const bool row = i0 < nvf;
    double acc = 0.0;
    if constexpr (E::STREAM_LD) {
#pragma unroll
      for (int k = 0; k < NLD; ++k) {
        const int j = hh + H * k;
        if (j < NX) acc -= gl[k] * sdx[j];
      }
    } else {
      for (int j = hh; j < NX; j += H) acc -= LD[ir0 + j * LDV] * sdx[j];
    }
    if (!impact)
      for (int j = hh; j < NU; j += H) acc += Lam[ir0 + (NP + j) * LDV] * sdu[j];
    if (H == 2) acc += __shfl_xor(acc, 32, 64);
    if (row && hh == 0) {
      st_[i0] = acc;  // t_i
      double daf = acc - vLr;
      if (i0 >= NV) daf = -daf;
      else sda[i0] = daf;  // the acceleration-limit rows expand with da
      dr[DL.off[RTOC_DIR_DAF] + i0] = daf;
    }
  }
  __syncthreads();
  {
    const bool row = i0 < nvf;
    double accl = 0.0;
    if (ir0 < NV) {
      if (hh == 0) {
        accl = vLaf + vQaa * st_[ir0] + (impact ? 1.0 : dt) * sg[ir0];
        if (NS > 0 && ns > 0)
          for (int l = 0; l < ns; ++l) accl += Phia[l + (size_t)ir0 * LDS_] * sxi[l];
      }
    } else {
      const int f = ir0 - NV;
      for (int k = hh; k < nf; k += H) accl += Qff[f + k * LDF] * st_[NV + k];
      for (int j = hh; j < NV; j += H) accl -= Qqf[j + f * NV] * sdx[j];
      if (hh == 0) accl += vLaf;
    }
    if (H == 2) accl += __shfl_xor(accl, 32, 64);
    if (use_dts) accl += dtsv * vHaf;
    if (row && hh == 0) {
      laf[i0] = accl;  // the reference updates data.laf() in place as well
      slaf[i0] = accl;
    }
  }
  RTOC_CPROF(36);
  // dnu_passive (:178-188): NP rows of NU + NX + NV terms, each row over PARTS lanes (lane = row + NP * part), summed through LDS
  if constexpr (NP > 0) {
    constexpr int PARTS = 64 / NP;
    __shared__ double spart[64];
    const int prow_ = lane % NP, part = lane / NP;
    double acc = 0.0;
    if (!impact && part < PARTS) {
      for (int j = part; j < NU; j += PARTS) acc -= Quuptr[prow_ + j * NP] * sdu[j];
      for (int j = part; j < NX; j += PARTS) acc -= Qxup[j + prow_ * NX] * sdx[j];
      for (int j = part; j < NV; j += PARTS) acc -= dt * Lam[prow_ + j * LDV] * sg[j];
    }
    spart[lane] = (part < PARTS) ? acc : 0.0;
    __syncthreads();
    if (!impact && lane < NP) {
      double sum = -lup[lane];
#pragma unroll
      for (int p_ = 0; p_ < PARTS; ++p_) sum += spart[lane + NP * p_];
      dr[DL.off[RTOC_DIR_DNUP] + lane] = sum;
    }
  } else {
    __syncthreads();
  }
  RTOC_CPROF(37);
  // ================= PDIPM expansion + fraction-to-boundary (constraints.cpp:360-458) =================
  if (a.con && !impact) {
    const int* no = a.nl.off;
    double fp = 1.0, fd = 1.0;
    auto one_row = [&](const int r, const rtoc_box_row& row, const double slack, const double dual, const double res, const double cmpl) {
      if (r < a.nrows && g.time_stage >= row.level) {
        const double dz = row.var == RTOC_VAR_U ? sdu[row.index]
                                                : (row.var == RTOC_VAR_V ? sdx[NV + row.index]
                                                                         : (row.var == RTOC_VAR_A ? sda[row.index] : sdx[row.index]));
        const double dslack = -row.sign * dz - res;
        const double ddual = -(dual * dslack + cmpl) / slack;
        nr[no[RTOC_CON_DSLACK] + r] = dslack;
        nr[no[RTOC_CON_DDUAL] + r] = ddual;
        const double fs = -a.tau * (slack / dslack), fdd = -a.tau * (dual / ddual);  // pdipm.hxx:121-142
        if (fs > 0.0 && fs < 1.0) fp = fmin(fp, fs);
        if (fdd > 0.0 && fdd < 1.0) fd = fmin(fd, fdd);
      }
    };
#pragma unroll
    for (int k = 0; k < PRE; ++k) one_row(lane + 64 * k, prow[k], pslack[k], pdual[k], pres[k], pcmpl[k]);
    for (int r = lane + 64 * PRE; r < a.nrows; r += 64)
      one_row(r, a.rows[r], nr[no[RTOC_CON_SLACK] + r], nr[no[RTOC_CON_DUAL] + r], nr[no[RTOC_CON_RESIDUAL] + r], nr[no[RTOC_CON_CMPL] + r]);
    // min over the wave, then over the instance: positive doubles order like their bit patterns
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      fp = fmin(fp, __shfl_xor(fp, off, 64));
      fd = fmin(fd, __shfl_xor(fd, off, 64));
    }
    if (lane == 0) {
      atomicMin(&a.steps[2 * b + 0], (unsigned long long)__double_as_longlong(fp));
      atomicMin(&a.steps[2 * b + 1], (unsigned long long)__double_as_longlong(fd));
    }
  }
  RTOC_CPROF(38);
  // dbetamu = -MJtJinv * laf (:201, impact :95)
  {
    const bool row = i0 < nvf;
    double acc = 0.0;
    for (int j = hh; j < nvf; j += H) acc -= Lam[ir0 + j * LDV] * slaf[j];
    if (H == 2) acc += __shfl_xor(acc, 32, 64);
    if (row && hh == 0) dr[DL.off[RTOC_DIR_DBETAMU] + i0] = acc;
  }
  RTOC_CPROF(39);
}

// updateSlack / updateDual (constraints_impl.hxx:167-182) with the per-instance step sizes
struct UpdArgs {
  double* con;
  const rtoc_box_row* rows;
  const rtoc_grid* grid;
  const double* steps;
  int nrows, nstages, batch;
  rtoc_record_layout nl;
};

static __global__ __launch_bounds__(64) void pdipm_update_kernel(UpdArgs a) {
  const int item = blockIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = item / nst1, st = item % nst1;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  if (g.type == RTOC_GRID_IMPACT) return;
  double* nr = a.con + ((size_t)b * a.nstages + st) * a.nl.stride;
  const int* no = a.nl.off;
  const double ps = a.steps[2 * b], ds = a.steps[2 * b + 1];
  for (int r = threadIdx.x; r < a.nrows; r += 64) {
    if (g.time_stage >= a.rows[r].level) {
      nr[no[RTOC_CON_SLACK] + r] += ps * nr[no[RTOC_CON_DSLACK] + r];
      nr[no[RTOC_CON_DUAL] + r] += ds * nr[no[RTOC_CON_DDUAL] + r];
    }
  }
}

static __global__ void fill_steps_kernel(double* steps, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) steps[i] = 1.0;
}

}  // namespace rtoc

static inline int rtoc_con_stride(const rtoc_dims* d) { return RTOC_CON_NFIELDS * ((d->nc_max + 7) & ~7); }
