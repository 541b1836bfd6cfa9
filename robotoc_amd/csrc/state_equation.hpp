// state_equation.hpp -- floating-base corrections of the linearised state equation.
//
// Replaces (reference src/dynamics/state_equation.cpp:68-109, impact_state_equation.cpp:57-72)
//   correctLinearizeStateEquation / correctLinearizeImpactStateEquation : the 6x6 corner of Fqq, Fqv
//       and the 6-heads of Fq, fq are multiplied by -Fqq_inv,
//   correctCostateDirection       : dlmd.head<6>() <- -Fqq_prev_inv^T dlmd.head<6>(),
//   computeInitialStateDirection  : dq0.head<6>()  <- -Fqq_prev_inv dq0.head<6>().
// Fqq_inv / Fqq_prev_inv (the SE3 Jlog inverses of SE3JacobianInverse) come from the CPU-side
// linearisation in the RTOC_BUF_SE3 record of every grid point.  O(36) flops per grid point:
// pure HBM latency, one 64-lane wave per grid point.
#pragma once
#include "device_utils.hpp"
#include "../../include/rtoc.h"

namespace rtoc {

struct SeArgs {
  double* kkt;
  double* dir;
  double* dx0;
  const double* se3;  // [batch][nstages][RTOC_SE3_STRIDE]
  const rtoc_grid* grid;
  int nstages, batch;
  rtoc_record_layout kl, dl;
  int nx;
  const double* dt_inst;  // per-instance time steps or nullptr (grid_dt)
};

// mode 0: correctLinearize(Impact)StateEquation on every non-terminal grid point
// mode 1: correctCostateDirection on every grid point
// mode 2: computeInitialStateDirection on the DX0 buffer (one block per instance)
template <int MODE>
__global__ __launch_bounds__(64) void state_correction_kernel(SeArgs a) {
  const int lane = threadIdx.x;
  const int i = lane % 6, j = lane / 6;  // lanes 0..35: entry (i, j) of a 6x6 block
  __shared__ double sIn[64];
  if (MODE == 2) {
    const int b = blockIdx.x;
    if (b >= a.batch) return;
    const double* inv = a.se3 + ((size_t)b * a.nstages) * RTOC_SE3_STRIDE + RTOC_SE3_FQQ_PREV_INV;
    double* dq = a.dx0 + (size_t)b * a.nx;
    if (lane < 6) sIn[lane] = dq[lane];
    __syncthreads();
    if (lane < 6) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) acc += inv[lane + 6 * k] * sIn[k];
      dq[lane] = -acc;
    }
    return;
  }
  const int item = blockIdx.x;
  const int b = item / a.nstages, st = item % a.nstages;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  const double* se = a.se3 + (size_t)item * RTOC_SE3_STRIDE;
  if (MODE == 1) {
    double* dl = a.dir + (size_t)item * a.dl.stride + a.dl.off[RTOC_DIR_DLMDGMM];
    const double* inv = se + RTOC_SE3_FQQ_PREV_INV;
    if (lane < 6) sIn[lane] = dl[lane];
    __syncthreads();
    if (lane < 6) {
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) acc += inv[k + 6 * lane] * sIn[k];  // Fqq_prev_inv^T
      dl[lane] = -acc;
    }
    return;
  }
  if (g.type == RTOC_GRID_TERMINAL) return;
  const bool impact = g.type == RTOC_GRID_IMPACT;
  const int nx = a.nx, nv = nx / 2;
  double* kr = a.kkt + (size_t)item * a.kl.stride;
  double* Fxx = kr + a.kl.off[RTOC_KKT_FXX];
  double* Fx = kr + a.kl.off[RTOC_KKT_FX];
  double* fx = kr + a.kl.off[RTOC_KKT_FFX];
  const double* inv = se + RTOC_SE3_FQQ_INV;
  // Fqq_tmp = Fqq.topLeftCorner<6,6>(), Fq_tmp, fq_tmp
  if (lane < 36) sIn[lane] = Fxx[i + (size_t)j * nx];
  if (lane >= 36 && lane < 42) sIn[lane] = Fx[lane - 36];
  if (lane >= 42 && lane < 48) sIn[lane] = fx[lane - 42];
  __syncthreads();
  if (lane < 36) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc += inv[i + 6 * k] * sIn[k + 6 * j];
    Fxx[i + (size_t)j * nx] = -acc;                                        // Fqq corner (:81 / impact :69)
    if (!impact) Fxx[i + (size_t)(nv + j) * nx] = -grid_dt(a.grid, a.dt_inst, b, a.nstages, st) * inv[i + 6 * j];  // Fqv corner (:82)
  } else if (lane < 42) {
    const int r = lane - 36;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc += inv[r + 6 * k] * sIn[36 + k];
    Fx[r] = -acc;  // Fq head (:84 / impact :70)
  } else if (lane < 48 && !impact) {
    const int r = lane - 42;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) acc += inv[r + 6 * k] * sIn[42 + k];
    fx[r] = -acc;  // fq head (:86)
  }
}

}  // namespace rtoc
