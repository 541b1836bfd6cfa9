// integrate_solution.hpp -- the Euclidean part of SplitSolution::integrate.
//
// Replaces SplitSolution::integrate (reference src/core/split_solution.cpp:58-90) as called by
// {Intermediate,Impact,Terminal}Stage::updatePrimal (intermediate_stage.cpp:187-195, impact_stage.cpp:155-162,
// terminal_stage.cpp:139-148) from DirectMultipleShooting::integrateSolution
// (direct_multiple_shooting.cpp:212-241): every member is advanced by primal_step x direction.
// robot.integrateConfiguration (:62) is Pinocchio's manifold update; its joint (Euclidean) part is done
// here, the 7 floating-base entries of q are left to the CPU side.  Pure streaming: HBM-bound.
#pragma once
#include "device_utils.hpp"
#include "../../include/rtoc.h"

namespace rtoc {

struct IntArgs {
  double* sol;
  const double* dir;
  const double* steps;  // [batch][2]: primal, dual
  const rtoc_grid* grid;
  int nstages, batch;
  int nv, nu, np, nf_max, ns_max;
  rtoc_record_layout sl, dl;
};

static __global__ __launch_bounds__(64) void integrate_solution_kernel(IntArgs a) {
  const int lane = threadIdx.x;
  const int item = blockIdx.x;
  const int b = item / a.nstages, st = item % a.nstages;
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  const bool impact = g.type == RTOC_GRID_IMPACT;
  const double step = a.steps[2 * b];
  double* s = a.sol + (size_t)item * a.sl.stride;
  const double* d = a.dir + (size_t)item * a.dl.stride;
  const int nv = a.nv, nvf = nv + a.nf_max;
  auto axpy = [&](int sfield, int n, const double* dv) {
    double* dst = s + a.sl.off[sfield];
    for (int i = lane; i < n; i += 64) dst[i] += step * dv[i];
  };
  const double* dx = d + a.dl.off[RTOC_DIR_DX];
  const double* dl = d + a.dl.off[RTOC_DIR_DLMDGMM];
  const double* daf = d + a.dl.off[RTOC_DIR_DAF];
  const double* dbm = d + a.dl.off[RTOC_DIR_DBETAMU];
  // joint part of q (:62): fixed base q += step dq; floating base q[7+j] += step dq[6+j]
  {
    double* q = s + a.sl.off[RTOC_SOL_Q];
    const int nb = a.np == 6 ? 6 : 0;
    for (int i = lane; i < nv - nb; i += 64) q[(nb ? 7 : 0) + i] += step * dx[nb + i];
  }
  axpy(RTOC_SOL_V, nv, dx + nv);                                   // (:63)
  if (!impact) {
    axpy(RTOC_SOL_A, nv, daf);                                     // a += step da (:65)
    axpy(RTOC_SOL_U, a.nu, d + a.dl.off[RTOC_DIR_DU]);             // (:67)
  } else {
    axpy(RTOC_SOL_A, nv, daf);                                     // dv += step ddv (:71); slot A holds dv
    for (int i = lane; i < a.nu; i += 64) s[a.sl.off[RTOC_SOL_U] + i] = 0.0;  // u.setZero() (:72)
  }
  axpy(RTOC_SOL_LMD, nv, dl);                                      // (:74)
  axpy(RTOC_SOL_GMM, nv, dl + nv);                                 // (:75)
  axpy(RTOC_SOL_BETA, nv, dbm);                                    // (:76)
  if (a.np == 6 && !impact) axpy(RTOC_SOL_NUP, a.np, d + a.dl.off[RTOC_DIR_DNUP]);  // (:77-79)
  if (g.dimf > 0) {
    axpy(RTOC_SOL_F, g.dimf, daf + nv);                            // (:81)
    axpy(RTOC_SOL_MU, g.dimf, dbm + nv);                           // (:83)
  }
  if (g.dims > 0 && !impact) axpy(RTOC_SOL_XI, g.dims, d + a.dl.off[RTOC_DIR_DXI]);  // (:86-89)
  (void)nvf;
}

}  // namespace rtoc
