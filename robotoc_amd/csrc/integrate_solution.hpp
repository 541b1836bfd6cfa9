// integrate_solution.hpp -- SplitSolution::integrate.
//
// Replaces SplitSolution::integrate (reference src/core/split_solution.cpp:58-90) as called by
// {Intermediate,Impact,Terminal}Stage::updatePrimal (intermediate_stage.cpp:187-195, impact_stage.cpp:155-162,
// terminal_stage.cpp:139-148) from DirectMultipleShooting::integrateSolution
// (direct_multiple_shooting.cpp:212-241): every member is advanced by primal_step x direction.
// robot.integrateConfiguration (:62) is Pinocchio's manifold update: joints additively, a free-flyer base by the SE(3)
// exponential -- M <- M exp6(step dq_base): rotation exp(w), translation R V(w) v, quaternion product and
// re-normalisation (restated from the Lie-group formulas; Pinocchio absent).  Pure streaming: HBM-bound.
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
  // q (:62): fixed base q += step dq; floating base: q[7+j] += step dq[6+j] and the base by the SE(3) exponential
  {
    double* q = s + a.sl.off[RTOC_SOL_Q];
    const int nb = a.np == 6 ? 6 : 0;
    for (int i = lane; i < nv - nb; i += 64) q[(nb ? 7 : 0) + i] += step * dx[nb + i];
    // step 0 (instances the convergence mask froze): the iterate is kept bit for bit, no re-normalisation
    if (nb && lane == 0 && step != 0.0) {
      const double vx = step * dx[0], vy = step * dx[1], vz = step * dx[2], wx = step * dx[3], wy = step * dx[4], wz = step * dx[5];
      const double th = sqrt(wx * wx + wy * wy + wz * wz);
      double A, B;
      if (th < 1e-8) {
        A = 0.5, B = 1.0 / 6.0;
      } else {
        A = (1.0 - cos(th)) / (th * th), B = (th - sin(th)) / (th * th * th);
      }
      // V(w) v = v + A w x v + B w x (w x v)
      const double cx = wy * vz - wz * vy, cy = wz * vx - wx * vz, cz = wx * vy - wy * vx;
      const double ex = wy * cz - wz * cy, ey = wz * cx - wx * cz, ez = wx * cy - wy * cx;
      const double ux = vx + A * cx + B * ex, uy = vy + A * cy + B * ey, uz = vz + A * cz + B * ez;
      const double x = q[3], y = q[4], z = q[5], w = q[6];
      // R(quat) (V v)
      q[0] += (1 - 2 * (y * y + z * z)) * ux + 2 * (x * y - z * w) * uy + 2 * (x * z + y * w) * uz;
      q[1] += 2 * (x * y + z * w) * ux + (1 - 2 * (x * x + z * z)) * uy + 2 * (y * z - x * w) * uz;
      q[2] += 2 * (x * z - y * w) * ux + 2 * (y * z + x * w) * uy + (1 - 2 * (x * x + y * y)) * uz;
      // quat (x) (sin(t/2)/t w, cos(t/2)), normalised
      const double sc = th < 1e-8 ? 0.5 - th * th / 48.0 : sin(0.5 * th) / th, ew = cos(0.5 * th);
      const double e0 = sc * wx, e1 = sc * wy, e2 = sc * wz;
      const double r0 = w * e0 + x * ew + y * e2 - z * e1, r1 = w * e1 - x * e2 + y * ew + z * e0,
                   r2 = w * e2 + x * e1 - y * e0 + z * ew, r3 = w * ew - x * e0 - y * e1 - z * e2;
      const double n = 1.0 / sqrt(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
      q[3] = r0 * n, q[4] = r1 * n, q[5] = r2 * n, q[6] = r3 * n;
    }
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
