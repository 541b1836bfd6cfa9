// kkt_error.hpp -- squared KKT residual of every OCP instance, evaluated at the evalKKT boundary.
//
// Replaces the kkt_error accumulation of IntermediateStage / ImpactStage / TerminalStage::evalKKT
// (reference src/ocp/intermediate_stage.cpp:132: data.KKTError() + kkt_residual.KKTError(), summed over
// the horizon by DirectMultipleShooting::evalKKT, direct_multiple_shooting.cpp:129-159) and the final
// sqrt of OCPSolver::KKTError() (src/solver/ocp_solver.cpp:429-431; the STO term is not part of it here):
//   SplitKKTResidual::KKTError()      Fx^2 + P^2 + lx^2 + lu^2 + la^2 + ldv^2 + lf^2     (split_kkt_residual.hxx:90-104)
//   ContactDynamicsData::KKTError()   IDC^2 + lu_passive^2                               (contact_dynamics_data.hpp:204-206)
//   ConstraintComponentData::KKTError() residual^2 + cmpl^2 of the active rows           (constraint_component_data.hpp:122-124)
// on the PRE-condensation records (the reference evaluates it before condenseSlackAndDual /
// condenseContactDynamics).  One wave per instance, stages in order, fixed lane-strided summation
// order and a butterfly at the end: deterministic.  Pure streaming read (HBM-bound, ~11 KB/stage).
#pragma once
#include "device_utils.hpp"
#include "../../include/rtoc.h"

namespace rtoc {

struct KktErrArgs {
  const double* kkt;
  const double* cdd;   // may be null (no contact-dynamics terms)
  const double* con;   // may be null (no constraint rows)
  const rtoc_box_row* rows;
  const rtoc_grid* grid;
  double* out;         // [batch] sqrt of the sum
  int nstages, batch, nrows, cone_contacts, cone_dim, cone_rows, nc_max;
  int nv, nu, np, nx;
  rtoc_record_layout kl, cl, nl;
};

static __global__ __launch_bounds__(64) void kkt_error_kernel(KktErrArgs a) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  if (b >= a.batch) return;
  double acc = 0.0;
  auto sq = [&](const double* p, int n) {
    for (int i = lane; i < n; i += 64) {
      const double v = p[i];
      acc += v * v;
    }
  };
  for (int st = 0; st < a.nstages; ++st) {
    const rtoc_grid g = a.grid[st];
    const size_t rec = (size_t)b * a.nstages + st;
    const double* kr = a.kkt + rec * a.kl.stride;
    const bool terminal = g.type == RTOC_GRID_TERMINAL, impact = g.type == RTOC_GRID_IMPACT;
    sq(kr + a.kl.off[RTOC_KKT_LX], a.nx);
    if (terminal) continue;
    sq(kr + a.kl.off[RTOC_KKT_FX], a.nx);
    if (!impact) {
      sq(kr + a.kl.off[RTOC_KKT_LU], a.nu);
      if (g.dims > 0) sq(kr + a.kl.off[RTOC_KKT_PRES], g.dims);
    }
    if (a.cdd) {
      const double* cr = a.cdd + rec * a.cl.stride;
      sq(cr + a.cl.off[RTOC_CDD_LA], a.nv);  // la (contact grids) / ldv (impact grids)
      sq(cr + a.cl.off[RTOC_CDD_LF], g.dimf);
      sq(cr + a.cl.off[RTOC_CDD_IDC], a.nv + g.dimf);
      if (!impact) sq(cr + a.cl.off[RTOC_CDD_LUP], a.np);
    }
    if (a.con) {
      const double* nr = a.con + rec * a.nl.stride;
      if (!impact)
        for (int r = lane; r < a.nrows; r += 64)
          if (g.time_stage >= a.rows[r].level) {
            const double x = nr[a.nl.off[RTOC_CON_RESIDUAL] + r], y = nr[a.nl.off[RTOC_CON_CMPL] + r];
            acc += x * x + y * y;
          }
      if (a.cone_contacts > 0) {
        const int row0 = a.nc_max - a.cone_rows * a.cone_contacts, n = a.cone_rows * (g.dimf / a.cone_dim);
        for (int r = lane; r < n; r += 64) {
          const double x = nr[a.nl.off[RTOC_CON_RESIDUAL] + row0 + r], y = nr[a.nl.off[RTOC_CON_CMPL] + row0 + r];
          acc += x * x + y * y;
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) a.out[b] = sqrt(acc);
}

}  // namespace rtoc
