// unconstr_constraints.hpp -- the joint-limit inequality rows (PDIPM) of the unconstrained solver iteration, on the device.
//
// The contact path receives the rows' slack / dual / residual / cmpl from the CPU side (Constraints::linearizeConstraints
// needs nothing but the iterate for joint limits, yet it sits with the cost there).  The unconstrained path is closed on
// the device (rtoc_unconstr_update_solution), so the whole life of the rows runs here, per (instance, grid point), one lane
// per row -- the six Joint{Position,Velocity,Torques}{Lower,Upper}Limit components (reference
// src/constraints/joint_position_lower_limit.cpp:40-87 and its five siblings; include/robotoc/constraints/pdipm.hxx):
//   INIT      setSlackAndDual: slack = -g, clipped at sqrt(barrier); dual = barrier / slack         (pdipm.hxx:12-23)
//   LINEARIZE evalConstraint + evalDerivatives: residual = g + slack, cmpl = slack dual - barrier, l_z += sign dual
//   CONDENSE  condenseSlackAndDual: Q_zz += dual / slack, cond = (dual residual - cmpl) / slack, l_z += sign cond
//   EXPAND    expandSlackAndDual: dslack = -sign dz - residual, ddual = -(dual dslack + cmpl) / slack, and the
//             fraction-to-boundary step sizes (pdipm.hxx:121-142) into RTOC_BUF_STEP (atomicMin on the bit pattern)
// with g = sign * z - bound (rtoc_box_row), z an entry of q, v, u or -- contact path only -- a.  Record convention of the unconstrained path
// (rtoc_unconstr_condense): the u rows act on CDD.la (= lu) and CDD.Qaa (= diag Quu); q / v rows on KKT.lx and diag Qxx.
// A row is active on a grid point iff time_stage >= level, never on the terminal one (constraints_data.cpp:20-45).
// The contact path (rtoc_contact_init_constraints / rtoc_contact_eval_kkt) uses INIT and LINEARIZE with `contact` set: the
// u rows then act on KKT.lu, impact grids (time_stage = -1) carry no rows; condensation and expansion of the rows are the
// contact path's own kernels (condense.hpp).
#pragma once
#include "device_utils.hpp"
#include "../../include/rtoc.h"

namespace rtoc {

struct UboxArgs {
  const double* sol;
  double* kkt;
  double* cdd;
  double* con;
  const double* dir;
  const rtoc_box_row* rows;
  const int* entry;           // CSR of the rows per primal entry (q_0.., v_0.., u_0..): [ne + 1] offsets, then row ids
  const double* bounds;
  const rtoc_grid* grid;
  unsigned long long* steps;  // [batch][2] bit patterns
  int nstages, batch, nrows, nv, nu, mode;
  double barrier, tau;
  int sol_stride, kkt_stride, cdd_stride, con_stride, dir_stride;
  int o_q, o_v, o_u, o_a;       // RTOC_BUF_SOL
  int o_qxx, o_lx;              // RTOC_BUF_KKT
  int o_qaa, o_la;              // RTOC_BUF_CDD (diag Quu, lu)
  int o_dx, o_du;               // RTOC_BUF_DIR
  int contact;                  // contact path (rtoc_contact_eval_kkt): the u rows act on KKT.lu; INIT / LINEARIZE only
  int q_shift, o_lu;            // nq - nv (the joint entries of q sit one further with a free-flyer's quaternion)
  rtoc_record_layout nl;
};
enum { UBOX_INIT = 0, UBOX_LINEARIZE = 1, UBOX_CONDENSE = 2, UBOX_EXPAND = 3 };

static __global__ __launch_bounds__(64) void unconstr_box_kernel(UboxArgs a) {
  const int lane = threadIdx.x;
  const int nst1 = a.nstages - 1;
  const int b = blockIdx.x / nst1, st = blockIdx.x % nst1;  // the terminal grid point has no rows
  if (b >= a.batch) return;
  const rtoc_grid g = a.grid[st];
  const size_t rec = (size_t)b * a.nstages + st;
  const double* const s = a.sol + rec * a.sol_stride;
  double* const kr = a.kkt ? a.kkt + rec * a.kkt_stride : nullptr;
  double* const cr = a.cdd ? a.cdd + rec * a.cdd_stride : nullptr;
  double* const nr = a.con + rec * a.con_stride;
  const double* const dr = a.dir ? a.dir + rec * a.dir_stride : nullptr;
  const int* const no = a.nl.off;
  const int nv = a.nv, nx = 2 * nv;
  double fp = 1.0, fd = 1.0;
  auto value_of = [&](const rtoc_box_row& w) {
    return w.var == RTOC_VAR_Q ? s[a.o_q + w.index + a.q_shift]
                               : w.var == RTOC_VAR_V ? s[a.o_v + w.index] : w.var == RTOC_VAR_A ? s[a.o_a + w.index] : s[a.o_u + w.index];
  };
  if (a.mode == UBOX_LINEARIZE || a.mode == UBOX_CONDENSE) {
    // one lane per primal entry, its rows in row order: a lower and an upper limit meet on the same entry, and every
    // entry is accumulated by a single lane (deterministic, no atomics) -- like the box rows of condense_kernel
    const int ne = 3 * nv + a.nu;  // q, v, u, a (rtoc_set_constraint_rows)
    const int* const rowid = a.entry + (ne + 1);
    for (int t = lane; t < ne; t += 64) {
      double grad = 0.0, hess = 0.0;
      for (int e = a.entry[t]; e < a.entry[t + 1]; ++e) {
        const int r = rowid[e];
        const rtoc_box_row w = a.rows[r];
        if (g.time_stage < w.level) continue;
        const double slack = nr[no[RTOC_CON_SLACK] + r], dual = nr[no[RTOC_CON_DUAL] + r];
        if (a.mode == UBOX_LINEARIZE) {
          nr[no[RTOC_CON_RESIDUAL] + r] = (w.sign * value_of(w) - a.bounds[r]) + slack;
          nr[no[RTOC_CON_CMPL] + r] = slack * dual - a.barrier;
          grad += w.sign * dual;
        } else {
          const double cond = (dual * nr[no[RTOC_CON_RESIDUAL] + r] - nr[no[RTOC_CON_CMPL] + r]) / slack;
          nr[no[RTOC_CON_COND] + r] = cond;
          hess += dual / slack;
          grad += w.sign * cond;
        }
      }
      if (a.contact) {   // LINEARIZE only: the condensation of these rows is condense_kernel's
        if (t < 2 * nv) kr[a.o_lx + t] += grad;
        else if (a.entry[t + 1] > a.entry[t]) {
          if (t < 2 * nv + a.nu) kr[a.o_lu + (t - 2 * nv)] += grad;
          else cr[a.o_la + (t - 2 * nv - a.nu)] += grad;   // JointAcceleration*Limit::evalDerivatives: la -/+= dual
        }
      } else if (t >= 2 * nv + a.nu) {
        // (acceleration rows exist on the contact path only: rtoc_set_constraint_rows refuses them without contacts)
      } else if (t < 2 * nv) {
        kr[a.o_lx + t] += grad;
        kr[a.o_qxx + t + (size_t)t * nx] += hess;
      } else {
        cr[a.o_la + (t - 2 * nv)] += grad;
        cr[a.o_qaa + (t - 2 * nv)] += hess;
      }
    }
    return;
  }
  for (int r = lane; r < a.nrows; r += 64) {
    const rtoc_box_row w = a.rows[r];
    if (g.time_stage < w.level) continue;
    if (a.mode == UBOX_INIT) {
      double slack = -(w.sign * value_of(w) - a.bounds[r]);
      const double sb = sqrt(a.barrier);
      if (slack < sb) slack = sb;
      nr[no[RTOC_CON_SLACK] + r] = slack;
      nr[no[RTOC_CON_DUAL] + r] = a.barrier / slack;
      continue;
    }
    // UBOX_EXPAND
    const double slack = nr[no[RTOC_CON_SLACK] + r], dual = nr[no[RTOC_CON_DUAL] + r];
    const double dz = w.var == RTOC_VAR_Q ? dr[a.o_dx + w.index] : w.var == RTOC_VAR_V ? dr[a.o_dx + nv + w.index] : dr[a.o_du + w.index];  // (no RTOC_VAR_A here)
    const double residual = nr[no[RTOC_CON_RESIDUAL] + r], cmpl = nr[no[RTOC_CON_CMPL] + r];
    const double dslack = -w.sign * dz - residual;
    const double ddual = -(dual * dslack + cmpl) / slack;
    nr[no[RTOC_CON_DSLACK] + r] = dslack;
    nr[no[RTOC_CON_DDUAL] + r] = ddual;
    const double fs = -a.tau * (slack / dslack), fdd = -a.tau * (dual / ddual);
    if (fs > 0.0 && fs < 1.0 && fs < fp) fp = fs;
    if (fdd > 0.0 && fdd < 1.0 && fdd < fd) fd = fdd;
  }
  if (a.mode == UBOX_EXPAND) {
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
}

}  // namespace rtoc
