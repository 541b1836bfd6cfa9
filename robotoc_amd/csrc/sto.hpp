// sto.hpp -- the switching-time half of OCPSolver::updateSolution on the device.
//
// reference: src/sto/switching_time_optimization.cpp (initConstraints :45-52, evalKKT :79-137, computeStepSizes :140-158,
// maxPrimal/DualStepSize :161-178, integrateSolution :181-206), src/sto/sto_constraints.cpp (minimum dwell times as PDIPM rows:
// setSlackAndDual :148-172, evalConstraint :175-188, linearizeConstraints :191-198, condenseSlackAndDual :201-211,
// expandSlackAndDual :214-220, step sizes :223-236, updates :239-254, computeDwellTimes :257-276),
// include/robotoc/constraints/pdipm.hxx, and TimeDiscretization::correctTimeSteps (src/ocp/time_discretization.cpp:179-221).
//
// The batch shares the grid STRUCTURE (event order, grid points per phase); with switching-time optimisation every instance
// owns its event times ts[b][e] (e = 0 .. nev-1 in grid order: ContactSequence::impactTime / liftTime), hence its own time
// steps dt_inst[b][stage] (device_utils.hpp: grid_dt), its own dwell-time rows (nev + 1 of them) and its own step.  All of it is
// scalar work, O(#grid points) per instance: one thread per instance.
//
// Dwell-time rows: g_p = min_dwell_p - dwell_p <= 0 for phase p, dwell_p = ts_p - ts_{p-1} (ts_{-1} = t0, ts_nev = t0 + T).
// Jacobian with respect to the event times: J[p][p] = -1 (p < nev), J[p][p-1] = +1 (p >= 1) (sto_constraints.cpp:160-167), so
// (J^T v)_e = v_{e+1} - v_e and diag(J^T diag(c) J)_e = c_e + c_{e+1}.  The reference scatters only the DIAGONAL of Qtt into the
// grid points (switching_time_optimization.cpp:105-118).
//
// STO cost (STOCostFunction, src/sto/sto_cost_function.cpp): user-defined components, none shipped with the reference (the
// examples pass an empty one); a host that has components evaluates them and hands the gradient / Hessian diagonal over
// (rtoc_sto_set_cost_terms), like rtoc_sto_eval_kkt always did.
#pragma once
#include "device_utils.hpp"

namespace rtoc {

#define RTOC_STO_MAX_EVENTS 15  // nev + 1 dwell-time rows <= 16

// per-instance constraint record: [6][RTOC_STO_MAX_EVENTS + 1] = slack, dual, residual, cmpl, dslack, ddual
#define RTOC_STO_CON_STRIDE (6 * (RTOC_STO_MAX_EVENTS + 1))

struct StoDevArgs {
  double* kkt;
  const double* dir;
  const rtoc_grid* grid;
  double* ts;               // [batch][nev] event times
  double* dt_inst;          // [batch][nstages]
  double* con;              // [batch][RTOC_STO_CON_STRIDE]
  const double* min_dwell;  // [nev + 1]
  const double* cost_lt;    // [batch][nev] or nullptr
  const double* cost_qtt;   // [batch][nev] or nullptr
  double* lt;               // [batch][nev] out: gradient handed to the scatter
  double* qtt;              // [batch][nev] out: Hessian diagonal handed to the scatter
  double* err;              // [batch] squared STO KKT term
  double* kkterr;           // [batch] OCPSolver::KKTError() (sqrt), updated in place
  double* steps;            // [batch][2] max primal / dual step
  const int* active;        // [batch] or nullptr: 0 = the instance has converged and keeps its iterate
  int nstages, batch, nev;
  int kkt_stride, scal_off, dir_stride, dts_off;
  double t0, T, barrier, tau, sto_reg;
};

__device__ __forceinline__ void sto_dwell_times(const StoDevArgs& a, const double* ts, double* dwell) {
  double prev = a.t0;
  for (int e = 0; e < a.nev; ++e) {
    dwell[e] = ts[e] - prev;
    prev = ts[e];
  }
  dwell[a.nev] = a.t0 + a.T - prev;
}

// TimeDiscretization::correctTimeSteps (time_discretization.cpp:179-221), dt only (grid times are read by nothing on the device)
static __global__ void sto_time_steps_kernel(StoDevArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  const int N = a.nstages - 1;
  const double* ts = a.ts + (size_t)b * a.nev;
  double* dt = a.dt_inst + (size_t)b * a.nstages;
  int prev_stage = 0, e = 0;
  double prev_t = a.t0;
  for (int i = 0; i < N; ++i) {
    const int ty = a.grid[i].type, tn = a.grid[i + 1].type;
    if (ty == RTOC_GRID_IMPACT) {
      const double te = e < a.nev ? ts[e] : prev_t;
      const double d = (te - prev_t) / (double)a.grid[i - 1].num_grids_in_phase;
      for (int j = prev_stage; j <= i - 1; ++j) dt[j] = d;
      dt[i] = 0.0;
      prev_t = te;
      prev_stage = i + 1;
      ++e;
      ++i;  // (:197: the grid point after an impact is not examined)
    } else if (tn == RTOC_GRID_LIFT) {
      const double te = e < a.nev ? ts[e] : prev_t;
      const double d = (te - prev_t) / (double)a.grid[i].num_grids_in_phase;
      for (int j = prev_stage; j <= i; ++j) dt[j] = d;
      prev_t = te;
      prev_stage = i + 1;
      ++e;
    } else if (tn == RTOC_GRID_TERMINAL) {
      const double d = (a.t0 + a.T - prev_t) / (double)a.grid[i].num_grids_in_phase;
      for (int j = prev_stage; j <= i; ++j) dt[j] = d;
    }
  }
  dt[N] = 0.0;
}

// SwitchingTimeOptimization::initConstraints -> STOConstraints::setSlackAndDual (sto_constraints.cpp:148-172)
static __global__ void sto_init_kernel(StoDevArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  constexpr int NP = RTOC_STO_MAX_EVENTS + 1;
  double dwell[NP];
  sto_dwell_times(a, a.ts + (size_t)b * a.nev, dwell);
  double* c = a.con + (size_t)b * RTOC_STO_CON_STRIDE;
  const double sb = sqrt(a.barrier);
  for (int p = 0; p <= a.nev; ++p) {
    double slack = -(a.min_dwell[p] - dwell[p]);
    if (slack < sb) slack = sb;                 // pdipm::setSlackAndDualPositive (pdipm.hxx:12-23)
    c[0 * NP + p] = slack;
    c[1 * NP + p] = a.barrier / slack;
    c[2 * NP + p] = c[3 * NP + p] = c[4 * NP + p] = c[5 * NP + p] = 0.0;
  }
}

// SwitchingTimeOptimization::evalKKT (:79-137): quadratizeCost (handed-over terms + regularisation), linearizeConstraints,
// condenseSlackAndDual, the scatter into h / Qtt of the grid points, the STO term of the KKT error, and
// OCPSolver::KKTError() = sqrt(dms kkt_error + sto kkt_error) (ocp_solver.cpp:429-431) in place of the dms-only value.
static __global__ void sto_eval_kkt_dev_kernel(StoDevArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  constexpr int NP = RTOC_STO_MAX_EVENTS + 1, MAXP = NP + 1;
  const int N = a.nstages - 1, nev = a.nev;
  double dwell[NP], lt[NP], qtt[NP];
  sto_dwell_times(a, a.ts + (size_t)b * nev, dwell);
  double* c = a.con + (size_t)b * RTOC_STO_CON_STRIDE;
  for (int e = 0; e < nev; ++e) {
    lt[e] = a.cost_lt ? a.cost_lt[(size_t)b * nev + e] : 0.0;
    qtt[e] = a.sto_reg + (a.cost_qtt ? a.cost_qtt[(size_t)b * nev + e] : 0.0);
  }
  double err = 0.0;
  double dual[NP], cond[NP], dos[NP];
  for (int p = 0; p <= nev; ++p) {
    const double slack = c[0 * NP + p];
    dual[p] = c[1 * NP + p];
    const double residual = a.min_dwell[p] - dwell[p] + slack;   // evalConstraint (:186)
    const double cmpl = slack * dual[p] - a.barrier;
    c[2 * NP + p] = residual;
    c[3 * NP + p] = cmpl;
    err += residual * residual + cmpl * cmpl;                     // ConstraintComponentData::KKTError
    cond[p] = (dual[p] * residual - cmpl) / slack;                // computeCondensingCoeffcient
    dos[p] = dual[p] / slack;
  }
  for (int e = 0; e < nev; ++e) {
    lt[e] += dual[e + 1] - dual[e];    // linearizeConstraints: lt += J^T dual
    lt[e] += cond[e + 1] - cond[e];    // condenseSlackAndDual: lt += J^T cond
    qtt[e] += dos[e] + dos[e + 1];     // diag(J^T diag(dual / slack) J)
  }
  // scatter (:105-118)
  double* k = a.kkt + (size_t)b * a.nstages * a.kkt_stride + a.scal_off;
  int ev = 0;
  for (int i = 0; i < N && ev < nev; ++i) {
    const int ty = a.grid[i].type;
    if (ty == RTOC_GRID_IMPACT || ty == RTOC_GRID_LIFT) {
      double* sc = k + (size_t)(ty == RTOC_GRID_IMPACT ? i + 1 : i) * a.kkt_stride;
      sc[RTOC_KKT_SCAL_H] -= lt[ev];
      sc[RTOC_KKT_SCAL_QTT] += qtt[ev];
      ++ev;
    }
  }
  if (a.lt)
    for (int e = 0; e < nev; ++e) a.lt[(size_t)b * nev + e] = lt[e], a.qtt[(size_t)b * nev + e] = qtt[e];
  // per-phase Hamiltonian sums and their differences across STO-enabled events (:120-136)
  double h[MAXP];
  for (int p = 0; p < MAXP; ++p) h[p] = 0.0;
  int phase = 0;
  for (int i = 0; i < N; ++i) {
    const int ty = a.grid[i].type;
    if (ty == RTOC_GRID_IMPACT || ty == RTOC_GRID_LIFT) ++phase;
    if (phase < MAXP) h[phase] += k[(size_t)i * a.kkt_stride + RTOC_KKT_SCAL_H];
  }
  int e2 = 0;
  for (int i = 0; i < N; ++i) {
    const rtoc_grid g = a.grid[i];
    if ((g.type == RTOC_GRID_IMPACT && a.grid[i + 1].sto) || (g.type == RTOC_GRID_LIFT && g.sto)) {
      if (e2 + 1 < MAXP) {
        const double hd = h[e2] - h[e2 + 1];
        err += hd * hd;
      }
      ++e2;
    }
  }
  a.err[b] = err;
  if (a.kkterr) {
    const double e0 = a.kkterr[b];
    a.kkterr[b] = sqrt(e0 * e0 + err);
  }
}

// SwitchingTimeOptimization::computeStepSizes + maxPrimal / maxDualStepSize (:140-178): dts of the events from the direction,
// expandSlackAndDual, fraction-to-boundary, min-ed into the step sizes the stages left in RTOC_BUF_STEP (ocp_solver.cpp:129-132)
static __global__ void sto_step_sizes_kernel(StoDevArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  constexpr int NP = RTOC_STO_MAX_EVENTS + 1;
  const int N = a.nstages - 1, nev = a.nev;
  double dts[NP];
  int ev = 0;
  const double* d = a.dir + (size_t)b * a.nstages * a.dir_stride + a.dts_off;
  for (int i = 0; i < N && ev < nev; ++i) {
    const int ty = a.grid[i].type;
    if (ty == RTOC_GRID_IMPACT || ty == RTOC_GRID_LIFT) dts[ev++] = d[(size_t)i * a.dir_stride];
  }
  double* c = a.con + (size_t)b * RTOC_STO_CON_STRIDE;
  double ps = 1.0, ds = 1.0;
  for (int p = 0; p <= nev; ++p) {
    // dslack = -J dts - residual,  (J dts)_p = -dts_p (p < nev) + dts_{p-1} (p >= 1)
    double Jd = 0.0;
    if (p < nev) Jd -= dts[p];
    if (p >= 1) Jd += dts[p - 1];
    const double slack = c[0 * NP + p], dual = c[1 * NP + p];
    const double dslack = -Jd - c[2 * NP + p];
    const double ddual = -(dual * dslack + c[3 * NP + p]) / slack;   // pdipm::computeDualDirection
    c[4 * NP + p] = dslack;
    c[5 * NP + p] = ddual;
    const double fs = -a.tau * (slack / dslack), fd = -a.tau * (dual / ddual);   // pdipm::fractionToBoundary
    if (fs > 0.0 && fs < 1.0 && fs < ps) ps = fs;
    if (fd > 0.0 && fd < 1.0 && fd < ds) ds = fd;
  }
  double* st = a.steps + 2 * (size_t)b;
  if (ps < st[0]) st[0] = ps;
  if (ds < st[1]) st[1] = ds;
}

// SwitchingTimeOptimization::integrateSolution (:181-206): ts += primal step x dts, slack / dual updates
static __global__ void sto_integrate_kernel(StoDevArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  constexpr int NP = RTOC_STO_MAX_EVENTS + 1;
  const int N = a.nstages - 1, nev = a.nev;
  const double ps = a.steps[2 * (size_t)b], ds = a.steps[2 * (size_t)b + 1];
  const double* d = a.dir + (size_t)b * a.nstages * a.dir_stride + a.dts_off;
  double* ts = a.ts + (size_t)b * nev;
  int ev = 0;
  for (int i = 0; i < N && ev < nev; ++i) {
    const int ty = a.grid[i].type;
    if (ty == RTOC_GRID_IMPACT || ty == RTOC_GRID_LIFT) {
      ts[ev] += ps * d[(size_t)i * a.dir_stride];
      ++ev;
    }
  }
  double* c = a.con + (size_t)b * RTOC_STO_CON_STRIDE;
  for (int p = 0; p <= nev; ++p) {
    c[0 * NP + p] += ps * c[4 * NP + p];
    c[1 * NP + p] += ds * c[5 * NP + p];
  }
}

}  // namespace rtoc
