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
  double* partial;     // [batch][nstages] squared residual per grid point
  int nstages, batch, nrows, cone_contacts, cone_dim, cone_rows, nc_max, impact_cones;
  int nv, nu, np, nx;
  rtoc_record_layout kl, cl, nl;
};

// One wave per (grid point, instance): the squared residual of that grid point into partial[b][st]; the second kernel
// adds the grid points of an instance in grid order and takes the root -- the same fixed summation order for every
// launch geometry (deterministic), and a single OCP no longer walks its horizon serially (0.2 ms -> a few us).
static __global__ __launch_bounds__(64) void kkt_error_kernel(KktErrArgs a) {
  const int lane = threadIdx.x;
  const int st = blockIdx.x, b = blockIdx.y;
  if (b >= a.batch || st >= a.nstages) return;
  double acc = 0.0;
  auto sq = [&](const double* p, int n) {
    for (int i = lane; i < n; i += 64) {
      const double v = p[i];
      acc += v * v;
    }
  };
  {
    const rtoc_grid g = a.grid[st];
    const size_t rec = (size_t)b * a.nstages + st;
    const double* kr = a.kkt + rec * a.kl.stride;
    const bool terminal = g.type == RTOC_GRID_TERMINAL, impact = g.type == RTOC_GRID_IMPACT;
    sq(kr + a.kl.off[RTOC_KKT_LX], a.nx);
    if (!terminal) {
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
        if (a.cone_contacts > 0 && (!impact || a.impact_cones)) {
          const int row0 = a.nc_max - a.cone_rows * a.cone_contacts, n = a.cone_rows * (g.dimf / a.cone_dim);
          for (int r = lane; r < n; r += 64) {
            const double x = nr[a.nl.off[RTOC_CON_RESIDUAL] + row0 + r], y = nr[a.nl.off[RTOC_CON_CMPL] + row0 + r];
            acc += x * x + y * y;
          }
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (lane == 0) a.partial[(size_t)b * a.nstages + st] = acc;
}

// ---- DirectMultipleShooting::evalOCP's performance index (line search) ---------------------------------------------
// What {Intermediate,Impact,Terminal}Stage::evalOCP accumulate per grid point (src/ocp/intermediate_stage.cpp:52-81,
// impact_stage.cpp:53-76, terminal_stage.cpp:51-66) and LineSearch::lineSearchFilterMethod reads (src/line_search/line_search.cpp:
// 56-83): cost (the value the cost kernel stored), cost_barrier = - barrier sum log(slack) of the active rows
// (pdipm.hxx:195-200), primal_feasibility = l1 norms of the rows' residuals, of [ID; C] (contact_dynamics_data.hpp:195-197) and
// of Fx, P (split_kkt_residual.hxx:109-115); the terminal grid point carries its cost only.  On records linearised and NOT yet
// condensed.  partial: [batch][nstages][2] = (cost + barrier, violation); the reduction adds them in grid order.
struct EvalOcpArgs {
  const double* kkt;
  const double* cdd;
  const double* con;      // may be null
  const double* costval;  // [batch][nstages]
  const rtoc_box_row* rows;
  const rtoc_grid* grid;
  double* partial;
  int nstages, batch, nrows, cone_contacts, cone_dim, cone_rows, nc_max, impact_cones;
  int nv, nx;
  double barrier;
  rtoc_record_layout kl, cl, nl;
};

static __global__ __launch_bounds__(64) void eval_ocp_kernel(EvalOcpArgs a) {
  const int lane = threadIdx.x;
  const int st = blockIdx.x, b = blockIdx.y;
  if (b >= a.batch || st >= a.nstages) return;
  double viol = 0.0, bar = 0.0;
  auto l1 = [&](const double* p, int n) {
    for (int i = lane; i < n; i += 64) viol += fabs(p[i]);
  };
  const rtoc_grid g = a.grid[st];
  const size_t rec = (size_t)b * a.nstages + st;
  const bool terminal = g.type == RTOC_GRID_TERMINAL, impact = g.type == RTOC_GRID_IMPACT;
  if (!terminal) {
    const double* kr = a.kkt + rec * a.kl.stride;
    const double* cr = a.cdd + rec * a.cl.stride;
    l1(kr + a.kl.off[RTOC_KKT_FX], a.nx);
    if (!impact && g.dims > 0) l1(kr + a.kl.off[RTOC_KKT_PRES], g.dims);
    l1(cr + a.cl.off[RTOC_CDD_IDC], a.nv + g.dimf);
    if (a.con) {
      const double* nr = a.con + rec * a.nl.stride;
      if (!impact)
        for (int r = lane; r < a.nrows; r += 64)
          if (g.time_stage >= a.rows[r].level) {
            viol += fabs(nr[a.nl.off[RTOC_CON_RESIDUAL] + r]);
            bar -= log(nr[a.nl.off[RTOC_CON_SLACK] + r]);
          }
      if (a.cone_contacts > 0 && (!impact || a.impact_cones)) {
        const int row0 = a.nc_max - a.cone_rows * a.cone_contacts, n = a.cone_rows * (g.dimf / a.cone_dim);
        for (int r = lane; r < n; r += 64) {
          viol += fabs(nr[a.nl.off[RTOC_CON_RESIDUAL] + row0 + r]);
          bar -= log(nr[a.nl.off[RTOC_CON_SLACK] + row0 + r]);
        }
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) viol += __shfl_xor(viol, off, 64), bar += __shfl_xor(bar, off, 64);
  if (lane == 0) {
    a.partial[2 * rec] = a.costval[rec] + a.barrier * bar;
    a.partial[2 * rec + 1] = viol;
  }
}

// out: [2][batch] = cost + cost_barrier | primal_feasibility
static __global__ void eval_ocp_reduce_kernel(const double* partial, double* out, int nstages, int batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  double c = 0.0, v = 0.0;
  for (int st = 0; st < nstages; ++st) c += partial[2 * ((size_t)b * nstages + st)], v += partial[2 * ((size_t)b * nstages + st) + 1];
  out[b] = c;
  out[batch + b] = v;
}

// ---- LineSearch::lineSearchFilterMethod's backtracking loop, per instance (line_search.cpp:56-83) ----
// alpha: the trial step of every instance; active: 1 while the instance is still backtracking.
struct LsArgs {
  double* steps;        // [batch][2] RTOC_BUF_STEP (in: max primal step; out: accepted step)
  double* trial_steps;  // [batch][2] = (alpha, 0): the trial iterate moves the primal variables and the slacks only
  double* alpha;
  int* active;
  const int* accepted;
  int* nactive;
  int batch;
  double rate, min_step;
};
static __global__ void ls_begin_kernel(LsArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  const double s = a.steps[2 * b];
  a.alpha[b] = s;
  const int on = s > a.min_step ? 1 : 0;   // while (primal_step_size > settings_.min_step_size)
  a.active[b] = on;
  a.trial_steps[2 * b] = on ? s : 0.0;
  a.trial_steps[2 * b + 1] = 0.0;
  if (on) atomicAdd(a.nactive, 1);
}
static __global__ void ls_advance_kernel(LsArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch || !a.active[b]) return;
  if (a.accepted[b]) {            // filter_.augment done by the filter kernel; return primal_step_size
    a.active[b] = 0;
    a.steps[2 * b] = a.alpha[b];
    return;
  }
  const double s = a.alpha[b] * a.rate;   // primal_step_size *= step_size_reduction_rate
  a.alpha[b] = s;
  if (s > a.min_step) {
    a.trial_steps[2 * b] = s;
    atomicAdd(a.nactive, 1);
  } else {                       // the loop ends without an accepted trial: the reduced step is returned as it is
    a.active[b] = 0;
    a.steps[2 * b] = s;
  }
}

// ---- LineSearch::meritBacktrackingLineSearch (line_search.cpp:87-128), per instance ----
// penaltyParam (:120-128): (1 + margin_rate) x the largest SplitSolution::lagrangeMultiplierLinfNorm over the grid
// (split_solution.cpp:126-134: lmd, gmm, beta, nu_passive of a floating base, mu of the active contact dimensions, xi of the active
// switching-constraint rows).  One wave per instance.
struct LsPenaltyArgs {
  const double* sol;
  const rtoc_grid* grid;
  double* penalty;   // [batch]
  int nstages, batch, nv, np;
  rtoc_record_layout sl;
  double margin;
};
static __global__ __launch_bounds__(64) void ls_penalty_kernel(LsPenaltyArgs a) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= a.batch) return;
  double m = 0.0;
  auto linf = [&](const double* p, int n) {
    for (int i = lane; i < n; i += 64) m = fmax(m, fabs(p[i]));
  };
  for (int st = 0; st < a.nstages; ++st) {
    const rtoc_grid g = a.grid[st];
    const double* s = a.sol + ((size_t)b * a.nstages + st) * a.sl.stride;
    linf(s + a.sl.off[RTOC_SOL_LMD], a.nv);
    linf(s + a.sl.off[RTOC_SOL_GMM], a.nv);
    // (the terminal record too: SplitSolution::lagrangeMultiplierLinfNorm of s[N] takes every field, line_search.cpp:120-128 --
    //  beta, nu_passive are zero there unless the caller uploaded something else; the terminal stage has no contact forces, mu of
    //  s[N] is never written by the solver and stays out)
    linf(s + a.sl.off[RTOC_SOL_BETA], a.nv);
    if (a.np > 0) linf(s + a.sl.off[RTOC_SOL_NUP], a.np);
    if (g.type == RTOC_GRID_TERMINAL) continue;
    linf(s + a.sl.off[RTOC_SOL_MU], g.dimf);
    if (g.type != RTOC_GRID_IMPACT && g.switching_constraint) linf(s + a.sl.off[RTOC_SOL_XI], g.dims);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
  if (lane == 0) a.penalty[b] = m * (1.0 + a.margin);
}
// phase 0: every instance gets the trial step eps (the directional derivative's trial, :96-103);
// phase 1: dd = (merit(eps) - merit) / eps;
// phase 2: armijoCondition (:111-117) of the active instances' trial at step alpha -> accepted.
struct LsMeritArgs {
  const double* cur;      // [2][batch] cost + barrier | violation of the iterate
  const double* trial;    // [2][batch] of the trial iterate
  const double* penalty;  // [batch]
  double* dd;             // [batch] directional derivative of the merit function
  double* trial_steps;    // [batch][2]
  const double* alpha;    // [batch]
  const int* active;
  int* accepted;
  int batch, phase;
  double eps, armijo;
};
static __global__ void ls_merit_kernel(LsMeritArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  if (a.phase == 0) {
    a.trial_steps[2 * b] = a.eps;
    a.trial_steps[2 * b + 1] = 0.0;
    return;
  }
  const double merit = a.cur[b] + a.penalty[b] * a.cur[a.batch + b];
  const double merit_trial = a.trial[b] + a.penalty[b] * a.trial[a.batch + b];
  if (a.phase == 1) {
    a.dd[b] = (1.0 / a.eps) * (merit_trial - merit);
    return;
  }
  a.accepted[b] = (a.active[b] && merit_trial < merit + a.armijo * a.alpha[b] * a.dd[b]) ? 1 : 0;
}

static __global__ void kkt_error_reduce_kernel(const double* partial, double* out, int nstages, int batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  double acc = 0.0;
  for (int st = 0; st < nstages; ++st) acc += partial[(size_t)b * nstages + st];
  out[b] = sqrt(acc);
}

// SwitchingTimeOptimization::evalKKT downstream of the STO cost / dwell-time constraints (reference
// src/sto/switching_time_optimization.cpp:105-137): scatter of the per-event gradient lt and of diag(Qtt_) into the grid
// point after an impact / the lift grid point, then the STO term of the KKT error (squared differences of the per-phase
// Hamiltonian sums across STO-enabled events).  One thread per instance: O(#grid points) scalar work.
struct StoArgs {
  double* kkt;
  const rtoc_grid* grid;
  const double* lt;    // [batch][nev]
  const double* qtt;   // [batch][nev]
  double* err;         // [batch] squared STO KKT error
  int nstages, batch, nev, stride, scal_off;
};

static __global__ void sto_eval_kkt_kernel(StoArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.batch) return;
  const int N = a.nstages - 1;
  double* k = a.kkt + (size_t)b * a.nstages * a.stride + a.scal_off;
  int ev = 0;
  for (int i = 0; i < N && ev < a.nev; ++i) {
    const int ty = a.grid[i].type;
    if (ty == RTOC_GRID_IMPACT || ty == RTOC_GRID_LIFT) {
      double* sc = k + (size_t)(ty == RTOC_GRID_IMPACT ? i + 1 : i) * a.stride;
      sc[RTOC_KKT_SCAL_H] -= a.lt[(size_t)b * a.nev + ev];
      sc[RTOC_KKT_SCAL_QTT] += a.qtt[(size_t)b * a.nev + ev];
      ++ev;
    }
  }
  // per-phase Hamiltonian sums h_[grid.phase] (:120-124), GridInfo::phase = number of impact / lift grids so far;
  // then, exactly as the reference walks them (:126-136), ONE running index over the STO-enabled events
  constexpr int MAXP = 32;
  double h[MAXP];
  for (int p = 0; p < MAXP; ++p) h[p] = 0.0;
  int phase = 0;
  for (int i = 0; i < N; ++i) {
    const int ty = a.grid[i].type;
    if (ty == RTOC_GRID_IMPACT || ty == RTOC_GRID_LIFT) ++phase;
    if (phase < MAXP) h[phase] += k[(size_t)i * a.stride + RTOC_KKT_SCAL_H];
  }
  double err = 0.0;
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
}


// ---- LineSearchFilter of every instance (src/line_search/line_search_filter.cpp) -------------------
struct FilterArgs {
  double* filt;      // [batch][CAP][2] (cost, violation)
  int* nfilt;        // [batch]
  const double* cost;
  const double* viol;
  const int* mask;   // may be nullptr
  int* accepted;
  int count, cap;
  double cost_rate, viol_rate;
  int seed_empty;    // 1: only instances whose filter is empty take part (line_search.cpp:58-62: seed it with the current iterate)
};

static __global__ void line_search_filter_kernel(FilterArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.count) return;
  if (a.mask && !a.mask[b]) {
    a.accepted[b] = 0;
    return;
  }
  double* f = a.filt + (size_t)b * a.cap * 2;
  int n = a.nfilt[b];
  if (a.seed_empty && n != 0) {
    a.accepted[b] = 0;
    return;
  }
  const double c = a.cost[b], v = a.viol[b];
  // isAccepted (:26-39): an empty filter accepts; otherwise ANY entry that the pair improves on
  bool ok = n == 0;
  for (int e = 0; e < n && !ok; ++e)
    ok = (c < f[2 * e] - a.cost_rate * f[2 * e + 1]) || (v < (1.0 - a.viol_rate) * f[2 * e + 1]);
  a.accepted[b] = ok ? 1 : 0;
  if (!ok) return;
  // augment (:42-60): erase the entries the new pair dominates, keep the order, append
  int w = 0;
  for (int e = 0; e < n; ++e) {
    const double ce = f[2 * e], ve = f[2 * e + 1];
    if (!(ce <= c && ve <= v)) {
      f[2 * w] = ce;
      f[2 * w + 1] = ve;
      ++w;
    }
  }
  if (w == a.cap) {  // full: drop the oldest
    for (int e = 1; e < w; ++e) {
      f[2 * (e - 1)] = f[2 * e];
      f[2 * (e - 1) + 1] = f[2 * e + 1];
    }
    --w;
  }
  f[2 * w] = c;
  f[2 * w + 1] = v;
  a.nfilt[b] = w + 1;
}

}  // namespace rtoc
