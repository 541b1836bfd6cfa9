// TEST INFRASTRUCTURE ONLY (oracle/_ref/librtoc_ref.so).  Runs the reference's own SwitchingTimeOptimization::evalKKT
// (src/sto/switching_time_optimization.cpp:79-137) with its real STOCostFunction (src/sto/sto_cost_function.cpp) and
// STOConstraints (minimum dwell times, src/sto/sto_constraints.cpp) over a grid table, so that the part the device takes
// over -- the scatter of lt_ / diag(Qtt_) into h / Qtt and the Hamiltonian-difference term of the KKT error
// (rtoc_sto_eval_kkt, held to oracle/rtoc_oracle_condense.c:orc_sto_eval_kkt) -- is checked against the reference's
// sources, fed with the lt_ / Qtt_ the reference itself computed.  Only the per-grid scalars h and Qtt of the KKT
// containers matter to that function; everything else in them stays at its constructor value.
//
// Not the reference's: the STO cost *component* below (the reference ships only the abstract base,
// include/robotoc/sto/sto_cost_function_component_base.hpp; a quadratic in the event times stands in for a user's).
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "robotoc/core/kkt_matrix.hpp"
#include "robotoc/core/kkt_residual.hpp"
#include "robotoc/core/performance_index.hpp"
#include "robotoc/ocp/ocp.hpp"
#include "robotoc/ocp/time_discretization.hpp"
#include "robotoc/planner/contact_sequence.hpp"
#include "robotoc/sto/sto_constraints.hpp"
#include "robotoc/sto/sto_cost_function.hpp"
// the event gradient / Hessian the reference hands from its cost and constraints to the scatter are private members
#define private public
#include "robotoc/sto/switching_time_optimization.hpp"
#undef private

#include "../../include/rtoc_layout.h"

using namespace robotoc;

namespace {
class QuadraticEventTimeCost final : public STOCostFunctionComponentBase {
 public:
  QuadraticEventTimeCost(const double* w, const double* tref, int nev) : w_(w, w + nev), tref_(tref, tref + nev) {}
  std::vector<double> times(const TimeDiscretization& td) const {
    std::vector<double> ts;
    for (int i = 0; i < td.size() - 1; ++i)
      if (td[i].type == GridType::Impact || td[i].type == GridType::Lift) ts.push_back(td[i].t);
    return ts;
  }
  double evalCost(const TimeDiscretization& td) const override {
    const auto ts = times(td);
    double c = 0.0;
    for (size_t e = 0; e < ts.size(); ++e) c += 0.5 * w_[e] * (ts[e] - tref_[e]) * (ts[e] - tref_[e]);
    return c;
  }
  void evalCostDerivatives(const TimeDiscretization& td, Eigen::VectorXd& lt) const override {
    const auto ts = times(td);
    for (size_t e = 0; e < ts.size(); ++e) lt.coeffRef(e) += w_[e] * (ts[e] - tref_[e]);
  }
  void evalCostHessian(const TimeDiscretization& td, Eigen::MatrixXd& Qtt) const override {
    for (size_t e = 0; e < w_.size(); ++e) Qtt.coeffRef(e, e) += w_[e];
  }

 private:
  std::vector<double> w_, tref_;
};
}  // namespace

extern "C" {
// grid / t: the discretisation ([n] grid points, t their times); h, qtt: [n] in/out, SplitKKTResidual::h and
// SplitKKTMatrix::Qtt of every grid point (after DirectMultipleShooting::evalKKT); min_dwell: [nev + 1];
// cost_w / cost_tref: [nev] or NULL (no cost component); out_lt / out_qtt_diag: [nev] what the reference scattered;
// out_perf: [4] = PerformanceIndex::kkt_error of the STO problem, the dwell-time constraints' own KKTError() (so that
// their difference is the Hamiltonian term), dual_feasibility, cost.  Returns the number of events, < 0 on error.
int ref_sto_eval_kkt(const rtoc_grid* grid, const double* t, int n, double* h, double* qtt, const double* min_dwell, double barrier,
                     double fraction_to_boundary, double sto_reg, const double* cost_w, const double* cost_tref, double* out_lt,
                     double* out_qtt_diag, double* out_perf) {
  std::vector<GridInfo> gi(n);
  int phase = 0;
  for (int i = 0; i < n; ++i) {
    GridInfo& o = gi[i];
    o.type = grid[i].type == RTOC_GRID_IMPACT ? GridType::Impact
                                              : (grid[i].type == RTOC_GRID_LIFT ? GridType::Lift
                                                                                : (grid[i].type == RTOC_GRID_TERMINAL ? GridType::Terminal
                                                                                                                      : GridType::Intermediate));
    // GridInfo::phase counts the impact / lift grids up to and including this one (time_discretization.cpp:70-126)
    if (o.type == GridType::Impact || o.type == GridType::Lift) ++phase;
    o.phase = phase;
    o.t = t[i];
    o.dt = grid[i].dt;
    o.sto = grid[i].sto != 0;
    o.sto_next = grid[i].sto_next != 0;
    o.switching_constraint = grid[i].switching_constraint != 0;
    o.num_grids_in_phase = grid[i].num_grids_in_phase;
    o.stage = i;
  }
  const int nev = phase;
  const TimeDiscretization td(gi);
  const Robot robot(8, 2, std::vector<ContactType>());
  KKTMatrix km(n, SplitKKTMatrix(robot));
  KKTResidual kr(n, SplitKKTResidual(robot));
  for (int i = 0; i < n; ++i) km[i].Qtt = qtt[i], kr[i].h = h[i];
  OCP ocp;
  ocp.sto_cost = std::make_shared<STOCostFunction>();
  if (cost_w && nev > 0) ocp.sto_cost->add("event_times", std::make_shared<QuadraticEventTimeCost>(cost_w, cost_tref, nev));
  ocp.sto_constraints =
      std::make_shared<STOConstraints>(std::vector<double>(min_dwell, min_dwell + nev + 1), barrier, fraction_to_boundary);
  SwitchingTimeOptimization sto(ocp);
  sto.setRegularization(sto_reg);
  sto.initConstraints(td);
  sto.evalKKT(td, km, kr);
  for (int i = 0; i < n; ++i) qtt[i] = km[i].Qtt, h[i] = kr[i].h;
  for (int e = 0; e < nev; ++e) out_lt[e] = sto.lt_.coeff(e), out_qtt_diag[e] = sto.Qtt_.coeff(e, e);
  const PerformanceIndex& p = sto.getEval();
  out_perf[0] = p.kkt_error;
  out_perf[1] = nev > 0 ? sto.constraint_data_.KKTError() : 0.0;
  out_perf[2] = p.dual_feasibility;
  out_perf[3] = p.cost;
  return nev;
}
}  // extern "C"
