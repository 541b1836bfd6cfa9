// robotoc_hip_unconstr_solver.hpp -- robotoc::UnconstrOCPSolver over the C ABI, the whole iteration on the device.
//
// Mirrors include/robotoc/solver/unconstr_ocp_solver.hpp:33-230 / src/solver/unconstr_ocp_solver.cpp: same method
// names, argument order, solve() loop (:121-160) and convergence test.  Unlike the OCPSolver shell
// (robotoc_hip_solver.hpp), which takes the Pinocchio / cost half of evalKKT from a StageDataSource, nothing of
// updateSolution (:96-118) is left on the host here: for a fixed-base robot without contacts the cost
// (ConfigurationSpaceCost), the state equation, the rigid-body linearisation, the condensation, the Riccati recursion,
// the expansion and the solution update all run in rtoc_unconstr_update_solution (include/rtoc_robot.h).
// The six joint-limit components of the reference's examples run on the device too (unconstr_constraints.hpp).  What the
// reference's OCP can carry beyond that -- other cost components, the line search -- is not part of this path
// (std::logic_error if the line search is requested).
#ifndef ROBOTOC_HIP_UNCONSTR_SOLVER_HPP_
#define ROBOTOC_HIP_UNCONSTR_SOLVER_HPP_

#include <chrono>
#include <cmath>
#include <string>

#include "../../include/rtoc_robot.h"
#include "robotoc_hip_solver.hpp"

namespace robotoc {

// include/robotoc/ocp/ocp.hpp (unconstrained use): robot, cost, horizon length T, number of intervals N
struct UnconstrOCP {
  rtoc_robot_model robot;
  rtoc_configuration_cost cost;
  double T = 0.0;
  int N = 0;
  int device = 0;
  // the joint-limit components of the reference's Constraints (Joint{Position,Velocity,Torques}{Lower,Upper}Limit): one
  // rtoc_box_row + bound per row, g(z) = sign z - bound <= 0; empty = no inequality constraints
  std::vector<rtoc_box_row> constraint_rows;
  std::vector<double> constraint_bounds;
  double barrier_param = 1.0e-03;  // ConstraintsBase default
};

class UnconstrOCPSolver {
 public:
  explicit UnconstrOCPSolver(const UnconstrOCP& ocp, const SolverOptions& solver_options = SolverOptions())
      : ocp_(ocp), dt_(ocp.T / ocp.N) {
    if (ocp.T <= 0) throw std::out_of_range("[UnconstrOCPSolver] invalid argument: T must be positive!");  // unconstr_ocp_solver.cpp ctor checks
    if (ocp.N <= 0) throw std::out_of_range("[UnconstrOCPSolver] invalid argument: N must be positive!");
    const int nv = ocp.robot.nv;
    dims_.dimv = nv, dims_.dimu = nv, dims_.dim_passive = 0, dims_.max_dimf = 0;
    if (ocp.constraint_rows.size() != ocp.constraint_bounds.size()) throw std::invalid_argument("[UnconstrOCPSolver] one bound per constraint row");
    rtoc_dims d = {nv, nv, 0, 0, 0, static_cast<int>((ocp.constraint_rows.size() + 7) / 8 * 8)};
    rtoc_ctx* c = nullptr;
    check(rtoc_create(&d, ocp.N + 1, 1, ocp.device, &c), "rtoc_create");
    ctx_.reset(c, [](rtoc_ctx* p) { rtoc_destroy(p); });
    check(rtoc_get_layout(c, &L_), "rtoc_get_layout");
    check(rtoc_set_robot_model(c, &ocp.robot), "rtoc_set_robot_model");
    check(rtoc_set_configuration_cost(c, &ocp.cost), "rtoc_set_configuration_cost");
    if (!ocp.constraint_rows.empty())
      check(rtoc_set_constraint_rows(c, ocp.constraint_rows.data(), static_cast<int>(ocp.constraint_rows.size())), "rtoc_set_constraint_rows");
    s_.assign(ocp.N + 1, SplitSolution(dims_));
    setSolverOptions(solver_options);
    discretize(0.0);
  }
  UnconstrOCPSolver() {}
  // value semantics like the reference (unconstr_ocp_solver.hpp:58-73): a copy owns a deep copy of the device context
  // (rtoc_clone: buffers, model, cost, bounds, slack / dual state)
  UnconstrOCPSolver(const UnconstrOCPSolver& o)
      : ocp_(o.ocp_), dt_(o.dt_), dims_(o.dims_), L_(o.L_), s_(o.s_), lqr_policy_(o.lqr_policy_), solver_options_(o.solver_options_),
        solver_statistics_(o.solver_statistics_), kkt_error_(o.kkt_error_), host_solution_valid_(o.host_solution_valid_),
        device_solution_valid_(o.device_solution_valid_) {
    if (o.ctx_) {
      rtoc_ctx* c = nullptr;
      check(rtoc_clone(o.ctx_.get(), &c), "rtoc_clone");
      ctx_.reset(c, [](rtoc_ctx* p) { rtoc_destroy(p); });
    }
  }
  UnconstrOCPSolver& operator=(const UnconstrOCPSolver& o) {
    if (this != &o) {
      UnconstrOCPSolver tmp(o);
      *this = std::move(tmp);
    }
    return *this;
  }
  UnconstrOCPSolver(UnconstrOCPSolver&&) = default;
  UnconstrOCPSolver& operator=(UnconstrOCPSolver&&) = default;

  void setSolverOptions(const SolverOptions& solver_options) {
    // line_search_.set(solver_options.line_search_settings) (unconstr_ocp_solver.cpp:80): the filter method of
    // UnconstrLineSearch::computeStepSize (unconstr_line_search.cpp:37-67) runs on the device inside rtoc_unconstr_update_solution --
    // trial iterates, evalOCP, the filter of every instance
    const LineSearchSettings& ls = solver_options.line_search_settings;
    check(rtoc_set_line_search(ctx_.get(), solver_options.enable_line_search ? 1 : 0, ls.step_size_reduction_rate, ls.min_step_size,
                               ls.filter_cost_reduction_rate, ls.filter_constraint_violation_reduction_rate), "rtoc_set_line_search");
    solver_options_ = solver_options;
    if (!ocp_.constraint_rows.empty())
      check(rtoc_set_constraint_bounds(ctx_.get(), ocp_.constraint_bounds.data(), static_cast<int>(ocp_.constraint_bounds.size()),
                                       ocp_.barrier_param, solver_options.fraction_to_boundary_rule), "rtoc_set_constraint_bounds");
  }
  // discretize (:85-88): N uniform intervals, no events
  void discretize(const double) {
    std::vector<rtoc_grid> g(ocp_.N + 1);
    for (int i = 0; i <= ocp_.N; ++i) {
      g[i] = rtoc_grid{};
      g[i].type = i == ocp_.N ? RTOC_GRID_TERMINAL : RTOC_GRID_INTERMEDIATE;
      g[i].num_grids_in_phase = ocp_.N;
      g[i].time_stage = i;
      g[i].dt = i == ocp_.N ? 0.0 : dt_;
    }
    check(rtoc_set_grid(ctx_.get(), g.data(), ocp_.N + 1), "rtoc_set_grid");
  }
  // initConstraints (:91-93): setSlackAndDual of the joint-limit rows at the current iterate
  void initConstraints() {
    if (ocp_.constraint_rows.empty()) return;
    if (!device_solution_valid_) uploadSolution();
    check(rtoc_unconstr_init_constraints(ctx_.get()), "rtoc_unconstr_init_constraints");
  }

  // updateSolution (:96-118)
  void updateSolution(const double, const Vec& q, const Vec& v) {
    const int nv = dims_.dimv;
    if (q.size() != nv || v.size() != nv) throw std::out_of_range("[UnconstrOCPSolver] invalid argument: q, v must have dimv entries!");
    std::vector<double> x0(2 * nv);
    for (int i = 0; i < nv; ++i) x0[i] = q(i), x0[nv + i] = v(i);
    check(rtoc_set_initial_state(ctx_.get(), x0.data(), 1), "rtoc_set_initial_state");
    if (!device_solution_valid_) uploadSolution();
    double err = 0.0;
    check(rtoc_unconstr_update_solution(ctx_.get(), dt_, &err, 1), "rtoc_unconstr_update_solution");
    kkt_error_ = err;
    host_solution_valid_ = false;
    double steps[2] = {1.0, 1.0};
    if (!ocp_.constraint_rows.empty() || solver_options_.enable_line_search)
      check(rtoc_download(ctx_.get(), RTOC_BUF_STEP, 0, steps, 2), "rtoc_download(RTOC_BUF_STEP)");
    solver_statistics_.primal_step_size.push_back(steps[0]);
    solver_statistics_.dual_step_size.push_back(steps[1]);
  }

  // solve (:121-160)
  void solve(const double t, const Vec& q, const Vec& v, const bool init_solver = true) {
    const auto t0 = std::chrono::high_resolution_clock::now();
    if (init_solver) {
      initConstraints();
      if (solver_options_.enable_line_search) check(rtoc_line_search_clear(ctx_.get()), "rtoc_line_search_clear");   // line_search_.clearHistory() (:135)
    }
    solver_statistics_.clear();
    for (int iter = 0; iter < solver_options_.max_iter; ++iter) {
      updateSolution(t, q, v);
      solver_statistics_.performance_index.push_back(kkt_error_ * kkt_error_);
      solver_statistics_.iter = iter + 1;
      if (KKTError() < solver_options_.kkt_tol) {
        solver_statistics_.convergence = true;
        break;
      }
    }
    if (solver_options_.enable_benchmark)
      solver_statistics_.cpu_time = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
  }

  const SolverStatistics& getSolverStatistics() const { return solver_statistics_; }
  const SplitSolution& getSolution(const int stage) {
    syncSolution();
    return s_.at(stage);
  }
  std::vector<Vec> getSolution(const std::string& name) {  // :176-214
    syncSolution();
    std::vector<Vec> out;
    for (int i = 0; i <= ocp_.N; ++i) {
      if (name == "q") out.push_back(s_[i].q);
      else if (name == "v") out.push_back(s_[i].v);
      else if (i < ocp_.N && name == "a") out.push_back(s_[i].a);
      else if (i < ocp_.N && name == "u") out.push_back(s_[i].u);
    }
    return out;
  }
  // getLQRPolicy (:217-219): K (dimu x dimx, acceleration as the Riccati control), k
  const std::vector<LQRPolicy>& getLQRPolicy() {
    lqr_policy_.assign(ocp_.N, LQRPolicy(dims_));
    std::vector<double> b(static_cast<size_t>(ocp_.N + 1) * L_.ric.stride);
    check(rtoc_download(ctx_.get(), RTOC_BUF_RIC, 0, b.data(), b.size()), "rtoc_download(RTOC_BUF_RIC)");
    const int nv = dims_.dimv;
    for (int i = 0; i < ocp_.N; ++i) {
      const double* r = &b[static_cast<size_t>(i) * L_.ric.stride];
      std::copy(r + L_.ric.off[RTOC_RIC_K], r + L_.ric.off[RTOC_RIC_K] + 2 * nv * nv, lqr_policy_[i].Kt.data());
      std::copy(r + L_.ric.off[RTOC_RIC_KV], r + L_.ric.off[RTOC_RIC_KV] + nv, lqr_policy_[i].k.data());
    }
    return lqr_policy_;
  }
  void setSolution(const std::string& name, const Vec& value) {  // :222-256
    syncSolution();
    for (int i = 0; i <= ocp_.N; ++i) {
      Vec* dst = name == "q" ? &s_[i].q : name == "v" ? &s_[i].v : (i < ocp_.N && name == "a") ? &s_[i].a : (i < ocp_.N && name == "u") ? &s_[i].u : nullptr;
      if (name != "q" && name != "v" && name != "a" && name != "u")
        throw std::invalid_argument("[UnconstrOCPSolver] invalid arugment: name must be q, v, a, or u!");
      if (!dst) continue;
      if (value.size() != dst->size()) throw std::out_of_range("[UnconstrOCPSolver] invalid argument: value has the wrong size!");
      *dst = value;
    }
    device_solution_valid_ = false;
  }
  // KKTError(t, q, v) (:259-265): linearise at the current iterate and evaluate, without updating it
  double KKTError(const double, const Vec& q, const Vec& v) {
    const int nv = dims_.dimv;
    std::vector<double> x0(2 * nv);
    for (int i = 0; i < nv; ++i) x0[i] = q(i), x0[nv + i] = v(i);
    check(rtoc_set_initial_state(ctx_.get(), x0.data(), 1), "rtoc_set_initial_state");
    if (!device_solution_valid_) uploadSolution();
    check(rtoc_unconstr_eval_kkt(ctx_.get(), dt_), "rtoc_unconstr_eval_kkt");
    check(rtoc_kkt_error(ctx_.get(), &kkt_error_, 1), "rtoc_kkt_error");
    return kkt_error_;
  }
  double KKTError() const { return kkt_error_; }  // :268-270, of the iterate updateSolution linearised at
  double T() const { return ocp_.T; }
  int N() const { return ocp_.N; }
  rtoc_ctx* context() const { return ctx_.get(); }

 private:
  static void check(const int rc, const char* what) {
    if (rc != RTOC_OK) throw std::runtime_error(std::string("[UnconstrOCPSolver] ") + what + ": " + rtoc_error_string(rc));
  }
  void uploadSolution() {
    std::vector<double> b(static_cast<size_t>(ocp_.N + 1) * L_.sol.stride, 0.0);
    for (int i = 0; i <= ocp_.N; ++i) StageDumpSource::packSolution(L_, s_[i], &b[static_cast<size_t>(i) * L_.sol.stride]);
    check(rtoc_upload(ctx_.get(), RTOC_BUF_SOL, 0, b.data(), b.size()), "rtoc_upload(RTOC_BUF_SOL)");
    device_solution_valid_ = host_solution_valid_ = true;
  }
  void syncSolution() {
    if (host_solution_valid_) return;
    std::vector<double> b(static_cast<size_t>(ocp_.N + 1) * L_.sol.stride);
    check(rtoc_download(ctx_.get(), RTOC_BUF_SOL, 0, b.data(), b.size()), "rtoc_download(RTOC_BUF_SOL)");
    for (int i = 0; i <= ocp_.N; ++i) StageDumpSource::unpackSolution(L_, &b[static_cast<size_t>(i) * L_.sol.stride], s_[i]);
    host_solution_valid_ = true;
  }

  UnconstrOCP ocp_;
  double dt_ = 0.0;
  RobotDims dims_;
  std::shared_ptr<rtoc_ctx> ctx_;
  rtoc_layout L_;
  Solution s_;
  std::vector<LQRPolicy> lqr_policy_;
  SolverOptions solver_options_;
  SolverStatistics solver_statistics_;
  double kkt_error_ = 0.0;
  bool host_solution_valid_ = true, device_solution_valid_ = false;
};

}  // namespace robotoc
#endif
