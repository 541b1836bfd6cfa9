// robotoc_hip_device_source.hpp -- a StageDataSource that linearises ON THE DEVICE.
//
// The OCPSolver shell (robotoc_hip_solver.hpp) takes the Pinocchio / cost half of evalKKT from a StageDataSource.  For an
// OCP whose cost is a ConfigurationSpaceCost and that has no inequality rows and no switching constraints, that half
// exists on the device as well (include/rtoc_robot.h: rtoc_contact_eval_kkt = cost + state equation on the manifold +
// rigid-body linearisation): with this source nothing of OCPSolver::updateSolution (src/solver/ocp_solver.cpp:111-145) runs
// on the host -- the solution stays resident in RTOC_BUF_SOL, the host sees KKT errors and, on request, the iterate.
#ifndef ROBOTOC_HIP_DEVICE_SOURCE_HPP_
#define ROBOTOC_HIP_DEVICE_SOURCE_HPP_

#include "../../include/rtoc_robot.h"
#include "robotoc_hip_solver.hpp"

namespace robotoc {

class ConfigurationCostSource : public StageDataSource {
 public:
  // grid: the discretisation (TimeDiscretization of the contact sequence); active[i]: contact mask of grid point i;
  // contact_positions: [grid point][contact][3]; s0: initial guess
  ConfigurationCostSource(const rtoc_robot_model& model, const rtoc_configuration_cost& cost, const std::vector<GridInfo>& grid,
                          const std::vector<unsigned>& active, const std::vector<double>& contact_positions, const Solution& s0)
      : model_(model), cost_(cost), td_(grid), active_(active), cpos_(contact_positions), s0_(s0) {
    if (active.size() != grid.size() || contact_positions.size() != grid.size() * static_cast<size_t>(model.ncontacts) * 3)
      throw std::invalid_argument("[ConfigurationCostSource] one contact mask and ncontacts positions per grid point");
  }
  RobotDims robot() const override {
    const bool ff = model_.type[0] == RTOC_JOINT_FREE_FLYER;
    int dimf = 0;
    for (int k = 0; k < model_.ncontacts; ++k) dimf += model_.contact_type[k] == RTOC_CONTACT_SURFACE ? 6 : 3;
    RobotDims r;
    r.dimv = model_.nv, r.dimu = ff ? model_.nv - 6 : model_.nv, r.dim_passive = ff ? 6 : 0, r.max_dimf = dimf;
    return r;
  }
  int ncMax() const override { return 0; }
  const TimeDiscretization& timeDiscretization() const override { return td_; }
  void configure(rtoc_ctx* ctx) override {
    chk(rtoc_set_robot_model(ctx, &model_), "rtoc_set_robot_model");
    chk(rtoc_set_configuration_cost(ctx, &cost_), "rtoc_set_configuration_cost");
    scheduled_ = false;
  }
  void setInitialState(rtoc_ctx* ctx, const Vec& q, const Vec& v) override {
    std::vector<double> x0(q.size() + v.size());
    for (int i = 0; i < q.size(); ++i) x0[i] = q(i);
    for (int i = 0; i < v.size(); ++i) x0[q.size() + i] = v(i);
    chk(rtoc_set_initial_state(ctx, x0.data(), 1), "rtoc_set_initial_state");
  }
  void linearize(rtoc_ctx* ctx, const Solution&) override {
    if (!scheduled_) {  // needs the grid, which the solver sets after configure()
      chk(rtoc_set_contact_schedule(ctx, active_.data(), cpos_.data(), nullptr), "rtoc_set_contact_schedule");
      scheduled_ = true;
    }
    chk(rtoc_contact_eval_kkt(ctx), "rtoc_contact_eval_kkt");
  }
  // computed on the device with the linearisation (state_equation.cpp:99-109, Fqq_prev_inv correction included)
  bool initialStateDirectionOnDevice() const override { return true; }
  void initialStateDirection(const Vec&, const Vec&, const Solution&, Vec&) const override {}
  void initialSolution(Solution& s) const override {
    for (size_t i = 0; i < s.size() && i < s0_.size(); ++i) s[i] = s0_[i];
  }

 private:
  static void chk(const int rc, const char* what) {
    if (rc != RTOC_OK) throw std::runtime_error(std::string("[ConfigurationCostSource] ") + what + ": " + rtoc_error_string(rc));
  }
  rtoc_robot_model model_;
  rtoc_configuration_cost cost_;
  TimeDiscretization td_;
  std::vector<unsigned> active_;
  std::vector<double> cpos_;
  Solution s0_;
  bool scheduled_ = false;
};

}  // namespace robotoc
#endif
